// zr_camera.hip — the camera pass (deferred-scene pass, ZE:3417-3480) as triangle-level binning: k_hiz_build, k_select, k_geom<HIZ>,
// k_scan_tri, k_index, k_tile<MODE, LAST>, k_sky_tiles.  See zr_dev.h for the map of the kernel files.
#include "zr_dev.h"
#include "zr_raster.h"

// Hi-Z pyramid of the key buffer: level 0 = max depth per 8x8 pixel block (1.0 where a pixel is still empty), each further
// level the max over 2x2 blocks of the previous one.  One workgroup per 64x64 pixel region builds all four levels in LDS.
// regions[]: the 64 x 64 regions that hold tiles this context owns (x | y << 16): a super-tile is whole regions, and the pyramid's texels
// over other ranks' regions stay 0 from zr_create on ("hidden": nothing is drawn there) - a rank of eight builds an eighth.
__global__ __launch_bounds__(256) void k_hiz_build(const unsigned long long* __restrict__ vis64, uint32_t W, uint32_t H, ZrHiz Z, const uint32_t* __restrict__ regions)
{
    __shared__ float l0[8][8];
    const uint32_t rx = regions[blockIdx.x] & 0xFFFFu, ry = regions[blockIdx.x] >> 16, tid = threadIdx.x;
    // 256 threads: thread t handles pixel-block (t & 7, (t >> 3) & 7) quarter (t >> 6): 4 threads per 8x8 block, a 4x4 sub-block each
    const uint32_t bx = tid & 7u, by = (tid >> 3) & 7u, q = tid >> 6;
    float m = 0.0f;
    bool any = false;
    for (uint32_t i = 0; i < 16u; ++i) {
        const uint32_t px = rx * 64u + bx * 8u + (q & 1u) * 4u + (i & 3u), py = ry * 64u + by * 8u + (q >> 1) * 4u + (i >> 2);
        if (px < W && py < H) {
            // a tile of another rank never receives a fragment here: it must not keep the meshlets that straddle it alive
            const bool mine = Z.tile_world <= 1u || tile_owner(px / TILE, py / TILE, Z.tile_world) == Z.tile_rank;
            if (mine) m = __builtin_fmaxf(m, zr_u2f((uint32_t)(vis64[(size_t)py * W + px] >> 32)));
            any = true;
        }
    }
    if (!any) m = 0.0f;
    {   // the finest level: this thread's 4 x 4 pixels
        const uint32_t fx = rx * 16u + bx * 2u + (q & 1u), fy = ry * 16u + by * 2u + (q >> 1);
        if (fx < Z.fw && fy < Z.fh) Z.fine[(size_t)fy * Z.fw + fx] = m;
    }
    // combine the 4 quarters (lanes tid, tid+64, tid+128, tid+192) through LDS
    __shared__ float part[4][64];
    part[q][tid & 63u] = m;
    __syncthreads();
    if (tid < 64u) {
        const float v = __builtin_fmaxf(__builtin_fmaxf(part[0][tid], part[1][tid]), __builtin_fmaxf(part[2][tid], part[3][tid]));
        l0[by][bx] = v;
        const uint32_t gx = rx * 8u + bx, gy = ry * 8u + by;
        if (gx < Z.hw[0] && gy < Z.hh[0]) Z.lvl[0][(size_t)gy * Z.hw[0] + gx] = v;
    }
    __syncthreads();
    if (tid < 16u) {            // level 1: 4x4 per region
        const uint32_t x = tid & 3u, y = tid >> 2;
        const float v = __builtin_fmaxf(__builtin_fmaxf(l0[2 * y][2 * x], l0[2 * y][2 * x + 1]), __builtin_fmaxf(l0[2 * y + 1][2 * x], l0[2 * y + 1][2 * x + 1]));
        const uint32_t gx = rx * 4u + x, gy = ry * 4u + y;
        if (gx < Z.hw[1] && gy < Z.hh[1]) Z.lvl[1][(size_t)gy * Z.hw[1] + gx] = v;
    }
    if (tid >= 64u && tid < 68u) {   // level 2: 2x2 per region
        const uint32_t x = (tid - 64u) & 1u, y = (tid - 64u) >> 1;
        float v = 0.0f;
        for (uint32_t j = 0; j < 4u; ++j) for (uint32_t i = 0; i < 4u; ++i) v = __builtin_fmaxf(v, l0[4 * y + j][4 * x + i]);
        const uint32_t gx = rx * 2u + x, gy = ry * 2u + y;
        if (gx < Z.hw[2] && gy < Z.hh[2]) Z.lvl[2][(size_t)gy * Z.hw[2] + gx] = v;
    }
    if (tid == 128u) {               // level 3: the region
        float v = 0.0f;
        for (uint32_t j = 0; j < 8u; ++j) for (uint32_t i = 0; i < 8u; ++i) v = __builtin_fmaxf(v, l0[j][i]);
        if (rx < Z.hw[3] && ry < Z.hh[3]) Z.lvl[3][(size_t)ry * Z.hw[3] + rx] = v;
    }
}

// ------------------------------------------------------------------------------------------------ triangle-binned camera pass
//
// A meshlet-binned rasteriser re-transforms a meshlet's vertices and re-tests all of its triangles in every tile the meshlet touches
// (2.5 on average in the camera pass) and walks the survivors in whatever mix of sizes the queue hands a wave.
// Here a meshlet is processed ONCE: k_geom transforms its vertices, applies the exact per-triangle tests (facing, degenerate, no
// pixel centre, Hi-Z in round 2) and emits one 32-byte record per (triangle, owned tile) - vertices relative to the tile, three depths,
// the primitive id - plus its tile id; k_scan_tri lays the tiles' ranges out, k_index writes the records' positions in tile order (an
// index list), and k_tile's lanes gather them and do nothing but edge setup + walk on live triangles.  Same arithmetic, same keys as the meshlet-binned path
// (kept in -DZR_DIAG builds for A/B): the frame is the same bit for bit.

// Which meshlet-instances does this round draw?  (The split of the two-pass occlusion culling, as k_bin_count makes it.)
// Compacted per workgroup: one global atomic per 1024 work items (atomics on one address run at ~10 ns apiece on this part).
#define ZR_SELECT_THREADS 256
__global__ __launch_bounds__(256) void k_select(ZrPass P, const ZrObject* __restrict__ objs, const uint32_t* __restrict__ work,
                                                const uint32_t* __restrict__ rects, ZrHiz Z, ZrBinEntry* __restrict__ sel,
                                                ZrDevStats* __restrict__ stats, int slot)
{
    // 1024 work items per workgroup of 256 threads: beside the other lane's kernels a small workgroup finds room where 1024 threads
    // wait for a whole CU (this kernel sits on the camera pipeline's critical path), and the compaction still costs one atomic per 1024
    __shared__ uint32_t wcount[16], wbase[16], nocc;
    const uint32_t n_vis = P.use_worklist ? stats->n_vis_work[1] : P.n_work;
    if (blockIdx.x * 1024u >= n_vis) return;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) nocc = 0;
    bool take[4]; uint32_t w[4]; unsigned long long m[4];
    uint32_t n_occ = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t k = blockIdx.x * 1024u + (uint32_t)j * 256u + threadIdx.x;
        take[j] = false; w[j] = 0;
        bool occluded = false;
        if (k < n_vis) {
            w[j] = P.use_worklist ? work[k] : k;
            take[j] = rects[k] != ZR_RECT_CULLED;
            if (take[j] && Z.phase) {
                const bool was_visible = Z.vis_prev[w[j]] == (uint8_t)Z.vis_stamp;
                if (Z.phase == 1u) take[j] = was_visible;
                else if (was_visible) take[j] = false;
                else if (hiz_occluded(Z, Z.pxrect[k], Z.zmin[k])) { take[j] = false; occluded = true; }
            }
        }
        m[j] = __ballot(take[j]);
        if (lane == 0) wcount[j * 4 + (int)wv] = (uint32_t)__popcll(m[j]);
        n_occ += (uint32_t)__popcll(__ballot(occluded));
    }
    __syncthreads();
    if (lane == 0 && n_occ) atomicAdd(&nocc, n_occ);
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int i = 0; i < 16; ++i) { wbase[i] = tot; tot += wcount[i]; }
        const uint32_t base = tot ? atomicAdd(&stats->n_sel[slot], tot) : 0u;
        for (int i = 0; i < 16; ++i) wbase[i] += base;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!take[j]) continue;
        // the work id is decoded here, lane-parallel: k_geom's wave starts every load of the meshlet from this one record
        const ZrObject* __restrict__ O = objs + find_object_work(objs, (int)P.n_objects, w[j]);
        const uint32_t local = w[j] - O->work_base;
        const uint32_t inst_i = local / O->n_meshlets, mi = local - inst_i * O->n_meshlets;
        const XkMeshlet* __restrict__ ml = O->meshlets + mi;
        ZrBinEntry be;
        const uint4 mh = ld_global((const uint4*)ml);            // VertexOffset, VertexCount, TriangleOffset, TriangleCount
        be.mpos = O->mpos + mh.x; be.mtri = O->mtri + ld_global(&ml->BindlessContext); be.inst = O->inst + inst_i;
        be.counts = mh.y | mh.w << 8 | (O->instanced ? 1u << 16 : 0u);
        be.prim_base = O->prim_base + inst_i * O->n_tris;
        sel[wbase[j * 4 + (int)wv] + (uint32_t)__popcll(m[j] & ((1ull << lane) - 1ull))] = be;
    }
    if (threadIdx.x == 0 && nocc) atomicAdd(&stats->hiz_culled, nocc);
}

// ---- triangle records ----
// 32 bytes per (triangle, tile): three snapped vertices RELATIVE TO THE TILE'S ORIGIN as int16 pairs (a small triangle - every edge under
// 64 px - that reaches the tile has its vertices within [-16384, 24576] sub-pixel units of it) with their depth bits, and the primitive id:
//   plane A: (X0 | Y0 << 16, z0, X1 | Y1 << 16, z1)      plane B: (X2 | Y2 << 16, z2, prim, 0)
// plus the tile id in a separate dword stream (k_index reads 4 bytes per record, not the record, to find where it goes).  Records live
// in chunks of ZR_TPOOL_CHUNK, structure-of-arrays inside a chunk (every store and load of a wave is one contiguous run); chunk_fill[c] =
// records in chunk c.  Both rounds of a frame use the chunks from 0: round 1's records have been moved and rasterised by then.
__device__ __forceinline__ uint32_t pack_xy(int X, int Y) { return ((uint32_t)X & 0xFFFFu) | ((uint32_t)Y << 16); }
struct RecWriter {                 // wave-uniform state of one record stream of a wave
    uint32_t cur, fill;            // the chunk being filled (>= n_chunks: the pool ran dry) and its fill
};
// room for `n` more records (wave-uniform): closes the chunk and takes one from the pool when it would overflow; false: pool dry
__device__ __forceinline__ bool rec_reserve(RecWriter& W, uint32_t n, uint32_t lane, const ZrTriBins& B, ZrDevStats* __restrict__ stats, int slot)
{
    if (W.cur < B.n_chunks && W.fill + n > ZR_TPOOL_CHUNK) {
        uint32_t nx_c = 0;
        if (lane == 0) { B.chunk_fill[W.cur] = W.fill; nx_c = B.n_waves + atomicAdd(&stats->pool_next[slot], 1u); }
        W.cur = min((uint32_t)__builtin_amdgcn_readfirstlane((int)nx_c), B.n_chunks);
        W.fill = 0;
    }
    if (W.cur >= B.n_chunks) { if (lane == 0) { stats->overflow = 1u; stats->overflow_sticky = 1u; } return false; }
    return true;
}
__device__ __forceinline__ void rec_store(const ZrTriBins& B, uint32_t pos, const int4& r0, const int4& r1, const int4& r2, uint32_t prim, uint32_t tile, int tx, int ty)
{
    const int ox = tx * (TILE * 256), oy = ty * (TILE * 256);
    B.recA[pos] = make_uint4(pack_xy(r0.x - ox, r0.y - oy), (uint32_t)r0.z, pack_xy(r1.x - ox, r1.y - oy), (uint32_t)r1.z);
    B.recB[pos] = make_uint4(pack_xy(r2.x - ox, r2.y - oy), (uint32_t)r2.z, prim, 0u);
    B.rtile[pos] = tile;
}
struct RecTri { SV a, b, c; uint32_t prim; };
__device__ __forceinline__ RecTri rec_load(const uint4 qa, const uint4 qb)
{
    RecTri t;
    t.a.X = (int)(short)(qa.x & 0xFFFFu); t.a.Y = (int)qa.x >> 16; t.a.z = zr_u2f(qa.y); t.a.rw = 0.0f;
    t.b.X = (int)(short)(qa.z & 0xFFFFu); t.b.Y = (int)qa.z >> 16; t.b.z = zr_u2f(qa.w); t.b.rw = 0.0f;
    t.c.X = (int)(short)(qb.x & 0xFFFFu); t.c.Y = (int)qb.x >> 16; t.c.z = zr_u2f(qb.y); t.c.rw = 0.0f;
    t.prim = qb.z;
    return t;
}

// Max depth already in the key buffer (per the pyramid Z) over the pixel blocks a snapped box touches: 4 x 4 blocks for a box under 16
// pixels, else 8 x 8 (blocks of other ranks' tiles hold 0).  A triangle whose least vertex depth lies behind it cannot win a pixel.
__device__ __forceinline__ float pyramid_max(const ZrHiz& Z, int x0, int y0, int x1, int y1)
{
    float h = 0.0f;
    if (max(x1 - x0, y1 - y0) < 16) {
        for (int by = y0 >> 2; by <= (y1 >> 2); ++by)
            for (int bx = x0 >> 2; bx <= (x1 >> 2); ++bx) h = __builtin_fmaxf(h, Z.fine[(size_t)by * Z.fw + (size_t)bx]);
    } else {
        for (int by = y0 >> 3; by <= (y1 >> 3); ++by)
            for (int bx = x0 >> 3; bx <= (x1 >> 3); ++bx) h = __builtin_fmaxf(h, Z.lvl[0][(size_t)by * Z.hw[0] + (size_t)bx]);
    }
    return h;
}

// One wave per selected meshlet-instance: vertices -> LDS, then a lane per triangle.
// Triangles that pass the exact tests (facing, a pixel centre of the target inside the snapped box) become records, one per (triangle,
// owned tile), in the wave's own chunks (wave k starts in chunk k and takes further ones from a pool: one atomic per ZR_TPOOL_CHUNK
// records; a round is ONE launch whatever the scene's size).
// ROUND 2 (HIZ = true): a meshlet whose snapped vertex box lies behind this frame's pyramid is dropped after the vertex phase, and every
// triangle is tested once more by itself against the 4 x 4-pixel level (the meshlet's blocks stay in LDS for that).
template <bool HIZ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_geom(ZrPass P, const ZrBinEntry* __restrict__ sel, ZrHiz Z, ZrTriBins B, uint32_t* __restrict__ tile_count,
            ZrDevStats* __restrict__ stats, int slot, unsigned long long* __restrict__ vis64)
{
    __shared__ int4 vstage[4][WAVE];
    __shared__ float hzs[4][WAVE];
    const uint32_t lane = threadIdx.x & 63u, wv = wave_uniform(threadIdx.x >> 6);
    const uint32_t n = stats->n_sel[slot];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t wave_id = blockIdx.x * 4u + wv, n_waves = gridDim.x * 4u;
    constexpr bool pyramid = HIZ;
    if (HIZ && wave_id == 0u && lane == 0u) stats->survivors[slot] = n;      // (k_tile_slow takes the meshlets dropped behind the pyramid off)
    RecWriter Wd;                                        // the wave's record stream
    Wd.cur = wave_id; Wd.fill = 0;
    uint32_t culled = 0;
    for (uint32_t i = wave_id; i < n; i += n_waves) {
        const uint4* __restrict__ rec = (const uint4*)(sel + i);
        const uint4 e0 = rec[0], e1 = rec[1];
        const float4* __restrict__ mp = (const float4*)(((unsigned long long)wave_uniform(e0.y) << 32) | wave_uniform(e0.x));
        const uint2* __restrict__ tw = (const uint2*)(((unsigned long long)wave_uniform(e0.w) << 32) | wave_uniform(e0.z));
        const ZrInstance* __restrict__ ip = (const ZrInstance*)(((unsigned long long)wave_uniform(e1.y) << 32) | wave_uniform(e1.x));
        const uint32_t counts = wave_uniform(e1.z), pbase = wave_uniform(e1.w);
        const uint32_t vcount = counts & 255u, tcount = (counts >> 8) & 255u;
        const bool instanced = (counts >> 16) & 1u;
        uint2 tri_w[2];
        tri_w[0] = lane < tcount ? ld_global(tw + lane) : make_uint2(0u, 0u);
        tri_w[1] = lane + WAVE < tcount ? ld_global(tw + lane + WAVE) : make_uint2(0u, 0u);
        const float4 pp = lane < vcount ? ld_global(mp + lane) : make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        const ZrInstance I = ld_record(ip);

        lds_fence();   // this wave's previous readers are done with its staging area
        bool flagged;
        int lo2 = 0x7FFF7FFF, hi2 = (int)0x80008000, zb = 0x7FFFFFFF;      // this lane's share of the meshlet's pixel box / least depth
        {
            const zf4 c = zr_mat4_point(P.PVM, vs_position(zr3(pp.x, pp.y, pp.z), I, instanced));
            const float FM = 3.402823466e38f, gb = ZR_GUARD * c.w;
            const bool fin = __builtin_fabsf(c.x) <= FM && __builtin_fabsf(c.y) <= FM && __builtin_fabsf(c.z) <= FM && __builtin_fabsf(c.w) <= FM;
            const bool odd = !fin || c.x < -c.w || c.x > c.w || c.y < -c.w || c.y > c.w || c.z < 0.0f || c.z > c.w ||
                             !(c.w > 0.0f) || __builtin_fabsf(c.x) > gb || __builtin_fabsf(c.y) > gb;
            flagged = __ballot(lane < vcount && odd) != 0ull;
            if (lane < vcount) {
                const uint32_t f = flagged ? vertex_flags(c) : 0u;
                SV sv; sv.X = 0; sv.Y = 0; sv.z = 0.0f; sv.rw = 0.0f;
                if (!(f & 129u)) sv = project(c, P.hw, P.hh);
                vstage[wv][lane] = make_int4(sv.X, sv.Y, (int)zr_f2u(sv.z), (int)f);      // snapped x, y (absolute), depth, clip flags
                if (pyramid && !flagged) {
                    lo2 = (clamp16((sv.X - 128 + 255) >> 8) & 0xFFFF) | (clamp16((sv.Y - 128 + 255) >> 8) << 16);
                    hi2 = (clamp16((sv.X - 128) >> 8) & 0xFFFF) | (clamp16((sv.Y - 128) >> 8) << 16);
                    zb = (int)zr_f2u(sv.z + 0.0f);
                }
            }
        }
        bool hz_local = false;           // wave-uniform: hzs[wv] holds this meshlet's 4 x 4-pixel blocks, (hz_x0, hz_y0) the first one
        int hz_x0 = 0, hz_y0 = 0;
        if (pyramid && !flagged) {      // every vertex inside the frustum: the box of the snapped vertices bounds every fragment
            const int lo = wave_pkmin16(lo2), hi = wave_pkmax16(hi2);
            const int px0 = max(0, (int)(short)(lo & 0xFFFF)), py0 = max(0, lo >> 16);
            const int px1 = min((int)P.W - 1, (int)(short)(hi & 0xFFFF)), py1 = min((int)P.H - 1, hi >> 16);
            bool gone = px0 > px1 || py0 > py1;                  // no pixel centre inside
            if (!gone) {
                const uint32_t fx0 = (uint32_t)px0 >> 2, fy0 = (uint32_t)py0 >> 2, fx1 = (uint32_t)px1 >> 2, fy1 = (uint32_t)py1 >> 2;
                if (fx1 - fx0 < 8u && fy1 - fy0 < 8u) {
                    // a box of up to 32 x 32 pixels: its <= 8 x 8 blocks of the 4 x 4 level, a lane each - one load, one wave reduction;
                    // the values stay in LDS for the per-triangle tests below (no dependent global load per triangle)
                    const uint32_t x = fx0 + (lane & 7u), y = fy0 + (lane >> 3);
                    const float v = (x <= fx1 && y <= fy1) ? Z.fine[(size_t)y * Z.fw + x] : 0.0f;
                    hzs[wv][lane] = v;
                    hz_x0 = (int)fx0; hz_y0 = (int)fy0; hz_local = true;
                    if (HIZ) { const float zm = zr_u2f((uint32_t)wave_min(zb)); gone = zm >= 0.0f && zm > wave_fmax(v); }
                } else if (HIZ) {
                    const float zm = zr_u2f((uint32_t)wave_min(zb));
                    gone = hiz_occluded(Z, make_uint2((uint32_t)px0 | (uint32_t)py0 << 16, (uint32_t)px1 | (uint32_t)py1 << 16), zm);
                }
            }
            // (round 1 keeps a meshlet whose box holds no pixel centre: its triangles fail their own test below, nothing is deferred)
            if (HIZ && gone) { ++culled; continue; }
        }
        lds_fence();

#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const uint32_t t0 = (uint32_t)round * WAVE;
            if (t0 >= tcount) break;
            const uint32_t t = t0 + lane;
            int4 r0 = make_int4(0, 0, 0, 0), r1 = r0, r2 = r0;
            const uint32_t prim = pbase + tri_w[round].y;
            bool alive = false, is_slow = false, hidden = false;
            int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
            uint32_t i0 = 0, i1 = 0, i2 = 0;
            uint32_t slow_rect = 0xFFFF0000u;         // the tiles a slow triangle can touch: (0, 0)-(255, 255) = every tile, or an unclipped one's snapped box
            if (t < tcount) {
                i0 = tri_w[round].x & 255u; i1 = (tri_w[round].x >> 8) & 255u; i2 = (tri_w[round].x >> 16) & 255u;
                r0 = vstage[wv][i0]; r1 = vstage[wv][i1]; r2 = vstage[wv][i2];
                int cls = flagged ? classify((uint32_t)r0.w, (uint32_t)r1.w, (uint32_t)r2.w) : 1;
                if (cls == 1 && !tri_is_small(r0.x, r0.y, r1.x, r1.y, r2.x, r2.y)) {
                    // a big triangle goes to the list every owned tile tries - unless it faces away or its snapped box holds no pixel
                    // centre of the target (raster_sub's own first tests, in 64 bits here: big coordinates)
                    const long long A = (long long)(r1.x - r0.x) * (r2.y - r0.y) - (long long)(r2.x - r0.x) * (r1.y - r0.y);
                    const int bx0 = max((imin3(r0.x, r1.x, r2.x) - 128 + 255) >> 8, 0), bx1 = min((imax3(r0.x, r1.x, r2.x) - 128) >> 8, (int)P.W - 1);
                    const int by0 = max((imin3(r0.y, r1.y, r2.y) - 128 + 255) >> 8, 0), by1 = min((imax3(r0.y, r1.y, r2.y) - 128) >> 8, (int)P.H - 1);
                    cls = (A < 0 && bx0 <= bx1 && by0 <= by1) ? 2 : 0;
                    if (cls == 2) slow_rect = (uint32_t)(bx0 / TILE) | (uint32_t)(by0 / TILE) << 8 | (uint32_t)(bx1 / TILE) << 16 | (uint32_t)(by1 / TILE) << 24;
                }
                if (cls == 2) is_slow = true;
                else if (cls == 1) {
                    // the tests of tri_prefilter / raster_sub that do not depend on the tile: facing + degenerate (edges below 2^14:
                    // the area fits 32 bits), pixel centres of the TARGET inside the snapped box, then the pyramid
                    const int A = (r1.x - r0.x) * (r2.y - r0.y) - (r2.x - r0.x) * (r1.y - r0.y);
                    x0 = max((imin3(r0.x, r1.x, r2.x) - 128 + 255) >> 8, 0); x1 = min((imax3(r0.x, r1.x, r2.x) - 128) >> 8, (int)P.W - 1);
                    y0 = max((imin3(r0.y, r1.y, r2.y) - 128 + 255) >> 8, 0); y1 = min((imax3(r0.y, r1.y, r2.y) - 128) >> 8, (int)P.H - 1);
                    alive = A < 0 && x0 <= x1 && y0 <= y1;
                    if (pyramid && alive && !flagged) {
                        const float tz = __builtin_fminf(__builtin_fminf(zr_u2f((uint32_t)r0.z), zr_u2f((uint32_t)r1.z)), zr_u2f((uint32_t)r2.z));
                        float h = 0.0f;
                        if (hz_local) {       // (a triangle's box lies inside its meshlet's)
                            for (int by = (y0 >> 2) - hz_y0; by <= (y1 >> 2) - hz_y0; ++by)
                                for (int bx = (x0 >> 2) - hz_x0; bx <= (x1 >> 2) - hz_x0; ++bx) h = __builtin_fmaxf(h, hzs[wv][by * 8 + bx]);
                        } else h = pyramid_max(Z, x0, y0, x1, y1);
                        hidden = tz > h;
                    } else if (HIZ && alive) {   // (a flagged meshlet's unclipped triangle: vertices in front of the near plane, depths valid)
                        const float tz = __builtin_fminf(__builtin_fminf(zr_u2f((uint32_t)r0.z), zr_u2f((uint32_t)r1.z)), zr_u2f((uint32_t)r2.z));
                        hidden = tz > pyramid_max(Z, x0, y0, x1, y1);
                    }
                }
            }
            // ---- slow triangles: the three clip-space vertices go to the list every owned tile tries
            const unsigned long long ms = __ballot(is_slow);
            if (ms) {
                uint32_t base = 0;
                if (lane == (uint32_t)__builtin_ctzll(ms)) base = atomicAdd(&stats->n_slow[slot], (uint32_t)__popcll(ms));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(ms));
                if (is_slow) {
                    const uint32_t pos_r = base + (uint32_t)__popcll(ms & lt), pos = pos_r + (slot == 2 ? B.slow_cap / 2u : 0u);
                    if (pos_r < B.slow_cap / 2u) {
                        const uint32_t li[3] = { i0, i1, i2 };
                        for (int k = 0; k < 3; ++k) {
                            const float4 pk = ld_global(mp + li[k]);
                            const zf4 cc = zr_mat4_point(P.PVM, vs_position(zr3(pk.x, pk.y, pk.z), I, instanced));
                            B.slow[4u * pos + (uint32_t)k] = make_uint4(zr_f2u(cc.x), zr_f2u(cc.y), zr_f2u(cc.z), zr_f2u(cc.w));
                        }
                        B.slow[4u * pos + 3u] = make_uint4(prim, slow_rect, 0u, 0u);
                    } else { stats->overflow = 1u; stats->overflow_sticky = 1u; }
                }
            }
            // ---- one record per (triangle, owned tile); ranks within a tile are handed out by k_index
            const bool draw = alive && !hidden;
            const int tx0 = x0 / TILE, ty0 = y0 / TILE;
            const int nx = draw ? x1 / TILE - tx0 + 1 : 0, ny = draw ? y1 / TILE - ty0 + 1 : 0, ntile = nx * ny;
            for (int step = 0; __ballot(step < ntile) != 0ull; ++step) {
                bool emit = step < ntile;
                uint32_t tile = 0;            // the record's tile
                int rtx = 0, rty = 0;
                if (emit) {
                    // (a small triangle spans at most 3 x 3 tiles: the step's row by comparisons, not by a division)
                    const int sy = (step >= nx) + (step >= 2 * nx), sx = step - sy * nx;
                    rtx = tx0 + sx; rty = ty0 + sy;
                    if (P.tile_world > 1u && tile_owner((uint32_t)rtx, (uint32_t)rty, P.tile_world) != P.tile_rank) emit = false;
                    tile = (uint32_t)rty * P.tiles_x + (uint32_t)rtx;
                }
                const unsigned long long me = __ballot(emit);
                if (!me) continue;
                if (!rec_reserve(Wd, (uint32_t)__popcll(me), lane, B, stats, slot)) continue;
                // count per tile: one add per (wave, tile), all of a step's in one instruction, and nobody waits for them.  The step's records
                // are laid down tile by tile (a lane's place = its tile group's start + its rank in the group): a tile's records then form
                // runs of whole cache lines in the chunk, which is what k_tile's gather through the index list reads
                unsigned long long pend = me;
                uint32_t cnt = 0, mypos = 0, gbase = 0;
                while (pend) {
                    const int leader = __builtin_ctzll(pend);
                    const uint32_t tl = (uint32_t)__builtin_amdgcn_readlane((int)tile, leader);
                    const unsigned long long same = __ballot(emit && tile == tl) & pend;
                    const uint32_t ns = (uint32_t)__popcll(same);
                    if ((int)lane == leader) cnt = ns;
                    if (!HIZ && (same >> lane & 1ull)) mypos = gbase + (uint32_t)__popcll(same & lt);
                    gbase += ns;
                    pend &= ~same;
                }
                // (round 2 emits a few records per step: laid down in lane order they leave the wave as whole-line stores; grouped, the same
                // bytes went out as scattered 16-byte writes - 20 MB of write requests for 5.5 MB of records)
                if (HIZ) mypos = (uint32_t)__popcll(me & lt);
                if (cnt) atomicAdd(&tile_count[tile * ZR_TSTRIDE], cnt);
                if (emit) rec_store(B, Wd.cur * ZR_TPOOL_CHUNK + Wd.fill + mypos, r0, r1, r2, prim, tile, rtx, rty);
                Wd.fill += gbase;
            }
        }
    }
    if (lane == 0) {
        if (Wd.cur < B.n_chunks) B.chunk_fill[Wd.cur] = Wd.fill;
        B.wave_culled[wave_id] = HIZ ? culled : 0u;
    }
}

// Exclusive scan of the per-tile record counts into tile_offset and k_tile's work units of <= `unit` records of ONE tile (the counters
// and cursors of the tiles sit ZR_TSTRIDE words apart: atomics on one cache line queue up behind each other, and neighbouring tiles are
// hit together); books the round.  ONE workgroup: every workgroup of k_index scanning the counts for itself was tried (a launch less on
// the camera pipeline's critical path) and is as fast at 1080p but four times slower at 3840 x 2160 (8 160 tiles per scan, 32 KB of LDS
// per workgroup: k_index 1.1 ms instead of 0.3).
__global__ __launch_bounds__(1024) void k_scan_tri(const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_offset,
                                                   uint4* __restrict__ chunk_tab, uint32_t chunk_cap, const uint32_t* __restrict__ owned_tiles, uint32_t n_tiles,
                                                   uint32_t sorted_cap, ZrDevStats* __restrict__ stats, int slot, uint32_t unit)
{
    // (n_tiles = the tiles this context owns, owned_tiles their indices: only they can hold records - a rank of eight scans an eighth)
    __shared__ uint32_t wtot[16], cwtot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t per = (n_tiles + 1023u) / 1024u;
    const uint32_t b = min(n_tiles, tid * per), e = min(n_tiles, b + per);
    uint32_t s = 0, cs = 0;
    for (uint32_t j = b; j < e; ++j) { const uint32_t t = tile_count[owned_tiles[j] * ZR_TSTRIDE]; s += t; cs += (t + unit - 1u) / unit; }
    // scan: inside the wave by shuffles, across the 16 waves through LDS - one barrier (this kernel is one workgroup on the critical path)
    uint32_t incl = s, cincl = cs;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o), cv = (uint32_t)__shfl_up((int)cincl, o);
        if ((int)lane >= o) { incl += v; cincl += cv; }
    }
    if (lane == 63u) { wtot[wv] = incl; cwtot[wv] = cincl; }
    __syncthreads();
    uint32_t wpre = 0, cwpre = 0, tot = 0, ctot = 0;
    for (uint32_t i = 0; i < 16u; ++i) { if (i < wv) { wpre += wtot[i]; cwpre += cwtot[i]; } tot += wtot[i]; ctot += cwtot[i]; }
    uint32_t run = wpre + incl - s, crun = cwpre + cincl - cs;
    for (uint32_t j = b; j < e; ++j) {
        const uint32_t i = owned_tiles[j];
        const uint32_t t = tile_count[i * ZR_TSTRIDE], nu = (t + unit - 1u) / unit;
        tile_offset[i] = run;
        for (uint32_t k = 0; k < nu; ++k)          // k_tile's work units: (tile, first record, end) - one load there, not a search
            if (crun + k < chunk_cap) chunk_tab[crun + k] = make_uint4(i, run + k * unit, run + min(t, (k + 1u) * unit), 0u);
        run += t; crun += nu;
    }
    if (tid == 0) {
        stats->bin_entries[slot] = tot;               // triangle records of the round
        stats->n_chunks[slot] = min(ctot, chunk_cap);
        stats->chunk_counter[slot] = 0;
        stats->survivors[slot] = stats->n_sel[slot];
        if (ctot > chunk_cap || tot > sorted_cap) { stats->overflow = 1u; stats->overflow_sticky = 1u; }
    }
}

// Every record -> its place in its tile's stretch of the tile-ordered INDEX LIST sidx[] (4 bytes per record: the record stays where k_geom
// wrote it and k_tile gathers it).  A cursor per tile is advanced once per (wave, distinct tile) - the 64 records of a wave
// come meshlet by meshlet, so they name a handful of tiles - because atomics on one address run at about 10 ns apiece on this part and
// there are half a million records: the lanes first sort themselves into tile groups (scalar work, no memory), then every group's first
// lane issues its add in ONE instruction.  One wave per record chunk.
__global__ __launch_bounds__(256) void k_index(ZrTriBins B, const ZrDevStats* __restrict__ stats, int slot,
                                               const uint32_t* __restrict__ tile_offset, uint32_t* __restrict__ tile_cursor)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t used = min(B.n_waves + stats->pool_next[slot], B.n_chunks);
    const unsigned long long lt = (1ull << lane) - 1ull;
    // A chunk's four stretches of 64 records go through the dependent steps TOGETHER (tile ids -> tile offsets -> one cursor add per
    // (stretch, tile) group -> stores): the kernel waits for memory three times per chunk, not three times per stretch (it spent 74 % of
    // its wave-cycles waiting: round 3's counters).
    constexpr uint32_t NB = ZR_TPOOL_CHUNK / 64u;
    for (uint32_t ch = blockIdx.x * 4u + wv; ch < used; ch += gridDim.x * 4u) {
        const uint32_t n = B.chunk_fill[ch], r0 = ch * ZR_TPOOL_CHUNK;
        bool have[NB]; uint32_t tile[NB], off[NB], rank[NB], cnt[NB], b[NB]; int first[NB];
#pragma unroll
        for (uint32_t k = 0; k < NB; ++k) {
            const uint32_t j = k * 64u + lane;
            have[k] = j < n;
            tile[k] = have[k] ? B.rtile[r0 + j] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < NB; ++k) off[k] = have[k] ? tile_offset[tile[k]] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < NB; ++k) {
            rank[k] = 0; cnt[k] = 0; first[k] = (int)lane;
            unsigned long long pend = __ballot(have[k]);
            while (pend) {
                const int leader = __builtin_ctzll(pend);
                const uint32_t tl = (uint32_t)__builtin_amdgcn_readlane((int)tile[k], leader);
                const unsigned long long same = __ballot(have[k] && tile[k] == tl) & pend;
                if (same >> lane & 1ull) { first[k] = leader; rank[k] = (uint32_t)__popcll(same & lt); cnt[k] = (uint32_t)__popcll(same); }
                pend &= ~same;
            }
            b[k] = 0;
            if (have[k] && first[k] == (int)lane) b[k] = atomicAdd(&tile_cursor[tile[k] * ZR_TSTRIDE], cnt[k]);
        }
#pragma unroll
        for (uint32_t k = 0; k < NB; ++k) {
            const uint32_t bb = (uint32_t)__shfl((int)b[k], first[k]);
            const uint32_t dst = off[k] + bb + rank[k];
            if (have[k] && dst < B.sorted_cap) B.sidx[dst] = r0 + k * 64u + lane;
        }
    }
}

// Persistent workgroups pull work units: <= ZR_TBATCHES batches of <= ZR_TCHUNK records of one tile, contiguous in the index list; lane per
// triangle: gather, edge setup + walk into the tile's LDS keys; a unit's keys are merged into the frame key buffer once.  Nothing else.
// The kernel also leaves the per-tile counters and the record pool as the next round's k_geom wants them (zero).
// Sorted walk.  The 64 lanes of a wave walk their triangles' boxes in lock step: a row loop as long as the tallest box, a column loop per
// row as long as the widest box still alive there - with a unit's records in arrival order 35 % of the lanes' iterations were live
// (simulated on the benchmark frame, 42.9 column iterations per 64 records for 15.0 of work).  A unit's <= 512
// records therefore go through a counting sort in LDS first, keyed by the clipped box (height, then width, each capped at 15): the
// waves then walk batches of like boxes (31.9 iterations in the same simulation).  The order of the keys' minimum does not matter.
#define ZR_TSORT_BINS 256u
// LAST (the frame's last round): the workgroups then also draw the frame's SLOW triangles (clipped, or with an edge of 64 px or more:
// round 1's in the first half of the list, round 2's in the second) - the usual frame has none, and as a kernel of its own that check
// cost the camera lane 25-50 us of waiting for room beside the shadow rasteriser - and fold k_geom's per-wave Hi-Z tallies into the
// statistics.  The clipper is inlined under this kernel's own register budget (it spills; the path is rare).
template <int MODE, bool LAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8)))
void k_tile(ZrPass P, const uint4* __restrict__ chunk_tab, ZrTriBins B, uint32_t* __restrict__ tile_count,
            uint32_t* __restrict__ tile_cursor, uint32_t n_tiles, ZrDevStats* __restrict__ stats, int slot,
            unsigned long long* __restrict__ vis64, const uint32_t* __restrict__ owned_tiles, uint32_t n_owned)
{
    static_assert(ZR_TCHUNK == 512u && TILE == 32, "two records per thread; box coordinates in 5 bits");
    __shared__ unsigned long long keys64[TILE_PIX];
    __shared__ uint4 srecA[ZR_TCHUNK], srecB[ZR_TCHUNK];
    __shared__ uint32_t hist[ZR_TSORT_BINS], wsum[4];
    __shared__ uint32_t cur_unit;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t n_units = stats->n_chunks[slot];
    for (uint32_t i = blockIdx.x * 256u + tid; i < n_tiles; i += gridDim.x * 256u) { tile_count[i * ZR_TSTRIDE] = 0u; tile_cursor[i * ZR_TSTRIDE] = 0u; }
    if (blockIdx.x == 0 && tid == 0) { stats->pool_used[slot] = stats->pool_next[slot]; }
    uint32_t unit = blockIdx.x;
    bool first = true;
    for (;;) {
        if (unit >= n_units) break;
        for (uint32_t i = tid; i < TILE_PIX; i += 256u) keys64[i] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
        const uint4 ct = chunk_tab[unit];
        const uint32_t tile = ct.x, n_unit = min(ct.z, B.sorted_cap) - min(ct.y, B.sorted_cap);      // <= ZR_TCHUNK * ZR_TBATCHES
        const int tpx0 = (int)(tile % P.tiles_x) * TILE, tpy0 = (int)(tile / P.tiles_x) * TILE;
        TileCtx T;
        T.px0 = 0; T.py0 = 0; T.W = (int)P.W - tpx0; T.H = (int)P.H - tpy0;
        const int wx1 = min(TILE - 1, T.W - 1), wy1 = min(TILE - 1, T.H - 1);
      // a unit's batches of <= ZR_TCHUNK records go into the same keys: one clear and one merge per unit, not per batch
      for (uint32_t b0 = 0; b0 < n_unit; b0 += ZR_TCHUNK) {
        const uint32_t rbeg = ct.y + b0, n = min(n_unit - b0, ZR_TCHUNK);
        hist[tid] = 0u;
        __syncthreads();
        // ---- count: the thread's two records, their clipped boxes (raster_sub's own expressions), the rank among equal keys
        uint4 qa[2], qb[2]; uint32_t key[2], rank[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t j = tid + (uint32_t)k * 256u;
            key[k] = 0u; rank[k] = 0u;
            if (j < n) {
                const uint32_t src = B.sidx[rbeg + j];      // (a tile's records come in runs of one meshlet's: the gather reads whole cache lines mostly)
                qa[k] = B.recA[src]; qb[k] = B.recB[src];
                const int X0 = (int)(short)(qa[k].x & 0xFFFFu), Y0 = (int)qa[k].x >> 16, X1 = (int)(short)(qa[k].z & 0xFFFFu), Y1 = (int)qa[k].z >> 16;
                const int X2 = (int)(short)(qb[k].x & 0xFFFFu), Y2 = (int)qb[k].x >> 16;
                const int x0 = max((imin3(X0, X1, X2) - 128 + 255) >> 8, 0), x1 = min((imax3(X0, X1, X2) - 128) >> 8, wx1);
                const int y0 = max((imin3(Y0, Y1, Y2) - 128 + 255) >> 8, 0), y1 = min((imax3(Y0, Y1, Y2) - 128) >> 8, wy1);
                if (x0 <= x1 && y0 <= y1) {
                    qb[k].w = (uint32_t)x0 | (uint32_t)y0 << 8 | (uint32_t)x1 << 16 | (uint32_t)y1 << 24;
                    key[k] = (uint32_t)min(y1 - y0 + 1, 15) * 16u + (uint32_t)min(x1 - x0 + 1, 15);
                    rank[k] = atomicAdd(&hist[key[k]], 1u);
                }
            }
        }
        __syncthreads();
        // ---- exclusive scan of the 256 bins (bin 0 = records that reach no pixel of the tile: none, by k_geom's construction)
        {
            const uint32_t v = tid ? hist[tid] : 0u;
            uint32_t incl = v;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += u; }
            if (lane == 63u) wsum[wv] = incl;
            __syncthreads();
            uint32_t pre = 0;
            for (uint32_t i = 0; i < wv; ++i) pre += wsum[i];
            hist[tid] = pre + incl - v;
        }
        __syncthreads();
        const uint32_t n_live = wsum[0] + wsum[1] + wsum[2] + wsum[3];
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (key[k]) { const uint32_t sl = hist[key[k]] + rank[k]; srecA[sl] = qa[k]; srecB[sl] = qb[k]; }
        __syncthreads();
        // ---- walk: batches of 64 sorted records; wave w takes batches w and 7 - w (small boxes and big ones: even loads)
        const uint32_t n_batches = (n_live + 63u) >> 6;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t b = k == 0 ? wv : 7u - wv;
            const uint32_t j = b * 64u + lane;
            if (b < n_batches && j < n_live) {
                const uint4 a4 = srecA[j], b4 = srecB[j];
                const RecTri t = rec_load(a4, b4);
                raster_sub<MODE, true, true>(t.a, t.b, t.c, t.prim, T, keys64, nullptr, b4.w);
            }
        }
        __syncthreads();      // (the next batch rewrites hist / srec; the merge below reads the keys)
      }
        for (uint32_t i = tid; i < TILE_PIX; i += 256u) {
            const int px = tpx0 + (int)(i & (TILE - 1)), py = tpy0 + (int)(i / TILE);
            if (px >= (int)P.W || py >= (int)P.H) continue;
            const size_t p = (size_t)py * P.W + (size_t)px;
            const unsigned long long k = keys64[i];
            if ((uint32_t)k != ZR_EMPTY_PRIM) atomicMin(&vis64[p], k);      // (no read-and-compare first: the key buffer is empty but for this tile's other units)
        }
        // the second unit of a workgroup is fixed too (b + grid): when the grid's first units end together, 2 048 claims on one
        // counter would queue up for ~10 ns apiece; only later units (hot frames) come from the counter
        if (first) { first = false; __syncthreads(); unit += gridDim.x; continue; }
        if (tid == 0) cur_unit = 2u * gridDim.x + atomicAdd(&stats->chunk_counter[slot], 1u);
        __syncthreads();
        unit = cur_unit;
    }
    if (LAST) {
        if (slot == 2) {        // the meshlets round 2's k_geom dropped behind the pyramid: per-wave counts, strided over this grid
            uint32_t nc = 0;
            for (uint32_t i = blockIdx.x * 256u + tid; i < B.n_waves; i += gridDim.x * 256u) nc += B.wave_culled[i];
            nc = (uint32_t)wave_sum((int)nc);
            if (lane == 0u && nc) { atomicAdd(&stats->hiz_culled, nc); atomicAdd(&stats->hiz_culled_geom, nc); atomicSub(&stats->survivors[2], nc); }
        }
        const uint32_t half_cap = B.slow_cap / 2u;
        const uint32_t n_a = min(stats->n_slow[1], half_cap), n_b = slot == 2 ? min(stats->n_slow[2], half_cap) : 0u;
        if (n_a + n_b == 0u) return;
        __syncthreads();
        for (uint32_t ti = blockIdx.x; ti < n_owned; ti += gridDim.x) {
            const uint32_t tile = owned_tiles[ti];
            for (uint32_t i = tid; i < TILE_PIX; i += 256u) keys64[i] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
            __syncthreads();
            const int tpx0 = (int)(tile % P.tiles_x) * TILE, tpy0 = (int)(tile / P.tiles_x) * TILE;
            TileCtx T;
            T.px0 = 0; T.py0 = 0; T.W = (int)P.W - tpx0; T.H = (int)P.H - tpy0;
            const uint32_t ttx = tile % P.tiles_x, tty = tile / P.tiles_x;
            for (uint32_t jj = tid; jj < n_a + n_b; jj += 256u) {
                const uint32_t j = jj < n_a ? jj : half_cap + (jj - n_a);
                const uint4 q3 = B.slow[4u * j + 3u];
                // the tiles the triangle's snapped box reaches (k_geom), or all of them
                if (ttx < (q3.y & 255u) || tty < ((q3.y >> 8) & 255u) || ttx > ((q3.y >> 16) & 255u) || tty > (q3.y >> 24)) continue;
                const uint4 q0 = B.slow[4u * j], q1 = B.slow[4u * j + 1u], q2 = B.slow[4u * j + 2u];
                zf4 c0, c1, c2;
                c0.x = zr_u2f(q0.x); c0.y = zr_u2f(q0.y); c0.z = zr_u2f(q0.z); c0.w = zr_u2f(q0.w);
                c1.x = zr_u2f(q1.x); c1.y = zr_u2f(q1.y); c1.z = zr_u2f(q1.z); c1.w = zr_u2f(q1.w);
                c2.x = zr_u2f(q2.x); c2.y = zr_u2f(q2.y); c2.z = zr_u2f(q2.z); c2.w = zr_u2f(q2.w);
                raster_clipped_body<MODE>(c0, c1, c2, q3.x, T, P.hw, P.hh, tpx0 * 256, tpy0 * 256, keys64, nullptr);
            }
            __syncthreads();
            for (uint32_t i = tid; i < TILE_PIX; i += 256u) {
                const int px = tpx0 + (int)(i & (TILE - 1)), py = tpy0 + (int)(i / TILE);
                if (px >= (int)P.W || py >= (int)P.H) continue;
                const size_t p = (size_t)py * P.W + (size_t)px;
                const unsigned long long k = keys64[i];
                if ((uint32_t)k != ZR_EMPTY_PRIM && k < vis64[p]) atomicMin(&vis64[p], k);
            }
            __syncthreads();
        }
    }
}

// The skydome pass's visibility (ZE:3681-3691, SH/Skydome.vert): the dome's triangles against each other, LESS in draw order, into a key
// plane of their own (depth bits << 32 | triangle).  Workgroup per owned tile; every thread takes its share of the dome's few hundred
// triangles through the general path (classification, clipper, 64-bit walk: the dome surrounds the eye, most of its triangles cross
// the guard band), clipped to the tile; the tile's keys are stored whole, so the plane needs no clear.
__global__ __launch_bounds__(256) void k_sky_tiles(ZrPass P, const ZrObject* __restrict__ objs, const uint32_t* __restrict__ owned_tiles,
                                                   unsigned long long* __restrict__ sky64)
{
    __shared__ unsigned long long keys64[TILE_PIX];
    const uint32_t tid = threadIdx.x, tile = owned_tiles[blockIdx.x];
    for (uint32_t i = tid; i < TILE_PIX; i += 256u) keys64[i] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
    __syncthreads();
    const ZrObject* __restrict__ O = objs + P.sky_object;
    const int tpx0 = (int)(tile % P.tiles_x) * TILE, tpy0 = (int)(tile / P.tiles_x) * TILE;
    TileCtx T;
    T.px0 = 0; T.py0 = 0; T.W = (int)P.W - tpx0; T.H = (int)P.H - tpy0;
    const ZrInstance I = O->inst[0];
    for (uint32_t t = tid; t < O->n_tris; t += 256u) {
        zf4 c[3];
        for (int k = 0; k < 3; ++k) {
            const float4 q0 = ld_global((const float4*)(O->rverts + ld_global(O->indices + 3u * t + (uint32_t)k)));
            c[k] = zr_mat4_point(P.PVM, vs_position(zr3(q0.x, q0.y, q0.z), I, false));
        }
        if (classify(vertex_flags(c[0]), vertex_flags(c[1]), vertex_flags(c[2])) == 0) continue;
        raster_clipped<ZR_MODE_GBUFFER>(c[0], c[1], c[2], t, T, P.hw, P.hh, tpx0 * 256, tpy0 * 256, keys64, (uint32_t*)nullptr);
    }
    __syncthreads();
    for (uint32_t i = tid; i < TILE_PIX; i += 256u) {
        const int px = tpx0 + (int)(i & (TILE - 1)), py = tpy0 + (int)(i / TILE);
        if (px < (int)P.W && py < (int)P.H) sky64[(size_t)py * P.W + (size_t)px] = keys64[i];
    }
}

// ------------------------------------------------------------------------------------------------ launchers (C++ linkage, used by zr_host.cpp)

void zr_launch_hiz_build(const unsigned long long* vis64, uint32_t W, uint32_t H, const ZrHiz& Z, const uint32_t* regions, uint32_t n_regions, hipStream_t s)
{
    if (n_regions) hipLaunchKernelGGL(k_hiz_build, dim3(n_regions), dim3(256), 0, s, vis64, W, H, Z, regions);
}
void zr_launch_select(const ZrPass& P, const ZrObject* objs, const uint32_t* work, const uint32_t* rects, const ZrHiz& Z, const ZrTriBins& B, ZrDevStats* stats,
                      int slot, hipStream_t s)
{
    if (P.n_work == 0) return;
    hipLaunchKernelGGL(k_select, dim3((P.n_work + 1023) / 1024), dim3(ZR_SELECT_THREADS), 0, s, P, objs, work, rects, Z, B.sel, stats, slot);
}
void zr_launch_geom(const ZrPass& P, const ZrHiz& Z, const ZrTriBins& B, uint32_t* tile_count, ZrDevStats* stats, int slot, unsigned long long* vis64, hipStream_t s)
{
    const dim3 g(B.n_waves / 4u), b(256);
    if (Z.phase == 2u) hipLaunchKernelGGL(k_geom<true>, g, b, 0, s, P, B.sel, Z, B, tile_count, stats, slot, vis64);
    else hipLaunchKernelGGL(k_geom<false>, g, b, 0, s, P, B.sel, Z, B, tile_count, stats, slot, vis64);
}
void zr_launch_scan_tri(const uint32_t* tile_count, uint32_t* tile_offset, uint4* chunk_tab, uint32_t chunk_cap, const uint32_t* owned_tiles, uint32_t n_owned,
                        const ZrTriBins& B, ZrDevStats* stats, int slot, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan_tri, dim3(1), dim3(1024), 0, s, tile_count, tile_offset, chunk_tab, chunk_cap, owned_tiles, n_owned, B.sorted_cap, stats, slot, ZR_TCHUNK * ZR_TBATCHES);
}
void zr_launch_index(const ZrTriBins& B, const uint32_t* tile_offset, uint32_t* tile_cursor, const ZrDevStats* stats, int slot, hipStream_t s)
{
    hipLaunchKernelGGL(k_index, dim3(B.n_waves / 4u), dim3(256), 0, s, B, stats, slot, tile_offset, tile_cursor);
}
void zr_launch_tile(const ZrPass& P, const uint4* chunk_tab, const ZrTriBins& B, uint32_t* tile_count, uint32_t* tile_cursor, uint32_t n_tiles,
                    ZrDevStats* stats, int slot, unsigned long long* vis64, uint32_t n_blocks, hipStream_t s, bool last, const uint32_t* owned_tiles, uint32_t n_owned)
{
    if (last) hipLaunchKernelGGL((k_tile<ZR_MODE_GBUFFER, true>), dim3(n_blocks), dim3(256), 0, s, P, chunk_tab, B, tile_count, tile_cursor, n_tiles, stats, slot, vis64, owned_tiles, n_owned);
    else hipLaunchKernelGGL((k_tile<ZR_MODE_GBUFFER, false>), dim3(n_blocks), dim3(256), 0, s, P, chunk_tab, B, tile_count, tile_cursor, n_tiles, stats, slot, vis64, owned_tiles, n_owned);
}
void zr_launch_sky_tiles(const ZrPass& P, const ZrObject* objs, const uint32_t* owned_tiles, uint32_t n_owned, unsigned long long* sky64, hipStream_t s)
{
    if (n_owned) hipLaunchKernelGGL(k_sky_tiles, dim3(n_owned), dim3(256), 0, s, P, objs, owned_tiles, sky64);
}
