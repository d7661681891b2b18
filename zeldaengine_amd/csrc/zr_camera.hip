// zr_camera.hip — the camera pass (deferred-scene pass, ZE:3417-3480) as triangle-level binning: k_hiz_build, k_select, k_geom<HIZ>,
// k_plan, k_tile<MODE, LAST>, k_sky_tiles.  See zr_dev.h for the map of the kernel files.
#include "zr_dev.h"
#include "zr_raster.h"

#include <type_traits>

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
// pixel centre, Hi-Z in round 2) and appends one 32-byte record per (triangle, owned tile) - vertices relative to the tile, three depths,
// the primitive id - to that tile's BUCKET of the record arrays (laid out by k_plan from the previous frame's counts: "triangle records"
// below); k_tile's lanes read a bucket as one run and do nothing but edge setup + walk on live triangles.  Same arithmetic, same keys as the
// meshlet-binned path (kept in -DZR_DIAG builds for A/B): the frame is the same bit for bit.

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
// TILE BUCKETS.  Every tile owns a stretch [tile_base, tile_base + tile_cap) of the record arrays, laid out by k_plan at the end of the
// PREVIOUS frame from that frame's own per-tile counts (+ 25 % + 32): k_geom appends a meshlet's records for a tile with ONE returning add on
// the tile's cursor, k_tile reads its tile's stretch as one contiguous run.  No scan and no index list sit between the two kernels (they
// were k_scan_tri + k_index: two launches, 33 us per frame alone, on the camera lane's critical path, and a 20 MB index list).  What does
// not fit its bucket - the camera moved, a tile got busier than last frame - goes to ONE overflow region (in ZR_OVER_SECTIONS sections, by
// tile id modulo) with its tile id beside it, and the tile's first work unit picks its records out of its section: slower, exact, and rare.  A frame without a usable plan (first frame
// of a scene; the round structure changed) runs k_geom once more ahead of the round, counting only, and plans from that.
__device__ __forceinline__ uint32_t pack_xy(int X, int Y) { return ((uint32_t)X & 0xFFFFu) | ((uint32_t)Y << 16); }
typedef unsigned short zr_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b)      // (a.lo - b.lo) mod 2^16 | (a.hi - b.hi) mod 2^16 << 16: one v_pk_sub_u16
{
    zr_us2 x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
    const zr_us2 r = x - y;
    uint32_t o; __builtin_memcpy(&o, &r, 4);
    return o;
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
// the tiles a drawn triangle's clipped pixel box reaches, packed tx0 | ty0 << 8 | tx1 << 16 | ty1 << 24; ZR_NO_TILES: draws nothing
#define ZR_NO_TILES 0x0000FFFFu          // tx0 = ty0 = 255 > tx1 = ty1 = 0: touches no tile
__device__ __forceinline__ bool rect_has(uint32_t r, int tx, int ty)
{
    return (int)(r & 255u) <= tx && tx <= (int)((r >> 16) & 255u) && (int)((r >> 8) & 255u) <= ty && ty <= (int)(r >> 24);
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
// owned tile), appended to the tiles' buckets: the wave first counts what every tile of the meshlet's tile rectangle gets (ballots), then
// the lane of each such tile reserves the run with one returning add on the tile's cursor - ONE round trip per meshlet, whatever its
// triangles - and the records go out as one contiguous run per tile.
// ROUND 2 (HIZ = true): a meshlet whose snapped vertex box lies behind this frame's pyramid is dropped after the vertex phase, and every
// triangle is tested once more by itself against the 4 x 4-pixel level (the meshlet's blocks stay in LDS for that).
// COUNT: nothing is stored and nothing returned - the per-tile counts of the round for k_plan (a frame without a usable plan).
#ifndef ZR_GEOM_WAVES
#define ZR_GEOM_WAVES 8
#endif
template <bool HIZ, bool COUNT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ZR_GEOM_WAVES, ZR_GEOM_WAVES)))
void k_geom(ZrPass P, const ZrBinEntry* __restrict__ sel, ZrHiz Z, ZrTriBins B, ZrDevStats* __restrict__ stats, int slot)
{
    __shared__ int4 vstage[4][WAVE];
    __shared__ float hzs[4][WAVE];
    const uint32_t lane = threadIdx.x & 63u, wv = wave_uniform(threadIdx.x >> 6);
    const uint32_t n = stats->n_sel[slot];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t wave_id = blockIdx.x * 4u + wv, n_waves = gridDim.x * 4u;
    constexpr bool pyramid = HIZ;
    if (!COUNT && wave_id == 0u && lane == 0u) stats->survivors[slot] = n;      // (round 2: k_tile takes the meshlets dropped behind the pyramid off)
    uint32_t* __restrict__ const cursor = B.cursor + (size_t)(slot == 2 ? 1u : 0u) * B.n_tiles * ZR_TSTRIDE;
    uint32_t culled = 0;
    // where the overflow region begins this frame and what each of its sections holds (k_plan: the record arrays behind the last bucket)
    const uint32_t over_start = COUNT ? 0u : wave_uniform(B.plan[0]), sec_cap = COUNT ? 0u : wave_uniform(B.plan[1]);
    // the wave's next meshlet record is fetched (scalar loads: the address is the wave's) while the current one is worked on: the chain of
    // dependent round trips per meshlet is vertices -> bucket reservation, not record -> vertices -> reservation
    ZrBinEntry nx = ld_record(sel + min(wave_id, n ? n - 1u : 0u));
    for (uint32_t i = wave_id; i < n; i += n_waves) {
        const ZrBinEntry be = nx;
        if (i + n_waves < n) nx = ld_record(sel + (i + n_waves));
        const float4* __restrict__ mp = be.mpos;
        const uint2* __restrict__ tw = be.mtri;
        const ZrInstance* __restrict__ ip = be.inst;
        const uint32_t counts = be.counts, pbase = be.prim_base;
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

        // ---- per triangle (two per lane): the exact tests; what is left is the rectangle of tiles its clipped pixel box reaches
        uint32_t trect[2] = { ZR_NO_TILES, ZR_NO_TILES };
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
            const unsigned long long ms = COUNT ? 0ull : __ballot(is_slow);
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
                    } else { stats->overflow = 1u; stats->overflow_sticky = ZR_OVF_SLOW; }
                }
            }
            if (alive && !hidden) trect[round] = (uint32_t)(x0 / TILE) | (uint32_t)(y0 / TILE) << 8 | (uint32_t)(x1 / TILE) << 16 | (uint32_t)(y1 / TILE) << 24;
        }
        // ---- the meshlet's own rectangle of tiles (a small triangle spans at most 3 x 3; the meshlet's usually 1 .. 4 tiles in all)
        const unsigned long long any = __ballot(trect[0] != ZR_NO_TILES || trect[1] != ZR_NO_TILES);
        if (!any) continue;
        int TX0, TY0, TW, TH;
        {
            int lo = 0x7FFF7FFF, hi = (int)0x80008000;
#pragma unroll
            for (int round = 0; round < 2; ++round)
                if (trect[round] != ZR_NO_TILES) {
                    lo = op_pkmin(lo, (int)((trect[round] & 255u) | ((trect[round] >> 8) & 255u) << 16));
                    hi = op_pkmax(hi, (int)(((trect[round] >> 16) & 255u) | (trect[round] >> 24) << 16));
                }
            lo = wave_pkmin16(lo); hi = wave_pkmax16(hi);
            TX0 = lo & 0xFFFF; TY0 = lo >> 16; TW = (hi & 0xFFFF) - TX0 + 1; TH = (hi >> 16) - TY0 + 1;
        }
        const int n_rect = TW * TH;
        // The cells of the rectangle are taken 32 at a time (one turn, unless the meshlet reaches more than 32 tiles).  Every lane first
        // forms, per triangle, the MASK of the turn's cells its tile box covers - a row of 1 .. 3 bits, up to three times - so that "is my
        // triangle in cell k" costs an AND in the loops over the cells below (they were box comparisons: 4 extracts + 4 compares per
        // triangle per cell, most of the kernel's instructions).
        for (int c0 = 0; c0 < n_rect; c0 += 32) {
            const int nn = min(32, n_rect - c0);
            uint32_t cm[2] = { 0u, 0u };
            if (n_rect <= 32) {
#pragma unroll
                for (int round = 0; round < 2; ++round)
                    if (trect[round] != ZR_NO_TILES) {
                        const uint32_t t = trect[round];
                        const int s0 = ((int)((t >> 8) & 255u) - TY0) * TW + ((int)(t & 255u) - TX0);
                        const uint32_t w1 = ((t >> 16) & 255u) - (t & 255u), h1 = (t >> 24) - ((t >> 8) & 255u);      // (0 .. 2: a small triangle spans at most 3 tiles)
                        const uint32_t row = (2u << w1) - 1u;
                        uint32_t m = row << s0;
                        if (h1 >= 1u) m |= row << (s0 + TW);
                        if (h1 >= 2u) m |= row << (s0 + 2 * TW);
                        cm[round] = m;
                    }
            } else {
                int tx = TX0 + c0 % TW, ty = TY0 + c0 / TW;
                for (int k = 0; k < nn; ++k) {
                    if (rect_has(trect[0], tx, ty)) cm[0] |= 1u << k;
                    if (rect_has(trect[1], tx, ty)) cm[1] |= 1u << k;
                    if (++tx == TX0 + TW) { tx = TX0; ++ty; }
                }
            }
            // what every cell of the turn gets: lane k keeps cell k's numbers
            int cnt_i = 0, cnt0_i = 0, mtx = 0, mty = 0;
            {
                int tx = TX0 + c0 % TW, ty = TY0 + c0 / TW;
                for (int k = 0; k < nn; ++k) {
                    const uint32_t bit = 1u << k;
                    const int n0 = __popcll(__ballot((cm[0] & bit) != 0u)), n1 = __popcll(__ballot((cm[1] & bit) != 0u));
                    if ((int)lane == k) { cnt0_i = n0; cnt_i = n0 + n1; mtx = tx; mty = ty; }
                    if (++tx == TX0 + TW) { tx = TX0; ++ty; }
                }
            }
            uint32_t cnt = (uint32_t)cnt_i;
            const uint32_t cnt0 = (uint32_t)cnt0_i;
            if (P.tile_world > 1u && tile_owner((uint32_t)mtx, (uint32_t)mty, P.tile_world) != P.tile_rank) cnt = 0;      // another rank's tile
            const uint32_t tile = (uint32_t)mty * P.tiles_x + (uint32_t)mtx;
            if (COUNT) { if (cnt) atomicAdd(&cursor[tile * ZR_TSTRIDE], cnt); continue; }
            // one returning add per tile of the meshlet, all of them in one instruction: the run's place in the tile's bucket
            uint32_t run = 0, tbase = 0, tcap = 0;
            if (cnt) { run = atomicAdd(&cursor[tile * ZR_TSTRIDE], cnt); tbase = B.tile_base[tile]; tcap = B.tile_cap[tile]; }
            // what does not fit the bucket (records run .. run + cnt - 1 at places >= tcap) goes to the overflow region: rare
            // (the region has ZR_OVER_SECTIONS sections, a tile's records go to section tile % ZR_OVER_SECTIONS: a tile that spilled sifts one
            // section, not the region)
            const uint32_t n_over = (cnt && run + cnt > tcap) ? run + cnt - max(run, tcap) : 0u;
            uint32_t obase = 0;
            if (n_over) obase = atomicAdd(&B.over_cursor[(slot == 2 ? ZR_OVER_SECTIONS : 0u) + (tile & (ZR_OVER_SECTIONS - 1u))], n_over);
            // the records, tile by tile (wave-uniform loop): a tile's run is contiguous, round 0's triangles first
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                if ((uint32_t)round * WAVE >= tcount) break;
                // the triangle's vertices as the record holds them, but for the tile's origin: x | y << 16 (mod 2^16 each) and depth
                uint32_t q0 = 0u, q1 = 0u, q2 = 0u, z0 = 0u, z1 = 0u, z2 = 0u;
                if (cm[round]) {
                    const int4 r0 = vstage[wv][tri_w[round].x & 255u], r1 = vstage[wv][(tri_w[round].x >> 8) & 255u], r2 = vstage[wv][(tri_w[round].x >> 16) & 255u];
                    q0 = pack_xy(r0.x, r0.y); q1 = pack_xy(r1.x, r1.y); q2 = pack_xy(r2.x, r2.y);
                    z0 = (uint32_t)r0.z; z1 = (uint32_t)r1.z; z2 = (uint32_t)r2.z;
                }
                const uint32_t prim = pbase + tri_w[round].y;
                for (int k = 0; k < nn; ++k) {
                    const uint32_t ck = (uint32_t)__builtin_amdgcn_readlane((int)cnt, k);
                    if (!ck) continue;
                    const bool mine = (cm[round] & (1u << k)) != 0u;
                    const unsigned long long m = __ballot(mine);
                    if (!m) continue;
                    const int tx = __builtin_amdgcn_readlane(mtx, k), ty = __builtin_amdgcn_readlane(mty, k);
                    const uint32_t rk = (uint32_t)__builtin_amdgcn_readlane((int)run, k), ca = (uint32_t)__builtin_amdgcn_readlane((int)tcap, k);
                    if (mine) {
                        const uint32_t place = rk + (round ? (uint32_t)__builtin_amdgcn_readlane((int)cnt0, k) : 0u) + (uint32_t)__popcll(m & lt);
                        uint32_t pos;
                        if (place < ca) pos = (uint32_t)__builtin_amdgcn_readlane((int)tbase, k) + place;
                        else {
                            const uint32_t tl = (uint32_t)ty * P.tiles_x + (uint32_t)tx;
                            const uint32_t ol = (uint32_t)__builtin_amdgcn_readlane((int)obase, k) + (place - max(rk, ca));      // place in the tile's section
                            pos = ol < sec_cap ? over_start + (tl & (ZR_OVER_SECTIONS - 1u)) * sec_cap + ol : 0xFFFFFFFFu;
                            if (ol < sec_cap) B.over_tile[pos] = tl;
                            else { stats->overflow = 1u; stats->overflow_sticky = ZR_OVF_RECORDS; }      // the record arrays are full: the frame is incomplete, and says so
                        }
                        if (pos != 0xFFFFFFFFu) {
                            // tile-relative coordinates: both halves minus the tile's origin, each modulo 2^16 (= pack_xy of the differences)
                            const uint32_t o = pack_xy(tx * (TILE * 256), ty * (TILE * 256));
                            B.recA[pos] = make_uint4(pk_sub16(q0, o), z0, pk_sub16(q1, o), z1);
                            B.recB[pos] = make_uint4(pk_sub16(q2, o), z2, prim, 0u);
                        }
                    }
                }
            }
        }
    }
    // (the round's record count is the sum of the tiles' cursors: k_plan books it - 8 192 waves adding to one word here would queue for ~10 ns apiece)
    if (lane == 0 && !COUNT) B.wave_culled[wave_id] = HIZ ? culled : 0u;
}

#ifndef ZR_BUCKET_SLACK
#define ZR_BUCKET_SLACK 128u            // records a bucket holds beyond last frame's count + 25 %: a sphere's worth of triangles entering an empty tile
#endif
// The plan of the NEXT frame's record arrays, from this frame's per-tile counts (the greater of its rounds'): every owned tile gets a
// bucket of count + 25 % + ZR_BUCKET_SLACK records (all of them scaled down together should they not fit: the overflow region takes what spills) and
// ceil(bucket / unit) work units of k_tile; both rounds of a frame use the same buckets and the same units (a unit takes its share of
// whatever its tile's cursor says).  Also what used to be k_tile's first duty: the cursors are zero again for the next frame.
// ONE workgroup, after the resolve: the camera lane has nothing to do until the next frame begins, nothing waits for this launch.
// `exact`: the counts come from a count-only run of the very round that follows - the buckets are the counts themselves, nothing can spill.
__global__ __launch_bounds__(1024) void k_plan(ZrTriBins B, const uint32_t* __restrict__ owned_tiles, uint32_t n_owned, uint32_t unit, ZrDevStats* __restrict__ stats,
                                               uint32_t exact, uint32_t bucket_pct)
{
    __shared__ uint32_t wtot[16], cwtot[16];
    __shared__ unsigned long long gsum[16];
    __shared__ uint32_t nsum[16][2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t per = (n_owned + 1023u) / 1024u;
    const uint32_t b = min(n_owned, tid * per), e = min(n_owned, b + per);
    uint32_t* __restrict__ c0 = B.cursor; uint32_t* __restrict__ c1 = B.cursor + (size_t)B.n_tiles * ZR_TSTRIDE;
    // pass 1: what the buckets would like in total; and the records each round appended (the frame's statistics)
    unsigned long long want = 0;
    uint32_t n0 = 0, n1 = 0;
    for (uint32_t j = b; j < e; ++j) {
        const uint32_t t = owned_tiles[j], a0 = c0[t * ZR_TSTRIDE], a1 = c1[t * ZR_TSTRIDE], c = max(a0, a1);
        want += exact ? (unsigned long long)c : (unsigned long long)c + (c >> 2) + ZR_BUCKET_SLACK;
        n0 += a0; n1 += a1;
    }
    for (int o = 32; o > 0; o >>= 1) {
        want += (unsigned long long)__shfl_xor((long long)want, o);
        n0 += (uint32_t)__shfl_xor((int)n0, o); n1 += (uint32_t)__shfl_xor((int)n1, o);
    }
    if (lane == 0u) { gsum[wv] = want; nsum[wv][0] = n0; nsum[wv][1] = n1; }
    __syncthreads();
    unsigned long long all = 0;
    for (uint32_t i = 0; i < 16u; ++i) all += gsum[i];
    if (tid == 0 && stats) {
        uint32_t r0 = 0, r1 = 0, o0 = 0, o1 = 0;
        for (uint32_t i = 0; i < 16u; ++i) { r0 += nsum[i][0]; r1 += nsum[i][1]; }
        for (uint32_t i = 0; i < ZR_OVER_SECTIONS; ++i) { o0 += B.over_cursor[i]; o1 += B.over_cursor[ZR_OVER_SECTIONS + i]; }
        stats->bin_entries[1] = r0; stats->bin_entries[2] = r1;
        stats->pool_used[1] = o0; stats->pool_used[2] = o1;      // (pool_used: the rounds' overflow records)
    }
    const bool shrink = all > (unsigned long long)B.bucket_max;
    // pass 2: buckets and units (one scan of each)
    uint32_t s = 0, cs = 0;
    for (uint32_t j = b; j < e; ++j) {
        const uint32_t t = owned_tiles[j], c = max(c0[t * ZR_TSTRIDE], c1[t * ZR_TSTRIDE]);
        unsigned long long cap = exact ? (unsigned long long)c : (unsigned long long)c + (c >> 2) + ZR_BUCKET_SLACK;
        if (shrink) cap = cap * B.bucket_max / all;
        if (bucket_pct != 100u) cap = cap * bucket_pct / 100u;      // (zr_set_bucket_share)
        s += (uint32_t)cap; cs += max(1u, ((uint32_t)cap + unit - 1u) / unit);
    }
    uint32_t incl = s, cincl = cs;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o), cv = (uint32_t)__shfl_up((int)cincl, o);
        if ((int)lane >= o) { incl += v; cincl += cv; }
    }
    if (lane == 63u) { wtot[wv] = incl; cwtot[wv] = cincl; }
    __syncthreads();
    uint32_t wpre = 0, cwpre = 0, ctot = 0, rtot = 0;
    for (uint32_t i = 0; i < 16u; ++i) { if (i < wv) { wpre += wtot[i]; cwpre += cwtot[i]; } ctot += cwtot[i]; rtot += wtot[i]; }
    uint32_t run = wpre + incl - s, crun = cwpre + cincl - cs;
    for (uint32_t j = b; j < e; ++j) {
        const uint32_t t = owned_tiles[j], c = max(c0[t * ZR_TSTRIDE], c1[t * ZR_TSTRIDE]);
        unsigned long long cap = exact ? (unsigned long long)c : (unsigned long long)c + (c >> 2) + ZR_BUCKET_SLACK;
        if (shrink) cap = cap * B.bucket_max / all;
        if (bucket_pct != 100u) cap = cap * bucket_pct / 100u;      // (zr_set_bucket_share)
        const uint32_t nu = max(1u, ((uint32_t)cap + unit - 1u) / unit);
        B.tile_base[t] = run; B.tile_cap[t] = (uint32_t)cap;
        for (uint32_t k = 0; k < nu; ++k)
            if (crun + k < B.unit_cap) B.unit_tab[crun + k] = make_uint4(t, k, nu, 0u);
        run += (uint32_t)cap; crun += nu;
        c0[t * ZR_TSTRIDE] = 0u; c1[t * ZR_TSTRIDE] = 0u;
    }
    if (tid == 0) {
        *B.n_units = min(ctot, B.unit_cap);
        // everything behind the last bucket is the frame's overflow region, in ZR_OVER_SECTIONS equal sections (a frame after a camera cut
        // can put most of its records there: round 2 then draws what the stale pyramid hid, far beyond last frame's counts)
        B.plan[0] = rtot; B.plan[1] = (B.n_rec - rtot) / ZR_OVER_SECTIONS;
        for (uint32_t i = 0; i < 2u * ZR_OVER_SECTIONS; ++i) B.over_cursor[i] = 0u;
        if (ctot > B.unit_cap && stats) { stats->overflow = 1u; stats->overflow_sticky = ZR_OVF_UNITS; }
    }
}

// Persistent workgroups pull work units: a unit is part `part` of `parts` of ONE tile's bucket (k_plan: a tile whose bucket holds more than
// ZR_TCHUNK * ZR_TBATCHES records is shared by several units), walked in batches of <= ZR_TCHUNK records; lane per triangle: edge setup + walk
// into the tile's LDS keys; a unit's keys are merged into the frame key buffer once.  A tile's first unit also takes the tile's records out
// of the overflow region (what did not fit the bucket this frame), if there are any.
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
void k_tile(ZrPass P, ZrTriBins B, ZrDevStats* __restrict__ stats, int slot,
            unsigned long long* __restrict__ vis64, const uint32_t* __restrict__ owned_tiles, uint32_t n_owned)
{
    static_assert(ZR_TCHUNK == 512u && TILE == 32, "two records per thread; box coordinates in 5 bits");
    __shared__ unsigned long long keys64[TILE_PIX];
    __shared__ uint4 srecA[ZR_TCHUNK], srecB[ZR_TCHUNK];
    __shared__ uint32_t hist[ZR_TSORT_BINS], wsum[4];
    __shared__ uint32_t cur_unit, uh[4], ocnt[4], omore[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t n_units = *B.n_units, ri = slot == 2 ? 1u : 0u;
    const uint32_t over_start = wave_uniform(B.plan[0]), sec_cap = wave_uniform(B.plan[1]);      // this frame's overflow region (k_plan)
    const uint32_t* __restrict__ const cursor = B.cursor + (size_t)ri * B.n_tiles * ZR_TSTRIDE;
    if (blockIdx.x == 0 && tid == 0) stats->n_chunks[slot] = n_units;
    uint32_t unit = blockIdx.x;
    bool first = true;
    for (;;) {
        if (unit >= n_units) break;
        // the unit's header, read by ONE thread while the others clear the keys: its tile, its share of the tile's bucket (records [lo, lo + n_own)
        // of the min(cursor, cap) the bucket holds: equal shares, the last one short) and - a tile's first unit only - how much of the overflow
        // region it has to sift for records that name its tile
        if (tid == 0) {
            const uint4 ct = B.unit_tab[unit];
            const uint32_t n_all = cursor[ct.x * ZR_TSTRIDE], tcap = B.tile_cap[ct.x], n_in = min(n_all, tcap);
            const uint32_t len = (n_in + ct.z - 1u) / ct.z, lo = min(n_in, ct.y * len);
            uh[0] = ct.x; uh[1] = min(n_in - lo, len); uh[2] = B.tile_base[ct.x] + lo;
            uh[3] = (ct.y == 0u && n_all > tcap) ? min(B.over_cursor[ri * ZR_OVER_SECTIONS + (ct.x & (ZR_OVER_SECTIONS - 1u))], sec_cap) : 0u;
        }
        for (uint32_t i = tid; i < TILE_PIX; i += 256u) keys64[i] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
        __syncthreads();
        const uint32_t tile = wave_uniform(uh[0]), n_own = wave_uniform(uh[1]), rec0 = wave_uniform(uh[2]), n_sift = wave_uniform(uh[3]);
        const uint32_t n_unit = n_own + n_sift;
        if (n_unit == 0u) {           // nothing for this unit (an empty tile keeps its one unit): the next one
            if (first) { first = false; __syncthreads(); unit += gridDim.x; continue; }
            if (tid == 0) cur_unit = 2u * gridDim.x + atomicAdd(&stats->chunk_counter[slot], 1u);
            __syncthreads();
            unit = cur_unit;
            __syncthreads();
            continue;
        }
        const int tpx0 = (int)(tile % P.tiles_x) * TILE, tpy0 = (int)(tile / P.tiles_x) * TILE;
        TileCtx T;
        T.px0 = 0; T.py0 = 0; T.W = (int)P.W - tpx0; T.H = (int)P.H - tpy0;
        const int wx1 = min(TILE - 1, T.W - 1), wy1 = min(TILE - 1, T.H - 1);
        // The unit's batches of <= ZR_TCHUNK records go into the same keys (one clear and one merge per unit, not per batch): first the unit's
        // share of the bucket, one contiguous run; then - a tile's first unit, when the tile got more than its bucket holds - the tile's
        // records in its section of the overflow region.  Each WAVE sifts a quarter of the section's tile ids (64 per step, ballot +
        // count: no atomics, no barriers) into a list of its own in the record staging area, which is idle here, and stops above 64
        // entries; the four lists (<= 4 x 128 = ZR_TCHUNK) are one batch; the waves go on from where they stopped until the section is
        // done.  The usual frame has a few dozen such records for a few hundred tiles: one step, one small batch; a frame after a camera
        // cut reads each section once per tile of the section.
        uint32_t* const olist = (uint32_t*)srecA;
        // one batch: n records - the run [src0, src0 + n) of the arrays, or (LIST) the entries of the four sift lists end to end
        auto do_batch = [&](const uint32_t n, const uint32_t src0, auto list_tag) {
        constexpr bool LIST = decltype(list_tag)::value;
        hist[tid] = 0u;
        __syncthreads();
        // ---- count: the thread's two records, their clipped boxes (raster_sub's own expressions), the rank among equal keys
        uint4 qa[2], qb[2]; uint32_t key[2], rank[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t j = tid + (uint32_t)k * 256u;
            key[k] = 0u; rank[k] = 0u;
            if (j < n) {
                uint32_t src = src0 + j;
                if (LIST) {        // (read before anything is staged over the lists)
                    const uint32_t l1 = ocnt[0], l2 = l1 + ocnt[1], l3 = l2 + ocnt[2];
                    const uint32_t w = (j >= l1 ? 1u : 0u) + (j >= l2 ? 1u : 0u) + (j >= l3 ? 1u : 0u);
                    src = olist[w * 128u + (j - (w == 0u ? 0u : w == 1u ? l1 : w == 2u ? l2 : l3))];
                }
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
            const uint32_t incl = wave_incl_scan(v);      // (on the DPP network: no lane-index registers held across the unit loop)
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
        };
        for (uint32_t b0 = 0; b0 < n_own; b0 += ZR_TCHUNK) do_batch(min(n_own - b0, ZR_TCHUNK), rec0 + b0, std::false_type());
        if (n_sift) {
            const uint32_t sec0 = over_start + (tile & (ZR_OVER_SECTIONS - 1u)) * sec_cap;      // the tile's section (an index into the record arrays)
            const uint32_t quarter = ((n_sift + 255u) >> 8) << 6;
            uint32_t spos = wave_uniform(min(n_sift, wv * quarter));      // (wave-uniform, like send and wc)
            const uint32_t send = wave_uniform(min(n_sift, spos + quarter));
            for (bool more = true; more;) {
                uint32_t wc = 0u;
                while (spos < send && wc <= 64u) {
                    const uint32_t o = spos + lane;
                    const bool hit = o < send && B.over_tile[sec0 + o] == tile;
                    const unsigned long long m = __ballot(hit);
                    if (hit) olist[wv * 128u + wc + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = sec0 + o;
                    wc += (uint32_t)__popcll(m); spos += 64u;
                }
                if (lane == 0u) { ocnt[wv] = wc; omore[wv] = spos < send ? 1u : 0u; }
                __syncthreads();
                const uint32_t n = wave_uniform(ocnt[0] + ocnt[1] + ocnt[2] + ocnt[3]);
                more = (omore[0] | omore[1] | omore[2] | omore[3]) != 0u;
                if (n) do_batch(n, 0u, std::true_type());
                else __syncthreads();      // (this stretch of the section held other tiles' records only; ocnt / omore are rewritten next)
            }
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
        __syncthreads();      // (cur_unit is rewritten by the next claim)
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
void zr_launch_geom(const ZrPass& P, const ZrHiz& Z, const ZrTriBins& B, ZrDevStats* stats, int slot, bool count_only, hipStream_t s)
{
    const dim3 g(B.n_waves / 4u), b(256);
    if (count_only) {
        if (Z.phase == 2u) hipLaunchKernelGGL((k_geom<true, true>), g, b, 0, s, P, B.sel, Z, B, stats, slot);
        else hipLaunchKernelGGL((k_geom<false, true>), g, b, 0, s, P, B.sel, Z, B, stats, slot);
    } else if (Z.phase == 2u) hipLaunchKernelGGL((k_geom<true, false>), g, b, 0, s, P, B.sel, Z, B, stats, slot);
    else hipLaunchKernelGGL((k_geom<false, false>), g, b, 0, s, P, B.sel, Z, B, stats, slot);
}
void zr_launch_plan(const ZrTriBins& B, const uint32_t* owned_tiles, uint32_t n_owned, ZrDevStats* stats, bool exact, uint32_t bucket_pct, hipStream_t s)
{
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, s, B, owned_tiles, n_owned, ZR_TCHUNK * ZR_TBATCHES, stats, exact ? 1u : 0u, bucket_pct);
}
void zr_launch_tile(const ZrPass& P, const ZrTriBins& B, ZrDevStats* stats, int slot, unsigned long long* vis64, uint32_t n_blocks, hipStream_t s, bool last,
                    const uint32_t* owned_tiles, uint32_t n_owned)
{
    if (last) hipLaunchKernelGGL((k_tile<ZR_MODE_GBUFFER, true>), dim3(n_blocks), dim3(256), 0, s, P, B, stats, slot, vis64, owned_tiles, n_owned);
    else hipLaunchKernelGGL((k_tile<ZR_MODE_GBUFFER, false>), dim3(n_blocks), dim3(256), 0, s, P, B, stats, slot, vis64, owned_tiles, n_owned);
}
void zr_launch_sky_tiles(const ZrPass& P, const ZrObject* objs, const uint32_t* owned_tiles, uint32_t n_owned, unsigned long long* sky64, hipStream_t s)
{
    if (n_owned) hipLaunchKernelGGL(k_sky_tiles, dim3(n_owned), dim3(256), 0, s, P, objs, owned_tiles, sky64);
}
