// zr_shadow.hip — the shadow pass (ZE:3239-3393) as meshlet-level binning: k_bin_count / k_scan / k_bin_fill (per-tile lists of
// self-contained 32-byte meshlet records), k_raster_chunks<MODE, HIZ, DEFER, LATE> (persistent workgroups, a 44 x 44 key window in LDS),
// k_shadow_occlusion (the map as a running minimum hides casters), k_count_shadow.  See zr_dev.h for the map of the kernel files.
#include "zr_dev.h"
#include "zr_raster.h"

// Per-tile entry counts from the rects.  Counting goes through an LDS histogram per 1024 work items so that a hot
// tile costs one global atomic per workgroup instead of one per meshlet-instance (same-address atomics serialise).
__global__ __launch_bounds__(1024) void k_bin_count(ZrPass P, const uint32_t* __restrict__ work, uint32_t* __restrict__ rects,
                                                    uint32_t* __restrict__ tile_count, ZrHiz Z, ZrDevStats* __restrict__ stats, int slot)
{
    extern __shared__ uint32_t hist[];
    const uint32_t n_tiles = P.tiles_x * P.tiles_y;
    const int vslot = slot > 1 ? 1 : slot;               // camera rounds 1 and 2 share the cull results of slot 1
    const uint32_t n_vis = P.use_worklist ? stats->n_vis_work[vslot] : P.n_work;
    if (blockIdx.x * 1024u >= n_vis) return;            // the grid is sized for every meshlet-instance of the scene
    // (shadow pass: ownership of the map's tiles was decided per meshlet by k_cull_box - an accepted meshlet is listed in every tile of its rect)
    const bool shadow = P.mode == ZR_MODE_SHADOW;
    for (uint32_t i = threadIdx.x; i < n_tiles; i += 1024u) hist[i] = 0;
    __syncthreads();
    const uint32_t w = blockIdx.x * 1024u + threadIdx.x;
    uint32_t occluded = 0;
    if (w < n_vis) {
        uint32_t r = rects[w];
        if (r != ZR_RECT_CULLED && Z.phase) {            // two-pass occlusion culling: who is drawn in this round?
            const bool was_visible = Z.vis_prev[P.use_worklist ? work[w] : w] == (uint8_t)Z.vis_stamp;
            if (Z.phase == 1u) { if (!was_visible) r = ZR_RECT_CULLED; }
            else if (was_visible) r = ZR_RECT_CULLED;     // drawn in round 1
            else if (hiz_occluded(Z, Z.pxrect[w], Z.zmin[w])) { r = ZR_RECT_CULLED; rects[w] = r; occluded = 1; }
        }
        if (r != ZR_RECT_CULLED) {
            const uint32_t tx0 = r & 255u, ty0 = (r >> 8) & 255u, tx1 = (r >> 16) & 255u, ty1 = r >> 24;
            const float zt = tile_test_depth(Z, w);
            for (uint32_t ty = ty0; ty <= ty1; ++ty)
                for (uint32_t tx = tx0; tx <= tx1; ++tx) {
                    const uint32_t t = ty * P.tiles_x + tx;
                    if ((shadow || tile_owner(tx, ty, P.tile_world) == P.tile_rank) && !tile_hides(Z, zt, t)) atomicAdd(&hist[t], 1u);
                }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_tiles; i += 1024u) { const uint32_t c = hist[i]; if (c) atomicAdd(&tile_count[i], c); }
    // statistics: one global atomic per workgroup (same-address atomics serialise)
    __shared__ uint32_t tally;
    if (threadIdx.x == 0) tally = 0;
    __syncthreads();
    const uint32_t nocc = (uint32_t)__popcll(__ballot(occluded != 0));
    if ((threadIdx.x & 63u) == 0 && nocc) atomicAdd(&tally, nocc);
    __syncthreads();
    if (threadIdx.x == 0 && tally) atomicAdd(&stats->hiz_culled, tally);
}

// Exclusive scan of tile_count[0..n) into tile_offset[0..n] and of the per-tile work-unit counts ceil(count / chunk)
// into chunk_offset[0..n]; lays out the rasteriser's work units (tile, first entry, end, kind); zeroes tile_count and tile_cursor
// for the fill and resets the work counter.
__global__ __launch_bounds__(1024) void k_scan(uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_offset,
                                               uint32_t* __restrict__ tile_cursor, uint32_t* __restrict__ chunk_offset,
                                               uint4* __restrict__ chunk_tab, uint32_t chunk_cap,
                                               uint32_t n, uint32_t capacity, ZrDevStats* __restrict__ stats, int slot, uint32_t chunk)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t cpart[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t b = tid * per, e = min(n, b + per);
    uint32_t s = 0, cs = 0;
    for (uint32_t i = b; i < e; ++i) s += tile_count[i];
    part[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t v = (tid >= off) ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    for (uint32_t i = b; i < e; ++i) cs += (tile_count[i] + chunk - 1u) / chunk;
    cpart[tid] = cs;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t cv = (tid >= off) ? cpart[tid - off] : 0u;
        __syncthreads();
        cpart[tid] += cv;
        __syncthreads();
    }
    uint32_t run = part[tid] - s, crun = cpart[tid] - cs;
    for (uint32_t i = b; i < e; ++i) {
        const uint32_t c = tile_count[i];
        tile_offset[i] = run;
        chunk_offset[i] = crun;
        // one record per raster work unit: (tile, first entry, end, kind): the rasteriser finds its unit with one load, not a search
        const uint32_t nu = (c + chunk - 1u) / chunk;
        for (uint32_t k = 0; k < nu; ++k)
            if (crun + k < chunk_cap) chunk_tab[crun + k] = make_uint4(i, run + k * chunk, run + min(c, (k + 1u) * chunk), 0u);
        run += c; crun += nu;
        tile_count[i] = 0; tile_cursor[i] = 0;
    }
    if (slot == 0 && tid < 32u) stats->covered_part[tid] = 0;      // shadow pipeline: k_shadow_occlusion's tally (32 partial sums)
    if (tid == 1023) {
        tile_offset[n] = part[1023];
        chunk_offset[n] = cpart[1023];
        stats->bin_entries[slot] = part[1023];
        stats->n_chunks[slot] = min(cpart[1023], chunk_cap);
        stats->chunk_counter[slot] = 0;
        // what the kernels after this one accumulate for the pass starts from zero here: the shadow pipeline's block is not touched by
        // k_frame_begin (the pipeline does not wait for the camera lane)
        stats->survivors[slot] = 0; stats->n_slow[slot] = 0;
        if (slot == 0) stats->overflow = part[1023] > capacity ? 1u : 0u;      // the pipeline's own block: reset here.  (Camera slots - A/B builds -
        else if (part[1023] > capacity) stats->overflow = 1u;                  // share the lane's block: round 2 must not clear round 1's flag)
        if (slot == 0) { stats->n_chunks[1] = 0; stats->chunk_counter[1] = 0; stats->shadow_late = 0; }      // (k_shadow_occlusion's late units)
        if (part[1023] > capacity) stats->overflow_sticky = ZR_OVF_BINS;
    }
}

// Scatter meshlet-instance ids into the per-tile lists.  Same LDS aggregation as k_bin_count: the workgroup reserves
// a contiguous range per tile with one global atomic, then hands out slots from LDS.
__global__ __launch_bounds__(1024) void k_bin_fill(ZrPass P, const ZrObject* __restrict__ objs, const uint32_t* __restrict__ work,
                                                   const uint32_t* __restrict__ rects,
                                                   const uint32_t* __restrict__ tile_offset, uint32_t* __restrict__ tile_cursor,
                                                   ZrBinEntry* __restrict__ bins, ZrHiz Z, ZrDevStats* __restrict__ stats, int slot)
{
    extern __shared__ uint32_t hist[];
    __shared__ uint32_t tot;
    const uint32_t n_tiles = P.tiles_x * P.tiles_y;
    const int vslot = slot > 1 ? 1 : slot;
    const uint32_t n_vis = P.use_worklist ? stats->n_vis_work[vslot] : P.n_work;
    if (blockIdx.x * 1024u >= n_vis) return;
    const bool shadow = P.mode == ZR_MODE_SHADOW;       // (as in k_bin_count)
    for (uint32_t i = threadIdx.x; i < n_tiles; i += 1024u) hist[i] = 0;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    const uint32_t k = blockIdx.x * 1024u + threadIdx.x;
    uint32_t r = ZR_RECT_CULLED, w = 0;
    if (k < n_vis) {
        r = rects[k]; w = P.use_worklist ? work[k] : k;
        if (r != ZR_RECT_CULLED && Z.phase) {            // same split as k_bin_count (round 2's occluded items were marked CULLED there)
            const bool was_visible = Z.vis_prev[w] == (uint8_t)Z.vis_stamp;
            if ((Z.phase == 1u) != was_visible) r = ZR_RECT_CULLED;
        }
    }
    const uint32_t tx0 = r & 255u, ty0 = (r >> 8) & 255u, tx1 = (r >> 16) & 255u, ty1 = r >> 24;
    if (r != ZR_RECT_CULLED) {
        const float zt = tile_test_depth(Z, k);
        for (uint32_t ty = ty0; ty <= ty1; ++ty)
            for (uint32_t tx = tx0; tx <= tx1; ++tx) {
                const uint32_t t = ty * P.tiles_x + tx;
                if ((shadow || tile_owner(tx, ty, P.tile_world) == P.tile_rank) && !tile_hides(Z, zt, t)) atomicAdd(&hist[t], 1u);
            }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_tiles; i += 1024u) {
        const uint32_t c = hist[i];
        if (c) hist[i] = tile_offset[i] + atomicAdd(&tile_cursor[i], c);      // hist now holds the next free slot
    }
    __syncthreads();
    if (r != ZR_RECT_CULLED) {
        // decode the work id once; every tile of the rect gets the same self-contained record
        const ZrObject* __restrict__ O = objs + find_object_work(objs, (int)P.n_objects, w);
        const uint32_t local = w - O->work_base;
        const uint32_t inst_i = local / O->n_meshlets, m = local - inst_i * O->n_meshlets;
        const XkMeshlet* __restrict__ ml = O->meshlets + m;
        ZrBinEntry be;
        const uint4 mh = ld_global((const uint4*)ml);            // VertexOffset, VertexCount, TriangleOffset, TriangleCount
        be.mpos = O->mpos + mh.x; be.mtri = O->mtri + ld_global(&ml->BindlessContext); be.inst = O->inst + inst_i;
        be.counts = mh.y | mh.w << 8 | (O->instanced ? 1u << 16 : 0u);
        be.prim_base = O->prim_base + inst_i * O->n_tris;
        const float zt = tile_test_depth(Z, k);
        for (uint32_t ty = ty0; ty <= ty1; ++ty)
            for (uint32_t tx = tx0; tx <= tx1; ++tx) {
                const uint32_t t = ty * P.tiles_x + tx;
                if ((!shadow && tile_owner(tx, ty, P.tile_world) != P.tile_rank) || tile_hides(Z, zt, t)) continue;
                const uint32_t pos = atomicAdd(&hist[t], 1u);
                if (pos < P.bin_capacity) bins[pos] = be;
            }
    }
    const uint32_t cnt = (uint32_t)__popcll(__ballot(r != ZR_RECT_CULLED));
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd(&tot, cnt);
    __syncthreads();
    if (threadIdx.x == 0 && tot) atomicAdd(&stats->survivors[slot], tot);
}

// Persistent chunk rasteriser.  A chunk = up to ZR_CHUNK consecutive entries of ONE tile's bin list, so a hot tile is
// spread over many workgroups and the pass is bounded by total work, not by the fullest tile.  Every workgroup pulls
// chunk ids from one device counter until they run out (each wave reaches the exit test).  Per chunk: clear the
// tile's LDS keys, 4 waves rasterise the chunk's meshlets into them (ds_min), then the touched keys are merged into
// the frame-sized key buffer in HBM with global atomic min (skipped when the resident key already wins).
//   GBUFFER: vis64[W*H] (depth bits << 32 | prim), resolved later by k_resolve_gbuffer
//   SHADOW : the shadow map itself (float bits as uint): the merge IS the LESS_OR_EQUAL depth write
// DEFER: triangles that need the clipper (or the 64-bit walk) are not rasterised here but appended, with their tile, to `slow` for
// k_tile_slow: without the call to raster_clipped in its loop the kernel needs half the registers, i.e. twice the waves per SIMD
// fit - next to each other and next to the other lane's kernels.
// LATE (shadow pass, after k_shadow_occlusion): unit u is the ONE entry bins[bin_capacity - 1 - u], its tile in the record's prim_base
// (the shadow pass has no use for a primitive id); the units are counted in slot 1 of the pipeline's block, slow triangles stay in `slot`.
template <int MODE, bool HIZ, bool DEFER, bool LATE = false>
__global__ __launch_bounds__(RTHREADS) __attribute__((amdgpu_waves_per_eu(DEFER ? ZR_RASTER_WAVES_DEFER : ZR_RASTER_WAVES)))
void k_raster_chunks(ZrPass P, const ZrObject* __restrict__ objs, const uint4* __restrict__ chunk_tab,
                     const ZrBinEntry* __restrict__ bins, ZrDevStats* __restrict__ stats, int slot,
                     unsigned long long* __restrict__ vis64, uint32_t* __restrict__ shadow_bits,
                     const float* __restrict__ hiz0, uint32_t hiz0_w, uint32_t hiz0_h, uint4* __restrict__ slow, uint32_t slow_cap)
{
    __shared__ float hz[HIZ ? (TILE / 8) * (TILE / 8) : 1];
    __shared__ unsigned long long keys64[MODE == ZR_MODE_GBUFFER ? TILE_PIX : 1];
    __shared__ uint32_t keys32[MODE == ZR_MODE_SHADOW ? SPAN_PIX(MODE) : 1];
    __shared__ int4 vstage[RW][WAVE];
    // per-wave ring of surviving triangles, SoA: 3 x (tile-relative X | Y << 16, z) + prim.  Only small triangles (edges under 64 px)
    // that reach the tile are queued, so a relative coordinate lies within [-16384, 24576] sub-pixel units and fits 16 bits.
    __shared__ int queue[RW][7][QCAP];
    __shared__ uint32_t cur_chunk;

    // the wave index is made KNOWN-uniform: entry indices, bin records, the instance record and every pointer derived from them
    // then live in SGPRs and are fetched with scalar loads - some 30 VGPRs less in the hot loop (one more wave per SIMD, and this
    // kernel waits on dependent loads most of the time)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = wave_uniform(tid >> 6);
    const int cslot = LATE ? 1 : slot;
    const uint32_t n_chunks = LATE ? min(stats->n_chunks[1], P.bin_capacity - min(stats->bin_entries[slot], P.bin_capacity)) : stats->n_chunks[slot];

    // the first two chunks of a workgroup are its own index and that + the grid (no atomic: an empty pass costs nothing), later ones
    // come from the counter
    uint32_t chunk = blockIdx.x;
    bool first = true;
    for (;;) {
        if (chunk >= n_chunks) break;
        for (uint32_t i = tid; i < (uint32_t)SPAN_PIX(MODE); i += RTHREADS) {
            if (MODE == ZR_MODE_GBUFFER) keys64[i] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
            else keys32[i] = 0x3F800000u;
        }
        __syncthreads();
        // this work unit: (tile, first entry, end) as k_scan laid it out: one load, not a search over the tiles' chunk offsets
        uint4 ct;
        if (LATE) { const uint32_t idx = P.bin_capacity - 1u - chunk; ct = make_uint4(((const uint4*)bins)[2u * idx + 1u].w, idx, idx + 1u, 0u); }
        else ct = chunk_tab[chunk];
        const uint32_t tile = ct.x, beg = ct.y, end = min(ct.z, P.bin_capacity);
        // Everything below works in TILE-RELATIVE coordinates (origin = the tile's first pixel): edge functions, depth planes and
        // bounding boxes are built from coordinate differences, so the integers and floats are the ones absolute coordinates give.
        const int tpx0 = (int)(tile % P.tiles_x) * TILE, tpy0 = (int)(tile / P.tiles_x) * TILE;
        const int ox = tpx0 * 256, oy = tpy0 * 256;
        TileCtx T;
        T.px0 = 0; T.py0 = 0; T.W = (int)P.W - tpx0; T.H = (int)P.H - tpy0;
        if (HIZ) {      // this tile's finest pyramid texels (blocks past the target's edge hold no pixel: 0 = "hides everything")
            if (tid < (TILE / 8) * (TILE / 8)) {
                const uint32_t bx = (uint32_t)tpx0 / 8u + tid % (TILE / 8), by = (uint32_t)tpy0 / 8u + tid / (TILE / 8);
                hz[tid] = (bx < hiz0_w && by < hiz0_h) ? hiz0[(size_t)by * hiz0_w + bx] : 0.0f;
            }
            __syncthreads();
        }

        uint32_t qhead = 0, qn = 0;
        // The next entry's 32-byte record is fetched (vector loads, vmcnt-ordered) while the current one is processed; every
        // load of a meshlet then depends on that record alone.
        const uint4* __restrict__ rec = (const uint4*)bins;
        uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
        if (beg + wv < end) { n0 = rec[2u * (beg + wv)]; n1 = rec[2u * (beg + wv) + 1u]; }
        for (uint32_t e = beg + wv; e < end; e += RW) {
            const uint4 r0 = n0, r1 = n1;
            if (e + RW < end) { n0 = rec[2u * (e + RW)]; n1 = rec[2u * (e + RW) + 1u]; }
            const float4* __restrict__ mp = (const float4*)(((unsigned long long)wave_uniform(r0.y) << 32) | wave_uniform(r0.x));
            const uint2* __restrict__ tw = (const uint2*)(((unsigned long long)wave_uniform(r0.w) << 32) | wave_uniform(r0.z));
            const ZrInstance* __restrict__ ip = (const ZrInstance*)(((unsigned long long)wave_uniform(r1.y) << 32) | wave_uniform(r1.x));
            const uint32_t counts = wave_uniform(r1.z), pbase = wave_uniform(r1.w);
            const uint32_t vcount = counts & 255u, tcount = (counts >> 8) & 255u;
            const bool instanced = (counts >> 16) & 1u;

            // both rounds' triangle words and the vertex are requested together, before anything waits
            uint2 tri_w[2];
            tri_w[0] = lane < tcount ? ld_global(tw + lane) : make_uint2(0u, 0u);
            tri_w[1] = lane + WAVE < tcount ? ld_global(tw + lane + WAVE) : make_uint2(0u, 0u);
            const float4 pp = lane < vcount ? ld_global(mp + lane) : make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            const ZrInstance I = ld_record(ip);

            lds_fence();   // this wave's previous readers are done with its staging area
            // Almost every meshlet has no vertex outside the frustum at all: that is one wave-wide vote over ten comparisons (each a
            // 64-lane mask by itself); only a flagged meshlet pays for the per-vertex flag words and the per-triangle classification.
            bool flagged;
            {
                const zf4 c = zr_mat4_point(P.PVM, vs_position(zr3(pp.x, pp.y, pp.z), I, instanced));
                const float FM = 3.402823466e38f, gb = ZR_GUARD * c.w;
                const bool fin = __builtin_fabsf(c.x) <= FM && __builtin_fabsf(c.y) <= FM && __builtin_fabsf(c.z) <= FM && __builtin_fabsf(c.w) <= FM;
                const bool odd = !fin || c.x < -c.w || c.x > c.w || c.y < -c.w || c.y > c.w || c.z < 0.0f || c.z > c.w ||
                                 !(c.w > 0.0f) || __builtin_fabsf(c.x) > gb || __builtin_fabsf(c.y) > gb;
                flagged = __ballot(lane < vcount && odd) != 0ull;
                if (lane < vcount) {
                    const uint32_t f = flagged ? vertex_flags(c) : 0u;
                    SV s; s.X = 0; s.Y = 0; s.z = 0.0f; s.rw = 0.0f;
                    if (!(f & 129u)) s = project(c, P.hw, P.hh);
                    vstage[wv][lane] = make_int4(s.X - ox, s.Y - oy, (int)zr_f2u(s.z), (int)f);      // snapped x, y (tile-relative), depth, clip flags
                }
            }
            lds_fence();

            // phase 1: every triangle gets the cheap tests; survivors are compacted into this wave's LDS ring so that
            // phase 2 (setup + pixel walk) always runs with full lanes, across meshlet boundaries
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                const uint32_t t0 = (uint32_t)round * WAVE;
                if (t0 >= tcount || ZR_DIAG_SKIP(P.debug_skip) >= 2u) break;
                const uint32_t t = t0 + lane;
                bool alive = false;
                int4 r0 = make_int4(0, 0, 0, 0), r1 = r0, r2 = r0;
                const uint32_t prim = pbase + tri_w[round].y;
                if (t < tcount) {
                    const uint32_t i0 = tri_w[round].x & 255u, i1 = (tri_w[round].x >> 8) & 255u, i2 = (tri_w[round].x >> 16) & 255u;
                    r0 = vstage[wv][i0]; r1 = vstage[wv][i1]; r2 = vstage[wv][i2];
                    int cls = flagged ? classify((uint32_t)r0.w, (uint32_t)r1.w, (uint32_t)r2.w) : 1;
                    // a triangle with an edge of 64 pixels or more goes the clipper's way too: that route holds the 64-bit walk
                    // (it leaves a triangle that needs no clipping as it is, so the pixels are the same) - but only for the windows its
                    // snapped box reaches: a ground triangle under a 2048^2 map is met in thousands of tiles' lists and touches a few
                    if (cls == 1 && !tri_is_small(r0.x, r0.y, r1.x, r1.y, r2.x, r2.y))
                        cls = tri_prefilter<MODE, false>(r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, T) ? 2 : 0;
                    if (cls == 1) {
                        const float tz = HIZ ? __builtin_fminf(__builtin_fminf(zr_u2f((uint32_t)r0.z), zr_u2f((uint32_t)r1.z)), zr_u2f((uint32_t)r2.z)) : 0.0f;
                        alive = tri_prefilter<MODE, HIZ>(r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, T, tz, hz);
                    } else if (cls == 2) {
                        zf4 cc[3];
                        const uint32_t li[3] = { i0, i1, i2 };
                        for (int k = 0; k < 3; ++k) {
                            const float4 pk = ld_global(mp + li[k]);
                            cc[k] = zr_mat4_point(P.PVM, vs_position(zr3(pk.x, pk.y, pk.z), I, instanced));
                        }
                        if (DEFER) {
                            const uint32_t pos = atomicAdd(&stats->n_slow[slot], 1u);          // rare: one atomic apiece does
                            if (pos < slow_cap) {
                                for (int k = 0; k < 3; ++k) slow[4u * pos + (uint32_t)k] = make_uint4(zr_f2u(cc[k].x), zr_f2u(cc[k].y), zr_f2u(cc[k].z), zr_f2u(cc[k].w));
                                slow[4u * pos + 3u] = make_uint4(prim, tile, 0u, 0u);
                            } else { stats->overflow = 1u; stats->overflow_sticky = ZR_OVF_SLOW; }
                        } else raster_clipped<MODE>(cc[0], cc[1], cc[2], prim, T, P.hw, P.hh, ox, oy, keys64, keys32);
                    }
                }
                const unsigned long long mask = __ballot(alive);
                if (alive) {
                    const uint32_t slot = (qhead + qn + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))) & (QCAP - 1u);
                    int* q = &queue[wv][0][slot];
                    q[0 * QCAP] = (r0.x & 0xFFFF) | (r0.y << 16); q[1 * QCAP] = r0.z;
                    q[2 * QCAP] = (r1.x & 0xFFFF) | (r1.y << 16); q[3 * QCAP] = r1.z;
                    q[4 * QCAP] = (r2.x & 0xFFFF) | (r2.y << 16); q[5 * QCAP] = r2.z;
                    q[6 * QCAP] = (int)prim;
                }
                qn += (uint32_t)__popcll(mask);
                if (ZR_DIAG_SKIP(P.debug_skip) >= 1u) { qhead = (qhead + qn) & (QCAP - 1u); qn = 0; }
                if (qn >= WAVE) {
                    lds_fence();
                    const int* q = &queue[wv][0][(qhead + lane) & (QCAP - 1u)];
                    SV a, b, c;
                    a.X = (short)q[0 * QCAP]; a.Y = q[0 * QCAP] >> 16; a.z = zr_u2f((uint32_t)q[1 * QCAP]); a.rw = 0.0f;
                    b.X = (short)q[2 * QCAP]; b.Y = q[2 * QCAP] >> 16; b.z = zr_u2f((uint32_t)q[3 * QCAP]); b.rw = 0.0f;
                    c.X = (short)q[4 * QCAP]; c.Y = q[4 * QCAP] >> 16; c.z = zr_u2f((uint32_t)q[5 * QCAP]); c.rw = 0.0f;
                    raster_sub<MODE, true>(a, b, c, (uint32_t)q[6 * QCAP], T, keys64, keys32);
                    qhead = (qhead + WAVE) & (QCAP - 1u); qn -= WAVE;
                }
            }
        }
        if (qn) {      // flush the tail of this chunk
            lds_fence();
            if (lane < qn) {
                const int* q = &queue[wv][0][(qhead + lane) & (QCAP - 1u)];
                SV a, b, c;
                a.X = (short)q[0 * QCAP]; a.Y = q[0 * QCAP] >> 16; a.z = zr_u2f((uint32_t)q[1 * QCAP]); a.rw = 0.0f;
                b.X = (short)q[2 * QCAP]; b.Y = q[2 * QCAP] >> 16; b.z = zr_u2f((uint32_t)q[3 * QCAP]); b.rw = 0.0f;
                c.X = (short)q[4 * QCAP]; c.Y = q[4 * QCAP] >> 16; c.z = zr_u2f((uint32_t)q[5 * QCAP]); c.rw = 0.0f;
                raster_sub<MODE, true>(a, b, c, (uint32_t)q[6 * QCAP], T, keys64, keys32);
            }
            qhead = (qhead + qn) & (QCAP - 1u); qn = 0;
        }
        __syncthreads();

        // merge the touched keys into HBM
        for (uint32_t i = tid; i < (uint32_t)SPAN_PIX(MODE); i += RTHREADS) {
            const int px = tpx0 + (int)(i % (uint32_t)SPAN(MODE)), py = tpy0 + (int)(i / (uint32_t)SPAN(MODE));
            if (px >= (int)P.W || py >= (int)P.H) continue;
            const size_t p = (size_t)py * P.W + (size_t)px;
            if (MODE == ZR_MODE_GBUFFER) {
                const unsigned long long k = keys64[i];
                if ((uint32_t)k != ZR_EMPTY_PRIM && k < vis64[p]) atomicMin(&vis64[p], k);
            } else {
                // (a key still at its clear value 1.0 cannot lower the map, whose texels never exceed 1.0: its texel is not even read - the
                // untouched four fifths of a window's 1 936 texels were 15 MB of the pass's 25 MB of reads, profiles/r06_shadow_tcc.txt)
                const uint32_t k = keys32[i];
                if (k != 0x3F800000u && k < shadow_bits[p]) atomicMin(&shadow_bits[p], k);
            }
        }
        // (the next unit is claimed only when this one is done: claiming early costs more in tail balance than the atomic's latency)
        // ... and the second unit of a workgroup is fixed like the first (b + grid): claims on one counter queue up for ~10 ns apiece
        if (first) { first = false; __syncthreads(); chunk += gridDim.x; continue; }
        if (tid == 0) cur_chunk = 2u * gridDim.x + atomicAdd(&stats->chunk_counter[cslot], 1u);
        __syncthreads();   // keys are re-cleared at the top of the loop
        chunk = cur_chunk;
    }
}

// Occlusion culling of the shadow pass.  The map is a running minimum: a meshlet-instance whose least possible depth lies behind EVERY texel
// its box can reach, at any moment of the pass, cannot change the map - then or later - and need not be drawn; what is drawn is the same
// whatever was left out, so the map is bit for bit the one the full pass writes.  Which ones to try first is a guess taken from the
// previous frame (one byte per work item): the rasteriser's first launch draws the flagged ones (everything, on a scene's first frame),
// then this kernel tests EVERY survivor of the cull against the map as it stands (box and least depth from k_cull_box: conservative, the
// camera pass's Hi-Z bounds), flags "not hidden" for the next frame, and hands the unflagged ones that are not hidden to a late launch
// of the rasteriser.  A light or a scene that moves costs late work, never a wrong texel.
// A wave takes 64 survivors of the cull, a lane each for the item's record (box, least depth, flag), and writes their flags with ONE store
// (a byte stored per item by whichever lane happened to test it is a partial write of a cache line that lanes of other waves write too:
// 110 000 of those took 150 us).  The texels are read by TASKS: one per (item, row of its box), dealt out to the lanes by a prefix sum
// over the boxes' heights, so a 4 x 4 box costs 4 lane-loads and a 40 x 40 one 400, whatever mix a wave meets.  A task loads its row in
// spans of 4 texels (the map's rows are 4-byte aligned, nothing more is asked of a global load), masks what lies beyond the box's right
// edge, and folds its maximum into the item's word in LDS (ds_max).
struct __attribute__((packed, aligned(4))) ZrTexel4 { uint32_t x, y, z, w; };      // four texels of a map row, from any texel on
template <bool WORKLIST>
__global__ __launch_bounds__(256) void k_shadow_occlusion(ZrPass P, const ZrObject* __restrict__ objs, const uint32_t* __restrict__ work,
                                                          const uint32_t* __restrict__ rects, const uint2* __restrict__ pxrect,
                                                          const float* __restrict__ zmin, uint8_t* __restrict__ flags,
                                                          const uint32_t* __restrict__ shadow_bits, ZrBinEntry* __restrict__ bins,
                                                          ZrDevStats* __restrict__ stats, uint32_t retest)
{
    // retest: which quarter of the FLAGGED items is tested this frame (work id + retest divisible by 4; >= 4: all of them, a scene's first
    // frame).  A flagged item was drawn by the first launch whatever the test says - the test only decides whether it is drawn again next
    // frame - so it can wait up to three frames; an unflagged item is tested every frame (it is drawn if the test does not hide it).
    __shared__ uint32_t s_first[4][WAVE], s_far[4][WAVE];      // per wave: an item's first task, the farthest texel of its box so far
    __shared__ uint2 s_box[4][WAVE];
    // A workgroup takes 1 024 consecutive work items and first lists the survivors of the cull among them (a quarter, at 1 M instances):
    // the waves then work on full sets of 64 survivors.  (A wave's stretch is a chain of three or four dependent round trips to memory, and
    // 8 waves per SIMD is all there is to hide it: with the dead items in the lanes the kernel took 300 us for 11 M items, 2.75 M alive.)
    __shared__ uint32_t s_live[1024], s_nlive;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t n = WORKLIST ? stats->n_vis_work[0] : P.n_work;
    uint32_t n_occl = 0, n_late = 0;
    for (uint32_t blk = blockIdx.x * 1024u; blk < n; blk += gridDim.x * 1024u) {
      if (threadIdx.x == 0u) s_nlive = 0u;
      __syncthreads();
      {
        uint32_t rr[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) { const uint32_t k = blk + j * 256u + threadIdx.x; rr[j] = k < n ? rects[k] : ZR_RECT_CULLED; }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const unsigned long long m = __ballot(rr[j] != ZR_RECT_CULLED);
            uint32_t at = 0;
            if (lane == 0u && m) at = atomicAdd(&s_nlive, (uint32_t)__popcll(m));
            at = lane_bcast(at, 0u);
            if (rr[j] != ZR_RECT_CULLED) s_live[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = blk + j * 256u + threadIdx.x;
        }
      }
      __syncthreads();
      const uint32_t n_live = s_nlive;
      for (uint32_t base = wv * 64u; base < n_live; base += 256u) {
        const bool have = base + lane < n_live;
        const uint32_t k = have ? s_live[base + lane] : 0u;
        uint32_t r = ZR_RECT_CULLED, w = 0, zb = 0x80000000u;
        uint2 pr = make_uint2(0u, 0u);
        bool flagged = true;
        if (have) {
            r = rects[k]; pr = pxrect[k]; zb = zr_f2u(zmin[k]); w = WORKLIST ? work[k] : k;
            flagged = flags[w] != 0;
        }
        const bool live = r != ZR_RECT_CULLED;
        const uint32_t x0 = pr.x & 0xFFFFu, y0 = pr.x >> 16, x1 = pr.y & 0xFFFFu, y1 = pr.y >> 16;
        // (zmin < 0: the box touches the near plane or the guard band, or is not finite: drawn, never tested; boxes wider than 64 texels neither)
        const bool due = !flagged || retest >= 4u || ((w + retest) & 3u) == 0u;
        const bool test = live && due && (int)zb >= 0 && x1 - x0 < 64u && y1 - y0 < 64u;
        const uint32_t rows = test ? y1 - y0 + 1u : 0u;
        const uint32_t incl = wave_incl_scan(rows), total = lane_bcast(incl, 63u);
        lds_fence();      // the previous stretch's readers are done
        s_first[wv][lane] = incl - rows; s_far[wv][lane] = 0u; s_box[wv][lane] = pr;
        lds_fence();
        for (uint32_t t0 = 0; t0 < total; t0 += 64u) {
            const uint32_t t = t0 + lane;
            if (t < total) {
                // the item this task belongs to: the last one whose first task is <= t (items without rows share their successor's first task)
                uint32_t i = 0;
#pragma unroll
                for (uint32_t step = 32u; step; step >>= 1) if (s_first[wv][i + step] <= t) i += step;
                const uint2 b = s_box[wv][i];
                const uint32_t bx0 = b.x & 0xFFFFu, bx1 = b.y & 0xFFFFu, y = (b.x >> 16) + (t - s_first[wv][i]);
                const uint32_t* __restrict__ row = shadow_bits + (size_t)y * P.W;
                uint32_t far = 0;
                for (uint32_t x = bx0; x <= bx1; x += 4u) {
                    const uint32_t xs = min(x, P.W - 4u);
                    const ZrTexel4 v = *(const ZrTexel4*)(row + xs);
                    if (xs >= bx0 && xs <= bx1) far = max(far, v.x);
                    if (xs + 1u >= bx0 && xs + 1u <= bx1) far = max(far, v.y);
                    if (xs + 2u >= bx0 && xs + 2u <= bx1) far = max(far, v.z);
                    if (xs + 3u >= bx0 && xs + 3u <= bx1) far = max(far, v.w);
                }
                atomicMax(&s_far[wv][i], far);
            }
        }
        lds_fence();
        const bool hidden = test && zb > s_far[wv][lane];        // depth bits of [0, 1]: ordered as integers
        if (live) flags[w] = hidden ? 0u : 1u;
        if (live && !flagged && hidden) ++n_occl;
        if (live && !flagged && !hidden) {
            // late: one self-contained record per (tile, meshlet-instance), as k_bin_fill writes them, with the tile in prim_base
            ++n_late;
            const ZrObject* __restrict__ O = objs + find_object_work(objs, (int)P.n_objects, w);
            const uint32_t local = w - O->work_base;
            const uint32_t inst_i = local / O->n_meshlets, m = local - inst_i * O->n_meshlets;
            const XkMeshlet* __restrict__ ml = O->meshlets + m;
            ZrBinEntry be;
            const uint4 mh = ld_global((const uint4*)ml);            // VertexOffset, VertexCount, TriangleOffset, TriangleCount
            be.mpos = O->mpos + mh.x; be.mtri = O->mtri + ld_global(&ml->BindlessContext); be.inst = O->inst + inst_i;
            be.counts = mh.y | mh.w << 8 | (O->instanced ? 1u << 16 : 0u);
            const uint32_t room = P.bin_capacity - min(stats->bin_entries[0], P.bin_capacity);      // above the first launch's entries
            const uint32_t tx0 = r & 255u, ty0 = (r >> 8) & 255u, tx1 = (r >> 16) & 255u, ty1 = r >> 24;
            for (uint32_t ty = ty0; ty <= ty1; ++ty)
                for (uint32_t tx = tx0; tx <= tx1; ++tx) {
                    const uint32_t pos = atomicAdd(&stats->n_chunks[1], 1u);      // (late entries are few; a light that jumps pays ~10 ns apiece here)
                    be.prim_base = ty * P.tiles_x + tx;
                    if (pos < room) bins[P.bin_capacity - 1u - pos] = be;
                    else { stats->overflow = 1u; stats->overflow_sticky = ZR_OVF_LATE; }
                }
        }
      }
      __syncthreads();      // the list is rewritten by the next stretch
    }
    // the tally in 32 partial sums (the shadow pipeline's block has no other use for covered_part; zr_finish adds them up): one atomic per
    // workgroup on ONE word queued up for ~10 ns apiece - 17 us for config 3's 1 719 workgroups
    n_occl = (uint32_t)wave_sum((int)n_occl); n_late = (uint32_t)wave_sum((int)n_late);
    if (lane == 0u) {
        if (n_occl) atomicAdd(&stats->covered_part[(blockIdx.x * 4u + wv) & 31u], n_occl);
        if (n_late) atomicAdd(&stats->shadow_late, n_late);
    }
}

// statistics only (not part of the frame): shadow-map texels with depth < 1
__global__ void k_count_shadow(const uint32_t* __restrict__ bits, size_t n, ZrDevStats* __restrict__ stats)
{
    uint32_t c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += bits[i] != 0x3F800000u;
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(&stats->covered_shadow, c);
}

// ------------------------------------------------------------------------------------------------ launchers (C++ linkage, used by zr_host.cpp)

void zr_launch_bin_count(const ZrPass& P, const uint32_t* work, uint32_t* rects, uint32_t* tile_count, const ZrHiz& Z, ZrDevStats* stats,
                         int slot, hipStream_t s)
{
    if (P.n_work == 0) return;
    const uint32_t n_tiles = P.tiles_x * P.tiles_y;
    hipLaunchKernelGGL(k_bin_count, dim3((P.n_work + 1023) / 1024), dim3(1024), n_tiles * sizeof(uint32_t), s, P, work, rects, tile_count, Z, stats, slot);
}
void zr_launch_scan(uint32_t* tile_count, uint32_t* tile_offset, uint32_t* tile_cursor, uint32_t* chunk_offset, uint4* chunk_tab,
                    uint32_t chunk_cap, uint32_t n, uint32_t capacity, ZrDevStats* stats, int slot, hipStream_t s,
                    uint32_t chunk)
{
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, tile_count, tile_offset, tile_cursor, chunk_offset, chunk_tab, chunk_cap, n, capacity, stats, slot, chunk);
}
void zr_launch_bin_fill(const ZrPass& P, const ZrObject* objs, const uint32_t* work, const uint32_t* rects, const uint32_t* tile_offset,
                        uint32_t* tile_cursor, ZrBinEntry* bins, const ZrHiz& Z, ZrDevStats* stats, int slot, hipStream_t s)
{
    if (P.n_work == 0) return;
    const uint32_t n_tiles = P.tiles_x * P.tiles_y;
    hipLaunchKernelGGL(k_bin_fill, dim3((P.n_work + 1023) / 1024), dim3(1024), n_tiles * sizeof(uint32_t), s, P, objs, work, rects,
                       tile_offset, tile_cursor, bins, Z, stats, slot);
}
void zr_launch_raster_chunks(const ZrPass& P, const ZrObject* objs, const uint4* chunk_tab,
                             const ZrBinEntry* bins, ZrDevStats* stats, int slot, unsigned long long* vis64, uint32_t* shadow_bits,
                             uint32_t n_blocks, const ZrHiz& Z, hipStream_t s, uint4* slow, uint32_t slow_cap, const uint32_t* tiles, uint32_t n_tiles, int stage)
{
    const float* none = nullptr;
#ifdef ZR_DIAG      // the camera pass through this rasteriser: A/B builds only (ZR_FLAG_MESHLET_BINS)
    if (P.mode == ZR_MODE_GBUFFER && Z.phase == 2u) {
        hipLaunchKernelGGL((k_raster_chunks<ZR_MODE_GBUFFER, true, false>), dim3(n_blocks), dim3(RTHREADS), 0, s, P, objs, chunk_tab, bins, stats, slot, vis64, shadow_bits, (const float*)Z.lvl[0], Z.hw[0], Z.hh[0], (uint4*)nullptr, 0u);
        return;
    }
    if (P.mode == ZR_MODE_GBUFFER) {
        hipLaunchKernelGGL((k_raster_chunks<ZR_MODE_GBUFFER, false, false>), dim3(n_blocks), dim3(RTHREADS), 0, s, P, objs, chunk_tab, bins, stats, slot, vis64, shadow_bits, none, 0u, 0u, (uint4*)nullptr, 0u);
        return;
    }
#endif
    if (P.mode == ZR_MODE_GBUFFER) return;      // (not reached: the product's camera pass is triangle-binned)
    if (slow) {    // shadow pass: clipped triangles go through a list + k_tile_slow
        if (stage == 2) hipLaunchKernelGGL((k_raster_chunks<ZR_MODE_SHADOW, false, true, true>), dim3(n_blocks), dim3(RTHREADS), 0, s, P, objs, chunk_tab, bins, stats, slot, vis64, shadow_bits, none, 0u, 0u, slow, slow_cap);
        else hipLaunchKernelGGL((k_raster_chunks<ZR_MODE_SHADOW, false, true>), dim3(n_blocks), dim3(RTHREADS), 0, s, P, objs, chunk_tab, bins, stats, slot, vis64, shadow_bits, none, 0u, 0u, slow, slow_cap);
        if (n_tiles && stage != 1) hipLaunchKernelGGL((k_tile_slow<ZR_MODE_SHADOW, true>), dim3(std::min<uint32_t>(n_tiles, ZR_SLOW_BLOCKS)), dim3(256), 0, s, P, tiles, n_tiles, slow, slow_cap, stats, slot,
                                        (unsigned long long*)nullptr, shadow_bits, (const uint32_t*)nullptr, 0u);
    } else if (stage == 2)
        hipLaunchKernelGGL((k_raster_chunks<ZR_MODE_SHADOW, false, false, true>), dim3(n_blocks), dim3(RTHREADS), 0, s, P, objs, chunk_tab, bins, stats, slot, vis64, shadow_bits, none, 0u, 0u, (uint4*)nullptr, 0u);
    else
        hipLaunchKernelGGL((k_raster_chunks<ZR_MODE_SHADOW, false, false>), dim3(n_blocks), dim3(RTHREADS), 0, s, P, objs, chunk_tab, bins, stats, slot, vis64, shadow_bits, none, 0u, 0u, (uint4*)nullptr, 0u);
}
void zr_launch_shadow_occlusion(const ZrPass& P, const ZrObject* objs, const uint32_t* work, const uint32_t* rects, const uint2* pxrect,
                                const float* zmin, uint8_t* flags, const uint32_t* shadow_bits, ZrBinEntry* bins, ZrDevStats* stats,
                                uint32_t n_blocks, uint32_t retest, hipStream_t s)
{
    if (P.n_work == 0) return;
    const dim3 g(std::min<uint32_t>((P.n_work + 1023u) / 1024u, n_blocks)), b(256);
    if (P.use_worklist) hipLaunchKernelGGL(k_shadow_occlusion<true>, g, b, 0, s, P, objs, work, rects, pxrect, zmin, flags, shadow_bits, bins, stats, retest);
    else hipLaunchKernelGGL(k_shadow_occlusion<false>, g, b, 0, s, P, objs, work, rects, pxrect, zmin, flags, shadow_bits, bins, stats, retest);
}
void zr_launch_count_shadow(const uint32_t* bits, size_t n, ZrDevStats* stats, hipStream_t s)
{
    hipLaunchKernelGGL(k_count_shadow, dim3(256), dim3(256), 0, s, bits, n, stats);
}
