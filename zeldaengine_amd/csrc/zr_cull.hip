// zr_cull.hip — scene preparation and culling: k_instance_prep (XkInstanceData -> ZrInstance, once per zr_object_add), k_cull_instances
// (big scenes: whole-mesh sphere vs frustum per instance -> compacted work list), k_cull_box<MODE> (lane per meshlet-instance: frustum,
// normal cone, owned region, then the 8 corners of the meshlet's box -> a tile rect / pixel box / least depth that BOUND the exact ones;
// compacts the camera pass's round-1 list) and, in -DZR_DIAG builds, the exact wave-per-survivor k_cull<MODE>.  See zr_dev.h for the map.
#include "zr_dev.h"

// ------------------------------------------------------------------------------------------------ instance prep

// MakeRotMatrix (SH/Common.glsl:60-87): rotMat = mz * my * mx; mx(R.x) turns about Y, my(R.y) about Z, mz(R.z) about X
__global__ void k_instance_prep(const XkInstanceData* __restrict__ in, ZrInstance* __restrict__ out, uint32_t n, uint32_t instanced)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ZrInstance I;
    if (!instanced) {
        for (int k = 0; k < 9; ++k) I.R[k] = (k % 4 == 0) ? 1.0f : 0.0f;
        I.t[0] = I.t[1] = I.t[2] = 0.0f; I.s = 1.0f;
    } else {
        XkInstanceData d = in[i];
        float s, c, mx[9], my[9], mz[9], t[9];
        zr_sincos(d.InstanceRotation[0], s, c);
        mx[0] = c; mx[1] = 0; mx[2] = s;  mx[3] = 0; mx[4] = 1; mx[5] = 0;  mx[6] = -s; mx[7] = 0; mx[8] = c;
        zr_sincos(d.InstanceRotation[1], s, c);
        my[0] = c; my[1] = s; my[2] = 0;  my[3] = -s; my[4] = c; my[5] = 0;  my[6] = 0; my[7] = 0; my[8] = 1;
        zr_sincos(d.InstanceRotation[2], s, c);
        mz[0] = 1; mz[1] = 0; mz[2] = 0;  mz[3] = 0; mz[4] = c; mz[5] = s;  mz[6] = 0; mz[7] = -s; mz[8] = c;
        zr_mat3_mul(mz, my, t);
        zr_mat3_mul(t, mx, I.R);
        I.t[0] = d.InstancePosition[0]; I.t[1] = d.InstancePosition[1]; I.t[2] = d.InstancePosition[2];
        I.s = d.InstancePScale;
    }
    I._pad[0] = I._pad[1] = I._pad[2] = 0.0f;
    out[i] = I;
}

__device__ __forceinline__ int find_object_inst(const ZrObject* __restrict__ objs, int n, uint32_t g)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (objs[mid].inst_base <= g) lo = mid; else hi = mid - 1; }
    return lo;
}

// Multi-GPU: can a bounding sphere (centre `co` after the instance transform, radius ri there) reach a tile this rank owns?
// View space: centre (x, y, -d), radius r, d > r (the eye is outside, the whole sphere in front of it).  In the x-d plane the lines
// through the eye tangent to the circle have slopes (x d +- r sqrt(x^2 + d^2 - r^2)) / (d^2 - r^2): every point of the sphere
// projects between them; ndc = Proj[0][0] * slope (Proj[1][1] for y).  Conservative: the radius is rounded up, a slack covers the
// arithmetic here and in the rasteriser's own transform, the pixel range gets a margin of one.  "true" whenever in doubt.
__device__ __forceinline__ bool sphere_screen(const ZrPass& P, zf3 co, float ri, float& sx0, float& sy0, float& sx1, float& sy1, float& d_near)
{   // false: no bound (the eye is inside, or the numbers are out of range).  Screen-space extent (pixels, y down) of the sphere.
    const zf4 cv = zr_mat4_point(P.VM, co);
    const float d = -cv.z;
    const float r = __builtin_fmaf(ri, 1.003f, 1e-6f * (__builtin_fabsf(cv.x) + __builtin_fabsf(cv.y) + __builtin_fabsf(d)) + 1e-30f);
    const float den = __builtin_fmaf(d, d, -(r * r));
    if (!(d > r && den > 0.0f && d < 3.0e18f)) return false;
    const float tx = r * __builtin_sqrtf(__builtin_fmaxf(__builtin_fmaf(cv.x, cv.x, den), 0.0f));
    const float ty = r * __builtin_sqrtf(__builtin_fmaxf(__builtin_fmaf(cv.y, cv.y, den), 0.0f));
    const float ax = ((cv.x * d - tx) / den) * P.p00, bx = ((cv.x * d + tx) / den) * P.p00;
    const float ay = ((cv.y * d - ty) / den) * P.p11, by = ((cv.y * d + ty) / den) * P.p11;
    float nx0 = __builtin_fminf(ax, bx), nx1 = __builtin_fmaxf(ax, bx), ny0 = __builtin_fminf(ay, by), ny1 = __builtin_fmaxf(ay, by);
    const float sl = 1e-5f;
    nx0 -= sl * (1.0f + __builtin_fabsf(nx0)); nx1 += sl * (1.0f + __builtin_fabsf(nx1));
    ny0 -= sl * (1.0f + __builtin_fabsf(ny0)); ny1 += sl * (1.0f + __builtin_fabsf(ny1));
    sx0 = __builtin_fmaf(nx0, P.hw, P.hw); sx1 = __builtin_fmaf(nx1, P.hw, P.hw);
    sy0 = __builtin_fmaf(ny0, P.hh, P.hh); sy1 = __builtin_fmaf(ny1, P.hh, P.hh);
    if (!(sx0 >= -1.0e9f && sx1 <= 1.0e9f && sy0 >= -1.0e9f && sy1 <= 1.0e9f)) return false;      // NaN or huge
    d_near = d - r;
    return true;
}
__device__ __forceinline__ bool sphere_bounds(const ZrPass& P, zf3 co, float ri, int& px0, int& py0, int& px1, int& py1, float& d_near)
{   // false: no bound.  An empty box (px0 > px1 or py0 > py1) = off the target.  One pixel of margin on every side.
    float sx0, sy0, sx1, sy1;
    if (!sphere_screen(P, co, ri, sx0, sy0, sx1, sy1, d_near)) return false;
    px0 = max(0, (int)__builtin_floorf(sx0) - 1); px1 = min((int)P.W - 1, (int)__builtin_floorf(sx1) + 1);
    py0 = max(0, (int)__builtin_floorf(sy0) - 1); py1 = min((int)P.H - 1, (int)__builtin_floorf(sy1) + 1);
    return true;
}
// The sphere's screen extent holds no pixel centre of the target (centre i lies at i + 0.5; 1/32 pixel of slack covers the snapping
// of vertices to 1/256 pixel and the rasteriser's own rounding): nothing inside it can produce a fragment.
__device__ __forceinline__ bool sphere_holds_no_centre(const ZrPass& P, zf3 co, float ri)
{
    float sx0, sy0, sx1, sy1, dn;
    if (!sphere_screen(P, co, ri, sx0, sy0, sx1, sy1, dn)) return false;
    const int px0 = max(0, (int)__builtin_ceilf(sx0 - 0.53125f)), px1 = min((int)P.W - 1, (int)__builtin_floorf(sx1 - 0.46875f));
    const int py0 = max(0, (int)__builtin_ceilf(sy0 - 0.53125f)), py1 = min((int)P.H - 1, (int)__builtin_floorf(sy1 - 0.46875f));
    return px0 > px1 || py0 > py1;
}
// the tiles a meshlet with the pixel box (px0, py0)-(px1, py1) is listed for (shadow pass: see ZR_SHADOW_APRON)
template <int MODE>
__device__ __forceinline__ uint32_t pack_tile_rect(int px0, int py0, int px1, int py1)
{
    if (MODE == ZR_MODE_SHADOW) { px1 = max(px0, px1 - ZR_SHADOW_APRON); py1 = max(py0, py1 - ZR_SHADOW_APRON); }
    return (uint32_t)(px0 / TILE) | (uint32_t)(py0 / TILE) << 8 | (uint32_t)(px1 / TILE) << 16 | (uint32_t)(py1 / TILE) << 24;
}
__device__ __forceinline__ bool sphere_reaches_owned_tile(const ZrPass& P, zf3 co, float ri)
{
    int px0, py0, px1, py1; float dn;
    if (!sphere_bounds(P, co, ri, px0, py0, px1, py1, dn)) return true;
    if (px0 > px1 || py0 > py1) return false;                                                    // off the target altogether
    const uint32_t sh = 5u + ZR_SUPERTILE_SHIFT;      // TILE == 32 pixels: pixel -> super-tile (checked where rect_cull is set)
    for (uint32_t sy = (uint32_t)py0 >> sh; sy <= (uint32_t)py1 >> sh; ++sy)
        for (uint32_t sx = (uint32_t)px0 >> sh; sx <= (uint32_t)px1 >> sh; ++sx)
            if ((sx + sy * ZR_SUPERTILE_SKEW) % P.tile_world == P.tile_rank) return true;
    return false;
}

// Level 1 of the cull hierarchy: one lane per instance, whole-mesh bounding sphere against the frustum (same inflated
// bounds as the meshlet test, so it is conservative).  The meshlet-instances of the surviving instances are appended to
// work[]; one atomic per wave reserves the range.  Also applies the shadow-pass filters (skydome, instance partition).
#define ZR_CI_THREADS 1024u
#define ZR_CI_PER 4u                        // instances per thread: one reservation per 4 096 instances
// instance g (global ordinal) against the pass's instance-level tests; nm / wbase: its meshlet-instances, co / radius: its bounding sphere
// after the instance transform (object space of PVM)
template <int MODE>
__device__ __forceinline__ bool instance_test(const ZrPass& P, const ZrObject* __restrict__ objs, uint32_t g, uint32_t& nm, uint32_t& wbase, zf3& co, float& radius)
{
    const ZrObject* __restrict__ O = objs + find_object_inst(objs, (int)P.n_objects, g);
    const uint32_t inst_i = g - O->inst_base;
    bool vis = true;
    co = zr3(0.0f, 0.0f, 0.0f); radius = 0.0f;
    // the skydome is not a shadow caster (ZE:4709-4720); with N GPUs each draws every N-th instance into its own copy of
    // the shadow map and the copies are min-reduced (depth test LESS_OR_EQUAL is a min, so the split is exact)
    if (MODE == ZR_MODE_SHADOW && ((O->flags & ZR_OBJ_SKY) || inst_i % P.inst_world != P.inst_rank)) vis = false;
    if (vis && (P.frustum_ok | P.rect_cull | P.sphere_ok)) {
        const ZrInstance I = ld_record(O->inst + inst_i);
        const bool instanced = O->instanced != 0;
        co = vs_position(zr3(O->mesh_center[0], O->mesh_center[1], O->mesh_center[2]), I, instanced);
        radius = O->mesh_radius * (instanced ? __builtin_fabsf(I.s) : 1.0f);
        const zf4 cw4 = zr_mat4_point(P.M, co);
        float rw = radius * P.m_scale;
        rw = __builtin_fmaf(rw, 1.001f, 1e-5f * (__builtin_fabsf(cw4.x) + __builtin_fabsf(cw4.y) + __builtin_fabsf(cw4.z) + 1.0f));
        for (int k = 0; k < 6 && P.frustum_ok; ++k) {
            const float d = __builtin_fmaf(P.planes[k][0], cw4.x, __builtin_fmaf(P.planes[k][1], cw4.y,
                            __builtin_fmaf(P.planes[k][2], cw4.z, P.planes[k][3])));
            if (d < -rw) vis = false;
        }
        // (shadow pass with the MAP owned by light-space super-tiles, zr_set_shadow_tiles: the same reject against the map's tiles)
        if (vis && P.rect_cull && !sphere_reaches_owned_tile(P, co, radius)) vis = false;
        // a whole instance between the pixel (texel) centres: a million instances under a 1024^2 shadow map are mostly that
        if (vis && P.sphere_ok && P.frustum_ok && sphere_holds_no_centre(P, co, radius)) vis = false;
    }
    nm = O->n_meshlets; wbase = O->work_base + inst_i * nm;
    return vis;
}
// workgroup-wide (ZR_CI_THREADS) exclusive scan of per-thread counts + ONE global reservation on *counter; returns this thread's offset.
// (A returning atomic per wave on one address: 15 600 of them at a million instances queued up for 0.7 ms.)
__device__ __forceinline__ uint32_t block_reserve(uint32_t mine, uint32_t* __restrict__ counter, uint32_t* wsum, uint32_t* base_s)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += v; }
    __syncthreads();                                      // (wsum / base_s may still be read from a previous call)
    if (lane == 63u) wsum[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t i = 0; i < ZR_CI_THREADS / 64u; ++i) { const uint32_t t = wsum[i]; wsum[i] = tot; tot += t; }
        *base_s = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    return *base_s + wsum[wv] + incl - mine;
}
template <int MODE>
__global__ __launch_bounds__(ZR_CI_THREADS) void k_cull_instances(ZrPass P, const ZrObject* __restrict__ objs, uint32_t* __restrict__ work,
                                                                  ZrDevStats* __restrict__ stats, int slot)
{
    __shared__ uint32_t wsum[ZR_CI_THREADS / 64u], base_s;
    uint32_t nm[ZR_CI_PER], wbase[ZR_CI_PER], mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < ZR_CI_PER; ++j) {
        const uint32_t g = (blockIdx.x * ZR_CI_PER + j) * ZR_CI_THREADS + threadIdx.x;
        nm[j] = 0; wbase[j] = 0;
        zf3 co; float radius;
        if (g < P.n_inst_total && !instance_test<MODE>(P, objs, g, nm[j], wbase[j], co, radius)) nm[j] = 0;
        if (g >= P.n_inst_total) nm[j] = 0;
        mine += nm[j];
    }
    uint32_t off = block_reserve(mine, &stats->n_vis_work[slot], wsum, &base_s);
#pragma unroll
    for (uint32_t j = 0; j < ZR_CI_PER; ++j) { for (uint32_t m = 0; m < nm[j]; ++m) work[off + m] = wbase[j] + m; off += nm[j]; }
}

// Level 2, in two stages inside one wavefront that owns ZR_CULL_GROUP consecutive work items (every rejection is exact or conservative:
// sphere-vs-frustum and the normal-cone test use inflated bounds; "all vertices outside one clip plane"
// and "snapped bounding box holds no pixel centre" are exact):
//   A  lane per meshlet-instance: decode, load the meshlet record and the instance, shadow-pass filters, bounding sphere
//      against the frustum, normal cone against the eye;
//   B  wave per survivor, ZR_CULL_BATCH of them at a time: the batch's vertex loads are issued together, then each survivor
//      gets the lane-per-vertex transform exactly as the rasteriser will redo it, its clip flags and its snapped bounding box.
// A wave therefore waits for memory a few times per group instead of three times per meshlet.
// Outputs per work item k: rects[k] (packed tile rect or ZR_RECT_CULLED) and, for the camera pass, the pixel box and the least
// vertex depth the Hi-Z test uses (zmin < 0: not testable).
#ifndef ZR_CULL_BATCH
#define ZR_CULL_BATCH 2
#endif
#ifndef ZR_CULL_GROUP
#define ZR_CULL_GROUP 8u                     // work items per wave (stage A uses that many lanes): enough waves to fill the chip
#endif

// Stage A for one work item k (the calling lane's): decode, sphere vs frustum, normal cone, owned-region reject.
struct CullItem {
    const float4* mposv;          // first vertex of the meshlet in the flattened position array
    const ZrObject* O;
    uint32_t vcount, instanced, m, inst_i, w, tcount, tri_base;
    ZrInstance I;
    zf3 sph_c; float sph_r;       // the meshlet's bounding sphere after the instance transform (object space of PVM)
};
template <int MODE, bool WORKLIST>
__device__ __forceinline__ bool cull_stage_a(const ZrPass& P, const ZrObject* __restrict__ objs, const uint32_t* __restrict__ work,
                                             uint8_t* __restrict__ vis_clear, uint32_t k, CullItem& it)
{
    bool alive = true;
    const uint32_t w = WORKLIST ? work[k] : k;
    (void)vis_clear;                                     // (visibility marks are frame stamps: nothing to clear, see ZrHiz::vis_stamp)
    const ZrObject* __restrict__ O = objs + find_object_work(objs, (int)P.n_objects, w);
    const uint32_t local = w - O->work_base, nm = O->n_meshlets;
    const uint32_t inst_i = local / nm, m = local - inst_i * nm;
    const XkMeshlet* __restrict__ mlp = O->meshlets + m;
    // the 64-byte record as aligned 16-byte words: [16] centre.xyz radius  [32] apex.xyz axis.x  [48] axis.yz cutoff
    const float4* __restrict__ mq = (const float4*)mlp;
    const float4 q0 = ld_global(mq), bs = ld_global(mq + 1), q2 = ld_global(mq + 2), q3 = ld_global(mq + 3);
    const float4 cn = make_float4(q2.w, q3.x, q3.y, q3.z);      // axis.xyz, cutoff
    it.O = O; it.m = m; it.inst_i = inst_i; it.w = w;
    it.mposv = O->mpos + zr_f2u(q0.x); it.vcount = zr_f2u(q0.y); it.tcount = zr_f2u(q0.w); it.tri_base = zr_f2u(q3.w);      // VertexOffset, VertexCount, TriangleCount, BindlessContext
    it.I = ld_record(O->inst + inst_i);
    const ZrInstance& I = it.I;
    const uint32_t instanced = O->instanced != 0 ? 1u : 0u;
    it.instanced = instanced;
    it.sph_c = vs_position(zr3(bs.x, bs.y, bs.z), I, instanced != 0); it.sph_r = bs.w * (instanced ? __builtin_fabsf(I.s) : 1.0f);
    // the skydome is not a shadow caster (ZE:4709-4720); with N GPUs each draws every N-th instance (see k_cull_instances)
    if (MODE == ZR_MODE_SHADOW && ((O->flags & ZR_OBJ_SKY) || inst_i % P.inst_world != P.inst_rank)) alive = false;
    if (alive && (P.frustum_ok | P.cone_ok | P.rect_cull)) {
        const zf3 co = vs_position(zr3(bs.x, bs.y, bs.z), I, instanced != 0);
        const zf4 cw4 = zr_mat4_point(P.M, co);
        const zf3 cw = zr3(cw4.x, cw4.y, cw4.z);
        float rw = bs.w * (instanced ? __builtin_fabsf(I.s) : 1.0f) * P.m_scale;
        rw = __builtin_fmaf(rw, 1.001f, 1e-5f * (__builtin_fabsf(cw.x) + __builtin_fabsf(cw.y) + __builtin_fabsf(cw.z) + 1.0f));
        if (P.frustum_ok) {
            for (int q = 0; q < 6; ++q) {
                const float d = __builtin_fmaf(P.planes[q][0], cw.x, __builtin_fmaf(P.planes[q][1], cw.y,
                                __builtin_fmaf(P.planes[q][2], cw.z, P.planes[q][3])));
                if (d < -rw) alive = false;
            }
        }
        if (MODE == ZR_MODE_GBUFFER && P.cone_ok && cn.w < 1.0f && (!instanced || I.s > 0.0f)) {
            // meshoptimizer's bounding-sphere cone test, widened by ~1 degree (0.02 L): every triangle of the
            // cluster is back-facing for this eye  <=  dot(c - eye, axis) >= cutoff*|c - eye| + radius
            zf3 ax = zr3(cn.x, cn.y, cn.z);
            if (instanced) ax = zr_rowvec_mat3(ax, I.R);
            const zf3 aw = zr3(__builtin_fmaf(P.M[8], ax.z, __builtin_fmaf(P.M[4], ax.y, P.M[0] * ax.x)),
                               __builtin_fmaf(P.M[9], ax.z, __builtin_fmaf(P.M[5], ax.y, P.M[1] * ax.x)),
                               __builtin_fmaf(P.M[10], ax.z, __builtin_fmaf(P.M[6], ax.y, P.M[2] * ax.x)));
            const zf3 d = cw - zr3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
            const float L = zr_length(d);
            if (zr_dot(d, aw) >= __builtin_fmaf(cn.w + 0.02f, L, rw)) alive = false;
        }
        // multi-GPU: nothing of this meshlet can land on a tile this rank owns -> no vertex of it is transformed here
        if (alive && P.rect_cull &&
            !sphere_reaches_owned_tile(P, co, bs.w * (instanced ? __builtin_fabsf(I.s) : 1.0f))) alive = false;
    }
    return alive;
}

#ifdef ZR_DIAG      // the exact 64-lane cull: A/B builds only (ZR_SHADOW_BOX_CULL=0, ZR_FLAG_MESHLET_BINS)
template <int MODE, bool WORKLIST>
__global__ __launch_bounds__(256) void k_cull(ZrPass P, const ZrObject* __restrict__ objs, const uint32_t* __restrict__ work,
                                              uint32_t* __restrict__ rects, uint2* __restrict__ pxrect, float* __restrict__ zmin,
                                              uint8_t* __restrict__ vis_clear, const ZrDevStats* __restrict__ stats, int slot)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = WORKLIST ? stats->n_vis_work[slot] : P.n_work;
    const uint32_t wave0 = wave_uniform(blockIdx.x * 4u + (threadIdx.x >> 6)), n_waves = gridDim.x * 4u;
    for (uint32_t base = wave0 * ZR_CULL_GROUP; base < n; base += n_waves * ZR_CULL_GROUP) {
        const uint32_t k = base + lane;
        // ---------------------------------------------------------------- stage A: lane per meshlet-instance
        const bool mine = lane < ZR_CULL_GROUP && k < n;
        CullItem it;
        it.mposv = nullptr; it.O = nullptr; it.vcount = 0; it.instanced = 0; it.m = 0; it.inst_i = 0; it.w = 0; it.tcount = 0; it.tri_base = 0;
        for (int i = 0; i < 9; ++i) it.I.R[i] = 0.0f;
        it.I.t[0] = it.I.t[1] = it.I.t[2] = 0.0f; it.I.s = 1.0f;
        const bool alive = mine && cull_stage_a<MODE, WORKLIST>(P, objs, work, vis_clear, k, it);
        const float4* mposv = it.mposv;
        const uint32_t vcount = it.vcount, instanced = it.instanced;
        const ZrInstance& I = it.I;
        uint32_t out_rect = ZR_RECT_CULLED; uint2 out_px = make_uint2(0u, 0u); float out_z = -1.0f;
        const uint32_t mp_lo = (uint32_t)(unsigned long long)mposv, mp_hi = (uint32_t)((unsigned long long)mposv >> 32);

        // ---------------------------------------------------------------- stage B: wave per survivor, batched
        unsigned long long live = __ballot(alive);
        while (live) {
            uint32_t src[ZR_CULL_BATCH]; float4 pp[ZR_CULL_BATCH];
#pragma unroll
            for (int c = 0; c < ZR_CULL_BATCH; ++c) {
                src[c] = 64u;
                if (live) { src[c] = (uint32_t)__builtin_ctzll(live); live &= live - 1ull; }
                pp[c] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
                if (src[c] < 64u) {
                    const float4* __restrict__ mp = (const float4*)(((unsigned long long)lane_bcast(mp_hi, src[c]) << 32) | lane_bcast(mp_lo, src[c]));
                    if (lane < lane_bcast(vcount, src[c])) pp[c] = mp[lane];
                }
            }
#pragma unroll
            for (int c = 0; c < ZR_CULL_BATCH; ++c) {
                if (src[c] >= 64u) break;
                ZrInstance J;
                for (int i = 0; i < 9; ++i) J.R[i] = lane_bcast(I.R[i], src[c]);
                J.t[0] = lane_bcast(I.t[0], src[c]); J.t[1] = lane_bcast(I.t[1], src[c]); J.t[2] = lane_bcast(I.t[2], src[c]);
                J.s = lane_bcast(I.s, src[c]);
                const bool inst = lane_bcast(instanced, src[c]) != 0u;
                const uint32_t vc = lane_bcast(vcount, src[c]);
                // The clip tests of vertex_flags() as wave-wide votes: a comparison IS a 64-lane mask on this machine, so "every vertex
                // outside plane k" / "some vertex needs the clipper" / "some vertex is not finite" cost one v_cmp each and no cross-lane
                // reduction.  For a plain vertex the first / last pixel centre its snapped position can bound is formed per lane
                // ((X - 128 + 255) >> 8 and (X - 128) >> 8 are monotonic, so min / max commute with them); the four box sides travel as
                // two packed int16 pairs: 2 wave reductions (+ 1 for the depth).
                const bool valid = lane < vc;
                const zf4 cl = zr_mat4_point(P.PVM, vs_position(zr3(pp[c].x, pp[c].y, pp[c].z), J, inst));
                const float FM = 3.402823466e38f;
                const bool fin = __builtin_fabsf(cl.x) <= FM && __builtin_fabsf(cl.y) <= FM && __builtin_fabsf(cl.z) <= FM && __builtin_fabsf(cl.w) <= FM;
                const float gb = ZR_GUARD * cl.w;
                const bool clip = cl.z < 0.0f || !(cl.w > 0.0f) || __builtin_fabsf(cl.x) > gb || __builtin_fabsf(cl.y) > gb;
                const unsigned long long vm = __ballot(valid);
                const bool any_nonfinite = __ballot(valid && !fin) != 0ull;
                const bool any_clip = __ballot(valid && clip) != 0ull;
                const bool all_outside = __ballot(valid && cl.x < -cl.w) == vm || __ballot(valid && cl.x > cl.w) == vm ||
                                         __ballot(valid && cl.y < -cl.w) == vm || __ballot(valid && cl.y > cl.w) == vm ||
                                         __ballot(valid && cl.z < 0.0f) == vm || __ballot(valid && cl.z > cl.w) == vm;
                int lo2 = 0x7FFF7FFF, hi2 = (int)0x80008000;
                int zb = 0x7FFFFFFF;           // least NDC depth over the vertices, as ordered int bits (depths here are >= 0)
                if (valid && fin && !clip) {
                    const SV sv = project(cl, P.hw, P.hh);
                    lo2 = (clamp16((sv.X - 128 + 255) >> 8) & 0xFFFF) | (clamp16((sv.Y - 128 + 255) >> 8) << 16);
                    hi2 = (clamp16((sv.X - 128) >> 8) & 0xFFFF) | (clamp16((sv.Y - 128) >> 8) << 16);
                    zb = (int)zr_f2u(sv.z + 0.0f);
                }
                uint32_t r = ZR_RECT_CULLED; uint2 pr = make_uint2(0u, 0u); float zm = -1.0f;
                if (any_nonfinite || !all_outside) {
                    int px0 = 0, py0 = 0, px1 = (int)P.W - 1, py1 = (int)P.H - 1;
                    bool any = true;
                    if (!any_nonfinite && !any_clip) {
                        const int lo = wave_pkmin16(lo2), hi = wave_pkmax16(hi2);
                        px0 = max(px0, (int)(short)(lo & 0xFFFF)); py0 = max(py0, lo >> 16);
                        px1 = min(px1, (int)(short)(hi & 0xFFFF)); py1 = min(py1, hi >> 16);
                        any = px0 <= px1 && py0 <= py1;
                        if (any && MODE == ZR_MODE_GBUFFER) {      // unclipped meshlet (so every z >= 0): usable for the Hi-Z test
                            zm = zr_u2f((uint32_t)wave_min(zb));
                            pr = make_uint2((uint32_t)px0 | (uint32_t)py0 << 16, (uint32_t)px1 | (uint32_t)py1 << 16);
                        }
                    }
                    if (any) r = pack_tile_rect<MODE>(px0, py0, px1, py1);
                }
                if (lane == src[c]) { out_rect = r; out_px = pr; out_z = zm; }
            }
        }
        if (mine) {
            rects[k] = out_rect;
            if (MODE == ZR_MODE_GBUFFER && pxrect) { pxrect[k] = out_px; zmin[k] = out_z; }
        }
    }
}

#endif   // ZR_DIAG

// The triangle-binned camera pass needs no tile rectangle from the cull - k_geom tests every triangle exactly - only "is it gone" and,
// for the Hi-Z test of round 2, a pixel box and a least depth that BOUND the meshlet's.  Those come from the eight corners of the
// meshlet's object-space box instead of its 64 vertices, a lane per meshlet-instance instead of a wave: about a twentieth of
// k_cull<GBUFFER>'s instructions.  The bounds are conservative by construction: a vertex lies in the box, the transforms are affine up
// to rounding, and the rounding of both the corners' and the vertices' arithmetic is covered by an explicit margin (8 ulps of the
// magnitudes involved, carried through the divide; at least one pixel) - culling more is never possible, only a little less.
template <int MODE, bool WORKLIST>
__global__ __launch_bounds__(256) void k_cull_box(ZrPass P, const ZrObject* __restrict__ objs, const uint32_t* __restrict__ work,
                                                  uint32_t* __restrict__ rects, uint2* __restrict__ pxrect, float* __restrict__ zmin,
                                                  uint8_t* __restrict__ vis_clear, ZrDevStats* __restrict__ stats, int slot,
                                                  ZrBinEntry* __restrict__ sel, const uint8_t* __restrict__ vis_prev, uint32_t vis_stamp)
{
    // sel != nullptr (camera pass): the survivors that round 1 draws - all of them, or with vis_prev those that owned a pixel last
    // frame - are compacted into sel[] right here (what k_select does for round 2), one global atomic per 256 work items
    __shared__ uint32_t wcount[4], wbase[4];
    const uint32_t n = WORKLIST ? stats->n_vis_work[slot] : P.n_work;
    const float rs_x = __builtin_fabsf(P.PVM[0]) + __builtin_fabsf(P.PVM[4]) + __builtin_fabsf(P.PVM[8]);
    const float rs_y = __builtin_fabsf(P.PVM[1]) + __builtin_fabsf(P.PVM[5]) + __builtin_fabsf(P.PVM[9]);
    const float rs_z = __builtin_fabsf(P.PVM[2]) + __builtin_fabsf(P.PVM[6]) + __builtin_fabsf(P.PVM[10]);
    const float rs_w = __builtin_fabsf(P.PVM[3]) + __builtin_fabsf(P.PVM[7]) + __builtin_fabsf(P.PVM[11]);
    for (uint32_t k0 = blockIdx.x * 256u; k0 < n; k0 += gridDim.x * 256u) {
        const uint32_t k = k0 + threadIdx.x;
        CullItem it;
        it.O = nullptr; it.w = 0;
        uint32_t r = ZR_RECT_CULLED; uint2 pr = make_uint2(0u, 0u); float zm = -1.0f;
        const uint32_t r_all = (P.tiles_x - 1u) << 16 | (P.tiles_y - 1u) << 24;          // every tile: extents unknown
        if (k < n && cull_stage_a<MODE, WORKLIST>(P, objs, work, vis_clear, k, it)) {
            const float4 lo = ld_global(it.O->mbox + 2u * it.m), hi = ld_global(it.O->mbox + 2u * it.m + 1u);
            const float FM = 3.402823466e38f, U = 9.5367431640625e-7f;           // 8 ulps
            bool fin = true, clip = false;
            float mag = 0.0f, mx = 0.0f, my = 0.0f, mz = 0.0f, mw = 0.0f, wmin = FM;
            float nxl = FM, nxh = -FM, nyl = FM, nyh = -FM, zl = FM;
            uint32_t out_all = 63u;        // bit q: every corner beyond plane q (-x, +x, -y, +y, near, far)
            zf4 cl[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const zf3 wp = vs_position(zr3((q & 1) ? hi.x : lo.x, (q & 2) ? hi.y : lo.y, (q & 4) ? hi.z : lo.z), it.I, it.instanced != 0);
                cl[q] = zr_mat4_point(P.PVM, wp);
                mx = __builtin_fmaxf(mx, __builtin_fabsf(cl[q].x)); my = __builtin_fmaxf(my, __builtin_fabsf(cl[q].y));
                mz = __builtin_fmaxf(mz, __builtin_fabsf(cl[q].z)); mw = __builtin_fmaxf(mw, __builtin_fabsf(cl[q].w));
                fin = fin && __builtin_fabsf(cl[q].x) <= FM && __builtin_fabsf(cl[q].y) <= FM && __builtin_fabsf(cl[q].z) <= FM && __builtin_fabsf(cl[q].w) <= FM;
            }
            // what the corners' and the vertices' clip coordinates can differ from exact arithmetic by: 8 ulps of the largest terms
            // of the two affine maps (instance: |s p| + |t|; PVM: |row| . |position| + |translation| + |result|)
            const float pm = __builtin_fmaxf(__builtin_fabsf(lo.x), __builtin_fabsf(hi.x)) + __builtin_fmaxf(__builtin_fabsf(lo.y), __builtin_fabsf(hi.y)) +
                             __builtin_fmaxf(__builtin_fabsf(lo.z), __builtin_fabsf(hi.z));
            mag = it.instanced ? __builtin_fmaf(3.0f * __builtin_fabsf(it.I.s), pm, __builtin_fabsf(it.I.t[0]) + __builtin_fabsf(it.I.t[1]) + __builtin_fabsf(it.I.t[2])) : pm;
            const float ew = U * (mag + 1.0f);
            const float ex = __builtin_fmaf(ew, rs_x, U * (mx + __builtin_fabsf(P.PVM[12]))), ey = __builtin_fmaf(ew, rs_y, U * (my + __builtin_fabsf(P.PVM[13])));
            const float ez = __builtin_fmaf(ew, rs_z, U * (mz + __builtin_fabsf(P.PVM[14]))), eW = __builtin_fmaf(ew, rs_w, U * (mw + __builtin_fabsf(P.PVM[15])));
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const zf4 c = cl[q];
                const float gb = ZR_GUARD * c.w;
                clip = clip || c.z < ez || !(c.w > eW) || __builtin_fabsf(c.x) > gb || __builtin_fabsf(c.y) > gb;
                uint32_t o = 0;
                if (c.x < -c.w - (ex + eW)) o |= 1u;
                if (c.x > c.w + (ex + eW)) o |= 2u;
                if (c.y < -c.w - (ey + eW)) o |= 4u;
                if (c.y > c.w + (ey + eW)) o |= 8u;
                if (c.z < -ez) o |= 16u;
                if (c.z > c.w + (ez + eW)) o |= 32u;
                out_all &= o;
                wmin = __builtin_fminf(wmin, c.w);
            }
            if (!fin) r = r_all;                                // not finite: drawn, never occlusion-tested (the rasteriser sorts it out)
            else if (out_all) r = ZR_RECT_CULLED;               // the whole box is beyond one frustum plane
            else if (clip) r = r_all;                           // touches the near plane / guard band: drawn, not occlusion-tested
            else {
                const float rwm = 1.0f / (wmin - eW);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float rw = 1.0f / cl[q].w;
                    const float x = __builtin_fmaf(cl[q].x * rw, P.hw, P.hw), y = __builtin_fmaf(cl[q].y * rw, P.hh, P.hh);
                    nxl = __builtin_fminf(nxl, x); nxh = __builtin_fmaxf(nxh, x); nyl = __builtin_fminf(nyl, y); nyh = __builtin_fmaxf(nyh, y);
                    zl = __builtin_fminf(zl, cl[q].z * rw);
                }
                // |d(x / w)| <= (ex + |x / w| eW) / w, |x / w| <= ZR_GUARD here; 1/64 pixel on top for the divide, the viewport fma and the
                // snapping to 1/256 pixel
                const float px_e = __builtin_fmaf(P.hw, (ex + ZR_GUARD * eW) * rwm, 0.015625f), py_e = __builtin_fmaf(P.hh, (ey + ZR_GUARD * eW) * rwm, 0.015625f);
                const float z_e = __builtin_fmaf(ez + eW, rwm, 1e-6f);
                if (px_e < 64.0f && py_e < 64.0f) {
                    // pixel centres the snapped vertices can bound: centre i is at i + 0.5
                    int px0 = (int)__builtin_ceilf(nxl - px_e - 0.5f), py0 = (int)__builtin_ceilf(nyl - py_e - 0.5f);
                    int px1 = (int)__builtin_floorf(nxh + px_e - 0.5f), py1 = (int)__builtin_floorf(nyh + py_e - 0.5f);
                    px0 = max(px0, 0); py0 = max(py0, 0); px1 = min(px1, (int)P.W - 1); py1 = min(py1, (int)P.H - 1);
                    zm = __builtin_fmaxf(zl - z_e, 0.0f);
                    if (P.sphere_ok) {      // the bounding sphere bounds the same vertices: the tighter of the two on every side
                        int sx0, sy0, sx1, sy1; float dn;
                        if (sphere_bounds(P, it.sph_c, it.sph_r, sx0, sy0, sx1, sy1, dn)) {
                            px0 = max(px0, sx0); py0 = max(py0, sy0); px1 = min(px1, sx1); py1 = min(py1, sy1);
                            // ndc depth of a point d in front of the eye: pz_a + pz_b / d, growing with d (pz_b < 0)
                            const float zs = P.pz_a + P.pz_b / dn;
                            zm = __builtin_fmaxf(zm, zs - __builtin_fmaf(1e-6f, __builtin_fabsf(P.pz_a) + __builtin_fabsf(zs), z_e));
                        }
                    }
                    if (px0 <= px1 && py0 <= py1) {
                        r = pack_tile_rect<MODE>(px0, py0, px1, py1);
                        pr = make_uint2((uint32_t)px0 | (uint32_t)py0 << 16, (uint32_t)px1 | (uint32_t)py1 << 16);
                        // Shadow map owned by light-space super-tiles (zr_set_shadow_tiles): a meshlet is this rank's work when its texel box
                        // - the box itself, not the apron-shrunk rectangle it is LISTED under - reaches a tile the rank owns.  It is then drawn
                        // whole, in every window it is listed for (the listing tile of a meshlet that straddles a border may be the
                        // neighbour's): the owned tiles end up exact, whatever lands on the others is not sent anywhere.
                        // (The camera pass goes without: what the box test would drop there falls to k_select's Hi-Z test at the same price -
                        // a rank of eight: k_cull_box + 8 us, k_select unchanged - and k_geom emits records for owned tiles only.)
                        if (MODE == ZR_MODE_SHADOW && P.tile_world > 1u) {
                            bool mine = false;
                            for (int ty = py0 / TILE; ty <= py1 / TILE; ++ty)
                                for (int tx = px0 / TILE; tx <= px1 / TILE; ++tx)
                                    mine = mine || tile_owner((uint32_t)tx, (uint32_t)ty, P.tile_world) == P.tile_rank;
                            if (!mine) r = ZR_RECT_CULLED;
                        }
                    }
                } else r = r_all;
            }
        }
        if (k < n) {
            rects[k] = r;
            if (pxrect) { pxrect[k] = pr; zmin[k] = zm; }       // camera pass: round 2's Hi-Z test; shadow pass: k_shadow_occlusion
        }
        if (MODE == ZR_MODE_GBUFFER && sel) {
            const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
            const bool take = r != ZR_RECT_CULLED && (vis_prev == nullptr || vis_prev[it.w] == (uint8_t)vis_stamp);
            const unsigned long long m = __ballot(take);
            if (lane == 0) wcount[wv] = (uint32_t)__popcll(m);
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
                const uint32_t base = tot ? atomicAdd(&stats->n_sel[1], tot) : 0u;
                wbase[0] = base; wbase[1] = base + wcount[0]; wbase[2] = wbase[1] + wcount[1]; wbase[3] = wbase[2] + wcount[2];
            }
            __syncthreads();
            if (take) {
                const ZrObject* __restrict__ O = it.O;
                ZrBinEntry be;
                be.mpos = it.mposv; be.mtri = O->mtri + it.tri_base; be.inst = O->inst + it.inst_i;
                be.counts = it.vcount | it.tcount << 8 | (it.instanced ? 1u << 16 : 0u);
                be.prim_base = O->prim_base + it.inst_i * O->n_tris;
                sel[wbase[wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = be;
            }
            __syncthreads();      // wcount / wbase are reused by the next stretch
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers (C++ linkage, used by zr_host.cpp)

void zr_launch_instance_prep(const XkInstanceData* in, ZrInstance* out, uint32_t n, uint32_t instanced, hipStream_t s)
{
    hipLaunchKernelGGL(k_instance_prep, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n, instanced);
}
#ifdef ZR_DIAG
void zr_launch_cull(const ZrPass& P, const ZrObject* objs, uint32_t* work, uint32_t* rects, const ZrHiz& Z, ZrDevStats* stats,
                    int slot, uint32_t n_waves, hipStream_t s)
{
    if (P.n_work == 0) return;
    const dim3 gi((P.n_inst_total + ZR_CI_THREADS * ZR_CI_PER - 1u) / (ZR_CI_THREADS * ZR_CI_PER)), bi(ZR_CI_THREADS), b(256);
    // one wave per ZR_CULL_GROUP work items; with the work list the count is only known on the device: a fixed grid strides over it
    const uint32_t all = (uint32_t)(((uint64_t)P.n_work + 4u * ZR_CULL_GROUP - 1) / (4u * ZR_CULL_GROUP));
    const uint32_t blocks = P.use_worklist ? std::min<uint32_t>(all, std::max<uint32_t>(1u, n_waves / 4u)) : all;
    if (P.mode == ZR_MODE_GBUFFER) {
        if (P.use_worklist) {
            hipLaunchKernelGGL(k_cull_instances<ZR_MODE_GBUFFER>, gi, bi, 0, s, P, objs, work, stats, slot);
            hipLaunchKernelGGL((k_cull<ZR_MODE_GBUFFER, true>), dim3(blocks), b, 0, s, P, objs, work, rects, Z.pxrect, Z.zmin, Z.vis_now, stats, slot);
        } else hipLaunchKernelGGL((k_cull<ZR_MODE_GBUFFER, false>), dim3(blocks), b, 0, s, P, objs, work, rects, Z.pxrect, Z.zmin, Z.vis_now, stats, slot);
    } else {
        if (P.use_worklist) {
            hipLaunchKernelGGL(k_cull_instances<ZR_MODE_SHADOW>, gi, bi, 0, s, P, objs, work, stats, slot);
            hipLaunchKernelGGL((k_cull<ZR_MODE_SHADOW, true>), dim3(blocks), b, 0, s, P, objs, work, rects, (uint2*)nullptr, (float*)nullptr, (uint8_t*)nullptr, stats, slot);
        } else hipLaunchKernelGGL((k_cull<ZR_MODE_SHADOW, false>), dim3(blocks), b, 0, s, P, objs, work, rects, (uint2*)nullptr, (float*)nullptr, (uint8_t*)nullptr, stats, slot);
    }
}
#endif
void zr_launch_cull_box(const ZrPass& P, const ZrObject* objs, uint32_t* work, uint32_t* rects, const ZrHiz& Z, ZrDevStats* stats,
                        int slot, hipStream_t s, ZrBinEntry* sel, const uint8_t* vis_prev, bool reuse_list)
{
    const uint32_t vis_stamp = Z.vis_stamp;
    if (P.n_work == 0) return;
    const dim3 gi((P.n_inst_total + ZR_CI_THREADS * ZR_CI_PER - 1u) / (ZR_CI_THREADS * ZR_CI_PER)), bi(ZR_CI_THREADS), b(256);
    const dim3 g(std::min<uint32_t>((P.n_work + 255u) / 256u, 8192u));
    if (P.mode == ZR_MODE_GBUFFER) {
        if (P.use_worklist) {
            if (!reuse_list) hipLaunchKernelGGL(k_cull_instances<ZR_MODE_GBUFFER>, gi, bi, 0, s, P, objs, work, stats, slot);
            hipLaunchKernelGGL((k_cull_box<ZR_MODE_GBUFFER, true>), g, b, 0, s, P, objs, work, rects, Z.pxrect, Z.zmin, Z.vis_now, stats, slot, sel, vis_prev, vis_stamp);
        } else hipLaunchKernelGGL((k_cull_box<ZR_MODE_GBUFFER, false>), g, b, 0, s, P, objs, work, rects, Z.pxrect, Z.zmin, Z.vis_now, stats, slot, sel, vis_prev, vis_stamp);
    } else {
        if (P.use_worklist) {
            if (!reuse_list) hipLaunchKernelGGL(k_cull_instances<ZR_MODE_SHADOW>, gi, bi, 0, s, P, objs, work, stats, slot);
            hipLaunchKernelGGL((k_cull_box<ZR_MODE_SHADOW, true>), g, b, 0, s, P, objs, work, rects, Z.pxrect, Z.zmin, (uint8_t*)nullptr, stats, slot, (ZrBinEntry*)nullptr, (const uint8_t*)nullptr, vis_stamp);
        } else hipLaunchKernelGGL((k_cull_box<ZR_MODE_SHADOW, false>), g, b, 0, s, P, objs, work, rects, Z.pxrect, Z.zmin, (uint8_t*)nullptr, stats, slot, (ZrBinEntry*)nullptr, (const uint8_t*)nullptr, vis_stamp);
    }
}
