// zr_kernels.hip — CDNA4 (gfx950) kernels of the deferred render path.
//
//   k_instance_prep   XkInstanceData -> ZrInstance (rotation matrix), once per zr_object_add
//   k_cull_instances  big scenes only: whole-mesh sphere vs frustum per instance -> compacted work list
//   k_cull_box<MODE>  lane per meshlet-instance: bounds tests (frustum, normal cone, owned region), then the 8 corners of the
//                     meshlet's object-space box through the vertex transform -> a tile rect / pixel box / least depth that BOUND
//                     the exact ones; the camera pass's round-1 list is compacted here too
//   k_cull<MODE>      (shadow pass with ZR_SHADOW_BOX_CULL=0, camera pass with ZR_FLAG_MESHLET_BINS) the exact version: lane-per-
//                     meshlet bounds tests, then wave-per-survivor lane-per-vertex transform -> exact snapped screen box
//   shadow pass (and the camera pass with ZR_FLAG_MESHLET_BINS): meshlet-level binning
//   k_bin_count / k_scan / k_bin_fill   per-tile lists of self-contained 32-byte meshlet records (LDS histograms)
//   k_raster_chunks<MODE, HIZ, DEFER>  persistent workgroups pull chunks (<= ZR_CHUNK entries of one 32x32 tile's list): per wave,
//                     stage a meshlet's transformed vertices in LDS, test its <=128 triangles (2 per lane), compact the
//                     survivors, rasterise them lane-per-triangle into the tile's LDS depth/visibility keys with ds_min;
//                     merge touched keys into HBM (atomic min); DEFER: clipped triangles go to a list for k_tile_slow
//   camera pass: triangle-level binning
//   k_select          round 2: the meshlet-instances the Hi-Z pyramid does not hide -> 32-byte records
//   k_geom<HIZ>       wave per meshlet-instance: vertices once, exact per-triangle tests (round 2: + the pyramid per meshlet and per triangle),
//                     one 32-byte record per (triangle, tile) + its tile id
//   k_scan_tri / k_index   per-tile offsets and work units; the records' positions in tile order (an index list)
//   k_tile / k_tile_slow   lane per record, streamed: edge set-up + walk into the tile's LDS keys; clipped triangles through raster_clipped
//   k_hiz_build       max-depth pyramid of the key buffer after round 1 (two-pass occlusion culling of the camera pass)
//   k_sky_tiles       the skydome's triangles into a key plane of their own (drawn after lighting, depth-tested, colour only)
//   k_resolve_gbuffer BaseScene.frag per pixel from the key buffer; SoA GBuffer planes, coalesced row stores; marks the
//                     meshlet-instances that own a pixel (next frame's round 1)
//   k_lighting        BaseLighting.frag per pixel (PCF 5x5, per-tile light list, ambient, cubemap IBL, gamma, debug views)
//   k_gbuffer_vis     debug view 9: the GBufferVis mosaic
//   k_untile          multi-GPU composite: all-gathered packed tiles -> row-major frame
//
// Replaces: SH/Shadowmap*.vert, SH/Base*.vert, SH/BaseScene.frag, SH/Background.vert + SH/BaseLighting.frag,
// the fixed-function rasteriser/ROP state of RHICreateGraphicsPipelines (ZE:5094-5201) and the draw loops of
// RecordCommandBuffer (ZE:3239-3540).  Raster rules: DESIGN.md §4.
#include "zr_math.h"
#include "zr_types.h"

#include <algorithm>
#include <cstddef>

#define WAVE 64
#define TILE ZR_TILE
#define TILE_PIX (TILE * TILE)
// Shadow pass: a workgroup rasterises into a WINDOW = its tile plus an apron of ZR_SHADOW_APRON texels to the right and below, and a
// meshlet is listed only for the tiles that its box WITHOUT its last APRON columns / rows touches: every texel of the box still lies in
// the window of a listed tile, and a meshlet up to APRON + 1 texels across (the usual caster under a 1024^2 map: 9 texels) is listed ONCE
// where the plain tile grid listed it 1.64 times - and transformed and tested it as often.  The pass's depth test is a min, so texels that
// two windows both draw come out the same; the windows' keys are merged into the map with atomicMin as before.
// (A/B on the whole frame: apron 0 / 4 / 8 / 12 / 16 / 32 -> 5 030 / 5 060 / 5 165 / 5 187 / 4 995 / 4 830 Mpixel/s with 179 950 / - / 126 105 /
// - / 111 000 / 110 000 list entries for 110 000 meshlets: 12 is the widest window whose keys leave room for six workgroups per CU.)
#ifndef ZR_SHADOW_APRON
#define ZR_SHADOW_APRON 12
#endif
#define SPAN(MODE) ((MODE) == ZR_MODE_SHADOW ? TILE + ZR_SHADOW_APRON : TILE)       // edge of the key window of a rasteriser workgroup
#define SPAN_PIX(MODE) (SPAN(MODE) * SPAN(MODE))
#define QCAP 128u
#define RW (ZR_TILE >= 64 ? 8 : 4)          // waves per rasteriser workgroup: one 64x64 tile's keys (32 KB) are shared by 8 waves
#define RTHREADS (RW * WAVE)
#ifndef ZR_RASTER_WAVES
#define ZR_RASTER_WAVES 4                    // waves per SIMD the tile rasteriser is compiled for (5 fits only with ~25 VGPRs spilled to scratch: +100 MB of traffic per frame for 2 % less time alone, nothing side by side)
#endif
#ifndef ZR_RASTER_WAVES_DEFER
#define ZR_RASTER_WAVES_DEFER 6              // ... and the variant without the clipper in its loop (DEFER)
#endif
// Diagnostic work-skipping switches (attribution of kernel time) exist only in -DZR_DIAG builds: the product library has none.
#ifdef ZR_DIAG
#define ZR_DIAG_SKIP(x) (x)
#else
#define ZR_DIAG_SKIP(x) 0u
#endif

// ------------------------------------------------------------------------------------------------ helpers

struct SV { int X, Y; float z, rw; };          // snapped screen vertex (1/256 px), NDC depth, 1/w

__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// Ordering point between LDS accesses of ONE wave (a lane reads what another lane of the same wave wrote).  The LDS
// executes a wave's DS instructions in issue order, so no s_waitcnt is needed: only the compiler must not reorder.
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

__device__ __forceinline__ int find_object_work(const ZrObject* __restrict__ objs, int n, uint32_t w)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (objs[mid].work_base <= w) lo = mid; else hi = mid - 1; }
    return lo;
}
__device__ __forceinline__ int find_object_prim(const ZrObject* __restrict__ objs, int n, uint32_t p)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (objs[mid].prim_base <= p) lo = mid; else hi = mid - 1; }
    return lo;
}

// Loads through pointers that were themselves read from memory (ZrObject's fields, a bin record's three addresses).  The compiler cannot
// know what such a pointer points into and emits FLAT loads for it: those count on BOTH wait counters (every LDS access then waits for
// them and they for it), keep their 64-bit addresses in vector registers and can never be scalar.  All of these point into device memory
// that no kernel of the frame writes while it is read - scene data, or records an earlier kernel laid down - so the loads are spelled in
// the global address space (an SGPR base + a 32-bit lane offset) or, for a record every lane reads, in the constant one (scalar loads).
#define ZR_AS_GLOBAL __attribute__((address_space(1)))
#define ZR_AS_CONST __attribute__((address_space(4)))
typedef float zr_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t zr_u2v __attribute__((ext_vector_type(2)));
typedef uint32_t zr_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_global(const float4* p) { const zr_f4v v = *(const ZR_AS_GLOBAL zr_f4v*)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 ld_global(const uint2* p) { const zr_u2v v = *(const ZR_AS_GLOBAL zr_u2v*)p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ uint4 ld_global(const uint4* p) { const zr_u4v v = *(const ZR_AS_GLOBAL zr_u4v*)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint32_t ld_global(const uint32_t* p) { return *(const ZR_AS_GLOBAL uint32_t*)p; }
__device__ __forceinline__ float ld_global(const float* p) { return *(const ZR_AS_GLOBAL float*)p; }
__device__ __forceinline__ uint8_t ld_global(const uint8_t* p) { return *(const ZR_AS_GLOBAL uint8_t*)p; }
// a whole record (dwords): scalar loads when the address is wave-uniform, vector loads when it is not
template <class T> __device__ __forceinline__ T ld_record(const T* p)
{
    static_assert(sizeof(T) % 4 == 0, "dwords");
    T v;
    uint32_t* d = (uint32_t*)&v;
    const ZR_AS_CONST uint32_t* q = (const ZR_AS_CONST uint32_t*)p;
#pragma unroll
    for (uint32_t i = 0; i < sizeof(T) / 4; ++i) d[i] = q[i];
    return v;
}

// Base.vert:26 / BaseInstanced.vert:70 / Shadowmap*.vert: object-space position fed to PVM
__device__ __forceinline__ zf3 vs_position(zf3 p, const ZrInstance& I, bool instanced)
{
    if (!instanced) return p;
    zf3 q = zr_rowvec_mat3(p * I.s, I.R);
    return zr3(q.x + I.t[0], q.y + I.t[1], q.z + I.t[2]);
}
// outNormal = (M * vec4(normalize(n), 1)).xyz [* mat3(rotMat)] — the w = 1 is the engine's own (Base.vert:29)
__device__ __forceinline__ zf3 vs_normal(zf3 n, const ZrInstance& I, bool instanced, const float* M)
{
    zf4 mn = zr_mat4_point(M, zr_normalize(n));
    zf3 r = zr3(mn.x, mn.y, mn.z);
    return instanced ? zr_rowvec_mat3(r, I.R) : r;
}

// bit0 non-finite, bits1-6 outside {x<-w, x>w, y<-w, y>w, z<0, z>w}, bit7 needs clipping
__device__ __forceinline__ uint32_t vertex_flags(zf4 c)
{
    const float FM = 3.402823466e38f;
    uint32_t f = 0;
    if (!(__builtin_fabsf(c.x) <= FM && __builtin_fabsf(c.y) <= FM && __builtin_fabsf(c.z) <= FM && __builtin_fabsf(c.w) <= FM)) f |= 1u;
    if (c.x < -c.w) f |= 2u;
    if (c.x > c.w) f |= 4u;
    if (c.y < -c.w) f |= 8u;
    if (c.y > c.w) f |= 16u;
    if (c.z < 0.0f) f |= 32u;
    if (c.z > c.w) f |= 64u;
    float g = ZR_GUARD * c.w;
    if (c.z < 0.0f || !(c.w > 0.0f) || __builtin_fabsf(c.x) > g || __builtin_fabsf(c.y) > g) f |= 128u;
    return f;
}
// 0 discard, 1 fast path, 2 clip path
__device__ __forceinline__ int classify(uint32_t f0, uint32_t f1, uint32_t f2)
{
    if ((f0 | f1 | f2) & 1u) return 0;
    if (f0 & f1 & f2 & 0x7Eu) return 0;
    return ((f0 | f1 | f2) & 128u) ? 2 : 1;
}

__device__ __forceinline__ SV project(zf4 c, float hw, float hh)
{
    SV s;
    s.rw = 1.0f / c.w;                     // one IEEE reciprocal, then multiplies (the perspective divide)
    const float nx = c.x * s.rw, ny = c.y * s.rw;
    const float xs = __builtin_fmaf(nx, hw, hw), ys = __builtin_fmaf(ny, hh, hh);
    s.X = (int)__builtin_floorf(__builtin_fmaf(xs, 256.0f, 0.5f));
    s.Y = (int)__builtin_floorf(__builtin_fmaf(ys, 256.0f, 0.5f));
    s.z = c.z * s.rw;
    return s;
}

// Multi-GPU ownership of a tile (zelda_render.h: zr_tile_owner)
__device__ __forceinline__ uint32_t tile_owner(uint32_t tx, uint32_t ty, uint32_t world)
{
    return ((tx >> ZR_SUPERTILE_SHIFT) + (ty >> ZR_SUPERTILE_SHIFT) * ZR_SUPERTILE_SKEW) % world;
}

__device__ __forceinline__ int imin3(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(a, max(b, c)); }

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

// ------------------------------------------------------------------------------------------------ cull + bin

// Wave-wide reductions on the DPP network (no LDS round trips): an inclusive scan over each row of 16 lanes (row_shr 1, 2, 4,
// 8), then row_bcast15 / row_bcast31 fold the rows; lane 63 holds the result, which is broadcast through an SGPR.
// `idn` is the operation's identity (what lanes without a source contribute).
#define ZR_DPP_STEP(OP, ctrl, rmask) r = OP(r, __builtin_amdgcn_update_dpp(idn, r, ctrl, rmask, 0xF, false))
#define ZR_WAVE_REDUCE(OP)                                                                 \
    int r = v;                                                                              \
    ZR_DPP_STEP(OP, 0x111, 0xF); ZR_DPP_STEP(OP, 0x112, 0xF); ZR_DPP_STEP(OP, 0x114, 0xF);   \
    ZR_DPP_STEP(OP, 0x118, 0xF); ZR_DPP_STEP(OP, 0x142, 0xA); ZR_DPP_STEP(OP, 0x143, 0xC);   \
    return __builtin_amdgcn_readlane(r, 63)
__device__ __forceinline__ int op_min(int a, int b) { return min(a, b); }
__device__ __forceinline__ int op_max(int a, int b) { return max(a, b); }
__device__ __forceinline__ int op_or(int a, int b) { return a | b; }
__device__ __forceinline__ int op_and(int a, int b) { return a & b; }
__device__ __forceinline__ int op_add(int a, int b) { return a + b; }
__device__ __forceinline__ int wave_min(int v) { const int idn = 0x7FFFFFFF; ZR_WAVE_REDUCE(op_min); }
__device__ __forceinline__ int wave_max(int v) { const int idn = (int)0x80000000; ZR_WAVE_REDUCE(op_max); }
__device__ __forceinline__ int wave_sum(int v) { const int idn = 0; ZR_WAVE_REDUCE(op_add); }
__device__ __forceinline__ uint32_t wave_or(uint32_t u) { const int idn = 0, v = (int)u; ZR_WAVE_REDUCE(op_or); }
__device__ __forceinline__ uint32_t wave_and(uint32_t u) { const int idn = -1, v = (int)u; ZR_WAVE_REDUCE(op_and); }
typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int op_pkmin(int a, int b) { short2_t x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4); x = __builtin_elementwise_min(x, y); int r; __builtin_memcpy(&r, &x, 4); return r; }
__device__ __forceinline__ int op_pkmax(int a, int b) { short2_t x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4); x = __builtin_elementwise_max(x, y); int r; __builtin_memcpy(&r, &x, 4); return r; }
// two int16 lanes per register: one reduction for (x, y) pairs
__device__ __forceinline__ int wave_pkmin16(int v) { const int idn = 0x7FFF7FFF; ZR_WAVE_REDUCE(op_pkmin); }
__device__ __forceinline__ int wave_pkmax16(int v) { const int idn = (int)0x80008000; ZR_WAVE_REDUCE(op_pkmax); }
__device__ __forceinline__ int clamp16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int op_fmin(int a, int b) { return (int)zr_f2u(__builtin_fminf(zr_u2f((uint32_t)a), zr_u2f((uint32_t)b))); }
__device__ __forceinline__ int op_fmax(int a, int b) { return (int)zr_f2u(__builtin_fmaxf(zr_u2f((uint32_t)a), zr_u2f((uint32_t)b))); }
__device__ __forceinline__ float wave_fmin(float f) { const int idn = 0x7F800000, v = (int)zr_f2u(f); return zr_u2f((uint32_t)[&]() { ZR_WAVE_REDUCE(op_fmin); }()); }
__device__ __forceinline__ float wave_fmax(float f) { const int idn = (int)0xFF800000, v = (int)zr_f2u(f); return zr_u2f((uint32_t)[&]() { ZR_WAVE_REDUCE(op_fmax); }()); }

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
__device__ __forceinline__ float lane_bcast(float v, uint32_t src) { return zr_u2f((uint32_t)__builtin_amdgcn_readlane((int)zr_f2u(v), (int)src)); }
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src); }

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

// Conservative occlusion test of one meshlet-instance against the pyramid: true when every pixel of its snapped bounding box
// already holds a depth smaller than the least depth the meshlet can produce, i.e. its fragments would all fail LESS.
__device__ __forceinline__ bool hiz_occluded(const ZrHiz& Z, uint2 pr, float zmin)
{
    if (!(zmin >= 0.0f)) return false;
    const uint32_t x0 = pr.x & 0xFFFFu, y0 = pr.x >> 16, x1 = pr.y & 0xFFFFu, y1 = pr.y >> 16;
    uint32_t l = 0;                                        // 0: the 4 x 4 pixel level, 1..4: lvl[0..3]
    // the finest level at which the box spans at most 4 texels per axis
    while (l < 4u && (((x1 >> (2u + l)) - (x0 >> (2u + l))) > 3u || ((y1 >> (2u + l)) - (y0 >> (2u + l))) > 3u)) ++l;
    const uint32_t sh = 2u + l;
    const uint32_t tx0 = x0 >> sh, ty0 = y0 >> sh, tx1 = x1 >> sh, ty1 = y1 >> sh;
    if (tx1 - tx0 > 3u || ty1 - ty0 > 3u) return false;     // wider than 4x4 texels of the coarsest level: not tested
    const float* __restrict__ L = l == 0u ? Z.fine : Z.lvl[l - 1u];
    const uint32_t hw = l == 0u ? Z.fw : Z.hw[l - 1u];
    // 16 independent loads (clamped repeats at the far edges) instead of a data-dependent loop: one memory latency, not sixteen
    float hmax = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j)
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i)
            hmax = __builtin_fmaxf(hmax, L[(size_t)min(ty0 + j, ty1) * hw + min(tx0 + i, tx1)]);
    // fragment depths are clamped to their triangle's vertex depths (shade_key), so zmin bounds them exactly
    return zmin > hmax;
}

// Round 2, per (meshlet-instance, tile) pair: the pyramid level whose texels are the raster tiles tells whether the whole
// tile is already nearer than anything the meshlet can produce.  k_bin_count and k_bin_fill must agree: both call these.
__device__ __forceinline__ float tile_test_depth(const ZrHiz& Z, uint32_t k)
{
    if (TILE != 32 || Z.phase != 2u) return -1.0f;
    return Z.zmin[k];                                   // < 0: the meshlet is not occlusion-tested
}
__device__ __forceinline__ bool tile_hides(const ZrHiz& Z, float zt, uint32_t tile)
{
    return zt >= 0.0f && zt > Z.lvl[2][tile];           // level 2 = 32 x 32 pixel blocks = tiles, same row pitch (tiles_x)
}

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
        if (part[1023] > capacity) stats->overflow_sticky = 1u;
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

// ------------------------------------------------------------------------------------------------ rasteriser

struct TileCtx {
    int px0, py0;                 // tile origin in pixels
    int W, H;                     // target extent
};

// Cheap per-triangle rejection, identical in effect to the early-outs of raster_sub: degenerate or back-facing
// (GBUFFER only), or no pixel centre of this tile inside the snapped bounding box.
// HIZ (camera pass, round 2): hz[] holds the tile's 4 x 4 pyramid texels (max depth per 8 x 8 pixel block after round 1); a
// triangle whose least vertex depth lies behind every block its clipped box touches cannot win a pixel (fragment depths
// are clamped to the vertex depths).
template <int MODE, bool HIZ = false>
__device__ __forceinline__ bool tri_prefilter(int X0, int Y0, int X1, int Y1, int X2, int Y2, const TileCtx& T,
                                              float zmin = 0.0f, const float* __restrict__ hz = nullptr)
{
    if (MODE == ZR_MODE_GBUFFER) {      // the shadow pass is two-sided: its (rare) degenerate triangles are left to raster_sub
        const long long A = (long long)(X1 - X0) * (Y2 - Y0) - (long long)(X2 - X0) * (Y1 - Y0);
        if (A >= 0) return false;
    }
    const int x0 = max((imin3(X0, X1, X2) - 128 + 255) >> 8, T.px0), x1 = min((imax3(X0, X1, X2) - 128) >> 8, min(T.px0 + SPAN(MODE) - 1, T.W - 1));
    const int y0 = max((imin3(Y0, Y1, Y2) - 128 + 255) >> 8, T.py0), y1 = min((imax3(Y0, Y1, Y2) - 128) >> 8, min(T.py0 + SPAN(MODE) - 1, T.H - 1));
    if (!(x0 <= x1 && y0 <= y1)) return false;
    if (HIZ) {
        const int bx0 = (x0 - T.px0) >> 3, bx1 = (x1 - T.px0) >> 3, by0 = (y0 - T.py0) >> 3, by1 = (y1 - T.py0) >> 3;
        float h = 0.0f;
        for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx) h = __builtin_fmaxf(h, hz[by * (TILE / 8) + bx]);
        if (zmin > h) return false;
    }
    return true;
}

template <int MODE>
__device__ __forceinline__ void shade_key(int x, int y, float fy, const SV& v0, float gx, float gy, float zlo, float zhi,
                                          float bias, uint32_t prim, const TileCtx& T, unsigned long long* __restrict__ keys64, uint32_t* __restrict__ keys32)
{
    const float fx = (float)(x * 256 + 128 - v0.X);
    float z = __builtin_fmaf(gy, fy, __builtin_fmaf(gx, fx, v0.z));
    z = __builtin_fminf(__builtin_fmaxf(z, zlo), zhi);      // fragments stay within their vertices' depths (Hi-Z relies on it)
    z = z + 0.0f;
    const int li = (y - T.py0) * SPAN(MODE) + (x - T.px0);
    if (MODE == ZR_MODE_GBUFFER) {
        if (z >= 0.0f && z < 1.0f)   // depth clip (depthClampEnable FALSE) + LESS against the 1.0 clear
            atomicMin(&keys64[li], (unsigned long long)zr_f2u(z) << 32 | prim);
    } else {
        if (z >= 0.0f && z <= 1.0f) {
            const float zb = __builtin_fminf(__builtin_fmaxf(z + bias, 0.0f), 1.0f);
            atomicMin(&keys32[li], zr_f2u(zb));
        }
    }
}

// Rasterise one snapped triangle into the tile's LDS keys.
//   GBUFFER: key = depth_bits << 32 | prim, ds_min_u64  == depth test LESS in draw order (ties: lower prim wins)
//   SHADOW : key = biased depth bits,        ds_min_u32  == depth test LESS_OR_EQUAL, depth write only
// Coverage is exact integer arithmetic (edge functions of the snapped vertices, top-left rule as a -1 bias); the
// 32-bit loop is taken when every edge value met while walking the clipped bounding box fits, and is bit-identical.
// SMALL: every lane's triangle is small (every edge component below 2^14 sub-pixel units = 64 px) and given in TILE-RELATIVE
// coordinates.  The whole setup then stays in 32 bits: the clipped box lies in the tile, so |P - v| < 2^14 + 2^13 for every corner P of
// the walk and vertex v, an edge value is a difference of two products below 1.5 * 2^28 (|E| < 2^29.6 at the box origin), and the walk
// adds at most 31 steps of 256 |e| < 2^22 per axis (< 2^28): everything stays below 2^31; the area is a difference of two products
// below 2^28.  The tile kernels instantiate ONLY this form in their hot loop - bigger triangles take the clipper's route
// (raster_clipped, a call), which holds the general form - so the loop's register budget carries no 64-bit edge state.  Both forms
// produce the same integers.
#define ZR_SMALL_EDGE (1 << 14)
// every component of every EDGE below ZR_SMALL_EDGE  <=>  the snapped box is narrower than that on both axes (the widest edge
// component along an axis IS the box's extent along it).  All three edges: a triangle whose two edges at vertex 0 are short can still
// have a long third one (found by tests/test_gpu_fuzz.py: such a triangle overflowed the tile-relative 16-bit coordinates).
__device__ __forceinline__ bool tri_is_small(int x0, int y0, int x1, int y1, int x2, int y2)
{
    return imax3(x0, x1, x2) - imin3(x0, x1, x2) < ZR_SMALL_EDGE && imax3(y0, y1, y2) - imin3(y0, y1, y2) < ZR_SMALL_EDGE;
}
// BOXED: the caller hands over the clipped box it already formed with these very expressions (k_tile sorts its records by it).
template <int MODE, bool SMALL, bool BOXED = false>
__device__ __forceinline__ void raster_sub(const SV& v0, const SV& v1, const SV& v2, uint32_t prim, const TileCtx& T,
                                           unsigned long long* __restrict__ keys64, uint32_t* __restrict__ keys32, uint32_t box = 0u)
{
    const int dX1 = v1.X - v0.X, dY1 = v1.Y - v0.Y, dX2 = v2.X - v0.X, dY2 = v2.Y - v0.Y;
    constexpr bool all_small = SMALL;
    long long A;
    if (all_small) A = (long long)(dX1 * dY2 - dX2 * dY1);
    else A = (long long)dX1 * dY2 - (long long)dX2 * dY1;
    if (A == 0) return;
    // Vulkan facing: a = -A/2 in framebuffer coordinates; COUNTER_CLOCKWISE front  <=>  A < 0 (ZE:5113-5123)
    if (MODE == ZR_MODE_GBUFFER && A > 0) return;
    int x0, y0, x1, y1;
    if (BOXED) { x0 = (int)(box & 255u); y0 = (int)((box >> 8) & 255u); x1 = (int)((box >> 16) & 255u); y1 = (int)(box >> 24); }
    else {
        x0 = (imin3(v0.X, v1.X, v2.X) - 128 + 255) >> 8; x1 = (imax3(v0.X, v1.X, v2.X) - 128) >> 8;
        y0 = (imin3(v0.Y, v1.Y, v2.Y) - 128 + 255) >> 8; y1 = (imax3(v0.Y, v1.Y, v2.Y) - 128) >> 8;
        x0 = max(x0, T.px0); y0 = max(y0, T.py0);
        x1 = min(x1, min(T.px0 + SPAN(MODE) - 1, T.W - 1)); y1 = min(y1, min(T.py0 + SPAN(MODE) - 1, T.H - 1));
    }
    if (x0 > x1 || y0 > y1) return;

    const int sgn = A > 0 ? 1 : -1;
    // oriented edges (inside positive): e0 = v1->v2, e1 = v2->v0, e2 = v0->v1; top-left rule folded in as a bias
    const int ex0 = sgn * (v2.X - v1.X), ey0 = sgn * (v2.Y - v1.Y);
    const int ex1 = sgn * (v0.X - v2.X), ey1 = sgn * (v0.Y - v2.Y);
    const int ex2 = sgn * (v1.X - v0.X), ey2 = sgn * (v1.Y - v0.Y);
    const int Px0 = x0 * 256 + 128, Py0 = y0 * 256 + 128;
    const int tl0 = ((ey0 < 0) || (ey0 == 0 && ex0 > 0)) ? 0 : 1, tl1 = ((ey1 < 0) || (ey1 == 0 && ex1 > 0)) ? 0 : 1;
    const int tl2 = ((ey2 < 0) || (ey2 == 0 && ex2 > 0)) ? 0 : 1;

    // depth plane anchored at vertex 0, gradients per sub-pixel unit
    const float invA = 1.0f / (float)A;
    const float a1 = (float)(v2.Y - v0.Y) * invA, b1 = (float)(v0.X - v2.X) * invA;
    const float a2 = (float)(v0.Y - v1.Y) * invA, b2 = (float)(v1.X - v0.X) * invA;
    const float dz1 = v1.z - v0.z, dz2 = v2.z - v0.z;
    const float gx = __builtin_fmaf(a2, dz2, a1 * dz1), gy = __builtin_fmaf(b2, dz2, b1 * dz1);
    const float zlo = __builtin_fminf(__builtin_fminf(v0.z, v1.z), v2.z), zhi = __builtin_fmaxf(__builtin_fmaxf(v0.z, v1.z), v2.z);
    float bias = 0.0f;
    if (MODE == ZR_MODE_SHADOW) {
        // vkCmdSetDepthBias(1.25, 0, 7.5) on D32 (ZE:3280-3287): o = m * slope + r * constant, r = 2^(e - 23)
        const float m = __builtin_fmaxf(__builtin_fabsf(gx), __builtin_fabsf(gy)) * 256.0f;
        const float zm = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v0.z), __builtin_fabsf(v1.z)), __builtin_fabsf(v2.z));
        const uint32_t e = zr_f2u(zm) & 0x7F800000u;
        const float r = (e > (23u << 23) && e < 0x7F800000u) ? zr_u2f(e - (23u << 23)) : 0.0f;
        bias = __builtin_fmaf(m, 7.5f, r * 1.25f);
    }
    int e0 = 0, e1 = 0, e2 = 0;
    long long E0 = 0, E1 = 0, E2 = 0;
    bool fits;
    if (all_small) {
        e0 = ex0 * (Py0 - v1.Y) - ey0 * (Px0 - v1.X) - tl0;
        e1 = ex1 * (Py0 - v2.Y) - ey1 * (Px0 - v2.X) - tl1;
        e2 = ex2 * (Py0 - v0.Y) - ey2 * (Px0 - v0.X) - tl2;
        fits = true;
    } else {
        E0 = (long long)ex0 * (Py0 - v1.Y) - (long long)ey0 * (Px0 - v1.X) - tl0;
        E1 = (long long)ex1 * (Py0 - v2.Y) - (long long)ey1 * (Px0 - v2.X) - tl1;
        E2 = (long long)ex2 * (Py0 - v0.Y) - (long long)ey2 * (Px0 - v0.X) - tl2;
        // |E| anywhere in the walk <= |E at the origin| + nx*|sx| + ny*|sy|
        const long long nx = x1 - x0 + 1, ny = y1 - y0 + 1;
        const long long lim = 0x3FFFFFFFll;
        const long long m0 = (E0 < 0 ? -E0 : E0) + 256 * (nx * (ey0 < 0 ? -(long long)ey0 : ey0) + ny * (ex0 < 0 ? -(long long)ex0 : ex0));
        const long long m1 = (E1 < 0 ? -E1 : E1) + 256 * (nx * (ey1 < 0 ? -(long long)ey1 : ey1) + ny * (ex1 < 0 ? -(long long)ex1 : ex1));
        const long long m2 = (E2 < 0 ? -E2 : E2) + 256 * (nx * (ey2 < 0 ? -(long long)ey2 : ey2) + ny * (ex2 < 0 ? -(long long)ex2 : ex2));
        fits = m0 < lim && m1 < lim && m2 < lim;
        if (fits) { e0 = (int)E0; e1 = (int)E1; e2 = (int)E2; }
    }
    if (fits) {
        const int sx0 = -ey0 * 256, sx1 = -ey1 * 256, sx2 = -ey2 * 256, sy0 = ex0 * 256, sy1 = ex1 * 256, sy2 = ex2 * 256;
        for (int y = y0; y <= y1; ++y) {
            int r0 = e0, r1 = e1, r2 = e2;
            const float fy = (float)(y * 256 + 128 - v0.Y);
            for (int x = x0; x <= x1; ++x) {
                if ((r0 | r1 | r2) >= 0) shade_key<MODE>(x, y, fy, v0, gx, gy, zlo, zhi, bias, prim, T, keys64, keys32);
                r0 += sx0; r1 += sx1; r2 += sx2;
            }
            e0 += sy0; e1 += sy1; e2 += sy2;
        }
    } else {
        const long long sx0 = -(long long)ey0 * 256, sx1 = -(long long)ey1 * 256, sx2 = -(long long)ey2 * 256;
        const long long sy0 = (long long)ex0 * 256, sy1 = (long long)ex1 * 256, sy2 = (long long)ex2 * 256;
        for (int y = y0; y <= y1; ++y) {
            long long r0 = E0, r1 = E1, r2 = E2;
            const float fy = (float)(y * 256 + 128 - v0.Y);
            for (int x = x0; x <= x1; ++x) {
                if ((r0 | r1 | r2) >= 0) shade_key<MODE>(x, y, fy, v0, gx, gy, zlo, zhi, bias, prim, T, keys64, keys32);
                r0 += sx0; r1 += sx1; r2 += sx2;
            }
            E0 += sy0; E1 += sy1; E2 += sy2;
        }
    }
}

// Sutherland-Hodgman against near (z >= 0) and the 4x guard band; intersections always run inside -> outside.
__device__ __forceinline__ float plane_dist(zf4 c, int plane)
{
    switch (plane) {
    case 0: return c.z;
    case 1: return __builtin_fmaf(ZR_GUARD, c.w, c.x);
    case 2: return __builtin_fmaf(ZR_GUARD, c.w, -c.x);
    case 3: return __builtin_fmaf(ZR_GUARD, c.w, c.y);
    default: return __builtin_fmaf(ZR_GUARD, c.w, -c.y);
    }
}
__device__ __forceinline__ zf4 lerp4(zf4 in, zf4 out, float t)
{
    zf4 r;
    r.x = __builtin_fmaf(t, out.x - in.x, in.x); r.y = __builtin_fmaf(t, out.y - in.y, in.y);
    r.z = __builtin_fmaf(t, out.z - in.z, in.z); r.w = __builtin_fmaf(t, out.w - in.w, in.w);
    return r;
}
// T and the keys are tile-relative (see k_raster_chunks): (ox, oy) = the tile's origin in sub-pixel units is taken off after projecting.
template <int MODE>
__device__ __forceinline__ void raster_clipped_body(zf4 c0, zf4 c1, zf4 c2, uint32_t prim, TileCtx T, float hw, float hh, int ox, int oy,
                                                    unsigned long long* keys64, uint32_t* keys32)
{
    zf4 a[10], b[10];
    int na = 3;
    a[0] = c0; a[1] = c1; a[2] = c2;
    for (int plane = 0; plane < 5; ++plane) {
        int nb = 0;
        for (int i = 0; i < na; ++i) {
            const zf4 p = a[i], q = a[(i + 1 == na) ? 0 : i + 1];
            const float dp = plane_dist(p, plane), dq = plane_dist(q, plane);
            const bool ip = dp >= 0.0f, iq = dq >= 0.0f;
            if (ip) b[nb++] = p;
            if (ip != iq) b[nb++] = ip ? lerp4(p, q, dp / (dp - dq)) : lerp4(q, p, dq / (dq - dp));
        }
        na = nb;
        if (na < 3) return;
        for (int i = 0; i < na; ++i) a[i] = b[i];
    }
    for (int i = 0; i < na; ++i) if (!(a[i].w > 0.0f)) return;
    SV s0 = project(a[0], hw, hh); s0.X -= ox; s0.Y -= oy;
    SV sp = project(a[1], hw, hh); sp.X -= ox; sp.Y -= oy;
    for (int i = 2; i < na; ++i) {
        SV sn = project(a[i], hw, hh); sn.X -= ox; sn.Y -= oy;
        raster_sub<MODE, false>(s0, sp, sn, prim, T, keys64, keys32);
        sp = sn;
    }
}
// (a call in the rasterisers' loops - their register budget must not carry the clipper's; inlined in k_tile_slow, whose own budget is set
// so that it finds room beside the shadow rasteriser)
template <int MODE>
__device__ __noinline__ void raster_clipped(zf4 c0, zf4 c1, zf4 c2, uint32_t prim, TileCtx T, float hw, float hh, int ox, int oy,
                                            unsigned long long* keys64, uint32_t* keys32)
{
    raster_clipped_body<MODE>(c0, c1, c2, prim, T, hw, hh, ox, oy, keys64, keys32);
}

// ------------------------------------------------------------------------------------------------ GBuffer resolve

struct Bary { float b0, b1, b2; };
struct FastSetup { int X0, Y0; float a1, b1, a2, b2, rw0, rw1, rw2; };

__device__ __forceinline__ Bary bary_screen(const FastSetup& s, int px, int py)
{
    const float fx = (float)(px * 256 + 128 - s.X0), fy = (float)(py * 256 + 128 - s.Y0);
    const float l1 = __builtin_fmaf(s.b1, fy, s.a1 * fx), l2 = __builtin_fmaf(s.b2, fy, s.a2 * fx);
    const float l0 = (1.0f - l1) - l2;
    const float q0 = l0 * s.rw0, q1 = l1 * s.rw1, q2 = l2 * s.rw2;
    const float inv = 1.0f / ((q0 + q1) + q2);
    Bary r; r.b0 = q0 * inv; r.b1 = q1 * inv; r.b2 = q2 * inv;
    return r;
}
// clipped triangles: 2D-homogeneous interpolation from the unclipped clip-space vertices
__device__ __forceinline__ Bary bary_homog(const zf4* c, float hw, float hh, int px, int py)
{
    const float u = (((float)px + 0.5f) - hw) / hw, v = (((float)py + 0.5f) - hh) / hh;
    float k[3];
    for (int i = 0; i < 3; ++i) {
        const zf4 p = c[(i + 1) % 3], q = c[(i + 2) % 3];
        const float kx = __builtin_fmaf(p.y, q.w, -(q.y * p.w));
        const float ky = __builtin_fmaf(q.x, p.w, -(p.x * q.w));
        const float kz = __builtin_fmaf(p.x, q.y, -(q.x * p.y));
        k[i] = __builtin_fmaf(kx, u, __builtin_fmaf(ky, v, kz));
    }
    const float inv = 1.0f / ((k[0] + k[1]) + k[2]);
    Bary r; r.b0 = k[0] * inv; r.b1 = k[1] * inv; r.b2 = k[2] * inv;
    return r;
}
__device__ __forceinline__ float interp1(Bary b, float a0, float a1, float a2)
{
    return __builtin_fmaf(b.b2, a2, __builtin_fmaf(b.b1, a1, b.b0 * a0));
}
__device__ __forceinline__ zf3 interp3(Bary b, zf3 a0, zf3 a1, zf3 a2)
{
    return zr3(interp1(b, a0.x, a1.x, a2.x), interp1(b, a0.y, a1.y, a2.y), interp1(b, a0.z, a1.z, a2.z));
}

// ComputeNormal(fragPosition, fragTexCoord, fragNormal, texNormal), SH/Common.glsl:113-127; ts = zr_tangent_space_normal(texNormal)
__device__ __forceinline__ zf3 compute_normal(zf3 pos_dx, zf3 pos_dy, float s1, float t1, float s2, float t2, zf3 fragN, zf3 ts)
{
    // (vec3 / scalar: one IEEE reciprocal, three multiplies - DESIGN.md section 4)
    const float rdet = 1.0f / __builtin_fmaf(s1, t2, -(s2 * t1));
    zf3 T = zr3(__builtin_fmaf(t2, pos_dx.x, -(t1 * pos_dy.x)) * rdet,
                __builtin_fmaf(t2, pos_dx.y, -(t1 * pos_dy.y)) * rdet,
                __builtin_fmaf(t2, pos_dx.z, -(t1 * pos_dy.z)) * rdet);
    const zf3 N = zr_normalize(fragN);
    T = zr_normalize(T - N * zr_dot(N, T));
    const zf3 B = zr_normalize(zr_cross(N, T));
    const zf3 w = zr3(__builtin_fmaf(N.x, ts.z, __builtin_fmaf(B.x, ts.y, T.x * ts.x)),
                      __builtin_fmaf(N.y, ts.z, __builtin_fmaf(B.y, ts.y, T.y * ts.x)),
                      __builtin_fmaf(N.z, ts.z, __builtin_fmaf(B.z, ts.y, T.z * ts.x)));
    return zr_normalize(w);
}

// ---- material sampling: texture(sampler2D, uv) with LINEAR mag/min/mip, REPEAT (RHICreateSampler, ZE:6523-6557) ----
__device__ __forceinline__ int tex_idx_clamp(float f, int hi) { f = __builtin_fminf(__builtin_fmaxf(f, 0.0f), (float)hi); return (int)f; }
// A texel as the filter sees it: sRGB channels decoded to linear through `lut` (the format conversion comes before filtering), UNORM
// channels as their 8-bit CODE.  The filter - bilinear, trilinear, the anisotropic average - is linear, so the codes are filtered and the
// result is scaled by 1 / 255 ONCE per channel (tex_unorm8_scale, where a sample is finished) instead of every texel being divided first:
// the same real number, rounded once at the end (Vulkan leaves the precision of filtering to the implementation; the oracle states the
// same).  Decoding was two thirds of the sampled resolve's instructions: 8 texels x 13 channels per tap.
__device__ __forceinline__ float tex_decode(uint32_t v, bool srgb, const float* __restrict__ lut) { return srgb ? lut[v] : (float)v; }
// x / 255 of a filtered code: fma(x, k_hi, x * k_lo), k_hi + k_lo = 1 / 255 to 48 bits (c / 255 correctly rounded for an integer c)
__device__ __forceinline__ float tex_unorm8_scale(float x) { return __builtin_fmaf(x, ZR_UNORM8_HI, x * ZR_UNORM8_LO); }
__device__ __forceinline__ zf4 tex_finish(zf4 r, bool srgb)
{
    if (!srgb) { r.x = tex_unorm8_scale(r.x); r.y = tex_unorm8_scale(r.y); r.z = tex_unorm8_scale(r.z); }
    r.w = tex_unorm8_scale(r.w);
    return r;
}
__device__ __forceinline__ zf4 tex_fetch(const uint8_t* __restrict__ lvl, uint32_t w, int x, int y, bool srgb, const float* __restrict__ lut)
{
    const uint32_t t = ld_global((const uint32_t*)(lvl + ((size_t)y * w + (size_t)x) * 4));
    zf4 r;
    r.x = tex_decode(t & 255u, srgb, lut); r.y = tex_decode((t >> 8) & 255u, srgb, lut);
    r.z = tex_decode((t >> 16) & 255u, srgb, lut); r.w = tex_decode(t >> 24, false, lut);
    return r;
}
__device__ __forceinline__ zf4 tex_bilinear(const ZrTex& T, int level, float u, float v, bool srgb, const float* __restrict__ lut)
{
    size_t off = 0;
    uint32_t w = T.w >> level, h = T.h >> level;
    if (w != 0u && h != 0u && (T.w & (T.w - 1u)) == 0u && (T.h & (T.h - 1u)) == 0u) {
        // power-of-two image, level above the 1 x N tail: sum_{l < level} (w h) >> 2 l = (w h - (w h >> 2 level)) * 4 / 3 texels, exactly
        const uint32_t sz = T.w * T.h;                    // (images are at most 16384^2 texels: 2^28)
        off = (size_t)((sz - (sz >> (2 * level))) / 3u) * 16u;
    } else
        for (int l = 0; l < level; ++l) { uint32_t lw = T.w >> l, lh = T.h >> l; if (!lw) lw = 1; if (!lh) lh = 1; off += (size_t)lw * lh * 4; }
    if (!w) w = 1; if (!h) h = 1;
    const uint8_t* __restrict__ lvl = T.data + off;
    const float ur = u - __builtin_floorf(u), vr = v - __builtin_floorf(v);
    const float x = __builtin_fmaf(ur, (float)w, -0.5f), y = __builtin_fmaf(vr, (float)h, -0.5f);
    const float fx = __builtin_floorf(x), fy = __builtin_floorf(y), a = x - fx, b = y - fy;
    int x0 = tex_idx_clamp(fx + 1.0f, (int)w) - 1, y0 = tex_idx_clamp(fy + 1.0f, (int)h) - 1;
    int x1 = x0 + 1; if (x1 >= (int)w) x1 = 0; if (x0 < 0) x0 = (int)w - 1;
    int y1 = y0 + 1; if (y1 >= (int)h) y1 = 0; if (y0 < 0) y0 = (int)h - 1;
    const zf4 t00 = tex_fetch(lvl, w, x0, y0, srgb, lut), t10 = tex_fetch(lvl, w, x1, y0, srgb, lut);
    const zf4 t01 = tex_fetch(lvl, w, x0, y1, srgb, lut), t11 = tex_fetch(lvl, w, x1, y1, srgb, lut);
    zf4 r;
    { const float top = __builtin_fmaf(a, t10.x - t00.x, t00.x), bot = __builtin_fmaf(a, t11.x - t01.x, t01.x); r.x = __builtin_fmaf(b, bot - top, top); }
    { const float top = __builtin_fmaf(a, t10.y - t00.y, t00.y), bot = __builtin_fmaf(a, t11.y - t01.y, t01.y); r.y = __builtin_fmaf(b, bot - top, top); }
    { const float top = __builtin_fmaf(a, t10.z - t00.z, t00.z), bot = __builtin_fmaf(a, t11.z - t01.z, t01.z); r.z = __builtin_fmaf(b, bot - top, top); }
    { const float top = __builtin_fmaf(a, t10.w - t00.w, t00.w), bot = __builtin_fmaf(a, t11.w - t01.w, t01.w); r.w = __builtin_fmaf(b, bot - top, top); }
    return r;
}
// trilinear between the two mip levels around lambda (already clamped to the chain)
__device__ __forceinline__ zf4 tex_trilinear(const ZrTex& T, float lambda, float u, float v, bool srgb, const float* __restrict__ lut)
{
    const float fl = __builtin_floorf(lambda);
    const int l0 = (int)fl, l1 = min(l0 + 1, (int)T.levels - 1);
    const float f = lambda - fl;
    const zf4 c0 = tex_bilinear(T, l0, u, v, srgb, lut), c1 = tex_bilinear(T, l1, u, v, srgb, lut);
    zf4 r;
    r.x = __builtin_fmaf(f, c1.x - c0.x, c0.x); r.y = __builtin_fmaf(f, c1.y - c0.y, c0.y);
    r.z = __builtin_fmaf(f, c1.z - c0.z, c0.z); r.w = __builtin_fmaf(f, c1.w - c0.w, c0.w);
    return r;
}
// texture(sampler2D, uv): LINEAR, LINEAR mips, REPEAT, anisotropy on with the device's maximum (ZE:6523-6557).  The anisotropic
// scheme is the one the Vulkan specification describes: N = min(ceil(Pmax / Pmin), 16) trilinear taps spread along the major
// screen axis at lambda = log2(Pmax / N), averaged; N = 1 is plain trilinear filtering.
#define ZR_MAX_ANISO 16
// The filter footprint depends on the image's size and mip count and on the derivatives only: a material's seven textures are
// usually of one size, so the resolve forms it once per pixel and size, not once per slot.
struct TexFootprint { uint32_t w, h, levels; int N; float lambda, du, dv; };
__device__ __forceinline__ TexFootprint tex_footprint(const ZrTex& T, float dudx, float dvdx, float dudy, float dvdy)
{
    TexFootprint F;
    F.w = T.w; F.h = T.h; F.levels = T.levels;
    const float W = (float)T.w, H = (float)T.h;
    const float ax = dudx * W, ay = dvdx * H, bx = dudy * W, by = dvdy * H;
    const float rx2 = __builtin_fmaf(ax, ax, ay * ay), ry2 = __builtin_fmaf(bx, bx, by * by);
    const bool xmajor = rx2 >= ry2;
    const float rmax2 = __builtin_fmaxf(rx2, ry2), rmin2 = __builtin_fminf(rx2, ry2);
    int N = 1;                                            // least N with N^2 * Pmin^2 >= Pmax^2, at most the limit
    while (N < ZR_MAX_ANISO && (float)(N * N) * rmin2 < rmax2) ++N;
    float lambda = 0.5f * zr_log2(rmax2);
    if (N > 1) lambda = lambda - zr_log2((float)N);
    F.lambda = __builtin_fminf(__builtin_fmaxf(lambda, 0.0f), (float)(T.levels - 1u));
    F.N = N;
    F.du = xmajor ? dudx : dudy; F.dv = xmajor ? dvdx : dvdy;
    return F;
}
__device__ __forceinline__ zf4 tex_sample_footprint(const ZrTex& T, const TexFootprint& F, bool srgb, const float* __restrict__ lut, float u, float v)
{
    const int N = F.N;
    if (N == 1) return tex_finish(tex_trilinear(T, F.lambda, u, v, srgb, lut), srgb);
    zf4 acc; acc.x = acc.y = acc.z = acc.w = 0.0f;
    for (int i = 1; i <= N; ++i) {
        const float off = (float)i / (float)(N + 1) - 0.5f;
        const zf4 s = tex_trilinear(T, F.lambda, __builtin_fmaf(F.du, off, u), __builtin_fmaf(F.dv, off, v), srgb, lut);
        acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
    }
    zf4 r; r.x = acc.x / (float)N; r.y = acc.y / (float)N; r.z = acc.z / (float)N; r.w = acc.w / (float)N;
    return tex_finish(r, srgb);
}
__device__ __forceinline__ zf4 tex_sample_image(const ZrTex& T, bool srgb, const float* __restrict__ lut,
                                             float u, float v, float dudx, float dvdx, float dudy, float dvdy)
{
    const TexFootprint F = tex_footprint(T, dudx, dvdx, dudy, dvdy);
    return tex_sample_footprint(T, F, srgb, lut, u, v);
}
// A material's seven slots (BaseScene.frag:30-36) at once.  Everything about a tap but the texels themselves - footprint, the two
// mip levels, the four texel addresses and the two weights per level - depends on the image's SIZE only, and a material's images are
// usually of one size: the slots of one size are sampled as a group, tap by tap, with that part formed once per tap instead of once
// per tap and slot (it was more than half of the sampled resolve's instructions).  Per slot the arithmetic and its order are
// those of tex_sample_footprint (acc = 0 + t1 + t2 ..., / N; a single tap's 0 + t is t: no filtered texel is -0).
struct TexGeo { uint32_t o00, o10, o01, o11; float a, b; };       // byte offsets of the four texels from the image's base
__device__ __forceinline__ TexGeo tex_geo(uint32_t W0, uint32_t H0, int level, float u, float v)
{
    uint32_t off = 0;
    uint32_t w = W0 >> level, h = H0 >> level;
    if (w != 0u && h != 0u && (W0 & (W0 - 1u)) == 0u && (H0 & (H0 - 1u)) == 0u) {
        const uint32_t sz = W0 * H0;                      // as in tex_bilinear
        off = ((sz - (sz >> (2 * level))) / 3u) * 16u;
    } else
        for (int l = 0; l < level; ++l) { uint32_t lw = W0 >> l, lh = H0 >> l; if (!lw) lw = 1; if (!lh) lh = 1; off += lw * lh * 4u; }
    if (!w) w = 1; if (!h) h = 1;
    const float ur = u - __builtin_floorf(u), vr = v - __builtin_floorf(v);
    const float x = __builtin_fmaf(ur, (float)w, -0.5f), y = __builtin_fmaf(vr, (float)h, -0.5f);
    const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    int x0 = tex_idx_clamp(fx + 1.0f, (int)w) - 1, y0 = tex_idx_clamp(fy + 1.0f, (int)h) - 1;
    int x1 = x0 + 1; if (x1 >= (int)w) x1 = 0; if (x0 < 0) x0 = (int)w - 1;
    int y1 = y0 + 1; if (y1 >= (int)h) y1 = 0; if (y0 < 0) y0 = (int)h - 1;
    TexGeo g;
    g.a = x - fx; g.b = y - fy;
    const uint32_t r0 = off + (uint32_t)y0 * w * 4u, r1 = off + (uint32_t)y1 * w * 4u;
    g.o00 = r0 + (uint32_t)x0 * 4u; g.o10 = r0 + (uint32_t)x1 * 4u; g.o01 = r1 + (uint32_t)x0 * 4u; g.o11 = r1 + (uint32_t)x1 * 4u;
    return g;
}
__device__ __forceinline__ zf4 tex_decode4(uint32_t t, bool srgb, const float* __restrict__ lut)
{
    zf4 r;
    r.x = tex_decode(t & 255u, srgb, lut); r.y = tex_decode((t >> 8) & 255u, srgb, lut);
    r.z = tex_decode((t >> 16) & 255u, srgb, lut); r.w = tex_decode(t >> 24, false, lut);
    return r;
}
__device__ __forceinline__ zf4 tex_bilinear_geo(const uint8_t* __restrict__ base, const TexGeo& g, bool srgb, const float* __restrict__ lut)
{
    const uint32_t u00 = ld_global((const uint32_t*)(base + g.o00)), u10 = ld_global((const uint32_t*)(base + g.o10));
    const uint32_t u01 = ld_global((const uint32_t*)(base + g.o01)), u11 = ld_global((const uint32_t*)(base + g.o11));
    const zf4 t00 = tex_decode4(u00, srgb, lut), t10 = tex_decode4(u10, srgb, lut), t01 = tex_decode4(u01, srgb, lut), t11 = tex_decode4(u11, srgb, lut);
    const float a = g.a, b = g.b;
    zf4 r;
    { const float top = __builtin_fmaf(a, t10.x - t00.x, t00.x), bot = __builtin_fmaf(a, t11.x - t01.x, t01.x); r.x = __builtin_fmaf(b, bot - top, top); }
    { const float top = __builtin_fmaf(a, t10.y - t00.y, t00.y), bot = __builtin_fmaf(a, t11.y - t01.y, t01.y); r.y = __builtin_fmaf(b, bot - top, top); }
    { const float top = __builtin_fmaf(a, t10.z - t00.z, t00.z), bot = __builtin_fmaf(a, t11.z - t01.z, t01.z); r.z = __builtin_fmaf(b, bot - top, top); }
    { const float top = __builtin_fmaf(a, t10.w - t00.w, t00.w), bot = __builtin_fmaf(a, t11.w - t01.w, t01.w); r.w = __builtin_fmaf(b, bot - top, top); }
    return r;
}
// The packed material (ZrObject::packed, 16 B per texel = the 13 channels BaseScene.frag reads): one 16-byte load per texel instead of
// seven 4-byte ones.  The sampled resolve is bound by the vector cache's line rate (a wave's 64 lanes scatter over the image), so the
// number of loads is what counts: 8 per tap instead of 56.  Per channel the arithmetic and its order are tex_sample_footprint's.
typedef float zr_f2 __attribute__((ext_vector_type(2)));
// channel pair (2 j, 2 j + 1) of a packed texel, decoded.  Two channels ride in one register pair from here on: gfx950 issues
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (two IEEE fp32 operations per lane) at the cost of one scalar-float instruction when
// few waves share a SIMD (tools/valu_calib), and the 13 channels of a tap are independent of each other.
__device__ __forceinline__ zr_f2 pk_decode2(const uint4& t, int j, const float* __restrict__ lut)
{
    const uint32_t w = j < 2 ? t.x : j < 4 ? t.y : j < 6 ? t.z : t.w;
    const uint32_t v0 = (w >> (16 * (j & 1))) & 255u, v1 = (w >> (16 * (j & 1) + 8)) & 255u;
    // UNORM channels enter the filter as their codes (one byte -> float conversion each; scaled by 1 / 255 once, after the filter)
    zr_f2 r; r.x = (float)v0; r.y = (float)v1;
    if (j == 0) { r.x = lut[v0]; r.y = lut[v1]; }          // bytes 0..2: base colour rgb, through the sRGB table
    if (j == 1) r.x = lut[v0];
    return r;
}
__device__ __forceinline__ void tex_sample_packed(const ZrTex& T, const float* __restrict__ lut, float u, float v,
                                                  float dudx, float dvdx, float dudy, float dvdy, float (&out)[ZR_PK_CHANNELS])
{
    const TexFootprint F = tex_footprint(T, dudx, dvdx, dudy, dvdy);
    const float fl = __builtin_floorf(F.lambda);
    const int l0 = (int)fl, l1 = min(l0 + 1, (int)F.levels - 1);
    const float f = F.lambda - fl;
    const zr_f2 f2 = { f, f };
    const int N = F.N;
    constexpr int PAIRS = (ZR_PK_CHANNELS + 1) / 2;
    zr_f2 acc[PAIRS];
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) acc[j] = (zr_f2){ 0.0f, 0.0f };
    const uint8_t* __restrict__ base = T.data;
    for (int i = 1; i <= N; ++i) {
        float uu = u, vv = v;
        if (N > 1) {
            const float off = (float)i / (float)(N + 1) - 0.5f;
            uu = __builtin_fmaf(F.du, off, u); vv = __builtin_fmaf(F.dv, off, v);
        }
        const TexGeo g0 = tex_geo(F.w, F.h, l0, uu, vv), g1 = tex_geo(F.w, F.h, l1, uu, vv);     // offsets for 4-byte texels: x 4 here
        const uint4 a00 = ld_global((const uint4*)(base + (size_t)g0.o00 * 4u)), a10 = ld_global((const uint4*)(base + (size_t)g0.o10 * 4u));
        const uint4 a01 = ld_global((const uint4*)(base + (size_t)g0.o01 * 4u)), a11 = ld_global((const uint4*)(base + (size_t)g0.o11 * 4u));
        const uint4 b00 = ld_global((const uint4*)(base + (size_t)g1.o00 * 4u)), b10 = ld_global((const uint4*)(base + (size_t)g1.o10 * 4u));
        const uint4 b01 = ld_global((const uint4*)(base + (size_t)g1.o01 * 4u)), b11 = ld_global((const uint4*)(base + (size_t)g1.o11 * 4u));
        const zr_f2 a0 = { g0.a, g0.a }, b0 = { g0.b, g0.b }, a1 = { g1.a, g1.a }, b1 = { g1.b, g1.b };
#pragma unroll
        for (int j = 0; j < PAIRS; ++j) {                  // per channel: tex_bilinear's and tex_trilinear's expressions, in their order
            const zr_f2 s00 = pk_decode2(a00, j, lut), s10 = pk_decode2(a10, j, lut), s01 = pk_decode2(a01, j, lut), s11 = pk_decode2(a11, j, lut);
            const zr_f2 t00 = pk_decode2(b00, j, lut), t10 = pk_decode2(b10, j, lut), t01 = pk_decode2(b01, j, lut), t11 = pk_decode2(b11, j, lut);
            const zr_f2 top0 = __builtin_elementwise_fma(a0, s10 - s00, s00), bot0 = __builtin_elementwise_fma(a0, s11 - s01, s01);
            const zr_f2 c0 = __builtin_elementwise_fma(b0, bot0 - top0, top0);
            const zr_f2 top1 = __builtin_elementwise_fma(a1, t10 - t00, t00), bot1 = __builtin_elementwise_fma(a1, t11 - t01, t01);
            const zr_f2 c1 = __builtin_elementwise_fma(b1, bot1 - top1, top1);
            acc[j] = acc[j] + __builtin_elementwise_fma(f2, c1 - c0, c0);
        }
    }
#pragma unroll
    for (int k = 0; k < ZR_PK_CHANNELS; ++k) {
        const float a = (k & 1) ? acc[k / 2].y : acc[k / 2].x;
        const float m = N > 1 ? a / (float)N : a;
        out[k] = k < 3 ? m : tex_unorm8_scale(m);          // channels 0..2 = base colour (sRGB, already linear); the rest are filtered codes
    }
}
#define ZR_MATERIAL_SLOTS 7
__device__ __forceinline__ void tex_sample_material(const ZrObject* __restrict__ O, const float* __restrict__ lut, float u, float v,
                                                    float dudx, float dvdx, float dudy, float dvdy, zf4 (&out)[ZR_MATERIAL_SLOTS])
{
    uint32_t todo = 0;                                    // slots that hold an image and are not sampled yet
#pragma unroll
    for (int s = 0; s < ZR_MATERIAL_SLOTS; ++s) {
        if (O->tex[s].data != nullptr) todo |= 1u << s;
        else { out[s].x = O->texc[s][0]; out[s].y = O->texc[s][1]; out[s].z = O->texc[s][2]; out[s].w = O->texc[s][3]; }   // constant slot: decoded once on the host
    }
    while (todo) {                                        // one turn per image size among the slots (one, as a rule)
        const int lead = __builtin_ctz(todo);
        const ZrTex& TL = O->tex[lead];
        const TexFootprint F = tex_footprint(TL, dudx, dvdx, dudy, dvdy);
        uint32_t grp = 0;
#pragma unroll
        for (int s = 0; s < ZR_MATERIAL_SLOTS; ++s)
            if ((todo >> s & 1u) && O->tex[s].w == F.w && O->tex[s].h == F.h && O->tex[s].levels == F.levels) grp |= 1u << s;
        todo &= ~grp;
        const float fl = __builtin_floorf(F.lambda);
        const int l0 = (int)fl, l1 = min(l0 + 1, (int)F.levels - 1);
        const float f = F.lambda - fl;
        const int N = F.N;
        zf4 acc[ZR_MATERIAL_SLOTS];
#pragma unroll
        for (int s = 0; s < ZR_MATERIAL_SLOTS; ++s) acc[s].x = acc[s].y = acc[s].z = acc[s].w = 0.0f;
        for (int i = 1; i <= N; ++i) {
            float uu = u, vv = v;
            if (N > 1) {
                const float off = (float)i / (float)(N + 1) - 0.5f;
                uu = __builtin_fmaf(F.du, off, u); vv = __builtin_fmaf(F.dv, off, v);
            }
            const TexGeo g0 = tex_geo(F.w, F.h, l0, uu, vv), g1 = tex_geo(F.w, F.h, l1, uu, vv);
#pragma unroll
            for (int s = 0; s < ZR_MATERIAL_SLOTS; ++s) {
                if (!(grp >> s & 1u)) continue;
                const uint8_t* __restrict__ base = O->tex[s].data;
                const zf4 c0 = tex_bilinear_geo(base, g0, s == 0, lut), c1 = tex_bilinear_geo(base, g1, s == 0, lut);
                acc[s].x += __builtin_fmaf(f, c1.x - c0.x, c0.x); acc[s].y += __builtin_fmaf(f, c1.y - c0.y, c0.y);
                acc[s].z += __builtin_fmaf(f, c1.z - c0.z, c0.z); acc[s].w += __builtin_fmaf(f, c1.w - c0.w, c0.w);
            }
        }
#pragma unroll
        for (int s = 0; s < ZR_MATERIAL_SLOTS; ++s) {
            if (!(grp >> s & 1u)) continue;
            if (N > 1) { acc[s].x = acc[s].x / (float)N; acc[s].y = acc[s].y / (float)N; acc[s].z = acc[s].z / (float)N; acc[s].w = acc[s].w / (float)N; }
            out[s] = tex_finish(acc[s], s == 0);
        }
    }
}
// IMAGES = false: the caller knows that no slot of the scene holds an image (every material constant, the common synthetic
// case): the filter is not even instantiated, which keeps eight inlined copies of it out of the kernel's registers.
template <int IMAGES>
__device__ __forceinline__ zf4 tex_sample(const ZrTex& T, const float* __restrict__ constant, bool srgb, const float* __restrict__ lut,
                                          float u, float v, float dudx, float dvdx, float dudy, float dvdy)
{
    if (!IMAGES || T.data == nullptr) {      // constant slot: decoded once on the host
        zf4 r; r.x = constant[0]; r.y = constant[1]; r.z = constant[2]; r.w = constant[3];
        return r;
    }
    return tex_sample_image(T, srgb, lut, u, v, dudx, dvdx, dudy, dvdy);
}

// fp32 -> fp16 with the conversion unit (round to nearest even, denormals kept, overflow to inf: what zr_f32_to_f16 spells out
// in integer arithmetic for the host); NaN is canonicalised as there.
__device__ __forceinline__ uint32_t f32_to_f16_hw(float f)
{
    const _Float16 h = (_Float16)f;
    uint16_t b; __builtin_memcpy(&b, &h, 2);
    return (f != f) ? (((zr_f2u(f) >> 16) & 0x8000u) | 0x7E00u) : (uint32_t)b;
}
__device__ __forceinline__ float f16_to_f32_hw(uint32_t h)      // exact (every fp16 value is an fp32 value); quiet NaNs map as in zr_f16_to_f32
{
    const uint16_t b = (uint16_t)h;
    _Float16 v; __builtin_memcpy(&v, &b, 2);
    return (float)v;
}
// M * vec4(p, 1) when M may be the identity: for finite p every product with a zero entry is +-0, the sum is p (or a zero of either
// sign) and the final "+ M[12]" with M[12] = +0 turns -0 into +0 - which is exactly what p + 0.0f does.
__device__ __forceinline__ zf3 model_point(const ZrPass& P, zf3 p)
{
    if (P.m_identity) return zr3(p.x + 0.0f, p.y + 0.0f, p.z + 0.0f);
    const zf4 w = zr_mat4_point(P.M, p);
    return zr3(w.x, w.y, w.z);
}

// The fragment of primitive `prim` at pixel (px, py): what the rasteriser and the vertex stage hand a fragment shader (Base.vert /
// BaseInstanced.vert outputs interpolated perspective-correctly, and their fine derivatives over the pixel's 2 x 2 quad).
struct PixGeom {
    const ZrObject* O; uint32_t tri, inst_i;
    zf3 P0, N0, pos_dx, pos_dy;            // fragPosition, fragNormal, dFdx / dFdy(fragPosition)
    float u0, v0, s1, t1, s2, t2;          // fragTexCoord, dFdx(uv) = (s1, t1), dFdy(uv) = (s2, t2)
    Bary b0;                               // the pixel's own barycentrics (for whatever else is interpolated)
};
__device__ __forceinline__ PixGeom pixel_geom(const ZrPass& P, const ZrObject* __restrict__ objs, uint32_t prim, int px, int py,
                                              uint8_t* __restrict__ vis_now = nullptr, uint32_t vis_mark = 1u)
{
    PixGeom g;
    const ZrObject* __restrict__ O = objs + find_object_prim(objs, (int)P.n_objects, prim);
    const uint32_t local = prim - O->prim_base;
    const uint32_t inst_i = local / O->n_tris, tri = local - inst_i * O->n_tris;
    const bool instanced = O->instanced != 0;
    const ZrInstance I = ld_record(O->inst + inst_i);
    g.O = O; g.tri = tri; g.inst_i = inst_i;
    // visibility history for next frame's round 1: this meshlet-instance owns a pixel
    if (vis_now) vis_now[O->work_base + inst_i * O->n_meshlets + ld_global(O->tri_meshlet + tri)] = (uint8_t)vis_mark;
    zf4 clip[3]; zf3 WP[3], WN[3]; float U[3], V[3]; uint32_t fl[3];
    for (int k = 0; k < 3; ++k) {
        const float4* __restrict__ rv = (const float4*)(O->rverts + ld_global(O->indices + 3u * tri + (uint32_t)k));
        const float4 q0 = ld_global(rv), q1 = ld_global(rv + 1);       // position.xyz u | normalize(normal).xyz v
        const zf3 pos = vs_position(zr3(q0.x, q0.y, q0.z), I, instanced);
        clip[k] = zr_mat4_point(P.PVM, pos);
        WP[k] = model_point(P, pos);
        // outNormal = (M * vec4(normalize(n), 1)).xyz [* mat3(rotMat)], Base.vert:29 / BaseInstanced.vert:73
        const zf3 mn = model_point(P, zr3(q1.x, q1.y, q1.z));
        WN[k] = instanced ? zr_rowvec_mat3(mn, I.R) : mn;
        U[k] = q0.w; V[k] = q1.w;
        fl[k] = vertex_flags(clip[k]);
    }
    const int cls = classify(fl[0], fl[1], fl[2]);
    const int qx = px ^ 1, qy = py ^ 1;           // 2x2 quad partners (fine derivatives, helper lanes extrapolate)
    Bary b0, bh, bv;
    if (cls == 1) {
        const SV s0 = project(clip[0], P.hw, P.hh), s1 = project(clip[1], P.hw, P.hh), s2 = project(clip[2], P.hw, P.hh);
        const long long A = (long long)(s1.X - s0.X) * (s2.Y - s0.Y) - (long long)(s2.X - s0.X) * (s1.Y - s0.Y);
        const float invA = 1.0f / (float)A;
        FastSetup fs;
        fs.X0 = s0.X; fs.Y0 = s0.Y;
        fs.a1 = (float)(s2.Y - s0.Y) * invA; fs.b1 = (float)(s0.X - s2.X) * invA;
        fs.a2 = (float)(s0.Y - s1.Y) * invA; fs.b2 = (float)(s1.X - s0.X) * invA;
        fs.rw0 = s0.rw; fs.rw1 = s1.rw; fs.rw2 = s2.rw;
        b0 = bary_screen(fs, px, py); bh = bary_screen(fs, qx, py); bv = bary_screen(fs, px, qy);
    } else {
        b0 = bary_homog(clip, P.hw, P.hh, px, py); bh = bary_homog(clip, P.hw, P.hh, qx, py); bv = bary_homog(clip, P.hw, P.hh, px, qy);
    }
    const zf3 P0 = interp3(b0, WP[0], WP[1], WP[2]), Ph = interp3(bh, WP[0], WP[1], WP[2]), Pv = interp3(bv, WP[0], WP[1], WP[2]);
    g.P0 = P0; g.b0 = b0;
    g.N0 = interp3(b0, WN[0], WN[1], WN[2]);
    const float u0 = interp1(b0, U[0], U[1], U[2]), uh = interp1(bh, U[0], U[1], U[2]), uv_ = interp1(bv, U[0], U[1], U[2]);
    const float v0 = interp1(b0, V[0], V[1], V[2]), vh = interp1(bh, V[0], V[1], V[2]), vv = interp1(bv, V[0], V[1], V[2]);
    const float sx = (px & 1) ? 1.0f : -1.0f, sy = (py & 1) ? 1.0f : -1.0f;
    g.pos_dx = (P0 - Ph) * sx; g.pos_dy = (P0 - Pv) * sy;
    g.u0 = u0; g.v0 = v0;
    g.s1 = (u0 - uh) * sx; g.t1 = (v0 - vh) * sx; g.s2 = (u0 - uv_) * sy; g.t2 = (v0 - vv) * sy;
    return g;
}

// BaseScene.frag:26-48 for the pixel (px, py) whose winning primitive is `prim`
// returns true when the pixel holds scene geometry (not empty, not sky)
template <int IMAGES>
__device__ __forceinline__ bool resolve_pixel(const ZrPass& P, const ZrObject* __restrict__ objs, uint32_t prim, float depth,
                                              int px, int py, const GBufferPtrs& G, const float* __restrict__ lut,
                                              uint8_t* __restrict__ vis_now, uint32_t vis_mark = 1u)
{
    const size_t p = (size_t)py * P.W + (size_t)px;
    if (prim == ZR_EMPTY_PRIM) {   // clears, ZE:3427-3433
        G.depth[p] = 1.0f; G.scene_color[p] = 0xFF000000u; G.gA[p] = 0u; G.gB[p] = 0xFF000000u; G.gC[p] = 0xFF000000u;
        G.gD[p] = make_uint2(0u, 0x3C000000u);
        if (P.write_overlay) G.overlay[p] = 0u;
        return false;
    }
    const PixGeom g = pixel_geom(P, objs, prim, px, py, vis_now, vis_mark);
    const ZrObject* __restrict__ O = g.O;
    const zf3 P0 = g.P0, N0 = g.N0, pos_dx = g.pos_dx, pos_dy = g.pos_dy;
    const float u0 = g.u0, v0 = g.v0, s1 = g.s1, t1 = g.t1, s2 = g.s2, t2 = g.t2;

    if (O->flags & ZR_OBJ_SKY) {    // Skydome.frag: texture(skydomeSampler, uv).rgb, gamma - colour only, into the overlay plane: the pass
        // is drawn after the lighting quad (ZE:3681-3691) and no GBuffer attachment is written by it
        const zf4 sk = tex_sample<IMAGES>(O->tex[0], O->texc[0], true, lut, u0, v0, s1, t1, s2, t2);
        G.overlay[p] = zr_unorm(zr_pow(sk.x, 0.4545f), 255.0f) | zr_unorm(zr_pow(sk.y, 0.4545f), 255.0f) << 8 |
                       zr_unorm(zr_pow(sk.z, 0.4545f), 255.0f) << 16 | 255u << 24;
        return false;
    }
    if (P.write_overlay) G.overlay[p] = 0u;
    // texture(samplerN, fragTexCoord), BaseScene.frag:30-36; slot 0 (base colour) is R8G8B8A8_SRGB (ZE:5878).  Targets whose slots are
    // all constant were packed on the host (same zr_unorm), and so was the tangent-space normal of a constant normal map.
    uint32_t w_sc, w_gB, w_gC;
    zf3 ts;
    if (!IMAGES) {
        w_sc = O->c_scene_color; w_gB = O->c_gB; w_gC = O->c_gC;
        ts = zr3(O->ts_const[0], O->ts_const[1], O->ts_const[2]);
    } else {
        zf4 tb, tme, tro, tno, tao, tem, tms;
        if (IMAGES == 1) {                                 // every material of the scene that has images has them packed
            tb.x = O->texc[0][0]; tb.y = O->texc[0][1]; tb.z = O->texc[0][2]; tme.x = O->texc[1][0]; tro.x = O->texc[2][0];
            tno.x = O->texc[3][0]; tno.y = O->texc[3][1]; tno.z = O->texc[3][2]; tao.x = O->texc[4][0];
            tem.x = O->texc[5][0]; tem.y = O->texc[5][1]; tem.z = O->texc[5][2]; tms.x = O->texc[6][0];
            if (O->packed.data != nullptr) {
                float pk[ZR_PK_CHANNELS];
                tex_sample_packed(O->packed, lut, u0, v0, s1, t1, s2, t2, pk);
                const uint32_t cs = O->const_slots;         // a constant slot stays the constant (the oracle does not filter it)
                if (!(cs & 1u)) { tb.x = pk[ZR_PK_BC]; tb.y = pk[ZR_PK_BC + 1]; tb.z = pk[ZR_PK_BC + 2]; }
                if (!(cs & 2u)) tme.x = pk[ZR_PK_ME];
                if (!(cs & 4u)) tro.x = pk[ZR_PK_RO];
                if (!(cs & 8u)) { tno.x = pk[ZR_PK_NO]; tno.y = pk[ZR_PK_NO + 1]; tno.z = pk[ZR_PK_NO + 2]; }
                if (!(cs & 16u)) tao.x = pk[ZR_PK_AO];
                if (!(cs & 32u)) { tem.x = pk[ZR_PK_EM]; tem.y = pk[ZR_PK_EM + 1]; tem.z = pk[ZR_PK_EM + 2]; }
                if (!(cs & 64u)) tms.x = pk[ZR_PK_MS];
            }
        } else {
            zf4 ms[ZR_MATERIAL_SLOTS];
            tex_sample_material(O, lut, u0, v0, s1, t1, s2, t2, ms);
            tb = ms[0]; tme = ms[1]; tro = ms[2]; tno = ms[3]; tao = ms[4]; tem = ms[5]; tms = ms[6];
        }
        const float Rough = __builtin_fmaxf(0.01f, tro.x);
        w_sc = zr_unorm(tem.x, 255.0f) | zr_unorm(tem.y, 255.0f) << 8 | zr_unorm(tem.z, 255.0f) << 16 | zr_unorm(tms.x, 255.0f) << 24;
        w_gB = zr_unorm(tme.x, 255.0f) | zr_unorm(1.0f, 255.0f) << 8 | zr_unorm(Rough, 255.0f) << 16 | 255u << 24;
        w_gC = zr_unorm(tb.x, 255.0f) | zr_unorm(tb.y, 255.0f) << 8 | zr_unorm(tb.z, 255.0f) << 16 | zr_unorm(tao.x, 255.0f) << 24;
        ts = (O->const_slots & 8u) ? zr3(O->ts_const[0], O->ts_const[1], O->ts_const[2]) : zr_tangent_space_normal(zr3(tno.x, tno.y, tno.z));
    }
    const zf3 Nw = compute_normal(pos_dx, pos_dy, s1, t1, s2, t2, N0, ts);
    const zf3 Nn = zr_normalize(Nw);
    const zf3 NP = zr3((Nn.x + 1.0f) / 2.0f, (Nn.y + 1.0f) / 2.0f, (Nn.z + 1.0f) / 2.0f);
    G.depth[p] = depth;
    G.scene_color[p] = w_sc;
    G.gA[p] = zr_unorm(NP.z, 1023.0f) | zr_unorm(NP.y, 1023.0f) << 10 | zr_unorm(NP.x, 1023.0f) << 20 | 3u << 30;
    G.gB[p] = w_gB;
    G.gC[p] = w_gC;
    G.gD[p] = make_uint2(f32_to_f16_hw(P0.x) | f32_to_f16_hw(P0.y) << 16, f32_to_f16_hw(P0.z) | 0x3C000000u);
    return true;
}

// ------------------------------------------------------------------------------------------------ tile raster kernels

// First launch of a frame: zero the frame statistics (all but the sticky overflow latch) and, when the uniforms changed, copy
// XkView from the pinned host ring slot into this frame's device copy.  (The runtime's own hipMemcpyAsync / hipMemsetAsync
// paths cost two extra launches per frame, and the copy path stalls the host for milliseconds the first times it is used.)
__global__ __launch_bounds__(1024) void k_frame_begin(uint32_t* __restrict__ stats, uint32_t n_stats, const uint32_t* __restrict__ view_src,
                                                      uint32_t* __restrict__ view_dst, uint32_t n_view,
                                                      uint32_t* __restrict__ n_vis_camera, uint32_t rebuild_lists)
{
    const uint32_t i0 = blockIdx.x * 1024u + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < n_stats) stats[threadIdx.x] = 0u;
    // the passes' work lists (k_cull_instances) stand while camera / light matrices and scene do: only a list about to be rebuilt starts from 0
    // (the shadow pass's length sits in the shadow pipeline's own block and is reset on that pipeline's stream, see shadow_pass)
    if (blockIdx.x == 0 && threadIdx.x == 1u && (rebuild_lists & 2u)) *n_vis_camera = 0u;
    if (view_src) for (uint32_t i = i0; i < n_view; i += gridDim.x * 1024u) view_dst[i] = view_src[i];
}

__global__ void k_fill32(uint32_t* __restrict__ p, uint32_t v, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_fill64(unsigned long long* __restrict__ p, unsigned long long v, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
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
                            } else { stats->overflow = 1u; stats->overflow_sticky = 1u; }
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
                const uint32_t k = keys32[i];
                if (k < shadow_bits[p]) atomicMin(&shadow_bits[p], k);
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
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)      // inclusive prefix sum over the wave's lanes (rows on the DPP network, then across)
{
    const int idn = 0;
    int r = (int)v;
    ZR_DPP_STEP(op_add, 0x111, 0xF); ZR_DPP_STEP(op_add, 0x112, 0xF); ZR_DPP_STEP(op_add, 0x114, 0xF); ZR_DPP_STEP(op_add, 0x118, 0xF);
    ZR_DPP_STEP(op_add, 0x142, 0xA); ZR_DPP_STEP(op_add, 0x143, 0xC);
    return (uint32_t)r;
}
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
                    else { stats->overflow = 1u; stats->overflow_sticky = 1u; }
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

// The slow triangles of the round (they need the clipper, or have an edge of 64 px or more): every owned tile tries every one of
// them through raster_clipped.  A SMALL persistent grid (ZR_SLOW_BLOCKS workgroups stride over the owned tiles): the usual round has no
// slow triangle, and this launch sits on the camera lane's critical path - as one workgroup per owned tile (2 040 at 1080p, 8 KB of LDS
// each) it cost 35 us beside the shadow rasteriser just to find room and return; a few dozen workgroups come and go like k_scan_tri's one.
#ifndef ZR_SLOW_BLOCKS
#define ZR_SLOW_BLOCKS 256u
#endif
template <int MODE, bool BY_TILE>
__global__ __launch_bounds__(256) void k_tile_slow(ZrPass P, const uint32_t* __restrict__ owned_tiles, uint32_t n_owned, const uint4* __restrict__ slow,
                                                   uint32_t slow_cap, ZrDevStats* __restrict__ stats, int slot,
                                                   unsigned long long* __restrict__ vis64, uint32_t* __restrict__ shadow_bits,
                                                   const uint32_t* __restrict__ wave_culled, uint32_t n_waves)
{
    __shared__ unsigned long long keys64[MODE == ZR_MODE_GBUFFER ? TILE_PIX : 1];
    __shared__ uint32_t keys32[MODE == ZR_MODE_SHADOW ? SPAN_PIX(MODE) : 1];
    // camera pass: round 1's triangles sit in the first half of the list, round 2's in the second; one launch after round 2 draws both
    // (slot = 2), or round 1's alone in a one-round frame (slot = 1)
    const uint32_t half = BY_TILE ? slow_cap : slow_cap / 2u;
    const uint32_t n_a = min(stats->n_slow[BY_TILE ? slot : 1], half), n_b = (!BY_TILE && slot == 2) ? min(stats->n_slow[2], half) : 0u;
    if (!BY_TILE && slot == 2 && wave_culled) {
        // the meshlets round 2's k_geom dropped behind the pyramid: the per-wave counts, strided over this grid (one atomic per wave of
        // THAT kernel on one address would queue up for ~10 ns apiece and hold its end)
        uint32_t nc = 0;
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_waves; i += gridDim.x * 256u) nc += wave_culled[i];
        nc = (uint32_t)wave_sum((int)nc);
        if ((threadIdx.x & 63u) == 0u && nc) { atomicAdd(&stats->hiz_culled, nc); atomicAdd(&stats->hiz_culled_geom, nc); atomicSub(&stats->survivors[2], nc); }
    }
    if (n_a + n_b == 0u) return;
    const uint32_t tid = threadIdx.x;
    for (uint32_t ti = blockIdx.x; ti < n_owned; ti += gridDim.x) {
        const uint32_t tile = owned_tiles[ti];
        for (uint32_t i = tid; i < (uint32_t)SPAN_PIX(MODE); i += 256u) {
            if (MODE == ZR_MODE_GBUFFER) keys64[i] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
            else keys32[i] = 0x3F800000u;
        }
        __syncthreads();
        const int tpx0 = (int)(tile % P.tiles_x) * TILE, tpy0 = (int)(tile / P.tiles_x) * TILE;
        TileCtx T;
        T.px0 = 0; T.py0 = 0; T.W = (int)P.W - tpx0; T.H = (int)P.H - tpy0;
        for (uint32_t jj = tid; jj < n_a + n_b; jj += 256u) {
            const uint32_t j = jj < n_a ? jj : half + (jj - n_a);
            const uint4 q3 = slow[4u * j + 3u];
            if (BY_TILE && q3.y != tile) continue;          // (the meshlet-binned rasteriser lists a triangle once per tile of its meshlet)
            if (!BY_TILE) {                                 // camera pass: the tiles the triangle's snapped box reaches (k_geom), or all of them
                const uint32_t ttx = tile % P.tiles_x, tty = tile / P.tiles_x;
                if (ttx < (q3.y & 255u) || tty < ((q3.y >> 8) & 255u) || ttx > ((q3.y >> 16) & 255u) || tty > (q3.y >> 24)) continue;
            }
            const uint4 q0 = slow[4u * j], q1 = slow[4u * j + 1u], q2 = slow[4u * j + 2u];
            zf4 c0, c1, c2;
            c0.x = zr_u2f(q0.x); c0.y = zr_u2f(q0.y); c0.z = zr_u2f(q0.z); c0.w = zr_u2f(q0.w);
            c1.x = zr_u2f(q1.x); c1.y = zr_u2f(q1.y); c1.z = zr_u2f(q1.z); c1.w = zr_u2f(q1.w);
            c2.x = zr_u2f(q2.x); c2.y = zr_u2f(q2.y); c2.z = zr_u2f(q2.z); c2.w = zr_u2f(q2.w);
            raster_clipped<MODE>(c0, c1, c2, q3.x, T, P.hw, P.hh, tpx0 * 256, tpy0 * 256, keys64, keys32);
        }
        __syncthreads();
        for (uint32_t i = tid; i < (uint32_t)SPAN_PIX(MODE); i += 256u) {
            const int px = tpx0 + (int)(i % (uint32_t)SPAN(MODE)), py = tpy0 + (int)(i / (uint32_t)SPAN(MODE));
            if (px >= (int)P.W || py >= (int)P.H) continue;
            const size_t p = (size_t)py * P.W + (size_t)px;
            if (MODE == ZR_MODE_GBUFFER) {
                const unsigned long long k = keys64[i];
                if ((uint32_t)k != ZR_EMPTY_PRIM && k < vis64[p]) atomicMin(&vis64[p], k);
            } else {
                const uint32_t k = keys32[i];
                if (k < shadow_bits[p]) atomicMin(&shadow_bits[p], k);
            }
        }
        __syncthreads();      // the keys are cleared again for the next tile
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

// BaseScene.frag for every pixel of the owned tiles, from the frame's key buffer; resets the keys for the next frame.
// IMAGES 0: no material images in the scene; 1: every material with images has the packed form; 2: per-slot sampling.
// TB = threads per workgroup.  A tile is 256 threads x 4 pixels either way; without images that is one workgroup.  The sampled variants
// run as four independent single-wave workgroups per tile: their waves differ a lot in length (tap counts 1..16 at silhouettes) and
// hold 177+ registers, so a four-wave workgroup that waits for one slot on EVERY SIMD and retires with its slowest wave left the
// SIMDs at 1.46 resident waves of the 2 that fit.
#ifndef ZR_RESOLVE_IMG_WAVES
#define ZR_RESOLVE_IMG_WAVES 3
#endif
// PPT = pixels per thread (ZR_PIXELS_PER_THREAD; the note above k_lighting says why it is 1).
template <int IMAGES, int TB, int PPT>
__global__ __launch_bounds__(TB, TB == 64 ? ZR_RESOLVE_IMG_WAVES : 1) void k_resolve_gbuffer(ZrPass P, const ZrObject* __restrict__ objs,
                                                        const uint32_t* __restrict__ owned_tiles,
                                                        unsigned long long* __restrict__ vis64, GBufferPtrs G,
                                                        const float* __restrict__ srgb_lut, const float* __restrict__ unorm_lut,
                                                        uint8_t* __restrict__ vis_now, ZrDevStats* __restrict__ stats, uint32_t vis_mark)
{
    __shared__ uint32_t covered_s;
    __shared__ float tlut[IMAGES ? 512 : 1];       // texel decode tables of the sampler (see tex_decode)
    constexpr uint32_t T = TILE_PIX / (uint32_t)PPT, PARTS = T / (uint32_t)TB;          // threads / workgroups per tile
    const uint32_t tid = threadIdx.x + (blockIdx.x % PARTS) * (uint32_t)TB;             // the thread's place among the tile's T
    if (IMAGES) for (uint32_t i = threadIdx.x; i < 256u; i += (uint32_t)TB) { tlut[i] = srgb_lut[i]; tlut[256u + i] = unorm_lut[i]; }   // (the barrier below orders it)
    const float* __restrict__ dlut = IMAGES ? tlut : srgb_lut;
    const uint32_t tile = owned_tiles[blockIdx.x / PARTS];
    const int tx0 = (int)(tile % P.tiles_x) * TILE, ty0 = (int)(tile / P.tiles_x) * TILE;
    if (threadIdx.x == 0) covered_s = 0;
    __syncthreads();
    uint32_t ncov = 0;
    // row-major within the tile -> 128 B (256 B for GBufferD / keys) contiguous row segments per wave.  The thread's four keys are fetched
    // (and reset) together: four independent loads in flight instead of one at the head of each pixel's chain of dependent loads.
    unsigned long long keys[PPT];
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)PPT; ++q) {
        const uint32_t i = tid + q * T;
        const int px = tx0 + (int)(i & (TILE - 1)), py = ty0 + (int)(i / TILE);
        keys[q] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
        if (px < (int)P.W && py < (int)P.H) {
            const size_t p = (size_t)py * P.W + (size_t)px;
            keys[q] = vis64[p];
            vis64[p] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)PPT; ++q) {
        const uint32_t i = tid + q * T;
        const int px = tx0 + (int)(i & (TILE - 1)), py = ty0 + (int)(i / TILE);
        if (px >= (int)P.W || py >= (int)P.H) continue;
        const unsigned long long k = keys[q];
        ncov += resolve_pixel<IMAGES>(P, objs, (uint32_t)k, zr_u2f((uint32_t)(k >> 32)), px, py, G, dlut, vis_now, vis_mark) ? 1u : 0u;
        if (G.prim) G.prim[(size_t)py * P.W + (size_t)px] = (uint32_t)k;      // forward variant (k_forward): the depth test's winner
        if (IMAGES != 0 && P.sky_keys != nullptr) {
            // The skydome (ZE:3681-3691: drawn last, depth test LESS against the scene's depth, colour only).  Its triangles were
            // resolved among themselves into a key plane of their own; the dome shows where that depth is less than the scene's.
            // (The scene pixel above cleared the overlay word; the GBuffer keeps what the scene pass wrote, hidden or not.)
            const unsigned long long ks = P.sky_keys[(size_t)py * P.W + (size_t)px];
            if ((uint32_t)ks != ZR_EMPTY_PRIM && zr_u2f((uint32_t)(ks >> 32)) < zr_u2f((uint32_t)(k >> 32)))
                resolve_pixel<IMAGES>(P, objs, objs[P.sky_object].prim_base + (uint32_t)ks, zr_u2f((uint32_t)(ks >> 32)), px, py, G, dlut, nullptr);
        }
    }
    if (ncov) atomicAdd(&covered_s, ncov);
    __syncthreads();
    // (one add per workgroup; spread over 32 words - 32 000 single-wave workgroups on ONE address would queue for 10 ns apiece)
    if (threadIdx.x == 0 && covered_s) atomicAdd(&stats->covered_part[blockIdx.x & 31u], covered_s);
}

// statistics only (not part of the frame): shadow-map texels with depth < 1
__global__ void k_count_shadow(const uint32_t* __restrict__ bits, size_t n, ZrDevStats* __restrict__ stats)
{
    uint32_t c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += bits[i] != 0x3F800000u;
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(&stats->covered_shadow, c);
}

// ------------------------------------------------------------------------------------------------ lighting

__device__ __forceinline__ int idx_clamp(float f, int hi) { f = __builtin_fminf(__builtin_fmaxf(f, 0.0f), (float)hi); return (int)f; }

__device__ __forceinline__ zf3 cube_fetch(const CubeDesc& C, const float* __restrict__ lut, uint32_t dim0, int level, int face, int x, int y)
{
    uint32_t d = dim0 >> level; if (d == 0) d = 1;
    const uint8_t* p = C.levels[level] + ((size_t)d * d * (size_t)face + (size_t)y * d + (size_t)x) * 4;
    const uint32_t t = *(const uint32_t*)p;
    return zr3(lut[t & 255u], lut[(t >> 8) & 255u], lut[(t >> 16) & 255u]);
}
__device__ __forceinline__ zf3 lerp3(float a, zf3 x, zf3 y)
{
    return zr3(__builtin_fmaf(a, y.x - x.x, x.x), __builtin_fmaf(a, y.y - x.y, x.y), __builtin_fmaf(a, y.z - x.z, x.z));
}
__device__ __forceinline__ zf3 cube_bilinear(const CubeDesc& C, const float* __restrict__ lut, uint32_t dim0, int level, int face, float s, float t)
{
    uint32_t d = dim0 >> level; if (d == 0) d = 1;
    const float u = __builtin_fmaf(s, (float)d, -0.5f), v = __builtin_fmaf(t, (float)d, -0.5f);
    const float fu = __builtin_floorf(u), fv = __builtin_floorf(v);
    const float a = u - fu, b = v - fv;
    const int x0 = idx_clamp(fu, (int)d - 1), x1 = idx_clamp(fu + 1.0f, (int)d - 1);
    const int y0 = idx_clamp(fv, (int)d - 1), y1 = idx_clamp(fv + 1.0f, (int)d - 1);
    const zf3 top = lerp3(a, cube_fetch(C, lut, dim0, level, face, x0, y0), cube_fetch(C, lut, dim0, level, face, x1, y0));
    const zf3 bot = lerp3(a, cube_fetch(C, lut, dim0, level, face, x0, y1), cube_fetch(C, lut, dim0, level, face, x1, y1));
    return lerp3(b, top, bot);
}
// textureLod(samplerCube, R, lod): Vulkan face selection (z wins ties over y over x), trilinear, faces clamp-to-edge
__device__ __forceinline__ zf3 cube_sample(const CubeDesc& C, const float* __restrict__ lut, uint32_t dim0, int nlevels, zf3 R, float lod)
{
    const float ax = __builtin_fabsf(R.x), ay = __builtin_fabsf(R.y), az = __builtin_fabsf(R.z);
    int face; float sc, tc, ma;
    if (az >= ax && az >= ay) { ma = az; if (R.z >= 0.0f) { face = 4; sc = R.x; tc = -R.y; } else { face = 5; sc = -R.x; tc = -R.y; } }
    else if (ay >= ax)        { ma = ay; if (R.y >= 0.0f) { face = 2; sc = R.x; tc = R.z; }  else { face = 3; sc = R.x; tc = -R.z; } }
    else                      { ma = ax; if (R.x >= 0.0f) { face = 0; sc = -R.z; tc = -R.y; } else { face = 1; sc = R.z; tc = -R.y; } }
    const float rma = 1.0f / ma;
    const float s = __builtin_fmaf(sc * rma, 0.5f, 0.5f), t = __builtin_fmaf(tc * rma, 0.5f, 0.5f);
    const float l = __builtin_fminf(__builtin_fmaxf(lod, 0.0f), (float)(nlevels - 1));
    const float fl = __builtin_floorf(l);
    const int l0 = (int)fl, l1 = min(l0 + 1, nlevels - 1);
    return lerp3(l - fl, cube_bilinear(C, lut, dim0, l0, face, s, t), cube_bilinear(C, lut, dim0, l1, face, s, t));
}

__device__ __forceinline__ float F_Schlick(float f0, float f90, float u) { return __builtin_fmaf(f90 - f0, zr_pow5(1.0f - u), f0); }   // SH/Common.glsl:134
__device__ __forceinline__ float Fr_DisneyDiffuse(float NdotV, float NdotL, float LdotH, float r)                                     // :148
{
    const float E_bias = __builtin_fmaf(0.5f, r, 0.0f * (1.0f - r));
    const float E_factor = __builtin_fmaf(1.0f / 1.51f, r, 1.0f * (1.0f - r));
    const float fd90 = __builtin_fmaf((2.0f * LdotH) * LdotH, r, E_bias);
    return (F_Schlick(1.0f, fd90, NdotL) * F_Schlick(1.0f, fd90, NdotV)) * E_factor;
}
__device__ __forceinline__ float V_SmithGGXCorrelated(float NdotV, float NdotL, float r)                                             // :161
{
    const float a2 = r * r;
    const float GGXV = NdotL * __builtin_sqrtf(__builtin_fmaf(NdotV * NdotV, 1.0f - a2, a2));
    const float GGXL = NdotV * __builtin_sqrtf(__builtin_fmaf(NdotL * NdotL, 1.0f - a2, a2));
    const float GGX = GGXV + GGXL;
    return GGX > 0.0f ? 0.5f / GGX : 0.0f;
}
__device__ __forceinline__ float D_GGX(float NdotH, float r)                                                                         // :178
{
    const float a2 = r * r;
    const float f = __builtin_fmaf(__builtin_fmaf(NdotH, a2, -NdotH), NdotH, 1.0f);
    return a2 / ((3.14159265359f * f) * f);
}

typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load from a 4-byte aligned address

// ShadowDepthProject + texture(LINEAR, clamp-to-edge) of the D32 shadow map (SH/Common.glsl:307-319; sampler ZE:2532-2537)
__device__ __forceinline__ float shadow_tap(const float* __restrict__ S, int SD, float sx, float sy, float sz, float sw, float ox, float oy)
{
    float f = 1.0f;
    if (sz > -1.0f && sz < 1.0f) {
        const float dim = (float)SD;
        const float u = __builtin_fmaf(sx + ox, dim, -0.5f), v = __builtin_fmaf(sy + oy, dim, -0.5f);
        const float fu = __builtin_floorf(u), fv = __builtin_floorf(v), a = u - fu, b = v - fv;
        const int x0 = idx_clamp(fu, SD - 1), x1 = idx_clamp(fu + 1.0f, SD - 1);
        const int y0 = idx_clamp(fv, SD - 1), y1 = idx_clamp(fv + 1.0f, SD - 1);
        const float t00 = S[(size_t)y0 * SD + x0], t10 = S[(size_t)y0 * SD + x1];
        const float t01 = S[(size_t)y1 * SD + x0], t11 = S[(size_t)y1 * SD + x1];
        const float top = __builtin_fmaf(a, t10 - t00, t00), bot = __builtin_fmaf(a, t11 - t01, t01);
        const float dist = __builtin_fmaf(b, bot - top, top);
        if (sw > 0.0f && dist < sz) f = 0.1f;
    }
    return f;
}

// What BaseLighting.frag:174-221 and Base.frag:62-112 have in common, word for word: the PCF factor, (1) direct lighting over the
// directional then the point lights, (2) the lambert indirect term, (3) the image-based reflection.  Inputs as the shader holds them at that
// point (N: normalize(Normal) of the unpacked GBufferA in the deferred shader, ComputeNormal()'s result in the forward one).
// USE_MASK: the point lights are the set bits of lmask (the tile's light list, k_lighting), walked in ascending order.
template <bool USE_MASK>
__device__ __forceinline__ void shade_surface(const ZrLightParams& L, const XkView* __restrict__ view, const float* __restrict__ shadowmap,
                                              const CubeDesc& C, const float* __restrict__ slut, const uint32_t* lmask,
                                              uint32_t nDir, uint32_t nPoint, float maxmips, float dxy, zf3 cam,
                                              zf3 BaseColor, float Metallic, float Roughness, zf3 N, float AO, zf3 Pw,
                                              zf3& Direct, zf3& Indirect, zf3& RefC, float& ShadowFactor)
{
    const zf3 Vv = zr_normalize(cam - Pw);
    const float NdotV = zr_saturate(zr_dot(N, Vv));

    const zf4 s4 = zr_mat4_point(L.SB, Pw);
    // shadowCoord / shadowCoord.w (SH/Common.glsl:296): IEEE divisions - the PCF comparison below is the shader's one discontinuity,
    // and a reciprocal-multiply moved its ties
    const float sx = s4.x / s4.w, sy = s4.y / s4.w, sz = s4.z / s4.w, sw = s4.w / s4.w;
    // ComputePCF r = 2 (SH/Common.glsl:323-342): 25 taps of ShadowDepthProject.  A tap's texel column / row and bilinear
    // weight depend only on its x / y offset, so they are formed once per axis (5 + 5) instead of once per tap (25 + 25);
    // every tap still evaluates fma(sx + ox, dim, -0.5) etc. with the same operands, i.e. the same bits.
    float sum = 0.0f;
    if (sz > -1.0f && sz < 1.0f && !(ZR_DIAG_SKIP(L.debug_skip) & 1u)) {
        const int SDi = (int)L.SD;
        const float dim = (float)SDi;
        int cx0[5], cx1[5], ry0[5], ry1[5]; float wa[5], wb[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float off = dxy * (float)(k - 2);
            const float u = __builtin_fmaf(sx + off, dim, -0.5f), v = __builtin_fmaf(sy + off, dim, -0.5f);
            const float fu = __builtin_floorf(u), fv = __builtin_floorf(v);
            wa[k] = u - fu; wb[k] = v - fv;
            cx0[k] = idx_clamp(fu, SDi - 1); cx1[k] = idx_clamp(fu + 1.0f, SDi - 1);
            ry0[k] = idx_clamp(fv, SDi - 1) * SDi; ry1[k] = idx_clamp(fv + 1.0f, SDi - 1) * SDi;
        }
        // Column pattern of the five x offsets (-3, -1.5, 0, 1.5, 3 texels) when nothing is clamped: pairs start at
        // cb, cb+1|cb+2, cb+3, cb+4|cb+5, cb+6 - all inside an 8-texel span.  Then each tap row is TWO 16-byte loads per lane
        // instead of ten 4-byte ones (the texture path processes ~4 lane addresses per clock whatever their width), and the
        // taps pick their texels from registers.  Any deviation (map edge, a rounding oddity) takes the per-texel path.
        const int cb = cx0[0];
        const bool p1 = cx0[1] == cb + 2, p3 = cx0[3] == cb + 5;
        const bool pattern = cx1[0] == cb + 1 && (cx0[1] == cb + 1 || p1) && cx1[1] == cx0[1] + 1 && cx0[2] == cb + 3 && cx1[2] == cb + 4 &&
                             (cx0[3] == cb + 4 || p3) && cx1[3] == cx0[3] + 1 && cx0[4] == cb + 6 && cx1[4] == cb + 7;
        if (pattern) {
            // taps accumulate in the reference's order (x outer, y inner): keep the 25 outcomes (one bit each), add afterwards
            uint32_t lit = 0u;
#pragma unroll
            for (int y = 0; y < 5; ++y) {
                const float4_u a0 = *(const float4_u*)(shadowmap + ry0[y] + cb), a1 = *(const float4_u*)(shadowmap + ry0[y] + cb + 4);
                const float4_u b0 = *(const float4_u*)(shadowmap + ry1[y] + cb), b1 = *(const float4_u*)(shadowmap + ry1[y] + cb + 4);
                const float t00[5] = { a0.x, p1 ? a0.z : a0.y, a0.w, p3 ? a1.y : a1.x, a1.z };
                const float t10[5] = { a0.y, p1 ? a0.w : a0.z, a1.x, p3 ? a1.z : a1.y, a1.w };
                const float t01[5] = { b0.x, p1 ? b0.z : b0.y, b0.w, p3 ? b1.y : b1.x, b1.z };
                const float t11[5] = { b0.y, p1 ? b0.w : b0.z, b1.x, p3 ? b1.z : b1.y, b1.w };
#pragma unroll
                for (int x = 0; x < 5; ++x) {
                    const float top = __builtin_fmaf(wa[x], t10[x] - t00[x], t00[x]), bot = __builtin_fmaf(wa[x], t11[x] - t01[x], t01[x]);
                    const float dist = __builtin_fmaf(wb[y], bot - top, top);
                    if (sw > 0.0f && dist < sz) lit |= 1u << (x * 5 + y);
                }
            }
#pragma unroll
            for (int x = 0; x < 5; ++x)
#pragma unroll
                for (int y = 0; y < 5; ++y) sum += ((lit >> (x * 5 + y)) & 1u) ? 0.1f : 1.0f;
        } else {
#pragma unroll
            for (int x = 0; x < 5; ++x)
#pragma unroll
                for (int y = 0; y < 5; ++y) {
                    const float t00 = shadowmap[ry0[y] + cx0[x]], t10 = shadowmap[ry0[y] + cx1[x]];
                    const float t01 = shadowmap[ry1[y] + cx0[x]], t11 = shadowmap[ry1[y] + cx1[x]];
                    const float top = __builtin_fmaf(wa[x], t10 - t00, t00), bot = __builtin_fmaf(wa[x], t11 - t01, t01);
                    const float dist = __builtin_fmaf(wb[y], bot - top, top);
                    sum += (sw > 0.0f && dist < sz) ? 0.1f : 1.0f;
                }
        }
    } else sum = 25.0f;      // every tap returns 1.0: 25 exact additions
    ShadowFactor = sum * 0.04f;       // ShadowFactor / Count (25 taps)

    Direct = zr3(0.0f, 0.0f, 0.0f);
    const zf3 Nn = zr_normalize(N);                      // Apply*Light and refract() re-normalise N
    const zf3 DiffuseColor = BaseColor * (1.0f - Metallic);
    // lights in the shader's order: directional, then point (with a tile list: only its set bits, ascending)
    const uint32_t n_lights = (ZR_DIAG_SKIP(L.debug_skip) & 2u) ? 0u : nDir + nPoint;
    uint32_t mword = 0u, mnext = 0u;       // remaining bits of the current mask word, index of the next word
    for (uint32_t li = 0; li < n_lights; ++li) {
        if (USE_MASK && li >= nDir) {
            while (mword == 0u && mnext * 32u < nPoint) mword = lmask[mnext++];
            if (mword == 0u) break;
            const uint32_t b = (uint32_t)__builtin_ctz(mword);
            mword &= mword - 1u;
            li = nDir + (mnext - 1u) * 32u + b;
            if (li >= n_lights) break;
        }
        const bool isdir = li < nDir;
        const XkLight* __restrict__ Lt = isdir ? &view->DirectionalLights[li] : &view->PointLights[li - nDir];
        const zf3 lp = zr3(Lt->Position[0], Lt->Position[1], Lt->Position[2]);
        // A light whose radiance factor is exactly 0 adds fma(0, bxdf, Direct) = Direct: skip its BxDF.  That is the
        // case beyond a point light's radius (attenuation 1 - clamp(d, 0, r) / r = 0) and for N.L <= 0.  The skip
        // needs finite colour * intensity (0 * finite = 0); the test is wave-uniform per light.
        const bool lfinite = __builtin_fabsf(Lt->Color[0]) <= 3.402823466e38f && __builtin_fabsf(Lt->Color[1]) <= 3.402823466e38f &&
                             __builtin_fabsf(Lt->Color[2]) <= 3.402823466e38f && __builtin_fabsf(Lt->Color[3]) <= 3.402823466e38f;
        float att = 1.0f;
        zf3 Lv;
        if (!isdir) {
            const float falloff = Lt->Direction[3];
            // far outside the radius (1e-6 relative margin on the squared distance covers every rounding in dist): the exact
            // test below would give att == 0, so the distance and the quotient need not be formed
            const zf3 dl = lp - Pw;
            const float d2 = zr_dot(dl, dl);
            if (lfinite && falloff > 0.0f && d2 > (falloff * falloff) * 1.000001f) continue;
            // distance(light_pos, position) and normalize(light_pos - position) share ONE inversesqrt: length = d2 * inversesqrt(d2)
            // (0 for d2 = 0; GLSL derives sqrt's precision from inversesqrt's), direction = dl * inversesqrt(d2)
            const float rd = zr_rsqrt(d2);
            const float dist = d2 > 0.0f ? d2 * rd : 0.0f;
            // remap(dist, 0, falloff, 0, 1), SH/Common.glsl:43-47.  The quotient stays an IEEE division: falloff / falloff must be
            // exactly 1 beyond the radius (the tile light lists and the skips around here rest on att == 0 there)
            att = 1.0f - zr_clamp(dist, 0.0f, falloff) / falloff;
            if (lfinite && att == 0.0f) continue;
            Lv = dl * rd;
        } else Lv = zr_normalize(zr3(Lt->Direction[0], Lt->Direction[1], Lt->Direction[2]));
        // ApplyDirectionalLight / ApplyPointLight (SH/Common.glsl:364-372, 399-416)
        const float ndotl = zr_clamp(zr_dot(Nn, Lv), 0.0f, 1.0f);
        if (lfinite && ndotl == 0.0f) continue;
        const zf3 Hh = zr_normalize(Vv + Lv);
        const float LdotH = zr_saturate(zr_dot(Lv, Hh)), NdotH = zr_saturate(zr_dot(N, Hh)), NdotL = zr_saturate(zr_dot(N, Lv));
        // DefaultLitBxDF (SH/Common.glsl:259-282): F0 = 0.04, F90 = saturate(50 * 0.04)
        const float F = F_Schlick(0.04f, zr_saturate(50.0f * 0.04f), LdotH);
        const float Vis = V_SmithGGXCorrelated(NdotV, NdotL, Roughness);
        const float Dg = D_GGX(NdotH, Roughness);
        const float Fr = (F * Dg) * Vis;
        const float Fd = Fr_DisneyDiffuse(NdotV, NdotL, LdotH, Roughness);
        const zf3 bx = zr3(__builtin_fmaf(DiffuseColor.x * (1.0f - F), Fd, Fr), __builtin_fmaf(DiffuseColor.y * (1.0f - F), Fd, Fr),
                           __builtin_fmaf(DiffuseColor.z * (1.0f - F), Fd, Fr));
        const float k = ndotl * Lt->Color[3];
        zf3 rad = zr3(k * Lt->Color[0], k * Lt->Color[1], k * Lt->Color[2]);
        if (isdir) {
            Direct = zr3(__builtin_fmaf(rad.x * bx.x, ShadowFactor, Direct.x), __builtin_fmaf(rad.y * bx.y, ShadowFactor, Direct.y),
                         __builtin_fmaf(rad.z * bx.z, ShadowFactor, Direct.z));
        } else {
            rad = rad * att;
            Direct = zr3(__builtin_fmaf(rad.x, bx.x, Direct.x), __builtin_fmaf(rad.y, bx.y, Direct.y), __builtin_fmaf(rad.z, bx.z, Direct.z));
        }
    }
    // (2) indirect, BaseLighting.frag:210
    Indirect = zr3((((DiffuseColor.x * ZR_INV_PI) * AO) * 0.3f) * ShadowFactor,
                             (((DiffuseColor.y * ZR_INV_PI) * AO) * 0.3f) * ShadowFactor,
                             (((DiffuseColor.z * ZR_INV_PI) * AO) * 0.3f) * ShadowFactor);
    // (3) reflection, :213-221
    const zf3 bcl = zr3(zr_clamp(BaseColor.x, 0.04f, 1.0f), zr_clamp(BaseColor.y, 0.04f, 1.0f), zr_clamp(BaseColor.z, 0.04f, 1.0f));
    const float dsf0 = (0.04f * 2.0f) * 0.5f;
    const zf3 RSpec = zr3(__builtin_fmaf(Metallic, bcl.x, (1.0f - Metallic) * dsf0), __builtin_fmaf(Metallic, bcl.y, (1.0f - Metallic) * dsf0),
                          __builtin_fmaf(Metallic, bcl.z, (1.0f - Metallic) * dsf0));
    // EnvBRDFApproxLazarov, SH/Common.glsl:201-211
    const float rx = __builtin_fmaf(Roughness, -1.0f, 1.0f), ry = __builtin_fmaf(Roughness, -0.0275f, 0.0425f);
    const float rz = __builtin_fmaf(Roughness, -0.572f, 1.04f), rw = __builtin_fmaf(Roughness, 0.022f, -0.04f);
    const float a004 = __builtin_fmaf(__builtin_fminf(rx * rx, zr_exp2(-9.28f * NdotV)), rx, ry);
    const float ABx = __builtin_fmaf(-1.04f, a004, rz), ABy = __builtin_fmaf(1.04f, a004, rw);
    const float F90 = zr_saturate(50.0f * RSpec.y);
    const zf3 RBRDF = zr3(__builtin_fmaf(RSpec.x, ABx, F90 * ABy), __builtin_fmaf(RSpec.y, ABx, F90 * ABy), __builtin_fmaf(RSpec.z, ABx, F90 * ABy));
    const float eta = 1.00f / 1.52f;
    const float dNI = zr_dot(Nn, Vv);
    const float kk = __builtin_fmaf(-(eta * eta), __builtin_fmaf(-dNI, dNI, 1.0f), 1.0f);
    zf3 R;
    if (kk < 0.0f) R = zr3(0.0f, 0.0f, 0.0f);
    else {
        const float q = __builtin_fmaf(eta, dNI, __builtin_sqrtf(kk));
        R = zr3(__builtin_fmaf(eta, Vv.x, -(q * Nn.x)), __builtin_fmaf(eta, Vv.y, -(q * Nn.y)), __builtin_fmaf(eta, Vv.z, -(q * Nn.z)));
    }
    // ComputeReflectionMipFromRoughness, SH/Common.glsl:191-198
    const float MIPS = (maxmips - 1.0f) - __builtin_fmaf(-1.2f, zr_log2(__builtin_fmaxf(Roughness, 0.001f)), 1.0f);
    const zf3 RL = (ZR_DIAG_SKIP(L.debug_skip) & 4u) ? zr3(0.0f, 0.0f, 0.0f) : cube_sample(C, slut, L.cube_dim, (int)L.cube_levels, R, MIPS) * 10.0f;
    const float RV = zr_saturate((zr_pow(NdotV + AO, Roughness * Roughness) - 1.0f) + AO);   // GetSpecularOcclusion :226
    RefC = zr3((RL.x * RV) * RBRDF.x, (RL.y * RV) * RBRDF.y, (RL.z * RV) * RBRDF.z);
}

// BaseLighting.frag:147-254 for every pixel of the owned tiles (the full-screen quad of ZE:3531-3540)
// Shape of the per-pixel kernels' grids (A/B'd on the whole two-lane frame, not on the kernel alone: what counts is what the pass
// leaves to the other lane while it runs).  Pixels per thread 1 instead of 4: + 3.5 % (the pass itself takes LONGER beside the camera
// lane, 123 -> 187 us, and the camera lane's short kernels stop starving: hiz 80 -> 40 us); 512-thread workgroups for the lighting
// pass: + 1 % more (128 threads: - 8 %, 1 024: - 2 %; single-wave workgroups at 4 pixels per thread: - 3 %).  One pixel per thread
// is also what a rank of a multi-GPU job needs, whose few tiles would otherwise fill a quarter of the machine.
#ifndef ZR_LIGHT_TB
#define ZR_LIGHT_TB 512
#endif
#ifndef ZR_LIGHT_WAVES
#define ZR_LIGHT_WAVES 4             // waves per SIMD k_lighting is compiled for (see the note at the kernel)
#endif
#ifndef ZR_PIXELS_PER_THREAD
#define ZR_PIXELS_PER_THREAD 1       // of k_resolve_gbuffer and k_lighting: 1, 2 or 4 (a tile is 1 024 pixels; workgroups per tile follow)
#endif
static_assert(TILE_PIX / ZR_PIXELS_PER_THREAD >= ZR_LIGHT_TB && TILE_PIX / ZR_PIXELS_PER_THREAD >= 256, "a tile's threads must fill at least one workgroup");
#ifndef ZR_RESOLVE_TB
#define ZR_RESOLVE_TB 256
#endif
template <bool LIGHT_LIST, bool BACKGROUND, int TB, int PPT>      // PPT: pixels per thread, 4 or 1 (as in k_resolve_gbuffer)
// (compiled for exactly 4 waves per SIMD: left to itself the allocator takes 127 VGPRs, told so it makes do with 97 - the same four
// waves, but 120 registers per SIMD left for the other lane's kernels; 5 or 6 waves (95 / 80 VGPRs) are faster alone, not beside)
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(ZR_LIGHT_WAVES, ZR_LIGHT_WAVES))) void k_lighting(ZrLightParams L, const XkView* __restrict__ view,
                                                  const uint32_t* __restrict__ owned_tiles, GBufferPtrs G,
                                                  const float* __restrict__ shadowmap, CubeDesc C,
                                                  const float* __restrict__ srgb_lut, const float* __restrict__ unorm_lut,
                                                  uint32_t* __restrict__ out)
{
    // UNORM loads are IEEE quotients c / 255 and c / 1023: 14 per pixel, served from an LDS copy of the host-built table
    __shared__ float tl[512];            // [0, 256) sRGB decode (24 cubemap fetches per pixel), [256, 512) c / 255: tex_decode's layout
    __shared__ float u10[1024];
    float* const slut = tl; float* const u8 = tl + 256;
    constexpr uint32_t T = TILE_PIX / (uint32_t)PPT, PARTS = T / (uint32_t)TB, WAVES = (uint32_t)TB / 64u;
    const uint32_t tid = threadIdx.x + (blockIdx.x % PARTS) * (uint32_t)TB;      // the thread's place among the tile's T
    const uint32_t tile_slot = blockIdx.x / PARTS;
    for (uint32_t i = threadIdx.x; i < 256u; i += (uint32_t)TB) { u8[i] = unorm_lut[i]; slut[i] = srgb_lut[i]; }
    for (uint32_t i = threadIdx.x; i < 1024u; i += (uint32_t)TB) u10[i] = unorm_lut[256u + i];
    __syncthreads();
    if (L.clear_next) {      // the clear of the next frame's shadow pass (depth 1.0, ZE:3248), a slice per workgroup: saves a launch
        const uint32_t per = (L.clear_n + gridDim.x - 1u) / gridDim.x, b = blockIdx.x * per;
        for (uint32_t i = threadIdx.x; i < per && b + i < L.clear_n; i += (uint32_t)TB) L.clear_next[b + i] = 0x3F800000u;
    }
    const uint32_t tile = owned_tiles[tile_slot];
    const int tx0 = (int)(tile % L.tiles_x) * TILE, ty0 = (int)(tile / L.tiles_x) * TILE;
    const zf3 cam = zr3(view->CameraInfo[0], view->CameraInfo[1], view->CameraInfo[2]);
    const uint32_t nDir = (uint32_t)view->LightsCount[0], nPoint = (uint32_t)view->LightsCount[1];
    const float maxmips = (float)(uint32_t)view->LightsCount[3];
    const float dxy = 1.5f * 1.0f / (float)L.SD;

    // Tile light list (LIGHT_LIST: four or more point lights): a light whose sphere of influence misses the bounding box of the tile's world
    // positions would be skipped by every pixel's own exact test below (|lp - P| >= the box distance per axis, and the squared
    // sums are monotonic), so it is dropped for the whole tile.  The list is a bitmask, walked in ascending order: the
    // accumulation order over lights is unchanged.  Pixels with Mask = 0 do not count: their colour is (...) * 0 -> stored 0.
    __shared__ float bbp[LIGHT_LIST ? WAVES : 1][6];
    __shared__ uint32_t lmask[LIGHT_LIST ? XK_MAX_POINT_LIGHTS_NUM / 32 : 1];
    constexpr bool use_mask = LIGHT_LIST;
    if constexpr (LIGHT_LIST) {
        float lo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, hi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
        bool odd = false;                    // a non-finite position: keep every light
        for (uint32_t i = tid; i < TILE_PIX; i += T) {
            const int px = tx0 + (int)(i & (TILE - 1)), py = ty0 + (int)(i / TILE);
            if (px >= (int)L.W || py >= (int)L.H) continue;
            const size_t p = (size_t)py * L.W + (size_t)px;
            if ((G.scene_color[p] >> 24) == 0u) continue;
            const uint2 D = G.gD[p];
            const float q[3] = { f16_to_f32_hw(D.x & 0xFFFFu), f16_to_f32_hw(D.x >> 16), f16_to_f32_hw(D.y & 0xFFFFu) };
            for (int a = 0; a < 3; ++a) {
                if (!(__builtin_fabsf(q[a]) <= 3.402823466e38f)) odd = true;
                lo[a] = __builtin_fminf(lo[a], q[a]); hi[a] = __builtin_fmaxf(hi[a], q[a]);
            }
        }
        for (int a = 0; a < 3; ++a) { lo[a] = wave_fmin(lo[a]); hi[a] = wave_fmax(hi[a]); }
        const bool wodd = __ballot(odd) != 0ull;
        if ((threadIdx.x & 63u) == 0u) {
            float* o = bbp[threadIdx.x >> 6];
            o[0] = wodd ? -__builtin_inff() : lo[0]; o[1] = wodd ? -__builtin_inff() : lo[1]; o[2] = wodd ? -__builtin_inff() : lo[2];
            o[3] = wodd ? __builtin_inff() : hi[0]; o[4] = wodd ? __builtin_inff() : hi[1]; o[5] = wodd ? __builtin_inff() : hi[2];
        }
        for (uint32_t i = threadIdx.x; i < XK_MAX_POINT_LIGHTS_NUM / 32; i += (uint32_t)TB) lmask[i] = 0u;
        __syncthreads();
        float blo[3], bhi[3];
        for (int a = 0; a < 3; ++a) {
            blo[a] = bbp[0][a]; bhi[a] = bbp[0][3 + a];
            for (uint32_t w = 1; w < WAVES; ++w) { blo[a] = __builtin_fminf(blo[a], bbp[w][a]); bhi[a] = __builtin_fmaxf(bhi[a], bbp[w][3 + a]); }
        }
        for (uint32_t li = threadIdx.x; li < nPoint; li += (uint32_t)TB) {
            const XkLight* __restrict__ Lt = &view->PointLights[li];
            const bool lfinite = __builtin_fabsf(Lt->Color[0]) <= 3.402823466e38f && __builtin_fabsf(Lt->Color[1]) <= 3.402823466e38f &&
                                 __builtin_fabsf(Lt->Color[2]) <= 3.402823466e38f && __builtin_fabsf(Lt->Color[3]) <= 3.402823466e38f;
            const float falloff = Lt->Direction[3];
            bool keep = true;
            if (lfinite && falloff > 0.0f) {
                zf3 e;      // per axis: how far the light lies outside the box (0 inside); |lp - P| is at least that for every P in it
                e.x = __builtin_fmaxf(0.0f, __builtin_fmaxf(blo[0] - Lt->Position[0], Lt->Position[0] - bhi[0]));
                e.y = __builtin_fmaxf(0.0f, __builtin_fmaxf(blo[1] - Lt->Position[1], Lt->Position[1] - bhi[1]));
                e.z = __builtin_fmaxf(0.0f, __builtin_fmaxf(blo[2] - Lt->Position[2], Lt->Position[2] - bhi[2]));
                if (zr_dot(e, e) > (falloff * falloff) * 1.000001f) keep = false;
            }
            if (keep) atomicOr(&lmask[li >> 5], 1u << (li & 31u));
        }
        __syncthreads();
    }

    for (uint32_t i = tid; i < TILE_PIX; i += T) {
        const int px = tx0 + (int)(i & (TILE - 1)), py = ty0 + (int)(i / TILE);
        if (px >= (int)L.W || py >= (int)L.H) continue;
        const size_t p = (size_t)py * L.W + (size_t)px;
        const uint32_t sc = G.scene_color[p], A = G.gA[p], B = G.gB[p], Cc = G.gC[p];
        const uint2 D = G.gD[p];
        // what every path ends with: the skydome / background drawn over the lit quad in view 0 (ZE:3681-3699), then the store
        auto emit = [&](uint32_t rgba) {
            if (L.debug_view == 0u) {
                const uint32_t ov = L.has_overlay ? G.overlay[p] : 0u;
                if (ov) rgba = ov;
                else if (BACKGROUND && L.bg_enabled && 1.0f <= G.depth[p]) {
                    const float u = ((float)px + 0.5f) / (float)L.W, v = ((float)py + 0.5f) / (float)L.H;
                    const float one4[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
                    const zf4 bgc = tex_sample<2>(L.bg, one4, true, tl, u, v, 1.0f / (float)L.W, 0.0f, 0.0f, 1.0f / (float)L.H);
                    rgba = zr_unorm(zr_pow(bgc.x, 0.4545f), 255.0f) | zr_unorm(zr_pow(bgc.y, 0.4545f), 255.0f) << 8 |
                           zr_unorm(zr_pow(bgc.z, 0.4545f), 255.0f) << 16 | 255u << 24;
                }
            }
            if (L.packed_out) out[(size_t)tile_slot * TILE_PIX + i] = rgba;
            else out[p] = rgba;
        };
        // A pixel nothing was drawn to holds the clear values of every target (ZE:3427-3433), so the shader computes the same
        // colour for all of them: it was computed once (zr_launch_lighting's one-pixel pre-launch of this very kernel).
        if (L.empty_rgba != nullptr && sc == 0xFF000000u && A == 0u && B == 0xFF000000u && Cc == 0xFF000000u && D.x == 0u && D.y == 0x3C000000u) {
            emit(*L.empty_rgba);
            continue;
        }
        const zf3 BaseColor = zr3(u8[Cc & 255u], u8[(Cc >> 8) & 255u], u8[(Cc >> 16) & 255u]);
        const float Metallic = zr_saturate(u8[B & 255u]);
        float Roughness = zr_saturate(u8[(B >> 16) & 255u]);
        const zf3 Normal = zr3(__builtin_fmaf(u10[(A >> 20) & 1023u], 2.0f, -1.0f), __builtin_fmaf(u10[(A >> 10) & 1023u], 2.0f, -1.0f),
                               __builtin_fmaf(u10[A & 1023u], 2.0f, -1.0f));
        const float AO = zr_saturate(u8[Cc >> 24]);
        const float Mask = u8[sc >> 24];
        Roughness = __builtin_fmaxf(0.01f, Roughness);
        const zf3 N = zr_normalize(Normal);
        const zf3 Pw = zr3(f16_to_f32_hw(D.x & 0xFFFFu), f16_to_f32_hw(D.x >> 16), f16_to_f32_hw(D.y & 0xFFFFu));
        zf3 Direct, Indirect, RefC; float ShadowFactor;
        shade_surface<use_mask>(L, view, shadowmap, C, slut, lmask, nDir, nPoint, maxmips, dxy, cam, BaseColor, Metallic, Roughness, N, AO, Pw,
                                Direct, Indirect, RefC, ShadowFactor);

        zf3 Final = ((Direct + Indirect) + RefC) * Mask;
        Final = zr3(zr_pow(Final.x, 0.4545f), zr_pow(Final.y, 0.4545f), zr_pow(Final.z, 0.4545f));
        zf3 o;
        switch (L.debug_view) {
        case 0: o = Final; break;
        case 1: o = zr3(zr_pow(BaseColor.x, 0.4545f), zr_pow(BaseColor.y, 0.4545f), zr_pow(BaseColor.z, 0.4545f)); break;
        case 2: o = zr3(Metallic, Metallic, Metallic); break;
        case 3: o = zr3(Roughness, Roughness, Roughness); break;
        case 4: o = Normal; break;
        case 5: o = zr3(AO, AO, AO); break;
        case 7: o = RefC; break;
        case 8: o = zr3(ShadowFactor, ShadowFactor, ShadowFactor); break;
        case 9: o = Final; break;       // GBufferVis: k_gbuffer_vis then overwrites the eight mosaic cells
        case 6: {   // fragColor of the full-screen quad: Background.vert:10-17 vertex colours over its two triangles
            const float u = ((float)px + 0.5f) / (float)L.W, v = ((float)py + 0.5f) / (float)L.H;
            o = v >= u ? zr3(1.0f - v, u, v - u) : zr3(1.0f - u, v, u - v);
            break;
        }
        default: o = Final * ShadowFactor; break;
        }
        emit(zr_unorm(o.x, 255.0f) | zr_unorm(o.y, 255.0f) << 8 | zr_unorm(o.z, 255.0f) << 16 | 255u << 24);
    }
}

// GBufferVis (SH/BaseLighting.frag:42-145, SPEC_CONSTANTS 9).  Runs after k_lighting has written FinalColor everywhere: the
// lighting quad samples every GBuffer target again at UV = fragTexCoord * 3 / (1 - EmptyRatio) through LINEAR / REPEAT samplers
// (ZE:2811-2847) and shows a 3 x 3 mosaic; the centre cell and whatever lies outside the cells keep FinalColor.  Bilinear
// weights are snapped to 8 fractional bits (a stated choice, like sampling hardware): with EmptyRatio = 0 every sample is
// exactly texel (3x + 1, 3y + 1) mod (W, H).
struct GTexel { float v[20]; };     // SceneColor, GBufferA, B, C, D as five vec4
__device__ __forceinline__ GTexel gbuffer_texel(const GBufferPtrs& G, uint32_t W, int x, int y)
{
    const size_t p = (size_t)y * W + (size_t)x;
    const uint32_t sc = G.scene_color[p], A = G.gA[p], B = G.gB[p], C = G.gC[p];
    const uint2 D = G.gD[p];
    GTexel t;
    for (int k = 0; k < 4; ++k) {
        t.v[k] = (float)((sc >> (8 * k)) & 255u) / 255.0f;
        t.v[8 + k] = (float)((B >> (8 * k)) & 255u) / 255.0f;
        t.v[12 + k] = (float)((C >> (8 * k)) & 255u) / 255.0f;
    }
    t.v[4] = (float)((A >> 20) & 1023u) / 1023.0f; t.v[5] = (float)((A >> 10) & 1023u) / 1023.0f;
    t.v[6] = (float)(A & 1023u) / 1023.0f; t.v[7] = (float)(A >> 30) / 3.0f;
    t.v[16] = zr_f16_to_f32(D.x & 0xFFFFu); t.v[17] = zr_f16_to_f32(D.x >> 16);
    t.v[18] = zr_f16_to_f32(D.y & 0xFFFFu); t.v[19] = zr_f16_to_f32(D.y >> 16);
    return t;
}
__device__ __forceinline__ int wrap_index(int i, int n) { const int m = i % n; return m < 0 ? m + n : m; }

__global__ __launch_bounds__(256) void k_gbuffer_vis(ZrLightParams L, const XkView* __restrict__ view, GBufferPtrs G,
                                                     const float* __restrict__ shadowmap, CubeDesc C,
                                                     const float* __restrict__ srgb_lut, uint32_t* __restrict__ out)
{
    const uint32_t px = blockIdx.x * 16u + (threadIdx.x & 15u), py = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (px >= L.W || py >= L.H) return;
    const float ERx = view->ViewportInfo[2] / view->ViewportInfo[0], ERy = view->ViewportInfo[3] / view->ViewportInfo[1];
    const float tx = ((float)px + 0.5f) / (float)L.W, ty = ((float)py + 0.5f) / (float)L.H;        // fragTexCoord
    const float UVx = (tx * 3.0f) / (1.0f - ERx), UVy = (ty * 3.0f) / (1.0f - ERy);
    const float Sx = (1.0f - ERx) / 3.0f, Sy = (1.0f - ERy) / 3.0f;                                 // Step
    int cell = -1; float bx = 0.0f, by = 0.0f;
    if (tx < Sx && ty < Sy) { cell = 0; bx = 1.0f; by = 1.0f; }
    else if (tx < Sx * 2.0f && ty < Sy) { cell = 1; bx = 2.0f; by = 1.0f; }
    else if (tx < Sx * 3.0f && ty < Sy) { cell = 2; bx = 3.0f; by = 1.0f; }
    else if (tx < Sx && ty < Sy * 2.0f) { cell = 3; bx = 1.0f; by = 2.0f; }
    else if (tx < 1.0f && ty < Sy * 2.0f && tx > Sx * 2.0f) { cell = 4; bx = 3.0f; by = 2.0f; }
    else if (tx < Sx && ty < Sx * 3.0f) { cell = 5; bx = 1.0f; by = 3.0f; }                         // Step.x * 3: as the shader has it
    else if (tx < Sx * 2.0f && tx > Sx && ty < Sy * 3.0f && ty > Sy * 2.0f) { cell = 6; bx = 2.0f; by = 3.0f; }
    else if (tx < Sx * 3.0f && tx > Sx * 2.0f && ty < Sy * 3.0f && ty > Sy * 2.0f) { cell = 7; bx = 3.0f; by = 3.0f; }
    if (cell < 0) return;                                                                           // FinalColor stays
    zf3 o;
    if (tx > Sx * (bx - ERx) || ty > Sy * (by - ERy)) o = zr3(1.0f, 1.0f, 1.0f);                      // the cells' white frames
    else {
        // texture(sampler2D, UV): one mip level, LINEAR, REPEAT
        float x = __builtin_fmaf(UVx, (float)L.W, -0.5f), y = __builtin_fmaf(UVy, (float)L.H, -0.5f);
        if (!(__builtin_fabsf(x) < 1.0e9f)) x = 0.0f;
        if (!(__builtin_fabsf(y) < 1.0e9f)) y = 0.0f;
        const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
        const float ax = __builtin_floorf(__builtin_fmaf(x - fx, 256.0f, 0.5f)) / 256.0f;
        const float ay = __builtin_floorf(__builtin_fmaf(y - fy, 256.0f, 0.5f)) / 256.0f;
        const int x0 = wrap_index((int)fx, (int)L.W), x1 = wrap_index((int)fx + 1, (int)L.W);
        const int y0 = wrap_index((int)fy, (int)L.H), y1 = wrap_index((int)fy + 1, (int)L.H);
        const GTexel t00 = gbuffer_texel(G, L.W, x0, y0), t10 = gbuffer_texel(G, L.W, x1, y0);
        const GTexel t01 = gbuffer_texel(G, L.W, x0, y1), t11 = gbuffer_texel(G, L.W, x1, y1);
        float g[20];
        for (int k = 0; k < 20; ++k) {
            const float top = __builtin_fmaf(ax, t10.v[k] - t00.v[k], t00.v[k]), bot = __builtin_fmaf(ax, t11.v[k] - t01.v[k], t01.v[k]);
            g[k] = __builtin_fmaf(ay, bot - top, top);
        }
        const zf3 BaseColor = zr3(g[12], g[13], g[14]);
        const float Metallic = zr_saturate(g[8]);
        const float Roughness = __builtin_fmaxf(0.01f, zr_saturate(g[10]));
        const zf3 Normal = zr3(__builtin_fmaf(g[4], 2.0f, -1.0f), __builtin_fmaf(g[5], 2.0f, -1.0f), __builtin_fmaf(g[6], 2.0f, -1.0f));
        const float AO = zr_saturate(g[15]);
        const zf3 N = zr_normalize(Normal);
        const zf3 Pw = zr3(g[16], g[17], g[18]);
        switch (cell) {
        case 0: o = zr3(zr_pow(BaseColor.x, 0.4545f), zr_pow(BaseColor.y, 0.4545f), zr_pow(BaseColor.z, 0.4545f)); break;
        case 1: o = zr3(Metallic, Metallic, Metallic); break;
        case 2: o = zr3(Roughness, Roughness, Roughness); break;
        case 3: o = N; break;
        case 4: o = zr3(AO, AO, AO); break;
        case 5: o = zr3(0.0f, 0.0f, 0.0f); break;
        case 6: {
            const zf3 cam = zr3(view->CameraInfo[0], view->CameraInfo[1], view->CameraInfo[2]);
            const zf3 Vv = zr_normalize(cam - Pw), Nn = zr_normalize(N);
            const float eta = 1.00f / 1.52f;
            const float dNI = zr_dot(Nn, Vv);
            const float kk = __builtin_fmaf(-(eta * eta), __builtin_fmaf(-dNI, dNI, 1.0f), 1.0f);
            zf3 R;
            if (kk < 0.0f) R = zr3(0.0f, 0.0f, 0.0f);
            else {
                const float q = __builtin_fmaf(eta, dNI, __builtin_sqrtf(kk));
                R = zr3(__builtin_fmaf(eta, Vv.x, -(q * Nn.x)), __builtin_fmaf(eta, Vv.y, -(q * Nn.y)), __builtin_fmaf(eta, Vv.z, -(q * Nn.z)));
            }
            o = cube_sample(C, srgb_lut, L.cube_dim, (int)L.cube_levels, R, 0.0f) * 10.0f;
            break;
        }
        default: {
            const zf4 s4 = zr_mat4_point(L.SB, Pw);
            const float sx = s4.x / s4.w, sy = s4.y / s4.w, sz = s4.z / s4.w, sw = s4.w / s4.w;
            const float dxy = 1.5f * 1.0f / (float)L.SD;
            float sum = 0.0f;
            for (int xo = -2; xo <= 2; ++xo)
                for (int yo = -2; yo <= 2; ++yo) sum += shadow_tap(shadowmap, (int)L.SD, sx, sy, sz, sw, dxy * (float)xo, dxy * (float)yo);
            const float sf = sum * 0.04f;
            o = zr3(sf, sf, sf);
            break;
        }
        }
    }
    out[(size_t)py * L.W + px] = zr_unorm(o.x, 255.0f) | zr_unorm(o.y, 255.0f) << 8 | zr_unorm(o.z, 255.0f) << 16 | 255u << 24;
}

// ------------------------------------------------------------------------------------------------ forward variant
// Base.frag:46-144 - the engine built with ENABLE_DEFERRED_SHADING false (ZE:93): the main render pass clears colour (0,0,0,1) and depth
// (ZE:3517-3519, 2366-2373) and Base.frag shades every fragment that passes LESS straight into the swapchain image (pipelines
// ZE:2749-2801, draws ZE:3544-3680).  Here the winner of the depth test is already known per pixel (the resolve keeps its primitive id in
// G.prim when the context shades forward), so the shader runs once per covered pixel, as with an early depth test and no overdraw.
// Against the deferred pair: the fetched material and ComputeNormal()'s result are used as floats (no render-target format in between),
// AO is not saturated, there is no Mask, every view multiplies FinalColor by ShadowFactor AFTER the gamma (:114-121), and the debug table
// is Base.frag's own (:123-143: base colour without gamma, AmbientOcclution.rgb, the interpolated vertex colour, no GBufferVis).
// The skydome and the background follow in the same render pass (view 0 only, ZE:3681-3699) exactly as in the deferred frame.
template <bool IMAGES>
__global__ __launch_bounds__(256) void k_forward(ZrPass P, ZrLightParams L, const XkView* __restrict__ view, const ZrObject* __restrict__ objs,
                                                 const uint32_t* __restrict__ owned_tiles, GBufferPtrs G, const float* __restrict__ shadowmap,
                                                 CubeDesc C, const float* __restrict__ srgb_lut, const float* __restrict__ unorm_lut,
                                                 uint32_t* __restrict__ out)
{
    __shared__ float tl[512];            // [0, 256) sRGB decode, [256, 512) c / 255: tex_decode's layout
    for (uint32_t i = threadIdx.x; i < 256u; i += 256u) { tl[i] = srgb_lut[i]; tl[256u + i] = unorm_lut[i]; }
    __syncthreads();
    const uint32_t tile_slot = blockIdx.x / (TILE_PIX / 256u), i = threadIdx.x + (blockIdx.x % (TILE_PIX / 256u)) * 256u;
    const uint32_t tile = owned_tiles[tile_slot];
    const int px = (int)(tile % L.tiles_x) * TILE + (int)(i & (TILE - 1)), py = (int)(tile / L.tiles_x) * TILE + (int)(i / TILE);
    if (px >= (int)L.W || py >= (int)L.H) return;
    const size_t p = (size_t)py * L.W + (size_t)px;
    const uint32_t prim = G.prim[p];
    uint32_t rgba = 0xFF000000u;         // clearValues[0].color, ZE:3517
    if (prim != ZR_EMPTY_PRIM) {
        const PixGeom g = pixel_geom(P, objs, prim, px, py);
        const ZrObject* __restrict__ O = g.O;
        // texture(samplerN, fragTexCoord), Base.frag:50-54 (emissive and mask are bound but not fetched)
        zf4 ms[ZR_MATERIAL_SLOTS];
        if (IMAGES) tex_sample_material(O, tl, g.u0, g.v0, g.s1, g.t1, g.s2, g.t2, ms);
        else for (int k = 0; k < ZR_MATERIAL_SLOTS; ++k) { ms[k].x = O->texc[k][0]; ms[k].y = O->texc[k][1]; ms[k].z = O->texc[k][2]; ms[k].w = O->texc[k][3]; }
        const zf3 BaseColor = zr3(ms[0].x, ms[0].y, ms[0].z);
        const float Metallic = zr_saturate(ms[1].x);
        const float Roughness = __builtin_fmaxf(0.01f, zr_saturate(ms[2].x));
        const zf3 ts = (O->const_slots & 8u) ? zr3(O->ts_const[0], O->ts_const[1], O->ts_const[2]) : zr_tangent_space_normal(zr3(ms[3].x, ms[3].y, ms[3].z));
        const zf3 Normal = compute_normal(g.pos_dx, g.pos_dy, g.s1, g.t1, g.s2, g.t2, g.N0, ts);
        const zf3 AmbientOcclution = zr3(ms[4].x, ms[4].y, ms[4].z);
        const zf3 cam = zr3(view->CameraInfo[0], view->CameraInfo[1], view->CameraInfo[2]);
        zf3 Direct, Indirect, RefC; float ShadowFactor;
        shade_surface<false>(L, view, shadowmap, C, tl, nullptr, (uint32_t)view->LightsCount[0], (uint32_t)view->LightsCount[1],
                             (float)(uint32_t)view->LightsCount[3], 1.5f * 1.0f / (float)L.SD, cam,
                             BaseColor, Metallic, Roughness, Normal, AmbientOcclution.x, g.P0, Direct, Indirect, RefC, ShadowFactor);
        zf3 Final = (Direct + Indirect) + RefC;
        Final = zr3(zr_pow(Final.x, 0.4545f), zr_pow(Final.y, 0.4545f), zr_pow(Final.z, 0.4545f));
        zf3 o;
        switch (L.debug_view) {
        case 1: o = BaseColor; break;
        case 2: o = zr3(Metallic, Metallic, Metallic); break;
        case 3: o = zr3(Roughness, Roughness, Roughness); break;
        case 4: o = Normal; break;
        case 5: o = AmbientOcclution; break;
        case 6: {   // fragColor = inColor (Base.vert:28), interpolated like every other varying
            const uint32_t* __restrict__ ix = O->indices + 3u * g.tri;
            const XkVertex* __restrict__ v0 = O->verts + ld_global(ix), * __restrict__ v1 = O->verts + ld_global(ix + 1), * __restrict__ v2 = O->verts + ld_global(ix + 2);
            o = interp3(g.b0, zr3(v0->Color[0], v0->Color[1], v0->Color[2]), zr3(v1->Color[0], v1->Color[1], v1->Color[2]),
                        zr3(v2->Color[0], v2->Color[1], v2->Color[2]));
            break;
        }
        case 7: o = RefC; break;
        case 8: o = zr3(ShadowFactor, ShadowFactor, ShadowFactor); break;
        default: o = Final * ShadowFactor; break;      // cases 0, 9 and default
        }
        rgba = zr_unorm(o.x, 255.0f) | zr_unorm(o.y, 255.0f) << 8 | zr_unorm(o.z, 255.0f) << 16 | 255u << 24;
    }
    if (L.debug_view == 0u) {            // skydome, then the background quad at depth 1 (ZE:3681-3699)
        const uint32_t ov = L.has_overlay ? G.overlay[p] : 0u;
        if (ov) rgba = ov;
        else if (L.bg_enabled && 1.0f <= G.depth[p]) {
            const float u = ((float)px + 0.5f) / (float)L.W, v = ((float)py + 0.5f) / (float)L.H;
            const float one4[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
            const zf4 bgc = tex_sample<2>(L.bg, one4, true, tl, u, v, 1.0f / (float)L.W, 0.0f, 0.0f, 1.0f / (float)L.H);
            rgba = zr_unorm(zr_pow(bgc.x, 0.4545f), 255.0f) | zr_unorm(zr_pow(bgc.y, 0.4545f), 255.0f) << 8 |
                   zr_unorm(zr_pow(bgc.z, 0.4545f), 255.0f) << 16 | 255u << 24;
        }
    }
    if (L.packed_out) out[(size_t)tile_slot * TILE_PIX + i] = rgba;
    else out[p] = rgba;
}

// Multi-GPU composite: gathered[rank][slot][TILE_PIX] -> frame; tile_map[t] = owner * slots_per_rank + slot of tile t
__global__ __launch_bounds__(256) void k_untile(const uint32_t* __restrict__ gathered, const uint32_t* __restrict__ tile_map,
                                                uint32_t* __restrict__ frame, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles)
{
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t* src = gathered + (size_t)tile_map[tile] * TILE_PIX;
    const uint32_t tx0 = (tile % tiles_x) * TILE, ty0 = (tile / tiles_x) * TILE;
    for (uint32_t i = threadIdx.x; i < TILE_PIX; i += 256u) {
        const uint32_t px = tx0 + (i & (TILE - 1)), py = ty0 + i / TILE;
        if (px < W && py < H) frame[(size_t)py * W + px] = src[i];
    }
}

// The other direction, for a plane that is NOT written tile-major by its producer (the shadow map): plane -> packed[slot][TILE_PIX] for
// the tiles in `tiles` (slot = place in the list); texels beyond the plane's edge are filled with `pad`.
__global__ __launch_bounds__(256) void k_pack_tiles(const uint32_t* __restrict__ plane, const uint32_t* __restrict__ tiles, uint32_t* __restrict__ packed,
                                                    uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t pad)
{
    const uint32_t tile = tiles[blockIdx.x];
    uint32_t* dst = packed + (size_t)blockIdx.x * TILE_PIX;
    const uint32_t tx0 = (tile % tiles_x) * TILE, ty0 = (tile / tiles_x) * TILE;
    for (uint32_t i = threadIdx.x; i < TILE_PIX; i += 256u) {
        const uint32_t px = tx0 + (i & (TILE - 1)), py = ty0 + i / TILE;
        dst[i] = (px < W && py < H) ? plane[(size_t)py * W + px] : pad;
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
void zr_launch_bin_count(const ZrPass& P, const uint32_t* work, uint32_t* rects, uint32_t* tile_count, const ZrHiz& Z, ZrDevStats* stats,
                         int slot, hipStream_t s)
{
    if (P.n_work == 0) return;
    const uint32_t n_tiles = P.tiles_x * P.tiles_y;
    hipLaunchKernelGGL(k_bin_count, dim3((P.n_work + 1023) / 1024), dim3(1024), n_tiles * sizeof(uint32_t), s, P, work, rects, tile_count, Z, stats, slot);
}
void zr_launch_hiz_build(const unsigned long long* vis64, uint32_t W, uint32_t H, const ZrHiz& Z, const uint32_t* regions, uint32_t n_regions, hipStream_t s)
{
    if (n_regions) hipLaunchKernelGGL(k_hiz_build, dim3(n_regions), dim3(256), 0, s, vis64, W, H, Z, regions);
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
void zr_launch_frame_begin(ZrDevStats* stats, const XkView* view_src_pinned, XkView* view_dst, uint32_t rebuild_lists, hipStream_t s)
{
    static_assert(sizeof(XkView) % 4 == 0 && offsetof(ZrDevStats, overflow_sticky) % 4 == 0, "dword copies");
    hipLaunchKernelGGL(k_frame_begin, dim3(view_src_pinned ? 4 : 1), dim3(1024), 0, s, (uint32_t*)stats, (uint32_t)(offsetof(ZrDevStats, overflow_sticky) / 4),
                       (const uint32_t*)view_src_pinned, (uint32_t*)view_dst, (uint32_t)(sizeof(XkView) / 4), &stats->n_vis_work[1], rebuild_lists);
}
void zr_launch_fill32(uint32_t* p, uint32_t v, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_fill32, dim3(1024), dim3(256), 0, s, p, v, n);
}
void zr_launch_fill64(unsigned long long* p, unsigned long long v, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_fill64, dim3(1024), dim3(256), 0, s, p, v, n);
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
void zr_launch_resolve_gbuffer(const ZrPass& P, const ZrObject* objs, const uint32_t* owned_tiles, uint32_t n_owned,
                               unsigned long long* vis64, const GBufferPtrs& G, const float* srgb_lut, const float* unorm_lut, uint8_t* vis_now,
                               ZrDevStats* stats, hipStream_t s, uint32_t vis_mark)
{
    if (n_owned == 0) return;
    // one pixel per thread (ZR_PIXELS_PER_THREAD): see the note above k_lighting
#define ZR_LAUNCH_RESOLVE(IM, TB, PPT) hipLaunchKernelGGL((k_resolve_gbuffer<IM, TB, PPT>), dim3(n_owned * (TILE_PIX / (PPT) / (TB))), dim3(TB), 0, s, P, objs, owned_tiles, vis64, G, srgb_lut, unorm_lut, vis_now, stats, vis_mark)
    if (P.images == 1u) ZR_LAUNCH_RESOLVE(1, 64, ZR_PIXELS_PER_THREAD);
    else if (P.images) ZR_LAUNCH_RESOLVE(2, 64, ZR_PIXELS_PER_THREAD);
    else ZR_LAUNCH_RESOLVE(0, ZR_RESOLVE_TB, ZR_PIXELS_PER_THREAD);
#undef ZR_LAUNCH_RESOLVE
}
void zr_launch_sky_tiles(const ZrPass& P, const ZrObject* objs, const uint32_t* owned_tiles, uint32_t n_owned, unsigned long long* sky64, hipStream_t s)
{
    if (n_owned) hipLaunchKernelGGL(k_sky_tiles, dim3(n_owned), dim3(256), 0, s, P, objs, owned_tiles, sky64);
}
void zr_launch_count_shadow(const uint32_t* bits, size_t n, ZrDevStats* stats, hipStream_t s)
{
    hipLaunchKernelGGL(k_count_shadow, dim3(256), dim3(256), 0, s, bits, n, stats);
}
void zr_launch_lighting(const ZrLightParams& L, const XkView* view, const uint32_t* owned_tiles, uint32_t n_owned,
                        const GBufferPtrs& G, const float* shadowmap, const CubeDesc& C, const float* lut, const float* unorm_lut,
                        uint32_t* out, hipStream_t s)
{
    if (n_owned == 0) return;
    // with several point lights each tile first builds its light list (L.light_list: decided on the host from the light count)
#define ZR_LAUNCH_LIGHTING(LL, BG) hipLaunchKernelGGL((k_lighting<LL, BG, ZR_LIGHT_TB, ZR_PIXELS_PER_THREAD>), dim3(n_owned * (TILE_PIX / ZR_PIXELS_PER_THREAD / ZR_LIGHT_TB)), dim3(ZR_LIGHT_TB), 0, s, L, view, owned_tiles, G, shadowmap, C, lut, unorm_lut, out)
    if (L.light_list) { if (L.bg_enabled) ZR_LAUNCH_LIGHTING(true, true); else ZR_LAUNCH_LIGHTING(true, false); }
    else { if (L.bg_enabled) ZR_LAUNCH_LIGHTING(false, true); else ZR_LAUNCH_LIGHTING(false, false); }
#undef ZR_LAUNCH_LIGHTING
}
void zr_launch_forward(const ZrPass& P, const ZrLightParams& L, const XkView* view, const ZrObject* objs, const uint32_t* owned_tiles, uint32_t n_owned,
                       const GBufferPtrs& G, const float* shadowmap, const CubeDesc& C, const float* lut, const float* unorm_lut, uint32_t* out, hipStream_t s)
{
    if (n_owned == 0) return;
    if (P.images) hipLaunchKernelGGL((k_forward<true>), dim3(n_owned * (TILE_PIX / 256u)), dim3(256), 0, s, P, L, view, objs, owned_tiles, G, shadowmap, C, lut, unorm_lut, out);
    else hipLaunchKernelGGL((k_forward<false>), dim3(n_owned * (TILE_PIX / 256u)), dim3(256), 0, s, P, L, view, objs, owned_tiles, G, shadowmap, C, lut, unorm_lut, out);
}
void zr_launch_gbuffer_vis(const ZrLightParams& L, const XkView* view, const GBufferPtrs& G, const float* shadowmap, const CubeDesc& C,
                           const float* lut, uint32_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_gbuffer_vis, dim3((L.W + 15) / 16, (L.H + 15) / 16), dim3(256), 0, s, L, view, G, shadowmap, C, lut, out);
}
void zr_launch_untile(const uint32_t* gathered, const uint32_t* tile_map, uint32_t* frame, uint32_t W, uint32_t H, uint32_t tiles_x,
                      uint32_t n_tiles, hipStream_t s)
{
    hipLaunchKernelGGL(k_untile, dim3(n_tiles), dim3(256), 0, s, gathered, tile_map, frame, W, H, tiles_x, n_tiles);
}
void zr_launch_pack_tiles(const uint32_t* plane, const uint32_t* tiles, uint32_t n_tiles, uint32_t* packed, uint32_t W, uint32_t H, uint32_t tiles_x,
                          uint32_t pad, hipStream_t s)
{
    if (n_tiles) hipLaunchKernelGGL(k_pack_tiles, dim3(n_tiles), dim3(256), 0, s, plane, tiles, packed, W, H, tiles_x, pad);
}
