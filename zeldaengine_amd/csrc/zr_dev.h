// zr_dev.h — common ground of the kernel files (csrc/zr_*.hip): tile / wave constants, address-space-qualified loads, the vertex stage
// (Base.vert / BaseInstanced.vert / Shadowmap*.vert), wave reductions on the DPP network, the Hi-Z tests, the per-pixel kernels' grid shape.
//
// The render path is one .hip per pass; each compiles alone (-fno-gpu-rdc), what they share is header-only and inlined:
//   zr_cull.hip      k_instance_prep, k_cull_instances, k_cull_box<MODE> (+ the exact k_cull<MODE> of -DZR_DIAG builds)
//   zr_shadow.hip    shadow pass, meshlet-level binning: k_bin_count / k_scan / k_bin_fill, k_raster_chunks<MODE, HIZ, DEFER, LATE>,
//                    k_shadow_occlusion, k_count_shadow
//   zr_camera.hip    camera pass, triangle-level binning: k_hiz_build, k_select, k_geom<HIZ>, k_scan_tri, k_index, k_tile, k_sky_tiles
//   zr_resolve.hip   k_resolve_gbuffer: BaseScene.frag per pixel from the key buffer into the SoA GBuffer planes
//   zr_lighting.hip  k_lighting (BaseLighting.frag), k_gbuffer_vis (debug view 9)
//   zr_forward.hip   k_forward: the forward variant, Base.frag
//   zr_frame.hip     k_frame_begin, fills, k_untile / k_pack_tiles (multi-GPU composite)
//   zr_raster.h      the rasteriser proper (raster_sub, the clipper, k_tile_slow): shared by the shadow and the camera pass
//   zr_texture.h     texture(sampler2D): mips, trilinear, anisotropic; zr_surface.h: interpolation, ComputeNormal, BaseScene.frag's body
//   zr_shade.h       Common.glsl's BxDF, PCF, cubemap sampling and the body BaseLighting.frag and Base.frag share
//
// Replaces: SH/Shadowmap*.vert, SH/Base*.vert, SH/BaseScene.frag, SH/Background.vert + SH/BaseLighting.frag, SH/Base.frag,
// the fixed-function rasteriser/ROP state of RHICreateGraphicsPipelines (ZE:5094-5201) and the draw loops of
// RecordCommandBuffer (ZE:3239-3540).  Raster rules: DESIGN.md §4.
#pragma once
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

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)      // inclusive prefix sum over the wave's lanes (rows on the DPP network, then across)
{
    const int idn = 0;
    int r = (int)v;
    ZR_DPP_STEP(op_add, 0x111, 0xF); ZR_DPP_STEP(op_add, 0x112, 0xF); ZR_DPP_STEP(op_add, 0x114, 0xF); ZR_DPP_STEP(op_add, 0x118, 0xF);
    ZR_DPP_STEP(op_add, 0x142, 0xA); ZR_DPP_STEP(op_add, 0x143, 0xC);
    return (uint32_t)r;
}
__device__ __forceinline__ float lane_bcast(float v, uint32_t src) { return zr_u2f((uint32_t)__builtin_amdgcn_readlane((int)zr_f2u(v), (int)src)); }
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src); }

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

// Shape of the per-pixel kernels' grids (A/B'd on the whole two-lane frame, not on the kernel alone: what counts is what the pass
// leaves to the other lane while it runs).  Pixels per thread 1 instead of 4: + 3.5 % (the pass itself takes LONGER beside the camera
// lane, 123 -> 187 us, and the camera lane's short kernels stop starving: hiz 80 -> 40 us); 512-thread workgroups for the lighting
// pass: + 1 % more (128 threads: - 8 %, 1 024: - 2 %; single-wave workgroups at 4 pixels per thread: - 3 %).  One pixel per thread
// is also what a rank of a multi-GPU job needs, whose few tiles would otherwise fill a quarter of the machine.
#ifndef ZR_LIGHT_TB
#define ZR_LIGHT_TB 512
#endif
#ifndef ZR_LIGHT_WAVES
#define ZR_LIGHT_WAVES 5             // waves per SIMD k_lighting is compiled for (see the note at the kernel)
#endif
#ifndef ZR_PIXELS_PER_THREAD
#define ZR_PIXELS_PER_THREAD 1       // of k_resolve_gbuffer and k_lighting: 1, 2 or 4 (a tile is 1 024 pixels; workgroups per tile follow)
#endif
static_assert(TILE_PIX / ZR_PIXELS_PER_THREAD >= ZR_LIGHT_TB && TILE_PIX / ZR_PIXELS_PER_THREAD >= 256, "a tile's threads must fill at least one workgroup");
#ifndef ZR_RESOLVE_TB
#define ZR_RESOLVE_TB 256
#endif
