// zr_texture.h — texture(sampler2D, uv) as RHICreateSampler sets it up (ZE:6523-6557): RGBA8 mip chains, LINEAR / REPEAT, trilinear,
// anisotropic; the packed-material form (13 channels of seven same-sized slots in one 16-byte texel).  DESIGN.md §4 has the arithmetic.
#pragma once
#include "zr_dev.h"

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
