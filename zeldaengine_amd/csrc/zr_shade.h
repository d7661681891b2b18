// zr_shade.h — SH/Common.glsl's lighting functions (BxDF :134-282, PCF :294-342, Apply*Light :364-416), textureLod of the cubemap, and
// shade_surface: the body BaseLighting.frag:174-221 and Base.frag:62-112 share word for word.
#pragma once
#include "zr_dev.h"

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
