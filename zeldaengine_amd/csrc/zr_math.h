// zr_math.h — fp32 shader math of the HIP renderer (device + host, gfx950).
//
// Every kernel is compiled with -ffp-contract=off and no fast-math: each fma below is
// explicit, '/' and sqrtf are the IEEE-correct expansions (inversesqrt is zr_rsqrt, a fixed sequence of
// fmas: see there), and the transcendental kernels
// are small polynomial evaluations (the GLSL the engine ships leaves their precision to
// the driver; Vulkan's bounds are far looser than these).  That makes a frame a pure
// function of its inputs: the same scene gives the same bytes on every launch and on
// every GPU of a screen-tile partition, which the multi-GPU composite relies on.
//
// GLSL being restated: Engine/ZeldaEngine/Shaders/Common.glsl (SH/Common.glsl below).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZR_HD __host__ __device__ __forceinline__

// c / 255.0f for an integer c in 0..255 without a division: fma(c, ZR_UNORM8_HI, c * ZR_UNORM8_LO) is the correctly rounded quotient
// for all 256 values (k_hi = fl(1 / 255), k_lo = fl(1 / 255 - k_hi); checked exhaustively in exact arithmetic and again by zr_create)
#define ZR_UNORM8_HI 0x1.010102p-8f
#define ZR_UNORM8_LO -0x1.fdfdfep-33f

struct zf3 { float x, y, z; };
struct zf4 { float x, y, z, w; };

ZR_HD uint32_t zr_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
ZR_HD float zr_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

ZR_HD zf3 zr3(float x, float y, float z) { zf3 r; r.x = x; r.y = y; r.z = z; return r; }
ZR_HD zf3 operator+(zf3 a, zf3 b) { return zr3(a.x + b.x, a.y + b.y, a.z + b.z); }
ZR_HD zf3 operator-(zf3 a, zf3 b) { return zr3(a.x - b.x, a.y - b.y, a.z - b.z); }
ZR_HD zf3 operator*(zf3 a, float s) { return zr3(a.x * s, a.y * s, a.z * s); }
ZR_HD float zr_dot(zf3 a, zf3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
ZR_HD zf3 zr_cross(zf3 a, zf3 b)
{
    return zr3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
               __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
ZR_HD float zr_length(zf3 a) { return __builtin_sqrtf(zr_dot(a, a)); }
// inversesqrt(x) of the shader stages (normalize, distance).  GLSL gives inversesqrt 2 ulp and leaves x <= 0 undefined; an IEEE
// 1 / sqrt costs 27 instructions on this part, two of them quarter-rate.  This build fixes ONE cheaper evaluation (the CPU oracle states
// the same sequence): an integer seed (relative error 3.4 %), one third-order step y (1 + e/2 + 3 e^2/8) with e = 1 - x y^2 (-> 1.3e-5),
// one residual step y + (y/2) e - every fma explicit, 11 full-rate instructions.  Checked over every normal float against the
// correctly rounded value: error < 1.2 ulp, 85.6 % correctly rounded (the KAT tests pin samples and the bound).
// Specials: NaN -> NaN; +inf -> 0; zero, denormals (flushed, as GLSL allows) and negatives (undefined in GLSL) -> +inf, so
// normalize(0) is NaN as with the IEEE form.
ZR_HD float zr_rsqrt(float x)
{
    float y = zr_u2f(0x5F3759DFu - (zr_f2u(x) >> 1));
    float t = x * y, e = __builtin_fmaf(-t, y, 1.0f);
    y = __builtin_fmaf(y, e * __builtin_fmaf(0.375f, e, 0.5f), y);
    t = x * y; e = __builtin_fmaf(-t, y, 1.0f);
    y = __builtin_fmaf(y * 0.5f, e, y);
    const float sp = (x == __builtin_inff()) ? 0.0f : __builtin_inff();
    return ((x >= 1.17549435e-38f && x < __builtin_inff()) || x != x) ? y : sp;
}
// normalize(v) = v * inversesqrt(dot(v, v)) in shader code
ZR_HD zf3 zr_normalize(zf3 a) { return a * zr_rsqrt(zr_dot(a, a)); }
// glm::normalize on the HOST (lookAt, ZE:4612-4618): there the engine's own x86 code runs, v * (1 / sqrt(dot)) in IEEE arithmetic
ZR_HD zf3 zr_normalize_ieee(zf3 a) { return a * (1.0f / __builtin_sqrtf(zr_dot(a, a))); }
// 1 / 3.14159265359 (SH/Common.glsl PI): `x / PI` is evaluated as x * ZR_INV_PI (division is a 2.5-ulp operation in GLSL)
#define ZR_INV_PI 0.318309886f
// normalize(2.0 * normalize(texNormal) - 1.0), SH/Common.glsl:125-126: a per-object constant when the normal map is one texel
ZR_HD zf3 zr_tangent_space_normal(zf3 texN)
{
    const zf3 n = zr_normalize(texN);
    return zr_normalize(zr3(__builtin_fmaf(2.0f, n.x, -1.0f), __builtin_fmaf(2.0f, n.y, -1.0f), __builtin_fmaf(2.0f, n.z, -1.0f)));
}
ZR_HD float zr_saturate(float t) { return __builtin_fminf(__builtin_fmaxf(t, 0.0f), 1.0f); }   // SH/Common.glsl:23
ZR_HD float zr_clamp(float t, float a, float b) { return __builtin_fminf(__builtin_fmaxf(t, a), b); }

// M * vec4(p, 1), M column-major
ZR_HD zf4 zr_mat4_point(const float* m, zf3 p)
{
    zf4 r;
    r.x = __builtin_fmaf(m[8], p.z, __builtin_fmaf(m[4], p.y, m[0] * p.x)) + m[12];
    r.y = __builtin_fmaf(m[9], p.z, __builtin_fmaf(m[5], p.y, m[1] * p.x)) + m[13];
    r.z = __builtin_fmaf(m[10], p.z, __builtin_fmaf(m[6], p.y, m[2] * p.x)) + m[14];
    r.w = __builtin_fmaf(m[11], p.z, __builtin_fmaf(m[7], p.y, m[3] * p.x)) + m[15];
    return r;
}
// v * mat3(R): component j = dot(v, column j), R column-major 3x3
ZR_HD zf3 zr_rowvec_mat3(zf3 v, const float* R)
{
    return zr3(__builtin_fmaf(v.z, R[2], __builtin_fmaf(v.y, R[1], v.x * R[0])),
               __builtin_fmaf(v.z, R[5], __builtin_fmaf(v.y, R[4], v.x * R[3])),
               __builtin_fmaf(v.z, R[8], __builtin_fmaf(v.y, R[7], v.x * R[6])));
}
ZR_HD void zr_mat4_mul(const float* A, const float* B, float* C)
{
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            C[c * 4 + r] = __builtin_fmaf(A[12 + r], B[c * 4 + 3], __builtin_fmaf(A[8 + r], B[c * 4 + 2],
                           __builtin_fmaf(A[4 + r], B[c * 4 + 1], A[r] * B[c * 4 + 0])));
}
ZR_HD void zr_mat3_mul(const float* A, const float* B, float* C)
{
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
            C[c * 3 + r] = __builtin_fmaf(A[6 + r], B[c * 3 + 2], __builtin_fmaf(A[3 + r], B[c * 3 + 1], A[r] * B[c * 3 + 0]));
}

// ---- transcendental kernels: Cody-Waite reduction + short minimax polynomials -----------------------------

// sin/cos for the per-instance rotation (MakeRotMatrix, SH/Common.glsl:60-87); |x| up to ~1e5 rad
ZR_HD void zr_sincos(float x, float& s, float& c)
{
    float k = __builtin_rintf(x * 0.636619772367581343f);
    float r = __builtin_fmaf(-k, 1.5703125f, x);
    r = __builtin_fmaf(-k, 4.837512969970703125e-4f, r);
    r = __builtin_fmaf(-k, 7.54978995489188216e-8f, r);
    float r2 = r * r;
    float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(r2, ps, -1.6666654611e-1f);
    float sn = __builtin_fmaf(r2 * r, ps, r);
    float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(r2, pc, 4.166664568298827e-2f);
    float cs = __builtin_fmaf(r2 * r2, pc, __builtin_fmaf(r2, -0.5f, 1.0f));
    int q = ((int)k) & 3;
    float so = (q & 1) ? cs : sn;
    float co = (q & 1) ? sn : cs;
    if (q == 1 || q == 2) co = -co;
    if (q >= 2) so = -so;
    s = so; c = co;
}

ZR_HD float zr_exp2(float x)
{
    if (!(x >= -126.0f)) return (x != x) ? x : 0.0f;
    if (x >= 128.0f) return __builtin_inff();
    float n = __builtin_rintf(x);
    float f = x - n;
    float p = __builtin_fmaf(f, 1.5252733804059840e-5f, 1.5403530393381606e-4f);
    p = __builtin_fmaf(f, p, 1.3333558146428443e-3f);
    p = __builtin_fmaf(f, p, 9.618129107628477e-3f);
    p = __builtin_fmaf(f, p, 5.550410866482158e-2f);
    p = __builtin_fmaf(f, p, 2.402265069591007e-1f);
    p = __builtin_fmaf(f, p, 6.931471805599453e-1f);
    p = __builtin_fmaf(f, p, 1.0f);
    int ni = (int)n;
    if (ni > 127) { p *= 2.0f; ni -= 1; }
    return p * zr_u2f((uint32_t)(ni + 127) << 23);
}

ZR_HD float zr_log2(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x *= 8388608.0f; e = -23; }
    uint32_t u = zr_f2u(x);
    e += (int)(u >> 23) - 127;
    float m = zr_u2f((u & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356237f) { m *= 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = __builtin_fmaf(f, 7.0376836292e-2f, -1.1514610310e-1f);
    p = __builtin_fmaf(f, p, 1.1676998740e-1f);
    p = __builtin_fmaf(f, p, -1.2420140846e-1f);
    p = __builtin_fmaf(f, p, 1.4249322787e-1f);
    p = __builtin_fmaf(f, p, -1.6668057665e-1f);
    p = __builtin_fmaf(f, p, 2.0000714765e-1f);
    p = __builtin_fmaf(f, p, -2.4999993993e-1f);
    p = __builtin_fmaf(f, p, 3.3333331174e-1f);
    float ln = __builtin_fmaf(f * z, p, __builtin_fmaf(z, -0.5f, f));
    return __builtin_fmaf(ln, 1.44269504088896341f, (float)e);
}

// GLSL pow(x, y) := exp2(y * log2(x))
ZR_HD float zr_pow(float x, float y) { return zr_exp2(y * zr_log2(x)); }
// pow(1 - u, 5.0) in F_Schlick (SH/Common.glsl:136): the argument is never negative there
ZR_HD float zr_pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }

// ---- format conversion ---------------------------------------------------------------------------------

// UNORM store: NaN -> 0, clamp, floor(c * max + 0.5)
ZR_HD uint32_t zr_unorm(float c, float maxv)
{
    c = __builtin_fminf(__builtin_fmaxf(c, 0.0f), 1.0f);
    return (uint32_t)__builtin_floorf(__builtin_fmaf(c, maxv, 0.5f));
}

// fp32 -> fp16 bits, round to nearest even, overflow to inf (R16G16B16A16_SFLOAT store)
ZR_HD uint32_t zr_f32_to_f16(float f)
{
    uint32_t u = zr_f2u(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return sign | 0x7E00u;
    if (a >= 0x47800000u) return sign | 0x7C00u;
    if (a < 0x38800000u) {
        if (a < 0x33000000u) return sign;
        uint32_t e = a >> 23;
        uint32_t m = (a & 0x007FFFFFu) | 0x00800000u;
        uint32_t shift = 126u - e;
        uint32_t h = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return sign | h;
    }
    uint32_t h = (a - 0x38000000u) >> 13;
    uint32_t rem = a & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return sign | h;
}
ZR_HD float zr_f16_to_f32(uint32_t h)
{
    uint32_t sign = (h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return zr_u2f(sign);
        float v = (float)m * 5.9604644775390625e-8f;
        return sign ? -v : v;
    }
    if (e == 31) return zr_u2f(sign | 0x7F800000u | (m << 13));
    return zr_u2f(sign | ((e + 112u) << 23) | (m << 13));
}
