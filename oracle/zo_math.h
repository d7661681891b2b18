/*
 * zo_math.h — scalar fp32 math of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  The product (zeldaengine_amd/) never includes this.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or KATs and its
 * renderer cannot be built here (needs Vulkan + 9 absent submodules), so this
 * restatement is pinned only by hand-derived known-answer values (SURVEY App. C,
 * tests/test_oracle_kat.py) and by the golden livelink JSON.
 *
 * GLSL leaves the precision of sin/cos/pow/exp2/log2 and the evaluation order of
 * fp expressions to the implementation.  The oracle fixes ONE evaluation: every
 * expression below is written op by op (explicit fmaf, IEEE / and sqrtf, inversesqrt
 * as the fixed fma sequence zo_rsqrt; a vector / scalar is one IEEE reciprocal and
 * multiplies; x / PI and x / 25 are multiplications by the rounded reciprocal) and is
 * compiled with -ffp-contract=off, so results are reproducible bit for bit.  The
 * transcendental functions are small polynomial kernels (Cody-Waite reduction +
 * Cephes-style minimax coefficients) instead of libm, for the same reason.
 */
#ifndef ZO_MATH_H
#define ZO_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } zo_v3;
typedef struct { float x, y, z, w; } zo_v4;

static inline uint32_t zo_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float zo_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---- transcendental kernels (all fp32, fixed op order) ---- */

/* sin and cos of x (radians).  k = rint(x*2/pi); r = x - k*pi/2 in three fma steps. */
static inline void zo_sincosf(float x, float* s, float* c)
{
    float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(r2, ps, -1.6666654611e-1f);
    float sn = fmaf(r2 * r, ps, r);
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(r2, pc, 4.166664568298827e-2f);
    float cs = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
    int q = ((int)k) & 3;
    float so = (q & 1) ? cs : sn;
    float co = (q & 1) ? sn : cs;
    if (q == 1 || q == 2) co = -co;
    if (q >= 2) so = -so;
    *s = so; *c = co;
}

/* 2^x.  x < -126 -> 0, x >= 128 -> +inf, NaN -> NaN. */
static inline float zo_exp2f(float x)
{
    if (!(x >= -126.0f)) return (x != x) ? x : 0.0f;
    if (x >= 128.0f) return INFINITY;
    float n = rintf(x);
    float f = x - n;
    float p = fmaf(f, 1.5252733804059840e-5f, 1.5403530393381606e-4f);
    p = fmaf(f, p, 1.3333558146428443e-3f);
    p = fmaf(f, p, 9.618129107628477e-3f);
    p = fmaf(f, p, 5.550410866482158e-2f);
    p = fmaf(f, p, 2.402265069591007e-1f);
    p = fmaf(f, p, 6.931471805599453e-1f);
    p = fmaf(f, p, 1.0f);
    int ni = (int)n;
    if (ni > 127) { p *= 2.0f; ni -= 1; }      /* x in [127.5,128): keep the scale a normal float */
    return p * zo_u2f((uint32_t)(ni + 127) << 23);
}

/* log2(x).  x == 0 -> -inf, x < 0 -> NaN, inf -> inf. */
static inline float zo_log2f(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x *= 8388608.0f; e = -23; }
    uint32_t u = zo_f2u(x);
    e += (int)(u >> 23) - 127;
    float m = zo_u2f((u & 0x007FFFFFu) | 0x3F800000u);   /* [1,2) */
    if (m > 1.41421356237f) { m *= 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = fmaf(f, 7.0376836292e-2f, -1.1514610310e-1f);
    p = fmaf(f, p, 1.1676998740e-1f);
    p = fmaf(f, p, -1.2420140846e-1f);
    p = fmaf(f, p, 1.4249322787e-1f);
    p = fmaf(f, p, -1.6668057665e-1f);
    p = fmaf(f, p, 2.0000714765e-1f);
    p = fmaf(f, p, -2.4999993993e-1f);
    p = fmaf(f, p, 3.3333331174e-1f);
    float ln = fmaf(f * z, p, fmaf(z, -0.5f, f));
    return fmaf(ln, 1.44269504088896341f, (float)e);
}

/* GLSL pow(x, y) = exp2(y * log2(x)) (the spec's own definition of its precision). */
static inline float zo_powf(float x, float y) { return zo_exp2f(y * zo_log2f(x)); }

/* pow(x, 5.0) of F_Schlick (Common.glsl:136) as repeated multiplication; x >= 0 there. */
static inline float zo_pow5f(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }

/* ---- GLSL helpers, fixed op order ---- */
static inline float zo_saturate(float t) { return fminf(fmaxf(t, 0.0f), 1.0f); }            /* Common.glsl:23 */
static inline float zo_clampf(float t, float a, float b) { return fminf(fmaxf(t, a), b); }
static inline zo_v3 zo_v3make(float x, float y, float z) { zo_v3 r = { x, y, z }; return r; }
static inline zo_v3 zo_add(zo_v3 a, zo_v3 b) { return zo_v3make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline zo_v3 zo_sub(zo_v3 a, zo_v3 b) { return zo_v3make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline zo_v3 zo_mul(zo_v3 a, zo_v3 b) { return zo_v3make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline zo_v3 zo_scale(zo_v3 a, float s) { return zo_v3make(a.x * s, a.y * s, a.z * s); }
static inline float zo_dot(zo_v3 a, zo_v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline zo_v3 zo_cross(zo_v3 a, zo_v3 b)
{
    return zo_v3make(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float zo_length(zo_v3 a) { return sqrtf(zo_dot(a, a)); }
/* inversesqrt(x) of the shader stages.  GLSL allows 2 ulp and leaves x <= 0 undefined; the evaluation fixed here (and, op for op,
 * in csrc/zr_math.h: zr_rsqrt): integer seed, one third-order step y (1 + e/2 + 3 e^2/8) with e = 1 - x y^2, one residual step
 * y + (y/2) e.  Error < 1.2 ulp over every normal float (tests/test_oracle_kat.py).  NaN -> NaN, +inf -> 0; zero, denormals
 * (flushed, as GLSL allows) and negatives -> +inf, so normalize(0) is NaN as with 1 / sqrt. */
static inline float zo_rsqrt(float x)
{
    float y = zo_u2f(0x5F3759DFu - (zo_f2u(x) >> 1));
    float t = x * y, e = fmaf(-t, y, 1.0f);
    y = fmaf(y, e * fmaf(0.375f, e, 0.5f), y);
    t = x * y; e = fmaf(-t, y, 1.0f);
    y = fmaf(y * 0.5f, e, y);
    float sp = (x == INFINITY) ? 0.0f : INFINITY;
    return ((x >= 1.17549435e-38f && x < INFINITY) || x != x) ? y : sp;
}
/* -DZO_LITERAL (oracle/_build/liboracle_literal.so, oracle/CONTRACT.md): the IEEE-literal reading of the same GLSL - inversesqrt as
 * 1 / sqrt, every `/` a correctly rounded division per component, texels converted (/ 255) before they are filtered.  It is what this
 * oracle was before the contract fixed cheaper admissible forms; tests/test_oracle_contract.py renders both and bounds the distance. */
#ifdef ZO_LITERAL
#define ZO_IS_LITERAL 1
static inline float zo_shader_rsqrt(float x) { return 1.0f / sqrtf(x); }
#else
#define ZO_IS_LITERAL 0
static inline float zo_shader_rsqrt(float x) { return zo_rsqrt(x); }
#endif
/* normalize(v) = v * inversesqrt(dot(v,v)) in shader code */
static inline zo_v3 zo_normalize(zo_v3 a) { return zo_scale(a, zo_shader_rsqrt(zo_dot(a, a))); }
/* vec3 / scalar (contract: one IEEE reciprocal and three multiplies; literal: three divisions) */
static inline zo_v3 zo_div_scalar(zo_v3 a, float s)
{
#ifdef ZO_LITERAL
    return zo_v3make(a.x / s, a.y / s, a.z / s);
#else
    const float r = 1.0f / s;
    return zo_v3make(a.x * r, a.y * r, a.z * r);
#endif
}
/* glm::normalize on the host (lookAt): v * (1 / sqrt(dot)), IEEE - the engine's own x86 arithmetic, not a shader's */
static inline zo_v3 zo_normalize_ieee(zo_v3 a) { return zo_scale(a, 1.0f / sqrtf(zo_dot(a, a))); }
#define ZO_INV_PI 0.318309886f     /* x / PI (Common.glsl) is evaluated as x * fl(1 / 3.14159265359) */
#ifdef ZO_LITERAL
static inline float zo_div_pi(float x) { return x / 3.14159265359f; }
static inline float zo_div_25(float x) { return x / 25.0f; }
#else
static inline float zo_div_pi(float x) { return x * ZO_INV_PI; }
static inline float zo_div_25(float x) { return x * 0.04f; }
#endif

/* column-major mat4 (glm): m[c*4+r] */
static inline void zo_mat4_mul(const float* A, const float* B, float* C) /* C = A*B; C may not alias */
{
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            C[c * 4 + r] = fmaf(A[12 + r], B[c * 4 + 3], fmaf(A[8 + r], B[c * 4 + 2],
                           fmaf(A[4 + r], B[c * 4 + 1], A[r] * B[c * 4 + 0])));
}
/* M * vec4(p, 1) */
static inline zo_v4 zo_mat4_point(const float* m, zo_v3 p)
{
    zo_v4 r;
    r.x = fmaf(m[8], p.z, fmaf(m[4], p.y, m[0] * p.x)) + m[12];
    r.y = fmaf(m[9], p.z, fmaf(m[5], p.y, m[1] * p.x)) + m[13];
    r.z = fmaf(m[10], p.z, fmaf(m[6], p.y, m[2] * p.x)) + m[14];
    r.w = fmaf(m[11], p.z, fmaf(m[7], p.y, m[3] * p.x)) + m[15];
    return r;
}

/* fp32 -> fp16 bits, round-to-nearest-even, overflow -> inf, NaN -> qNaN (VK_FORMAT_R16G16B16A16_SFLOAT store) */
static inline uint16_t zo_f32_to_f16(float f)
{
    uint32_t u = zo_f2u(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);
    if (a >= 0x47800000u) {                      /* >= 65536 -> inf; 65520..65536 rounds to inf below */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (a < 0x38800000u) {                       /* subnormal half or zero */
        if (a < 0x33000000u) return (uint16_t)sign;          /* < 2^-25 -> 0 */
        uint32_t e = a >> 23;                                /* 102..112 */
        uint32_t m = (a & 0x007FFFFFu) | 0x00800000u;
        uint32_t shift = 126u - e;                           /* 14..24 */
        uint32_t h = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((a - 0x38000000u) >> 13);
    uint32_t rem = a & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);                 /* carry into exponent handles 65520 -> inf */
}
static inline float zo_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return zo_u2f(sign);
        float v = (float)m * 5.9604644775390625e-8f;         /* m * 2^-24 */
        return sign ? -v : v;
    }
    if (e == 31) return zo_u2f(sign | 0x7F800000u | (m << 13));
    return zo_u2f(sign | ((e + 112u) << 23) | (m << 13));
}

/* float -> UNORM with N bits: NaN -> 0, clamp, round half up on c*max+0.5 (fixed evaluation) */
static inline uint32_t zo_unorm(float c, float maxv)
{
    c = fminf(fmaxf(c, 0.0f), 1.0f);
    return (uint32_t)floorf(fmaf(c, maxv, 0.5f));
}

#endif
