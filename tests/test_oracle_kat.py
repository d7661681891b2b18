"""Known-answer tests of the CPU oracle and accuracy of its fixed-order math kernels.

The reference ships no tests or vectors (parity unpinned); these hand-derived values are what pins the restatement.
"""
import ctypes as C
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L(oracle_lib):
    return oracle_lib.lib()


def test_brdf_terms(L):
    assert L.zo_kat_D_GGX(1.0, 0.5) == pytest.approx(1.2732395, rel=2e-6)
    assert L.zo_kat_D_GGX(0.8, 0.5) == pytest.approx(0.2942954, rel=2e-6)
    assert L.zo_kat_V_SmithGGXCorrelated(0.7, 0.6, 0.5) == pytest.approx(0.5121564, rel=2e-6)
    assert L.zo_kat_F_Schlick(0.04, 1.0, 0.5) == pytest.approx(0.07, rel=2e-6)
    assert L.zo_kat_Fr_DisneyDiffuse(0.7, 0.6, 0.9, 0.5) == pytest.approx(0.8317577, rel=2e-6)
    ab = (C.c_float * 2)()
    L.zo_kat_EnvBRDFApproxLazarov(0.5, 0.7, ab)
    assert ab[0] == pytest.approx(0.7183388, rel=3e-6) and ab[1] == pytest.approx(0.0066612, abs=2e-6)
    assert L.zo_kat_ReflectionMip(0.5, 11.0) == pytest.approx(7.8, abs=2e-6)
    assert L.zo_kat_ReflectionMip(1.0, 11.0) == pytest.approx(9.0, abs=2e-6)


def test_default_normal_and_srgb(L):
    ts = (C.c_float * 3)()
    L.zo_kat_default_normal_ts(ts)
    assert list(ts) == pytest.approx([-0.2701231, -0.2701231, 0.9241575], abs=2e-6)
    assert L.zo_kat_srgb8_to_linear(127) == pytest.approx(0.2122308, rel=2e-6)
    assert L.zo_kat_srgb8_to_linear(0) == 0.0 and L.zo_kat_srgb8_to_linear(255) == 1.0


def test_glm_matrices(L):
    m = (C.c_float * 16)()
    L.zo_kat_perspective(45.0, 16.0 / 9.0, 0.1, 45.0, m)
    assert m[0] == pytest.approx(1.3579951, rel=2e-6) and m[5] == pytest.approx(2.4142136, rel=2e-6)
    assert m[10] == pytest.approx(-1.0022272, rel=2e-6) and m[11] == -1.0 and m[14] == pytest.approx(-0.1002227, rel=2e-6)
    eye = (C.c_float * 3)(5, 5, 5); ctr = (C.c_float * 3)(0, 0, 0); up = (C.c_float * 3)(0, 0, 1)
    L.zo_kat_lookat(eye, ctr, up, m)
    M = np.array(list(m), dtype=np.float64).reshape(4, 4).T       # column-major -> math layout
    R = M[:3, :3]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and np.linalg.det(R) > 0
    assert np.allclose(M @ np.array([5, 5, 5, 1.0]), [0, 0, 0, 1], atol=1e-5)       # the eye maps to the origin
    assert np.allclose(M @ np.array([0, 0, 0, 1.0]), [0, 0, -math.sqrt(75), 1], atol=1e-5)   # RH: looks down -z


def test_inversesqrt_of_the_shader_stages(L):
    """zo_rsqrt (= csrc/zr_math.h: zr_rsqrt): the fixed fma sequence that stands for GLSL inversesqrt in normalize() / distance().
    GLSL allows 2 ulp; the sequence stays below 1.2 ulp of the exact value over the whole range (every 997th normal float here, all
    2^31 of them in the scratch run quoted in DESIGN.md section 4), and the special values are the IEEE ones."""
    assert L.zo_kat_rsqrt(1.0) == 1.0 and L.zo_kat_rsqrt(4.0) == 0.5 and L.zo_kat_rsqrt(0.25) == 2.0
    assert L.zo_kat_rsqrt(2.0) == pytest.approx(0.70710678, rel=1.5e-7) and L.zo_kat_rsqrt(3.0) == pytest.approx(0.57735027, rel=1.5e-7)
    assert L.zo_kat_rsqrt_worst(0x00800000, 0x7F800000, 997) < 1.2
    assert L.zo_kat_rsqrt_worst(0x3F000000, 0x40800000, 1) < 1.2            # every float of [0.5, 4): all mantissas, both exponent parities
    inf = float("inf")
    assert L.zo_kat_rsqrt(0.0) == inf and L.zo_kat_rsqrt(inf) == 0.0 and math.isnan(L.zo_kat_rsqrt(float("nan")))
    assert L.zo_kat_rsqrt(1e-40) == inf and L.zo_kat_rsqrt(-1.0) == inf      # denormals are flushed; x < 0 is undefined in GLSL


def test_sincos_accuracy(L):
    out = (C.c_float * 2)()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-700, 700, 4000), rng.uniform(-4, 4, 2000), [0.0, math.pi, 565.48]]).astype(np.float32)
    worst = 0.0
    for x in xs:
        L.zo_kat_sincos(float(x), out)
        worst = max(worst, abs(out[0] - math.sin(float(x))), abs(out[1] - math.cos(float(x))))
    assert worst < 3e-7        # Vulkan allows 2^-11 absolute error for sin/cos; this is ~2 ulp at 1.0


def test_exp2_log2_pow_accuracy(L):
    rng = np.random.default_rng(1)
    for x in rng.uniform(-20, 20, 3000).astype(np.float32):
        r = L.zo_kat_exp2(float(x))
        assert r == pytest.approx(2.0 ** float(x), rel=4e-7)
    for x in np.exp(rng.uniform(-30, 30, 3000)).astype(np.float32):
        r = L.zo_kat_log2(float(x))
        assert abs(r - math.log2(float(x))) <= 4e-7 * max(1.0, abs(math.log2(float(x))))
    for x in rng.uniform(0, 12, 3000).astype(np.float32):
        assert L.zo_kat_pow(float(x), 0.4545) == pytest.approx(float(x) ** 0.4545, rel=3e-6, abs=1e-30)
    assert L.zo_kat_pow(0.0, 0.4545) == 0.0
    assert math.isnan(L.zo_kat_pow(-1.0, 0.4545))
    assert L.zo_kat_exp2(-200.0) == 0.0 and math.isinf(L.zo_kat_exp2(200.0))
    assert L.zo_kat_log2(0.0) == -math.inf


def test_f16_conversion_matches_ieee(L):
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(-70000, 70000, 5000), rng.uniform(-1e-4, 1e-4, 3000), rng.uniform(-10, 10, 3000),
                         [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, 6.1e-5, 5.96e-8, 2.98e-8, 2.99e-8]]).astype(np.float32)
    want = xs.astype(np.float16).view(np.uint16)
    got = np.array([L.zo_kat_f32_to_f16(float(x)) for x in xs], dtype=np.uint16)
    assert np.array_equal(want, got)


def test_rot_matrix_conventions(L):
    """MakeRotMatrix quirk: R.x turns about Y, R.y about Z, R.z about X; (0, yaw, 0) is a rotation about Z."""
    e = (C.c_float * 3)(0.0, 0.5, 0.0)
    r = (C.c_float * 9)()
    L.zo_kat_rotmat(e, r)
    R = np.array(list(r)).reshape(3, 3)       # rows of this view = columns of the GLSL mat3
    c, s = math.cos(0.5), math.sin(0.5)
    assert np.allclose(R, [[c, s, 0], [-s, c, 0], [0, 0, 1]], atol=1e-6)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6)


def test_anisotropic_tap_count_and_lod(oracle_lib):
    """Vulkan's described anisotropic scheme: N = min(ceil(Pmax / Pmin), 16), lambda = log2(Pmax / N), clamped to the chain."""
    L = oracle_lib.lib()

    def q(ax, ay, bx, by, levels=8):
        out = np.zeros(3, dtype=np.float32)
        L.zo_kat_aniso(ax, ay, bx, by, levels, out.ctypes.data)
        return int(out[0]), float(out[1]), int(out[2])
    assert q(1, 0, 0, 1) == (1, 0.0, 1)                       # isotropic, magnification boundary
    assert q(4, 0, 0, 4) == (1, 2.0, 1)                       # isotropic minification: plain trilinear at log2(4)
    assert q(8, 0, 0, 1) == (8, 0.0, 1)                       # 8:1 along x: 8 taps of the base level
    assert q(0, 1, 8, 0) == (8, 0.0, 0)                       # the same along y
    n, l, _ = q(32, 0, 0, 1)
    assert n == 16 and abs(l - 1.0) < 1e-6                    # capped at 16 taps, the rest goes to the LOD
    n, l, _ = q(3.5, 0, 0, 1)
    assert n == 4 and l == 0.0                                # log2(3.5 / 4) < 0 clamps to the base level
    n, l, _ = q(6, 0, 0, 2)
    assert n == 3 and abs(l - 1.0) < 1e-6                     # Pmax / N = 2 texels
    assert q(5, 0, 0, 0)[0] == 16                             # degenerate minor axis
    assert q(0, 0, 0, 0) == (1, 0.0, 1)                       # no footprint at all (log2 0 = -inf clamps to level 0)
    assert q(1000, 0, 0, 1000, levels=4)[1] == 3.0            # clamped to the last level
