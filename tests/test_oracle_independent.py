"""The CPU oracle against an INDEPENDENT float64 evaluation of the reference's shaders (tests/independent_eval.py).

The reference cannot be run here and ships no golden images, so the oracle's parity is unpinned; the renderer and the
oracle share one author's reading of the GLSL and agree bit for bit by construction.  This test narrows the gap from the other
side: a second restatement, written from the GLSL text alone in another language, precision and evaluation order, must land on
the same GBuffer codes and the same lit colours - within one code (LSB) on at least 99.9 % of the pixels (SURVEY 8c's tolerance).
It takes the oracle's visibility buffer (who owns each pixel) as given; what stays unpinned is stated in DESIGN.md section 6.
"""
import math

import numpy as np
import pytest

import independent_eval as ie
from zeldaengine_amd import abi, scenes

DEFAULT_TEXELS = [(127, 127, 127, 255), (0, 0, 0, 255), (255, 255, 255, 255), (127, 127, 255, 255), (255, 255, 255, 255), (0, 0, 0, 255),
                  (255, 255, 255, 255)]                      # default_{grey,black,white,normal,white,black,white}.png, ZE:4951-4978
FACES = [(200, 60, 40, 255), (40, 180, 70, 255), (50, 80, 210, 255), (220, 200, 60, 255), (150, 150, 160, 255), (30, 30, 35, 255)]


def _const(rgba, n=2):
    return np.tile(np.array(rgba, dtype=np.uint8), (n, n, 1))


class Scene:
    """Collects draws for both the oracle and the independent evaluator (engine draw order: non-instanced first, ZE:3445-3476)."""

    def __init__(self):
        self.items = []

    def add(self, mesh, texels=None, instances=None):
        self.items.append({"verts": mesh[0], "idx": mesh[1], "texel": list(texels or DEFAULT_TEXELS), "instances": instances})

    def load(self, o):
        o.set_cubemap([_const(c, 4) for c in FACES])
        self._keep = []
        for it in self.items:
            mat = None
            if it["texel"] != DEFAULT_TEXELS:
                mat, k = abi.make_material([_const(t) for t in it["texel"]])
                self._keep.append(k)
            o.object_add(o.mesh_create(it["verts"], it["idx"]), mat, it["instances"])

    def draws(self):
        out, base = [], 0
        for instanced in (False, True):
            for it in self.items:
                if (it["instances"] is not None) != instanced:
                    continue
                d = dict(it)
                d["prim_base"] = base
                base += (len(it["idx"]) // 3) * (1 if it["instances"] is None else len(it["instances"]))
                out.append(d)
        return out


def _lights(n_dir, n_point):
    w = scenes.sample_world()
    d, _, s = scenes.lights_from_world(w)
    w["PointLights"] = scenes.sample_point_lights(n_point)
    _, p, _ = scenes.lights_from_world(w)
    for l in p:
        l["Direction"][3] = 6.0           # a radius that reaches the geometry
    return d[:n_dir], p, s


def scene_mixed():
    s = Scene()
    s.add(scenes.grid_plane(14.0, 3, 0.0))
    s.add(scenes.box((0.9, 0.6, 0.5), (1.2, -0.8, 0.5)), [(200, 40, 30, 255), (255, 255, 255, 255), (90, 90, 90, 255), (127, 127, 255, 255),
                                                            (180, 180, 180, 255), (0, 20, 40, 255), (255, 255, 255, 255)])
    s.add(scenes.uv_sphere(12, 6, 0.6), None, scenes.generate_instances(12, 1.0, 4.5, 0.5, 1.2, seed=3))
    return s, abi.make_camera((5.0, 4.0, 3.5), (0.0, 0.0, 0.4)), _lights(1, 4), 0.0


def scene_single_sphere_no_sun():
    """config 2's shape: no directional light -> lookAt(0, 0) = NaN shadow matrices -> every PCF tap returns 1"""
    s = Scene()
    s.add(scenes.uv_sphere(16, 8, 1.0))
    return s, abi.make_camera((1.9, 1.7, 1.3), (0.0, 0.0, 0.0)), _lights(0, 1), 0.0


def scene_rolled_and_clipped():
    """a rotated stage (model != identity), metallic / rough materials, instanced boxes, the ground crossing the near plane"""
    s = Scene()
    s.add(scenes.grid_plane(40.0, 2, 0.0), [(90, 140, 60, 255), (0, 0, 0, 255), (200, 200, 200, 255), (127, 127, 255, 255), (255, 255, 255, 255),
                                             (0, 0, 0, 255), (255, 255, 255, 255)])
    s.add(scenes.box((0.4, 0.4, 0.9), (0, 0, 0.9)), [(220, 220, 230, 255), (230, 230, 230, 255), (60, 60, 60, 255), (140, 120, 250, 255),
                                                      (255, 255, 255, 255), (10, 0, 0, 255), (255, 255, 255, 255)],
          scenes.generate_instances(9, 1.5, 5.0, 0.6, 1.4, seed=21))
    s.add(scenes.uv_sphere(10, 5, 0.8), None, scenes.generate_instances(5, 1.0, 3.0, 0.5, 1.0, seed=4))
    return s, abi.make_camera((3.0, -4.0, 1.2), (0.0, 0.0, 0.6), fov=60.0), _lights(1, 16), 0.35


SCENES = {"mixed": scene_mixed, "single_sphere_no_sun": scene_single_sphere_no_sun, "rolled_and_clipped": scene_rolled_and_clipped}


def _within_one(a, b):
    return np.all(np.abs(a - b) <= 1, axis=-1)


@pytest.mark.parametrize("name", sorted(SCENES))
def test_oracle_agrees_with_an_independent_float64_evaluation(oracle_lib, name):
    W, H, SD = 192, 128, 256
    scene, cam, (d, p, sp), roll = SCENES[name]()
    o = oracle_lib.Oracle(W, H, SD)
    scene.load(o)
    o.update_uniforms(cam, d, p, sp, roll, 0.0, 0.0)
    o.render(0)
    mvp, _sh, view = o.get_frame()
    prim = o.visibility()
    assert (prim != 0xFFFFFFFF).sum() > 0.2 * W * H

    # ---- BaseScene.frag: the five colour targets of every covered pixel, recomputed from the scene description
    mine = ie.base_scene(scene.draws(), mvp, prim, W, H)
    ys, xs = mine["yx"]
    got = {"scene_color": ie.unpack_rgba8(o.gbuffer(1))[ys, xs], "a": ie.unpack_a2r10g10b10(o.gbuffer(2))[ys, xs],
           "b": ie.unpack_rgba8(o.gbuffer(3))[ys, xs], "c": ie.unpack_rgba8(o.gbuffer(4))[ys, xs]}
    want = {"scene_color": ie.unorm(mine["scene_color"], 8), "b": ie.unorm(mine["b"], 8), "c": ie.unorm(mine["c"], 8),
            "a": np.concatenate([ie.unorm(mine["a"][:, :3], 10), ie.unorm(mine["a"][:, 3:], 2)], axis=1)}
    for k in ("scene_color", "b", "c"):
        assert np.array_equal(got[k], want[k]), "GBuffer %s: constant material slots must come out exactly" % k
    ok_a = _within_one(got["a"], want["a"])
    d_vals, d_codes = ie.unpack_rgba16f(o.gbuffer(5))
    mine_d = mine["d"].astype(np.float16).view(np.uint16)
    # one fp16 step - or what 1/256 of a pixel step moves the position by: the oracle interpolates from vertices snapped to the
    # 1/256-pixel grid (a stated raster choice), this evaluation from the unsnapped ones, and far away on a
    # grazing ground plane a pixel step spans more world units than an fp16 step
    budget = np.hstack([mine["d_per_pixel"] / 256.0, np.zeros((len(ys), 1))])
    ok_d = np.all((np.abs(ie.f16_ordinal(d_codes[ys, xs]) - ie.f16_ordinal(mine_d)) <= 1) | (np.abs(d_vals[ys, xs] - mine["d"]) <= budget), axis=-1)
    frac_a, frac_d = ok_a.mean(), ok_d.mean()
    assert frac_a >= 0.999, "normals (A2R10G10B10): only %.4f of %d pixels within one code" % (frac_a, len(ok_a))
    assert frac_d >= 0.999, "world position (fp16): only %.4f of %d pixels within one ulp" % (frac_d, len(ok_d))

    # ---- BaseLighting.frag: every pixel of the quad from the oracle's GBuffer, shadow map and uniforms
    gb = {"scene_color": ie.unpack_rgba8(o.gbuffer(1)) / 255.0, "b": ie.unpack_rgba8(o.gbuffer(3)) / 255.0, "c": ie.unpack_rgba8(o.gbuffer(4)) / 255.0,
          "a": ie.unpack_a2r10g10b10(o.gbuffer(2)) / np.array([1023.0, 1023.0, 1023.0, 3.0]), "d": d_vals}
    lit = ie.lighting(gb, o.shadowmap(), view, FACES, W, H)
    want_rgb = ie.unorm(lit, 8)
    have = o.color().astype(np.int64)
    ok = _within_one(have[..., :3], want_rgb)
    assert (have[..., 3] == 255).all()
    assert ok.mean() >= 0.999, "lit colour: only %.4f of the pixels within one LSB (worst %d)" % (ok.mean(), np.abs(have[..., :3] - want_rgb).max())
    assert len(np.unique(have.reshape(-1, 4), axis=0)) > 200        # a real picture: lights, shadow, reflection all vary


def test_rotation_matrix_convention_against_the_oracle(oracle_lib):
    """MakeRotMatrix + `v * mat3(rotMat)` (SH/BaseInstanced.vert:38-70): the naming trap (mx turns about Y ...) and the row-vector
    product, checked numerically against the oracle's KAT entry point for a few Euler triples."""
    import ctypes as C
    L = oracle_lib.lib()
    rng = np.random.default_rng(5)
    for _ in range(8):
        e = rng.uniform(-math.pi, math.pi, 3).astype(np.float32)
        out = np.zeros(9, dtype=np.float32)
        L.zo_kat_rotmat(e.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        mine = ie.make_rot_matrix(e.astype(np.float64))
        assert np.allclose(out.reshape(3, 3).T, mine, atol=2e-6)      # the oracle stores mat3(rotMat) column-major
