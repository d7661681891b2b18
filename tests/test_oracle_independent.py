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


def scene_random(seed):
    """seeded mixtures: 2-5 draws of plane / box / sphere with random constant materials (metallic, rough, emissive, masked, odd normal
    texels), with and without instances, 0-1 directional and 0-24 point lights, the camera anywhere around, the stage rolled"""
    import math
    rng = np.random.default_rng(7000 + seed)
    s = Scene()

    def texels():
        if rng.random() < 0.3:
            return None
        t = [tuple(int(x) for x in rng.integers(0, 256, 3)) + (255,) for _ in range(7)]
        t[3] = (int(rng.integers(100, 156)), int(rng.integers(100, 156)), int(rng.integers(200, 256)), 255)      # a plausible normal texel
        t[6] = (int(rng.choice([255, 255, 255, 0, 128])), 0, 0, 255)                                              # the lighting mask
        return t
    s.add(scenes.grid_plane(float(rng.choice([10.0, 24.0, 60.0])), int(rng.integers(2, 5)), 0.0), texels())
    for _ in range(int(rng.integers(1, 5))):
        kind = int(rng.integers(0, 3))
        mesh = [scenes.uv_sphere(12, 6, 0.6), scenes.uv_sphere(16, 8, 0.9), scenes.box((0.7, 0.5, 0.6), (0.0, 0.0, 0.6))][kind]
        inst = scenes.generate_instances(int(rng.integers(2, 20)), 0.8, float(rng.uniform(3.0, 9.0)), 0.4, 1.3, seed=int(rng.integers(1, 1 << 30))) if rng.random() < 0.7 else None
        s.add(mesh, texels(), inst)
    a, rad = rng.uniform(0, 2 * math.pi), float(rng.choice([3.5, 6.0, 11.0]))
    cam = abi.make_camera((rad * math.cos(a), rad * math.sin(a), float(rng.choice([0.6, 2.0, 5.0]))), (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), 0.4),
                          fov=float(rng.choice([40.0, 55.0, 70.0])))
    return s, cam, _lights(int(rng.integers(0, 2)), int(rng.choice([0, 1, 3, 8, 24]))), float(rng.uniform(0.0, 1.0))


SCENES = {"mixed": scene_mixed, "single_sphere_no_sun": scene_single_sphere_no_sun, "rolled_and_clipped": scene_rolled_and_clipped}
for _k in range(10):
    SCENES["random_%02d" % _k] = (lambda k: (lambda: scene_random(k)))(_k)


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
    assert (prim != 0xFFFFFFFF).sum() > (0.2 * W * H if not name.startswith("random") else 1500)

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
    # The shader's one discontinuity is the PCF comparison `dist < z`: where a tap's filtered depth and the reference depth agree to a few
    # float32 ulps, float32 and float64 may decide differently and the pixel jumps by a tap's worth of light.  Such pixels are found by
    # evaluating with the reference depth moved by +-4e-7 (a few ulps of a depth near 1) and accepted when the oracle's colour lies within
    # what that spans.
    lo_hi = [ie.unorm(ie.lighting(gb, o.shadowmap(), view, FACES, W, H, pcf_eps=e), 8) for e in (-4e-7, 4e-7)]
    lo, hi = np.minimum(np.minimum(lo_hi[0], lo_hi[1]), want_rgb) - 1, np.maximum(np.maximum(lo_hi[0], lo_hi[1]), want_rgb) + 1
    on_edge = (lo_hi[0] != lo_hi[1]).any(axis=-1)
    ok = ok | (on_edge & np.all((have[..., :3] >= lo) & (have[..., :3] <= hi), axis=-1))
    assert on_edge.mean() < 0.1
    assert (have[..., 3] == 255).all()
    assert ok.mean() >= 0.999, "lit colour: only %.4f of the pixels within one LSB (worst %d)" % (ok.mean(), np.abs(have[..., :3] - want_rgb).max())
    assert len(np.unique(have.reshape(-1, 4), axis=0)) > 200        # a real picture: lights, shadow, reflection all vary


@pytest.mark.parametrize("name", ["mixed", "single_sphere_no_sun", "rolled_and_clipped", "random_01", "random_04", "random_08"])
def test_oracle_forward_variant_agrees_with_an_independent_float64_evaluation(oracle_lib, name):
    """The forward variant (SH/Base.frag:46-123, zo_set_shading) against the same float64 statement of the GLSL: here nothing separates the
    stages (no GBuffer to read back), so the evaluator shades its OWN interpolants - from unsnapped vertices, where the oracle interpolates
    from vertices on the 1/256-pixel grid - and the agreement is that of the whole fragment shader."""
    W, H, SD = 192, 128, 256
    scene, cam, (d, p, sp), roll = SCENES[name]()
    o = oracle_lib.Oracle(W, H, SD)
    scene.load(o)
    o.update_uniforms(cam, d, p, sp, roll, 0.0, 0.0)
    o.set_shading(True)
    o.render(0)
    mvp, _sh, view = o.get_frame()
    prim = o.visibility()
    mine = ie.base_scene(scene.draws(), mvp, prim, W, H)
    ys, xs = mine["yx"]
    gb = {k: np.zeros((H, W, 4)) for k in ("scene_color", "a", "b", "c", "d")}
    gb["scene_color"][ys, xs] = mine["scene_color"]; gb["b"][ys, xs] = mine["b"]; gb["c"][ys, xs] = mine["c"]; gb["d"][ys, xs] = mine["d"]
    gb["a"][ys, xs, :3] = (mine["normal"] + 1.0) / 2.0          # lighting() undoes exactly this
    have = o.color().astype(np.int64)
    covered = prim != 0xFFFFFFFF
    assert (have[~covered] == (0, 0, 0, 255)).all(), "an empty pixel is the render pass's clear colour (ZE:3517)"
    shade = lambda eps: ie.unorm(ie.lighting(gb, o.shadowmap(), view, FACES, W, H, pcf_eps=eps, forward=True), 8)      # noqa: E731
    want, lo_hi = shade(0.0), [shade(-4e-7), shade(4e-7)]       # (the PCF comparison's ties, as in the deferred test above)
    ok = _within_one(have[..., :3], want)
    lo, hi = np.minimum(np.minimum(lo_hi[0], lo_hi[1]), want) - 1, np.maximum(np.maximum(lo_hi[0], lo_hi[1]), want) + 1
    on_edge = (lo_hi[0] != lo_hi[1]).any(axis=-1)
    ok = ok | (on_edge & np.all((have[..., :3] >= lo) & (have[..., :3] <= hi), axis=-1))
    frac = ok[covered].mean()
    print(name, "forward: within one LSB on %.5f of %d covered pixels, worst %d" % (frac, covered.sum(), np.abs(have[..., :3] - want)[covered].max()))
    assert frac >= 0.999, "forward colour: only %.4f of the covered pixels within one LSB" % frac
    assert (have[..., 3] == 255).all()


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


# ---------------------------------------------------------------------------------------------------------------- raster rules
# The oracle's coverage, facing, depth test and depth bias against tests/independent_raster.py (exact integers / fractions, written from
# the rules).  Vertices are given on the 1/256-pixel grid through identity matrices (w = 1), so the oracle's transform and snap are exact
# and both sides see the same integers: what is compared is the RULE, not the arithmetic that feeds it.

def _raster_case(oracle_lib, tris_px, zs, W=32, H=32):
    """tris_px: per triangle three (x, y) in pixels, multiples of 1/256; zs: three depths each.  -> (oracle, sub-pixel triangles)"""
    import independent_raster as ir
    o = oracle_lib.Oracle(W, H, W)                      # shadow map of the same size: the shadow pass sees the same triangles
    verts = np.zeros(3 * len(tris_px), dtype=abi.XkVertex)
    k = 0
    for tri, z in zip(tris_px, zs):
        for (x, y), zz in zip(tri, z):
            verts[k]["Position"] = (x / (W / 2.0) - 1.0, y / (H / 2.0) - 1.0, zz)     # ndc = clip (w = 1): exact in float32
            verts[k]["Normal"] = (0.0, 0.0, 1.0); verts[k]["Color"] = (1.0, 1.0, 1.0)
            k += 1
    o.object_add(o.mesh_create(verts, np.arange(len(verts), dtype=np.uint32)))
    d, p, sp = _lights(1, 1)
    o.update_uniforms(abi.make_camera(), d, p, sp, 0.0, 0.0, 0.0)
    cam, sh, view = o.get_frame()
    ident = np.eye(4, dtype=np.float32).reshape(16)
    for m in (cam, sh):
        m["Model"] = ident; m["View"] = ident; m["Proj"] = ident
    o.set_frame(cam, sh, view)
    o.render(0, 3)                                       # shadow + deferred-scene passes
    sub = [[(int(round(x * ir.SUB)), int(round(y * ir.SUB))) for x, y in tri] for tri in tris_px]
    for tri_px, tri in zip(tris_px, sub):
        assert all(abs(x * ir.SUB - X) < 1e-9 and abs(y * ir.SUB - Y) < 1e-9 for (x, y), (X, Y) in zip(tri_px, tri)), "vertices must sit on the grid"
    return o, sub


def _check_raster(oracle_lib, tris_px, zs, W=32, H=32, min_ambiguous_free=0.98):
    import independent_raster as ir
    o, sub = _raster_case(oracle_lib, tris_px, zs, W, H)
    zs32 = [[float(np.float32(v)) for v in z] for z in zs]
    # ---- deferred-scene pass: who owns each pixel, and at what depth
    win, dep, amb = ir.render(sub, zs32, W, H, shadow=False)
    vis, depth = o.visibility(), o.gbuffer(0)
    checked = 0
    for y in range(H):
        for x in range(W):
            if amb[y][x]:
                continue
            checked += 1
            want = 0xFFFFFFFF if win[y][x] is None else win[y][x]
            assert int(vis[y, x]) == want, "pixel (%d, %d): oracle %d, rules %s" % (x, y, int(vis[y, x]), win[y][x])
            if win[y][x] is not None:
                assert abs(float(depth[y, x]) - float(dep[y][x])) <= 3e-7, (x, y, float(depth[y, x]), float(dep[y][x]))
    assert checked >= min_ambiguous_free * W * H
    # ---- shadow pass: cull NONE, LESS_OR_EQUAL, biased depth
    _, sdep, _ = ir.render(sub, zs32, W, H, shadow=True)
    sm = o.shadowmap()
    for y in range(H):
        for x in range(W):
            want = 1.0 if sdep[y][x] is None else float(sdep[y][x])
            assert abs(float(sm[y, x]) - want) <= 6e-7, "shadow texel (%d, %d): oracle %.9g, rules %.9g" % (x, y, float(sm[y, x]), want)
    return vis


def test_top_left_rule_on_shared_and_centre_crossing_edges(oracle_lib):
    """Edges that run exactly through pixel centres: a quad cut along a diagonal through the centres, its outer edges ON the centres of
    rows / columns (top and left ones count, bottom and right ones do not), a fan around a vertex that sits on a centre (exactly one
    triangle may own that pixel), both windings (the back-facing twin is culled in the deferred-scene pass, drawn in the shadow pass)."""
    c = 0.5
    quad = [[(4 + c, 4 + c), (4 + c, 12 + c), (12 + c, 12 + c)], [(4 + c, 4 + c), (12 + c, 12 + c), (12 + c, 4 + c)]]
    fan_centre = (22 + c, 9 + c)
    ring = [(26.0, 9.5), (25.0, 13.0), (22.5, 14.0), (19.0, 12.0), (18.0, 9.5), (19.5, 5.0), (22.5, 4.0), (25.5, 5.5)]
    fan = [[fan_centre, ring[(i + 1) % 8], ring[i]] for i in range(8)]
    back = [[(6.0, 20.0), (14.0, 20.5), (8.0, 28.0)]]              # the other winding: culled (BACK) - but a shadow caster
    front = [[(18.0, 20.0), (22.0, 28.0), (28.5, 20.5)]]
    tris = quad + fan + back + front
    zs = [[0.5, 0.5, 0.5]] * 2 + [[0.25, 0.5, 0.5]] * 8 + [[0.3, 0.4, 0.6]] + [[0.7, 0.2, 0.4]]
    vis = _check_raster(oracle_lib, tris, zs)
    import independent_raster as ir
    assert ir.front_facing([(int(x * 256), int(y * 256)) for x, y in quad[0]]) and not ir.front_facing([(int(x * 256), int(y * 256)) for x, y in back[0]])
    # the quad: its top row and left column of centres are in, the bottom row and right column are out; 8 x 8 pixels, each owned once
    assert (vis[4:12, 4:12] <= 1).all() and (vis[12, 4:13] == 0xFFFFFFFF).all() and (vis[4:13, 12] == 0xFFFFFFFF).all()
    assert ((vis[4:12, 4:12] == 0).sum(), (vis[4:12, 4:12] == 1).sum()) in ((36, 28), (28, 36))
    assert vis[9, 22] in range(2, 10)                              # the fan's hub pixel has exactly one owner


def _on_grid(tris):
    return [[(round(x * 256) / 256.0, round(y * 256) / 256.0) for x, y in t] for t in tris]


def test_slivers_degenerates_and_subpixel_triangles(oracle_lib):
    tris = _on_grid([[(3.0, 3.25), (29.0, 3.5), (3.0, 3.75)],               # a sliver thinner than a pixel, crossing a row of centres
            [(3.0, 8.0), (29.0, 8.6), (3.0, 8.2)],                 # ... and one that threads between two rows
            [(5.0, 12.0), (9.0, 16.0), (13.0, 20.0)],              # zero area (collinear)
            [(20.0, 12.0), (20.0, 12.0), (24.0, 16.0)],            # zero area (repeated vertex)
            [(10.6, 24.6), (10.9, 25.4), (11.4, 24.7)],            # lies between the centres: no fragment
            [(20.5, 24.5), (20.5, 25.5), (21.5, 25.5)],            # vertices ON three centres
            [(26.0, 22.0), (26.25, 30.0), (26.75, 22.0)]])         # a vertical sliver
    zs = [[0.5, 0.6, 0.5], [0.1, 0.2, 0.3], [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], [0.4, 0.4, 0.4], [0.3, 0.35, 0.4], [0.9, 0.95, 0.9]]
    _check_raster(oracle_lib, tris, zs)


def test_depth_order_ties_and_clip_against_exact_arithmetic(oracle_lib):
    """Overlapping triangles: nearer wins whatever the draw order, the first drawn keeps a pixel on equal depth (LESS), fragments beyond
    z = 1 are clipped per fragment (the plane leaves the depth range inside the triangle), z = 1 never beats the clear."""
    big = lambda z: [[(2.0, 2.0), (2.0, 30.0), (30.0, 30.0)], z]
    cases = [big([0.6, 0.6, 0.6]), big([0.4, 0.4, 0.4]), big([0.4, 0.4, 0.4]), big([0.5, 0.2, 0.8]),
             [[(4.0, 2.0), (30.0, 28.0), (30.0, 2.0)], [0.25, 0.5, 1.25]],           # leaves the depth range inside the triangle (far side:
                                                                                      # clipped per fragment; the near side goes through
                                                                                      # the oracle's polygon clipper, whose re-snapped
                                                                                      # intersections are its own stated choice)
             [[(16.0, 4.0), (20.0, 12.0), (24.0, 4.0)], [1.0, 1.0, 1.0]]]            # at the clear depth: never visible
    _check_raster(oracle_lib, [t for t, _ in cases], [z for _, z in cases], min_ambiguous_free=0.9)


def test_random_triangles_on_the_subpixel_grid(oracle_lib):
    rng = np.random.default_rng(17)
    tris, zs = [], []
    for _ in range(160):
        cx, cy = rng.uniform(2, 30, 2)
        pts = [(float(np.round((cx + rng.uniform(-6, 6)) * 256) / 256), float(np.round((cy + rng.uniform(-6, 6)) * 256) / 256)) for _ in range(3)]
        tris.append(pts)
        zs.append([float(np.float32(v)) for v in rng.uniform(0.05, 0.95, 3)])
    _check_raster(oracle_lib, tris, zs, min_ambiguous_free=0.97)


# ---------------------------------------------------------------- the samplers against a float64 restatement of the Vulkan rules
import independent_sampler as isamp


def _noise_image(w, h, seed, smooth=False):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    if smooth:                                               # large flat-ish areas as well as noise
        yy, xx = np.mgrid[0:h, 0:w]
        img[..., 0] = (xx * 255 // max(1, w - 1)).astype(np.uint8)
        img[..., 1] = (yy * 255 // max(1, h - 1)).astype(np.uint8)
    return img


@pytest.mark.parametrize("w,h,srgb", [(64, 64, 0), (64, 64, 1), (48, 20, 1), (33, 7, 0), (1, 16, 1), (5, 5, 0)])
def test_mip_chains_are_linear_blits_in_linear_light(oracle_lib, w, h, srgb):
    """RHIGenerateMipmaps (ZE:6348-6433): floor(log2(max(w, h))) + 1 levels (ZE:6887), each the LINEAR blit of the one before
    (sRGB formats are filtered in linear light).  Every 8-bit code of the oracle's chain is the rounding of the float64 value."""
    o = oracle_lib.Oracle(8, 8, 8)
    img = _noise_image(w, h, seed=w * 131 + h + srgb, smooth=True)
    chain = o.tex_mips(img, srgb)
    assert len(chain) == int(math.floor(math.log2(max(w, h)))) + 1
    assert np.array_equal(chain[0], img)
    for l in range(1, len(chain)):
        assert chain[l].shape[:2] == (max(1, h >> l), max(1, w >> l))
        real = isamp.blit_half_real(chain[l - 1], srgb)
        err = np.abs(chain[l].astype(np.float64) - real)
        assert err.max() <= 0.5 + 2e-3, (l, err.max())       # the rounding of the real value (float32 noise only at exact ties)
    o.close()


@pytest.mark.parametrize("srgb", [0, 1])
def test_material_sampler_against_float64(oracle_lib, srgb):
    """texture(sampler2D, uv) with derivatives: footprint -> tap count / LOD / major axis, REPEAT addressing, trilinear taps averaged.
    600 random samples over magnified, minified, 2:1 .. beyond-16:1 footprints and coordinates outside [0, 1]."""
    o = oracle_lib.Oracle(8, 8, 8)
    rng = np.random.default_rng(77 + srgb)
    img = _noise_image(64, 32, seed=5 + srgb)
    chain = o.tex_mips(img, srgb)
    checked, taps_seen = 0, set()
    for k in range(600):
        uv = rng.uniform(-2.0, 3.0, 2)
        scale = 2.0 ** rng.uniform(-9.0, -1.0)                 # footprint from 1/8 texel to half the image
        ang = rng.uniform(0, 2 * math.pi)
        ratio = rng.choice([1.0, 1.37, 2.4, 5.5, 11.3, 15.6, 40.0])
        major = np.array([math.cos(ang), math.sin(ang)]) * scale
        minor = np.array([-math.sin(ang), math.cos(ang)]) * scale / ratio
        duv = (major[0], major[1], minor[0], minor[1]) if k % 2 else (minor[0], minor[1], major[0], major[1])
        if k % 10 == 0:                                        # an exactly isotropic footprint of t texels: one trilinear tap
            t = np.float32(2.0 ** rng.uniform(-3.0, 5.0))
            duv = (t / 64, 0.0, 0.0, t / 32)
        duv32 = np.asarray(duv, np.float32); uv32 = np.asarray(uv, np.float32)
        want, n, lam, margin = isamp.sample_2d(chain, srgb, uv32.astype(np.float64), duv32.astype(np.float64))
        if margin < 1e-3:
            continue                                           # Pmax / Pmin on an integer: N or N + 1 taps are both right
        got = o.tex_sample(img, srgb, uv32, duv32)
        assert np.allclose(got, want, atol=3e-5, rtol=0), (k, n, lam, got, want)
        checked += 1; taps_seen.add(n)
    assert checked > 550 and {1, 2, 3, 6, 12, 16} <= taps_seen
    # exact cases: the centre of a texel at magnification is that texel; one whole-image footprint is the last level's only texel
    px = o.tex_sample(img, srgb, ((10 + 0.5) / 64, (7 + 0.5) / 32), (0, 0, 0, 0))
    assert np.allclose(px, isamp.decode(img, srgb)[7, 10], atol=1e-6)
    far = o.tex_sample(img, srgb, (0.3, 0.8), (4.0, 0, 0, 4.0))
    assert np.allclose(far, isamp.decode(chain[-1], srgb)[0, 0], atol=1e-6)
    o.close()


def test_cubemap_chain_and_textureLod_against_float64(oracle_lib):
    """textureLod(samplerCube, R, lod): face selection table (ties: z over y over x), bilinear inside the face, linear between
    levels; the chain is a 2x2 box in linear light.  Faces carry noise, so a wrong face, a flipped axis or a wrong level shows."""
    dim = 32
    rng = np.random.default_rng(3)
    faces = [rng.integers(0, 256, (dim, dim, 4), dtype=np.uint8) for _ in range(6)]
    o = oracle_lib.Oracle(8, 8, 8)
    o.set_cubemap(faces)
    chain = o.cube_mips(dim)
    assert len(chain) == 6 and np.array_equal(chain[0], np.stack(faces))
    for l in range(1, len(chain)):
        rgb, alpha = isamp.cube_half_real(chain[l - 1])
        assert np.abs(chain[l][..., :3].astype(np.float64) - rgb).max() <= 0.5 + 2e-3
        assert np.abs(chain[l][..., 3].astype(np.float64) - alpha).max() <= 0.5
    dirs = list(rng.normal(size=(400, 3)))
    dirs += [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1),          # face centres
             (1, 1, 0.3), (1, 0.2, 1), (0.1, 1, 1), (1, 1, 1), (-1, 1, -1), (2, -2, 0.5)]  # ties between major axes
    for k, r in enumerate(dirs):
        r32 = np.asarray(r, np.float32)
        for lod in (0.0, 0.37, 1.0, 2.5, 4.99, 5.0, 7.0, -1.0):
            got = o.cube_sample(r32, lod)
            want = isamp.sample_cube(chain, r32.astype(np.float64), np.float32(lod))
            assert np.allclose(got, want, atol=3e-5, rtol=0), (k, r, lod, got, want)
    o.close()


# ---------------------------------------------------------------------------------------------------------------- the clipper
def test_near_plane_and_guard_band_clipping_against_exact_homogeneous_coverage(oracle_lib):
    """Triangles that cross the eye plane, the near plane and the guard band, through a perspective whose arithmetic is exact in float32
    (dyadic entries: clip = (x, y, -1.5 z - 1.5, -z), near 1, far 3).  What the oracle's clipper + re-projection + snap draws is compared
    with the exact coverage of the UNCLIPPED triangle in homogeneous coordinates (tests/independent_raster.covers_clip_space) cut at
    0 <= depth <= 1, on every pixel whose centre is not within 6/256 px of an edge or of a clip boundary (the clipper's new vertices are
    snapped like any other, which moves its edges by up to a sub-pixel step): owner, and depth to within what it varies over those 6/256 px.  Both windings: one is drawn, the
    other culled."""
    import independent_raster as ir
    from fractions import Fraction as Fr
    W = H = 48
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 1.0; P[1, 1] = 1.0; P[2, 2] = -1.5; P[2, 3] = -1.5; P[3, 2] = -1.0      # row-major here; stored column-major below
    cases = [   # object-space triangles (x, y, z), dyadic; z < 0 is in front of the eye
        [(-0.5, -0.75, -2.5), (0.75, -0.5, -1.25), (0.25, 2.0, 0.5)],       # one vertex behind the eye
        [(-2.0, -0.25, 0.25), (1.5, -0.5, 0.75), (0.0, 0.75, -2.0)],        # two vertices behind the eye
        [(-0.75, -0.5, -0.5), (0.5, -0.75, -0.75), (0.25, 0.5, -1.5)],      # between eye and near plane: crosses z = 0 only
        [(-40.0, -1.0, -1.5), (30.0, -0.75, -1.75), (0.5, 35.0, -2.0)],     # far outside the guard band on three sides
        [(-0.5, -0.5, -2.5), (0.5, -0.5, -3.5), (0.0, 0.75, -4.0)],         # crosses the far plane: per-fragment depth clip
    ]
    delta = Fr(6, 256)
    for ci, tri in enumerate(cases):
        for flip in (False, True):
            t3 = [tri[0], tri[2], tri[1]] if flip else tri
            o = oracle_lib.Oracle(W, H, 8)
            verts = np.zeros(3, dtype=abi.XkVertex)
            for k, p in enumerate(t3):
                verts[k]["Position"] = p; verts[k]["Normal"] = (0.0, 0.0, 1.0); verts[k]["Color"] = (1.0, 1.0, 1.0)
            o.object_add(o.mesh_create(verts, np.arange(3, dtype=np.uint32)))
            d, pl, sp = _lights(1, 1)
            o.update_uniforms(abi.make_camera(), d, pl, sp, 0.0, 0.0, 0.0)
            cam, sh, view = o.get_frame()
            ident = np.eye(4, dtype=np.float32).reshape(16)
            for m in (cam, sh):
                m["Model"] = ident; m["View"] = ident
            cam["Proj"] = P.T.reshape(16)                   # column-major storage
            o.set_frame(cam, sh, view)
            o.render(0, 2)
            vis, depth = o.visibility(), o.gbuffer(0)
            clip = [(Fr(x), Fr(y), Fr(-3, 2) * Fr(z) - Fr(3, 2), -Fr(z)) for x, y, z in t3]
            # facing of the visible part: the rule of independent_raster.front_facing (front = negative signed area, y down) on det's sign
            front = ir.orientation_clip_space(clip) < 0
            decisive = drawn = 0
            for py in range(H):
                for px in range(W):
                    probes = []
                    for dx, dy in ((-delta, -delta), (delta, -delta), (-delta, delta), (delta, delta), (0, 0)):
                        X = (Fr(2 * px + 1, 2) + dx) / Fr(W, 2) - 1
                        Y = (Fr(2 * py + 1, 2) + dy) / Fr(H, 2) - 1
                        c = ir.covers_clip_space(clip, X, Y)
                        probes.append(None if c is None or not (0 <= c[0] <= 1) else c[0])
                    inside = [q is not None for q in probes]
                    if any(inside) != all(inside):
                        continue                                              # near an edge, the near plane or the far plane
                    if all(inside) and any(abs(q) < Fr(1, 2000) or abs(1 - q) < Fr(1, 2000) for q in probes):
                        continue
                    decisive += 1
                    want = all(inside) and front
                    got = int(vis[py, px]) != 0xFFFFFFFF
                    assert got == want, "case %d%s pixel (%d, %d): oracle %s, exact %s" % (ci, " flipped" if flip else "", px, py, got, want)
                    if want:
                        drawn += 1
                        # (the clipper's vertices are snapped to 1/256 px: near the near plane the depth changes fast across a pixel, so the
                        # tolerance is what the depth does over the probe square, 12/256 px wide)
                        tol = 2e-4 + float(max(probes) - min(probes))
                        assert abs(float(depth[py, px]) - float(probes[4])) < tol, (ci, px, py, float(depth[py, px]), float(probes[4]), tol)
            assert decisive > W * H * 0.8
            if front:
                assert drawn > 5, (ci, flip, drawn)
            o.close()
