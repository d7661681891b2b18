"""GPU parity over RANDOM combinations of the path's features (every target bit for bit against the CPU oracle, as everywhere).

The named tests exercise features one or two at a time; here a seeded generator draws small scenes that mix them: instanced and
non-instanced draws of several meshes, constant / packed / mixed-size sampled materials, skydome and background on or off, 0-2 directional
and 0-40 point lights, cameras inside and outside the dome looking along and across the ground (near-plane clipping, guard band,
anisotropic footprints), the sun anywhere (grazing shadow projections), odd target and shadow-map sizes (partial tiles, shadow windows
cut by the map's edge), debug views, culls switched off, one stream or two lanes, and two frames in a row with a moved camera
(visibility history, stale Hi-Z); now and then 70 000 instances (the instance pre-pass), a model matrix that is sheared or mirrored, a
vertex that is NaN or infinite, instances of scale 0 / negative / denormal.
"""
import math
import os

import numpy as np
import pytest

from parity_util import compare_all
from zeldaengine_amd import abi, scenes

pytestmark = pytest.mark.gpu


def _image(rng, w, h, kind):
    if kind == 0:
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.zeros((h, w, 4), np.uint8)
        img[..., 0] = (xx * 255 // max(1, w - 1)); img[..., 1] = (yy * 255 // max(1, h - 1))
        img[..., 2] = (((xx // 3) + (yy // 2)) & 1) * 200 + 30
    img[..., 3] = 255
    return img


def _material(rng):
    """None (engine defaults), constants as images, one image size for all image slots (the packed form), or mixed sizes"""
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return None
    sizes = [(int(rng.choice([4, 16, 24, 64])), int(rng.choice([4, 8, 32, 48])))]
    if kind == 3:
        sizes.append((int(rng.choice([5, 32])), int(rng.choice([3, 16]))))
    imgs = []
    for slot in range(7):
        r = rng.random()
        if r < 0.3:
            imgs.append(None)
        elif r < 0.45 or kind == 1:
            imgs.append(np.tile(rng.integers(0, 256, 4, dtype=np.uint8), (4, 4, 1)))       # a constant image
        else:
            w, h = sizes[int(rng.integers(0, len(sizes)))]
            img = _image(rng, w, h, int(rng.integers(0, 2)))
            if slot == 6:
                img[..., 0] = 255                                                        # mask r = 1: lit
            if slot == 3:
                img[..., :2] = (img[..., :2] // 4 + 96); img[..., 2] = 255               # a plausible normal map
            imgs.append(img)
    return imgs


def _scene(seed):
    rng = np.random.default_rng(seed)
    W, H = int(rng.choice([160, 200, 257, 320])), int(rng.choice([96, 120, 131, 200]))
    SD = int(rng.choice([64, 100, 128, 256]))
    meshes = [scenes.uv_sphere(), scenes.uv_sphere(12, 6, 0.5), scenes.box((0.6, 0.4, 0.5), (0.0, 0.0, 0.5)), scenes.grid_plane(float(rng.choice([12.0, 30.0, 70.0])), int(rng.integers(2, 7)), 0.0)]
    draws = []
    n_draws = int(rng.integers(2, 6))
    for k in range(n_draws):
        mi = 3 if k == 0 else int(rng.integers(0, 3))
        mat = _material(rng)
        inst = None
        if mi != 3 and rng.random() < 0.7:
            inst = scenes.generate_instances(int(rng.integers(1, 120)), float(rng.uniform(0.2, 2.0)), float(rng.uniform(3.0, 30.0)), 0.2, float(rng.uniform(0.4, 2.0)), seed=int(rng.integers(1, 1 << 30)))
        uvscale = float(rng.choice([1.0, 3.0, 9.0]))
        draws.append((mi, mat, inst, uvscale))
    sky = rng.random() < 0.5
    bg = rng.random() < 0.3
    w = scenes.sample_world()
    if rng.random() < 0.6:                               # the sun somewhere else: another shadow projection, grazing ones included
        a, el = rng.uniform(0, 2 * math.pi), float(rng.choice([0.15, 0.6, 1.2, 1.5]))
        dist = float(rng.choice([8.0, 28.0, 60.0]))
        pos = [dist * math.cos(a) * math.cos(el), dist * math.sin(a) * math.cos(el), dist * math.sin(el)]
        w["DirectionalLights"][0]["Position"] = pos; w["DirectionalLights"][0]["Direction"] = pos
    d, _, s = scenes.lights_from_world(w)
    w["PointLights"] = scenes.sample_point_lights(int(rng.choice([0, 1, 3, 8, 16, 40])))
    _, p, _ = scenes.lights_from_world(w)
    if rng.random() < 0.2:
        d = d[:0]
    ang = rng.uniform(0, 2 * math.pi)
    rad = float(rng.choice([3.0, 8.0, 25.0]))
    eye = (rad * math.cos(ang), rad * math.sin(ang), float(rng.choice([0.3, 1.5, 6.0])))
    look = (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), float(rng.uniform(0.0, 1.0)))
    cam = dict(position=eye, lookat=look, fov=float(rng.choice([35.0, 45.0, 70.0])), znear=float(rng.choice([0.05, 0.1, 0.5])), zfar=float(rng.choice([20.0, 45.0, 200.0])))
    flags = int(rng.choice([0, 0, 0, abi.FLAG_NO_HIZ, abi.FLAG_NO_FRUSTUM_CULL | abi.FLAG_NO_CONE_CULL, abi.FLAG_SERIAL_PASSES]))
    if seed % 2:          # every other scene: the shadow pass's occlusion culling whatever the scene's size (frame 2 runs on frame 1's flags)
        flags |= abi.FLAG_SHADOW_OCCLUSION
    view = int(rng.choice([0, 0, 0, 0, 1, 2, 3, 4, 5, 7, 8, 9]))
    # rarer ingredients
    extra = {}
    r = rng.random()
    if r < 0.08:          # the instance-level pre-pass and its work lists: >= 65 536 instances of a 12-triangle box
        extra["many"] = scenes.generate_instances(66000 + int(rng.integers(0, 9000)), 0.5, float(rng.choice([6.0, 14.0])), 0.02, 0.12, seed=int(rng.integers(1, 1 << 30)))
    elif r < 0.16:        # a model matrix that is not rigid (cone and sphere tests must switch themselves off), or mirrored
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = np.diag(rng.choice([0.5, 1.0, 2.5, -1.0], 3)).astype(np.float32)
        m[0, 1] = np.float32(rng.choice([0.0, 0.4]))
        extra["model"] = m
    elif r < 0.22:        # a vertex that is not a number / not finite: its triangles are dropped, nothing else is
        extra["bad_vertex"] = (int(rng.integers(0, 3)), float(rng.choice([np.nan, np.inf, -np.inf, 3.0e38])))
    elif r < 0.28:        # instances with zero and negative scale among the others
        extra["odd_scale"] = True
    elif r < 0.36:        # a projection that is not the engine's: off-centre, un-flipped, stretched (both passes' own matrices are touched:
        #                   the sphere / cone / owned-region tests must notice that their assumptions do not hold)
        extra["proj"] = (float(rng.choice([0.0, 0.15, -0.4])), float(rng.choice([0.0, -0.2, 0.3])), float(rng.choice([1.0, -1.0])), float(rng.choice([1.0, 0.6, 1.7])))
    return dict(W=W, H=H, SD=SD, meshes=meshes, draws=draws, sky=sky, bg=bg, lights=(d, p, s), cam=cam, flags=flags, view=view,
                roll=(float(rng.uniform(0, 1)), float(rng.uniform(0, 1))), extra=extra)


def _build(r, sc):
    r.set_cubemap(scenes.synthetic_cubemap(16))
    keep = []
    ids = {}
    extra = sc.get("extra", {})
    for di, (mi, mat, inst, uvscale) in enumerate(sc["draws"]):
        key = (mi, uvscale)
        if key not in ids:
            v, idx = sc["meshes"][mi]
            v = v.copy(); v["TexCoord"] *= uvscale
            if "bad_vertex" in extra and mi == extra["bad_vertex"][0]:
                v["Position"][len(v) // 2][1] = extra["bad_vertex"][1]
            ids[key] = r.mesh_create(v, idx)
        m = None
        if mat is not None:
            m, k = abi.make_material(mat); keep.append(k)
        if inst is not None and extra.get("odd_scale") and len(inst) > 3:
            inst = inst.copy(); inst["InstancePScale"][0] = 0.0; inst["InstancePScale"][1] = -0.7; inst["InstancePScale"][2] = 1e-30
        r.object_add(ids[key], m, inst)
    if "many" in extra:
        r.object_add(r.mesh_create(*scenes.box((0.5, 0.5, 0.5), (0.0, 0.0, 0.5))), None, extra["many"])
    if sc["sky"]:
        r.set_skydome(*scenes.sky_dome(), scenes.synthetic_sky_image(64, 32))
    if sc["bg"]:
        r.set_background(scenes.synthetic_sky_image(48, 40)[:, ::-1].copy())
    r._keepalive = keep


def _raw_frame(r, sc):
    """zr_set_frame with the engine's matrices altered: another model matrix and / or another projection, both passes"""
    extra = sc.get("extra", {})
    if "model" not in extra and "proj" not in extra:
        return
    cm, sh, view = r.get_frame()
    for u in (cm, sh):
        if "model" in extra:
            u["Model"] = (np.asarray(u["Model"], np.float32).reshape(4, 4).T @ extra["model"]).T.reshape(16)
        if "proj" in extra:
            ox, oy, fy, sx = extra["proj"]
            P = np.asarray(u["Proj"], np.float32).reshape(4, 4).T.copy()
            P[0, 2] += np.float32(ox); P[1, 2] += np.float32(oy)          # off-centre frustum
            P[1, :] *= np.float32(fy)                                     # un-flipped y: handedness reverses
            P[0, 0] *= np.float32(sx)                                     # stretched
            u["Proj"] = P.T.reshape(16)
    r.set_frame(cm, sh, view)


N_SEEDS = int(os.environ.get("ZR_FUZZ_SEEDS", "48"))      # (a longer hunt: ZR_FUZZ_SEEDS=2000 [ZR_FUZZ_BASE=5000] pytest tests/test_gpu_fuzz.py -m gpu)
BASE = int(os.environ.get("ZR_FUZZ_BASE", "1000"))


@pytest.mark.parametrize("seed", list(range(N_SEEDS)))
def test_random_scene_matches_the_oracle(oracle_lib, gpu_engine, seed):
    sc = _scene(BASE + seed)
    o = oracle_lib.Oracle(sc["W"], sc["H"], sc["SD"])
    g = gpu_engine.Renderer(sc["W"], sc["H"], sc["SD"], flags=sc["flags"])
    for r in (o, g):
        _build(r, sc)
    d, p, s = sc["lights"]
    cam = dict(sc["cam"])
    for frame in range(2):
        if frame == 1:                                   # the camera moves: the second frame runs on the first one's visibility history
            e = cam["position"]
            cam["position"] = (e[0] * 0.9 - 0.3 * e[1], e[1] * 0.9 + 0.3 * e[0], e[2] + 0.2)
        for r in (o, g):
            r.update_uniforms(abi.make_camera(**cam), d, p, s, sc["roll"][0], sc["roll"][1] + 0.05 * frame, 1.0 + frame)
            _raw_frame(r, sc)
        o.render(sc["view"])
        g.render(sc["view"]); g.finish()
        diff = {k: v for k, v in compare_all(o, g).items() if v}
        assert not diff, "seed %d frame %d (flags %d, view %d, %dx%d, shadow %d): %r" % (seed, frame, sc["flags"], sc["view"], sc["W"], sc["H"], sc["SD"], diff)
    st = g.stats()
    assert st["overflow"] == 0
    g.close()
    o.close()


@pytest.mark.parametrize("seed", list(range(max(8, N_SEEDS // 4))))
def test_random_scene_over_a_tile_partition(oracle_lib, gpu_engine, seed):
    """The same random scenes as N rank contexts (2..8 ranks: super-tile ownership, owned-region reject, per-rank visibility history): every
    rank's packed tiles are the oracle's pixels of the tiles it owns, on two frames."""
    from zeldaengine_amd import dist as zdist
    sc = _scene(BASE + 100000 + seed)
    if sc["view"] == 9:
        sc["view"] = 0
    world = int(np.random.default_rng(seed).choice([2, 3, 4, 8]))
    o = oracle_lib.Oracle(sc["W"], sc["H"], sc["SD"])
    ranks = [gpu_engine.Renderer(sc["W"], sc["H"], sc["SD"], flags=sc["flags"], tile_rank=r, tile_world=world) for r in range(world)]
    for r in [o] + ranks:
        _build(r, sc)
    d, p, s = sc["lights"]
    cam = dict(sc["cam"])
    for frame in range(2):
        if frame == 1:
            e = cam["position"]
            cam["position"] = (e[0] * 0.9 - 0.3 * e[1], e[1] * 0.9 + 0.3 * e[0], e[2] + 0.2)
        for r in [o] + ranks:
            r.update_uniforms(abi.make_camera(**cam), d, p, s, sc["roll"][0], sc["roll"][1] + 0.05 * frame, 1.0 + frame)
            _raw_frame(r, sc)
        o.render(sc["view"])
        want = o.color()
        for rk, g in enumerate(ranks):
            g.render(sc["view"])
            assert np.array_equal(g.read_tiles(), zdist.pack_tiles(want, rk, world)), \
                "seed %d frame %d rank %d of %d (flags %d, view %d, %dx%d)" % (seed, frame, rk, world, sc["flags"], sc["view"], sc["W"], sc["H"])
            assert g.stats()["overflow"] == 0
    for g in ranks:
        g.close()
    o.close()


@pytest.mark.parametrize("seed,size", [(3, (1600, 900)), (11, (1920, 1080)), (17, (2048, 857))])
def test_random_scene_at_full_size(oracle_lib, gpu_engine, seed, size):
    """A few of the random scenes at target sizes of thousands of tiles (the small ones above are 15 to 70): the oracle's rasteriser and
    per-pixel stages on all cores."""
    sc = _scene(BASE + 500000 + seed)
    sc["W"], sc["H"], sc["SD"] = size[0], size[1], 1024
    o = oracle_lib.Oracle(sc["W"], sc["H"], sc["SD"])
    o.set_threads(16)
    g = gpu_engine.Renderer(sc["W"], sc["H"], sc["SD"], flags=sc["flags"])
    for r in (o, g):
        _build(r, sc)
    d, p, s = sc["lights"]
    cam = dict(sc["cam"])
    for frame in range(2):
        if frame == 1:
            e = cam["position"]
            cam["position"] = (e[0] * 0.9 - 0.3 * e[1], e[1] * 0.9 + 0.3 * e[0], e[2] + 0.2)
        for r in (o, g):
            r.update_uniforms(abi.make_camera(**cam), d, p, s, sc["roll"][0], sc["roll"][1] + 0.05 * frame, 1.0 + frame)
        o.render(sc["view"])
        g.render(sc["view"]); g.finish()
        diff = {k: v for k, v in compare_all(o, g).items() if v}
        assert not diff, "seed %d frame %d at %dx%d: %r" % (seed, frame, sc["W"], sc["H"], diff)
    g.close()
    o.close()


@pytest.mark.parametrize("abs_seed,size,sd", [(900030, (1000, 1000), 2048), (900126, (1000, 1000), 2048), (900156, (1280, 720), 2048)])
def test_big_casters_under_a_large_shadow_map(oracle_lib, gpu_engine, abs_seed, size, sd):
    """Scenes a hunt with the generator above at large sizes found: a ground plane that crosses the light's near plane is listed for every
    tile of the shadow map (its box cannot be projected) and its triangles - 64 texels and more across - go the clipper's way; under a
    2048^2 map that is 4 096 lists, each of which used to hand every such triangle on, whether it reached the tile or not, until
    the list of handed-on triangles (2^18) overflowed: ZR_ERR_OVERFLOW for scenes that are nothing special.  Now a list hands a big
    triangle on only if its snapped box reaches the list's window, and the list grows with the map."""
    sc = _scene(abs_seed)
    sc["W"], sc["H"], sc["SD"] = size[0], size[1], sd
    o = oracle_lib.Oracle(sc["W"], sc["H"], sc["SD"]); o.set_threads(16)
    g = gpu_engine.Renderer(sc["W"], sc["H"], sc["SD"], flags=sc["flags"])
    for r in (o, g):
        _build(r, sc)
    d, p, s = sc["lights"]
    for r in (o, g):
        r.update_uniforms(abi.make_camera(**sc["cam"]), d, p, s, sc["roll"][0], sc["roll"][1], 1.0)
        _raw_frame(r, sc)
    o.render(sc["view"])
    g.render(sc["view"]); g.finish()                       # (ZR_ERR_OVERFLOW would raise here)
    diff = {k: v for k, v in compare_all(o, g).items() if v}
    assert not diff and g.stats()["overflow"] == 0, (abs_seed, diff)
    g.close(); o.close()


def _random_meshlets(rng, oracle_lib, v, idx, scattered):
    """A valid meshlet partition of (v, idx) that no clusteriser would make: triangles in random order (scattered: every meshlet a
    random handful from all over the mesh - wide cones, huge spheres) or in runs of random length, cut at 64 vertices / 124 triangles;
    bounds and cones from the ORACLE's meshopt_computeMeshletBounds restatement (zo_meshlet_bounds), not from the library's."""
    import ctypes as C
    tris = idx.reshape(-1, 3)
    order = rng.permutation(len(tris)) if scattered else np.arange(len(tris))
    ml, mv, mt, flat_order = [], [], [], []
    i = 0
    while i < len(order):
        want = int(rng.integers(1, 125))
        local, lt = {}, []
        while i < len(order) and len(lt) < want:
            t = tris[order[i]]
            new = [x for x in dict.fromkeys(int(a) for a in t) if x not in local]
            if len(local) + len(new) > 64:
                break
            for x in new:
                local[x] = len(local)
            lt.append([local[int(a)] for a in t]); flat_order.append(int(order[i])); i += 1
        rec = np.zeros((), dtype=abi.XkMeshlet)
        rec["VertexOffset"] = len(mv); rec["VertexCount"] = len(local); rec["TriangleOffset"] = len(mt); rec["TriangleCount"] = len(lt)
        verts = np.asarray(list(local.keys()), np.uint32); tb = np.asarray(lt, np.uint8).reshape(-1)
        out = np.zeros((), dtype=abi.XkMeshlet)
        assert oracle_lib.lib().zo_meshlet_bounds(v.ctypes.data_as(C.c_void_p), verts.ctypes.data_as(C.c_void_p), tb.ctypes.data_as(C.c_void_p), len(lt), out.ctypes.data_as(C.c_void_p)) == 0
        for f in ("BoundsCenter", "BoundsRadius", "ConeApex", "ConeAxis", "ConeCutoff"):
            rec[f] = out[f]
        ml.append(rec); mv += list(verts); mt += list(tb)
        while len(mt) % 4:
            mt.append(0)                                  # (the container pads a meshlet's triangle bytes to four)
    return np.asarray(ml, dtype=abi.XkMeshlet), np.asarray(mv, np.uint32), np.asarray(mt, np.uint8), np.asarray(flat_order)


@pytest.mark.parametrize("seed", list(range(max(6, N_SEEDS // 4))))
def test_random_meshlet_partitions_from_the_caller(oracle_lib, gpu_engine, seed):
    """zr_mesh_set_meshlets with partitions a clusteriser would never produce, bounds and cones computed by the oracle's independent
    restatement of meshopt's formulas: every cull (frustum, cone, box, Hi-Z, owned region) must stay invisible whatever the meshlets
    look like.  The oracle draws the flattened index buffer of the same partition (ZE:4733-4756)."""
    rng = np.random.default_rng(40000 + seed)
    sc = _scene(BASE + 300000 + seed)
    sc["extra"] = {}
    o = oracle_lib.Oracle(sc["W"], sc["H"], sc["SD"])
    g = gpu_engine.Renderer(sc["W"], sc["H"], sc["SD"], flags=sc["flags"] & ~3)        # the culls stay ON here
    for r in (o, g):
        r.set_cubemap(scenes.synthetic_cubemap(16))
    for di, (mi, mat, inst, uvscale) in enumerate(sc["draws"]):
        v, idx = sc["meshes"][mi]
        ml, mv, mt, order = _random_meshlets(rng, oracle_lib, v, idx, scattered=bool(rng.integers(0, 2)))
        flat = idx.reshape(-1, 3)[order].reshape(-1)
        o.object_add(o.mesh_create(v, flat), None, inst)
        m = g.mesh_create(v, idx)
        g.mesh_set_meshlets(m, ml, mv, mt)
        g.object_add(m, None, inst)
    d, p, s = sc["lights"]
    cam = dict(sc["cam"])
    for frame in range(2):
        if frame == 1:
            e = cam["position"]
            cam["position"] = (e[0] * 0.9 - 0.3 * e[1], e[1] * 0.9 + 0.3 * e[0], e[2] + 0.2)
        for r in (o, g):
            r.update_uniforms(abi.make_camera(**cam), d, p, s, sc["roll"][0], sc["roll"][1] + 0.05 * frame, 1.0 + frame)
        o.render(sc["view"])
        g.render(sc["view"]); g.finish()
        diff = {k: v_ for k, v_ in compare_all(o, g).items() if v_}
        assert not diff, "seed %d frame %d: %r" % (seed, frame, diff)
    g.close(); o.close()
