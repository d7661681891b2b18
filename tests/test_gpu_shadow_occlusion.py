"""GPU parity of the shadow pass's occlusion culling: the map must not depend on what was left out, nor on the flags of the last frame.

The rasteriser's first launch draws the meshlet-instances that were not hidden last frame; k_shadow_occlusion then tests every survivor of the
cull against the map (conservative box + least depth from k_cull_box), rewrites the flags and hands the unflagged ones that are not hidden
to a late launch.  Every frame of a sequence - no history, exact history, a light that moves a little / jumps, a scene that changes - is
compared bit for bit with the oracle (which draws every triangle) or, at sizes the oracle cannot reach, with a context that has the
culling off (ZR_FLAG_NO_SHADOW_OCCLUSION).  It is on by itself from one meshlet-instance per five texels of the map on (the 70 000-instance
scenes of test_gpu_hiz.py run it against the oracle that way); ZR_FLAG_SHADOW_OCCLUSION forces it for the small scenes here.
"""
import math

import numpy as np
import pytest

from parity_util import compare_all
from zeldaengine_amd import abi, scenes

pytestmark = pytest.mark.gpu
ON, OFF = abi.FLAG_SHADOW_OCCLUSION, abi.FLAG_NO_SHADOW_OCCLUSION


def _identical(o, g, what=""):
    d = compare_all(o, g)
    bad = {k: v for k, v in d.items() if v}
    assert not bad, "%s: HIP path differs from the oracle: %r" % (what, bad)


def _pile(r, n=900, seed=3):
    """Spheres many deep under the light, a big occluder above most of them, a ground plane that crosses the light's near plane."""
    r.set_cubemap(scenes.synthetic_cubemap(16))
    r.object_add(r.mesh_create(*scenes.grid_plane(60.0, 8, 0.0)))
    r.object_add(r.mesh_create(*scenes.box((4.0, 4.0, 0.1), (0.0, 0.0, 6.0))))
    r.object_add(r.mesh_create(*scenes.uv_sphere()), None, scenes.generate_instances(n, 0.5, 7.0, 0.2, 0.6, seed=seed))


def _lights():
    w = scenes.sample_world()
    d, _, s = scenes.lights_from_world(w)
    w["PointLights"] = scenes.sample_point_lights(4)
    _, p, _ = scenes.lights_from_world(w)
    return d, p, s


LIGHTS = [
    (6.0, 0.0, 14.0),      # no history: everything is drawn by the first launch
    (6.0, 0.0, 14.0),      # exact history
    (6.3, 0.4, 14.0),      # the light creeps: a few late ones
    (6.6, 0.8, 13.8),
    (-9.0, 7.0, 9.0),      # the light jumps: the flags say little
    (-9.0, 7.0, 9.0),
]


def test_sequence_against_the_oracle(oracle_lib, gpu_engine):
    W, H, SD = 320, 180, 256
    o = oracle_lib.Oracle(W, H, SD)
    g = gpu_engine.Renderer(W, H, SD, flags=ON)
    n = gpu_engine.Renderer(W, H, SD, flags=OFF)
    for r in (o, g, n):
        _pile(r)
    d, p, s = _lights()
    cam = abi.make_camera((9.0, -7.0, 6.0), (0.0, 0.0, 1.0), fov=50.0)
    hidden, late = [], []
    for i, lpos in enumerate(LIGHTS):
        d[0]["Position"][:3] = lpos; d[0]["Direction"][:3] = lpos
        for r in (o, g, n):
            r.update_uniforms(cam, d, p, s, 0.0, 0.01 * i, 1.0)
        o.render(0)
        g.render(); g.finish()
        n.render(); n.finish()
        assert np.array_equal(o.shadowmap().view(np.uint32), g.shadowmap().view(np.uint32)), "shadow map, frame %d" % i
        _identical(o, g, "frame %d" % i)
        sg, sn = g.stats(), n.stats()
        assert sg["overflow"] == 0 and sn["shadow_occluded"] == 0 and sn["shadow_late"] == 0
        # every survivor of the cull is drawn by the first launch, drawn late, or left out - and the cull itself does not change
        assert sg["survivors"][0] == sn["survivors"][0], (i, sg, sn)
        assert sg["covered_shadow_texels"] == sn["covered_shadow_texels"]
        hidden.append(sg["shadow_occluded"]); late.append(sg["shadow_late"])
    assert hidden[0] == 0 and late[0] == 0          # first frame of a scene: one launch draws it all
    assert hidden[1] > 100 and late[1] == 0, (hidden, late)      # still light: what was hidden is hidden, nothing is late
    assert late[2] > 0 or late[4] > 0, (hidden, late)            # a moved light uncovers some
    for r in (g, n):
        r.close()


def test_scene_change_resets_the_flags(oracle_lib, gpu_engine):
    """Work items are renumbered when the scene changes: the flags of the old scene must not be read for the new one."""
    W, H, SD = 256, 144, 128
    g = gpu_engine.Renderer(W, H, SD, flags=ON)
    d, p, s = _lights()
    d[0]["Position"][:3] = (6.0, 0.0, 14.0); d[0]["Direction"][:3] = (6.0, 0.0, 14.0)
    cam = abi.make_camera((9.0, -7.0, 6.0), (0.0, 0.0, 1.0), fov=50.0)
    for seed, count in ((3, 400), (8, 700), (3, 150)):
        g.scene_clear()
        _pile(g, count, seed)
        o = oracle_lib.Oracle(W, H, SD)
        _pile(o, count, seed)
        for r in (o, g):
            r.update_uniforms(cam, d, p, s, 0.0, 0.0, 1.0)
        o.render(0)
        for k in range(3):
            g.render(); g.finish()
            assert np.array_equal(o.shadowmap().view(np.uint32), g.shadowmap().view(np.uint32)), (seed, count, k)
            _identical(o, g, "scene %d/%d frame %d" % (seed, count, k))
            st = g.stats()
            assert (st["shadow_occluded"] > 0) == (k > 0) and st["shadow_late"] == 0, (seed, count, k, st)
    g.close()


@pytest.mark.parametrize("sd", [32, 100, 2048])
def test_map_sizes(oracle_lib, gpu_engine, sd):
    """Maps narrower than a test span's alignment cases (a row is read 4 texels at a time, clamped to the last 4), not a multiple of
    the tile, and larger than one window per meshlet."""
    W, H = 192, 108
    o = oracle_lib.Oracle(W, H, sd)
    g = gpu_engine.Renderer(W, H, sd, flags=ON)
    for r in (o, g):
        _pile(r, 300 if sd < 2048 else 120)
    d, p, s = _lights()
    cam = abi.make_camera((9.0, -7.0, 6.0), (0.0, 0.0, 1.0), fov=50.0)
    for i, lpos in enumerate(LIGHTS[:4]):
        d[0]["Position"][:3] = lpos; d[0]["Direction"][:3] = lpos
        for r in (o, g):
            r.update_uniforms(cam, d, p, s, 0.0, 0.0, 1.0)
        o.render(0)
        g.render(); g.finish()
        assert np.array_equal(o.shadowmap().view(np.uint32), g.shadowmap().view(np.uint32)), "shadow map %d, frame %d" % (sd, i)
        assert g.stats()["overflow"] == 0
    _identical(o, g, "map %d" % sd)
    g.close()


def test_full_size_config3_forced_with_an_orbiting_light(gpu_engine):
    """BASELINE config 3 (110 000 meshlet-instances under a 1024^2 map - below the density at which the culling switches itself on): forced
    on against off, a still light and one that orbits 1.7 degrees per frame: the same map and the same frame, bit for bit."""
    from zeldaengine_amd import engine as eng
    cfg = scenes.config3(10000, cube_dim=64)
    ctx = []
    for flags in (ON, OFF):
        g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024, flags=flags)
        eng.load_scene(g, cfg)
        ctx.append(g)
    d = cfg["dir"].copy()
    seen_late = 0
    for i in range(8):
        a = 0.0 if i < 3 else 0.03 * (i - 2)
        d["Position"][0][:3] = (20.0 * math.cos(a), 20.0 * math.sin(a), 20.0); d["Direction"][0][:3] = d["Position"][0][:3]
        out = []
        for g in ctx:
            g.update_uniforms(cfg["camera"], d, cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)
            g.render(); g.finish()
            out.append((g.shadowmap(), g.color(), g.stats()))
        (sa, ca, ta), (sb, cb, tb) = out
        assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)), "shadow map, frame %d" % i
        assert np.array_equal(ca, cb), "frame %d" % i
        assert ta["survivors"] == tb["survivors"] and ta["overflow"] == 0, (i, ta, tb)
        if i in (1, 2):
            assert ta["shadow_occluded"] > 20000 and ta["shadow_late"] == 0, ta      # a quarter of the casters lies behind the others
        seen_late += ta["shadow_late"]
    assert seen_late > 0
    for g in ctx:
        g.close()


def test_split_shadow_ranks_keep_flags_of_their_own(gpu_engine):
    """zr_set_shadow_partition (the multi-GPU `split` mode): a rank draws its share of the casters, with flags of its own; the element-wise
    minimum of the ranks' maps is the full map, with the culling on or off."""
    W, H, SD = 256, 144, 256
    d, p, s = _lights()
    cam = abi.make_camera((9.0, -7.0, 6.0), (0.0, 0.0, 1.0), fov=50.0)
    full = gpu_engine.Renderer(W, H, SD, flags=OFF)
    ranks = [gpu_engine.Renderer(W, H, SD, flags=ON) for _ in range(2)]
    for r in [full] + ranks:
        _pile(r, 600)
    for k, r in enumerate(ranks):
        r.set_shadow_partition(k, 2)
    for i, lpos in enumerate(LIGHTS[:4]):
        d[0]["Position"][:3] = lpos; d[0]["Direction"][:3] = lpos
        maps = []
        for r in [full] + ranks:
            r.update_uniforms(cam, d, p, s, 0.0, 0.0, 1.0)
            r.render(); r.finish()
            maps.append(r.shadowmap().view(np.uint32))
        assert np.array_equal(maps[0], np.minimum(maps[1], maps[2])), "frame %d" % i
    assert sum(r.stats()["shadow_occluded"] for r in ranks) > 0
    for r in [full] + ranks:
        r.close()
