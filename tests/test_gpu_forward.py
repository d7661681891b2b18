"""The forward variant: SH/Base.frag:46-144 through zr_set_shading(ZR_SHADING_FORWARD), bit for bit against the oracle.

The engine picks one of its two scene pipelines at compile time (ZE:93); Base.frag shades straight into the frame from unquantised inputs,
without Mask, as FinalColor * ShadowFactor, with a debug table of its own (:123-143).  Every test renders the scene on the oracle
(zo_set_shading) and through the C-ABI and compares the frame - and everything else the frame leaves behind - exactly.
"""
import numpy as np
import pytest

from parity_util import compare_all
from test_gpu_scenes import _mixed_scene, _std_frame
from zeldaengine_amd import abi, dist as zdist, scenes

pytestmark = pytest.mark.gpu


def _pair(oracle_lib, gpu_engine, W, H, SD, build, frame, flags=0):
    o = oracle_lib.Oracle(W, H, SD)
    g = gpu_engine.Renderer(W, H, SD, flags=flags)
    for r in (o, g):
        build(r)
        frame(r)
        r.set_shading(True)
    return o, g


def _same(o, g, what):
    bad = {k: v for k, v in compare_all(o, g).items() if v}
    assert not bad, "%s: HIP path differs from the oracle: %r" % (what, bad)


def _coloured_mixed_scene(r):
    """_mixed_scene plus a draw whose vertex colours are not (1, 1, 1): view 6 of Base.frag shows the interpolated fragColor."""
    _mixed_scene(r, 12)
    v, idx = scenes.box((0.6, 0.6, 0.9), (1.5, -1.0, 0.9))
    v = v.copy()
    rng = np.random.default_rng(5)
    v["Color"] = rng.random((len(v), 3), dtype=np.float32)
    r.object_add(r.mesh_create(v, idx))


@pytest.mark.parametrize("view", list(range(10)) + [12])
def test_forward_views_on_the_mixed_scene(oracle_lib, gpu_engine, view):
    o, g = _pair(oracle_lib, gpu_engine, 192, 128, 128, _coloured_mixed_scene, _std_frame())
    for _ in range(2):                      # second frame: visibility history, Hi-Z pyramid, two rounds
        o.render(view); g.render(view)
    g.finish()
    _same(o, g, "forward view %d" % view)
    c = g.color()
    assert (c[..., 3] == 255).all() and (c[..., :3] != 0).any()
    if view in (0, 9, 12):                  # Base.frag:123,141,143: the same output for 0, 9 and anything unlisted
        o.render(0)
        assert np.array_equal(o.color(), c)


def test_forward_differs_from_deferred_where_the_shaders_do(oracle_lib, gpu_engine):
    """Empty pixels hold the clear colour (the deferred quad shades them), covered ones carry the extra ShadowFactor and no 8 / 10 / 16-bit
    round trip; switching back restores the deferred frame exactly, on the same context."""
    g = gpu_engine.Renderer(192, 128, 128)
    _mixed_scene(g, 12); _std_frame()(g)
    g.render(); g.finish()
    deferred, depth = g.color(), g.gbuffer(0)
    g.set_shading(True)
    g.render(); g.render(); g.finish()
    forward = g.color()
    assert np.array_equal(depth.view(np.uint32), g.gbuffer(0).view(np.uint32))
    empty = depth == 1.0
    assert empty.any() and (forward[empty][:, :3] == 0).all() and (deferred[empty][:, :3] != 0).any()
    assert (forward[~empty] != deferred[~empty]).any()
    g.set_shading(False)
    g.render(); g.finish()
    assert np.array_equal(deferred, g.color())
    with pytest.raises(gpu_engine.ZeldaRenderError):
        g.render_shadow(); g.set_shading(True)


def test_forward_config3_reduced(oracle_lib, gpu_engine):
    cfg = scenes.config3(1500, 640, 360)
    o = oracle_lib.Oracle(640, 360, 512); oracle_lib.load_scene(o, cfg)
    g = gpu_engine.Renderer(640, 360, 512); gpu_engine.load_scene(g, cfg)
    o.set_threads(8)
    for r in (o, g):
        r.set_shading(True)
    o.render()
    g.render(); g.render(); g.finish()
    _same(o, g, "config 3 (1500 spheres), forward")
    assert o.covered_pixels() > 20000


def test_forward_with_sampled_materials_sky_and_background(oracle_lib, gpu_engine):
    """Image slots go through the same filter as BaseScene.frag's (all seven slots of a Profab material; Base.frag reads five of them),
    the skydome and the background follow in the same pass (view 0 only, ZE:3681-3699)."""
    cfg = scenes.config3(300, 320, 180, textured=True)
    bg = scenes.synthetic_sky_image(90, 60)[:, ::-1].copy()
    o = oracle_lib.Oracle(320, 180, 256); oracle_lib.load_scene(o, cfg)
    g = gpu_engine.Renderer(320, 180, 256); gpu_engine.load_scene(g, cfg)
    for r in (o, g):
        r.set_skydome(*scenes.sky_dome(), scenes.synthetic_sky_image())
        r.set_background(bg)
        r.set_shading(True)
    for view in (0, 1, 5, 4):
        o.render(view); g.render(view); g.finish()
        _same(o, g, "textured forward view %d" % view)


def test_forward_on_tile_partitioned_contexts(gpu_engine):
    """Rank contexts of a 3-rank job shade forward into packed tiles that assemble to the single context's frame."""
    cfg = scenes.config3(800, 640, 360)
    single = gpu_engine.Renderer(640, 360, 256); gpu_engine.load_scene(single, cfg)
    single.set_shading(True); single.render(); single.render(); single.finish()
    want = single.color()
    for r in range(3):
        g = gpu_engine.Renderer(640, 360, 256, tile_rank=r, tile_world=3); gpu_engine.load_scene(g, cfg)
        g.set_shading(True); g.render(); g.render(); g.finish()
        assert np.array_equal(g.read_tiles(), zdist.pack_tiles(want, r, 3))
        g.close()
