"""GPU parity proper: the HIP path (through the C-ABI) against the CPU oracle, bit for bit.

Integer/packed targets (depth bits, UNORM8, A2R10G10B10, fp16 position, shadow-map bits) and the final RGBA8
frame must be IDENTICAL: both sides evaluate the same fixed op sequence, so the tolerance is 0.
"""
import numpy as np
import pytest

from parity_util import compare_all, render_both

pytestmark = pytest.mark.gpu


def _assert_identical(diffs):
    bad = {k: v for k, v in diffs.items() if v}
    assert not bad, "HIP path differs from the oracle: %r" % bad


def test_config2_single_sphere_512(oracle_lib, gpu_engine):
    from zeldaengine_amd import scenes
    o, g = render_both(oracle_lib, gpu_engine, scenes.config2())
    assert o.covered_pixels() > 1000
    assert g.stats()["covered_pixels"] == o.covered_pixels()
    _assert_identical(compare_all(o, g))


def test_config3_small_instanced_with_shadows(oracle_lib, gpu_engine):
    from zeldaengine_amd import scenes
    o, g = render_both(oracle_lib, gpu_engine, scenes.config3(400, 640, 360))
    assert (o.shadowmap() < 1.0).sum() > 1000
    _assert_identical(compare_all(o, g))


def test_culling_is_invisible(oracle_lib, gpu_engine):
    """Frustum + cone culling must not change a single word of any target."""
    from zeldaengine_amd import abi, engine, scenes
    cfg = scenes.config3(300, 480, 270)
    frames = []
    for flags in (0, abi.FLAG_NO_FRUSTUM_CULL | abi.FLAG_NO_CONE_CULL):
        g = engine.Renderer(cfg["width"], cfg["height"], flags=flags)
        engine.load_scene(g, cfg)
        g.render()
        frames.append((g.color(), [g.gbuffer(t) for t in range(6)], g.shadowmap(), g.stats()))
    assert frames[0][3]["survivors"][1] < frames[1][3]["survivors"][1]
    assert np.array_equal(frames[0][0], frames[1][0])
    for a, b in zip(frames[0][1], frames[1][1]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert np.array_equal(frames[0][2].view(np.uint32), frames[1][2].view(np.uint32))
