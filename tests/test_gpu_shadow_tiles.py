"""GPU: the shadow MAP owned by light-space super-tiles (zr_set_shadow_tiles), every rank's context on the one GPU.

Rank r of N draws the casters whose texel box can reach a tile of the map it owns (rank-local work list at the sizes that take one,
instance- and meshlet-level rejects, occlusion flags of its own) - whole, in every window they are listed for - so the tiles it owns
must equal the single-GPU map's bit for bit; zr_shadow_pack -> (all-gather, here: a concatenation on the device) -> zr_shadow_unpack
must leave every rank with the whole map, and the lit frame must be the single-GPU frame.  With a light that moves every frame and
the occlusion history in use.  (RCCL itself needs a GPU per rank: tests/test_gpu_0_multiproc.py runs the same loop over gloo.)
"""
import math

import numpy as np
import pytest

from zeldaengine_amd import abi, dist as zdist, scenes

pytestmark = pytest.mark.gpu


def _scene(r, n_inst, small):
    r.set_cubemap(scenes.synthetic_cubemap(16))
    r.object_add(r.mesh_create(*scenes.grid_plane(40.0, 6, 0.0)))                       # big triangles: the clipper / slow list
    r.object_add(r.mesh_create(*scenes.box((3.0, 3.0, 0.1), (1.0, -1.0, 5.0))))         # an occluder over part of the crowd
    mesh = scenes.uv_sphere(8, 5) if small else scenes.uv_sphere()
    r.object_add(r.mesh_create(*mesh), None, scenes.generate_instances(n_inst, 1.0, 13.0, 0.08 if small else 0.2, 0.3 if small else 0.7, seed=5))


def _lights():
    w = scenes.sample_world()
    d, _, s = scenes.lights_from_world(w)
    w["PointLights"] = scenes.sample_point_lights(4)
    _, p, _ = scenes.lights_from_world(w)
    return d, p, s


def _light(i):
    a = 0.4 + 0.11 * i
    return (15.0 * math.cos(a), 15.0 * math.sin(a), 13.0 + 2.0 * math.sin(1.7 * a))


@pytest.mark.parametrize("world,n_inst,small,flags", [
    (2, 900, False, abi.FLAG_SHADOW_OCCLUSION),        # no work list; occlusion culling forced
    (3, 70000, True, 0),                               # rank-local work lists (>= 65 536 instances); occlusion on by density
    (8, 2500, False, abi.FLAG_SHADOW_OCCLUSION),
    (8, 70000, True, abi.FLAG_NO_SHADOW_OCCLUSION),
])
def test_owned_tiles_assemble_to_the_single_gpu_map(gpu_engine, world, n_inst, small, flags):
    import torch
    W, H, SD, FRAMES = 320, 180, 512, 5
    dev = torch.device("cuda", 0)
    single = gpu_engine.Renderer(W, H, SD, flags=flags)
    ranks = [gpu_engine.Renderer(W, H, SD, flags=flags) for _ in range(world)]
    for r in [single] + ranks:
        _scene(r, n_inst, small)
    for k, r in enumerate(ranks):
        r.set_shadow_tiles(k, world)
    nb = ranks[0].shadow_tiles_bytes()
    lay = zdist.tile_layout(SD, SD, world)
    assert nb == lay["slots_per_rank"] * 32 * 32 * 4
    packed = [torch.ones(nb // 4, dtype=torch.float32, device=dev) for _ in range(world)]
    d, p, s = _lights()
    cam = abi.make_camera((12.0, -9.0, 7.0), (0.0, 0.0, 0.5), fov=50.0)
    drawn = []
    for i in range(FRAMES):
        lp = _light(i // 2 if i < 3 else i)              # frames 0, 1 share a light (exact history), then it moves every frame
        d[0]["Position"][:3] = lp; d[0]["Direction"][:3] = lp
        for r in [single] + ranks:
            r.update_uniforms(cam, d, p, s, 0.0, 0.01 * i, 1.0)
        single.render(); single.finish()
        want = single.shadowmap().view(np.uint32)
        for k, r in enumerate(ranks):
            r.render_geometry()
            r.shadow_pack(packed[k].data_ptr())
        torch.cuda.synchronize()
        # (1) every rank's own map: exact on the tiles it owns
        for k, r in enumerate(ranks):
            # the map as the rank's shadow pass left it (no unpack yet): read through the pack it just made
            got = packed[k].cpu().numpy().view(np.uint32).reshape(lay["slots_per_rank"], 32, 32)
            ref = zdist.pack_tiles(want, k, world, pad=np.uint32(0x3F800000))
            assert np.array_equal(got, ref), "frame %d: rank %d of %d: %d texels of its owned tiles differ" % (i, k, world, int((got != ref).sum()))
        # (2) the exchange: all-gather = concatenation in rank order; every rank scatters it and lights its frame
        gathered = torch.cat(packed).contiguous()
        torch.cuda.synchronize()
        for r in ranks:
            r.shadow_unpack(gathered.data_ptr())
            r.render_lighting()
        for k, r in enumerate(ranks):
            r.finish()
            assert np.array_equal(r.shadowmap().view(np.uint32), want), "frame %d: rank %d: the scattered map differs" % (i, k)
            assert np.array_equal(r.color(), single.color()), "frame %d: rank %d: lit frame differs" % (i, k)
            assert r.stats()["overflow"] == 0
        drawn.append([r.stats()["survivors"][0] for r in ranks] + [single.stats()["survivors"][0]])
    # the share: a rank's shadow pass keeps about 1 / N of the casters (+ those that straddle a super-tile border), not all of them
    last = drawn[-1]
    assert max(last[:-1]) < (0.85 if world == 2 else 0.6) * last[-1], drawn
    assert len(np.unique(want)) > 50
    for r in [single] + ranks:
        r.close()


def test_mode_switches_and_errors(gpu_engine):
    r = gpu_engine.Renderer(64, 64, 256)
    with pytest.raises(gpu_engine.ZeldaRenderError):
        r.shadow_pack(1234)                              # not owned by tiles
    r.set_shadow_tiles(1, 4)
    with pytest.raises(gpu_engine.ZeldaRenderError):
        r.set_shadow_partition(1, 4)                     # one way of sharing the pass at a time
    r.set_shadow_tiles(0, 1)
    r.set_shadow_partition(1, 4)
    with pytest.raises(gpu_engine.ZeldaRenderError):
        r.set_shadow_tiles(1, 4)
    r.close()
