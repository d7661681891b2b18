"""Two frames in flight with a light that moves EVERY frame, at a size that takes the instance-level work lists (>= 65 536 instances).

zr_render only enqueues: frame N + 1's camera lane (k_frame_begin first) starts while frame N's shadow pipeline still runs on the render
stream.  Whatever the two share must be ordered by the streams themselves - round 4 zeroed the shadow work list's length from the camera
lane (frame N + 1) under frame N's shadow pipeline, and no test saw it because every comparison called finish() right after the frame it
read.  Here frame N is copied out ON THE DEVICE, in stream order (zr_copy_frame_async), frame N + 1 is enqueued behind it without a
host synchronisation, and the copies are compared afterwards with a one-stream context (ZR_FLAG_SERIAL_PASSES) that renders the same
sequence with a finish() after every frame.
"""
import math

import numpy as np
import pytest

from zeldaengine_amd import abi, scenes

pytestmark = pytest.mark.gpu


def _scene(r, n_inst):
    r.set_cubemap(scenes.synthetic_cubemap(16))
    r.object_add(r.mesh_create(*scenes.grid_plane(40.0, 4, 0.0)))
    r.object_add(r.mesh_create(*scenes.uv_sphere(8, 5)), None, scenes.generate_instances(n_inst, 1.0, 14.0, 0.05, 0.2, seed=11))


def _light(i):
    a = 0.3 + 0.09 * i
    return (16.0 * math.cos(a), 16.0 * math.sin(a), 14.0 + 1.5 * math.sin(2.0 * a))


@pytest.mark.parametrize("flags", [0, abi.FLAG_SHADOW_OCCLUSION, abi.FLAG_NO_LIST_REUSE])
def test_every_frame_of_a_queued_sequence_matches_the_serial_context(gpu_engine, flags):
    import torch
    W, H, SD, N, FRAMES = 320, 180, 512, 70000, 14
    ref = gpu_engine.Renderer(W, H, SD, flags=flags | abi.FLAG_SERIAL_PASSES)
    g = gpu_engine.Renderer(W, H, SD, flags=flags)
    for r in (ref, g):
        _scene(r, N)
    w = scenes.sample_world()
    d, _, s = scenes.lights_from_world(w)
    w["PointLights"] = scenes.sample_point_lights(4)
    _, p, _ = scenes.lights_from_world(w)
    cam = abi.make_camera((12.0, -9.0, 7.0), (0.0, 0.0, 0.5), fov=50.0)
    want = []
    for i in range(FRAMES):
        lp = _light(i)
        d[0]["Position"][:3] = lp; d[0]["Direction"][:3] = lp
        ref.update_uniforms(cam, d, p, s, 0.0, 0.01 * i, 1.0)
        ref.render(); ref.finish()
        want.append((ref.color().copy(), ref.shadowmap().view(np.uint32).copy()))
    assert ref.stats()["work_items"][0] >= 65536              # the work-list path
    dev = torch.device("cuda", 0)
    col = [torch.zeros(W * H, dtype=torch.int32, device=dev) for _ in range(FRAMES)]
    sha = [torch.zeros(SD * SD, dtype=torch.int32, device=dev) for _ in range(FRAMES)]
    torch.cuda.synchronize()
    for i in range(FRAMES):                                   # back to back: no finish() until every frame is enqueued
        lp = _light(i)
        d[0]["Position"][:3] = lp; d[0]["Direction"][:3] = lp
        g.update_uniforms(cam, d, p, s, 0.0, 0.01 * i, 1.0)
        g.render()
        g.copy_frame_async(col[i].data_ptr(), sha[i].data_ptr())
    g.finish()
    assert g.stats()["overflow"] == 0
    for i in range(FRAMES):
        got_s = sha[i].cpu().numpy().view(np.uint32).reshape(SD, SD)
        got_c = col[i].cpu().numpy().view(np.uint8).reshape(H, W, 4)
        assert np.array_equal(got_s, want[i][1]), "shadow map of queued frame %d: %d texels differ" % (i, int((got_s != want[i][1]).sum()))
        assert np.array_equal(got_c, want[i][0]), "colour of queued frame %d" % i
    assert len(np.unique(want[-1][1])) > 50                    # a map with casters in it, not a clear
    ref.close(); g.close()
