"""Small oracle frames whose SHA-256 is pinned under tests/golden/ (regression pins of the oracle's own arithmetic: any change to
the restated evaluation order, filters or raster rules must be deliberate).  Shared by make_golden_oracle.py and the test."""
import hashlib

import numpy as np


def _digest(o, view=0):
    o.render(view)
    h = hashlib.sha256()
    h.update(o.color().tobytes())
    for t in range(6):
        h.update(o.gbuffer(t).tobytes())
    h.update(o.shadowmap().tobytes())
    return h.hexdigest()


def case_textured_aniso(pyoracle, abi, scenes):
    """Striped, sRGB, mip-mapped ground plane at a grazing angle (anisotropic taps 1..16), a normal-mapped box, spheres."""
    o = pyoracle.Oracle(160, 96, 64)
    o.set_cubemap(scenes.synthetic_cubemap(16))
    img = np.zeros((32, 32, 4), dtype=np.uint8); img[..., 3] = 255; img[:, ::2, :3] = 255; img[::4, :, 0] = 60
    nrm = np.zeros((16, 16, 4), dtype=np.uint8); nrm[..., 2] = 255; nrm[..., 3] = 255
    nrm[..., 0] = (np.arange(16) * 13 % 256)[None, :]; nrm[..., 1] = (np.arange(16) * 7 % 256)[:, None]
    mat, keep = abi.make_material([img, None, None, nrm, None, None, None])
    v, idx = scenes.grid_plane(40.0, 4, 0.0)
    v = v.copy(); v["TexCoord"] *= 3.0
    o.object_add(o.mesh_create(v, idx), mat)
    o.object_add(o.mesh_create(*scenes.box((0.7, 0.7, 0.7), (0, 0, 0.7))), mat)
    o.object_add(o.mesh_create(*scenes.uv_sphere()), None, scenes.generate_instances(30, 1.0, 6.0, 0.3, 0.8, seed=4))
    w = scenes.sample_world()
    d, p, s = scenes.lights_from_world(w)
    o.update_uniforms(abi.make_camera((0.0, -9.0, 0.6), (0.0, 6.0, 0.2), fov=55.0, znear=0.05, zfar=120.0), d, p, s, 0.3, 0.01, 2.0)
    return _digest(o), keep


def case_debug_view_9(pyoracle, abi, scenes):
    """GBufferVis mosaic with editor bars (bilinear re-sampling with snapped weights, white cell frames)."""
    cfg = scenes.config3(60, 150, 90)
    o = pyoracle.Oracle(150, 90, 64)
    pyoracle.load_scene(o, cfg)
    cam, sh, view = o.get_frame()
    view["ViewportInfo"][2] = 21.0; view["ViewportInfo"][3] = 12.0
    o.set_frame(cam, sh, view)
    return _digest(o, 9), None


def case_clipping_256_lights(pyoracle, abi, scenes):
    """Camera inside the crowd, huge ground quad: near-plane and guard-band clipping; 256 point lights."""
    o = pyoracle.Oracle(128, 80, 48)
    o.set_cubemap(None)
    o.object_add(o.mesh_create(*scenes.grid_plane(400.0, 2, 0.0)))
    o.object_add(o.mesh_create(*scenes.uv_sphere()), None, scenes.generate_instances(40, 0.2, 2.5, 0.8, 1.6, seed=3))
    w = scenes.sample_world()
    d, _, s = scenes.lights_from_world(w)
    w["PointLights"] = scenes.sample_point_lights(256)
    _, p, _ = scenes.lights_from_world(w)
    o.update_uniforms(abi.make_camera((0.3, -0.6, 0.45), (0.0, 4.0, 0.6), fov=75.0, znear=0.05, zfar=200.0), d, p, s, 0.0, 0.0, 1.0)
    return _digest(o), None


CASES = {"textured_aniso": case_textured_aniso, "debug_view_9": case_debug_view_9, "clipping_256_lights": case_clipping_256_lights}
