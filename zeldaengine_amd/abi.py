"""numpy / ctypes mirrors of include/zelda_abi.h (the ZeldaEngine submission structs).

Reference layouts: XkVertex ZE:417-422, XkInstanceData ZE:409-414, XkMeshlet ZE:689-701,
XkLight ZE:772-787, XkUniformBufferMVP ZE:384-388, XkView ZE:922-940
(ZE = Engine/ZeldaEngine/ZeldaEngine.cpp).
"""
import ctypes as C

import numpy as np

MAX_DIRECTIONAL_LIGHTS = 16
MAX_POINT_LIGHTS = 512
MAX_SPOT_LIGHTS = 16
SHADOWMAP_DIM = 1024
TILE = 32

XkVertex = np.dtype([("Position", "<f4", 3), ("Normal", "<f4", 3), ("Color", "<f4", 3), ("TexCoord", "<f4", 2)])
XkInstanceData = np.dtype([("InstancePosition", "<f4", 3), ("InstanceRotation", "<f4", 3), ("InstancePScale", "<f4"),
                           ("InstanceTexIndex", "u1"), ("_pad", "u1", 3)])
XkMeshlet = np.dtype([("VertexOffset", "<u4"), ("VertexCount", "<u4"), ("TriangleOffset", "<u4"), ("TriangleCount", "<u4"),
                      ("BoundsCenter", "<f4", 3), ("BoundsRadius", "<f4"), ("ConeApex", "<f4", 3), ("ConeAxis", "<f4", 3),
                      ("ConeCutoff", "<f4"), ("BindlessContext", "<u4")])
XkMeshletFileVertex = np.dtype([("pos", "<f4", 3), ("nrm", "<f4", 3), ("uv", "<f4", 2)])
XkLight = np.dtype([("Position", "<f4", 4), ("Color", "<f4", 4), ("Direction", "<f4", 4), ("LightInfo", "<f4", 4)])
XkUniformBufferMVP = np.dtype([("Model", "<f4", 16), ("View", "<f4", 16), ("Proj", "<f4", 16)])
XkView = np.dtype([("ViewProjSpace", "<f4", 16), ("ShadowmapSpace", "<f4", 16), ("LocalToWorld", "<f4", 16),
                   ("CameraInfo", "<f4", 4), ("ViewportInfo", "<f4", 4),
                   ("DirectionalLights", XkLight, MAX_DIRECTIONAL_LIGHTS), ("PointLights", XkLight, MAX_POINT_LIGHTS),
                   ("SpotLights", XkLight, MAX_SPOT_LIGHTS), ("LightsCount", "<i4", 4),
                   ("Time", "<f4"), ("zNear", "<f4"), ("zFar", "<f4")])

assert XkVertex.itemsize == 44 and XkInstanceData.itemsize == 32 and XkMeshlet.itemsize == 64
assert XkLight.itemsize == 64 and XkUniformBufferMVP.itemsize == 192 and XkView.itemsize == 35068


class Image(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("tex", Image * 7)]


class Camera(C.Structure):
    _fields_ = [("Position", C.c_float * 3), ("Lookat", C.c_float * 3), ("Speed", C.c_float), ("FOV", C.c_float),
                ("zNear", C.c_float), ("zFar", C.c_float)]


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("shadow_dim", C.c_uint32), ("debug_view", C.c_uint32),
                ("device", C.c_int32), ("tile_rank", C.c_uint32), ("tile_world", C.c_uint32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("work_items", C.c_uint64 * 2), ("survivors", C.c_uint64 * 2), ("bin_entries", C.c_uint64 * 2),
                ("covered_pixels", C.c_uint64), ("covered_shadow_texels", C.c_uint64), ("overflow", C.c_uint32), ("hiz_culled", C.c_uint32),
                ("round1_survivors", C.c_uint64), ("shadow_occluded", C.c_uint32), ("shadow_late", C.c_uint32),
                ("hiz_culled_geom", C.c_uint32), ("struct_bytes", C.c_uint32)]


ABI_VERSION = 6      # ZR_ABI_VERSION of include/zelda_render.h


PASS_NAMES = ["cull_shadow", "shadow", "cull_camera", "gbuffer", "hiz", "gbuffer2", "resolve", "lighting", "composite", "total"]
GBUFFER_DTYPES = [np.dtype("<f4"), np.dtype("<u4"), np.dtype("<u4"), np.dtype("<u4"), np.dtype("<u4"), np.dtype("<u8")]

FLAG_NO_FRUSTUM_CULL = 1
FLAG_NO_CONE_CULL = 2
FLAG_SKIP_COMPOSITE = 4
FLAG_NO_HIZ = 8
FLAG_SERIAL_PASSES = 16
FLAG_PACKED_TILES = 32
FLAG_NO_RECT_CULL = 64
FLAG_MESHLET_BINS = 128
FLAG_NO_LIST_REUSE = 256
FLAG_NO_SHADOW_OCCLUSION = 512
FLAG_SHADOW_OCCLUSION = 1024

OK, ERR_ARG, ERR_DEVICE, ERR_OOM, ERR_PARSE, ERR_IO, ERR_STATE, ERR_OVERFLOW, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7, -8


def make_light(position=(0, 0, 0), type_=0, color=(1, 1, 1), intensity=1.0, direction=(0, 0, 1), radius=0.0,
               extra=(0, 0, 0, 0)):
    """XkLight(const XkLightDesc&), ZE:781-787."""
    l = np.zeros((), dtype=XkLight)
    l["Position"] = (*position, float(type_))
    l["Color"] = (*color, intensity)
    l["Direction"] = (*direction, radius)
    l["LightInfo"] = extra
    return l


def make_camera(position=(5.0, 5.0, 5.0), lookat=(0.0, 0.0, 0.0), speed=2.5, fov=45.0, znear=0.1, zfar=45.0):
    """XkCameraDesc defaults, ZE:882-887."""
    return Camera((C.c_float * 3)(*position), (C.c_float * 3)(*lookat), speed, fov, znear, zfar)


def make_material(images=None):
    """images: list of 7 entries, each None (engine default) or an (H, W, 4) uint8 array.  Returns (Material, keepalive)."""
    m = Material()
    keep = []
    for i in range(7):
        img = images[i] if images else None
        if img is None:
            m.tex[i] = Image(None, 0, 0)
        else:
            a = np.ascontiguousarray(img, dtype=np.uint8)
            assert a.ndim == 3 and a.shape[2] == 4
            keep.append(a)
            m.tex[i] = Image(a.ctypes.data, a.shape[1], a.shape[0])
    return m, keep


DIST_SPLIT_SHADOW = 1      # ZR_DIST_SPLIT_SHADOW: casters i % world == rank + ncclAllReduce(min) of the maps
DIST_SHADOW_TILES = 2      # ZR_DIST_SHADOW_TILES: the map owned by light-space super-tiles + ncclAllGather of the packed tiles
SHADOW_MODES = ("replicated", "split", "tiles")


def shadow_mode(split_shadow=False, mode=None):
    """The multi-GPU shadow mode from the two ways hosts name it: mode in SHADOW_MODES, or the older split_shadow flag."""
    if mode is None:
        mode = "split" if split_shadow is True else (split_shadow if isinstance(split_shadow, str) else "replicated")
    if mode not in SHADOW_MODES:
        raise ValueError("shadow mode %r: one of %r" % (mode, SHADOW_MODES))
    return mode


def dist_flags(mode):
    return {"replicated": 0, "split": DIST_SPLIT_SHADOW, "tiles": DIST_SHADOW_TILES}[mode]
