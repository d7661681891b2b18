"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package (zeldaengine_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("zo_oracle.c", "zo_math.h", "zo_oracle.h")]
    outs = (_LIB, os.path.join(_HERE, "_build", "liboracle_literal.so"))
    if force or any(not os.path.exists(o) or any(os.path.getmtime(s) > os.path.getmtime(o) for s in src) for o in outs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_LIB_LITERAL = os.path.join(_HERE, "_build", "liboracle_literal.so")      # -DZO_LITERAL: the IEEE-literal evaluation (oracle/CONTRACT.md)
_libs = {}


def lib(literal=False):
    """The oracle (the contract's evaluation) or, literal=True, its IEEE-literal twin: same entry points."""
    path = _LIB_LITERAL if literal else _LIB
    _lib = _libs.get(path)
    if _lib is None:
        if not os.path.exists(path):
            build(force=True)
        L = C.CDLL(path)
        L.zo_create.restype = C.c_void_p
        L.zo_create.argtypes = [C.c_uint32] * 3
        L.zo_destroy.argtypes = [C.c_void_p]
        L.zo_mesh_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.zo_object_add.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
        L.zo_scene_clear.argtypes = [C.c_void_p]
        L.zo_set_cubemap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.zo_set_skydome.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.zo_set_background.argtypes = [C.c_void_p, C.c_void_p]
        L.zo_set_sky_flags.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.zo_update_uniforms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_float]
        L.zo_set_frame.argtypes = [C.c_void_p] * 4
        L.zo_get_frame.argtypes = [C.c_void_p] * 4
        L.zo_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.zo_set_shading.argtypes = [C.c_void_p, C.c_int]
        for n in ("zo_color", "zo_shadowmap", "zo_visibility"):
            getattr(L, n).restype = C.c_void_p
            getattr(L, n).argtypes = [C.c_void_p]
        L.zo_gbuffer.restype = C.c_void_p
        L.zo_gbuffer.argtypes = [C.c_void_p, C.c_int]
        L.zo_covered_pixels.restype = C.c_uint64
        L.zo_covered_pixels.argtypes = [C.c_void_p]
        L.zo_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.zo_meshlet_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        for n, args, res in [
            ("zo_kat_D_GGX", 2, C.c_float), ("zo_kat_V_SmithGGXCorrelated", 3, C.c_float),
            ("zo_kat_F_Schlick", 3, C.c_float), ("zo_kat_Fr_DisneyDiffuse", 4, C.c_float),
            ("zo_kat_ReflectionMip", 2, C.c_float), ("zo_kat_exp2", 1, C.c_float), ("zo_kat_log2", 1, C.c_float),
            ("zo_kat_pow", 2, C.c_float), ("zo_kat_rsqrt", 1, C.c_float)]:
            getattr(L, n).restype = res
            getattr(L, n).argtypes = [C.c_float] * args
        L.zo_kat_rsqrt_worst.restype = C.c_double
        L.zo_kat_rsqrt_worst.argtypes = [C.c_uint32] * 3
        L.zo_kat_srgb8_to_linear.restype = C.c_float
        L.zo_kat_srgb8_to_linear.argtypes = [C.c_uint32]
        L.zo_kat_f32_to_f16.restype = C.c_uint16
        L.zo_kat_f32_to_f16.argtypes = [C.c_float]
        L.zo_kat_EnvBRDFApproxLazarov.argtypes = [C.c_float, C.c_float, C.c_void_p]
        L.zo_kat_default_normal_ts.argtypes = [C.c_void_p]
        L.zo_kat_perspective.argtypes = [C.c_float] * 4 + [C.c_void_p]
        L.zo_kat_lookat.argtypes = [C.c_void_p] * 4
        L.zo_kat_sincos.argtypes = [C.c_float, C.c_void_p]
        L.zo_kat_rotmat.argtypes = [C.c_void_p, C.c_void_p]
        L.zo_kat_aniso.argtypes = [C.c_float] * 4 + [C.c_int, C.c_void_p]
        L.zo_kat_tex_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.zo_kat_tex_mip.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        L.zo_kat_cube_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.zo_kat_cube_mip.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib = _libs[path] = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Oracle:
    """Same call sequence as zeldaengine_amd.engine.Renderer, on the CPU."""

    def __init__(self, width, height, shadow_dim=1024, literal=False):
        self.L = lib(literal)
        self.W, self.H, self.SD = width, height, shadow_dim
        self.h = self.L.zo_create(width, height, shadow_dim)
        self._keep = []

    def close(self):
        if self.h:
            self.L.zo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def mesh_create(self, verts, idx):
        verts = np.ascontiguousarray(verts)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        r = self.L.zo_mesh_create(self.h, _ptr(verts), len(verts), _ptr(idx), len(idx))
        assert r >= 0
        return r

    def object_add(self, mesh, material=None, instances=None):
        mat = C.byref(material) if material is not None else None
        n = 0 if instances is None else len(instances)
        inst = np.ascontiguousarray(instances) if n else None
        r = self.L.zo_object_add(self.h, mesh, mat, _ptr(inst) if n else None, n)
        assert r >= 0
        return r

    def scene_clear(self):
        self.L.zo_scene_clear(self.h)

    def set_cubemap(self, faces):
        if faces is None:
            self.L.zo_set_cubemap(self.h, None, 0)
            return
        faces = [np.ascontiguousarray(f, dtype=np.uint8) for f in faces]
        arr = (C.c_void_p * 6)(*[f.ctypes.data for f in faces])
        assert self.L.zo_set_cubemap(self.h, arr, faces[0].shape[0]) == 0

    def set_skydome(self, verts, idx, image):
        from zeldaengine_amd import abi
        if image is None:
            self.L.zo_set_skydome(self.h, None, 0, None, 0, None)
            return
        verts = np.ascontiguousarray(verts); idx = np.ascontiguousarray(idx, dtype=np.uint32)
        img = np.ascontiguousarray(image, dtype=np.uint8)
        im = abi.Image(img.ctypes.data, img.shape[1], img.shape[0])
        self.L.zo_set_skydome(self.h, _ptr(verts), len(verts), _ptr(idx), len(idx), C.byref(im))

    def set_background(self, image):
        from zeldaengine_amd import abi
        if image is None:
            self.L.zo_set_background(self.h, None)
            return
        img = np.ascontiguousarray(image, dtype=np.uint8)
        im = abi.Image(img.ctypes.data, img.shape[1], img.shape[0])
        self.L.zo_set_background(self.h, C.byref(im))

    def set_sky_flags(self, enable_skydome=True, enable_background=True):
        self.L.zo_set_sky_flags(self.h, int(enable_skydome), int(enable_background))

    def update_uniforms(self, cam, dir_l, point_l, spot_l, roll_stage=0.0, roll_light=0.0, time=0.0):
        self.L.zo_update_uniforms(self.h, C.byref(cam), _ptr(dir_l), len(dir_l), _ptr(point_l), len(point_l),
                                  _ptr(spot_l), len(spot_l), roll_stage, roll_light, time)

    def get_frame(self):
        from zeldaengine_amd import abi
        cam = np.zeros((), dtype=abi.XkUniformBufferMVP)
        sh = np.zeros((), dtype=abi.XkUniformBufferMVP)
        view = np.zeros((), dtype=abi.XkView)
        self.L.zo_get_frame(self.h, cam.ctypes.data, sh.ctypes.data, view.ctypes.data)
        return cam, sh, view

    def set_frame(self, cam, sh, view):
        self.L.zo_set_frame(self.h, cam.ctypes.data, sh.ctypes.data, view.ctypes.data)

    def set_shading(self, forward):
        """False: the deferred frame; True: the forward variant (SH/Base.frag)."""
        self.L.zo_set_shading(self.h, int(bool(forward)))

    def render(self, debug_view=0, passes=7):
        self.L.zo_render(self.h, debug_view, passes)

    def _view(self, ptr, dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = (C.c_uint8 * n).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()

    def color(self):
        return self._view(self.L.zo_color(self.h), np.uint8, (self.H, self.W, 4))

    def gbuffer(self, target):
        dt = [np.float32, np.uint32, np.uint32, np.uint32, np.uint32, np.uint64][target]
        return self._view(self.L.zo_gbuffer(self.h, target), dt, (self.H, self.W))

    def shadowmap(self):
        return self._view(self.L.zo_shadowmap(self.h), np.float32, (self.SD, self.SD))

    def visibility(self):
        return self._view(self.L.zo_visibility(self.h), np.uint32, (self.H, self.W))

    # sampler KATs (tests/independent_sampler.py)
    def tex_sample(self, image, srgb, uv, duv):
        """texture(sampler2D, uv) of an (h, w, 4) uint8 image with the screen-space derivatives duv = (dudx, dvdx, dudy, dvdy)"""
        im = np.ascontiguousarray(image, np.uint8)
        a = np.asarray(uv, np.float32); d = np.asarray(duv, np.float32); out = np.zeros(4, np.float32)
        self.L.zo_kat_tex_sample(self.h, _ptr(im), im.shape[1], im.shape[0], int(srgb), _ptr(a), _ptr(d), _ptr(out))
        return out

    def tex_mips(self, image, srgb):
        im = np.ascontiguousarray(image, np.uint8)
        h, w = im.shape[:2]
        out, level = [], 0
        while True:
            lw, lh = max(1, w >> level), max(1, h >> level)
            buf = np.zeros((lh, lw, 4), np.uint8)
            n = self.L.zo_kat_tex_mip(self.h, _ptr(im), w, h, int(srgb), level, _ptr(buf))
            out.append(buf); level += 1
            if level >= n:
                return out

    def cube_sample(self, direction, lod):
        d = np.asarray(direction, np.float32); out = np.zeros(3, np.float32)
        self.L.zo_kat_cube_sample(self.h, _ptr(d), float(lod), _ptr(out))
        return out

    def cube_mips(self, dim):
        out, level = [], 0
        while (dim >> level) >= 1:
            d = dim >> level
            buf = np.zeros((6, d, d, 4), np.uint8)
            if self.L.zo_kat_cube_mip(self.h, level, _ptr(buf)) != d:
                break
            out.append(buf); level += 1
        return out

    def set_threads(self, n):
        """OpenMP team for the per-pixel stages (resolve, lighting): the all-cores CPU baseline.  Results do not change."""
        self.L.zo_set_threads(self.h, int(n))

    def covered_pixels(self):
        return int(self.L.zo_covered_pixels(self.h))


def load_scene(r, cfg):
    """Feed a scenes.config*() dict to an Oracle or a Renderer (same method names)."""
    r.set_cubemap(cfg.get("cubemap"))
    for o in cfg["objects"]:
        m = r.mesh_create(*o["mesh"])
        r.object_add(m, o.get("material"), o.get("instances"))
    r.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"])
