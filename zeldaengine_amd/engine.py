"""ctypes binding of libzelda_render.so — the HIP renderer behind the C-ABI of include/zelda_render.h.

`Renderer` mirrors the call sequence of XkZeldaEngineApp (ZE = Engine/ZeldaEngine/ZeldaEngine.cpp):
CreateEngineScene (ZE:4140) -> mesh_create / object_add, UpdateUniformBuffer (ZE:4585) -> update_uniforms,
DrawFrame (ZE:1940) -> render, plus read-back in place of the swapchain.

There is no CPU path here: the library fails loudly when the HIP extension or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZELDA_RENDER_LIB") or os.path.join(_HERE, "libzelda_render.so")   # env: A/B builds only

_lib = None


class ZeldaRenderError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("zelda_render error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Loads the in-tree shared library (built by zeldaengine_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libzelda_render.so is not built: run `python -m zeldaengine_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no fallback path")
    try:
        import torch  # noqa: F401  -- first, so that this process has ONE HIP runtime (torch's bundled libamdhip64.so)
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    sig = {
        "zr_create": [C.POINTER(abi.Config), C.POINTER(vp)],
        "zr_set_stream": [vp, vp],
        "zr_mesh_create": [vp, vp, u32, vp, u32, C.POINTER(u32)],
        "zr_mesh_set_meshlets": [vp, u32, vp, u32, vp, sz, vp, sz],
        "zr_mesh_build_meshlets": [vp, u32, u32, u32, C.c_float],
        "zr_mesh_get_meshlets": [vp, u32, vp, C.POINTER(u32), vp, C.POINTER(sz), vp, C.POINTER(sz)],
        "zr_meshlets_build": [vp, u32, vp, u32, u32, u32, C.c_float, vp, C.POINTER(u32), vp, C.POINTER(sz), vp, C.POINTER(sz), vp],
        "zr_object_add": [vp, u32, vp, vp, u32],
        "zr_scene_clear": [vp],
        "zr_set_limits": [vp, u32, u32],
        "zr_set_bucket_share": [vp, u32],
        "zr_object_count": [vp, C.POINTER(u32)],
        "zr_object_get_instances": [vp, u32, C.POINTER(u32), vp, C.POINTER(u32)],
        "zr_set_cubemap": [vp, vp, u32],
        "zr_set_skydome": [vp, vp, u32, vp, u32, vp],
        "zr_set_background": [vp, vp],
        "zr_set_sky_flags": [vp, C.c_int, C.c_int],
        "zr_update_uniforms": [vp, vp, vp, u32, vp, u32, vp, u32, C.c_float, C.c_float, C.c_float],
        "zr_set_frame": [vp, vp, vp, vp],
        "zr_get_frame": [vp, vp, vp, vp],
        "zr_set_debug_view": [vp, u32],
        "zr_set_shading": [vp, u32],
        "zr_render": [vp],
        "zr_render_shadow": [vp], "zr_render_gbuffer": [vp], "zr_render_lighting": [vp],
        "zr_render_geometry": [vp], "zr_stream_wait_shadow": [vp, vp],
        "zr_set_shadow_partition": [vp, u32, u32], "zr_set_shadow_buffer": [vp, vp],
        "zr_set_shadow_tiles": [vp, u32, u32], "zr_shadow_tiles_bytes": [vp, C.POINTER(sz)],
        "zr_shadow_pack": [vp, vp, vp], "zr_shadow_unpack": [vp, vp, vp],
        "zr_finish": [vp],
        "zr_get_pass_times": [vp, vp],
        "zr_get_pass_times_avg": [vp, u32, vp],
        "zr_set_timing_interval": [vp, u32],
        "zr_get_frame_latencies": [vp, u32, vp],
        "zr_get_frame_periods": [vp, u32, vp],
        "zr_get_stats": [vp, C.POINTER(abi.Stats), sz],
        "zr_read_color": [vp, vp, sz],
        "zr_read_gbuffer": [vp, C.c_int, vp, sz],
        "zr_read_shadowmap": [vp, vp, sz],
        "zr_copy_frame_async": [vp, vp, vp],
        "zr_tiles_device_buffer": [vp, C.POINTER(vp), C.POINTER(sz)],
        "zr_composite": [vp, vp],
        "zr_read_tiles": [vp, vp, sz],
        "zr_set_tiles_buffer": [vp, vp],
        "zr_tile_size": [],
        "zr_color_device_ptr": [vp, C.POINTER(vp)],
        "zr_tile_partition": [u32, u32, u32, u32, vp, C.POINTER(u32), C.POINTER(u32)],
        "zr_dist_unique_id": [vp, sz],
        "zr_dist_init": [vp, vp, sz, u32, u32, u32],
        "zr_dist_prepare": [vp, u32, u32, u32],
        "zr_dist_connect": [vp, vp, sz],
        "zr_dist_frame": [vp],
        "zr_dist_copy_frame_async": [vp, vp],
        "zr_profab_register": [vp, C.c_char_p, u32, vp],
        "zr_world_load_json": [vp, C.c_char_p, sz],
        "zr_set_asset_root": [vp, C.c_char_p],
        "zr_asset_path_search": [vp, C.c_char_p, vp, sz, C.POINTER(sz)],
        "zr_load_obj": [C.c_char_p, vp, C.POINTER(u32), vp, C.POINTER(u32)],
        "zr_load_png_rgba8": [C.c_char_p, vp, sz, C.POINTER(u32), C.POINTER(u32)],
        "zr_load_meshlet_file": [vp, C.c_char_p, C.POINTER(u32)],
        "zr_world_load_file": [vp, C.c_char_p],
        "zr_world_save_file": [vp, C.c_char_p],
        "zr_world_update_uniforms": [vp, C.c_float, C.c_float, C.c_float],
        "zr_world_save_json": [vp, vp, sz, C.POINTER(sz)],
        "zr_world_get_camera": [vp, vp],
        "zr_world_json_normalize": [C.c_char_p, sz, vp, sz, C.POINTER(sz)],
        "zr_livelink_bind_any": [vp, C.c_int],
        "zr_livelink_serve": [vp, C.c_uint16],
        "zr_livelink_port": [vp, C.POINTER(C.c_uint16)],
        "zr_livelink_poll": [vp, C.POINTER(C.c_int)],
        "zr_livelink_stop": [vp],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    L.zr_abi_version.argtypes = []
    L.zr_abi_version.restype = u32
    if L.zr_abi_version() != abi.ABI_VERSION:
        raise ImportError("libzelda_render.so speaks ABI version %d, this binding %d: rebuild (python -m zeldaengine_amd.build --force)"
                          % (L.zr_abi_version(), abi.ABI_VERSION))
    L.zr_tile_owner.argtypes = [u32, u32, u32]
    L.zr_tile_owner.restype = u32
    L.zr_destroy.argtypes = [vp]
    L.zr_destroy.restype = None
    L.zr_last_error.argtypes = [vp]
    L.zr_last_error.restype = C.c_char_p
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def build_meshlets(verts, idx, max_vertices=64, max_triangles=124, cone_weight=0.2):
    """Context-free clusteriser (host code only; works without a GPU).  Returns (meshlets, mverts, mtris, tri_order)."""
    L = lib()
    verts = np.ascontiguousarray(verts)
    assert verts.dtype == abi.XkVertex
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    nm, nmv, nmt = C.c_uint32(), C.c_size_t(), C.c_size_t()
    args = (_ptr(verts), len(verts), _ptr(idx), len(idx), max_vertices, max_triangles, cone_weight)
    rc = L.zr_meshlets_build(*args, None, C.byref(nm), None, C.byref(nmv), None, C.byref(nmt), None)
    if rc:
        raise ZeldaRenderError(rc, "zr_meshlets_build")
    ml = np.zeros(nm.value, dtype=abi.XkMeshlet)
    mv = np.zeros(nmv.value, dtype=np.uint32)
    mt = np.zeros(nmt.value, dtype=np.uint8)
    order = np.zeros(nmt.value // 3, dtype=np.uint32)
    rc = L.zr_meshlets_build(*args, _ptr(ml), C.byref(nm), _ptr(mv), C.byref(nmv), _ptr(mt), C.byref(nmt), _ptr(order))
    if rc:
        raise ZeldaRenderError(rc, "zr_meshlets_build")
    return ml, mv, mt, order


def load_obj(path):
    """LoadMeshAsset through the library's own OBJ reader (host code only) -> (XkVertex[], uint32 indices)."""
    L = lib()
    nv, ni = C.c_uint32(), C.c_uint32()
    rc = L.zr_load_obj(os.fsencode(path), None, C.byref(nv), None, C.byref(ni))
    if rc:
        raise ZeldaRenderError(rc, "zr_load_obj(%s)" % path)
    v = np.zeros(nv.value, dtype=abi.XkVertex)
    idx = np.zeros(ni.value, dtype=np.uint32)
    rc = L.zr_load_obj(os.fsencode(path), _ptr(v), C.byref(nv), _ptr(idx), C.byref(ni))
    if rc:
        raise ZeldaRenderError(rc, "zr_load_obj(%s)" % path)
    return v, idx


def load_png_rgba8(path):
    """LoadTextureAsset through the library's own PNG reader (host code only) -> (H, W, 4) uint8."""
    L = lib()
    w, h = C.c_uint32(), C.c_uint32()
    rc = L.zr_load_png_rgba8(os.fsencode(path), None, 0, C.byref(w), C.byref(h))
    if rc:
        raise ZeldaRenderError(rc, "zr_load_png_rgba8(%s)" % path)
    out = np.zeros((h.value, w.value, 4), dtype=np.uint8)
    rc = L.zr_load_png_rgba8(os.fsencode(path), _ptr(out), out.nbytes, C.byref(w), C.byref(h))
    if rc:
        raise ZeldaRenderError(rc, "zr_load_png_rgba8(%s)" % path)
    return out


def tile_partition(width, height, world, rank):
    """The library's own statement of the multi-GPU tile ownership (host code only) -> (owned tile indices, slots_per_rank)."""
    L = lib()
    n, spr = C.c_uint32(), C.c_uint32()
    rc = L.zr_tile_partition(width, height, world, rank, None, C.byref(n), C.byref(spr))
    if rc:
        raise ZeldaRenderError(rc, "zr_tile_partition")
    owned = np.zeros(n.value, dtype=np.uint32)
    L.zr_tile_partition(width, height, world, rank, _ptr(owned), C.byref(n), C.byref(spr))
    return owned, spr.value


def dist_unique_id():
    """ncclGetUniqueId through the library (128 bytes); rank 0 makes it, every rank passes it to Renderer.dist_init."""
    buf = C.create_string_buffer(128)
    rc = lib().zr_dist_unique_id(buf, 128)
    if rc:
        raise ZeldaRenderError(rc, "zr_dist_unique_id (librccl not loadable?)")
    return buf.raw


def world_json_normalize(text):
    """Context-free XkWorld Load -> Save (host code only).  Raises ZeldaRenderError(ZR_ERR_PARSE) with the loader's message."""
    L = lib()
    b = text.encode() if isinstance(text, str) else bytes(text)
    n = C.c_size_t()
    L.zr_world_json_normalize(b, len(b), None, 0, C.byref(n))
    buf = C.create_string_buffer(max(1, n.value))
    rc = L.zr_world_json_normalize(b, len(b), buf, n.value, C.byref(n))
    out = buf.raw[:n.value].decode(errors="replace")
    if rc:
        raise ZeldaRenderError(rc, out)
    return out


class Renderer:
    def __init__(self, width=1920, height=1080, shadow_dim=1024, device=0, tile_rank=0, tile_world=1, flags=0,
                 debug_view=0):
        self.L = lib()
        self.W, self.H, self.SD = width, height, shadow_dim
        self.tile_rank, self.tile_world = tile_rank, tile_world
        cfg = abi.Config(width, height, shadow_dim, debug_view, device, tile_rank, tile_world, flags)
        h = C.c_void_p()
        rc = self.L.zr_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise ZeldaRenderError(rc, "zr_create failed (no usable HIP device or bad config); there is no CPU fallback")
        self.h = h

    # ---- plumbing
    def _chk(self, rc):
        if rc != 0:
            raise ZeldaRenderError(rc, (self.L.zr_last_error(self.h) or b"").decode(errors="replace"))

    def close(self):
        if getattr(self, "h", None):
            self.L.zr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, hip_stream):
        self._chk(self.L.zr_set_stream(self.h, C.c_void_p(hip_stream)))

    # ---- scene
    def mesh_create(self, verts, idx):
        verts = np.ascontiguousarray(verts)
        assert verts.dtype == abi.XkVertex
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        mid = C.c_uint32()
        self._chk(self.L.zr_mesh_create(self.h, _ptr(verts), len(verts), _ptr(idx), len(idx), C.byref(mid)))
        return mid.value

    def mesh_set_meshlets(self, mesh, meshlets, mverts, mtris):
        meshlets = np.ascontiguousarray(meshlets)
        assert meshlets.dtype == abi.XkMeshlet
        mverts = np.ascontiguousarray(mverts, dtype=np.uint32)
        mtris = np.ascontiguousarray(mtris, dtype=np.uint8)
        self._chk(self.L.zr_mesh_set_meshlets(self.h, mesh, _ptr(meshlets), len(meshlets), _ptr(mverts), len(mverts),
                                              _ptr(mtris), len(mtris)))

    def mesh_build_meshlets(self, mesh, max_vertices=64, max_triangles=124, cone_weight=0.2):
        self._chk(self.L.zr_mesh_build_meshlets(self.h, mesh, max_vertices, max_triangles, cone_weight))

    def mesh_get_meshlets(self, mesh):
        nm, nmv, nmt = C.c_uint32(), C.c_size_t(), C.c_size_t()
        self._chk(self.L.zr_mesh_get_meshlets(self.h, mesh, None, C.byref(nm), None, C.byref(nmv), None, C.byref(nmt)))
        ml = np.zeros(nm.value, dtype=abi.XkMeshlet)
        mv = np.zeros(nmv.value, dtype=np.uint32)
        mt = np.zeros(nmt.value, dtype=np.uint8)
        self._chk(self.L.zr_mesh_get_meshlets(self.h, mesh, _ptr(ml), C.byref(nm), _ptr(mv), C.byref(nmv), _ptr(mt),
                                              C.byref(nmt)))
        return ml, mv, mt

    def object_add(self, mesh, material=None, instances=None):
        mat = C.cast(C.pointer(material), C.c_void_p) if material is not None else None
        n = 0 if instances is None else len(instances)
        inst = np.ascontiguousarray(instances) if n else None
        if n:
            assert inst.dtype == abi.XkInstanceData
        self._chk(self.L.zr_object_add(self.h, mesh, mat, _ptr(inst) if n else None, n))

    def scene_clear(self):
        self._chk(self.L.zr_scene_clear(self.h))

    def set_limits(self, record_chunks=0, slow_triangles=0):
        """Capacities of the triangle-record arrays (chunks of 256 records, `zr_record_chunk_size()`) and the clipped-triangle list; 0 = defaults."""
        self._chk(self.L.zr_set_limits(self.h, record_chunks, slow_triangles))

    def set_bucket_share(self, percent=100):
        """Plan every per-tile record bucket at `percent` of its size (zr_set_bucket_share): the rest goes the overflow route - same frame."""
        self._chk(self.L.zr_set_bucket_share(self.h, percent))

    def object_count(self):
        n = C.c_uint32()
        self._chk(self.L.zr_object_count(self.h, C.byref(n)))
        return n.value

    def object_get_instances(self, index):
        n, mesh = C.c_uint32(), C.c_uint32()
        self._chk(self.L.zr_object_get_instances(self.h, index, C.byref(mesh), None, C.byref(n)))
        inst = np.zeros(n.value, dtype=abi.XkInstanceData)
        if n.value:
            self._chk(self.L.zr_object_get_instances(self.h, index, C.byref(mesh), _ptr(inst), C.byref(n)))
        return mesh.value, (inst if n.value else None)

    def set_cubemap(self, faces):
        if faces is None:
            self._chk(self.L.zr_set_cubemap(self.h, None, 0))
            return
        faces = [np.ascontiguousarray(f, dtype=np.uint8) for f in faces]
        assert len(faces) == 6 and all(f.shape == faces[0].shape and f.shape[0] == f.shape[1] and f.shape[2] == 4 for f in faces)
        arr = (C.c_void_p * 6)(*[f.ctypes.data for f in faces])
        self._chk(self.L.zr_set_cubemap(self.h, arr, faces[0].shape[0]))

    def set_skydome(self, verts, idx, image):
        """Sky mesh + its sRGB RGBA8 image (H, W, 4); image None removes the pass (CreateSkydomePass, ZE:2690-2744)."""
        if image is None:
            self._chk(self.L.zr_set_skydome(self.h, None, 0, None, 0, None))
            return
        verts = np.ascontiguousarray(verts)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        img = np.ascontiguousarray(image, dtype=np.uint8)
        im = abi.Image(img.ctypes.data, img.shape[1], img.shape[0])
        self._chk(self.L.zr_set_skydome(self.h, _ptr(verts), len(verts), _ptr(idx), len(idx), C.cast(C.pointer(im), C.c_void_p)))

    def set_background(self, image):
        if image is None:
            self._chk(self.L.zr_set_background(self.h, None))
            return
        img = np.ascontiguousarray(image, dtype=np.uint8)
        im = abi.Image(img.ctypes.data, img.shape[1], img.shape[0])
        self._chk(self.L.zr_set_background(self.h, C.cast(C.pointer(im), C.c_void_p)))

    def set_sky_flags(self, enable_skydome=True, enable_background=True):
        self._chk(self.L.zr_set_sky_flags(self.h, int(enable_skydome), int(enable_background)))

    # ---- uniforms
    def update_uniforms(self, cam, dir_l, point_l, spot_l, roll_stage=0.0, roll_light=0.0, time=0.0):
        self._chk(self.L.zr_update_uniforms(self.h, C.cast(C.pointer(cam), C.c_void_p), _ptr(dir_l), len(dir_l),
                                            _ptr(point_l), len(point_l), _ptr(spot_l), len(spot_l),
                                            roll_stage, roll_light, time))

    def set_frame(self, cam, shadow, view):
        self._chk(self.L.zr_set_frame(self.h, cam.ctypes.data, shadow.ctypes.data, view.ctypes.data))

    def get_frame(self):
        cam = np.zeros((), dtype=abi.XkUniformBufferMVP)
        sh = np.zeros((), dtype=abi.XkUniformBufferMVP)
        view = np.zeros((), dtype=abi.XkView)
        self._chk(self.L.zr_get_frame(self.h, cam.ctypes.data, sh.ctypes.data, view.ctypes.data))
        return cam, sh, view

    def set_debug_view(self, v):
        self._chk(self.L.zr_set_debug_view(self.h, v))

    def set_shading(self, forward):
        """False: the deferred frame (default); True: the forward variant, SH/Base.frag (zr_set_shading)."""
        self._chk(self.L.zr_set_shading(self.h, 1 if forward else 0))

    # ---- frame
    def render(self, debug_view=None, passes=None):
        if debug_view is not None:
            self.set_debug_view(debug_view)
        self._chk(self.L.zr_render(self.h))

    def render_shadow(self):
        self._chk(self.L.zr_render_shadow(self.h))

    def render_gbuffer(self):
        self._chk(self.L.zr_render_gbuffer(self.h))

    def render_lighting(self):
        self._chk(self.L.zr_render_lighting(self.h))

    def render_geometry(self):
        """Shadow pass and deferred-scene pass side by side (not joined: see stream_wait_shadow)."""
        self._chk(self.L.zr_render_geometry(self.h))

    def stream_wait_shadow(self, hip_stream):
        """Make a HIP stream (its raw handle) wait for the shadow pass enqueued last."""
        self._chk(self.L.zr_stream_wait_shadow(self.h, hip_stream))

    def set_shadow_partition(self, rank, world):
        self._chk(self.L.zr_set_shadow_partition(self.h, rank, world))

    def set_shadow_tiles(self, rank, world):
        """The shadow MAP owned by light-space super-tiles: this context draws the casters that reach its tiles (exact there)."""
        self._chk(self.L.zr_set_shadow_tiles(self.h, rank, world))

    def shadow_tiles_bytes(self):
        n = C.c_size_t()
        self._chk(self.L.zr_shadow_tiles_bytes(self.h, C.byref(n)))
        return n.value

    def shadow_pack(self, packed_dev, hip_stream=None):
        self._chk(self.L.zr_shadow_pack(self.h, C.c_void_p(packed_dev), C.c_void_p(hip_stream) if hip_stream else None))

    def shadow_unpack(self, gathered_dev, hip_stream=None):
        self._chk(self.L.zr_shadow_unpack(self.h, C.c_void_p(gathered_dev), C.c_void_p(hip_stream) if hip_stream else None))

    def set_shadow_buffer(self, dev_ptr):
        self._chk(self.L.zr_set_shadow_buffer(self.h, C.c_void_p(dev_ptr) if dev_ptr else None))

    def finish(self):
        self._chk(self.L.zr_finish(self.h))

    def set_timing_interval(self, interval):
        """Record the per-pass hipEvents on every interval-th frame only (each record is a ~6 us bubble on the stream)."""
        self._chk(self.L.zr_set_timing_interval(self.h, interval))

    def pass_times(self, last_n=1):
        """Mean GPU milliseconds per pass over the last `last_n` (<= 64) timed frames."""
        ms = (C.c_float * len(abi.PASS_NAMES))()
        self._chk(self.L.zr_get_pass_times_avg(self.h, last_n, ms))
        return dict(zip(abi.PASS_NAMES, [float(x) for x in ms]))

    def frame_latencies(self, n=64):
        """Begin-to-end GPU milliseconds of the last n (<= 64) timed frames, newest first."""
        ms = (C.c_float * n)()
        got = self.L.zr_get_frame_latencies(self.h, n, ms)
        if got < 0:
            self._chk(got)
        return [float(ms[i]) for i in range(got)]

    def frame_periods(self, n=511):
        """GPU milliseconds between the ends of consecutive frames for the last n (<= 511) frames, newest first."""
        ms = (C.c_float * n)()
        got = self.L.zr_get_frame_periods(self.h, n, ms)
        if got < 0:
            self._chk(got)
        return [float(ms[i]) for i in range(got)]

    def stats(self):
        s = abi.Stats()
        self._chk(self.L.zr_get_stats(self.h, C.byref(s), C.sizeof(s)))
        return {"work_items": list(s.work_items), "survivors": list(s.survivors), "bin_entries": list(s.bin_entries),
                "covered_pixels": int(s.covered_pixels), "covered_shadow_texels": int(s.covered_shadow_texels), "overflow": int(s.overflow),
                "hiz_culled": int(s.hiz_culled), "hiz_culled_geom": int(s.hiz_culled_geom), "round1_survivors": int(s.round1_survivors),
                "shadow_occluded": int(s.shadow_occluded), "shadow_late": int(s.shadow_late)}

    # ---- read-back
    def color(self):
        out = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        self._chk(self.L.zr_read_color(self.h, _ptr(out), out.nbytes))
        return out

    def gbuffer(self, target):
        out = np.zeros((self.H, self.W), dtype=abi.GBUFFER_DTYPES[target])
        self._chk(self.L.zr_read_gbuffer(self.h, target, _ptr(out), out.nbytes))
        return out

    def shadowmap(self):
        out = np.zeros((self.SD, self.SD), dtype=np.float32)
        self._chk(self.L.zr_read_shadowmap(self.h, _ptr(out), out.nbytes))
        return out

    def copy_frame_async(self, color_dev=None, shadow_dev=None):
        """The frame enqueued last -> caller-owned device buffers (addresses), in stream order, no host synchronisation."""
        self._chk(self.L.zr_copy_frame_async(self.h, C.c_void_p(color_dev) if color_dev else None, C.c_void_p(shadow_dev) if shadow_dev else None))

    # ---- multi-GPU
    def tiles_device_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.L.zr_tiles_device_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_tiles_buffer(self, dev_ptr):
        self._chk(self.L.zr_set_tiles_buffer(self.h, C.c_void_p(dev_ptr) if dev_ptr else None))

    def read_tiles(self):
        _, nbytes = self.tiles_device_buffer()
        out = np.zeros((nbytes // (abi.TILE * abi.TILE * 4), abi.TILE, abi.TILE, 4), dtype=np.uint8)
        self._chk(self.L.zr_read_tiles(self.h, _ptr(out), out.nbytes))
        return out

    def composite(self, gathered_dev_ptr):
        self._chk(self.L.zr_composite(self.h, C.c_void_p(gathered_dev_ptr)))

    def color_device_ptr(self):
        p = C.c_void_p()
        self._chk(self.L.zr_color_device_ptr(self.h, C.byref(p)))
        return p.value

    def dist_init(self, unique_id, rank, world, split_shadow=False):
        """Native multi-GPU host: the library calls RCCL itself (zr_dist_frame = render + all-gather + composite).
        split_shadow: False / "replicated", True / "split", or "tiles" (abi.SHADOW_MODES)."""
        self._dist_id = bytes(unique_id)
        self._chk(self.L.zr_dist_init(self.h, self._dist_id, len(self._dist_id), rank, world, abi.dist_flags(abi.shadow_mode(split_shadow))))

    def dist_prepare(self, rank, world, split_shadow=False):
        """The local half of dist_init (librccl, collective stream, buffers): safe to fail on one rank alone."""
        self._chk(self.L.zr_dist_prepare(self.h, rank, world, abi.dist_flags(abi.shadow_mode(split_shadow))))

    def dist_connect(self, unique_id):
        """ncclCommInitRank: a collective - call it only when every rank's dist_prepare succeeded."""
        self._dist_id = bytes(unique_id)
        self._chk(self.L.zr_dist_connect(self.h, self._dist_id, len(self._dist_id)))

    def dist_frame(self):
        self._chk(self.L.zr_dist_frame(self.h))

    def dist_copy_frame_async(self, color_dev_ptr):
        """zr_dist_copy_frame_async: the last enqueued frame's composite into a caller-owned device buffer, in collective-stream order."""
        self._chk(self.L.zr_dist_copy_frame_async(self.h, C.c_void_p(color_dev_ptr)))

    # ---- world / livelink
    def profab_register(self, name, mesh, material=None):
        mat = C.cast(C.pointer(material), C.c_void_p) if material is not None else None
        self._chk(self.L.zr_profab_register(self.h, name.encode(), mesh, mat))

    def world_load_json(self, text):
        b = text.encode() if isinstance(text, str) else bytes(text)
        self._chk(self.L.zr_world_load_json(self.h, b, len(b)))

    def set_asset_root(self, path):
        """The engine's working directory (Profabs/, Content/): world loads then resolve Profabs and sky / cubemap / background files."""
        self._chk(self.L.zr_set_asset_root(self.h, os.fsencode(path) if path is not None else None))

    def asset_path_search(self, name):
        n = C.c_size_t()
        self._chk(self.L.zr_asset_path_search(self.h, name.encode(), None, 0, C.byref(n)))
        buf = C.create_string_buffer(max(1, n.value))
        self._chk(self.L.zr_asset_path_search(self.h, name.encode(), buf, n.value, C.byref(n)))
        return buf.raw[:n.value].decode()

    def load_meshlet_file(self, path):
        m = C.c_uint32()
        self._chk(self.L.zr_load_meshlet_file(self.h, os.fsencode(path), C.byref(m)))
        return m.value

    def world_load_file(self, path=None):
        self._chk(self.L.zr_world_load_file(self.h, os.fsencode(path) if path is not None else None))

    def world_save_file(self, path=None):
        self._chk(self.L.zr_world_save_file(self.h, os.fsencode(path) if path is not None else None))

    def world_update_uniforms(self, roll_stage=0.0, roll_light=0.0, time=0.0):
        self._chk(self.L.zr_world_update_uniforms(self.h, roll_stage, roll_light, time))

    def world_save_json(self):
        n = C.c_size_t()
        self._chk(self.L.zr_world_save_json(self.h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        self._chk(self.L.zr_world_save_json(self.h, buf, n.value, C.byref(n)))
        return buf.raw[:n.value].decode()

    def world_camera(self):
        cam = abi.Camera()
        self._chk(self.L.zr_world_get_camera(self.h, C.cast(C.pointer(cam), C.c_void_p)))
        return cam

    def livelink_serve(self, port=8080, bind_any=False):
        self._chk(self.L.zr_livelink_bind_any(self.h, int(bind_any)))
        self._chk(self.L.zr_livelink_serve(self.h, port))
        p = C.c_uint16()
        self._chk(self.L.zr_livelink_port(self.h, C.byref(p)))
        return p.value

    def livelink_poll(self):
        r = C.c_int()
        self._chk(self.L.zr_livelink_poll(self.h, C.byref(r)))
        return bool(r.value)

    def livelink_stop(self):
        self._chk(self.L.zr_livelink_stop(self.h))


def load_scene(r, cfg):
    """Feed a scenes.config*() dict to a Renderer."""
    r.set_cubemap(cfg.get("cubemap"))
    for o in cfg["objects"]:
        m = r.mesh_create(*o["mesh"])
        r.object_add(m, o.get("material"), o.get("instances"))
    r.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"])
