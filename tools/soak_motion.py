"""Soak: config 4 (1 M instances, 3840x2160) for N frames with the camera AND the shadow-casting light moving every frame, two contexts
side by side - every cull on / every cull off - compared bit for bit every K-th frame.  python tools/soak_motion.py [frames] [every] [point lights: 16 = config 4, 256 = config 5]"""
import sys, math, time
sys.path.insert(0, '.')
import numpy as np
from zeldaengine_amd import engine, scenes, abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
K = int(sys.argv[2]) if len(sys.argv) > 2 else 25
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cfg = scenes.config4(1000000, NP, cube_dim=64)
on = engine.Renderer(cfg["width"], cfg["height"], 1024)
off = engine.Renderer(cfg["width"], cfg["height"], 1024, flags=abi.FLAG_NO_FRUSTUM_CULL | abi.FLAG_NO_CONE_CULL | abi.FLAG_NO_HIZ | abi.FLAG_NO_SHADOW_OCCLUSION | abi.FLAG_NO_LIST_REUSE)
off.set_limits(record_chunks=1 << 20)      # without the culls one round draws everything: up to ~200 M records at some camera positions
for g in (on, off):
    engine.load_scene(g, cfg)
p0 = np.array(cfg["dir"]["Position"][0][:3], dtype=np.float64)
rad, a0 = math.hypot(p0[0], p0[1]), math.atan2(p0[1], p0[0])
bad = 0; t0 = time.time()
for i in range(N):
    d = cfg["dir"].copy()
    a = a0 + math.radians(0.7 * i)
    d["Position"][0][:3] = (rad * math.cos(a), rad * math.sin(a), p0[2] + 2.0 * math.sin(0.05 * i)); d["Direction"][0][:3] = d["Position"][0][:3]
    ca = math.radians(45.0 + 1.3 * i)
    cam = abi.make_camera((math.sqrt(50.0) * math.cos(ca), math.sqrt(50.0) * math.sin(ca), 5.0 + math.sin(0.03 * i)), (0.0, 0.0, 0.0))
    check = i % K == K - 1 or i == N - 1
    for g in ((on, off) if check else (on,)):
        g.update_uniforms(cam, d, cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)
        g.render()
    if check:
        # the culls-off context has skipped frames: its frame depends on the uniforms alone, which is the point
        try:
            on.finish(); off.finish()
        except engine.ZeldaRenderError as e:
            import ctypes as C
            for name, g in (("on", on), ("off", off)):
                st = abi.Stats(); rc = g.L.zr_get_stats(g.h, C.byref(st))
                print(name, "rc", rc, "work", list(st.work_items), "survivors", list(st.survivors), "bins", list(st.bin_entries), "occluded", st.shadow_occluded, "late", st.shadow_late, "overflow", st.overflow, flush=True)
            raise
        same = np.array_equal(on.shadowmap().view(np.uint32), off.shadowmap().view(np.uint32)) and np.array_equal(on.color(), off.color())
        st = on.stats()
        print("frame %4d  %s  occluded %d late %d hiz_culled %d overflow %d  (%.0f s)" % (i, "same" if same else "DIFFERENT", st["shadow_occluded"], st["shadow_late"], st["hiz_culled"], st["overflow"], time.time() - t0), flush=True)
        bad += 0 if same and st["overflow"] == 0 else 1
print("done: %d frames, %d bad checks" % (N, bad))
sys.exit(1 if bad else 0)
