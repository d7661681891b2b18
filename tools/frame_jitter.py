"""Where does a timed loop lose time?  Per-frame host enqueue time and GPU frame periods of bench.py's main loop, outliers listed."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zeldaengine_amd import dist as zdist, engine, scenes
cfg = scenes.config3(cube_dim=int(os.environ.get("CUBE", "1024")))
dr = zdist.DistributedRenderer(cfg["width"], cfg["height"], 1024, device_index=0, rank=0, world=1)
engine.load_scene(dr.r, cfg)
dr.r.set_timing_interval(int(os.environ.get("INTERVAL", "8")))
cam, d, p, s = cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"]
for rep in range(3):
    for i in range(10):
        dr.r.update_uniforms(cam, d, p, s, 0.0, 0.002 * i, 0.0); dr.frame()
    dr.synchronize(); torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for i in range(100):
        a = time.perf_counter()
        dr.r.update_uniforms(cam, d, p, s, 0.0, 0.002 * (10 + i), 0.0)
        b = time.perf_counter()
        dr.frame()
        host.append((b - a, time.perf_counter() - b))
    t1 = time.perf_counter()
    dr.synchronize()
    t2 = time.perf_counter()
    per = dr.r.frame_periods(99)[::-1]
    print("rep %d: enqueue %.1f us/frame, total %.1f us/frame; uniforms mean %.1f us, frame() mean %.1f us" % (
        rep, (t1 - t0) / 100 * 1e6, (t2 - t0) / 100 * 1e6, sum(h[0] for h in host) / 100 * 1e6, sum(h[1] for h in host) / 100 * 1e6))
    print("   host outliers (>300 us):", [(i, round(h[0] * 1e6), round(h[1] * 1e6)) for i, h in enumerate(host) if h[0] + h[1] > 300e-6])
    print("   GPU period outliers (>0.7 ms):", [(i + 1, round(x, 3)) for i, x in enumerate(per) if x > 0.7], "sum %.2f ms" % sum(per))
