import sys, time
sys.path.insert(0, "/root/repo")
import torch
from zeldaengine_amd import dist as zdist, engine, scenes
cfg = scenes.config3()
dr = zdist.DistributedRenderer(cfg["width"], cfg["height"], 1024, device_index=0, rank=0, world=1)
engine.load_scene(dr.r, cfg)
dr.r.set_timing_interval(0)
for _ in range(20): dr.frame()
dr.synchronize()
for n in (50, 200):
    t0 = time.perf_counter()
    for _ in range(n): dr.frame()
    t1 = time.perf_counter()
    dr.synchronize()
    t2 = time.perf_counter()
    print("n=%d enqueue %.1f us/frame, total %.1f us/frame" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
# the same with the uniforms updated every frame (pinned upload ring in use)
cam, d, p, s = cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"]
for n in (50, 50, 200):
    t0 = time.perf_counter()
    for i in range(n):
        dr.r.update_uniforms(cam, d, p, s, 0.001 * i, 0.0, 1.0)
        dr.frame()
    t1 = time.perf_counter()
    dr.synchronize()
    t2 = time.perf_counter()
    print("updating uniforms: n=%d enqueue %.1f us/frame, total %.1f us/frame" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
