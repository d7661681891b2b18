"""GPU period of every frame from the very first one after scene load (does the start of a run differ from its steady state?).
Usage: python tools/early_frames.py [frames] [idle_ms between scene load and the first frame] [update: 1 = uniforms rebuilt every frame]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from zeldaengine_amd import engine as gpu_engine, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
upd = len(sys.argv) > 3 and sys.argv[3] == "1"
cfg = scenes.config3(10000, cube_dim=1024)
g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024)
gpu_engine.load_scene(g, cfg)
time.sleep(idle * 1e-3)
host = []
for i in range(n):
    t = time.perf_counter()
    if upd:
        g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)
    g.render()
    host.append((time.perf_counter() - t) * 1e6)
g.finish()
p = g.frame_periods(n - 1)[::-1]
print("GPU periods us (frame 2..):", " ".join("%.0f" % (x * 1e3) for x in p))
print("host enqueue us (frame 1..):", " ".join("%.0f" % x for x in host))
# second batch on the same context after a pause: is it the context's age or the GPU's idleness?
time.sleep(0.2)
for i in range(n):
    g.render()
g.finish()
p = g.frame_periods(n - 1)[::-1]
print("after 200 ms idle, GPU periods us:", " ".join("%.0f" % (x * 1e3) for x in p))
g.close()
