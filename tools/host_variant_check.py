"""GPU frame periods of loops of n frames after a warm-up, and every period above 0.8 ms with its index: python tools/host_variant_check.py <warmup> <n> [<n> ...] [gc]
(How a 37 ms hole at frame ~160 of any loop that long was found: CPython's generational collector running a full collection over torch's
millions of objects while the GPU drains.  Not the library's: `gc` as the last argument disables the collector and the hole is gone;
bench.py's timed loops run with the collector off.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zeldaengine_amd import engine, scenes
if sys.argv[-1] == "gc":
    import gc
    sys.argv.pop(); gc.collect(); gc.disable()
cfg = scenes.config3(10000, cube_dim=64)
g = engine.Renderer(cfg["width"], cfg["height"], 1024)
engine.load_scene(g, cfg)
g.set_timing_interval(0)
k = [0]
def step():
    g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * k[0], 0.016 * k[0]); k[0] += 1
    g.render()
for i in range(int(sys.argv[1])): step()
g.finish()
for n in [int(x) for x in sys.argv[2:]]:
    w0 = time.perf_counter()
    for i in range(n): step()
    g.finish()
    w2 = time.perf_counter()
    per = np.asarray(g.frame_periods(min(n - 1, 500)))[::-1] * 1e3
    big = [(int(i), round(float(x))) for i, x in enumerate(per) if x > 800]
    print("   periods > 0.8 ms (index in this loop, us):", big[:12])
    print("warmup %s n %4d (frames %d..): total %.1f us/frame; GPU periods: first 10 %s ... median %.0f last 5 %s" % (sys.argv[1], n, k[0] - n, (w2 - w0) / n * 1e6,
          " ".join("%.0f" % x for x in per[:10]), np.median(per), " ".join("%.0f" % x for x in per[-5:])))
g.close()
