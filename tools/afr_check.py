"""Experiment: two whole-frame pipelines side by side (alternate frames on two contexts, each on one stream of its own) against the library's two-lane frame."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from zeldaengine_amd import engine, scenes, abi
cfg = scenes.config3(10000, cube_dim=64)
def mk(flags):
    g = engine.Renderer(cfg["width"], cfg["height"], 1024, flags=flags)
    engine.load_scene(g, cfg); g.set_timing_interval(0)
    return g
def loop(ctxs, frames, warm=20):
    import gc
    for i in range(warm):
        g = ctxs[i % len(ctxs)]
        g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i); g.render()
    for g in ctxs: g.finish()
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for i in range(frames):
        g = ctxs[i % len(ctxs)]
        g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * (warm + i), 0.016 * (warm + i)); g.render()
    for g in ctxs: g.finish()
    dt = (time.perf_counter() - t0) / frames * 1e3
    gc.enable()
    return dt
ctxs = [mk(abi.FLAG_SERIAL_PASSES) for _ in range(6)]
for n in (1, 2, 3, 4, 5, 6, 3, 4):
    print("%d one-stream contexts, alternate frames: %.4f ms" % (n, loop(ctxs[:n], 300)), flush=True)
