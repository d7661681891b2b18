"""Where the shadow pass's occlusion culling starts to pay: the benchmark's spheres at growing counts, forced on against off."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from zeldaengine_amd import engine, scenes, abi
def run(n, flags, frames=60):
    cfg = scenes.config3(n, cube_dim=64)
    g = engine.Renderer(cfg["width"], cfg["height"], 1024, flags=flags)
    engine.load_scene(g, cfg); g.set_timing_interval(0)
    def loop(k, b):
        for i in range(k):
            g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * (b + i), 0.016 * (b + i)); g.render()
        g.finish()
    loop(30, 0)
    t0 = time.perf_counter(); loop(frames, 30); dt = (time.perf_counter() - t0) / frames * 1e3
    st = g.stats(); g.close()
    return dt, st
for n in (10000, 20000, 30000, 50000):      # (100 000 of these spheres outgrow the default record pool on a scene's first frame: zr_set_limits)
    a, sa = run(n, abi.FLAG_SHADOW_OCCLUSION); b, sb = run(n, abi.FLAG_NO_SHADOW_OCCLUSION)
    print("instances %6d  work items per texel %.2f  on %.4f ms  off %.4f ms  (%+.1f %%)  hidden %d of %d" % (n, sa["work_items"][0] / 1048576.0, a, b, (b / a - 1) * 100, sa["shadow_occluded"], sa["survivors"][0]), flush=True)
