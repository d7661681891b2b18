"""Pass times of ONE rank's context of an N-rank job, alone on the GPU (where does a rank's frame go when N grows?).
Usage: python tools/rank_passes.py [world] [rank] [config] [serial|lanes] [replicated|tiles|split]"""
import sys, time
sys.path.insert(0, '/root/repo')
from zeldaengine_amd import engine, scenes, abi
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
config = int(sys.argv[3]) if len(sys.argv) > 3 else 3
serial = len(sys.argv) > 4 and sys.argv[4] == "serial"
cfg = scenes.config3(10000, cube_dim=1024) if config == 3 else scenes.config4(1000000, 256 if config == 5 else 16, cube_dim=1024)
g = engine.Renderer(cfg["width"], cfg["height"], 1024, tile_rank=rank, tile_world=world, flags=abi.FLAG_SERIAL_PASSES if serial else 0)
engine.load_scene(g, cfg)
mode = sys.argv[5] if len(sys.argv) > 5 else "split"
if mode == "split": g.set_shadow_partition(rank, world)
if mode == "tiles" and world > 1: g.set_shadow_tiles(rank, world)
def step(i):
    g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)
    g.render()
for i in range(10): step(i)
g.finish()
t = time.perf_counter()
for i in range(60): step(10 + i)
g.finish()
dt = (time.perf_counter() - t) / 60
import numpy as np
per = np.asarray(g.frame_periods(59))
host = []
for i in range(40):
    t0 = time.perf_counter(); step(70 + i); host.append(time.perf_counter() - t0)
g.finish()
print(mode + " world %d rank %d config %d%s: %.4f ms/frame wall, GPU period median %.4f ms, host call median %.1f us (blocks when two frames are in flight)"
      % (world, rank, config, " serial" if serial else "", dt * 1e3, float(np.median(per)), float(np.median(host)) * 1e6))
g.set_timing_interval(4)
for i in range(40): step(70 + i)
g.finish()
print("   passes us:", {k: round(v * 1e3, 1) for k, v in g.pass_times(8).items()})
print("   stats:", g.stats())
g.close()
