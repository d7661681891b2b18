"""Frame rate of the textured variant of the benchmark scene (seven 512^2 material images) under the current environment.
Usage: [ENV=..] python tools/textured_ab.py [flags]"""
import sys, time
sys.path.insert(0, '/root/repo')
from zeldaengine_amd import engine as gpu_engine, scenes, abi
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = scenes.config3(10000, cube_dim=1024, textured=True)
g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024, flags=flags)
gpu_engine.load_scene(g, cfg)
upd = len(sys.argv) > 2 and sys.argv[2] == "upd"
def step(i):
    if upd:
        g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)
    g.render()
for i in range(10): step(i)
g.finish()
t = time.perf_counter()
for i in range(60): step(10 + i)
g.finish()
dt = (time.perf_counter() - t) / 60
print("flags %d: %.4f ms/frame, %.0f Mpix/s" % (flags, dt * 1e3, cfg["width"] * cfg["height"] / dt / 1e6))
g.set_timing_interval(4)
for i in range(40): step(70 + i)
g.finish()
print("   ", {k: round(v * 1e3, 1) for k, v in g.pass_times(8).items()})
g.close()
