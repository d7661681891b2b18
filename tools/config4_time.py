"""Frame time and culling counters of config 4 (1 M instances, 3840x2160, 16 lights) / config 5 (256 lights): python tools/config4_time.py [n_point]"""
import sys, time
sys.path.insert(0, '/root/repo')
from zeldaengine_amd import engine as gpu_engine, scenes
n_point = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = scenes.config4(1000000, n_point)
g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024)
gpu_engine.load_scene(g, cfg)
for i in range(5): g.render()
g.finish()
t = time.perf_counter()
for i in range(30): g.render()
g.finish()
dt = (time.perf_counter() - t) / 30
st = g.stats()
print("config4 lights %d: %.3f ms/frame" % (n_point, dt * 1e3), {k: st[k] for k in ("survivors", "bin_entries", "hiz_culled", "round1_survivors", "covered_pixels", "covered_shadow_texels", "overflow")})
g.close()
