"""GPU period of every frame from the very first one after scene load (does the start of a run differ from its steady state?)."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from zeldaengine_amd import engine as gpu_engine, scenes
cfg = scenes.config3(10000, cube_dim=1024)
g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024)
gpu_engine.load_scene(g, cfg)
n = 60
for i in range(n):
    g.render()
g.finish()
p = g.frame_periods(n - 1)[::-1]
print("periods us (frame 1..):", " ".join("%.0f" % (x * 1e3) for x in p))
g.close()
