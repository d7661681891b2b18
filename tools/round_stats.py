"""Per-round counters of the camera pass on the benchmark scene, both rasterisers (run with ZR_DUMP_STATS=1 for the raw device block)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeldaengine_amd import engine as gpu_engine, scenes, abi
cfg = scenes.config3(10000, cube_dim=64)
for flags in (0, abi.FLAG_MESHLET_BINS):
    try:
        g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024, flags=flags)
    except gpu_engine.ZeldaRenderError:      # the A/B rasteriser: -DZR_DIAG libraries only (ZELDA_RENDER_LIB=...)
        continue
    gpu_engine.load_scene(g, cfg)
    for i in range(5): g.render()
    g.finish()
    print(flags, g.stats())
    g.close()
