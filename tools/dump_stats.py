"""Diagnostics (a -DZR_DIAG build with ZR_DUMP_STATS=1 prints the raw device block per zr_finish): config 3 with the camera orbiting 2 degrees per
frame - how many records miss their tile's bucket (overflow records) under motion.   usage: python tools/dump_stats.py [frames] [4 = config 4]"""
import math
import sys
sys.path.insert(0, ".")
from zeldaengine_amd import abi, engine, scenes
cfg = scenes.config4(1000000, 16, cube_dim=64) if (len(sys.argv) > 2 and sys.argv[2] == "4") else scenes.config3()
g = engine.Renderer(cfg["width"], cfg["height"], 1024)
engine.load_scene(g, cfg)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    a = math.radians(45.0 + 2.0 * i)
    cam = abi.make_camera((math.sqrt(50.0) * math.cos(a), math.sqrt(50.0) * math.sin(a), 5.0), (0.0, 0.0, 0.0))
    g.update_uniforms(cam, cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)
    g.render()
    try:
        g.finish()
    except Exception as e:
        print("frame", i, "ERROR", e)
    print(i, g.stats()["bin_entries"])
