"""Ad-hoc GPU check (run on the GPU box through gpurun): parity stats + pass timings, written to gpurun_out/."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle  # noqa: E402
from parity_util import compare_all, render_both  # noqa: E402
from zeldaengine_amd import engine, scenes  # noqa: E402

out = {}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for name, cfg in [("config2", scenes.config2()), ("config3_400", scenes.config3(400, 640, 360))]:
    t = time.time()
    o, g = render_both(pyoracle, engine, cfg)
    d = compare_all(o, g)
    out[name] = {"diffs": d, "stats": g.stats(), "times": g.pass_times(), "wall": time.time() - t}
    print(name, json.dumps(out[name]), flush=True)
    try:
        from PIL import Image
        Image.fromarray(g.color()).save(os.path.join(ROOT, "gpurun_out", name + "_gpu.png"))
        Image.fromarray(o.color()).save(os.path.join(ROOT, "gpurun_out", name + "_cpu.png"))
    except Exception as e:  # noqa: BLE001
        print("png:", e)

# full config 3 timing (no oracle)
cfg = scenes.config3()
g = engine.Renderer(cfg["width"], cfg["height"])
engine.load_scene(g, cfg)
for i in range(5):
    g.render()
g.finish()
times = []
for i in range(10):
    g.render()
    times.append(g.pass_times())
out["config3_full"] = {"stats": g.stats(), "times": times[-1],
                       "total_ms_median": float(np.median([t["total"] for t in times]))}
print("config3_full", json.dumps(out["config3_full"]), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gpu_check.json"), "w"), indent=1)
