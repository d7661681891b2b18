"""A hunt with tests/test_gpu_fuzz.py's scenes at LARGE target and shadow-map sizes (overflows are reported with the statistics, mismatches with the targets): python tests/hunt_fuzz_large.py <first seed> <last seed>"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import test_gpu_fuzz as t
from parity_util import compare_all
from oracle import pyoracle
from zeldaengine_amd import abi, engine
bad = 0
sizes = [(1280, 720), (1920, 1080), (1000, 1000), (2560, 601)]
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    sc = t._scene(900000 + seed)
    sc["W"], sc["H"] = sizes[seed % len(sizes)]; sc["SD"] = [256, 512, 1024, 2048][(seed // 4) % 4]
    g = engine.Renderer(sc["W"], sc["H"], sc["SD"], flags=sc["flags"])
    t._build(g, sc)
    d, p, s = sc["lights"]
    try:
        # two frames, the lights rolled between them: the second runs on the first one's visibility history and shadow-occlusion flags
        for frame in range(2):
            g.update_uniforms(abi.make_camera(**sc["cam"]), d, p, s, sc["roll"][0], sc["roll"][1] + 0.05 * (frame - 1), 1.0); t._raw_frame(g, sc)
            g.render(sc["view"]); g.finish()
    except engine.ZeldaRenderError as e:
        print("seed", seed, sc["W"], sc["H"], sc["SD"], "flags", sc["flags"], list(sc["extra"]), [(dd[0], None if dd[2] is None else len(dd[2])) for dd in sc["draws"]], str(e)[:60], flush=True)
        try: print("   stats", g.stats(), flush=True)
        except Exception as e2: print("   stats failed", e2)
        g.close(); continue
    o = pyoracle.Oracle(sc["W"], sc["H"], sc["SD"]); o.set_threads(16)
    t._build(o, sc)
    o.update_uniforms(abi.make_camera(**sc["cam"]), d, p, s, sc["roll"][0], sc["roll"][1], 1.0); t._raw_frame(o, sc)
    o.render(sc["view"])
    diff = {k: v for k, v in compare_all(o, g).items() if v}
    if diff:
        bad += 1; print("seed", seed, sc["W"], sc["H"], sc["SD"], sc["flags"], sc["view"], diff, flush=True)
    g.close(); o.close()
print("done", lo, hi, "bad", bad)
