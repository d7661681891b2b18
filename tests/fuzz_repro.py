"""Replay one seed of tests/test_gpu_fuzz.py and list the pixels whose depth differs from the oracle: python tests/fuzz_repro.py <seed> [flags]"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import test_gpu_fuzz as t
from oracle import pyoracle
from zeldaengine_amd import abi, engine
seed = int(sys.argv[1])
flags = int(sys.argv[2]) if len(sys.argv) > 2 else None
sc = t._scene(t.BASE + seed)
if flags is not None: sc["flags"] = flags
print({k: sc[k] for k in ("W", "H", "SD", "sky", "bg", "cam", "flags", "view")}, [(d[0], d[1] is not None, None if d[2] is None else len(d[2]), d[3]) for d in sc["draws"]])
o = pyoracle.Oracle(sc["W"], sc["H"], sc["SD"]); g = engine.Renderer(sc["W"], sc["H"], sc["SD"], flags=sc["flags"])
for r in (o, g): t._build(r, sc)
d, p, s = sc["lights"]
for r in (o, g):
    r.update_uniforms(abi.make_camera(**sc["cam"]), d, p, s, sc["roll"][0], sc["roll"][1], 1.0); t._raw_frame(r, sc)
o.render(sc["view"]); g.render(sc["view"]); g.finish()
do, dg = o.gbuffer(0), g.gbuffer(0)
bad = np.argwhere(do.view(np.uint32) != dg.view(np.uint32))
print("differing depth pixels:", len(bad))
vis = o.visibility()
for y, x in bad[:16]:
    print((int(x), int(y)), "oracle depth %.9g prim %d | gpu depth %.9g" % (do[y, x], int(vis[y, x]), dg[y, x]), "gA", hex(int(o.gbuffer(2)[y, x])), hex(int(g.gbuffer(2)[y, x])))
print(g.stats())
