"""Regenerates tests/golden/oracle_config3_40_96x64.sha256 (a regression pin of the oracle's own arithmetic)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from zeldaengine_amd import scenes  # noqa: E402

cfg = scenes.config3(40, 96, 64)
o = pyoracle.Oracle(96, 64, 128)
pyoracle.load_scene(o, cfg)
o.render()
h = hashlib.sha256(o.color().tobytes() + o.gbuffer(2).tobytes() + o.shadowmap().tobytes()).hexdigest()
open(os.path.join(ROOT, "tests", "golden", "oracle_config3_40_96x64.sha256"), "w").write(h + "\n")
print(h)
