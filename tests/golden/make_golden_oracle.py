"""Regenerates the oracle regression pins under tests/golden/ (SHA-256 of small oracle frames)."""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import pyoracle  # noqa: E402
from zeldaengine_amd import abi, scenes  # noqa: E402
import oracle_cases  # noqa: E402

cfg = scenes.config3(40, 96, 64)
o = pyoracle.Oracle(96, 64, 128)
pyoracle.load_scene(o, cfg)
o.render()
h = hashlib.sha256(o.color().tobytes() + o.gbuffer(2).tobytes() + o.shadowmap().tobytes()).hexdigest()
open(os.path.join(HERE, "oracle_config3_40_96x64.sha256"), "w").write(h + "\n")
print("config3_40_96x64", h)
for name, fn in oracle_cases.CASES.items():
    h, _keep = fn(pyoracle, abi, scenes)
    open(os.path.join(HERE, "oracle_%s.sha256" % name), "w").write(h + "\n")
    print(name, h)
