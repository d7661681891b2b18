"""Regenerates tests/golden/xkworld_untitled.json from the reference's own Python livelink script.

Runs only in the authoring container (needs /root/reference).  It imports
Engine/ZeldaPython/ZeldaUntitled.py (stdlib only; the module body builds the `xkWorld`
dict, the socket send is commented out at :162-164) and stores the exact bytes
`json.dumps(xkWorld)` would put on the wire (ZeldaUntitled.py:19,163).
Expected: 5727 bytes, sha256 533e3ec97cf90d66037de4799d911a0d65ed68f303b190c94731ccb68efd7f6e.
"""
import hashlib
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/Engine/ZeldaPython")
import ZeldaUntitled  # noqa: E402

payload = json.dumps(ZeldaUntitled.xkWorld).encode()
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xkworld_untitled.json")
with open(out, "wb") as f:
    f.write(payload)
print(len(payload), hashlib.sha256(payload).hexdigest())
