# usage (GPU box): bash tools/diag_attrib.sh   -> one-stream pass times with the rasteriser's pixel walk (1) / triangle phase (2) skipped
# (a -DZR_DIAG build: the product library has no such switches)
python - <<'PY'
from zeldaengine_amd import build
print(build.build(out=build.HERE + "/libzr_diag.so", extra_flags=["-DZR_DIAG"]))
PY
for sk in 0 1 2; do
  ZR_DEBUG_SKIP=$sk ZELDA_RENDER_LIB=$PWD/zeldaengine_amd/libzr_diag.so timeout -k 10 100 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/diag_$sk.json 2>/dev/null
  python - $sk <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/diag_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
p = d["one_stream"]["passes_ms"]
print("skip", sys.argv[1], {k: round(p[k] * 1e3, 1) for k in ("cull_shadow", "shadow", "cull_camera", "gbuffer", "hiz", "gbuffer2", "resolve", "lighting")})
PY
done
