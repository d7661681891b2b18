# usage: bash tools/ab_env2.sh lib.so "VAR1=a VAR2=b" "VAR1=c" ...   bench lines of one library under several environments
lib=$1; shift
for e in "$@"; do
  env $e ZELDA_RENDER_LIB=$PWD/zeldaengine_amd/$lib timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/ab_env.json 2>/dev/null || { echo "$e FAILED"; continue; }
  python - "$e" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_env.json") if l.startswith("{")][-1])
print("%-50s value %8.1f ms %.4f  %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["passes_ms"].items() if k != "composite"}))
PY
done
