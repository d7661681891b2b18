# usage: bash tools/ab_env.sh VAR v1 v2 ...   two-lane + one-stream bench lines with VAR set to each value ("-" = unset)
var=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/abe.json 2>/dev/null || { echo "$var=$v FAILED"; continue; }
  python - "$var=$v" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/abe.json") if l.startswith("{")][-1])
f = lambda p: {k: round(v * 1e3, 1) for k, v in p.items() if k not in ("composite",)}
print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "one-stream", (d.get("one_stream") or {}).get("ms_per_step"))
if d.get("passes_ms"): print("   two-lane  ", f(d["passes_ms"]))
PY
done
