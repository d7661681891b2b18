# usage: bash tools/ab_serial.sh lib1.so lib2.so ...   one-stream + two-lane pass times of library variants (zeldaengine_amd/<lib>)
for lib in "$@"; do
  ZELDA_RENDER_LIB=$PWD/zeldaengine_amd/$lib timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ab_$lib.json 2>/dev/null || { echo "$lib FAILED"; continue; }
  python - "$lib" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
f = lambda p: {k: round(v * 1e3, 1) for k, v in p.items() if k not in ("composite",)}
print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "moving", d.get("value_moving_camera"))
print("   two-lane  ", f(d["passes_ms"]))
print("   one-stream", d["one_stream"]["ms_per_step"], f(d["one_stream"]["passes_ms"]))
PY
done
