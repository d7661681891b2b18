# usage: bash tools/ab_bench.sh lib1.so lib2.so ...   (A/B variants built by zeldaengine_amd.build with out=...)
for lib in "$@"; do
  ZELDA_RENDER_LIB=$PWD/zeldaengine_amd/$lib timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ab_$lib.log 2>&1 || { echo "$lib FAILED"; tail -5 gpurun_out/ab_$lib.log; exit 1; }
  python - "$lib" <<PY
import json, sys
l=[x for x in open("gpurun_out/ab_%s.log" % sys.argv[1]) if x.startswith("{")][-1]
d=json.loads(l); print(sys.argv[1], d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in d["passes_ms"].items()}, d["stats"].get("bin_entries"))
PY
done
