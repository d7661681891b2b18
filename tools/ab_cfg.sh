# usage: bash tools/ab_cfg.sh "<bench args, e.g. --config 4 --steps 40 --warmup 8>" lib1.so lib2.so ...   (A/B variants in zeldaengine_amd/, as tools/ab_bench.sh)
args=$1; shift
for lib in "$@"; do
  ZELDA_RENDER_LIB=$PWD/zeldaengine_amd/$lib timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-extras > gpurun_out/abc_$lib.log 2>&1 || { echo "$lib FAILED"; tail -5 gpurun_out/abc_$lib.log; exit 1; }
  python - "$lib" <<PY
import json, sys
l=[x for x in open("gpurun_out/abc_%s.log" % sys.argv[1]) if x.startswith("{")][-1]
d=json.loads(l); print("%-22s %9.1f Mpx/s %.4f ms  gpu period %s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("frame_gpu_ms")))
PY
done
