# usage (GPU box): bash tools/diag_light.sh   -> lighting pass alone (one stream) + its VALU count with PCF (1) / lights (2) / reflection (4) skipped
# (a -DZR_DIAG build: the product library has no such switches)
python - <<'PY'
from zeldaengine_amd import build
print(build.build(out=build.HERE + "/libzr_diag.so", extra_flags=["-DZR_DIAG"]))
PY
cd /tmp && export TMPDIR=/tmp
for sk in 0 1 2 4 7; do
  R=$GRAFT_REPO_ROOT
  rm -rf $R/gpurun_out/dl
  ZR_DEBUG_SKIP_LIGHT=$sk ZELDA_RENDER_LIB=$R/zeldaengine_amd/libzr_diag.so timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/dl -o d -- python3 $R/bench.py --serial --steps 12 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 - $sk $R <<'PY'
import csv, glob, sys
sk, R = sys.argv[1], sys.argv[2]
v, t = [], []
for f in glob.glob(R + "/gpurun_out/dl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("void k_lighting") or r["Kernel_Name"].startswith("k_lighting"):
            v.append(float(r["Counter_Value"]))
for f in glob.glob(R + "/gpurun_out/dl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lighting" in r["Kernel_Name"]:
            t.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
big = [x for x in v if x > 1e6]; bt = [x for x in t if x > 20]
print("skip", sk, "VALU %.2f M" % (sum(big) / max(1, len(big)) / 1e6), "time %.1f us" % (sum(bt) / max(1, len(bt))))
PY
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/dl
