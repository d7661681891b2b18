# usage (GPU box): bash tools/kstats.sh [bench args]   -> one-stream per-kernel average durations (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ks
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ks -o s -- python3 $GRAFT_REPO_ROOT/bench.py --serial --steps 40 --warmup 10 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_ks/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("k_"): print("%-30s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf gpurun_out/prof_ks
