# usage (GPU box): bash tools/rank_kstats.sh <world> <rank> <config> [replicated|tiles|split]   -> one-stream per-kernel averages of one rank's context of an N-rank job
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_rk
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rk -o s -- python3 $GRAFT_REPO_ROOT/tools/rank_passes.py $1 $2 $3 serial ${4:-split} > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_rk/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("k_") and int(r["Calls"]) > 50: print("%-34s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (n[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf gpurun_out/prof_rk
