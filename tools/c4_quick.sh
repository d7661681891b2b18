# usage (GPU box): bash tools/c4_quick.sh [config]   -> bench line + per-kernel averages of config 4 / 5 (two-lane)
cfg=${1:-4}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c4q
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4q -o s -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/c4q.json 2>/dev/null
cd $GRAFT_REPO_ROOT; python - <<PY
import csv, glob, json
d = json.loads([l for l in open("gpurun_out/c4q.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], d["stats"])
f = glob.glob("gpurun_out/prof_c4q/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("k_") and float(r["Percentage"]) > 0.8: print("%-34s calls %4s avg %9.1f us %5.1f%%" % (n[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rm -rf gpurun_out/prof_c4q
