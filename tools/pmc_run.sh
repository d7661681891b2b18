# usage: bash tools/pmc_run.sh <tag> "<COUNTER COUNTER ...>"   -> gpurun_out/pmc_<tag>.txt (mean per kernel)
tag=$1; ctrs=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$tag
timeout -k 5 ${PMC_TIMEOUT:-300} rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc_$tag.err || { tail -5 $R/gpurun_out/pmc_$tag.err; exit 1; }
cd $R && python - "$tag" <<'PY'
import csv, glob, sys
from collections import defaultdict
tag = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_%s.txt" % tag, "w") as out:
    for k in sorted(acc):
        line = k.ljust(34) + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items()))
        print(line); out.write(line + "\n")
PY
rm -rf $R/gpurun_out/pmc_$tag
