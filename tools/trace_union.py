"""Busy time (union of kernel intervals over all streams) and period of the last frames of a rocprofv3 --kernel-trace run.
Usage: python tools/trace_union.py <trace_dir> [frames]"""
import csv, glob, sys
d = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").strip()
starts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "k_frame_begin"][-frames - 1:]
a, b = starts[0], starts[-1]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[a:b])
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
n = len(starts) - 1
print("frames %d: period %.1f us, GPU busy (union) %.1f us, sum of kernel durations %.1f us" %
      (n, span / n / 1e3, busy / n / 1e3, sum(e - s for s, e in iv) / n / 1e3))
