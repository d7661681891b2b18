"""Prints the dispatch sequence of the last frames of a rocprofv3 --kernel-trace run: per kernel position in the frame,
mean duration and mean gap to the previous dispatch (us).  Usage: python tools/trace_tail.py <trace_dir> [frames]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.split("(")[0].replace("void ", "").strip()


# a frame starts at k_frame_begin
starts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "k_frame_begin"]
starts = starts[-frames - 1:]
acc = defaultdict(lambda: [0.0, 0.0, 0])
order = []
for a, b in zip(starts[:-1], starts[1:]):
    for j in range(a, b):
        key = (j - a, short(rows[j]["Kernel_Name"]))
        dur = (int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])) / 1e3
        gap = (int(rows[j]["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"])) / 1e3 if j > a else 0.0
        if key not in acc:
            order.append(key)
        acc[key][0] += dur; acc[key][1] += gap; acc[key][2] += 1
tot = 0.0
for key in order:
    s = acc[key]
    print("%2d %-28s dur %8.2f us  gap %7.2f us  (n=%d)" % (key[0], key[1], s[0] / s[2], s[1] / s[2], s[2]))
    tot += (s[0] + s[1]) / s[2]
print("sum %.2f us" % tot)
