"""One frame's kernels on a time axis, from a rocprofv3 --kernel-trace run: start / end (us from the frame's k_frame_begin), queue.
Usage: python tools/timeline.py <trace_dir> [frames-from-the-end]     (kernels that started within that frame's period, either lane)"""
import csv, glob, sys
d = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").strip()
fb = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "k_frame_begin"]
a, b = fb[-back - 1], fb[-back]
t0 = int(rows[a]["Start_Timestamp"])
qs = {}
print("period %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
for r in rows[a:b + 12]:
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%s%-28s %8.1f -> %8.1f  (%6.1f)" % ("    " * 10 * 0 + ("" if q == 0 else " " * 44), short(r["Kernel_Name"]), s, e, e - s))
