"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k)
            for c, v in sorted(cs.items()):
                print("   %-24s n=%3d mean=%14.1f" % (c, len(v), sum(v) / len(v)))
