"""Copies the judged rocprofv3 summaries from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/make_profile_summary.py <tag> <kernel_stats_dir> <pmc_fetch_dir> <pmc_write_dir>

Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and refreshes profiles/pmc_traffic.json, the per-launch HBM
bytes bench.py reports as roofline.traffic.  Correction applied exactly as MI355X_MICROARCH.md (HBM section) prescribes:
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128 B read request, i.e. half the bytes of a
coalesced stream (calibrated here on k_count_shadow: a 4 MiB dword-per-lane read reports 2056 KiB), so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Collected in separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
shutil.copy(glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)[0], os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    return {"k_raster_chunks<0, false>": "k_raster<GBUFFER>", "k_raster_chunks<0, true>": "k_raster<GBUFFER,HiZ>",
            "k_raster_chunks<1, false>": "k_raster<SHADOW>", "k_cull<0, false>": "k_cull<GBUFFER>", "k_cull<1, false>": "k_cull<SHADOW>",
            "k_cull<0, true>": "k_cull<GBUFFER,worklist>", "k_cull<1, true>": "k_cull<SHADOW,worklist>"}.get(n, n)


def mean_counter(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = mean_counter(fetch_dir, "FETCH_SIZE"), mean_counter(write_dir, "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    out[k] = {"FETCH_SIZE_KiB": fetch.get(k, 0.0), "WRITE_SIZE_KiB": write.get(k, 0.0),
              "hbm_bytes_per_launch": int((2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024)}
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_pmc.json"), "w"), indent=1)
traffic = {k: v["hbm_bytes_per_launch"] for k, v in out.items()}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
