"""Copies the judged rocprofv3 summaries from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/make_profile_summary.py <tag> <kernel_stats_dir> <pmc_fetch_dir> <pmc_write_dir> <pmc_sq_dir> <bench_json>

Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and points profiles/current.json at them; bench.py quotes
`roofline.traffic`, `valu_roofline` and `frame_hbm` from that file ONLY when its `workload` equals the workload it is running.

HBM bytes, exactly as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE come from separate --pmc passes (they
do not fit one), are in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128 B read request, i.e. half the bytes of a coalesced
stream (calibrated on k_count_shadow: a 4 MiB dword-per-lane read reports 2056 KiB), so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Per-frame totals = sum over every dispatch of the frame's kernels / number of frames in that run (= dispatches of k_resolve_gbuffer).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir, fetch_dir, write_dir, sq_dir, bench_json = sys.argv[1:7]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
shutil.copy(glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)[0], os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])

SHORT = {"k_raster_chunks<0, false>": "k_raster<GBUFFER>", "k_raster_chunks<0, true>": "k_raster<GBUFFER,HiZ>",
         "k_raster_chunks<1, false>": "k_raster<SHADOW>", "k_raster_chunks<1, false, true>": "k_raster<SHADOW>", "k_raster_chunks<1, false, false>": "k_raster<SHADOW>",
         "k_raster_chunks<0, false, false>": "k_raster<GBUFFER>", "k_raster_chunks<0, true, false>": "k_raster<GBUFFER,HiZ>",
         "k_tile_slow<0, false>": "k_tile_slow<GBUFFER>", "k_tile_slow<1, true>": "k_tile_slow<SHADOW>",
         "k_cull_box<0, false>": "k_cull_box<GBUFFER>", "k_cull_box<1, false>": "k_cull_box<SHADOW>",
         "k_cull_box<0, true>": "k_cull_box<GBUFFER,worklist>", "k_cull_box<1, true>": "k_cull_box<SHADOW,worklist>", "k_cull<0, false>": "k_cull<GBUFFER>", "k_cull<1, false>": "k_cull<SHADOW>",
         "k_cull<0, true>": "k_cull<GBUFFER,worklist>", "k_cull<1, true>": "k_cull<SHADOW,worklist>"}
# (the triangle-binned camera pass's kernels keep their own names: k_cull_box<..>, k_select, k_geom<false|true>, k_scan_tri, k_index,
#  k_tile<0>, k_tile_slow<0>; k_scan_tri / k_index / k_tile / k_tile_slow run once per round, so launches_per_frame = 2)


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    n = SHORT.get(n, n)
    if n.startswith("k_lighting"):
        n = "k_lighting"
    if n.startswith("k_resolve_gbuffer"):
        n = "k_resolve_gbuffer"
    return n


def counters(d, counter):
    """-> ({kernel: [value per dispatch]})"""
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def is_frame_kernel(k):
    return k.startswith("k_") and k not in ("k_instance_prep", "k_fill64", "k_count_shadow")


fetch, write, valu = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE"), counters(sq_dir, "SQ_INSTS_VALU")
out = {}
for k in sorted(set(fetch) | set(write) | set(valu)):
    f, w, v = fetch.get(k, []), write.get(k, []), valu.get(k, [])
    mf, mw = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    nfr = len(fetch.get("k_resolve_gbuffer", [])) or 1
    out[k] = {"FETCH_SIZE_KiB": mf, "WRITE_SIZE_KiB": mw, "hbm_bytes_per_launch": int((2 * mf + mw) * 1024),
              "launches_per_frame": round(len(f) / nfr, 3),
              "hbm_bytes_per_frame": int((2 * sum(f) / nfr + sum(w) / (len(write.get("k_resolve_gbuffer", [])) or 1)) * 1024),
              "SQ_INSTS_VALU": (sum(v) / len(v) if v else None)}
if "k_lighting" in out:
    out["k_lighting"]["note"] = "two launches per frame: the full-screen pass + the one-pixel empty-colour pre-launch"


def per_frame(acc, scale):
    frames = len(acc.get("k_resolve_gbuffer", [])) or 1
    return sum(sum(v) for k, v in acc.items() if is_frame_kernel(k)) * scale / frames, frames


fb, nf = per_frame(fetch, 2048.0)
wb, nw = per_frame(write, 1024.0)
vi, nv = per_frame(valu, 1.0)
summary = {"tag": tag, "workload": bench["config"]["workload"], "bench_value_under_rocprof": bench["value"],
           "frames": {"fetch_pass": nf, "write_pass": nw, "sq_pass": nv},
           "hbm_bytes_per_frame": int(fb + wb), "valu_insts_per_frame": int(vi), "kernels": out,
           "method": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU in three separate passes of `python3 bench.py --steps 20 "
                     "--warmup 5 --no-cpu-baseline --no-extras`; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE correction)"}
json.dump(summary, open(os.path.join(ROOT, "profiles", tag + "_pmc.json"), "w"), indent=1)
json.dump({"pmc": tag + "_pmc.json", "kernel_stats": tag + "_kernel_stats.csv"}, open(os.path.join(ROOT, "profiles", "current.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}, indent=1))
for k, v in out.items():
    print("%-28s %12d B/launch  VALU %s" % (k, v["hbm_bytes_per_launch"], v["SQ_INSTS_VALU"]))
