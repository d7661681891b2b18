"""Copies the judged rocprofv3 summaries from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/make_profile_summary.py <tag> <kernel_stats_dir> <bench_json> <pmc_dir> [<pmc_dir> ...]
    python tools/make_profile_summary.py --calib <tag> <pmc_dir of tools/valu_calib>

Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and points profiles/current.json at them; bench.py quotes
`roofline.traffic`, `valu_roofline`, `frame_hbm` and the per-kernel issue / wait fractions from that file ONLY when its `workload`
equals the workload it is running.

HBM bytes, exactly as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE come from separate --pmc passes (they
do not fit one), are in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128 B read request, i.e. half the bytes of a coalesced
stream (calibrated on k_count_shadow: a 4 MiB dword-per-lane read reports 2056 KiB), so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Per-frame totals = sum over every dispatch of the frame's kernels / number of frames in that run (= dispatches of k_resolve_gbuffer).

Issue / wait accounting per kernel (SQ counters, one --pmc pass; the guide: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves, and WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES):
    valu_active_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES     share of its life a wave spends with a vector instruction in execution
    wait_mem_frac    = SQ_WAIT_ANY / SQ_WAVE_CYCLES             ... parked on s_waitcnt / a barrier (memory, LDS, other waves)
    wait_issue_frac  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES        ... ready but not issued (the SIMD is busy with other waves)
    waves_per_simd   = SQ_WAVE_CYCLES * 4 / (duration x clock x 1024 SIMDs)          mean resident waves
    valu_simd_busy   = SQ_ACTIVE_INST_VALU * 4 / (duration x clock x 1024 SIMDs)     vector-pipe busy share of the kernel's SIMD time
with duration = the kernel's average from the --kernel-trace run and clock = 2.4 GHz.  profiles/<tag>_valu_calib_pmc.json holds the
same figures for tools/valu_calib's saturated loops: what "busy" reads at best.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_HZ, SIMDS = 2.4e9, 1024

SHORT = {"k_raster_chunks<0, false>": "k_raster<GBUFFER>", "k_raster_chunks<0, true>": "k_raster<GBUFFER,HiZ>",
         "k_raster_chunks<1, false>": "k_raster<SHADOW>", "k_raster_chunks<1, false, true>": "k_raster<SHADOW>", "k_raster_chunks<1, false, false>": "k_raster<SHADOW>",
         "k_raster_chunks<1, false, true, false>": "k_raster<SHADOW>", "k_raster_chunks<1, false, false, false>": "k_raster<SHADOW>",
         "k_raster_chunks<1, false, true, true>": "k_raster<SHADOW,late>", "k_raster_chunks<1, false, false, true>": "k_raster<SHADOW,late>",      # after k_shadow_occlusion
         "k_shadow_occlusion<true>": "k_shadow_occlusion", "k_shadow_occlusion<false>": "k_shadow_occlusion",
         "k_raster_chunks<0, false, false>": "k_raster<GBUFFER>", "k_raster_chunks<0, true, false>": "k_raster<GBUFFER,HiZ>",
         "k_tile_slow<0, false>": "k_tile_slow<GBUFFER>", "k_tile_slow<1, true>": "k_tile_slow<SHADOW>",
         "k_tile<0, false>": "k_tile<0>", "k_tile<0, true>": "k_tile<0>",      # (the frame's last round also draws the slow triangles)
         "k_geom<false, false>": "k_geom<false>", "k_geom<true, false>": "k_geom<true>",      # (<.., true>: the count-only run of a frame without a bucket plan)
         "k_geom<false, true>": "k_geom<false,count>", "k_geom<true, true>": "k_geom<true,count>",
         "k_cull_box<0, false>": "k_cull_box<GBUFFER>", "k_cull_box<1, false>": "k_cull_box<SHADOW>",
         "k_cull_box<0, true>": "k_cull_box<GBUFFER,worklist>", "k_cull_box<1, true>": "k_cull_box<SHADOW,worklist>", "k_cull<0, false>": "k_cull<GBUFFER>", "k_cull<1, false>": "k_cull<SHADOW>",
         "k_cull<0, true>": "k_cull<GBUFFER,worklist>", "k_cull<1, true>": "k_cull<SHADOW,worklist>"}


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    n = SHORT.get(n, n)
    for base in ("k_lighting", "k_resolve_gbuffer", "k_calib"):
        if n.startswith(base) and base != "k_calib":
            n = base
    return n


def counters(dirs):
    """-> {counter: {kernel: [value per dispatch]}} over every counter_collection.csv under the given directories"""
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Counter_Name"]][short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v) if v else None


def issue_rows(acc, dur_ns):
    """per kernel: the issue / wait fractions of the module docstring"""
    out = {}
    for k in sorted(acc.get("SQ_WAVE_CYCLES", {})):
        wc = mean(acc["SQ_WAVE_CYCLES"][k])
        if not wc:
            continue
        g = lambda c: mean(acc.get(c, {}).get(k, []))      # noqa: E731
        row = {"SQ_WAVES": g("SQ_WAVES"), "SQ_INSTS_VALU": g("SQ_INSTS_VALU"), "SQ_WAVE_CYCLES": wc, "SQ_BUSY_CYCLES": g("SQ_BUSY_CYCLES"),
               "valu_active_frac": round(g("SQ_ACTIVE_INST_VALU") / wc, 4) if g("SQ_ACTIVE_INST_VALU") is not None else None,
               "active_any_frac": round(g("SQ_ACTIVE_INST_ANY") / wc, 4) if g("SQ_ACTIVE_INST_ANY") is not None else None,
               "wait_mem_frac": round(g("SQ_WAIT_ANY") / wc, 4) if g("SQ_WAIT_ANY") is not None else None,
               "wait_issue_frac": round(g("SQ_WAIT_INST_ANY") / wc, 4) if g("SQ_WAIT_INST_ANY") is not None else None}
        d = dur_ns.get(k)
        if d:
            simd_quads = d * 1e-9 * CLOCK_HZ * SIMDS / 4.0
            row["avg_duration_us"] = round(d / 1e3, 2)
            row["waves_per_simd"] = round(wc / simd_quads, 2)
            if g("SQ_ACTIVE_INST_VALU") is not None:
                row["valu_simd_busy"] = round(g("SQ_ACTIVE_INST_VALU") / simd_quads, 4)
            if g("SQ_INSTS_VALU"):
                row["valu_cycles_per_inst_simd"] = round(d * 1e-9 * CLOCK_HZ * SIMDS / g("SQ_INSTS_VALU"), 2)
        if g("SQ_THREAD_CYCLES_VALU") is not None and g("SQ_ACTIVE_INST_VALU"):
            row["lanes_active_frac"] = round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")), 4)       # mean share of the 64 lanes live per vector instruction
        for c in ("SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_SMEM", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_SCA"):
            if g(c) is not None:
                row[c] = g(c)
        out[k] = row
    return out


if sys.argv[1] == "--calib":
    tag, d = sys.argv[2], sys.argv[3]
    acc = counters([d])
    rows = {}
    for k in sorted(acc.get("SQ_WAVE_CYCLES", {})):
        # one dispatch per (op, waves per SIMD), three warm-up launches before each: keep every 4th, in launch order
        vals = {c: acc[c][k] for c in acc if k in acc[c]}
        n = len(vals["SQ_WAVE_CYCLES"])
        rows[k] = [{c: vals[c][i] for c in vals} for i in range(3, n, 4)]
    for k, lst in rows.items():
        for r in lst:
            wc = r.get("SQ_WAVE_CYCLES") or 1.0
            r["valu_active_frac"] = round(r.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)
            r["wait_issue_frac"] = round(r.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
            r["active_quads_per_valu_inst"] = round(r.get("SQ_ACTIVE_INST_VALU", 0.0) / max(1.0, r.get("SQ_INSTS_VALU", 1.0)), 3)
    json.dump({"tag": tag, "what": "SQ counters of tools/valu_calib's loops (1, 2, 4, 8 waves per SIMD per op, in launch order): how a saturated "
               "vector pipe reads on the counters the frame's kernels are judged with", "kernels": rows},
              open(os.path.join(ROOT, "profiles", tag + "_valu_calib_pmc.json"), "w"), indent=1)
    print("calibration rows:", {k: len(v) for k, v in rows.items()})
    sys.exit(0)

tag, stats_dir, bench_json = sys.argv[1:4]
pmc_dirs = sys.argv[4:]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
stats_csv = glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats_csv, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
dur = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(stats_csv)):
    k = short(r["Name"])
    dur[k][0] += float(r["TotalDurationNs"]); dur[k][1] += int(r["Calls"])
dur_ns = {k: v[0] / v[1] for k, v in dur.items() if v[1]}


def is_frame_kernel(k):
    return k.startswith("k_") and k not in ("k_instance_prep", "k_fill64", "k_count_shadow")


acc = counters(pmc_dirs)
fetch, write, valu = acc.get("FETCH_SIZE", {}), acc.get("WRITE_SIZE", {}), acc.get("SQ_INSTS_VALU", {})
issue = issue_rows(acc, dur_ns)
out = {}
for k in sorted(set(fetch) | set(write) | set(valu)):
    f, w, v = fetch.get(k, []), write.get(k, []), valu.get(k, [])
    mf, mw = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    nfr = len(fetch.get("k_resolve_gbuffer", [])) or 1
    out[k] = {"FETCH_SIZE_KiB": mf, "WRITE_SIZE_KiB": mw, "hbm_bytes_per_launch": int((2 * mf + mw) * 1024),
              "launches_per_frame": round(len(f) / nfr, 3),
              "hbm_bytes_per_frame": int((2 * sum(f) / nfr + sum(w) / (len(write.get("k_resolve_gbuffer", [])) or 1)) * 1024),
              "SQ_INSTS_VALU": (sum(v) / len(v) if v else None)}
    if k in issue:
        out[k]["issue"] = issue[k]
if "k_lighting" in out:
    out["k_lighting"]["note"] = "two launches per frame: the full-screen pass + the one-pixel empty-colour pre-launch (the means are over both)"


def per_frame(a, scale):
    frames = len(a.get("k_resolve_gbuffer", [])) or 1
    return sum(sum(v) for k, v in a.items() if is_frame_kernel(k)) * scale / frames, frames


fb, nf = per_frame(fetch, 2048.0)
wb, nw = per_frame(write, 1024.0)
vi, nv = per_frame(valu, 1.0)
va, _ = per_frame(acc.get("SQ_ACTIVE_INST_VALU", {}), 1.0)
summary = {"tag": tag, "workload": bench["config"]["workload"], "lib_sha16": bench["config"].get("lib_sha16"), "bench_value_under_rocprof": bench["value"],
           "frames": {"fetch_pass": nf, "write_pass": nw, "sq_pass": nv},
           "hbm_bytes_per_frame": int(fb + wb), "valu_insts_per_frame": int(vi), "valu_active_quads_per_frame": int(va), "kernels": out,
           "method": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ issue set | SQ LDS / memory set in four separate passes of `python3 bench.py "
                     "--steps 20 --warmup 5 --no-cpu-baseline --no-extras`; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE correction); "
                     "issue fractions: tools/make_profile_summary.py docstring"}
json.dump(summary, open(os.path.join(ROOT, "profiles", tag + "_pmc.json"), "w"), indent=1)
cur = {"pmc": tag + "_pmc.json", "kernel_stats": tag + "_kernel_stats.csv"}
if os.path.exists(os.path.join(ROOT, "profiles", tag + "_valu_calib.json")):
    cur["valu_calib"] = tag + "_valu_calib.json"
json.dump(cur, open(os.path.join(ROOT, "profiles", "current.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}, indent=1))
for k, v in out.items():
    i = v.get("issue", {})
    print("%-28s %12d B/launch  VALU %-12s active %-7s wait_mem %-7s wait_issue %-7s waves/SIMD %-5s simd_busy %s" % (
        k, v["hbm_bytes_per_launch"], v["SQ_INSTS_VALU"], i.get("valu_active_frac"), i.get("wait_mem_frac"), i.get("wait_issue_frac"),
        i.get("waves_per_simd"), i.get("valu_simd_busy")))
