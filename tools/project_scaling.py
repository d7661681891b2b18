#!/usr/bin/env python3
"""Scaling PROJECTION from one GPU (no N-GPU node is available to this repository: nothing here is a measured scaling curve).

    python3 tools/project_scaling.py [--configs 3,4,5] [--ranks 2,4,8] [--frames 30] > profiles/<tag>_scaling_projection.json

For every config and N, every rank's context (tile_rank = r, tile_world = N: the super-tile partition, owned-region reject, per-rank Hi-Z
history - exactly what rank r of an N-GPU job runs) renders the frame ALONE on this GPU; the projected frame time of the job is the
slowest rank's, plus what the collectives would take over xGMI when they are not hidden behind the next frame's rendering:
    all-gather of the packed RGBA8 tiles: 4 B x W x H in total; a rank's slice goes to its N - 1 peers over N - 1 of its 7 links at once
    (direct) = slice / 153 GB/s, or around a ring = (N - 1) x slice / 153 GB/s;
    split-shadow mode adds an all-reduce(min) of the 1024^2 map: 2 (N - 1) / N x 4 MiB / 153 GB/s around a ring.
zr_dist_frame overlaps the all-gather with the next frame's rendering (double-buffered): it bounds the period only when it is longer
than a frame.  The MIN all-reduce of `split` mode is NOT hidden: it sits between this rank's shadow raster and its lighting pass
(zr_dist.cpp: lighting waits on the reduced map), so the host lane of a split frame is shadow pipeline + all-reduce + lighting, and the
frame period is the longer of that lane and the measured two-lane period.  The all-reduce is priced as ring bandwidth + a latency of
RCCL_HOP_US per ring step (2 (N - 1) steps) - an ASSUMPTION, stated in the output; nothing here has run on more than one GPU.
Modes: `replicated` = every rank renders the whole shadow map (the all-gather is the only collective: the north star's mode, bench.py's
default); `split` = rank r rasterises the casters i % N == r (zr_set_shadow_partition; the reduce is modelled, not executed here).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
XGMI_LINK_GBS = 153.0
RCCL_HOP_US = 6.0          # assumed latency per ring step of a small RCCL collective on xGMI (launch + one hop); 2 (N - 1) steps per all-reduce


def time_context(engine, cfg, n_point, rank, world, split, frames, warmup=8):
    g = engine.Renderer(cfg["width"], cfg["height"], 1024, tile_rank=rank, tile_world=world)
    engine.load_scene(g, cfg)
    if split:
        g.set_shadow_partition(rank, world)
    g.set_timing_interval(0)
    for i in range(warmup):
        g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)
        g.render()
    g.finish()
    t0 = time.perf_counter()
    for i in range(frames):
        g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * (warmup + i), 0.016 * (warmup + i))
        g.render()
    g.finish()
    wall = (time.perf_counter() - t0) / frames * 1e3
    # the GPU's own frame period (time between the ends of consecutive frames), median over the timed frames: a host hiccup in one of a
    # few dozen frames does not move it; the wall-clock mean is kept beside it
    per = sorted(g.frame_periods(frames - 1))
    ms = per[len(per) // 2] if per else wall
    st = g.stats()
    g.close()
    # the host lane's passes (shadow pipeline, lighting) each ALONE on the GPU (one stream): what the lane needs when nothing holds it up -
    # the two-lane pass timers include the waits for the other lane and for the previous frame's lighting, so they cannot be added up
    from zeldaengine_amd import abi
    g = engine.Renderer(cfg["width"], cfg["height"], 1024, tile_rank=rank, tile_world=world, flags=abi.FLAG_SERIAL_PASSES)
    engine.load_scene(g, cfg)
    if split:
        g.set_shadow_partition(rank, world)
    g.set_timing_interval(1)
    for i in range(10):
        g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)
        g.render()
    g.finish()
    pt = g.pass_times(6)
    g.close()
    return ms, {"survivors": st["survivors"], "bin_entries": st["bin_entries"], "overflow": st["overflow"], "wall_ms": round(wall, 4),
                "host_lane_ms": round(pt["cull_shadow"] + pt["shadow"] + pt["lighting"], 4), "passes_alone_ms": {k: round(v, 4) for k, v in pt.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="3,4,5")
    ap.add_argument("--ranks", default="2,4,8")
    ap.add_argument("--frames", type=int, default=30)
    args = ap.parse_args()
    from zeldaengine_amd import engine, scenes
    out = {"what": "PROJECTION from rank contexts timed one at a time on ONE MI355X - not a measured scaling curve (see the module docstring)",
           "xgmi_link_gbs": XGMI_LINK_GBS, "assumed_rccl_latency_us_per_ring_step": RCCL_HOP_US, "configs": {}}
    for c in [int(x) for x in args.configs.split(",")]:
        n_point = 256 if c == 5 else 16
        cfg = scenes.config3(10000, cube_dim=64) if c == 3 else scenes.config4(1000000, n_point, cube_dim=64)
        W, H = cfg["width"], cfg["height"]
        frames = args.frames if c == 3 else max(8, args.frames // 3)
        one, _ = time_context(engine, cfg, n_point, 0, 1, False, frames)
        entry = {"resolution": [W, H], "n_gpus_1_ms": round(one, 4), "ranks": {}}
        for n in [int(x) for x in args.ranks.split(",")]:
            slice_b = 4.0 * W * H / n
            ag_direct = slice_b / (XGMI_LINK_GBS * 1e9) * 1e3
            ag_ring = (n - 1) * slice_b / (XGMI_LINK_GBS * 1e9) * 1e3
            ar = 2.0 * (n - 1) / n * 4.0 * 1024 * 1024 / (XGMI_LINK_GBS * 1e9) * 1e3 + 2.0 * (n - 1) * RCCL_HOP_US * 1e-3
            row = {"allgather_ms": {"direct": round(ag_direct, 4), "ring": round(ag_ring, 4)}, "shadow_allreduce_ms_ring": round(ar, 4)}
            for mode, split in (("replicated", False), ("split", True)):
                per, lane = [], []
                for r in range(n):
                    ms, st = time_context(engine, cfg, n_point, r, n, split, frames)
                    per.append(round(ms, 4)); lane.append(st["host_lane_ms"])
                    sys.stderr.write("config %d N %d %s rank %d: %.3f ms %s\n" % (c, n, mode, r, ms, st))
                # replicated: the all-gather of frame k runs beside frame k + 1 (hidden unless longer than a frame).  split: the all-reduce is ON
                # the host lane, between the shadow raster and the lighting pass of the same frame
                crit = max(per)
                if split:
                    crit = max(crit, max(lane) + ar)
                period = max(crit, ag_ring)
                row[mode] = {"per_rank_ms": per, "slowest_rank_ms": max(per), "host_lane_ms": lane, "allreduce_on_host_lane_ms": round(ar, 4) if split else 0.0,
                             "allgather_ms_ring_hidden": round(ag_ring, 4),
                             "projected_frame_ms": round(period, 4), "projected_speedup": round(one / period, 3),
                             "projected_efficiency": round(one / period / n, 3)}
            entry["ranks"][str(n)] = row
        out["configs"]["config%d" % c] = entry
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
