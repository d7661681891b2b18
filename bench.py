#!/usr/bin/env python3
"""bench.py — headline benchmark: Mpixels/s shaded, deferred PBR, 1920x1080, ~100k meshlets.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A step is one full frame of the hot path on synthetic, HBM-resident input:
    meshlet cull+bin (shadow) -> shadow raster -> meshlet cull+bin (camera) -> tile raster of last frame's visible set
    -> Hi-Z pyramid -> bin + tile raster of the rest (occlusion-tested) -> GBuffer write (resolve)
    -> deferred PBR + PCF lighting [-> RCCL all-gather of the packed RGBA8 tiles + untile, N > 1].
N > 1 partitions the SAME frame by screen tiles (tile t is rendered by rank t % N), so scaling is "strong".
value = W*H*steps / max-over-ranks(wall time of the K steps).

roofline: for the dominant kernel (largest mean hipEvent duration over the timed frames), algorithmic bytes
 / mean kernel time against the 8 TB/s HBM peak.  cpu_baseline: the scalar CPU oracle timed on rank 0 on a
bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def algorithmic_bytes(stats, cfg, n_tiles_owned_px):
    """SURVEY 8d per-frame algorithmic HBM bytes, split over the kernels that carry each term.

    GBuffer-write pass = 28 B x W*H (clear) + 28 B x covered px + geometry once per surviving meshlet-instance: the geometry
    term belongs to the rasteriser kernel (plus its 8 B visibility key per covered pixel), the 28 B terms to the resolve
    kernel, which is the one that writes the GBuffer.
    """
    geo = cfg["_geo_bytes_per_meshlet_instance"]
    SD = 1024
    return {
        "cull_shadow": 64 * stats["work_items"][0] + 4 * stats["survivors"][0],
        "shadow": geo * stats["survivors"][0] + 4 * SD * SD + 4 * stats["covered_shadow_texels"],
        "cull_camera": 64 * stats["work_items"][1] + 4 * stats["survivors"][1],
        "gbuffer": geo * stats["round1_survivors"] + 8 * stats["covered_pixels"] if stats["round1_survivors"] else
                   geo * stats["survivors"][1] + 8 * stats["covered_pixels"],
        "gbuffer2": geo * (stats["survivors"][1] - stats["round1_survivors"]) if stats["round1_survivors"] else 0,
        "resolve": 28 * n_tiles_owned_px + 28 * stats["covered_pixels"],
        "lighting": 28 * n_tiles_owned_px + 35068 + 4 * SD * SD,
    }


# passes that are ONE kernel launch (the roofline object is quoted on the longest of these); the cull / hiz passes are groups
KERNEL_OF_PASS = {"shadow": "k_raster<SHADOW>", "gbuffer": "k_raster<GBUFFER>", "gbuffer2": "k_raster<GBUFFER,HiZ>",
                  "resolve": "k_resolve_gbuffer", "lighting": "k_lighting"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--instances", type=int, default=10000, help="config 3 instance count (default: the metric's 10k)")
    ap.add_argument("--config", type=int, default=3, choices=[3, 4, 5],
                    help="3: the metric's workload (default); 4: 1M instances at 3840x2160; 5: config 4 with 256 point lights")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--replicated-shadow", action="store_true",
                    help="N > 1: every rank renders the whole shadow map (no shadow all-reduce; the all-gather of the composite is then the "
                         "only collective) instead of 1/N of the casters each + one MIN all-reduce of the 4 MB map")
    ap.add_argument("--timing-interval", type=int, default=0,
                    help="per-kernel hipEvents are recorded on every n-th timed frame (each record is a ~6 us stream bubble); "
                         "0 = steps // 6 clamped to 1..16, i.e. at least six timed frames")
    ap.add_argument("--cpu-sample-instances", type=int, default=10000)
    args = ap.parse_args()

    import torch
    from zeldaengine_amd import dist as zdist, engine, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal only (one-GPU box): ZR_BENCH_REHEARSAL=1 runs every rank on cuda:0 over gloo, to exercise the N > 1 frame loop end to
    # end (staged frame, shadow all-reduce, all-gather, composite) where RCCL cannot run; its numbers mean nothing.
    rehearsal = os.environ.get("ZR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    if args.config == 3:
        cfg = scenes.config3(args.instances)
    else:
        cfg = scenes.config4(args.instances if args.instances != 10000 else 1000000, 256 if args.config == 5 else 16)
    W, H = cfg["width"], cfg["height"]
    dr = zdist.DistributedRenderer(W, H, 1024, device_index=local_rank, rank=rank, world=world, split_shadow=not args.replicated_shadow)
    r = dr.r
    engine.load_scene(r, cfg)
    step = dr.frame        # render [+ ONE RCCL all-gather of the packed RGBA8 tiles + untile when world > 1]
    interval = args.timing_interval if args.timing_interval > 0 else max(1, min(16, args.steps // 6))
    interval = max(1, min(interval, args.steps))
    r.set_timing_interval(interval)

    for _ in range(args.warmup):
        step()
    dr.synchronize()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dr.synchronize()                # the render stream (render -> all-gather -> composite)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel means over the timed frames, from hipEvents recorded on the render stream
    times = r.pass_times(max(1, min(args.steps // interval, 64)))
    lat = sorted(r.frame_latencies(max(1, min(args.steps // interval, 64))))
    stats = r.stats()
    # Informative only (never `value`): the rate when the host reads every frame back over PCIe (zr_read_color), N = 1
    pcie_rate = None
    if world == 1:
        n_rb = max(5, min(20, args.steps))
        t1 = time.perf_counter()
        for _ in range(n_rb):
            step()
            r.color()
        pcie_rate = W * H * n_rb / (time.perf_counter() - t1) / 1e6
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = W * H * args.steps / elapsed / 1e6
        # geometry bytes per surviving meshlet-instance: mean over the mesh's meshlets of v*44 + t*3 + 64 + 32
        # (recomputed here from a fresh single-GPU-independent query of the mesh)
        mlt, _, _ = r.mesh_get_meshlets(0)
        cfg["_geo_bytes_per_meshlet_instance"] = float(np.mean(mlt["VertexCount"] * 44.0 + mlt["TriangleCount"] * 3.0 + 96.0))
        owned_px = 0
        tx = (W + 31) // 32
        ty = (H + 31) // 32
        for t in range(rank, tx * ty, world):
            x0, y0 = (t % tx) * 32, (t // tx) * 32
            owned_px += (min(W, x0 + 32) - x0) * (min(H, y0 + 32) - y0)
        alg = algorithmic_bytes(stats, cfg, owned_px)
        dom = max(KERNEL_OF_PASS, key=lambda k: times[k])
        achieved = alg[dom] / (times[dom] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(KERNEL_OF_PASS[dom])
            except Exception:      # noqa: BLE001
                traffic = None
        line = {
            "metric": "Mpixels/s shaded (deferred PBR, 1080p, 100k meshlets) + achieved HBM GB/s",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config%d: %d instanced 960-tri spheres (%d meshlet-instances), %dx%d, 1 directional + %d point "
                                   "lights, 1024^2 shadow map + 5x5 PCF, cubemap IBL" % (args.config, len(cfg["objects"][0]["instances"]),
                                                                                          stats["work_items"][1], W, H, len(cfg["point"])),
                       "resolution": [W, H], "parallelism": ("screen-tiles t%%%d, %s" % (world, "shadow map replicated" if args.replicated_shadow else
                                                                    "shadow casters i%%%d + MIN all-reduce" % world)) if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": KERNEL_OF_PASS[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel_ms": round(times[dom], 4), "algorithmic_bytes": int(alg[dom]),
                         "note": "two lanes (camera pipeline || shadow pipeline + previous frame's lighting): kernel_ms is the "
                                 "kernel's duration while sharing the GPU (alone: ZR_SERIAL_PASSES=1, profiles/r01_j_serial_kernel_stats.csv)"},
            "passes_ms": {k: round(v, 4) for k, v in times.items()},
            "passes_timing": "hipEvents on the render stream, every %d-th timed frame" % interval,
            "passes_gbs": {k: round(alg[k] / (times[k] * 1e-3) / 1e9, 2) for k in alg if times[k] > 0},
            "stats": stats,
            "pcie_inclusive_mpixels_s": None if pcie_rate is None else round(pcie_rate, 1),
            # GPU begin-to-end time of the sampled frames (two frames are in flight, so this exceeds ms_per_step)
            "frame_latency_ms": {"p10": round(lat[int(0.1 * (len(lat) - 1))], 4), "p50": round(lat[len(lat) // 2], 4),
                                 "p90": round(lat[int(0.9 * (len(lat) - 1) + 0.5)], 4), "samples": len(lat)} if lat else None,
        }
        if world == 1 and not args.no_cpu_baseline and args.config == 3:     # defined on the metric's workload only
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample_instances)
        print(json.dumps(line), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(n_inst):
    """The scalar CPU oracle on a bounded sample of the same workload (1 frame, 1 core)."""
    from oracle import pyoracle
    from zeldaengine_amd import scenes
    pyoracle.build()
    cfg = scenes.config3(n_inst)
    o = pyoracle.Oracle(cfg["width"], cfg["height"], 1024)
    pyoracle.load_scene(o, cfg)
    t0 = time.perf_counter()
    o.render()
    dt = time.perf_counter() - t0
    # the same frame again with the oracle's per-pixel stages (GBuffer resolve, lighting) on an OpenMP team of all host cores; the
    # rasteriser stays serial (it resolves the depth test in draw order).  Informative extra, same pixels.
    n_all = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_all = max(1, min(n_all, 16))               # a one-GPU box shares its host: 16 cores are this job's
    o.set_threads(n_all)
    o.render()                                   # first parallel region: thread team start-up
    t1 = time.perf_counter()
    o.render()
    dt_all = time.perf_counter() - t1
    return {"value": round(cfg["width"] * cfg["height"] / dt / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "1 frame of config 3 with %d of its 10000 instances at 1920x1080 (same lights, shadow map, PCF); "
                      "scalar C oracle, %.2f s" % (n_inst, dt),
            "value_all_cores": round(cfg["width"] * cfg["height"] / dt_all / 1e6, 4), "cores_all": n_all,
            "note_all_cores": "per-pixel stages on an OpenMP team, serial rasteriser; %.2f s" % dt_all}


if __name__ == "__main__":
    main()
