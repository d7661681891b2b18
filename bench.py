#!/usr/bin/env python3
"""bench.py — headline benchmark: Mpixels/s shaded, deferred PBR, 1920x1080, ~100k meshlets.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A step is one full frame of the hot path on synthetic, HBM-resident input, INCLUDING the engine's per-frame uniform work
(UpdateUniformBuffer, ZE:4585-4664: the point lights ride their spiral, XkView is rebuilt and uploaded every frame):
    meshlet cull+bin (shadow) -> shadow raster -> meshlet cull (camera) -> triangle records + tile raster of last frame's visible set
    -> Hi-Z pyramid -> triangle records + tile raster of the rest (occlusion-tested) -> GBuffer write (resolve)
    -> deferred PBR + PCF lighting [-> RCCL all-gather of the packed RGBA8 tiles + untile, N > 1].
N > 1 partitions the SAME frame by screen tiles, so scaling is "strong".  value = W*H*steps / max-over-ranks(wall time).

roofline: quoted for the pass the north star names, the GBuffer-write pass = the two camera rounds (k_geom + k_tile each) +
k_resolve_gbuffer, against the 8 TB/s HBM peak.  `achieved` = SURVEY 8(d)'s algorithmic bytes of that
pass / the summed mean duration of those kernels, measured live with HIP events on the stream they are launched on (the library's
camera lane).
`traffic` = the same kernels' FETCH_SIZE / WRITE_SIZE counter bytes from the committed rocprofv3 summary named in
`traffic_source`, only when that summary was collected on this very workload (else null).  `compulsory_bytes` / `compulsory_frac`:
the bytes any implementation of the pass must move (every pixel's 28 GBuffer bytes once + the scene's unique geometry once) over the same
time.  `issue_frac`: the pass's SQ_INSTS_VALU (same summary, same build) / the best instruction rate tools/valu_calib reached on this chip
/ the pass's time - how close the pass runs to the chip's measured vector issue rate.  `kernels` lists every kernel the same
way; `valu_roofline` states what actually bounds the frame (vector-ALU issue); `frame_hbm` is the whole frame's counter traffic
over the frame time.  cpu_baseline: the scalar CPU oracle timed on rank 0 on the same workload (N = 1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# Vector-ALU issue peak: a SIMD-32 issues a wave64 instruction over 2 cycles (MI355X_MICROARCH.md: 157.3 TFLOP/s FP32 vector = 32 lanes per
# clock per SIMD), 1 024 SIMDs at 2.4 GHz.  What plain loops of independent instructions actually reach on this chip is measured by
# tools/valu_calib (profiles/<tag>_valu_calib.json): one wave alone issues one instruction per ~8 cycles, two per SIMD reach 4, eight
# ~2.7 (0.9 T wave-instructions/s for v_mul_f32 / v_add_u32, 0.6-0.7 T for a three-source v_fma_f32) - both are quoted.
VALU_PEAK_PER_S = 256 * 4 * 2.4e9 / 2.0
PROFILE_INDEX = os.path.join(ROOT, "profiles", "current.json")     # written by tools/make_profile_summary.py

# the kernels of each timed pass (the cull / hiz passes are groups of launch-bound kernels).  A camera round of the triangle-binned
# pass is k_geom (vertices -> triangle records in their tiles' buckets) + k_tile; k_tile runs once per round under one name, so its
# profile row holds both rounds.  (k_plan - the next frame's buckets - runs behind the resolve, off the pass's path.)
KERNEL_OF_PASS = {"shadow": ("k_raster<SHADOW>", "k_tile_slow<SHADOW>"), "gbuffer": ("k_geom<false>", "k_tile<0>"),
                  "gbuffer2": ("k_geom<true>",), "resolve": ("k_resolve_gbuffer",), "lighting": ("k_lighting",)}
# kernels a pass runs only in some scenes (the shadow pass's occlusion culling: on from one meshlet-instance per five texels of the map)
KERNEL_OF_PASS_OPTIONAL = {"shadow": ("k_shadow_occlusion", "k_raster<SHADOW,late>")}
GBUFFER_WRITE_PASS = ("gbuffer", "gbuffer2", "resolve")
GBUFFER_WRITE_KERNELS = "k_geom + k_tile (x2 rounds; the last one also draws the slow triangles) + k_resolve_gbuffer"


def workload_name(config, n_inst, n_work, W, H, n_point, cube_dim):
    return ("config%d: %d instanced 960-tri spheres (%d meshlet-instances), %dx%d, 1 directional + %d point lights riding the engine's "
            "spiral (uniforms rebuilt every frame), 1024^2 shadow map + 5x5 PCF, %d^2 cubemap IBL" % (config, n_inst, n_work, W, H, n_point, cube_dim))


def algorithmic_bytes(stats, mesh, n_inst, owned_px):
    """SURVEY 8(d) per-frame algorithmic HBM bytes, literally, per pass.

    Cull       64 B x meshlets tested + 32 B x instances + 4 B x survivors written
    Shadow     44 B x unique verts + 4 B x indices + 32 B x instances + 4 B x 1024^2 clear + 4 B x covered shadow texels
    GBuffer-w  28 B x W*H (clear) + 28 B x covered px + geometry once per surviving meshlet-instance
               (vertexCount*44 + triangleCount*3 + 64 + 32, mean over the mesh's meshlets); split over the kernels that carry
               each term: geometry -> the two rasteriser launches, the 28 B terms -> k_resolve_gbuffer
    Lighting   28 B x W*H + 35 068 B UBO + 4 MB shadow map
    """
    SD = 1024
    geo = mesh["geo"]
    r1 = stats["round1_survivors"] if stats["round1_survivors"] else stats["survivors"][1]
    r2 = stats["survivors"][1] - r1
    return {
        "cull_shadow": 64 * stats["work_items"][0] + 32 * n_inst + 4 * stats["survivors"][0],
        "shadow": 44 * mesh["n_verts"] + 4 * mesh["n_idx"] + 32 * n_inst + 4 * SD * SD + 4 * stats["covered_shadow_texels"],
        "cull_camera": 64 * stats["work_items"][1] + 32 * n_inst + 4 * stats["survivors"][1],
        "gbuffer": geo * r1,
        "gbuffer2": geo * r2,
        "resolve": 28 * owned_px + 28 * stats["covered_pixels"],
        "lighting": 28 * owned_px + 35068 + 4 * SD * SD,
    }


def lib_sha16():
    """First 16 hex digits of the SHA-256 of the renderer library this process loads: profiles/<tag>_pmc.json records the one its counters
    were collected with, and a line whose library differs quotes no counter (a kernel change without a fresh profile would ship stale ones)."""
    import hashlib
    from zeldaengine_amd import engine
    try:
        return hashlib.sha256(open(engine.LIB_PATH, "rb").read()).hexdigest()[:16]
    except Exception:      # noqa: BLE001
        return None


def load_profile(workload, sha):
    """The committed rocprofv3 summary (kernel stats + separate FETCH_SIZE / WRITE_SIZE / SQ passes), if it is for this workload AND was
    collected with this very build of the library."""
    try:
        idx = json.load(open(PROFILE_INDEX))
        prof = json.load(open(os.path.join(ROOT, "profiles", idx["pmc"])))
    except Exception:      # noqa: BLE001
        return None
    if prof.get("workload") != workload or prof.get("lib_sha16") != sha:
        return None
    prof["_file"] = "profiles/" + idx["pmc"]
    try:        # the measured issue rates of this chip (tools/valu_calib), collected with the profile
        cal = json.load(open(os.path.join(ROOT, "profiles", idx["valu_calib"])))
        plain = [r for r in cal["rows"] if r["op"].startswith(("v_fma_f32 (", "v_mul_f32", "v_add_u32"))]
        best = max(plain, key=lambda r: r["ginst_per_s"])
        prof["_calib"] = {"file": "profiles/" + idx["valu_calib"], "best_per_s": best["ginst_per_s"] * 1e9, "best_op": best["op"],
                          "best_waves_per_simd": best["waves_per_simd"],
                          "one_wave_cycles_per_inst": min(r["cyc_per_inst_wave"] for r in plain if r["waves_per_simd"] == 1)}
    except Exception:      # noqa: BLE001
        prof["_calib"] = None
    return prof


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))] if xs else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--instances", type=int, default=10000, help="config 3 instance count (default: the metric's 10k)")
    ap.add_argument("--config", type=int, default=3, choices=[3, 4, 5],
                    help="3: the metric's workload (default); 4: 1M instances at 3840x2160; 5: config 4 with 256 point lights")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the moving-camera / textured / one-stream extra loops")
    ap.add_argument("--split-shadow", action="store_true",
                    help="N > 1: every rank rasterises 1/N of the casters and the maps are MIN all-reduced (4 MiB, on the shadow -> lighting "
                         "dependency of every frame).  NOT the default: no N > 1 run on hardware has compared the two modes yet "
                         "(profiles/r04_scaling_projection.json is a one-GPU projection), and the north star names one collective")
    ap.add_argument("--shadow-tiles", action="store_true",
                    help="N > 1: the shadow MAP is owned by light-space super-tiles like the frame (a rank draws the casters that reach its tiles; "
                         "one more all-gather, of the packed shadow tiles: 4 MiB in total, nothing reduced).  NOT the default either, for the "
                         "same reason")
    ap.add_argument("--replicated-shadow", action="store_true",
                    help="(the default for N > 1, kept so that older command lines still parse) every rank renders the whole 1024^2 shadow "
                         "map; the all-gather of the composite is the only collective")
    ap.add_argument("--python-dist", action="store_true", help="N > 1: torch.distributed frame loop (dist.py) instead of the library's native RCCL host")
    ap.add_argument("--timing-interval", type=int, default=0,
                    help="per-kernel hipEvents are recorded on every n-th timed frame (each record is a ~6 us stream bubble); "
                         "0 = steps // 12 clamped to 5..8 (a short run still leaves four frames in five undisturbed)")
    ap.add_argument("--cube-dim", type=int, default=1024, help="cubemap face edge (the engine's is 1024: 11 mips)")
    ap.add_argument("--flags", type=int, default=0, help="experiments: extra zr_config flags (e.g. 8 = ZR_FLAG_NO_HIZ)")
    ap.add_argument("--textured", action="store_true", help="profiling: config 3 with seven sampled 512^2 material images in the main loop "
                                                             "(the default line reports this variant as value_textured)")
    ap.add_argument("--serial", action="store_true", help="profiling: the whole frame on ONE stream (ZR_FLAG_SERIAL_PASSES), every kernel alone on the GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: start the N ranks as a CHILD process (torch.distributed.run, one rank per GPU) and relay
        # its output and exit code.  This parent never touches the GPU (nothing of torch or HIP is imported here) and never exec()s.
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    from zeldaengine_amd import abi, dist as zdist, engine, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal only (one-GPU box): ZR_BENCH_REHEARSAL=1 runs every rank on cuda:0 over gloo, to exercise the N > 1 frame loop end to
    # end where RCCL cannot run; its numbers mean nothing.
    rehearsal = os.environ.get("ZR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback")
    if world > 1 and not rehearsal and torch.cuda.device_count() < world:
        # RCCL refuses two ranks on one device (and would hang the others inside ncclCommInitRank): say so before any collective
        raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s); one process per GPU needs %d (ZR_BENCH_REHEARSAL=1 runs every rank "
                         "on cuda:0 over gloo to rehearse the frame loop: its numbers mean nothing)" % (world, torch.cuda.device_count(), world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Control plane (the ncclUniqueId hand-over, the timing barrier, the MAX over ranks) on gloo with CPU tensors; the frame's
        # collectives are the library's own RCCL communicator (zr_dist_*).  torch's RCCL communicator ("nccl" IS RCCL on ROCm) is created
        # lazily, i.e. only if the torch.distributed frame loop runs (--python-dist, or the agreed fallback): one communicator per rank.
        dist.init_process_group("gloo" if rehearsal else "cpu:gloo,cuda:nccl")

    def barrier():
        if world > 1:
            dist.all_reduce(torch.zeros(1, dtype=torch.int32))      # a CPU tensor: gloo

    args.split_shadow = world > 1 and args.split_shadow and not args.replicated_shadow
    args.shadow_mode = "replicated" if (world == 1 or args.replicated_shadow) else "tiles" if args.shadow_tiles else "split" if args.split_shadow else "replicated"
    n_point = 256 if args.config == 5 else 16
    if args.config == 3:
        cfg = scenes.config3(args.instances, cube_dim=args.cube_dim, textured=args.textured)
    else:
        cfg = scenes.config4(args.instances if args.instances != 10000 else 1000000, n_point, cube_dim=args.cube_dim)
    W, H = cfg["width"], cfg["height"]
    n_inst = len(cfg["objects"][0]["instances"])

    def make_renderer(cfg_, flags=0):
        native = world > 1 and not args.python_dist and not rehearsal
        dr_ = zdist.make_distributed(W, H, 1024, device_index=local_rank, rank=rank, world=world, flags=flags,
                                     split_shadow=args.shadow_mode, native=native)
        engine.load_scene(dr_.r, cfg_)
        return dr_

    dr = make_renderer(cfg, flags=(abi.FLAG_SERIAL_PASSES if args.serial else 0) | args.flags)
    r = dr.r
    fallback = getattr(dr, "native_fallback", None)
    interval = args.timing_interval if args.timing_interval > 0 else max(min(5, args.steps), min(8, args.steps // 12))
    interval = max(1, min(interval, args.steps))
    r.set_timing_interval(interval)

    cam0 = cfg["camera"]

    def uniforms_static(dr_, i):
        # UpdateUniformBuffer (ZE:4585-4664) every frame: RollLight advances, the point lights move along their spiral, XkView is
        # rebuilt on the host and uploaded; the camera stays (SURVEY 8d "camera static")
        dr_.r.update_uniforms(cam0, cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)

    def uniforms_orbit(dr_, i):
        a = math.radians(45.0 + 2.0 * i)                 # 2 degrees per frame around the scene at the default camera's radius / height
        cam = abi.make_camera((math.sqrt(50.0) * math.cos(a), math.sqrt(50.0) * math.sin(a), 5.0), (0.0, 0.0, 0.0))
        dr_.r.update_uniforms(cam, cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)

    def uniforms_light_orbit(dr_, i):
        # the directional light (the shadow caster) circles the scene at its own radius and height, 1 degree per frame: the shadow pass's work
        # list is rebuilt every frame and its occlusion flags (scenes dense enough to have them on) are a frame stale
        d = cfg["dir"].copy()
        p0 = np.array(cfg["dir"]["Position"][0][:3], dtype=np.float64)
        rad, a = math.hypot(p0[0], p0[1]), math.atan2(p0[1], p0[0]) + math.radians(1.0 * i)
        d["Position"][0][:3] = (rad * math.cos(a), rad * math.sin(a), p0[2]); d["Direction"][0][:3] = d["Position"][0][:3]
        dr_.r.update_uniforms(cam0, d, cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)

    def timed_loop(dr_, steps, warmup, per_frame):
        # (CPython's cyclic collector off while frames are enqueued: with torch imported a full collection takes ~37 ms of host time - a hole
        # of a hundred frames in the GPU's queue - and falls around the 160th frame of any loop that long: tools/host_variant_check.py)
        import gc
        gc.collect(); gc.disable()
        try:
            return timed_loop_body(dr_, steps, warmup, per_frame)
        finally:
            gc.enable()

    def timed_loop_body(dr_, steps, warmup, per_frame):
        for i in range(warmup):
            per_frame(dr_, i)
            dr_.frame()
        dr_.synchronize()
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            per_frame(dr_, warmup + i)
            dr_.frame()
        dr_.synchronize()                # render stream + camera lane + collective stream
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        timed_loop.local = el
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed_loop(dr, args.steps, args.warmup, uniforms_static)
    elapsed_local = timed_loop.local

    # per-kernel means over the sampled frames (hipEvents on the stream each kernel runs on), per-frame GPU periods of ALL timed frames
    n_s = max(1, min(args.steps // interval, 64))
    times = r.pass_times(n_s)
    lat = r.frame_latencies(n_s)
    periods = r.frame_periods(min(args.steps - 1, 511)) if args.steps > 1 else []
    stats = r.stats()
    mlt, _, _ = r.mesh_get_meshlets(0)
    mesh = {"geo": float(np.mean(mlt["VertexCount"] * 44.0 + mlt["TriangleCount"] * 3.0 + 96.0)),
            "n_verts": len(cfg["objects"][0]["mesh"][0]), "n_idx": len(cfg["objects"][0]["mesh"][1])}

    extras = {}
    if not args.no_extras:
        k = max(10, min(args.steps, 60))
        el = timed_loop(dr, k, 5, uniforms_orbit)
        extras["value_moving_camera"] = round(W * H * k / el / 1e6, 3)
        extras["moving_camera_note"] = "same workload, camera orbiting 2 deg/frame (stale Hi-Z history every frame), %d frames" % k
        el = timed_loop(dr, k, 5, uniforms_light_orbit)
        extras["value_moving_light"] = round(W * H * k / el / 1e6, 3)
        extras["moving_light_note"] = "same workload, the directional light orbiting 1 deg/frame (shadow work list rebuilt, shadow occlusion flags stale every frame), %d frames" % k
        if world == 1:
            n_rb = max(5, min(20, args.steps))        # informative only (never `value`): every frame read back over PCIe
            t1 = time.perf_counter()
            for i in range(n_rb):
                uniforms_static(dr, i)
                dr.frame()
                r.color()
            extras["pcie_inclusive_mpixels_s"] = round(W * H * n_rb / (time.perf_counter() - t1) / 1e6, 1)
    r.close()
    serial = None
    if not args.no_extras and world == 1:
        # every kernel alone on the GPU (one stream): what the rocprofv3 summaries under profiles/*_serial_* show
        ds = make_renderer(cfg, flags=abi.FLAG_SERIAL_PASSES)
        ds.r.set_timing_interval(1)
        k = max(10, min(args.steps, 40))
        el = timed_loop(ds, k, 5, uniforms_static)
        serial = {"ms_per_step": round(el / k * 1e3, 4), "passes_ms": {a: round(b, 4) for a, b in ds.r.pass_times(min(k, 64)).items()}}
        ds.r.close()
        if args.config == 3:
            cfg_t = scenes.config3(args.instances, cube_dim=args.cube_dim, textured=True)
            dt = make_renderer(cfg_t)
            dt.r.set_timing_interval(0)
            k = max(10, min(args.steps, 60))
            el = timed_loop(dt, k, 10, uniforms_static)
            extras["value_textured"] = round(W * H * k / el / 1e6, 3)
            extras["textured_note"] = "same scene with seven non-constant 512^2 material textures (trilinear + anisotropic sampling in the resolve), %d frames" % k
            dt.r.close()

    # N > 1: the configuration the north star assigns to a node - config 4 (1 M instances, 3840 x 2160) with the shadow map owned by light-space
    # tiles - timed on the same ranks right after the metric's loop, so that the FIRST run on a node shows the design that is meant to scale and
    # not only config 3 (0.38 ms of ~ 20 dependent launches per frame, which cannot).  A block of its own: `value` stays config 3.
    c4_block = None
    if world > 1 and not args.no_extras and args.config == 3 and not args.textured:
        try:
            c4_block = config4_tiles_block(args, world, rank, local_rank, rehearsal, timed_loop, gather_rows=lambda row: gather_json(row, world, dist, torch))
        except Exception as e:      # noqa: BLE001
            c4_block = {"error": "%s: %s" % (type(e).__name__, e)}

    rank_rows = None
    if world > 1:
        # every rank's own picture (wall time of its loop, GPU frame period, per-pass GPU times, survivors): the first hardware run of
        # N > 1 must be readable without a second one
        mine = {"rank": rank, "loop_ms_per_step": round(elapsed_local / args.steps * 1e3, 4),
                "frame_gpu_ms_median": round(pct(periods, 0.5), 4) if periods else None,
                "passes_ms": {k: round(v, 4) for k, v in times.items()},
                "survivors": stats["survivors"], "covered_pixels": stats["covered_pixels"], "overflow": stats["overflow"]}
        rows = gather_json(mine, world, dist, torch)
        rank_rows = {"slowest": max(rows, key=lambda r: r["frame_gpu_ms_median"] or r["loop_ms_per_step"])["rank"], "per_rank": rows}
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = W * H * args.steps / elapsed / 1e6
        owned_px = 0
        tx = (W + 31) // 32
        tiles = zdist.owned_tiles(rank, world, tx, (H + 31) // 32)
        for t in tiles:
            x0, y0 = (t % tx) * 32, (t // tx) * 32
            owned_px += (min(W, x0 + 32) - x0) * (min(H, y0 + 32) - y0)
        alg = algorithmic_bytes(stats, mesh, n_inst, owned_px)
        workload = workload_name(args.config, n_inst, stats["work_items"][1], W, H, len(cfg["point"]), args.cube_dim)
        if args.textured:
            workload += " [seven sampled 512^2 material images]"
        sha = lib_sha16()
        prof = load_profile(workload, sha) if world == 1 else None
        ptraffic = (prof or {}).get("kernels", {})

        def pass_kernels(p):
            return tuple(KERNEL_OF_PASS[p]) + tuple(k for k in KERNEL_OF_PASS_OPTIONAL.get(p, ()) if k in ptraffic)

        def pass_traffic(p):
            t = [ptraffic.get(k, {}).get("hbm_bytes_per_frame") for k in pass_kernels(p)]
            return int(sum(t)) if t and all(x is not None for x in t) else None

        def kernel_row(p, ms):
            row = {"pass": p, "kernels": list(pass_kernels(p)), "ms": round(ms[p], 4), "algorithmic_bytes": int(alg[p]),
                   "achieved_gbs": round(alg[p] / (ms[p] * 1e-3) / 1e9, 2) if ms[p] > 0 else None, "traffic": pass_traffic(p)}
            # what bounds each kernel, from the SQ counters of the committed profile (tools/make_profile_summary.py: share of a wave's life
            # with a vector instruction in execution / parked on memory or a barrier / ready but not issued; mean resident waves per SIMD)
            iss = {k: ptraffic[k]["issue"] for k in pass_kernels(p) if k in ptraffic and "issue" in ptraffic[k]}
            if iss:
                row["issue"] = {k: {f: v.get(f) for f in ("valu_active_frac", "wait_mem_frac", "wait_issue_frac", "waves_per_simd", "valu_simd_busy", "lanes_active_frac",
                                                          "valu_cycles_per_inst_simd")} for k, v in iss.items()}
            return row

        def gb_pass(ms):
            t = sum(ms[p] for p in GBUFFER_WRITE_PASS)
            b = sum(alg[p] for p in GBUFFER_WRITE_PASS)
            return t, b, (b / (t * 1e-3) / 1e9 if t > 0 else 0.0)

        t_gb, b_gb, gbs_gb = gb_pass(times)
        tr = [pass_traffic(p) for p in GBUFFER_WRITE_PASS]
        roofline = {
            "bound": "hbm",
            "kernel": "GBuffer-write pass (%s)" % GBUFFER_WRITE_KERNELS,
            "achieved": round(gbs_gb, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs_gb / HBM_PEAK_GBS, 6),
            "traffic": int(sum(tr)) if prof and all(t is not None for t in tr) else None,
            "traffic_source": prof["_file"] if prof else None,
            "traffic_note": None if prof else "no committed counter summary for this workload AND this build of the library (profiles/current.json; "
                                              "tools/collect_profiles.sh makes one): counters are not quoted from another build",
            "kernel_ms": round(t_gb, 4), "algorithmic_bytes": int(b_gb),
            "algorithmic_bytes_formula": "28*W*H + 28*covered_px + mean(v*44 + t*3 + 96) * camera survivors (SURVEY 8d)",
            "timing": "HIP events on the library's camera lane, mean over %d sampled frames; the lane shares the GPU with the shadow pipeline and "
                      "the previous frame's lighting" % n_s,
        }
        # compulsory bytes: what ANY implementation of the pass must move - every pixel's GBuffer values once (28 B) and the scene's unique
        # geometry once (the mesh's vertices and indices + one instance record per instance), whatever the number of instances of a meshlet
        comp = 28 * owned_px + 44 * mesh["n_verts"] + 4 * mesh["n_idx"] + 32 * n_inst
        roofline["compulsory_bytes"] = int(comp)
        roofline["compulsory_frac"] = round(comp / (t_gb * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if t_gb > 0 else None
        # issue_frac: the pass's vector instructions (SQ_INSTS_VALU of its kernels, committed counter summary of THIS build) / the best
        # rate tools/valu_calib reached on this chip / the pass's time: how close the pass runs to the chip's measured instruction issue
        pass_valu = None
        if prof:
            pv = [ptraffic.get(k, {}).get("SQ_INSTS_VALU") for p_ in GBUFFER_WRITE_PASS for k in pass_kernels(p_)]
            pl = [ptraffic.get(k, {}).get("launches_per_frame") for p_ in GBUFFER_WRITE_PASS for k in pass_kernels(p_)]
            if pv and all(x is not None for x in pv) and all(x is not None for x in pl):
                pass_valu = sum(a * b for a, b in zip(pv, pl))
        cal = (prof or {}).get("_calib")
        if pass_valu and cal and t_gb > 0:
            roofline["valu_wave_insts"] = int(pass_valu)
            roofline["issue_frac"] = round(pass_valu / cal["best_per_s"] / (t_gb * 1e-3), 4)
            roofline["issue_peak_per_s"] = cal["best_per_s"]
        else:
            roofline["issue_frac"] = None
        if serial:
            t1, b1, g1 = gb_pass(serial["passes_ms"])
            roofline["one_stream"] = {"kernel_ms": round(t1, 4), "achieved": round(g1, 3), "frac": round(g1 / HBM_PEAK_GBS, 6),
                                      "compulsory_frac": round(comp / (t1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if t1 > 0 else None,
                                      "issue_frac": round(pass_valu / cal["best_per_s"] / (t1 * 1e-3), 4) if (pass_valu and cal and t1 > 0) else None,
                                      "note": "the same kernels alone on the GPU (ZR_FLAG_SERIAL_PASSES run below)"}
        valu = None
        if prof and prof.get("valu_insts_per_frame"):
            mn = prof["valu_insts_per_frame"] / VALU_PEAK_PER_S * 1e3
            valu = {"valu_wave_insts_per_frame": int(prof["valu_insts_per_frame"]), "peak_per_s": VALU_PEAK_PER_S, "min_ms": round(mn, 4),
                    "frame_ms": round(ms_per_step, 4), "frac": round(mn / ms_per_step, 4), "source": prof["_file"],
                    "note": "sum of SQ_INSTS_VALU over the frame's kernels x 2 cycles / (1024 SIMDs x 2.4 GHz), the datasheet issue rate"}
            cal = prof.get("_calib")
            if cal:
                mm = prof["valu_insts_per_frame"] / cal["best_per_s"] * 1e3
                valu["measured"] = {"peak_per_s": cal["best_per_s"], "op": cal["best_op"], "waves_per_simd": cal["best_waves_per_simd"],
                                    "one_wave_cycles_per_inst": cal["one_wave_cycles_per_inst"], "min_ms": round(mm, 4),
                                    "frac": round(mm / ms_per_step, 4), "source": cal["file"],
                                    "note": "against the best rate tools/valu_calib reached on this chip with independent instructions"}
        frame_hbm = None
        if prof and prof.get("hbm_bytes_per_frame"):
            g = prof["hbm_bytes_per_frame"] / (ms_per_step * 1e-3) / 1e9
            frame_hbm = {"traffic": int(prof["hbm_bytes_per_frame"]), "gbs": round(g, 1), "frac": round(g / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes": int(sum(alg.values())), "source": prof["_file"]}
        line = {
            "metric": "Mpixels/s shaded (deferred PBR, 1080p, 100k meshlets) + achieved HBM GB/s",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "resolution": [W, H], "lib_sha16": sha,
                       "parallelism": ("screen super-tiles over %d ranks, %s, %s" % (world, "shadow casters i%%%d + MIN all-reduce" % world if args.shadow_mode == "split"
                                       else "shadow map owned by light-space super-tiles + all-gather of the packed shadow tiles" if args.shadow_mode == "tiles"
                                       else "shadow map replicated (all-gather of the composite is the only collective)",
                                       "torch.distributed loop" if (args.python_dist or rehearsal) else
                                       ("torch.distributed loop (the native RCCL host could not be brought up: %s)" % fallback if fallback
                                        else "native RCCL host (zr_dist_*)"))) if world > 1 else "single GPU"},
            "roofline": roofline,
            "kernels": [kernel_row(p, times) for p in KERNEL_OF_PASS],
            "valu_roofline": valu,
            "frame_hbm": frame_hbm,
            "passes_ms": {k: round(v, 4) for k, v in times.items()},
            "passes_timing": "hipEvents on the stream each kernel runs on, every %d-th timed frame (%d samples)" % (interval, n_s),
            "frame_gpu_ms": {"median": round(pct(periods, 0.5), 4), "p10": round(pct(periods, 0.1), 4), "p90": round(pct(periods, 0.9), 4),
                             "samples": len(periods), "what": "GPU time between the ends of consecutive frames, every timed frame"} if periods else None,
            "frame_latency_ms": {"p10": round(pct(lat, 0.1), 4), "p50": round(pct(lat, 0.5), 4), "p90": round(pct(lat, 0.9), 4),
                                 "samples": len(lat), "what": "first kernel to end of lighting (two frames are in flight)"} if lat else None,
            "one_stream": serial,
            "stats": stats,
        }
        # 'shaded' pixels: every pixel of the target goes through BaseLighting.frag; the ones that still hold the clear values of every GBuffer
        # target take the shader's constant result for them (computed once per frame by the same kernel) - stated, not hidden
        if W * H:
            cf = stats["covered_pixels"] / float(owned_px if world > 1 else W * H)
            line["covered_fraction"] = round(cf, 4)
            line["value_covered_pixels"] = round(value * cf, 3) if world == 1 else None
            line["covered_note"] = ("%d of the %d pixels hold scene geometry; the rest take the lighting shader's constant colour for a cleared GBuffer "
                                    "(value-preserving); value_covered_pixels = covered pixels / frame time" % (stats["covered_pixels"], owned_px if world > 1 else W * H))
        if rank_rows is not None:
            line["ranks"] = rank_rows
        if c4_block is not None:
            line["config4_tiles"] = c4_block
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline and args.config == 3:     # defined on the metric's workload only
            line["cpu_baseline"] = cpu_baseline(args.instances, args.cube_dim)
        print(json.dumps(line), flush=True)
    dr.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


def gather_json(obj, world, dist, torch):
    """every rank's small JSON object on every rank (CPU tensors: gloo, like the barrier - nothing here may make torch open an RCCL
    communicator of its own beside the library's)"""
    raw = json.dumps(obj).encode()
    buf = torch.zeros(8192, dtype=torch.uint8)
    buf[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [json.loads(bytes(o.tolist()).rstrip(b"\0").decode()) for o in outs]


def config4_tiles_block(args, world, rank, local_rank, rehearsal, timed_loop, gather_rows, steps=40, warmup=8):
    """BASELINE config 4 on the job's ranks with `--shadow-tiles` semantics: ms per frame (MAX over ranks, barriers as in the main loop)
    and every rank's own row.  ZR_BENCH_C4_INSTANCES shrinks the scene for rehearsals on one GPU."""
    from zeldaengine_amd import dist as zdist, engine, scenes
    n_inst = int(os.environ.get("ZR_BENCH_C4_INSTANCES", "1000000"))
    cfg = scenes.config4(n_inst, 16, cube_dim=args.cube_dim)
    W, H = cfg["width"], cfg["height"]
    native = not args.python_dist and not rehearsal
    dr = zdist.make_distributed(W, H, 1024, device_index=local_rank, rank=rank, world=world, flags=0, split_shadow="tiles", native=native)
    try:
        engine.load_scene(dr.r, cfg)
        dr.r.set_timing_interval(max(1, steps // 8))

        def uniforms(dr_, i):
            dr_.r.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)
        el = timed_loop(dr, steps, warmup, uniforms)
        per = dr.r.frame_periods(steps - 1)
        st = dr.r.stats()
        row = {"rank": rank, "loop_ms_per_frame": round(timed_loop.local / steps * 1e3, 4),
               "frame_gpu_ms_median": round(pct(per, 0.5), 4) if per else None,
               "passes_ms": {k: round(v, 4) for k, v in dr.r.pass_times(min(8, steps)).items()},
               "survivors": st["survivors"], "overflow": st["overflow"]}
        rows = gather_rows(row)
        fallback = getattr(dr, "native_fallback", None)
        return {"workload": "BASELINE config 4: %d instanced 960-triangle spheres (%d meshlet-instances), %dx%d, 1 directional + 16 point lights, 1024^2 shadow map"
                            % (n_inst, st["work_items"][1], W, H),
                "shadow_mode": "tiles (the map owned by light-space super-tiles; + one all-gather of the packed shadow tiles)",
                "host": "torch.distributed loop" if (args.python_dist or rehearsal or fallback) else "native RCCL host (zr_dist_*)",
                "native_fallback": fallback, "frames": steps, "warmup": warmup,
                "ms_per_frame": round(el / steps * 1e3, 4), "value_mpixels_s": round(W * H * steps / el / 1e6, 3),
                "note": "whole-job time per frame (MAX over ranks); never `value` - the metric is quoted on config 3",
                "per_rank": rows}
    finally:
        dr.close()


def launch_ranks(n):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <same arguments>` as a child, pass its stdout /
    stderr through, return its exit code (what the driver's launcher does for N > 1)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:       # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def cpu_baseline(n_inst, cube_dim):
    """The scalar CPU oracle on the same workload (1 frame, 1 core; then the per-pixel stages on all cores)."""
    from oracle import pyoracle
    from zeldaengine_amd import scenes
    pyoracle.build()
    cfg = scenes.config3(n_inst, cube_dim=cube_dim)
    o = pyoracle.Oracle(cfg["width"], cfg["height"], 1024)
    pyoracle.load_scene(o, cfg)
    t0 = time.perf_counter()
    o.render()
    dt = time.perf_counter() - t0
    # the same frame again on an OpenMP team of all host cores: the per-pixel stages (GBuffer resolve, lighting) by rows, the
    # rasteriser by contiguous slices of the draw sequence into per-thread targets merged in draw order.  Same pixels.
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
            "note_all_cores": "rasteriser (draw slices, merged in draw order) and per-pixel stages on an OpenMP team; %.2f s" % dt_all}


if __name__ == "__main__":
    main()
