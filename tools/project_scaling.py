#!/usr/bin/env python3
"""Scaling PROJECTION from one GPU (no N-GPU node is available to this repository: nothing here is a measured scaling curve).

    python3 tools/project_scaling.py [--configs 3,4,5] [--ranks 2,4,8] [--frames 30] [--modes replicated,tiles,split] > profiles/<tag>_scaling_projection.json

For every config and N, every rank's context (tile_rank = r, tile_world = N: the super-tile partition, owned-region reject, rank-local
work lists, per-rank Hi-Z and occlusion history - exactly what rank r of an N-GPU job runs) renders the frame ALONE on this GPU; the
projected frame time of the job is the slowest rank's, plus what the collectives would take over xGMI where they are not hidden:

  * all-gather of the packed RGBA8 frame tiles (4 B x W x H in total): zr_dist_frame overlaps it with the next frame's rendering
    (double-buffered), so it bounds the period only when it is longer than a frame;
  * `tiles` mode (ZR_DIST_SHADOW_TILES): all-gather of the packed shadow tiles, 4 MiB / N per rank.  NOT hidden: it sits between this
    rank's shadow pass and its lighting pass, so the host lane of a frame is shadow pipeline + pack + all-gather + unpack + lighting
    and the period is the longer of that lane and the measured two-lane period.  The pack and unpack kernels ARE executed here (the
    timed frame is render_geometry, zr_shadow_pack, zr_shadow_unpack, render_lighting); the all-gather between them is modelled;
  * `split` mode (ZR_DIST_SPLIT_SHADOW): all-reduce(min) of the whole 4 MiB map in the same place.

Link model (ASSUMPTIONS, stated in the output): 153 GB/s per xGMI link and direction; a collective costs RCCL_HOP_US per step on top of
its bytes.  All-gather: `direct` = every rank sends its slice to its N - 1 peers over N - 1 of its 7 links at once, one step;
`ring` = N - 1 steps of one slice each.  All-reduce: ring, 2 (N - 1) steps of 1 / N of the map.  The projection uses the RING figures
(the slower ones); the direct ones are printed beside them.
Modes: `replicated` = every rank renders the whole shadow map (the frame all-gather is the only collective: bench.py's default);
`tiles` = the map owned by light-space super-tiles (zr_set_shadow_tiles): a rank draws the casters that reach its tiles;
`split` = rank r rasterises the casters i % N == r (zr_set_shadow_partition).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
XGMI_LINK_GBS = 153.0
RCCL_HOP_US = 6.0          # assumed latency per step of a small RCCL collective on xGMI (launch + one hop)


class RankContext:
    """One rank's context, loaded once, timed under every shadow mode (the modes switch at run time)."""

    def __init__(self, engine, cfg, rank, world, flags=0):
        import torch
        self.torch = torch
        self.cfg, self.rank, self.world = cfg, rank, world
        self.g = engine.Renderer(cfg["width"], cfg["height"], 1024, tile_rank=rank, tile_world=world, flags=flags)
        engine.load_scene(self.g, cfg)
        self.i = 0
        self.bufs = None

    def set_mode(self, mode):
        g = self.g
        g.set_shadow_tiles(0, 1); g.set_shadow_partition(0, 1)
        self.mode = mode
        if self.world > 1 and mode == "split":
            g.set_shadow_partition(self.rank, self.world)
        if self.world > 1 and mode == "tiles":
            g.set_shadow_tiles(self.rank, self.world)
            nb = g.shadow_tiles_bytes()
            dev = self.torch.device("cuda", 0)
            self.bufs = (self.torch.ones(nb // 4, dtype=self.torch.float32, device=dev), self.torch.ones(nb // 4 * self.world, dtype=self.torch.float32, device=dev))

    def frame(self):
        c, g = self.cfg, self.g
        g.update_uniforms(c["camera"], c["dir"], c["point"], c["spot"], 0.0, 0.002 * self.i, 0.016 * self.i)
        self.i += 1
        if self.world > 1 and self.mode == "tiles":
            # (the gathered buffer holds depth 1.0 everywhere: the lighting pass then reads an empty map - same kernel, same bytes)
            g.render_geometry(); g.shadow_pack(self.bufs[0].data_ptr()); g.shadow_unpack(self.bufs[1].data_ptr()); g.render_lighting()
        else:
            g.render()

    def close(self):
        self.g.close()


def verify(ctx, mode, ref_color, ref_shadow, zdist, np):
    """A rank context's time counts only if it draws what the single context draws: its packed frame tiles (replicated / split: after a
    whole frame; the split map is a partial one and is not compared) and, with the map owned by tiles, its packed shadow tiles."""
    g, r, n = ctx.g, ctx.rank, ctx.world
    keep = ctx.i
    ctx.i = 1000
    if mode == "tiles":
        c = ctx.cfg
        g.update_uniforms(c["camera"], c["dir"], c["point"], c["spot"], 0.0, 0.002 * ctx.i, 0.016 * ctx.i)
        g.render_geometry(); g.shadow_pack(ctx.bufs[0].data_ptr()); ctx.torch.cuda.synchronize()
        lay = zdist.tile_layout(ref_shadow.shape[1], ref_shadow.shape[0], n)
        got = ctx.bufs[0].cpu().numpy().view(np.uint32).reshape(lay["slots_per_rank"], 32, 32)
        want = zdist.pack_tiles(ref_shadow, r, n, pad=np.uint32(0x3F800000))
        assert np.array_equal(got, want), "rank %d of %d: %d texels of its owned shadow tiles differ from the single context's map" % (r, n, int((got != want).sum()))
        g.shadow_unpack(ctx.bufs[1].data_ptr()); g.render_lighting(); g.finish()      # (an empty map: the lit tiles are not comparable here)
    else:
        ctx.frame(); g.finish()
        if mode == "replicated":
            got, want = g.read_tiles(), zdist.pack_tiles(ref_color, r, n)
            assert np.array_equal(got, want), "rank %d of %d: %d pixels of its tiles differ from the single context's frame" % (r, n, int((got != want).any(axis=-1).sum()))
    ctx.i = keep


def time_two_lanes(ctx, frames, warmup=8):
    g = ctx.g
    g.set_timing_interval(0)
    for _ in range(warmup):
        ctx.frame()
    g.finish()
    t0 = time.perf_counter()
    for _ in range(frames):
        ctx.frame()
    g.finish()
    wall = (time.perf_counter() - t0) / frames * 1e3
    # the GPU's own frame period (time between the ends of consecutive frames), median over the timed frames: a host hiccup in one of a
    # few dozen frames does not move it; the wall-clock mean is kept beside it
    per = sorted(g.frame_periods(frames - 1))
    ms = per[len(per) // 2] if per else wall
    st = g.stats()
    return ms, {"survivors": st["survivors"], "overflow": st["overflow"], "wall_ms": round(wall, 4)}


def time_alone(ctx):
    """the host lane's passes (shadow pipeline, lighting) each ALONE on the GPU (one stream): what the lane needs when nothing holds it
    up - the two-lane pass timers include the waits for the other lane and for the previous frame's lighting and cannot be added up"""
    g = ctx.g
    g.set_timing_interval(1)
    for _ in range(10):
        ctx.frame()
    g.finish()
    pt = g.pass_times(6)
    return round(pt["cull_shadow"] + pt["shadow"] + pt["lighting"], 4), {k: round(v, 4) for k, v in pt.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="3,4,5")
    ap.add_argument("--ranks", default="2,4,8")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--modes", default="replicated,tiles,split")
    args = ap.parse_args()
    import numpy as np
    from zeldaengine_amd import abi, dist as zdist, engine, scenes
    modes = args.modes.split(",")
    out = {"what": "PROJECTION from rank contexts timed one at a time on ONE MI355X - not a measured scaling curve (see the module docstring)",
           "verified": "before it is timed, every rank context's packed frame tiles (replicated) / packed shadow tiles (tiles) were compared with the single context's frame / map: identical, or this file would not exist",
           "xgmi_link_gbs": XGMI_LINK_GBS, "assumed_rccl_latency_us_per_step": RCCL_HOP_US, "collectives_priced_as": "ring", "configs": {}}
    for c in [int(x) for x in args.configs.split(",")]:
        n_point = 256 if c == 5 else 16
        cfg = scenes.config3(10000, cube_dim=64) if c == 3 else scenes.config4(1000000, n_point, cube_dim=64)
        W, H = cfg["width"], cfg["height"]
        frames = args.frames if c == 3 else max(8, args.frames // 3)
        one_ctx = RankContext(engine, cfg, 0, 1); one_ctx.set_mode("replicated")
        one, _ = time_two_lanes(one_ctx, frames)
        # what every rank context below must reproduce on its own tiles before its time counts: the single context's frame and shadow map
        # at fixed uniforms (frame index 1000: the point lights somewhere along their spiral)
        one_ctx.i = 1000; one_ctx.frame(); one_ctx.g.finish()
        ref_color, ref_shadow = one_ctx.g.color(), one_ctx.g.shadowmap().view(np.uint32).copy()
        one_ctx.close()
        entry = {"resolution": [W, H], "n_gpus_1_ms": round(one, 4), "ranks": {}}
        sys.stderr.write("config %d N 1: %.3f ms\n" % (c, one))
        for n in [int(x) for x in args.ranks.split(",")]:
            lat = RCCL_HOP_US * 1e-3
            fslice = 4.0 * W * H / n
            ag_direct = fslice / (XGMI_LINK_GBS * 1e9) * 1e3 + lat
            ag_ring = (n - 1) * (fslice / (XGMI_LINK_GBS * 1e9) * 1e3 + lat)
            sslice = 4.0 * 1024 * 1024 / n
            sg_direct = sslice / (XGMI_LINK_GBS * 1e9) * 1e3 + lat
            sg_ring = (n - 1) * (sslice / (XGMI_LINK_GBS * 1e9) * 1e3 + lat)
            ar = 2.0 * (n - 1) * (sslice / (XGMI_LINK_GBS * 1e9) * 1e3 + lat)
            row = {"frame_allgather_ms": {"direct": round(ag_direct, 4), "ring": round(ag_ring, 4)},
                   "shadow_tiles_allgather_ms": {"direct": round(sg_direct, 4), "ring": round(sg_ring, 4)}, "shadow_allreduce_ms_ring": round(ar, 4)}
            res = {m: {"per": [], "lane": [], "surv": []} for m in modes}
            for r in range(n):
                two = RankContext(engine, cfg, r, n)
                alone = RankContext(engine, cfg, r, n, flags=abi.FLAG_SERIAL_PASSES)
                for m in modes:
                    two.set_mode(m); alone.set_mode(m)
                    verify(two, m, ref_color, ref_shadow, zdist, np)
                    ms, st = time_two_lanes(two, frames)
                    lane, _ = time_alone(alone)
                    res[m]["per"].append(round(ms, 4)); res[m]["lane"].append(lane); res[m]["surv"].append(st["survivors"][0])
                    sys.stderr.write("config %d N %d %s rank %d: %.3f ms, host lane alone %.3f, shadow survivors %d %s\n" % (c, n, m, r, ms, lane, st["survivors"][0], st))
                two.close(); alone.close()
            for m in modes:
                per, lane = res[m]["per"], res[m]["lane"]
                on_lane = sg_ring if m == "tiles" else ar if m == "split" else 0.0
                crit = max(max(per), max(lane) + on_lane) if on_lane else max(per)
                period = max(crit, ag_ring)
                row[m] = {"per_rank_ms": per, "slowest_rank_ms": max(per), "host_lane_alone_ms": lane, "shadow_survivors": res[m]["surv"],
                          "collective_on_host_lane_ms": round(on_lane, 4), "frame_allgather_ms_ring_hidden": round(ag_ring, 4),
                          "projected_frame_ms": round(period, 4), "projected_speedup": round(one / period, 3),
                          "projected_efficiency": round(one / period / n, 3)}
            entry["ranks"][str(n)] = row
        out["configs"]["config%d" % c] = entry
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
