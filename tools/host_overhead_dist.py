"""Host time to enqueue one frame: plain zr_render, the native RCCL host (zr_dist_frame, world of one = the same calls as world N)
and the torch.distributed loop of dist.py with local stand-in collectives.  A tiny scene keeps the GPU ahead of the host, so the loop
rate IS the host cost (Python + ctypes + the library's enqueue work [+ torch's stream / event calls])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as tdist
from zeldaengine_amd import dist as zdist, engine, scenes


def rate(frame, sync, n=400):
    for _ in range(40):
        frame()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        frame()
    t1 = time.perf_counter()
    sync()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6


cfg = scenes.config3(16, 256, 144)
r = engine.Renderer(cfg["width"], cfg["height"], 256)
engine.load_scene(r, cfg); r.set_timing_interval(0)
print("plain zr_render:                 enqueue %.1f us/frame (loop %.1f)" % rate(r.render, r.finish))
nd = zdist.NativeDistributedRenderer(cfg["width"], cfg["height"], 256, device_index=0, rank=0, world=1)
engine.load_scene(nd.r, cfg); nd.r.set_timing_interval(0)
print("native host zr_dist_frame:       enqueue %.1f us/frame (loop %.1f)" % rate(nd.frame, nd.synchronize))


def fake_all_gather(out, inp, group=None, async_op=False):
    out[:inp.numel()].copy_(inp)


tdist.all_gather_into_tensor = fake_all_gather
tdist.all_reduce = lambda t, op=None, group=None, async_op=False: t.clamp_(max=1.0)
dr = zdist.DistributedRenderer(cfg["width"], cfg["height"], 256, device_index=0, rank=0, world=2)
engine.load_scene(dr.r, cfg); dr.r.set_timing_interval(0)
print("dist.py loop (stand-in gather):  enqueue %.1f us/frame (loop %.1f)" % rate(dr.frame, dr.synchronize))
