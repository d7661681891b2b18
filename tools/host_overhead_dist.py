"""Host time per DistributedRenderer.frame() for rank 0 of 2 on one GPU, with the two collectives replaced by local stand-ins
(what is measured: Python + ctypes + the library's enqueue work + torch's stream/event calls; RCCL's own launch cost is not)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import torch.distributed as tdist
from zeldaengine_amd import dist as zdist, engine, scenes


def fake_all_gather(out, inp, group=None, async_op=False):
    out[:inp.numel()].copy_(inp)


def fake_all_reduce(t, op=None, group=None, async_op=False):
    t.clamp_(max=1.0)


tdist.all_gather_into_tensor = fake_all_gather
tdist.all_reduce = fake_all_reduce
cfg = scenes.config3()
dr = zdist.DistributedRenderer(cfg["width"], cfg["height"], 1024, device_index=0, rank=0, world=2, split_shadow=True)
engine.load_scene(dr.r, cfg)
dr.r.set_timing_interval(0)
for _ in range(20):
    dr.frame()
dr.synchronize()
for n in (50, 200):
    t0 = time.perf_counter()
    for _ in range(n):
        dr.frame()
    t1 = time.perf_counter()
    dr.synchronize()
    t2 = time.perf_counter()
    print("world 2 (stand-in collectives): n=%d enqueue %.1f us/frame, total %.1f us/frame" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
dr.r.set_timing_interval(4)
for _ in range(64):
    dr.frame()
dr.synchronize()
print({k: round(v, 4) for k, v in dr.r.pass_times(16).items()})
print(dr.r.stats())
