"""Worker of tests/test_gpu_0_multiproc.py: one rank of a world-N DistributedRenderer, every rank on cuda:0, collectives over gloo
(RCCL needs one GPU per rank; the frame loop, the buffers and the collective calls are the same)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from zeldaengine_amd import dist as zdist, engine, scenes
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    cfg = scenes.config3(300, 416, 250)
    split = os.environ.get("ZR_TEST_SPLIT_SHADOW") == "1"     # default: all-gather of the composite is the only collective
    if os.environ.get("ZR_TEST_SPLIT_SHADOW") == "tiles":      # the map owned by light-space super-tiles + an all-gather of the packed tiles
        split = "tiles"
    if os.environ.get("ZR_TEST_NATIVE") == "1":
        # the native RCCL host cannot come up with every rank on one GPU (RCCL wants a device per rank): all ranks must notice, agree
        # and fall back to the torch.distributed loop together
        if os.environ.get("ZR_TEST_NATIVE_FAIL_RANK") == str(rank):
            # ONE rank cannot even prepare (no librccl / no memory there): the others must not be left inside ncclCommInitRank
            def broken(self, *a, **k):
                raise RuntimeError("injected: this rank cannot prepare the native host")
            engine.Renderer.dist_prepare = broken
        dr = zdist.make_distributed(cfg["width"], cfg["height"], 256, device_index=0, rank=rank, world=world, split_shadow=split, native=True)
        fb = getattr(dr, "native_fallback", None)
        print("rank %d native_fallback: %s" % (rank, fb), flush=True)
        assert isinstance(dr, zdist.DistributedRenderer) and fb, "expected the agreed fallback on a one-GPU box"
    else:
        dr = zdist.DistributedRenderer(cfg["width"], cfg["height"], 256, device_index=0, rank=rank, world=world, split_shadow=split)
    engine.load_scene(dr.r, cfg)
    for _ in range(5):                  # both halves of the double buffers, occlusion history in use
        dr.frame()
    dr.synchronize()
    got, got_shadow = dr.r.color(), dr.r.shadowmap()
    single = engine.Renderer(cfg["width"], cfg["height"], 256)
    engine.load_scene(single, cfg)
    single.render()
    ok = np.array_equal(got, single.color()) and np.array_equal(got_shadow.view(np.uint32), single.shadowmap().view(np.uint32))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dr.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("MP_DIST_OK" if int(flag.item()) == 1 else "MP_DIST_MISMATCH", flush=True)
    sys.exit(0 if int(flag.item()) == 1 else 3)


if __name__ == "__main__":
    main()
