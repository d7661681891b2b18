"""Multi-GPU path on CPU: the screen-tile partition + all-gather + untile logic with world_size 2 over gloo, and both ways of sharing the
shadow pass (casters split by instance + MIN all-reduce; the map owned by light-space super-tiles + all-gather of the packed tiles).

Each rank stands in for a GPU: it takes the frame the oracle rendered, keeps only the tiles it owns (pack_tiles = the
layout k_lighting writes), all-gathers the packed buffers and untiles (= k_untile).  The result must be the full frame on
every rank.  On the GPU box the same layout is exercised through the kernels (tests/test_gpu_tiles.py).
"""
import os
import socket

import numpy as np
import pytest

from zeldaengine_amd import dist as zdist


def test_layout_arithmetic():
    lay = zdist.tile_layout(1920, 1080, 1)
    assert lay == {"tiles_x": 60, "tiles_y": 34, "n_tiles": 2040, "slots_per_rank": 2040}
    assert zdist.tile_layout(3840, 2160, 8)["n_tiles"] == 8160
    for W, H in ((1920, 1080), (3840, 2160), (416, 250)):
        tx, ty = (W + 31) // 32, (H + 31) // 32
        for world in (1, 2, 3, 4, 8):
            owned = [zdist.owned_tiles(r, world, tx, ty) for r in range(world)]
            assert sorted(sum(owned, [])) == list(range(tx * ty))                     # a partition
            lay = zdist.tile_layout(W, H, world)
            assert max(map(len, owned)) == lay["slots_per_rank"]
            if W >= 1920:     # super-tiles of 128 px: shares within 25 % of each other at the benchmark sizes
                assert max(map(len, owned)) <= 1.25 * min(map(len, owned)), (W, H, world, list(map(len, owned)))
            # neighbouring super-tiles go to different ranks (the skew keeps columns from lining up)
            if world > 1:
                assert zdist.tile_owner(0, 0, world) != zdist.tile_owner(4, 0, world)
                assert world == 3 or zdist.tile_owner(0, 0, world) != zdist.tile_owner(0, 4, world)


def test_python_partition_is_the_librarys():
    """dist.py's numpy statement of the ownership and the C library's zr_tile_partition / zr_tile_owner must be one function."""
    from zeldaengine_amd import engine
    L = engine.lib()
    for W, H in ((1920, 1080), (3840, 2160), (352, 208), (33, 31)):
        tx, ty = (W + 31) // 32, (H + 31) // 32
        for world in (1, 2, 3, 5, 8):
            for r in range(world):
                owned, spr = engine.tile_partition(W, H, world, r)
                assert list(owned) == zdist.owned_tiles(r, world, tx, ty)
                assert spr == zdist.tile_layout(W, H, world)["slots_per_rank"]
            assert all(L.zr_tile_owner(x, y, world) == zdist.tile_owner(x, y, world) for x in range(0, tx, 3) for y in range(0, ty, 2))


@pytest.mark.parametrize("shape", [(64, 64), (100, 70), (33, 31)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_pack_untile_roundtrip(shape, world):
    rng = np.random.default_rng(7)
    frame = rng.integers(0, 256, size=(shape[1], shape[0], 4), dtype=np.uint8)
    gathered = np.stack([zdist.pack_tiles(frame, r, world) for r in range(world)])
    assert np.array_equal(zdist.untile(gathered, shape[0], shape[1]), frame)


def _rank_main(rank, world, port, W, H, q):
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    from zeldaengine_amd import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = scenes.config3(60, W, H)
    o = pyoracle.Oracle(W, H, 128)
    pyoracle.load_scene(o, cfg)
    o.render()
    full = o.color()
    mine = torch.from_numpy(zdist.pack_tiles(full, rank, world).reshape(-1).copy())
    gathered = torch.empty(mine.numel() * world, dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, mine)
    lay = zdist.tile_layout(W, H, world)
    frame = zdist.untile(gathered.numpy().reshape(world, lay["slots_per_rank"], 32, 32, 4), W, H)
    # shadow pass split: this rank draws instances i % world == rank; MIN all-reduce of the maps = the whole shadow map
    part = pyoracle.Oracle(W, H, 128)
    pcfg = dict(cfg)
    pcfg["objects"] = [{"mesh": cfg["objects"][0]["mesh"], "instances": cfg["objects"][0]["instances"][rank::world]}]
    pyoracle.load_scene(part, pcfg)
    part.render(0, 1)
    sh = torch.from_numpy(part.shadowmap().reshape(-1).copy())
    dist.all_reduce(sh, op=dist.ReduceOp.MIN)
    shadow_ok = bool(np.array_equal(sh.numpy().view(np.uint32), o.shadowmap().reshape(-1).view(np.uint32)))
    # shadow MAP owned by light-space super-tiles (zr_set_shadow_tiles): this rank draws the casters whose texel footprint can reach a tile
    # it owns, packs the owned tiles (zr_shadow_pack), the tiles are all-gathered and scattered (zr_shadow_unpack): no reduction
    SD = 256                            # 2 x 2 super-tiles of 128 texels: both ranks own some
    whole = pyoracle.Oracle(W, H, SD)
    pyoracle.load_scene(whole, cfg)
    whole.render(0, 1)
    full_map = whole.shadowmap()
    slay = zdist.tile_layout(SD, SD, world)
    mine_tiles = set(zdist.owned_tiles(rank, world, slay["tiles_x"], slay["tiles_y"]))
    inst = cfg["objects"][0]["instances"]
    keep = []
    for i in range(len(inst)):          # a caster is this rank's when it changes a texel of an owned tile (found with the oracle itself)
        one = pyoracle.Oracle(W, H, SD)
        ocfg = dict(cfg); ocfg["objects"] = [{"mesh": cfg["objects"][0]["mesh"], "instances": inst[i:i + 1]}]
        pyoracle.load_scene(one, ocfg)
        one.render(0, 1)
        ys, xs = np.nonzero(one.shadowmap() < 1.0)
        if any(((y // 32) * slay["tiles_x"] + x // 32) in mine_tiles for y, x in zip(ys, xs)):
            keep.append(i)
        one.close()
    share = pyoracle.Oracle(W, H, SD)
    scfg = dict(cfg); scfg["objects"] = [{"mesh": cfg["objects"][0]["mesh"], "instances": inst[keep]}]
    pyoracle.load_scene(share, scfg)
    share.render(0, 1)
    spacked = torch.from_numpy(zdist.pack_tiles(share.shadowmap(), rank, world, pad=1.0).reshape(-1).copy())
    sgathered = torch.empty(spacked.numel() * world, dtype=torch.float32)
    dist.all_gather_into_tensor(sgathered, spacked)
    smap = zdist.untile(sgathered.numpy().reshape(world, slay["slots_per_rank"], 32, 32), SD, SD)
    shadow_ok = shadow_ok and bool(np.array_equal(smap.view(np.uint32), full_map.view(np.uint32))) and 0 < len(keep) < len(inst)
    q.put((rank, bool(np.array_equal(frame, full)) and shadow_ok, int(mine.numel())))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_allgather_composite():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    W, H, world = 160, 96, 2
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == res[1][2] == zdist.tile_layout(W, H, world)["slots_per_rank"] * 32 * 32 * 4
