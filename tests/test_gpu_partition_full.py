"""GPU: the 8-GPU configurations AT FULL SIZE, partitioned - BASELINE configs 4 and 5 (1 000 000 instances / ~5 M meshlet-instances,
3840 x 2160; 16 and 256 point lights) as the eight rank contexts of an 8-rank job, every one on the one GPU.

Three frames (the second repeats the first one's uniforms: exact visibility / occlusion history; the third moves the shadow-casting
light and the point lights), twice: with the shadow map replicated on every rank (bench.py's default) and owned by light-space tiles
(`--shadow-tiles`, what a node should run).  Every frame: every rank's packed RGBA8 tiles must equal the single context's frame cut into
that rank's tiles, and in tiles mode its packed shadow tiles the single context's map - bit for bit - so the all-gathers (a concatenation
here; tests/test_gpu_dist_native.py runs the real loop) assemble the single-GPU frame.  What each rank kept of the work is asserted too.
"""
import math

import numpy as np
import pytest

from zeldaengine_amd import dist as zdist, scenes

pytestmark = pytest.mark.gpu
WORLD = 8


def _light(i):
    a = 0.785 + 0.09 * i
    return (28.3 * math.cos(a), 28.3 * math.sin(a) - 20.0 * math.sin(a - 0.785), 20.0 + 1.5 * i)


@pytest.mark.parametrize("n_point", [16, 256], ids=["config4", "config5"])
def test_eight_rank_contexts_assemble_the_full_size_frame(gpu_engine, n_point):
    import torch
    cfg = scenes.config4(1000000, n_point, cube_dim=64)
    W, H, SD = cfg["width"], cfg["height"], 1024
    dev = torch.device("cuda", 0)
    single = gpu_engine.Renderer(W, H, SD)
    gpu_engine.load_scene(single, cfg)
    ranks = [gpu_engine.Renderer(W, H, SD, tile_rank=k, tile_world=WORLD) for k in range(WORLD)]
    for r in ranks:
        gpu_engine.load_scene(r, cfg)
    d = cfg["dir"].copy()

    def uniforms(r, i):
        lp = _light(0 if i < 2 else i)
        d[0]["Position"][:3] = lp; d[0]["Direction"][:3] = lp
        r.update_uniforms(cfg["camera"], d, cfg["point"], cfg["spot"], 0.0, 0.0 if i < 2 else 0.05 * i, 0.016 * i)

    want = []
    for i in range(3):
        uniforms(single, i)
        single.render(); single.finish()
        want.append((single.color(), single.shadowmap().view(np.uint32).copy()))
    st1 = single.stats()
    assert st1["overflow"] == 0 and st1["work_items"][1] > 4000000
    assert not np.array_equal(want[1][1], want[2][1]) and np.array_equal(want[0][0], want[1][0])      # the light really moved in frame 2 only
    single.close()
    frame_tiles = [[zdist.pack_tiles(want[i][0], k, WORLD) for k in range(WORLD)] for i in range(3)]
    slay = zdist.tile_layout(SD, SD, WORLD)

    # ---- the shadow map replicated: the frame's all-gather is the only collective
    for i in range(3):
        for k, r in enumerate(ranks):
            uniforms(r, i)
            r.render()
        for k, r in enumerate(ranks):
            r.finish()
            got = r.read_tiles()
            assert np.array_equal(got, frame_tiles[i][k]), "replicated, frame %d, rank %d: %d pixels of its tiles differ" % (i, k, int((got != frame_tiles[i][k]).any(axis=-1).sum()))
            assert np.array_equal(r.shadowmap().view(np.uint32), want[i][1]), "replicated, frame %d, rank %d: shadow map" % (i, k)
    rep = [r.stats() for r in ranks]

    # ---- the map owned by light-space tiles: + one all-gather of the packed shadow tiles
    for k, r in enumerate(ranks):
        r.set_shadow_tiles(k, WORLD)
    nb = ranks[0].shadow_tiles_bytes()
    assert nb == slay["slots_per_rank"] * 32 * 32 * 4
    packed = [torch.ones(nb // 4, dtype=torch.float32, device=dev) for _ in range(WORLD)]
    for i in range(3):
        for k, r in enumerate(ranks):
            uniforms(r, i)
            r.render_geometry()
            r.shadow_pack(packed[k].data_ptr())
        torch.cuda.synchronize()
        for k in range(WORLD):
            got = packed[k].cpu().numpy().view(np.uint32).reshape(slay["slots_per_rank"], 32, 32)
            ref = zdist.pack_tiles(want[i][1], k, WORLD, pad=np.uint32(0x3F800000))
            assert np.array_equal(got, ref), "tiles, frame %d, rank %d: %d texels of its owned shadow tiles differ" % (i, k, int((got != ref).sum()))
        gathered = torch.cat(packed).contiguous()
        torch.cuda.synchronize()
        for r in ranks:
            r.shadow_unpack(gathered.data_ptr())
            r.render_lighting()
        for k, r in enumerate(ranks):
            r.finish()
            got = r.read_tiles()
            assert np.array_equal(got, frame_tiles[i][k]), "tiles, frame %d, rank %d: %d pixels of its tiles differ" % (i, k, int((got != frame_tiles[i][k]).any(axis=-1).sum()))
            assert np.array_equal(r.shadowmap().view(np.uint32), want[i][1]), "tiles, frame %d, rank %d: the scattered map differs" % (i, k)
    til = [r.stats() for r in ranks]
    # what a rank keeps: about an eighth of the camera pass either way; the whole shadow pass when the map is replicated, about an eighth
    # (+ the casters that straddle a super-tile border) when it is owned by tiles
    for k in range(WORLD):
        assert rep[k]["overflow"] == 0 and til[k]["overflow"] == 0
        assert rep[k]["survivors"][1] < 0.30 * st1["survivors"][1] and til[k]["survivors"][1] < 0.30 * st1["survivors"][1], (k, rep[k]["survivors"], st1["survivors"])
        assert rep[k]["survivors"][0] == st1["survivors"][0], (k, rep[k]["survivors"], st1["survivors"])
        assert til[k]["survivors"][0] < 0.25 * st1["survivors"][0], (k, til[k]["survivors"], st1["survivors"])
    assert sum(t["survivors"][0] for t in til) >= st1["survivors"][0]
    for r in ranks:
        r.close()
