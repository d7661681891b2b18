"""GPU: screen-tile partition through the kernels, world JSON + Profabs, the livelink server, full-size properties."""
import json
import time

import numpy as np
import pytest

from parity_util import compare_all
from zeldaengine_amd import abi, dist as zdist, livelink, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_tile_partition_composites_to_the_single_gpu_frame(gpu_engine, world):
    """Rank contexts on one GPU (run in turn), packed tiles gathered on the host, composited by k_untile."""
    import torch
    cfg = scenes.config3(300, 416, 250)
    single = gpu_engine.Renderer(cfg["width"], cfg["height"], 256)
    gpu_engine.load_scene(single, cfg)
    single.render()
    want = single.color()
    packs, ranks = [], []
    for r in range(world):
        g = gpu_engine.Renderer(cfg["width"], cfg["height"], 256, tile_rank=r, tile_world=world)
        gpu_engine.load_scene(g, cfg)
        g.render()
        packs.append(g.read_tiles())
        ranks.append(g)
        assert np.array_equal(packs[-1], zdist.pack_tiles(want, r, world))          # the packed layout is the documented one
    gathered = torch.from_numpy(np.stack(packs).reshape(-1).copy()).cuda()
    for g in ranks:
        g.composite(gathered.data_ptr())
        assert np.array_equal(g.color(), want)
    st = [g.stats() for g in ranks]
    assert sum(s["covered_pixels"] for s in st) == single.stats()["covered_pixels"]


def test_pipelined_distributed_renderer_with_a_stand_in_collective(gpu_engine, monkeypatch):
    """The two-stream, double-buffered frame loop of dist.DistributedRenderer for rank 0 of 2, with torch.distributed's
    all-gather replaced by a local stand-in that supplies rank 1's packed tiles (one GPU here; RCCL itself is not under test)."""
    import torch
    import torch.distributed as tdist
    cfg = scenes.config3(300, 416, 250)
    single = gpu_engine.Renderer(cfg["width"], cfg["height"], 256)
    gpu_engine.load_scene(single, cfg)
    single.render()
    want = single.color()
    other = gpu_engine.Renderer(cfg["width"], cfg["height"], 256, tile_rank=1, tile_world=2)
    gpu_engine.load_scene(other, cfg)
    other.render()
    other_tiles = torch.from_numpy(other.read_tiles().reshape(-1).copy()).cuda()
    calls = []

    other.set_shadow_partition(1, 2)     # rank 1's share of the shadow casters (its lit tiles above used the full map)
    other.render_shadow()
    other_shadow = torch.from_numpy(other.shadowmap().reshape(-1).copy()).cuda()
    other.render_gbuffer(); other.render_lighting()
    assert (other.shadowmap() >= single.shadowmap()).all() and (other.shadowmap() > single.shadowmap()).any()

    def fake_all_gather(out, inp, group=None, async_op=False):
        n = inp.numel()
        out[:n].copy_(inp)              # on the current (collective) stream, like the real op
        out[n:2 * n].copy_(other_tiles)
        calls.append(torch.cuda.current_stream().cuda_stream)

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        assert op == tdist.ReduceOp.MIN
        torch.minimum(t, other_shadow, out=t)
        calls.append(-1)
    monkeypatch.setattr(tdist, "all_gather_into_tensor", fake_all_gather)
    monkeypatch.setattr(tdist, "all_reduce", fake_all_reduce)
    dr = zdist.DistributedRenderer(cfg["width"], cfg["height"], 256, device_index=0, rank=0, world=2, split_shadow=True)
    gpu_engine.load_scene(dr.r, cfg)
    for _ in range(5):                  # exercises both halves of the double buffers
        dr.frame()
    dr.synchronize()
    gathers = [c for c in calls if c != -1]
    assert len(gathers) == 5 and calls.count(-1) == 5 and all(c == dr.comm_stream.cuda_stream for c in gathers)
    assert dr.comm_stream != dr.render_stream
    assert np.array_equal(dr.r.color(), want)
    assert np.array_equal(dr.r.shadowmap().view(np.uint32), single.shadowmap().view(np.uint32))     # min of the two shares = the whole map
    assert np.array_equal(dr.tiles[0].cpu().numpy(), dr.tiles[1].cpu().numpy())
    assert np.array_equal(dr.tiles[0].cpu().numpy().reshape(-1, 32, 32, 4), zdist.pack_tiles(want, 0, 2))
    dr.close()


def _register_sample_profabs(r):
    plane = r.mesh_create(*scenes.grid_plane(20.0, 4, 0.0))
    box = r.mesh_create(*scenes.box((0.5, 0.5, 0.5), (0, 0, 0.5)))
    sph = r.mesh_create(*scenes.uv_sphere())
    r.profab_register("terrain", plane)
    r.profab_register("rock_01", box)
    r.profab_register("rock_02", box)
    r.profab_register("grass_01", sph)
    return {"terrain": plane, "rock_01": box, "rock_02": box, "grass_01": sph}


def _oracle_from_renderer(oracle_lib, g, meshes, W, H, SD, cubemap):
    """Rebuild the renderer's scene on the oracle from what the library reports (instances it generated itself)."""
    o = oracle_lib.Oracle(W, H, SD)
    o.set_cubemap(cubemap)
    cache = {}
    for i in range(g.object_count()):
        mesh_id, inst = g.object_get_instances(i)
        if mesh_id not in cache:
            cache[mesh_id] = o.mesh_create(*meshes[mesh_id])
        o.object_add(cache[mesh_id], None, inst)
    o.set_frame(*g.get_frame())
    return o


def test_world_json_builds_the_scene_from_registered_profabs(oracle_lib, gpu_engine):
    W, H, SD = 384, 216, 256
    g = gpu_engine.Renderer(W, H, SD)
    cube = scenes.synthetic_cubemap(16)
    g.set_cubemap(cube)
    ids = _register_sample_profabs(g)
    world = scenes.sample_world()
    world["Objects"][3]["InstanceCount"] = 700                   # keep the oracle fast; grass_02 has no Profab -> draws nothing
    g.world_load_json(json.dumps(world))
    assert g.object_count() == 4
    counts = [None if g.object_get_instances(i)[1] is None else len(g.object_get_instances(i)[1]) for i in range(4)]
    assert counts == [None, None, 64, 700]                       # InstanceCount <= 1 -> non-instanced draw (ZE:4250-4267)
    cam = g.world_camera()
    assert list(cam.Lookat) == [0.0, 0.0, 0.5] and cam.FOV == 45.0
    saved = json.loads(g.world_save_json())
    assert saved["Objects"][3]["InstanceCount"] == 700 and len(saved["PointLights"]) == 16
    g.render(); g.finish()
    meshes = {ids["terrain"]: scenes.grid_plane(20.0, 4, 0.0), ids["rock_01"]: scenes.box((0.5, 0.5, 0.5), (0, 0, 0.5)),
              ids["grass_01"]: scenes.uv_sphere()}
    o = _oracle_from_renderer(oracle_lib, g, meshes, W, H, SD, cube)
    o.render()
    bad = {k: v for k, v in compare_all(o, g).items() if v}
    assert not bad, bad
    inst = g.object_get_instances(3)[1]
    d = np.linalg.norm(inst["InstancePosition"][:, :2], axis=1)
    assert d.min() >= 2.0 - 1e-5 and d.max() <= 8.0 + 1e-5 and (inst["InstancePosition"][:, 2] == 0).all()
    assert inst["InstancePScale"].min() >= 0.1 and inst["InstancePScale"].max() <= 0.5
    assert (inst["InstanceRotation"][:, 0] == 0).all() and inst["InstanceRotation"][:, 1].max() <= np.pi * 180.0


def test_profab_tree_on_disk_drops_in(oracle_lib, gpu_engine, tmp_path):
    """A content tree in the engine's layout (Profabs/<name>/{models,textures}) + a World.json -> frame equal to the oracle's."""
    from PIL import Image
    from zeldaengine_amd import assets
    root = tmp_path / "Profabs"
    meshes = {"terrain": scenes.grid_plane(18.0, 5, 0.0), "rock_02": scenes.uv_sphere(16, 8, 0.5)}
    for name, (v, idx) in meshes.items():
        (root / name / "models").mkdir(parents=True)
        (root / name / "textures").mkdir(parents=True)
        assets.write_obj(str(root / name / "models" / (name + ".obj")), v, idx)
    yy, xx = np.mgrid[0:32, 0:32]
    chk = np.where(((xx // 4 + yy // 4) & 1)[..., None].astype(bool), np.uint8(210), np.uint8(60)) * np.ones((1, 1, 3), np.uint8)
    Image.fromarray(chk.astype(np.uint8)).save(str(root / "terrain" / "textures" / "terrain_bc.png"))
    Image.fromarray(np.full((8, 8, 3), 128, np.uint8)).save(str(root / "rock_02" / "textures" / "rock_02_r.png"))
    W, H, SD = 320, 180, 128
    g = gpu_engine.Renderer(W, H, SD)
    ids = assets.register_profabs(g, str(root))
    assert sorted(ids) == ["rock_02", "terrain"]
    wpath = tmp_path / "World.json"
    wpath.write_text(json.dumps(scenes.sample_world()))
    assets.world_load_file(g, str(wpath))
    assert g.object_count() == 2                                # rock_01, grass_* have no directory: nothing drawn
    g.render(); g.finish()
    assets.world_save_file(g, str(tmp_path / "Saved.json"))
    assert json.loads((tmp_path / "Saved.json").read_text())["Objects"][2]["ProfabName"] == "rock_02"
    # oracle: same files, same instances
    o = oracle_lib.Oracle(W, H, SD)
    for i in range(g.object_count()):
        mesh_id, inst = g.object_get_instances(i)
        name = [n for n, m in ids.items() if m == [mesh_id]][0]
        v, idx = assets.load_obj(str(root / name / "models" / (name + ".obj")))
        tex = assets.find_profabs(str(root))[name][0][1]
        mat, keep = abi.make_material([assets.load_image_rgba8(p) if p else None for p in tex])
        o.object_add(o.mesh_create(v, idx), mat, inst)
    o.set_frame(*g.get_frame())
    o.render()
    bad = {k: v for k, v in compare_all(o, g).items() if v}
    assert not bad, bad
    assert len(np.unique(g.gbuffer(4))) > 50                    # the checker texture reached the GBuffer


def test_livelink_end_to_end(gpu_engine):
    """The reference client's bytes over TCP -> listener thread -> poll on the render thread -> new scene (ZE:1617-1710)."""
    g = gpu_engine.Renderer(128, 96, 64)
    _register_sample_profabs(g)
    port = g.livelink_serve(0)
    assert port > 0 and not g.livelink_poll()
    world = scenes.sample_world()
    world["Objects"][3]["InstanceCount"] = 50
    assert livelink.send_world(world, port=port, host="127.0.0.1") == b""
    for _ in range(100):
        if g.livelink_poll():
            break
        time.sleep(0.02)
    else:
        pytest.fail("the livelink payload never reached the render thread")
    assert g.object_count() == 4
    g.render(); g.finish()
    assert g.stats()["covered_pixels"] > 1000
    # a malformed payload is logged and ignored; the server keeps serving (the engine would terminate, ZE:1071-1073)
    livelink.sendDataToEngine("{not json", port=port, host="127.0.0.1")
    time.sleep(0.1)
    assert not g.livelink_poll()
    livelink.send_world(scenes.sample_world() | {"Objects": []}, port=port, host="127.0.0.1")
    for _ in range(100):
        if g.livelink_poll():
            break
        time.sleep(0.02)
    assert g.object_count() == 0
    g.livelink_stop()


def test_livelink_silent_client_and_oversized_scene_do_not_hurt(gpu_engine):
    """A client that connects and sends nothing must not park the listener (zr_livelink_stop / zr_destroy join it), and a payload
    asking for billions of instances is a schema error, not an allocation."""
    import socket
    g = gpu_engine.Renderer(64, 64, 64)
    _register_sample_profabs(g)
    port = g.livelink_serve(0)
    quiet = socket.create_connection(("127.0.0.1", port))
    time.sleep(0.3)                          # the listener is now inside the connection, waiting for the first segment
    t0 = time.time()
    g.livelink_stop()
    assert time.time() - t0 < 1.0
    quiet.close()
    port = g.livelink_serve(0)               # serving again works, and a later client is not blocked by an earlier silent one
    quiet = socket.create_connection(("127.0.0.1", port))
    huge = scenes.sample_world()
    huge["Objects"][3]["InstanceCount"] = 4000000000
    livelink.send_world(huge, port=port, host="127.0.0.1")      # queued behind the silent client: served once that one is dropped
    time.sleep(0.3)                          # (send_world returned after the half-close, i.e. after the payload was parsed)
    assert not g.livelink_poll()             # rejected at parse time
    with pytest.raises(gpu_engine.ZeldaRenderError) as e:
        g.world_load_json(json.dumps(huge))
    assert e.value.code == abi.ERR_PARSE and "InstanceCount" in str(e.value)
    quiet.close()
    g.close()                                # zr_destroy stops the listener


def test_full_size_config3_against_the_oracle(oracle_lib, gpu_engine):
    """BASELINE's own size: 10 000 instances / 140 000 meshlet-instances at 1920x1080, every target bit-exact."""
    cfg = scenes.config3()
    o = oracle_lib.Oracle(cfg["width"], cfg["height"], 1024)
    oracle_lib.load_scene(o, cfg)
    o.render()
    g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024)
    gpu_engine.load_scene(g, cfg)
    g.render(); g.finish()
    bad = {k: v for k, v in compare_all(o, g).items() if v}
    assert not bad, bad
    st = g.stats()
    assert st["covered_pixels"] == o.covered_pixels() and st["overflow"] == 0
    assert st["survivors"][1] < 0.7 * st["work_items"][1]          # frustum + cone culling bite
    # size-independent properties: determinism across frames, and culling is invisible at full size too
    first = (g.color(), g.gbuffer(2), g.shadowmap())
    g.render()
    assert np.array_equal(first[0], g.color()) and np.array_equal(first[1], g.gbuffer(2))
    h = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024, flags=abi.FLAG_NO_FRUSTUM_CULL | abi.FLAG_NO_CONE_CULL)
    gpu_engine.load_scene(h, cfg)
    h.render()
    assert np.array_equal(first[0], h.color()) and np.array_equal(first[2].view(np.uint32), h.shadowmap().view(np.uint32))
    assert h.stats()["survivors"][1] > st["survivors"][1]


def test_owned_region_reject_keeps_the_frame_and_cuts_per_rank_work(gpu_engine):
    """Super-tile ownership + the bounding-sphere reject of stage A: a rank transforms (stage B), bins and rasterises only what can
    reach its own screen region.  The packed tiles must not change (reject on / off), and the per-rank survivor count must fall to
    about 1 / world of the single-GPU count instead of staying at it."""
    cfg = scenes.config3(3000, 1280, 720, 9.0)
    single = gpu_engine.Renderer(cfg["width"], cfg["height"], 256, flags=abi.FLAG_NO_HIZ)
    gpu_engine.load_scene(single, cfg)
    single.render()
    want = single.color()
    total = single.stats()["survivors"][1]
    world = 4
    per_rank, per_rank_off = [], []
    for r in range(world):
        g = gpu_engine.Renderer(cfg["width"], cfg["height"], 256, tile_rank=r, tile_world=world, flags=abi.FLAG_NO_HIZ)
        h = gpu_engine.Renderer(cfg["width"], cfg["height"], 256, tile_rank=r, tile_world=world, flags=abi.FLAG_NO_HIZ | abi.FLAG_NO_RECT_CULL)
        for x in (g, h):
            gpu_engine.load_scene(x, cfg)
            x.render()
        assert np.array_equal(g.read_tiles(), zdist.pack_tiles(want, r, world))
        assert np.array_equal(g.read_tiles(), h.read_tiles())
        per_rank.append(g.stats()["survivors"][1]); per_rank_off.append(h.stats()["survivors"][1])
        g.close(); h.close()
    # without the reject every rank carries every frustum / cone survivor through the vertex stage and the binning kernels ...
    assert per_rank_off == [total] * world
    # ... with it a meshlet stays on the one rank (rarely two: the bounding sphere is a loose fit) that owns its super-tile
    assert total <= sum(per_rank) <= 1.45 * total, (total, per_rank)
    assert max(per_rank) < 0.45 * total, (total, per_rank)


def test_native_rccl_host_world_of_one(gpu_engine):
    """zr_dist_* with a communicator of one rank: the library loads librccl itself, ncclCommInitRank / ncclAllGather run on the
    hardware, and five pipelined frames (both halves of the double buffers) composite to the frame a plain context renders."""
    cfg = scenes.config3(300, 416, 250)
    single = gpu_engine.Renderer(cfg["width"], cfg["height"], 256)
    gpu_engine.load_scene(single, cfg)
    single.render()
    want = single.color()
    dr = zdist.NativeDistributedRenderer(cfg["width"], cfg["height"], 256, device_index=0, rank=0, world=1)
    gpu_engine.load_scene(dr.r, cfg)
    for _ in range(5):
        dr.frame()
    dr.synchronize()
    assert np.array_equal(dr.r.color(), want)
    assert np.array_equal(dr.r.read_tiles(), zdist.pack_tiles(want, 0, 1))
    with pytest.raises(gpu_engine.ZeldaRenderError):
        dr.r.dist_init(bytes(128), 0, 1)                  # already initialised
    dr.close()
    # a context without the packed path, or with the wrong rank, is refused before RCCL is touched
    g = gpu_engine.Renderer(64, 64, 64)
    with pytest.raises(gpu_engine.ZeldaRenderError):
        g.dist_init(bytes(128), 0, 1)
    with pytest.raises(gpu_engine.ZeldaRenderError):
        g.dist_init(bytes(128), 1, 2)
    g.close()


def test_tile_partition_with_clipped_triangles_and_hiz_rounds(gpu_engine):
    """Three rank contexts, a scene whose ground plane crosses the near plane and whose wall has triangles longer than 64 pixels
    (the triangle-binned pass's slow list, tried by every owned tile), three frames with the camera moving (Hi-Z rounds with a
    stale history on every rank): the ranks' packed tiles must be the single-GPU frame's, frame after frame."""
    W, H, SD, world = 416, 250, 128, 3

    def scene(r):
        r.set_cubemap(scenes.synthetic_cubemap(16))
        r.object_add(r.mesh_create(*scenes.grid_plane(60.0, 8, 0.0)))
        r.object_add(r.mesh_create(*scenes.box((3.0, 0.15, 1.6), (0.0, 0.0, 1.6))))
        r.object_add(r.mesh_create(*scenes.uv_sphere()), None, scenes.generate_instances(400, 1.0, 14.0, 0.3, 0.7, seed=5))

    single = gpu_engine.Renderer(W, H, SD)
    ranks = [gpu_engine.Renderer(W, H, SD, tile_rank=r, tile_world=world) for r in range(world)]
    for g in [single] + ranks:
        scene(g)
    w = scenes.sample_world()
    d, _, s = scenes.lights_from_world(w)
    w["PointLights"] = scenes.sample_point_lights(8)
    _, p, _ = scenes.lights_from_world(w)
    for i, (pos, look) in enumerate([((0.0, -6.0, 1.2), (0.0, 0.0, 1.0)), ((2.5, -5.0, 1.6), (0.0, 0.0, 1.0)), ((6.0, 1.0, 4.0), (0.0, 0.0, 0.5))]):
        cam = abi.make_camera(pos, look, fov=50.0)
        for g in [single] + ranks:
            g.update_uniforms(cam, d, p, s, 0.1 * i, 0.02 * i, 1.0 + i)
            g.render()
        want = single.color()
        for r, g in enumerate(ranks):
            assert np.array_equal(g.read_tiles(), zdist.pack_tiles(want, r, world)), "frame %d, rank %d" % (i, r)
            assert g.stats()["overflow"] == 0
    for g in [single] + ranks:
        g.close()
