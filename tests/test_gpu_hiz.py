"""GPU parity of the two-pass Hi-Z occlusion culling (camera pass): the frame must not depend on the visibility history.

Round 1 draws what owned a pixel last frame, a depth pyramid of that rejects hidden meshlet-instances, round 2 draws the
rest.  Every frame of a sequence (no history, exact history, stale history after camera motion and after the scene
animates) is compared bit for bit against the oracle, which draws every triangle.
"""
import numpy as np
import pytest

from parity_util import compare_all
from zeldaengine_amd import abi, scenes

pytestmark = pytest.mark.gpu


def _identical(o, g, what=""):
    d = compare_all(o, g)
    bad = {k: v for k, v in d.items() if v}
    assert not bad, "%s: HIP path differs from the oracle: %r" % (what, bad)


def _crowd(r, n=600, seed=5):
    """A wall close to the camera hides most of a crowd of spheres; a ground plane crosses the near plane."""
    r.set_cubemap(scenes.synthetic_cubemap(16))
    r.object_add(r.mesh_create(*scenes.grid_plane(60.0, 8, 0.0)))
    r.object_add(r.mesh_create(*scenes.box((3.0, 0.15, 1.6), (0.0, 0.0, 1.6))))
    r.object_add(r.mesh_create(*scenes.uv_sphere()), None, scenes.generate_instances(n, 1.0, 14.0, 0.3, 0.7, seed=seed))


def _lights():
    w = scenes.sample_world()
    d, _, s = scenes.lights_from_world(w)
    w["PointLights"] = scenes.sample_point_lights(8)
    _, p, _ = scenes.lights_from_world(w)
    return d, p, s


CAMS = [
    ((0.0, -6.0, 1.2), (0.0, 0.0, 1.0)),     # behind the wall
    ((0.0, -6.0, 1.2), (0.0, 0.0, 1.0)),     # same view again: exact history
    ((2.5, -5.0, 1.6), (0.0, 0.0, 1.0)),     # strafed: stale history, disocclusions
    ((6.0, 1.0, 4.0), (0.0, 0.0, 0.5)),      # far away from the last view: history nearly useless
    ((6.0, 1.0, 4.0), (0.0, 0.0, 0.5)),
]


def test_sequence_is_history_independent(oracle_lib, gpu_engine):
    W, H, SD = 384, 216, 256
    o = oracle_lib.Oracle(W, H, SD)
    g = gpu_engine.Renderer(W, H, SD)
    n = gpu_engine.Renderer(W, H, SD, flags=abi.FLAG_NO_HIZ)
    for r in (o, g, n):
        _crowd(r)
    d, p, s = _lights()
    culled = []
    for i, (pos, look) in enumerate(CAMS):
        cam = abi.make_camera(pos, look, fov=50.0)
        for r in (o, g, n):
            r.update_uniforms(cam, d, p, s, 0.1 * i, 0.02 * i, 1.0 + i)     # the stage and the lights move too
        o.render(0)
        g.render(); g.finish()
        n.render(); n.finish()
        _identical(o, g, "frame %d" % i)
        assert (g.color() == n.color()).all()
        sg, sn = g.stats(), n.stats()
        assert sn["hiz_culled"] == 0 and sn["round1_survivors"] == 0
        assert sg["covered_pixels"] == sn["covered_pixels"] == o.covered_pixels()
        # both rounds together never draw more than the single pass, and what Hi-Z rejected is exactly the difference
        assert sg["survivors"][1] + sg["hiz_culled"] == sn["survivors"][1]
        culled.append(sg["hiz_culled"])
        if i == 0:
            assert sg["round1_survivors"] == 0 and sg["hiz_culled"] == 0      # no history yet: one round
        else:
            assert sg["round1_survivors"] > 0
    assert culled[1] > 50, culled          # behind the wall much of the crowd is hidden


def test_scene_change_resets_history(oracle_lib, gpu_engine):
    """Work item numbering changes when the scene does: the history of the old scene must not be consulted."""
    W, H, SD = 256, 144, 128
    g = gpu_engine.Renderer(W, H, SD)
    d, p, s = _lights()
    cam = abi.make_camera((0.0, -6.0, 1.2), (0.0, 0.0, 1.0), fov=50.0)
    for seed, count in ((5, 300), (9, 500), (5, 120)):
        g.scene_clear()
        _crowd(g, count, seed)
        o = oracle_lib.Oracle(W, H, SD)
        _crowd(o, count, seed)
        for r in (o, g):
            r.update_uniforms(cam, d, p, s, 0.0, 0.0, 1.0)
        o.render(0)
        for k in range(2):
            g.render(); g.finish()
            _identical(o, g, "scene %d/%d frame %d" % (seed, count, k))
            st = g.stats()
            assert (st["round1_survivors"] > 0) == (k == 1)


def test_full_size_config3_second_frame(oracle_lib, gpu_engine, tmp_path):
    """BASELINE config 3 at 1920x1080, second frame (history exact): identical to the oracle, most hidden work rejected."""
    cfg = scenes.config3()
    from zeldaengine_amd import engine as eng
    o = oracle_lib.Oracle(cfg["width"], cfg["height"], 1024)
    oracle_lib.load_scene(o, cfg)
    o.render(0)
    g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024)
    eng.load_scene(g, cfg)
    for k in range(2):
        g.render(); g.finish()
    _identical(o, g, "config 3, frame 2")
    st = g.stats()
    assert st["hiz_culled"] > 10000, st
    assert st["round1_survivors"] < 0.5 * (st["survivors"][1] + st["hiz_culled"]), st


def test_tile_partitioned_ranks_keep_their_own_history(oracle_lib, gpu_engine):
    """Two rank contexts (tiles t % 2) on one GPU over a moving sequence: each rank's occlusion history covers its own tiles
    only, and the two packed halves still assemble to the oracle's frame on every frame."""
    from zeldaengine_amd import dist as zdist
    W, H, SD = 352, 208, 128
    o = oracle_lib.Oracle(W, H, SD)
    ranks = [gpu_engine.Renderer(W, H, SD, tile_rank=r, tile_world=2) for r in range(2)]
    for r in [o] + ranks:
        _crowd(r, 400, 3)
    d, p, s = _lights()
    for i, (pos, look) in enumerate(CAMS[:4]):
        cam = abi.make_camera(pos, look, fov=50.0)
        for r in [o] + ranks:
            r.update_uniforms(cam, d, p, s, 0.05 * i, 0.0, 1.0)
        o.render(0)
        want = o.color()
        for k, g in enumerate(ranks):
            g.render(); g.finish()
            assert np.array_equal(g.read_tiles(), zdist.pack_tiles(want, k, 2)), "frame %d rank %d" % (i, k)
        if i:
            assert all(g.stats()["round1_survivors"] > 0 for g in ranks)


def test_rank_contexts_with_small_buckets_draw_their_tiles_from_the_overflow_sections(oracle_lib, gpu_engine):
    """Four rank contexts (screen super-tiles) with every record bucket planned at 25 % of its size (zr_set_bucket_share) over a
    sequence of camera jumps: most of every owned tile's records take the overflow route - sections by tile number, of which a rank
    owns only some - and the four packed quarters still assemble to the oracle's frame on every frame."""
    from zeldaengine_amd import dist as zdist
    W, H, SD = 352, 208, 128
    o = oracle_lib.Oracle(W, H, SD)
    ranks = [gpu_engine.Renderer(W, H, SD, tile_rank=r, tile_world=4) for r in range(4)]
    for g in ranks:
        g.set_bucket_share(25)
    for r in [o] + ranks:
        _crowd(r, 900, 7)
    d, p, s = _lights()
    for i, (pos, look) in enumerate(CAMS[:4] + CAMS[:2]):
        cam = abi.make_camera(pos, look, fov=50.0)
        for r in [o] + ranks:
            r.update_uniforms(cam, d, p, s, 0.05 * i, 0.0, 1.0)
        o.render(0)
        want = o.color()
        for k, g in enumerate(ranks):
            g.render(); g.finish()
            assert g.stats()["overflow"] == 0
            assert np.array_equal(g.read_tiles(), zdist.pack_tiles(want, k, 4)), "frame %d rank %d" % (i, k)
    for g in ranks:
        g.close()


def _three_frames_with_a_light_step(g, cfg):
    """Two still frames, then the shadow-casting light turns one degree about the scene: the third frame runs on a visibility history
    and on shadow-occlusion flags that belong to the old light."""
    import math
    d = cfg["dir"].copy()
    p0 = np.array(cfg["dir"]["Position"][0][:3], dtype=np.float64)
    rad, a0 = math.hypot(p0[0], p0[1]), math.atan2(p0[1], p0[0])
    for i, deg in enumerate((0.0, 0.0, 1.0)):
        a = a0 + math.radians(deg)
        d["Position"][0][:3] = (rad * math.cos(a), rad * math.sin(a), p0[2]); d["Direction"][0][:3] = d["Position"][0][:3]
        g.update_uniforms(cfg["camera"], d, cfg["point"], cfg["spot"], 0.0, 0.0, 0.0)
        g.render()
    g.finish()


def test_config4_scale_culling_paths_agree(gpu_engine):
    """BASELINE config 4's size (1 M instances = 11 M meshlet-instances, 3840x2160; too big for the scalar oracle): the frame with
    every conservative cull on (instance pre-pass + work list, frustum, cone, two-pass Hi-Z, the shadow pass's occlusion culling; third
    frame, after the light has moved = histories in use and stale) must equal the frame with all of them off, in every target."""
    from zeldaengine_amd import engine as eng
    cfg = scenes.config4(1000000, 16)
    frames = []
    for flags in (0, abi.FLAG_NO_FRUSTUM_CULL | abi.FLAG_NO_CONE_CULL | abi.FLAG_NO_HIZ | abi.FLAG_NO_SHADOW_OCCLUSION):
        g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024, flags=flags)
        eng.load_scene(g, cfg)
        _three_frames_with_a_light_step(g, cfg)
        frames.append((g.color(), [g.gbuffer(t) for t in range(6)], g.shadowmap(), g.stats()))
        g.close()
    a, b = frames
    assert np.array_equal(a[0], b[0])
    for t in range(6):
        assert np.array_equal(a[1][t].view(np.uint8), b[1][t].view(np.uint8)), "GBuffer target %d" % t
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert a[3]["covered_pixels"] == b[3]["covered_pixels"] and a[3]["overflow"] == 0 and b[3]["overflow"] == 0
    # the shadow pass's occlusion culling switches itself on at this density (10 meshlet-instances per texel): a third of the casters is
    # left out, the light's step brings some back late
    assert a[3]["shadow_occluded"] > 500000 and a[3]["shadow_late"] > 0 and b[3]["shadow_occluded"] == 0 and b[3]["shadow_late"] == 0, (a[3], b[3])
    # (the bound was // 4 with round 2's 14-meshlet spheres: the 11 fuller meshlets of the slab x sector clusteriser have wider normal
    # cones - frustum + cone culls remove 33 % of the meshlet-instances where they removed 41 % - so fewer fall before the rasteriser)
    assert a[3]["hiz_culled"] > 100000 and a[3]["survivors"][1] < b[3]["survivors"][1] // 3


def test_camera_cut_at_config4_scale_matches_a_fresh_context(gpu_engine):
    """A camera CUT at 1 M instances / 3840x2160: the frame after it runs on a stale visibility history - round 2 then draws most of the
    scene, tens of millions of records beyond the buckets planned from the frame before, into the sections of the overflow region
    (the round's first bucketed build reported ZR_ERR_OVERFLOW here, and sifted quadratically before that).  Every target of that frame,
    and of the one after it, must equal what a context that has only ever seen the new camera draws."""
    import math
    from zeldaengine_amd import engine as eng
    cfg = scenes.config4(1000000, 16)

    def cam(deg, h):
        a = math.radians(deg)
        return abi.make_camera((math.sqrt(50.0) * math.cos(a), math.sqrt(50.0) * math.sin(a), h), (0.0, 0.0, 0.0))

    def frame(g, c):
        g.update_uniforms(c, cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.0, 0.0)
        g.render()

    def grab(g):
        g.finish()
        st = g.stats()
        assert st["overflow"] == 0, st
        return g.color(), [g.gbuffer(t) for t in range(6)], g.shadowmap(), st

    new_cam = cam(170.0, 2.0)
    a = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024)
    eng.load_scene(a, cfg)
    for _ in range(3):
        frame(a, cam(45.0, 5.0))                     # a settled history and plan for the OLD camera
    frame(a, new_cam)
    cut = grab(a)                                    # the frame after the cut
    frame(a, new_cam)
    after = grab(a)                                  # ... and the next one (planned from the cut frame's counts)
    a.close()
    b = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024)
    eng.load_scene(b, cfg)
    frame(b, new_cam)
    fresh = grab(b)
    b.close()
    for name, got in (("cut", cut), ("after the cut", after)):
        assert np.array_equal(got[0], fresh[0]), name
        for t in range(6):
            assert np.array_equal(got[1][t].view(np.uint8), fresh[1][t].view(np.uint8)), (name, "GBuffer target %d" % t)
        assert np.array_equal(got[2].view(np.uint32), fresh[2].view(np.uint32)), name
        assert got[3]["covered_pixels"] == fresh[3]["covered_pixels"], name
    # the cut frame really went the overflow way: far more records than the settled frames before it
    assert cut[3]["bin_entries"][1] > 3 * after[3]["bin_entries"][1], (cut[3], after[3])


def test_soak_two_frames_in_flight(oracle_lib, gpu_engine):
    """600 frames enqueued back to back (no host synchronisation in between) with the camera, the stage and the lights moving every
    frame: the double-buffered uniforms / GBuffer / shadow map of the two-frames-in-flight schedule and the pinned upload ring must
    never mix frames.  Every 75th frame is read back and compared with the oracle in all targets."""
    W, H, SD = 256, 144, 128
    o = oracle_lib.Oracle(W, H, SD)
    g = gpu_engine.Renderer(W, H, SD)
    for r in (o, g):
        _crowd(r, 250, 8)
    d, p, s = _lights()
    import math
    for i in range(600):
        a = 0.01 * i
        cam = abi.make_camera((7.0 * math.cos(a), 7.0 * math.sin(a), 1.0 + 0.6 * math.sin(3.0 * a)), (0.0, 0.0, 0.8), fov=50.0)
        g.update_uniforms(cam, d, p, s, 0.002 * i, 0.001 * i, 0.1 * i)
        g.render()
        if i % 75 == 74 or i == 599:
            o.update_uniforms(cam, d, p, s, 0.002 * i, 0.001 * i, 0.1 * i)
            o.render(0)
            g.finish()
            _identical(o, g, "frame %d" % i)


def test_config5_full_size_culling_paths_agree(gpu_engine):
    """BASELINE config 5 at its full size (1 M instances = 14 M meshlet-instances, 3840x2160, two-pass Hi-Z, 256 point lights with
    per-tile light lists; too big for the scalar oracle): two frames so that the visibility history and the pyramid are in use;
    every conservative cull on must equal every cull off, in every target."""
    from zeldaengine_amd import engine as eng
    cfg = scenes.config4(1000000, 256)
    assert len(cfg["point"]) == 256 and (cfg["width"], cfg["height"]) == (3840, 2160)
    frames = []
    for flags in (0, abi.FLAG_NO_FRUSTUM_CULL | abi.FLAG_NO_CONE_CULL | abi.FLAG_NO_HIZ | abi.FLAG_NO_SHADOW_OCCLUSION):
        g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024, flags=flags)
        eng.load_scene(g, cfg)
        _three_frames_with_a_light_step(g, cfg)
        frames.append((g.color(), [g.gbuffer(t) for t in range(6)], g.shadowmap(), g.stats()))
        g.close()
    a, b = frames
    assert np.array_equal(a[0], b[0])
    for t in range(6):
        assert np.array_equal(a[1][t].view(np.uint8), b[1][t].view(np.uint8)), "GBuffer target %d" % t
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert a[3]["covered_pixels"] == b[3]["covered_pixels"] and a[3]["overflow"] == 0 and b[3]["overflow"] == 0
    assert a[3]["shadow_occluded"] > 500000 and b[3]["shadow_occluded"] == 0, (a[3], b[3])
    # (// 1.9, not // 4: same clusteriser effect as above, and with 256 lights nothing else changes what the culls see; survivors[1]
    # counts the meshlet-instances that reach k_geom in either round - triangles the per-triangle pyramid test drops are not taken off)
    assert a[3]["round1_survivors"] > 0 and a[3]["hiz_culled"] > 100000 and a[3]["survivors"][1] < b[3]["survivors"][1] // 1.9
    assert len(np.unique(a[0].reshape(-1, 4), axis=0)) > 1000       # a lit frame, not a constant


def test_config5_reduced_against_the_oracle_with_a_moving_camera(oracle_lib, gpu_engine):
    """Config 5's ingredients together at a size the oracle finishes in seconds: 3 000 instanced spheres, 256 point lights (tile
    light lists), 640x360, Hi-Z history in use, the camera orbiting and the lights riding their spiral between frames."""
    import math
    from zeldaengine_amd import engine as eng
    cfg = scenes.config3(3000, 640, 360, 20.0, 256)
    o = oracle_lib.Oracle(cfg["width"], cfg["height"], 512)
    g = gpu_engine.Renderer(cfg["width"], cfg["height"], 512)
    oracle_lib.load_scene(o, cfg)
    eng.load_scene(g, cfg)
    culled = 0
    for i in range(3):
        a = 0.6 + 0.12 * i
        cam = abi.make_camera((8.0 * math.cos(a), 8.0 * math.sin(a), 4.0 + 0.3 * i), (0.0, 0.0, 0.3), fov=45.0)
        for r in (o, g):
            r.update_uniforms(cam, cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.01 * i, 0.5 * i)
        o.render(0)
        g.render(); g.finish()
        _identical(o, g, "reduced config 5, frame %d" % i)
        st = g.stats()
        assert st["covered_pixels"] == o.covered_pixels() and st["overflow"] == 0
        if i:
            assert st["round1_survivors"] > 0
            culled += st["hiz_culled"]
    assert culled > 0


def test_camera_pass_against_the_oracle_with_and_without_hiz_rounds(oracle_lib, gpu_engine):
    """The triangle-binned camera pass (k_cull_box, k_geom, k_tile) over the history sequence, with and without the Hi-Z rounds:
    both must give the oracle's frame, and the slow-triangle list must be exercised (the ground plane crosses the near plane; the
    wall's triangles are longer than 64 pixels).  (The meshlet-binned A/B rasteriser for the camera pass exists in -DZR_DIAG builds only:
    the product library refuses ZR_FLAG_MESHLET_BINS.)"""
    W, H, SD = 384, 216, 256
    o = oracle_lib.Oracle(W, H, SD)
    gs = [gpu_engine.Renderer(W, H, SD, flags=f) for f in (0, abi.FLAG_NO_HIZ)]
    for r in [o] + gs:
        _crowd(r, 300, 11)
    d, p, s = _lights()
    for i, (pos, look) in enumerate(CAMS[:4]):
        cam = abi.make_camera(pos, look, fov=50.0)
        for r in [o] + gs:
            r.update_uniforms(cam, d, p, s, 0.1 * i, 0.02 * i, 1.0 + i)
        o.render(0)
        for k, g in enumerate(gs):
            g.render(); g.finish()
            _identical(o, g, "frame %d, path %d" % (i, k))
            g.render(); g.finish()          # the same view once more: exact history
            _identical(o, g, "frame %d again, path %d" % (i, k))
    st = [g.stats() for g in gs]
    assert all(x["overflow"] == 0 for x in st)
    assert st[0]["covered_pixels"] == st[1]["covered_pixels"]
    for g in gs:
        g.close()
    with pytest.raises(gpu_engine.ZeldaRenderError):
        gpu_engine.Renderer(W, H, SD, flags=abi.FLAG_MESHLET_BINS)


def test_record_pool_chunks_beyond_the_first(oracle_lib, gpu_engine):
    """With NO_HIZ every frame is one round over all frustum / cone survivors: at 60 000 instances a wave of k_geom handles ~60
    meshlet-instances and the busiest tiles hold tens of thousands of records - buckets split over many work units of k_tile, planned from a
    count-only run of k_geom on the first frame.  Checked against the scalar oracle (57.6 M triangles: ~20 s of CPU)."""
    from zeldaengine_amd import engine as eng
    cfg = scenes.config3(60000, 1920, 1080)
    g = gpu_engine.Renderer(cfg["width"], cfg["height"], 1024, flags=abi.FLAG_NO_HIZ)
    eng.load_scene(g, cfg)
    g.render(); g.render(); g.finish()
    st = g.stats()
    assert st["overflow"] == 0
    assert st["bin_entries"][1] > 8192 * 256 * 2, st            # records: well beyond what the waves' own chunks (256 each) hold
    o = oracle_lib.Oracle(cfg["width"], cfg["height"], 1024)
    oracle_lib.load_scene(o, cfg)
    o.set_threads(16)
    o.render()
    _identical(o, g, "60 000 instances, one round")
    assert st["covered_pixels"] == o.covered_pixels()
    g.close()


def test_big_scene_shadow_path_with_a_moving_light(oracle_lib, gpu_engine):
    """Scenes with the instance-level pre-pass (>= 65 536 instances: whole-instance frustum / "no texel centre" rejects, compacted work
    list with one reservation per 4 096 instances) against the oracle, with the light moving every frame: 70 000 low-poly spheres lie
    many deep under a 256^2 shadow map."""
    W, H, SD = 320, 200, 256
    mesh = scenes.uv_sphere(8, 4, 0.5)
    inst = scenes.generate_instances(70000, 0.5, 9.0, 0.15, 0.5, seed=9)
    o = oracle_lib.Oracle(W, H, SD)
    g = gpu_engine.Renderer(W, H, SD)
    for r in (o, g):
        r.set_cubemap(scenes.synthetic_cubemap(8))
        r.object_add(r.mesh_create(*scenes.grid_plane(24.0, 4, 0.0)))
        r.object_add(r.mesh_create(*mesh), None, inst)
    d, p, s = _lights()
    cam = abi.make_camera((7.0, 6.0, 5.0), (0.0, 0.0, 0.3))
    for i, lpos in enumerate([(20.0, 0.0, 20.0), (20.0, 0.0, 20.0), (16.0, 9.0, 18.0), (-3.0, 19.0, 9.0)]):
        d[0]["Position"][:3] = lpos; d[0]["Direction"][:3] = lpos
        for r in (o, g):
            r.update_uniforms(cam, d, p, s, 0.0, 0.01 * i, 1.0)
        o.render(0)
        g.render(); g.finish()
        assert np.array_equal(o.shadowmap().view(np.uint32), g.shadowmap().view(np.uint32)), "shadow map, frame %d" % i
        _identical(o, g, "frame %d" % i)
        st = g.stats()
        assert st["overflow"] == 0 and st["survivors"][0] > 0 and st["work_items"][0] == 70001
    g.close()


@pytest.mark.gpu
def test_work_lists_stand_only_while_camera_light_and_scene_do(oracle_lib, gpu_engine):
    """The passes' instance-level work lists (k_cull_instances: scenes of >= 65 536 instances, and every tile-partitioned context) are
    rebuilt only when the pass's matrices or the scene change.  70 000 instances through: two still frames (the second reuses both
    lists), a moved camera (camera list rebuilt, shadow list kept), a still frame, a moved light (shadow list rebuilt), a changed scene
    (both rebuilt: the work items are renumbered), a still frame - every frame against the oracle, and against a context that rebuilds
    its lists every frame (ZR_FLAG_NO_LIST_REUSE)."""
    W, H, SD = 320, 200, 256
    mesh = scenes.uv_sphere(8, 4, 0.5)
    inst = scenes.generate_instances(70000, 0.5, 9.0, 0.15, 0.5, seed=11)
    o = oracle_lib.Oracle(W, H, SD)
    g = gpu_engine.Renderer(W, H, SD)
    h = gpu_engine.Renderer(W, H, SD, flags=abi.FLAG_NO_LIST_REUSE)
    for r in (o, g, h):
        r.set_cubemap(scenes.synthetic_cubemap(8))
        r.object_add(r.mesh_create(*scenes.grid_plane(24.0, 4, 0.0)))
        r.object_add(r.mesh_create(*mesh), None, inst)
    d, p, s = _lights()
    cams = [(7.0, 6.0, 5.0), (7.0, 6.0, 5.0), (-6.0, 7.5, 4.0), (-6.0, 7.5, 4.0), (-6.0, 7.5, 4.0), (-6.0, 7.5, 4.0), (-6.0, 7.5, 4.0)]
    lights = [(20.0, 0.0, 20.0)] * 4 + [(14.0, 11.0, 17.0)] * 3
    for i in range(7):
        if i == 5:          # the scene changes under a still camera and light
            extra = scenes.generate_instances(300, 1.0, 6.0, 0.3, 0.6, seed=12)
            for r in (o, g, h):
                r.object_add(r.mesh_create(*scenes.uv_sphere(12, 6, 0.5)), None, extra)
        d[0]["Position"][:3] = lights[i]; d[0]["Direction"][:3] = lights[i]
        cam = abi.make_camera(cams[i], (0.0, 0.0, 0.3))
        for r in (o, g, h):
            r.update_uniforms(cam, d, p, s, 0.0, 0.01 * i, 1.0)
        o.render(0)
        for r in (g, h):
            r.render(); r.finish()
        assert np.array_equal(o.shadowmap().view(np.uint32), g.shadowmap().view(np.uint32)), "shadow map, frame %d" % i
        _identical(o, g, "frame %d (lists reused)" % i)
        _identical(o, h, "frame %d (lists rebuilt)" % i)
        sg, sh = g.stats(), h.stats()
        assert sg["overflow"] == 0 and sg["survivors"] == sh["survivors"] and sg["covered_pixels"] == sh["covered_pixels"], (i, sg, sh)
    g.close(); h.close()
