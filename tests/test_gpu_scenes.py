"""GPU parity over the scene-submission surface: draw kinds, clipping, raw frames, caller meshlets, debug views, sizes.

All comparisons are bit-exact against the CPU oracle (see test_gpu_parity.py for the rationale).
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


def _const_image(rgba, n=4):
    return np.tile(np.array(rgba, dtype=np.uint8), (n, n, 1))


def _mixed_scene(r, instanced_boxes=40):
    """ground plane + big box (non-instanced draws), spheres + boxes (instanced draws), two materials."""
    r.set_cubemap(scenes.synthetic_cubemap(32))
    plane = r.mesh_create(*scenes.grid_plane(16.0, 6, 0.0))
    box = r.mesh_create(*scenes.box((0.8, 0.5, 0.4), (0.0, 0.0, 0.4)))
    sph = r.mesh_create(*scenes.uv_sphere())
    mat, keep = abi.make_material([_const_image((200, 40, 30, 255)), _const_image((255, 255, 255, 255)), _const_image((90, 90, 90, 255)),
                                   None, _const_image((180, 180, 180, 255)), _const_image((0, 20, 40, 255)), None])
    r._keepalive = keep
    r.object_add(sph, None, scenes.generate_instances(150, 1.5, 7.0, 0.2, 0.6, seed=7))
    r.object_add(plane)
    r.object_add(box, mat)
    r.object_add(box, mat, scenes.generate_instances(instanced_boxes, 2.0, 6.0, 0.3, 0.9, seed=11))


def _both(oracle_lib, gpu_engine, W, H, SD, build, frame, flags=0, debug_view=0):
    o = oracle_lib.Oracle(W, H, SD)
    g = gpu_engine.Renderer(W, H, SD, flags=flags)
    for r in (o, g):
        build(r)
        frame(r)
    o.render(debug_view)
    g.render(debug_view)
    g.finish()
    return o, g


def _std_frame(cam=None, roll_stage=0.0, roll_light=0.0, n_point=16):
    w = scenes.sample_world()
    d, _, s = scenes.lights_from_world(w)
    w["PointLights"] = scenes.sample_point_lights(n_point)
    _, p, _ = scenes.lights_from_world(w)

    def f(r):
        r.update_uniforms(cam or abi.make_camera(), d, p, s, roll_stage, roll_light, 1.0)
    return f


def test_mixed_draws_materials_and_animation(oracle_lib, gpu_engine):
    o, g = _both(oracle_lib, gpu_engine, 416, 240, 512, _mixed_scene, _std_frame(roll_stage=0.4, roll_light=0.03))
    assert o.covered_pixels() > 0.5 * 416 * 240
    _identical(o, g, "mixed scene")


def test_odd_resolution_and_small_shadow_map(oracle_lib, gpu_engine):
    o, g = _both(oracle_lib, gpu_engine, 333, 211, 96, _mixed_scene, _std_frame(abi.make_camera((4.0, -6.0, 3.0), (0.0, 0.0, 0.5), fov=60.0)))
    _identical(o, g, "333x211")


def test_near_plane_and_guard_band_clipping(oracle_lib, gpu_engine):
    """Camera just above a huge ground quad, inside a crowd of spheres: triangles cross the near plane and the guard band."""
    def build(r):
        r.set_cubemap(None)
        r.object_add(r.mesh_create(*scenes.grid_plane(400.0, 2, 0.0)))
        r.object_add(r.mesh_create(*scenes.box((30.0, 0.2, 3.0), (0.0, 2.0, 3.0))))
        r.object_add(r.mesh_create(*scenes.uv_sphere()), None, scenes.generate_instances(60, 0.2, 2.5, 0.8, 1.6, seed=3))
    cam = abi.make_camera((0.3, -0.6, 0.45), (0.0, 4.0, 0.6), fov=75.0, znear=0.05, zfar=200.0)
    o, g = _both(oracle_lib, gpu_engine, 320, 200, 256, build, _std_frame(cam))
    assert o.covered_pixels() > 0.6 * 320 * 200
    _identical(o, g, "clipping")


def test_raw_frame_without_the_y_flip(oracle_lib, gpu_engine):
    """zr_set_frame with the engine's matrices but Proj[1][1] un-flipped: handedness reverses, so cone culling must switch
    itself off (what is front-facing changes) and the frame must still match."""
    def frame(r):
        _std_frame()(r)
        cam, sh, view = r.get_frame()
        cam["Proj"][5] *= -1.0
        r.set_frame(cam, sh, view)
    o, g = _both(oracle_lib, gpu_engine, 256, 160, 128, _mixed_scene, frame)
    _identical(o, g, "un-flipped projection")


def test_caller_supplied_meshlets_follow_the_indirect_draw_order(oracle_lib, gpu_engine):
    """zr_mesh_set_meshlets flattens like CreateMeshVertexBuffers<XkMeshIndirect> (ZE:4733-4756): primitive order becomes
    meshlet order.  The oracle gets the equivalent flattened index buffer."""
    v, idx = scenes.uv_sphere()
    ml, mv, mt, order = gpu_engine.build_meshlets(v, idx, 48, 60, 0.5)
    flat = idx.reshape(-1, 3)[order].reshape(-1)
    inst = scenes.generate_instances(80, 1.0, 5.0, 0.3, 0.8, seed=5)
    o = oracle_lib.Oracle(320, 180, 128)
    o.object_add(o.mesh_create(v, flat), None, inst)
    g = gpu_engine.Renderer(320, 180, 128)
    m = g.mesh_create(v, idx)
    g.mesh_set_meshlets(m, ml, mv, mt)
    g.object_add(m, None, inst)
    for r in (o, g):
        r.set_cubemap(scenes.synthetic_cubemap(16))
        _std_frame()(r)
    o.render(); g.render(); g.finish()
    _identical(o, g, "caller meshlets")


def test_meshlet_file_from_the_tool_renders_like_the_engine_would(oracle_lib, gpu_engine, tmp_path):
    """Row a4 end to end: OBJ -> ZeldaMeshlet tool (assets.meshlet_tool, ZM:235-294) -> `.meshlet` file on disk -> LoadMeshletAsset
    (assets.read_meshlet, ZE:7046-7169) -> zr_mesh_set_meshlets -> GPU frame.  The oracle draws the flattened index buffer of
    CreateMeshVertexBuffers<XkMeshIndirect> (ZE:4733-4756) built independently here from the file's sections."""
    from zeldaengine_amd import assets
    v, idx = scenes.uv_sphere(24, 12, 0.8)
    obj = str(tmp_path / "ball.obj")
    assets.write_obj(obj, v, idx)
    path = str(tmp_path / "ball.meshlet")
    n = assets.meshlet_tool(obj, path)
    f = assets.read_meshlet(path)
    assert n == len(f["meshlets"]) > 1
    flat = []
    for ml in f["meshlets"]:
        for t in range(int(ml["TriangleCount"])):
            for k in range(3):
                flat.append(f["mverts"][int(ml["VertexOffset"]) + int(f["mtris"][int(ml["TriangleOffset"]) + 3 * t + k])])
    flat = np.asarray(flat, dtype=np.uint32)
    assert len(flat) == len(f["indices"])
    inst = scenes.generate_instances(40, 1.0, 4.0, 0.4, 0.9, seed=11)
    o = oracle_lib.Oracle(320, 180, 128)
    o.object_add(o.mesh_create(f["vertices"], flat), None, inst)
    o.object_add(o.mesh_create(f["vertices"], flat))
    g = gpu_engine.Renderer(320, 180, 128)
    m = g.mesh_create(f["vertices"], f["indices"])
    g.mesh_set_meshlets(m, f["meshlets"], f["mverts"], f["mtris"])
    g.object_add(m, None, inst)
    g.object_add(m)
    for r in (o, g):
        r.set_cubemap(scenes.synthetic_cubemap(16))
        _std_frame()(r)
    o.render(); g.render(); g.render(); g.finish()
    _identical(o, g, ".meshlet file")


@pytest.mark.parametrize("view", [1, 2, 3, 4, 5, 7, 8])
def test_debug_views(oracle_lib, gpu_engine, view):
    o, g = _both(oracle_lib, gpu_engine, 192, 128, 128, _mixed_scene, _std_frame(), debug_view=view)
    assert np.array_equal(o.color(), g.color())


def test_debug_view_9_gbuffer_mosaic(oracle_lib, gpu_engine):
    """SPEC_CONSTANTS 9 (BaseLighting.frag:42-145): 3x3 mosaic of the GBuffer re-sampled at 3x the texture coordinate."""
    W, H = 192, 129          # 192 = 3 * 64: the cell borders fall on pixel edges; 129 = 3 * 43
    o, g = _both(oracle_lib, gpu_engine, W, H, 128, _mixed_scene, _std_frame(), debug_view=9)
    mosaic = g.color()
    assert np.array_equal(o.color(), mosaic)
    # the centre cell keeps the lit frame, the bottom-left one is black, the top-left one is the gamma-encoded base colour
    o0, g0 = _both(oracle_lib, gpu_engine, W, H, 128, _mixed_scene, _std_frame(), debug_view=0)
    lit = g0.color()
    assert np.array_equal(mosaic[43:86, 64:128], lit[43:86, 64:128])
    assert (mosaic[86:, :64, :3] == 0).all()
    o1, g1 = _both(oracle_lib, gpu_engine, W, H, 128, _mixed_scene, _std_frame(), debug_view=1)
    assert np.array_equal(mosaic[:43, :64], g1.color()[1::3, 1::3][:43, :64])      # texel (3x + 1, 3y + 1)


def test_debug_view_9_with_editor_bars(oracle_lib, gpu_engine):
    """ViewportInfo.zw = the space ImGui's right / bottom bars take (ZE:4636): cells shrink, samples fall between texels
    (LINEAR filtering) and the cells get their white frames."""
    def frame(r):
        _std_frame()(r)
        cam, sh, view = r.get_frame()
        view["ViewportInfo"][2] = 37.0
        view["ViewportInfo"][3] = 21.0
        r.set_frame(cam, sh, view)
    o, g = _both(oracle_lib, gpu_engine, 200, 120, 64, _mixed_scene, frame, debug_view=9)
    assert np.array_equal(o.color(), g.color())
    assert (g.color()[..., :3] == 255).all(axis=2).sum() > 100            # the white frames exist


def test_skydome_and_background_passes(oracle_lib, gpu_engine):
    """N3: skydome mesh (unlit, depth LESS vs the deferred depth) + background quad at z = 1 (LESS_OR_EQUAL), view 0 only."""
    bg = scenes.synthetic_sky_image(90, 60)[:, ::-1].copy()

    def build_all(r):
        _mixed_scene(r, 10)
        r.set_skydome(*scenes.sky_dome(), scenes.synthetic_sky_image())
        r.set_background(bg)
    cam = abi.make_camera((6.0, 5.0, 2.0), (0.0, 0.0, 1.5), zfar=20.0)      # zFar 20: part of the dome is beyond the far plane
    o, g = _both(oracle_lib, gpu_engine, 352, 208, 128, build_all, _std_frame(cam))
    _identical(o, g, "sky + background")
    lit = _both(oracle_lib, gpu_engine, 352, 208, 128, lambda r: _mixed_scene(r, 10), _std_frame(cam))[1].color()
    both = g.color()
    assert (both != lit).any(axis=2).mean() > 0.2                          # the passes really drew
    # the dome is not a shadow caster and leaves the GBuffer untouched
    assert np.array_equal(g.shadowmap().view(np.uint32), _both(oracle_lib, gpu_engine, 352, 208, 128, lambda r: _mixed_scene(r, 10), _std_frame(cam))[1].shadowmap().view(np.uint32))
    # flags (XkWorld::EnableSkydome / EnableBackground) and debug views switch them off
    for r in (o, g):
        r.set_sky_flags(False, True)
    o.render(); g.render(); g.finish()
    _identical(o, g, "background only")
    for r in (o, g):
        r.set_sky_flags(True, False)
    o.render(3); g.render(3); g.finish()
    assert np.array_equal(o.color(), g.color())


def test_debug_view_6_quad_vertex_colour(oracle_lib, gpu_engine):
    o, g = _both(oracle_lib, gpu_engine, 200, 120, 64, lambda r: _mixed_scene(r, 5), _std_frame(), debug_view=6)
    assert np.array_equal(o.color(), g.color())
    c = g.color()
    assert c[0, 0, 0] > 250 and c[-1, 0, 2] > 250 and c[-1, -1, 1] > 250     # red top-left, blue bottom-left, green bottom-right


def test_empty_scene_and_scene_reuse(oracle_lib, gpu_engine):
    o = oracle_lib.Oracle(96, 64, 64)
    g = gpu_engine.Renderer(96, 64, 64)
    for r in (o, g):
        r.set_cubemap(scenes.synthetic_cubemap(8))
        _std_frame()(r)
    o.render(); g.render(); g.finish()
    _identical(o, g, "empty scene")
    assert g.stats()["covered_pixels"] == 0
    for r in (o, g):
        _mixed_scene(r, 5)
    o.render(); g.render(); g.finish()
    _identical(o, g, "after adding objects")
    for r in (o, g):
        r.scene_clear()
        r.object_add(r.mesh_create(*scenes.box()))
    o.render(); g.render(); g.finish()
    _identical(o, g, "after scene_clear")


def test_triangles_with_one_long_edge(oracle_lib, gpu_engine):
    """A triangle whose two edges at its FIRST vertex are under 64 pixels while the third is twice that is not a "small" triangle: the
    small-triangle forms keep tile-relative 16-bit coordinates and 32-bit edge products.  (Found by tests/test_gpu_fuzz.py: only
    the first vertex's edges were looked at, and such a triangle lost its part in the leftmost tile it touches.)  Pixel-space
    triangles through identity matrices, in both passes - shadow map of the target's size - and every rotation of the vertex order."""
    W, H = 192, 96
    base = [((70.3, 20.2, 0.40), (9.1, 30.7, 0.50), (131.6, 33.4, 0.45)),        # 62 / 63 px from vertex 0, 123 px between the others
            ((100.5, 60.1, 0.30), (40.2, 75.6, 0.35), (161.9, 70.3, 0.30)),
            ((20.0, 50.0, 0.60), (20.7, 90.9, 0.62), (80.2, 8.4, 0.58))]         # the long edge mostly vertical
    tris = []
    for t in base:
        for rot in range(3):
            tris.append(tuple(t[(k + rot) % 3] for k in range(3)))
            tris.append(tuple(t[(rot - k) % 3] for k in range(3)))              # the other winding: back face in the camera pass, drawn in the shadow pass
    verts = np.zeros(3 * len(tris), dtype=abi.XkVertex)
    for k, tri in enumerate(tris):
        for j, (x, y, z) in enumerate(tri):
            dz = 0.001 * (k // 6)                                                 # the copies of one triangle do not tie in depth
            verts[3 * k + j]["Position"] = (x / (W / 2.0) - 1.0, y / (H / 2.0) - 1.0, z + dz)
            verts[3 * k + j]["Normal"] = (0.0, 0.0, 1.0); verts[3 * k + j]["Color"] = (1.0, 1.0, 1.0)

    def build(r):
        r.set_cubemap(scenes.synthetic_cubemap(8))
        r.object_add(r.mesh_create(verts, np.arange(len(verts), dtype=np.uint32)))

    def frame(r):
        _std_frame()(r)
        cam, sh, view = r.get_frame()
        ident = np.eye(4, dtype=np.float32).reshape(16)
        for m in (cam, sh):
            m["Model"] = ident; m["View"] = ident; m["Proj"] = ident
        r.set_frame(cam, sh, view)
    o = oracle_lib.Oracle(W, H, 192)
    g = gpu_engine.Renderer(W, H, 192)
    for r in (o, g):
        build(r); frame(r)
    o.render(); g.render(); g.finish()
    assert (o.gbuffer(0) < 1.0).sum() > 1500 and (o.shadowmap() < 1.0).sum() > 1500
    _identical(o, g, "one long edge")


def test_many_lights_config5_style(oracle_lib, gpu_engine):
    """256 point lights (config 5's count): the zero-radiance skip in k_lighting must stay exact."""
    o, g = _both(oracle_lib, gpu_engine, 160, 96, 128, _mixed_scene, _std_frame(n_point=256))
    _identical(o, g, "256 lights")


def test_tile_light_lists_scattered_and_odd_lights(oracle_lib, gpu_engine):
    """300 point lights spread over the scene with small radii (most tiles see few of them), plus lights the tile test must
    not drop: zero / negative radius, non-finite colour, non-finite position.  Sky pixels (Mask = 0) are in the frame too."""
    rng = np.random.default_rng(11)
    w = scenes.sample_world()
    d, _, s = scenes.lights_from_world(w)
    n = 300
    p = np.zeros(n, dtype=abi.XkLight)
    for i in range(n):
        pos = (rng.uniform(-9, 9), rng.uniform(-9, 9), rng.uniform(0.1, 2.5))
        p[i] = abi.make_light(pos, 1, tuple(rng.uniform(0.1, 1.0, 3)), rng.uniform(2.0, 12.0), (0.0, 0.0, 1.0), rng.uniform(0.3, 2.5))
    p[7] = abi.make_light((1.0, 1.0, 1.0), 1, (1.0, 0.5, 0.2), 5.0, (0, 0, 1), 0.0)            # radius 0: 0/0
    p[19] = abi.make_light((-2.0, 1.0, 0.5), 1, (1.0, 0.5, 0.2), 5.0, (0, 0, 1), -1.0)
    p[33] = abi.make_light((50.0, 50.0, 1.0), 1, (float("nan"), 0.5, 0.2), 5.0, (0, 0, 1), 0.5)  # far away but NaN colour
    p[64] = abi.make_light((float("inf"), 0.0, 1.0), 1, (1.0, 0.5, 0.2), 5.0, (0, 0, 1), 1.0)
    p[65] = abi.make_light((float("nan"), 0.0, 1.0), 1, (1.0, 0.5, 0.2), 5.0, (0, 0, 1), 1.0)
    p[299] = abi.make_light((0.0, 0.0, 1.0), 1, (0.2, 0.9, 0.2), float("inf"), (0, 0, 1), 3.0)

    def frame(r):
        r.update_uniforms(abi.make_camera((7.0, -9.0, 4.0), (0.0, 0.0, 0.6), fov=55.0, zfar=80.0), d, p, s, 0.2, 0.0, 1.0)
    o, g = _both(oracle_lib, gpu_engine, 320, 192, 128, _mixed_scene, frame)
    assert (o.gbuffer(0) == 1.0).sum() > 1000            # sky pixels present
    _identical(o, g, "300 scattered lights")


def test_staged_frame_and_shadow_instance_partition(oracle_lib, gpu_engine):
    """zr_render_shadow/gbuffer/lighting == zr_render; per-rank shadow shares min-reduce to the full shadow map."""
    cfg = scenes.config3(500, 320, 200)
    full = gpu_engine.Renderer(cfg["width"], cfg["height"], 512)
    gpu_engine.load_scene(full, cfg)
    full.render()
    want_color, want_shadow = full.color(), full.shadowmap()
    shares = []
    for world in (1, 3):
        maps = []
        for rank in range(world):
            g = gpu_engine.Renderer(cfg["width"], cfg["height"], 512)
            gpu_engine.load_scene(g, cfg)
            g.set_shadow_partition(rank, world)
            g.render_shadow()
            with pytest.raises(gpu_engine.ZeldaRenderError):
                g.render_lighting()                      # stages must run in order
            g.render_gbuffer(); g.render_lighting(); g.finish()
            maps.append(g.shadowmap())
            if world == 1:
                assert np.array_equal(g.color(), want_color)
        shares.append(np.minimum.reduce(maps))
        assert np.array_equal(shares[-1].view(np.uint32), want_shadow.view(np.uint32))
    assert (maps[0] != want_shadow).any()                # a single share really is partial


def test_geometry_stage_runs_both_passes_side_by_side(oracle_lib, gpu_engine):
    """zr_render_geometry (shadow pass on the library's second stream next to the deferred-scene pass) + zr_stream_wait_shadow
    + zr_render_lighting, as a multi-GPU host drives it, over a few frames; and the single-stream flag gives the same frame."""
    import torch
    cfg = scenes.config3(500, 320, 200)
    full = gpu_engine.Renderer(cfg["width"], cfg["height"], 512, flags=abi.FLAG_SERIAL_PASSES)
    gpu_engine.load_scene(full, cfg)
    full.render()
    want_color, want_shadow = full.color(), full.shadowmap()
    g = gpu_engine.Renderer(cfg["width"], cfg["height"], 512)
    gpu_engine.load_scene(g, cfg)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    done = torch.cuda.Event()
    for _ in range(3):
        g.render_geometry()
        with pytest.raises(gpu_engine.ZeldaRenderError):
            g.render_gbuffer()                                   # already part of the geometry stage
        g.stream_wait_shadow(side.cuda_stream)                   # where a host would all-reduce the shadow map
        done.record(side)
        main.wait_event(done)
        g.stream_wait_shadow(main.cuda_stream)
        g.render_lighting()
    g.finish()
    assert np.array_equal(g.color(), want_color)
    assert np.array_equal(g.shadowmap().view(np.uint32), want_shadow.view(np.uint32))
    g.render(); g.finish()                                       # the fused call after staged frames
    assert np.array_equal(g.color(), want_color)


def test_instance_level_precull_path(oracle_lib, gpu_engine):
    """>= 65536 instances switch on the two-level cull (instance spheres, then meshlets of the survivors): same frame."""
    v, idx = scenes.uv_sphere(8, 4, 0.5)                      # 48 triangles: keeps the oracle quick
    inst = scenes.generate_instances(70000, 2.0, 40.0, 0.1, 0.5, seed=21)

    def build(r):
        r.set_cubemap(None)
        r.object_add(r.mesh_create(v, idx), None, inst)
        r.object_add(r.mesh_create(*scenes.grid_plane(100.0, 4, -0.3)))
    o, g = _both(oracle_lib, gpu_engine, 320, 180, 256, build, _std_frame(abi.make_camera((9.0, 9.0, 4.0), (0.0, 0.0, 0.0))))
    _identical(o, g, "instance-level pre-cull")
    st = g.stats()
    assert st["work_items"][1] >= 70000 and st["survivors"][1] < 0.6 * st["work_items"][1]


def test_errors_are_reported_not_swallowed(gpu_engine):
    g = gpu_engine.Renderer(64, 64, 64)
    with pytest.raises(gpu_engine.ZeldaRenderError) as e:
        g.render()
    assert e.value.code == -6                                   # no frame uniforms yet
    v, idx = scenes.box()
    bad = idx.copy(); bad[0] = 10 ** 6
    with pytest.raises(gpu_engine.ZeldaRenderError):
        g.mesh_create(v, bad)
    m = g.mesh_create(v, idx)
    with pytest.raises(gpu_engine.ZeldaRenderError) as e:
        g.object_add(m + 5)
    assert e.value.code == -1


def _textures(seed=0):
    """7 non-constant RGBA8 images of assorted (also non-power-of-two) sizes."""
    rng = np.random.default_rng(seed)
    def checker(w, h, a, b, cell):
        yy, xx = np.mgrid[0:h, 0:w]
        m = (((xx // cell) + (yy // cell)) & 1).astype(bool)
        img = np.empty((h, w, 4), np.uint8); img[m] = a; img[~m] = b
        return img
    def noise(w, h, lo, hi):
        img = rng.integers(lo, hi, size=(h, w, 4), dtype=np.uint8); img[..., 3] = 255
        return img
    nrm = noise(48, 20, 100, 156); nrm[..., 2] = 255
    return [checker(64, 64, (220, 60, 40, 255), (40, 90, 200, 255), 8), noise(16, 16, 0, 255), noise(37, 23, 30, 255), nrm,
            checker(8, 32, (255, 255, 255, 255), (120, 120, 120, 255), 2), noise(5, 3, 0, 60), checker(32, 32, (255,) * 4, (200, 200, 200, 255), 16)]


def test_textured_materials_trilinear_repeat_srgb(oracle_lib, gpu_engine):
    """N2: real material textures - blit-generated mips, sRGB base colour, REPEAT addressing, LOD from quad derivatives."""
    def build(r):
        r.set_cubemap(scenes.synthetic_cubemap(16))
        mat, keep = abi.make_material(_textures())
        r._keepalive = keep
        v, idx = scenes.grid_plane(24.0, 6, 0.0)
        v = v.copy(); v["TexCoord"] *= 5.0                     # tiles: exercises REPEAT and minification far away
        r.object_add(r.mesh_create(v, idx), mat)
        r.object_add(r.mesh_create(*scenes.uv_sphere()), mat, scenes.generate_instances(60, 1.0, 6.0, 0.4, 1.2, seed=9))
        mat2, keep2 = abi.make_material([None, None, None, _textures(3)[3], None, None, None])   # only the normal map sampled
        r._keepalive2 = keep2
        r.object_add(r.mesh_create(*scenes.box((0.7, 0.7, 0.7), (0, 0, 0.7))), mat2)
    o, g = _both(oracle_lib, gpu_engine, 400, 240, 256, build, _std_frame(abi.make_camera((5.0, 4.0, 2.5), (0.0, 0.0, 0.3))))
    assert len(np.unique(o.gbuffer(4))) > 200                  # base colour really varies
    _identical(o, g, "textured")


def test_materials_whose_images_share_a_size_are_sampled_packed(oracle_lib, gpu_engine):
    """The usual material: every image slot of one size.  The library interleaves such slots into one 16-byte texel per mip texel
    (ZrObject::packed) and the resolve reads a tap's texels of all seven slots with one load each; constant slots of such a material
    stay constants (the oracle does not filter them: (c + c + c) / 3 is not c).  Three materials - all seven slots sampled,
    three sampled + four constant (one of them a constant IMAGE), one image only - over a tiled ground seen at a grazing angle
    (every tap count) and instanced spheres (minification), with a skydome image beside them."""
    rng = np.random.default_rng(21)

    def noise(w, h, lo=0, hi=256):
        img = rng.integers(lo, hi, size=(h, w, 4), dtype=np.uint8); img[..., 3] = 255
        return img
    full = [noise(64, 32) for _ in range(7)]
    full[3] = noise(64, 32, 96, 160); full[3][..., 2] = 255                 # a plausible normal map
    full[6][..., 0] = 255                                                   # mask r = 1: lit
    some = [noise(16, 16), None, noise(16, 16, 40, 255), None, np.full((16, 16, 4), 200, np.uint8), None, None]
    some[6] = noise(16, 16); some[6][..., 0] = 255
    one = [None, None, noise(40, 24, 30, 255), None, None, None, None]

    def build(r):
        r.set_cubemap(scenes.synthetic_cubemap(16))
        r.set_skydome(*scenes.sky_dome(), scenes.synthetic_sky_image())
        keep = []
        mats = []
        for spec in (full, some, one):
            m, k = abi.make_material(spec); mats.append(m); keep.append(k)
        r._keepalive = keep
        v, idx = scenes.grid_plane(40.0, 6, 0.0)                # (its corners lie outside the dome, radius 20.48)
        v = v.copy(); v["TexCoord"] *= 7.0
        r.object_add(r.mesh_create(v, idx), mats[0])
        r.object_add(r.mesh_create(*scenes.uv_sphere()), mats[1], scenes.generate_instances(80, 1.0, 7.0, 0.3, 1.1, seed=4))
        r.object_add(r.mesh_create(*scenes.uv_sphere(12, 6, 0.5)), mats[0], scenes.generate_instances(40, 1.0, 9.0, 0.5, 1.4, seed=5))
        r.object_add(r.mesh_create(*scenes.box((0.9, 0.9, 0.9), (0, 0, 0.9))), mats[2])
    cam = abi.make_camera((0.0, -9.0, 0.8), (0.0, 6.0, 0.4), fov=55.0, znear=0.05, zfar=120.0)
    o, g = _both(oracle_lib, gpu_engine, 400, 240, 256, build, _std_frame(cam))
    assert len(np.unique(o.gbuffer(4))) > 500 and len(np.unique(o.gbuffer(1))) > 100
    _identical(o, g, "packed materials")


def test_geometry_behind_the_skydome_keeps_its_gbuffer(oracle_lib, gpu_engine):
    """The skydome is drawn after the lighting quad, depth-tested, colour only (ZE:3681-3691): scene geometry that lies OUTSIDE the dome
    (radius 20.48) is hidden by it in the image but stays in every GBuffer attachment, and is lit like any other pixel before the dome
    covers it.  The library resolves the dome in a key plane of its own for that (k_sky_tiles)."""
    def build(r):
        r.set_cubemap(scenes.synthetic_cubemap(16))
        r.set_skydome(*scenes.sky_dome(), scenes.synthetic_sky_image())
        r.object_add(r.mesh_create(*scenes.grid_plane(90.0, 6, 0.0)))                     # corners 64 units out: far beyond the dome
        r.object_add(r.mesh_create(*scenes.uv_sphere()), None, scenes.generate_instances(120, 2.0, 40.0, 0.5, 2.0, seed=8))   # inside and outside
    cam = abi.make_camera((0.0, -12.0, 2.0), (0.0, 30.0, 1.0), fov=60.0, znear=0.1, zfar=200.0)
    o, g = _both(oracle_lib, gpu_engine, 400, 240, 256, build, _std_frame(cam))
    depth = o.gbuffer(0)
    dome = (o.color() != 0).any(axis=2) & (depth < 1.0)
    assert (depth < 1.0).sum() > 400 * 240 // 3                   # the plane fills the lower half, dome pixels above AND over its far part
    _identical(o, g, "geometry behind the dome")
    g.render(); g.finish()                                        # and again with the visibility history of the first frame
    _identical(o, g, "geometry behind the dome, second frame")


def test_anisotropic_filtering_at_grazing_angles(oracle_lib, gpu_engine):
    """Samplers have anisotropy on at the device maximum (ZE:6540): a striped ground plane seen almost edge-on gives footprints
    from 1:1 near the camera to beyond 16:1 at the horizon, i.e. every tap count of the scheme."""
    def build(r):
        r.set_cubemap(None)
        img = np.zeros((64, 64, 4), dtype=np.uint8)
        img[..., 3] = 255
        img[:, ::2, :3] = 255                                   # one-texel stripes: the harshest case for minification
        img[::8, :, 0] = 40
        mat, keep = abi.make_material([img, None, None, None, None, None, None])
        r._keepalive = keep
        v, idx = scenes.grid_plane(60.0, 4, 0.0)
        v = v.copy(); v["TexCoord"] *= 3.0
        r.object_add(r.mesh_create(v, idx), mat)
    cam = abi.make_camera((0.0, -20.0, 0.35), (0.0, 20.0, 0.0), fov=50.0, znear=0.05, zfar=200.0)
    o, g = _both(oracle_lib, gpu_engine, 320, 200, 64, build, _std_frame(cam, n_point=2))
    bc = o.gbuffer(4)
    assert len(np.unique(bc)) > 50                             # filtered greys, not just the two stripe colours
    _identical(o, g, "anisotropic")


def test_record_pool_and_slow_list_overflow_are_reported(oracle_lib, gpu_engine, monkeypatch):
    """The triangle-binned camera pass keeps its records in a pool of chunks and its clipped triangles in a list; either running dry
    must surface as ZR_ERR_OVERFLOW at zr_finish (never as a silently incomplete frame), and the same scene must render exactly once
    the capacity is there again."""
    import math
    W, H, SD = 320, 200, 128

    def scene(r):
        r.set_cubemap(scenes.synthetic_cubemap(8))
        r.object_add(r.mesh_create(*scenes.grid_plane(40.0, 6, 0.0)))              # crosses the near plane: clipped triangles
        r.object_add(r.mesh_create(*scenes.uv_sphere()), None, scenes.generate_instances(400, 0.5, 6.0, 0.3, 0.8, seed=3))
        _std_frame()(r)

    o = oracle_lib.Oracle(W, H, SD)
    scene(o)
    o.render()
    for limits, what in (((8, 0), "triangle-record arrays"), ((0, 2), "slow-triangle list")):      # (record chunks, clipped triangles); 0 = default
        g = gpu_engine.Renderer(W, H, SD)
        g.set_limits(*limits)
        scene(g)
        g.render()
        with pytest.raises(gpu_engine.ZeldaRenderError) as e:
            g.finish()
        assert e.value.code == abi.ERR_OVERFLOW, (limits, e.value)
        assert what in str(e.value), (limits, str(e.value))       # zr_last_error names the capacity that ran full
        g.set_limits(0, 0)                             # the same context with the default pools: the frame is whole again
        g.render(); g.render(); g.finish()
        d = compare_all(o, g)
        assert not {k: v for k, v in d.items() if v}, (limits, d)
        g.close()
    g = gpu_engine.Renderer(W, H, SD)
    scene(g)
    g.render(); g.render(); g.finish()
    o.render()
    d = compare_all(o, g)
    assert not {k: v for k, v in d.items() if v}, d
    assert g.stats()["overflow"] == 0
    g.close()


def test_records_beyond_their_buckets_are_drawn_from_the_overflow_region(oracle_lib, gpu_engine):
    """The camera pass keeps a tile's triangle records in a bucket sized from the PREVIOUS frame's count; what a tile gets beyond it goes to
    the overflow region behind the buckets (64 sections by tile number) that the tile's first work unit sifts.  zr_set_bucket_share plans
    every bucket at that share of its size: part of every tile's records then takes the overflow route - a few dozen per tile (one
    sifted batch) with one scene, thousands per tile (several batches, the waves resuming their sift) with the other - and the frame must
    not change by a bit.  A camera that jumps between frames does the same to the default plan."""
    W, H, SD = 320, 200, 128

    def scene(r, n):
        r.set_cubemap(scenes.synthetic_cubemap(8))
        r.object_add(r.mesh_create(*scenes.uv_sphere()), None, scenes.generate_instances(n, 0.5, 6.0, 0.3, 0.8, seed=3))
        _std_frame()(r)

    for n_inst, pcts in ((60, (70, 30)), (1500, (90, 45, 5))):      # ~ 250 / ~ 6 400 records per tile
        o = oracle_lib.Oracle(W, H, SD); scene(o, n_inst); o.render()
        for pct in pcts:
            g = gpu_engine.Renderer(W, H, SD, flags=abi.FLAG_NO_HIZ)
            g.set_bucket_share(pct)
            scene(g, n_inst)
            for k in range(3):                         # counted and planned from its own counts, then planned from the frame before
                g.render(); g.finish()
                st = g.stats()
                assert st["overflow"] == 0
                bad = {k2: v for k2, v in compare_all(o, g).items() if v}
                assert not bad, (n_inst, pct, k, bad)
            g.close()
    # a camera cut: the plan is last frame's, every tile's count is new
    o = oracle_lib.Oracle(W, H, SD); g = gpu_engine.Renderer(W, H, SD)
    for r in (o, g):
        scene(r, 1500)
    for k, cam in enumerate([abi.make_camera(), abi.make_camera((-6.0, 3.0, 2.0), (1.0, 0.5, 0.0)), abi.make_camera((2.0, -7.0, 6.0), (0.0, 0.0, 0.0)), abi.make_camera()]):
        for r in (o, g):
            _std_frame(cam)(r)
        o.render(); g.render(); g.finish()
        bad = {k2: v for k2, v in compare_all(o, g).items() if v}
        assert not bad, (k, bad)
