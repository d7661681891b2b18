"""GPU: "existing scenes drop in" through the library's OWN file-system code, and through a native (C++) caller of the C-ABI.

A content tree is written to a temporary directory the way the engine lays it out (Profabs/<name>/{models,textures}, Content/Models,
Content/Textures, Content/World.json with the golden livelink payload's file names).  Then

* `zr_set_asset_root` + `zr_world_load_file`: the library finds the Profabs, reads OBJ and PNG files with its own loaders, applies the
  world's OverrideCubemap / OverrideSkydome / OverrideBackground through AssetPathSearch (ZE:7173-7263, ZE:4147-4183), and the frame is
  compared bit for bit with the oracle, which is fed by the Python loaders / PIL instead;
* `tools/zelda_headless` (C++, links -lzelda_render, built by __graft_entry__.build()) does the same from `main()`, including a
  `.meshlet` file written by the meshlet tool (LoadMeshletAsset, ZE:7046-7169), and its PPM must hold the same pixels.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from parity_util import compare_all
from zeldaengine_amd import abi, assets, build as zbuild, livelink, scenes

pytestmark = pytest.mark.gpu
W, H, SD = 320, 192, 256


def _png(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path)


def _pattern(h, w, seed, channels=3):
    rng = np.random.default_rng(seed)
    t = (np.arange(w, dtype=np.float32)[None, :] / w + np.arange(h, dtype=np.float32)[:, None] / h)
    img = np.zeros((h, w, channels), dtype=np.uint8)
    for k in range(channels):
        img[..., k] = (40 + 170 * ((t * (2 + k) + 0.13 * seed) % 1.0) + rng.integers(0, 20, size=(h, w))).astype(np.uint8)
    if channels == 4:
        img[..., 3] = 255
    return img


def _content_tree(root):
    """-> description of what was written: {profab: [(obj path, [7 texture paths or None])]} in the library's load order."""
    tex = os.path.join(root, "Content", "Textures")
    os.makedirs(tex)
    os.makedirs(os.path.join(root, "Content", "Models"))
    world = scenes.sample_world()
    for f, name in enumerate(world["Skydome"]["CubemapFileNames"]):
        _png(os.path.join(tex, name), _pattern(64, 64, 10 + f))
    _png(os.path.join(tex, world["Skydome"]["SkydomeFileName"]), _pattern(64, 128, 3))
    _png(os.path.join(tex, world["Background"]["BackgroundFileName"]), _pattern(48, 96, 4, 4))
    assets.write_obj(os.path.join(root, "Content", "Models", "skydome.obj"), *scenes.sky_dome(20.48, 16, 8))
    meshes = {"terrain": [("terrain", scenes.grid_plane(24.0, 6, 0.0))],
              "rock_01": [("rock", scenes.box((0.9, 0.6, 0.5), (1.5, -1.0, 0.5)))],
              "rock_02": [("a_pebble", scenes.box((0.5, 0.5, 0.3), (0, 0, 0.3))), ("b_ball", scenes.uv_sphere(12, 6, 0.5))],
              "grass_01": [("blade", scenes.uv_sphere(8, 4, 0.4))]}
    textured = {"terrain": {"bc": _pattern(32, 32, 21), "r": _pattern(16, 16, 22)}, "rock": {"bc": _pattern(16, 32, 23), "n": _pattern(16, 16, 24), "ao": _pattern(8, 8, 25)},
                "b_ball": {"m": _pattern(8, 8, 26), "ev": _pattern(8, 8, 27)}}
    desc = {}
    for name, models in meshes.items():
        md, td = os.path.join(root, "Profabs", name, "models"), os.path.join(root, "Profabs", name, "textures")
        os.makedirs(md); os.makedirs(td)
        desc[name] = []
        for stem, (v, idx) in sorted(models):
            obj = os.path.join(md, stem + ".obj")
            assets.write_obj(obj, v, idx)
            paths = []
            for suf in assets.TEXTURE_SUFFIXES:
                img = textured.get(stem, {}).get(suf)
                p = os.path.join(td, "%s_%s.png" % (stem, suf))
                if img is not None:
                    _png(p, img)
                paths.append(p if img is not None else None)
            desc[name].append((obj, paths))
        open(os.path.join(md, "readme.txt"), "w").write("not a model")          # non-.obj files are skipped (ZE:4949-4951)
    os.makedirs(os.path.join(root, "Profabs", "no_models_dir", "textures"))      # an asset set without models/: ignored
    world["Objects"][2]["InstanceCount"] = 24
    world["Objects"][3]["InstanceCount"] = 300
    world["Objects"][4]["InstanceCount"] = 40           # grass_02: no such Profab on disk -> nothing drawn, unless registered by the host
    world["MainCamera"]["Position"] = [6.0, 5.0, 4.0]
    with open(os.path.join(root, "Content", "World.json"), "w") as f:
        json.dump(world, f)
    return desc, world


def _oracle_for(oracle_lib, g, desc, world, root, extra_meshes=()):
    """The oracle's scene from the files on disk (Python loaders, PIL) and from what the library reports about its own scene."""
    o = oracle_lib.Oracle(W, H, SD)
    tex = os.path.join(root, "Content", "Textures")
    o.set_cubemap([assets.load_image_rgba8(os.path.join(tex, n)) for n in world["Skydome"]["CubemapFileNames"]])
    sv, si = assets.load_obj(os.path.join(root, "Content", "Models", "skydome.obj"))
    o.set_skydome(sv, si, assets.load_image_rgba8(os.path.join(tex, world["Skydome"]["SkydomeFileName"])))
    o.set_background(assets.load_image_rgba8(os.path.join(tex, world["Background"]["BackgroundFileName"])))
    o.set_sky_flags(world["Skydome"]["EnableSkydome"], world["Background"]["EnableBackground"])
    # mesh ids in the library: host-registered ones first (extra_meshes), then Profabs in the order the world names them
    by_id, keep = {}, []
    for k, m in enumerate(extra_meshes):
        by_id[k] = (m, None)
    next_id = len(extra_meshes)
    seen = set()
    for od in world["Objects"]:
        n = od["ProfabName"]
        if n in seen or n not in desc:
            continue
        seen.add(n)
        for obj, paths in desc[n]:
            by_id[next_id] = (assets.load_obj(obj), paths)
            next_id += 1
    cache = {}
    for i in range(g.object_count()):
        mesh_id, inst = g.object_get_instances(i)
        (v, idx), paths = by_id[mesh_id]
        if mesh_id not in cache:
            cache[mesh_id] = o.mesh_create(v, idx)
        mat = None
        if paths:
            mat, k = abi.make_material([assets.load_image_rgba8(p) if p else None for p in paths])
            keep.append(k)
        o.object_add(cache[mesh_id], mat, inst)
    o._keep2 = keep
    o.set_frame(*g.get_frame())
    return o


def test_content_tree_drops_in_through_the_library(oracle_lib, gpu_engine, tmp_path):
    root = str(tmp_path)
    desc, world = _content_tree(root)
    g = gpu_engine.Renderer(W, H, SD)
    g.set_asset_root(root)
    # AssetPathSearch: literal path, then Profabs/*/{models,textures}, then Content/*/{Models,Textures}
    assert g.asset_path_search("grassland_night_Y2.png") == os.path.join(root, "Content", "Textures", "grassland_night_Y2.png")
    assert g.asset_path_search("some/dir/rock_bc.png") == os.path.join(root, "Profabs", "rock_01", "textures", "rock_bc.png")
    assert g.asset_path_search("Content/Models/skydome.obj") == os.path.join(root, "Content", "Models", "skydome.obj")
    assert g.asset_path_search("nowhere.png") == os.path.join(root, "nowhere.png")
    g.world_load_file()                                  # Content/World.json
    assert g.object_count() == 1 + 1 + 2 + 1             # terrain, rock, two models of rock_02, grass_01; grass_02 has no directory
    g.render(); g.render(); g.finish()
    o = _oracle_for(oracle_lib, g, desc, world, root)
    o.render(0)
    bad = {k: v for k, v in compare_all(o, g).items() if v}
    assert not bad, bad
    assert o.covered_pixels() > 5000
    frame = g.color()
    assert len(np.unique(frame.reshape(-1, 4), axis=0)) > 500      # sky, background and textured materials all contribute
    # the world's override flags matter: without the skydome override the sky keeps whatever was set before (here: nothing)
    h = gpu_engine.Renderer(W, H, SD)
    h.set_asset_root(root)
    w2 = json.loads(json.dumps(world)); w2["Skydome"]["OverrideSkydome"] = False; w2["Background"]["OverrideBackground"] = False
    h.world_load_json(json.dumps(w2))
    h.render(); h.finish()
    assert not np.array_equal(h.color(), frame)
    # a missing override file is an error, not a silent default (the engine throws in LoadTextureAsset, ZE:6887-6890)
    w3 = json.loads(json.dumps(world)); w3["Background"]["BackgroundFileName"] = "does_not_exist.png"
    with pytest.raises(gpu_engine.ZeldaRenderError) as e:
        h.world_load_json(json.dumps(w3))
    assert e.value.code == abi.ERR_IO
    # World.json round trip on disk
    g.world_save_file("Content/Saved.json")
    assert gpu_engine.world_json_normalize(open(os.path.join(root, "Content", "Saved.json")).read()) == g.world_save_json()


def test_native_headless_driver_renders_the_same_pixels(oracle_lib, gpu_engine, tmp_path):
    root = str(tmp_path)
    desc, world = _content_tree(root)
    # a .meshlet file from the meshlet tool stands in for the missing grass_02 Profab (the XkMeshIndirect path)
    bv, bi = scenes.uv_sphere(16, 8, 0.45)
    assets.write_obj(os.path.join(root, "ball.obj"), bv, bi)
    assets.meshlet_tool(os.path.join(root, "ball.obj"), os.path.join(root, "ball.meshlet"))
    exe = zbuild.build_headless()
    ppm = os.path.join(root, "frame.ppm")
    out = subprocess.run([exe, "--root", root, "--world", "Content/World.json", "--meshlet", "ball.meshlet", "--profab", "grass_02",
                          "--size", "%dx%d" % (W, H), "--shadow", str(SD), "--frames", "2", "--out", ppm],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode(errors="replace")
    assert out.returncode == 0, text
    assert "objects 6" in text, text
    raw = open(ppm, "rb").read()
    header = ("P6\n%d %d\n255\n" % (W, H)).encode()
    assert raw.startswith(header) and len(raw) == len(header) + W * H * 3
    got = np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(H, W, 3)
    # the same steps through ctypes, and the oracle fed by the Python loaders
    g = gpu_engine.Renderer(W, H, SD)
    g.set_asset_root(root)
    m = g.load_meshlet_file("ball.meshlet")
    g.profab_register("grass_02", m)
    g.world_load_file("Content/World.json")
    assert g.object_count() == 6
    g.render(); g.render(); g.finish()
    assert np.array_equal(got, g.color()[..., :3]), "the native driver's frame differs from the ctypes-driven one"
    f = assets.read_meshlet(os.path.join(root, "ball.meshlet"))
    flat = []
    for ml in f["meshlets"]:
        for t in range(int(ml["TriangleCount"])):
            for k in range(3):
                flat.append(f["mverts"][int(ml["VertexOffset"]) + int(f["mtris"][int(ml["TriangleOffset"]) + 3 * t + k])])
    o = _oracle_for(oracle_lib, g, desc, world, root, extra_meshes=[(f["vertices"], np.asarray(flat, dtype=np.uint32))])
    o.render(0)
    bad = {k: v for k, v in compare_all(o, g).items() if v}
    assert not bad, bad
    assert np.array_equal(got, o.color()[..., :3])


def test_native_headless_driver_takes_the_scene_from_the_livelink(gpu_engine, tmp_path):
    """zelda_headless --livelink: the reference client's bytes arrive over TCP, the driver's frame loop picks them up (DrawFrame's
    bReloadScene, ZE:1943-1951) and renders; the frame equals the one rendered from the same JSON loaded from a file."""
    import re
    import time
    root = str(tmp_path)
    _desc, world = _content_tree(root)
    exe = zbuild.build_headless()
    ppm = os.path.join(root, "live.ppm")
    proc = subprocess.Popen([exe, "--root", root, "--livelink", "0", "--wait-ms", "20000", "--size", "%dx%d" % (W, H), "--shadow", str(SD),
                             "--frames", "2", "--out", ppm], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        line = proc.stdout.readline().decode()
        port = int(re.search(r"port (\d+)", line).group(1))
        for _ in range(50):
            if livelink.send_world(world, port=port, host="127.0.0.1") is not None:
                break
            time.sleep(0.1)
        rest = proc.communicate(timeout=120)[0].decode(errors="replace")
    finally:
        if proc.poll() is None:
            proc.kill()
    assert proc.returncode == 0, line + rest
    header = ("P6\n%d %d\n255\n" % (W, H)).encode()
    got = np.frombuffer(open(ppm, "rb").read()[len(header):], dtype=np.uint8).reshape(H, W, 3)
    g = gpu_engine.Renderer(W, H, SD)
    g.set_asset_root(root)
    g.world_load_json(json.dumps(world))
    g.render(); g.render(); g.finish()
    assert np.array_equal(got, g.color()[..., :3])


def test_hostile_inputs_are_errors_not_crashes(gpu_engine, tmp_path):
    """A corrupt `.meshlet` header must not be able to ask for gigabytes (every section count is checked against the bytes the file
    still holds), and file names in a world payload - the livelink is unauthenticated - must stay inside the content tree."""
    import struct
    root = str(tmp_path)
    _, world = _content_tree(root)
    g = gpu_engine.Renderer(64, 64, 64)
    g.set_asset_root(root)
    for payload in (struct.pack("<Q", (1 << 31)),                                        # 2^31 meshlets = 128 GiB, 8 bytes on disk
                    struct.pack("<Q", 3) + b"\0" * 64,                                   # three records promised, one present
                    struct.pack("<Q", 0) * 4 + struct.pack("<Q", (1 << 31) - 1)):        # empty sections, then a huge index count
        p = os.path.join(root, "bad.meshlet")
        open(p, "wb").write(payload)
        with pytest.raises(gpu_engine.ZeldaRenderError) as e:
            g.load_meshlet_file(p)
        assert e.value.code == abi.ERR_IO
    for key, section, bad in (("BackgroundFileName", "Background", "/etc/passwd"), ("BackgroundFileName", "Background", "../../secret.png"),
                              ("SkydomeFileName", "Skydome", "Content/../../x.png")):
        w = json.loads(json.dumps(world)); w[section][key] = bad
        with pytest.raises(gpu_engine.ZeldaRenderError) as e:
            g.world_load_json(json.dumps(w))
        assert e.value.code == abi.ERR_ARG, (bad, e.value)
    w = json.loads(json.dumps(world)); w["Objects"][0]["ProfabName"] = "../Content"
    with pytest.raises(gpu_engine.ZeldaRenderError) as e:
        g.world_load_json(json.dumps(w))
    assert e.value.code == abi.ERR_ARG
    g.world_load_json(json.dumps(world))          # and the sound world still loads on the same context
    g.render(); g.finish()
    g.close()
