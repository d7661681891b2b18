"""File side of the asset contract (CPU): OBJ ingest quirks, the .meshlet container, Profab directory search."""
import os

import numpy as np
import pytest

from zeldaengine_amd import abi, assets, engine, scenes

OBJ = """# two triangles sharing an edge, a quad face, per-corner uv and normal indices
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vt 0.5 0.25
vn 0 0 1
vn 0 1 0
vn 1 0 0
vn 0.6 0 0.8
f 1/1/4 2/2/4 3/3/4 4/4/4
f 1/5/1 3/3/2 -1/4/3
"""


def test_obj_ingest_quirks(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text(OBJ)
    v, idx = assets.load_obj(str(p))
    assert len(idx) == 9                                        # quad -> fan of 2, plus 1 triangle
    assert idx.tolist() == [0, 1, 2, 0, 2, 3, 4, 2, 3]          # vertex 1 with uv 5 is a new record; 3 and 4 are reused
    assert np.allclose(v["Color"], 1.0)
    assert np.allclose(v[1]["TexCoord"], (1.0, 1.0 - 0.0)) and np.allclose(v[4]["TexCoord"], (0.5, 0.75))     # v flipped
    # quirk (ZE:6927-6931): the normal comes from the POSITION index, not the face's normal index
    assert np.allclose(v[0]["Normal"], (0, 0, 1)) and np.allclose(v[1]["Normal"], (0, 1, 0)) and np.allclose(v[3]["Normal"], (0.6, 0, 0.8))
    fv, fidx = assets.load_obj_for_meshlet_tool(str(p))         # the tool indexes normals properly (ZM:201-203)
    assert np.allclose(fv[1]["nrm"], (0.6, 0, 0.8)) and len(fidx) == 9


def test_obj_roundtrip_of_a_generated_mesh(tmp_path):
    v, idx = scenes.uv_sphere(12, 6, 0.5)
    p = str(tmp_path / "s.obj")
    assets.write_obj(p, v, idx)
    v2, idx2 = assets.load_obj(p)
    assert np.array_equal(idx, idx2) and len(v) == len(v2)
    assert np.allclose(v["Position"], v2["Position"]) and np.allclose(v["TexCoord"], v2["TexCoord"], atol=1e-6)


def test_meshlet_container_layout_and_tool(tmp_path):
    v, idx = scenes.uv_sphere()
    obj = str(tmp_path / "sphere.obj")
    assets.write_obj(obj, v, idx)
    out = str(tmp_path / "sphere.meshlet")
    n = assets.meshlet_tool(obj, out)
    raw = open(out, "rb").read()
    assert int(np.frombuffer(raw[:8], "<u8")[0]) == n           # size_t count, then n * 64 bytes of Meshlet
    m = assets.read_meshlet(out)
    assert len(m["meshlets"]) == n and m["meshlets"].dtype.itemsize == 64
    assert m["mtris"].dtype == np.uint8 and m["file_vertices"].dtype.itemsize == 32
    off = 8 + 64 * n
    assert int(np.frombuffer(raw[off:off + 8], "<u8")[0]) == len(m["mverts"])
    assert len(raw) == 5 * 8 + 64 * n + 4 * len(m["mverts"]) + len(m["mtris"]) + 32 * len(m["file_vertices"]) + 4 * len(m["indices"])
    assert m["meshlets"]["TriangleCount"].sum() == len(m["indices"]) // 3 == 960
    assert (m["vertices"]["Color"] == 1.0).all()
    # the payload is accepted by the context-free validator path: rebuild from it gives the same partition sizes
    ml2, _, _, _ = engine.build_meshlets(m["vertices"], m["indices"])
    assert len(ml2) == n
    with open(out, "wb") as f:
        f.write(raw[:100])
    with pytest.raises(ValueError):
        assets.read_meshlet(out)


def test_profab_directory_search(tmp_path):
    from PIL import Image
    root = tmp_path / "Profabs"
    for name in ("rock_01", "broken"):
        os.makedirs(root / name / "models")
    os.makedirs(root / "rock_01" / "textures")
    v, idx = scenes.box()
    assets.write_obj(str(root / "rock_01" / "models" / "rock_a.obj"), v, idx)
    assets.write_obj(str(root / "rock_01" / "models" / "rock_b.obj"), v, idx)
    (root / "rock_01" / "models" / "notes.txt").write_text("ignored")
    Image.fromarray(np.full((4, 4, 3), 200, np.uint8)).save(str(root / "rock_01" / "textures" / "rock_a_bc.png"))
    Image.fromarray(np.full((2, 2, 3), 10, np.uint8)).save(str(root / "rock_01" / "textures" / "rock_a_r.png"))
    found = assets.find_profabs(str(root))
    assert list(found) == ["rock_01"]                           # "broken" has no textures dir: skipped (ZE:4936-4940)
    (obj_a, tex_a), (obj_b, tex_b) = found["rock_01"]
    assert obj_a.endswith("rock_a.obj") and obj_b.endswith("rock_b.obj")
    assert [t is not None for t in tex_a] == [True, False, True, False, False, False, False]
    assert tex_b == [None] * 7                                  # every slot falls back to the engine default
    img = assets.load_image_rgba8(tex_a[0])
    assert img.shape == (4, 4, 4) and (img[..., 3] == 255).all() and (img[..., :3] == 200).all()


# ---------------------------------------------------------------- the library's own (C++) loaders against the Python ones / PIL

def _obj_with_quirks(path):
    """Faces as quads and fans, negative indices, v//vn and v/vt forms mixed in, more normals than positions."""
    with open(path, "w") as f:
        f.write("# test\no thing\n")
        for p in [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0.5, 0.5, 1)]:
            f.write("v %g %g %g\n" % p)
        for t in [(0, 0), (1, 0), (1, 1), (0, 1), (0.5, 0.5)]:
            f.write("vt %g %g\n" % t)
        for n in [(0, 0, 1), (0, 0, -1), (1, 0, 0), (0, 1, 0), (0.6, 0, 0.8), (0, 0.6, 0.8)]:
            f.write("vn %g %g %g\n" % n)
        f.write("s off\nusemtl none\n")
        f.write("f 1/1/1 2/2/1 3/3/1 4/4/1\n")          # a quad: fan of two triangles
        f.write("f -5/1/2 -4/2/2 -1/5/6\n")             # negative (relative) indices
        f.write("f 2/2/3 3/3/3 5/5/5\nf 3/3 4/4 5/5\nf 4/4/4 1/1/4 5/5/5\n")


def test_native_obj_loader_matches_the_python_ingest(tmp_path):
    p = str(tmp_path / "quirks.obj")
    _obj_with_quirks(p)
    v0, i0 = assets.load_obj(p)
    v1, i1 = engine.load_obj(p)
    assert np.array_equal(i0, i1) and v0.tobytes() == v1.tobytes()
    # and on a mesh written by write_obj (the path Profab trees in the GPU tests take)
    v, idx = scenes.uv_sphere(12, 6, 0.7)
    q = str(tmp_path / "ball.obj")
    assets.write_obj(q, v, idx)
    v0, i0 = assets.load_obj(q)
    v1, i1 = engine.load_obj(q)
    assert np.array_equal(i0, i1) and v0.tobytes() == v1.tobytes()
    with pytest.raises(engine.ZeldaRenderError):
        engine.load_obj(str(tmp_path / "missing.obj"))


@pytest.mark.parametrize("mode,kw", [("RGBA", {}), ("RGB", {}), ("L", {}), ("LA", {}), ("P", {}), ("1", {}), ("RGB", {"interlace": True}),
                                     ("RGBA", {"compress_level": 0}), ("P", {"transparency": 3, "bits": 4})])
def test_native_png_loader_matches_pil(tmp_path, mode, kw):
    from PIL import Image
    rng = np.random.default_rng(7)
    w, h = 37, 23                                           # odd sizes: partial bytes at low bit depths, ragged Adam7 passes
    base = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    im = Image.fromarray(base, "RGBA")
    if mode == "P":
        im = im.convert("RGB").quantize(16 if kw.get("bits") == 4 else 200)
    else:
        im = im.convert(mode)
    path = str(tmp_path / ("t_%s.png" % mode))
    save_kw = {k: v for k, v in kw.items() if k != "interlace"}
    if kw.get("interlace"):
        # PIL cannot write Adam7; assemble one from its non-interlaced IDAT by hand
        path = _write_adam7_rgb(path, np.asarray(im.convert("RGB")))
    else:
        im.save(path, **save_kw)
    want = np.asarray(Image.open(path).convert("RGBA"))
    got = engine.load_png_rgba8(path)
    assert got.shape == want.shape and np.array_equal(got, want)


def _write_adam7_rgb(path, rgb):
    import struct
    import zlib
    h, w, _ = rgb.shape
    raw = b""
    for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        sub = rgb[y0::dy, x0::dx]
        if sub.size == 0:
            continue
        for row in sub:
            raw += b"\x00" + row.tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    return path


def _png_fix_crcs(data):
    """recompute every chunk's CRC after a mutation, so that the damage reaches the decoder instead of dying at the checksum"""
    import struct
    import zlib
    out, pos = bytearray(data[:8]), 8
    while pos + 12 <= len(data):
        n = struct.unpack(">I", data[pos:pos + 4])[0]
        if pos + 12 + n > len(data):
            break
        t, d = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        out += data[pos:pos + 8] + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
        pos += 12 + n
    return bytes(out + data[pos:])


N_MUT = int(os.environ.get("ZR_MUTATIONS", "150"))


def test_mutated_png_and_obj_files_are_errors_not_crashes(tmp_path):
    """The loaders read files a livelink world names: a damaged or hostile file must come back as an error code (or as some image /
    mesh), never as a crash or an out-of-bounds access (run under a host AddressSanitizer build with ZR_MUTATIONS=5000 when the loaders
    change).  Mutations: byte flips, truncations, length / dimension fields blown up, chunks duplicated; CRCs repaired half the time."""
    import struct
    from PIL import Image
    rng = np.random.default_rng(99)
    seeds = []
    for mode in ("RGBA", "RGB", "L", "LA", "P", "1"):
        im = Image.fromarray(rng.integers(0, 256, size=(19, 31, 4), dtype=np.uint8), "RGBA")
        im = im.convert("RGB").quantize(16) if mode == "P" else im.convert(mode)
        p = str(tmp_path / ("s_%s.png" % mode)); im.save(p)
        seeds.append(open(p, "rb").read())
    seeds.append(open(_write_adam7_rgb(str(tmp_path / "s_a7.png"), rng.integers(0, 256, size=(13, 17, 3), dtype=np.uint8)), "rb").read())
    ok = bad = 0
    for k in range(N_MUT):
        data = bytearray(seeds[k % len(seeds)])
        kind = int(rng.integers(0, 5))
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                data[int(rng.integers(8, len(data)))] ^= int(rng.integers(1, 256))
        elif kind == 1:
            data = data[:int(rng.integers(0, len(data)))]
        elif kind == 2:                                            # IHDR fields: width, height, depth, colour type, interlace
            off = 16 + int(rng.integers(0, 13))
            data[off] = int(rng.choice([0, 1, 3, 7, 16, 64, 255]))
        elif kind == 3:                                            # a chunk length blown up
            pos = 8
            lens = []
            while pos + 12 <= len(data):
                lens.append(pos); pos += 12 + struct.unpack(">I", data[pos:pos + 4])[0]
            at = lens[int(rng.integers(0, len(lens)))]
            data[at:at + 4] = struct.pack(">I", int(rng.choice([0, 1, 0x7FFFFFFF, 0xFFFFFFFF, 100000])))
        else:                                                      # IDAT duplicated / IEND moved
            i = bytes(data).find(b"IDAT")
            if i > 4:
                n = struct.unpack(">I", data[i - 4:i])[0]
                data = data[:i + 8 + n] + data[i - 4:i + 8 + n] + data[i + 8 + n:]
        blob = bytes(data)
        if k % 2 and len(blob) > 8:
            blob = _png_fix_crcs(blob)
        p = str(tmp_path / "m.png")
        open(p, "wb").write(blob)
        try:
            img = engine.load_png_rgba8(p)
            assert img.ndim == 3 and img.shape[2] == 4 and 0 < img.shape[0] <= 16384 and 0 < img.shape[1] <= 16384
            ok += 1
        except engine.ZeldaRenderError:
            bad += 1
    assert bad > N_MUT // 4 and ok + bad == N_MUT
    # OBJ: numbers replaced by junk, indices out of range / negative / huge, lines cut
    obj = (OBJ * 3).split("\\n")
    for k in range(N_MUT):
        lines = list(obj)
        for _ in range(int(rng.integers(1, 5))):
            i = int(rng.integers(0, len(lines)))
            toks = lines[i].split(" ")
            if len(toks) > 1:
                j = int(rng.integers(1, len(toks)))
                toks[j] = str(rng.choice(["", "nan", "1e999", "-0", "999999999999", "-7", "1/", "//", "1/2/3/4", "0/0/0", "4294967297/1/1", "x", "1e-400"]))
                lines[i] = " ".join(toks)
            if rng.random() < 0.2:
                lines[i] = lines[i][:int(rng.integers(0, len(lines[i]) + 1))]
        p = str(tmp_path / "m.obj")
        open(p, "w").write("\\n".join(lines))
        try:
            v, idx = engine.load_obj(p)
            assert len(idx) % 3 == 0 and (len(idx) == 0 or int(idx.max()) < len(v))
        except engine.ZeldaRenderError:
            pass
