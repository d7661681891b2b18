"""The loaders against the ENGINE'S OWN files (ZeldaEngine.cpp:6882-6948 consume them): Content/Models/*.obj, Content/Textures/*.png.

`tests/golden/content_digests.json` was made in the build container by `tests/golden/make_golden_content.py` with readers that share no
code with the product (PIL; a 30-line restatement of LoadMeshAsset): per PNG the size and the sha256 of the RGBA8 bytes, per OBJ the
vertex / index counts and the sha256 of the deduplicated XkVertex and index arrays.  Here the library's `zr_load_png_rgba8` / `zr_load_obj`
(host code of the C-ABI, no GPU needed) and the Python twins of `zeldaengine_amd.assets` must reproduce every digest.  The content tree
is the reference's and does not travel: where it is absent (the GPU box) the file-reading tests skip; the known answers that SURVEY App. C
derives from two of those files are asserted from the digests alone.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from zeldaengine_amd import assets, engine

HERE = os.path.dirname(os.path.abspath(__file__))
CONTENT = "/root/reference/Engine/ZeldaEngine/Content"
DIGESTS = json.load(open(os.path.join(HERE, "golden", "content_digests.json")))
needs_content = pytest.mark.skipif(not os.path.isdir(CONTENT), reason="the reference's content tree is not on this machine")


def test_digest_file_covers_the_shipped_content_and_the_known_answers():
    assert len(DIGESTS["textures"]) == 14 and len(DIGESTS["models"]) == 5
    # sphere.obj 960 triangles, cube.obj 48, stage.obj 576 (indices / 3); App. C: default_normal (127,127,255), default_grey 127
    m = DIGESTS["models"]
    assert (m["sphere.obj"]["indices"], m["cube.obj"]["indices"], m["stage.obj"]["indices"]) == (2880, 144, 1728 + 432)
    assert m["sphere.obj"]["vertices"] == 559 and m["cube.obj"]["vertices"] == 48
    t = DIGESTS["textures"]
    assert t["default_normal.png"]["first_texel"][:3] == [127, 127, 255]
    assert t["default_grey.png"]["first_texel"][:3] == [127, 127, 127]
    assert t["default_black.png"]["first_texel"][:3] == [0, 0, 0] and t["default_white.png"]["first_texel"][:3] == [255, 255, 255]
    for k in ("cubemap_X0.png", "cubemap_X1.png", "cubemap_Y2.png", "cubemap_Y3.png", "cubemap_Z4.png", "cubemap_Z5.png"):
        assert (t[k]["width"], t[k]["height"]) == (1024, 1024)      # maxMips = 11 (ZE:4308,6887)


@needs_content
@pytest.mark.parametrize("name", sorted(DIGESTS["textures"]))
def test_png_decoders_reproduce_the_engines_textures(name):
    want = DIGESTS["textures"][name]
    path = os.path.join(CONTENT, "Textures", name)
    for who, img in (("zr_load_png_rgba8", engine.load_png_rgba8(path)), ("assets.load_image_rgba8", assets.load_image_rgba8(path))):
        assert img.dtype == np.uint8 and img.shape == (want["height"], want["width"], 4), who
        assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == want["rgba8_sha256"], who
        assert img[0, 0].tolist() == want["first_texel"], who


@needs_content
@pytest.mark.parametrize("name", sorted(DIGESTS["models"]))
def test_obj_loaders_reproduce_the_engines_models(name):
    want = DIGESTS["models"][name]
    path = os.path.join(CONTENT, "Models", name)
    for who, (v, idx) in (("zr_load_obj", engine.load_obj(path)), ("assets.load_obj", assets.load_obj(path))):
        assert v.dtype.itemsize == 44 and idx.dtype == np.uint32, who
        assert (len(v), len(idx)) == (want["vertices"], want["indices"]), who
        assert hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() == want["xkvertex_sha256"], who
        assert hashlib.sha256(np.ascontiguousarray(idx).tobytes()).hexdigest() == want["index_sha256"], who
