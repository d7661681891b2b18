"""Regenerates tests/golden/content_digests.json from the engine's own shipped content (build container only).

The one kind of fixture the reference DOES hold is the content its loaders consume (`Content/Models/*.obj`, `Content/Textures/*.png`,
read by LoadMeshAsset / LoadTextureAsset, ZeldaEngine.cpp:6882-6948).  This script digests every such file with readers that share no
code with the product or its Python twins:

* PNG: PIL decodes, `convert("RGBA")` is what stb_image's STBI_rgb_alpha hands the engine (ZE:6885); digest = size + sha256 of the bytes
  (+ the first texel, which SURVEY App. C's known answers for default_normal / default_grey rest on);
* OBJ: the ~30-line reader below restates LoadMeshAsset from its text (ZE:6899-6948: polygons fanned as tinyobjloader does, the NORMAL taken
  by the corner's POSITION index :6927-6931, v flipped to 1 - v :6937, colour (1,1,1) :6933, vertices deduplicated on the whole record
  :520-535); digest = vertex / index counts + sha256 of the XkVertex array (44 B records) and of the uint32 index array.

Only the digests are committed (data: sizes, counts, hashes) - no byte of the content itself and no line of the reference's source.
`tests/test_content_pins.py` holds `zr_load_png_rgba8` / `zr_load_obj` and `assets.load_obj` / `load_image_rgba8` to them wherever the
content tree is present (it is not on the GPU box: CPU suite, skipped there).
"""
import hashlib
import json
import os
import struct
import sys

import numpy as np
from PIL import Image

CONTENT = "/root/reference/Engine/ZeldaEngine/Content"
HERE = os.path.dirname(os.path.abspath(__file__))


def digest_png(path):
    a = np.ascontiguousarray(np.array(Image.open(path).convert("RGBA"), dtype=np.uint8))
    return {"width": int(a.shape[1]), "height": int(a.shape[0]), "rgba8_sha256": hashlib.sha256(a.tobytes()).hexdigest(),
            "first_texel": [int(x) for x in a[0, 0]]}


def digest_obj(path):
    P, N, T, out, seen, index = [], [], [], bytearray(), {}, []
    f32 = lambda x: struct.unpack("<f", struct.pack("<f", float(x)))[0]      # noqa: E731  (the loader's real_t is float)
    for line in open(path, errors="replace"):
        w = line.split()
        if not w:
            continue
        if w[0] == "v":
            P.append([f32(x) for x in w[1:4]])
        elif w[0] == "vn":
            N.append([f32(x) for x in w[1:4]])
        elif w[0] == "vt":
            T.append([f32(w[1]), f32(w[2]) if len(w) > 2 else 0.0])
        elif w[0] == "f":
            c = []
            for tok in w[1:]:
                s = (tok.split("/") + ["", ""])[:3]
                vi = int(s[0]); vi = vi - 1 if vi > 0 else len(P) + vi
                ti = int(s[1]) if s[1] else 0; ti = ti - 1 if ti > 0 else len(T) + ti
                c.append((vi, ti))
            for k in range(1, len(c) - 1):
                for vi, ti in (c[0], c[k], c[k + 1]):
                    one_minus_v = struct.unpack("<f", struct.pack("<f", np.float32(1.0) - np.float32(T[ti][1])))[0]
                    rec = struct.pack("<11f", *P[vi], *N[vi], 1.0, 1.0, 1.0, T[ti][0], one_minus_v)
                    if rec not in seen:
                        seen[rec] = len(seen)
                        out += rec
                    index.append(seen[rec])
    idx = struct.pack("<%dI" % len(index), *index)
    return {"vertices": len(seen), "indices": len(index), "xkvertex_sha256": hashlib.sha256(bytes(out)).hexdigest(),
            "index_sha256": hashlib.sha256(idx).hexdigest()}


def main():
    if not os.path.isdir(CONTENT):
        sys.exit("no content tree at %s (build container only)" % CONTENT)
    d = {"source": "Engine/ZeldaEngine/Content (reference tree, read-only)", "textures": {}, "models": {}}
    for name in sorted(os.listdir(os.path.join(CONTENT, "Textures"))):
        if name.endswith(".png"):
            d["textures"][name] = digest_png(os.path.join(CONTENT, "Textures", name))
    for name in sorted(os.listdir(os.path.join(CONTENT, "Models"))):
        if name.endswith(".obj"):
            d["models"][name] = digest_obj(os.path.join(CONTENT, "Models", name))
    with open(os.path.join(HERE, "content_digests.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
        f.write("\n")
    print("%d textures, %d models" % (len(d["textures"]), len(d["models"])))
    for k, v in d["models"].items():
        print(k, v["vertices"], v["indices"])


if __name__ == "__main__":
    main()
