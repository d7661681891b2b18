"""Files of the engine's asset contract: OBJ ingest, `.meshlet` read/write, Profab directories, World.json.

Host-side helpers (load time, not the hot path) so that an existing ZeldaEngine content tree drops in:

* `load_obj`        LoadMeshAsset (ZE:6899-6948) with its quirks: vertices deduplicated on the whole record, normals indexed by
                    the POSITION index (ZE:6927-6931), uv.v flipped to 1 - v, colour (1, 1, 1).
* `load_obj_for_meshlet_tool`  the ZeldaMeshlet tool's ingest (ZM:184-216): normals by normal index, 32-byte vertex.
* `write_meshlet` / `read_meshlet`  the `.meshlet` container (ZM:52-122 == ZE:7089-7127): five sections, each a little-endian
                    size_t count followed by the raw array: Meshlet[64 B], u32 meshletVertices, u8 meshletTriangles,
                    Vertex[32 B], u32 indices.
* `meshlet_tool`    what `ZeldaMeshlet` main does (ZM:235-294): OBJ -> meshlets (64 v / 124 t / cone 0.2) -> file.
* `register_profabs`  CreateRenderObjectsFromProfabs' search (ZE:4922-5000): `Profabs/<name>/models/*.obj` with textures
                    `<model>_{bc,m,r,n,ao,ev,ms}.png`, falling back to the engine defaults.
* `world_load_file` / `world_save_file`  XkWorld::Load() from FilePath / Save() (ZE:1057-1068, 1149-1263).
"""
import os

import numpy as np

from . import abi

TEXTURE_SUFFIXES = ["bc", "m", "r", "n", "ao", "ev", "ms"]      # ZE:4951-4978


def _parse_obj(path):
    pos, nrm, uvs, faces = [], [], [], []
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                pos.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "vn":
                nrm.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "vt":
                uvs.append((float(t[1]), float(t[2]) if len(t) > 2 else 0.0))
            elif t[0] == "f":
                corners = []
                for c in t[1:]:
                    parts = c.split("/")
                    vi = int(parts[0])
                    ti = int(parts[1]) if len(parts) > 1 and parts[1] else 0
                    ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                    # OBJ indices are 1-based; negative = relative to the end (tinyobjloader semantics)
                    vi = vi - 1 if vi > 0 else len(pos) + vi
                    ti = ti - 1 if ti > 0 else (len(uvs) + ti if ti < 0 else -1)
                    ni = ni - 1 if ni > 0 else (len(nrm) + ni if ni < 0 else -1)
                    corners.append((vi, ti, ni))
                for k in range(1, len(corners) - 1):          # tinyobjloader triangulates polygons as a fan
                    faces.append((corners[0], corners[k], corners[k + 1]))
    return pos, nrm, uvs, faces


def load_obj(path):
    """-> (XkVertex[], uint32 indices), exactly as LoadMeshAsset ingests an OBJ."""
    pos, nrm, uvs, faces = _parse_obj(path)
    if not nrm or not uvs:
        raise ValueError("%s: the engine's loader indexes attrib.normals and attrib.texcoords unconditionally" % path)
    verts, index, lookup = [], [], {}
    for tri in faces:
        for vi, ti, _ni in tri:
            n = nrm[vi] if vi < len(nrm) else nrm[-1]          # quirk: normals[3 * vertex_index + k]
            rec = (pos[vi], n, (1.0, 1.0, 1.0), (uvs[ti][0], float(np.float32(1.0) - np.float32(uvs[ti][1]))))      # float arithmetic, as tinyobj's real_t
            key = tuple(np.float32(x).tobytes() for grp in rec for x in grp)
            if key not in lookup:
                lookup[key] = len(verts)
                verts.append(rec)
            index.append(lookup[key])
    v = np.zeros(len(verts), dtype=abi.XkVertex)
    for i, (p, n, c, t) in enumerate(verts):
        v[i]["Position"], v[i]["Normal"], v[i]["Color"], v[i]["TexCoord"] = p, n, c, t
    return v, np.asarray(index, dtype=np.uint32)


def load_obj_for_meshlet_tool(path):
    """-> (XkMeshletFileVertex[], uint32 indices) as buildMeshletsFromAsset ingests it (ZM:184-216)."""
    pos, nrm, uvs, faces = _parse_obj(path)
    verts, index, lookup = [], [], {}
    for tri in faces:
        for vi, ti, ni in tri:
            rec = (pos[vi], nrm[ni], (uvs[ti][0], float(np.float32(1.0) - np.float32(uvs[ti][1]))))
            key = tuple(np.float32(x).tobytes() for grp in rec for x in grp)
            if key not in lookup:
                lookup[key] = len(verts)
                verts.append(rec)
            index.append(lookup[key])
    v = np.zeros(len(verts), dtype=abi.XkMeshletFileVertex)
    for i, (p, n, t) in enumerate(verts):
        v[i]["pos"], v[i]["nrm"], v[i]["uv"] = p, n, t
    return v, np.asarray(index, dtype=np.uint32)


def write_obj(path, verts, idx):
    """Writes an XkVertex mesh so that load_obj(path) returns it again (v flipped back, one v/vt/vn per vertex)."""
    with open(path, "w") as f:
        for v in verts:
            f.write("v %.9g %.9g %.9g\n" % tuple(v["Position"]))
        for v in verts:
            f.write("vt %.9g %.9g\n" % (v["TexCoord"][0], 1.0 - float(np.float32(1.0) - np.float32(1.0) + v["TexCoord"][1])))
        for v in verts:
            f.write("vn %.9g %.9g %.9g\n" % tuple(v["Normal"]))
        for a, b, c in np.asarray(idx).reshape(-1, 3) + 1:
            f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, b, b, b, c, c, c))


# ---------------------------------------------------------------- .meshlet container

def _section(f, arr):
    f.write(np.uint64(len(arr)).tobytes())
    f.write(np.ascontiguousarray(arr).tobytes())


def write_meshlet(path, meshlets, mverts, mtris, vertices, indices):
    with open(path, "wb") as f:
        _section(f, np.asarray(meshlets, dtype=abi.XkMeshlet))
        _section(f, np.asarray(mverts, dtype=np.uint32))
        _section(f, np.asarray(mtris, dtype=np.uint8))
        _section(f, np.asarray(vertices, dtype=abi.XkMeshletFileVertex))
        _section(f, np.asarray(indices, dtype=np.uint32))


def read_meshlet(path):
    """-> dict(meshlets, mverts, mtris, file_vertices, indices, vertices=XkVertex[] as LoadMeshletAsset converts them)."""
    out = {}
    with open(path, "rb") as f:
        for name, dt in (("meshlets", abi.XkMeshlet), ("mverts", np.dtype("<u4")), ("mtris", np.dtype("u1")),
                         ("file_vertices", abi.XkMeshletFileVertex), ("indices", np.dtype("<u4"))):
            n = int(np.frombuffer(f.read(8), dtype="<u8")[0])
            raw = f.read(n * dt.itemsize)
            if len(raw) != n * dt.itemsize:
                raise ValueError("%s: truncated section '%s'" % (path, name))
            out[name] = np.frombuffer(raw, dtype=dt).copy()
    fv = out["file_vertices"]
    v = np.zeros(len(fv), dtype=abi.XkVertex)                # ZE:7141-7166
    v["Position"], v["Normal"], v["TexCoord"] = fv["pos"], fv["nrm"], fv["uv"]
    v["Color"] = 1.0
    out["vertices"] = v
    return out


def meshlet_tool(obj_path, out_path, max_vertices=64, max_triangles=124, cone_weight=0.2):
    """ZeldaMeshlet's job: OBJ -> .meshlet.  Returns the number of meshlets."""
    from . import engine
    fv, idx = load_obj_for_meshlet_tool(obj_path)
    xv = np.zeros(len(fv), dtype=abi.XkVertex)
    xv["Position"], xv["Normal"], xv["TexCoord"] = fv["pos"], fv["nrm"], fv["uv"]
    ml, mv, mt, _order = engine.build_meshlets(xv, idx, max_vertices, max_triangles, cone_weight)
    write_meshlet(out_path, ml, mv, mt, fv, idx)
    return len(ml)


# ---------------------------------------------------------------- Profabs

def load_image_rgba8(path):
    from PIL import Image                                      # stb_image with STBI_rgb_alpha in the engine (ZE:6885)
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGBA"), dtype=np.uint8))


def find_profabs(root):
    """-> {profab name: [(obj path, [7 texture paths or None])]} following ZE:4922-5000 (None = engine default texture)."""
    out = {}
    if not os.path.isdir(root):
        return out
    for name in sorted(os.listdir(root)):
        models, textures = os.path.join(root, name, "models"), os.path.join(root, name, "textures")
        if not (os.path.isdir(models) and os.path.isdir(textures)):
            continue
        entries = []
        for fn in sorted(os.listdir(models)):
            stem, ext = os.path.splitext(fn)
            if ext != ".obj":
                continue
            tex = []
            for suf in TEXTURE_SUFFIXES:
                p = os.path.join(textures, "%s_%s.png" % (stem, suf))
                tex.append(p if os.path.exists(p) else None)
            entries.append((os.path.join(models, fn), tex))
        out[name] = entries
    return out


def register_profabs(renderer, root):
    """Loads every Profab under `root` and registers it with the renderer; returns {name: [mesh ids]}."""
    ids = {}
    for name, entries in find_profabs(root).items():
        ids[name] = []
        for obj_path, tex in entries:
            v, idx = load_obj(obj_path)
            mesh = renderer.mesh_create(v, idx)
            images = [load_image_rgba8(p) if p else None for p in tex]
            mat, keep = abi.make_material(images)
            renderer.profab_register(name, mesh, mat)
            del keep
            ids[name].append(mesh)
    return ids


# ---------------------------------------------------------------- World.json

def world_load_file(renderer, path="Content/World.json"):
    with open(path, "rb") as f:
        renderer.world_load_json(f.read())


def world_save_file(renderer, path="Content/World.json"):
    with open(path, "w") as f:
        f.write(renderer.world_save_json())
