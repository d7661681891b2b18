"""Synthetic scenes for tests and bench.py: meshes, seeded instance scatter, the sample world.

Nothing here reads the reference tree.  The shipped OBJ models are not copied; `uv_sphere`
regenerates the topology of Content/Models/sphere.obj (32 segments x 16 rings, radius 0.5,
482 points / 960 triangles, Z-up) through the same ingest rules as LoadMeshAsset
(ZE:6899-6948: vertices deduplicated on the full record, normals taken per *position* index,
uv.v flipped, colour 1,1,1).
"""
import math
import random

import numpy as np

from . import abi


# ---------------------------------------------------------------- meshes

def _ingest(points, normals, uvs, faces):
    """LoadMeshAsset semantics: faces = list of ((p, t), (p, t), (p, t)) index pairs."""
    verts, index, lookup = [], [], {}
    for tri in faces:
        for p, t in tri:
            key = (p, t)
            if key not in lookup:
                lookup[key] = len(verts)
                verts.append((points[p], normals[p], (1.0, 1.0, 1.0), (uvs[t][0], 1.0 - uvs[t][1])))
            index.append(lookup[key])
    v = np.zeros(len(verts), dtype=abi.XkVertex)
    for i, (p, n, c, t) in enumerate(verts):
        v[i]["Position"], v[i]["Normal"], v[i]["Color"], v[i]["TexCoord"] = p, n, c, t
    return v, np.asarray(index, dtype=np.uint32)


def uv_sphere(segments=32, rings=16, radius=0.5):
    """Z-up UV sphere with pole fans; (32,16) -> 482 points, 960 triangles, 559 deduplicated vertices."""
    pts, nrm, uvs = [], [], []
    pts.append((0.0, 0.0, radius)); nrm.append((0.0, 0.0, 1.0))
    for i in range(1, rings):
        th = math.pi * i / rings
        for j in range(segments):
            ph = 2.0 * math.pi * j / segments
            n = (math.sin(th) * math.cos(ph), math.sin(th) * math.sin(ph), math.cos(th))
            pts.append((radius * n[0], radius * n[1], radius * n[2])); nrm.append(n)
    pts.append((0.0, 0.0, -radius)); nrm.append((0.0, 0.0, -1.0))
    south = len(pts) - 1

    def pid(i, j):
        return 1 + (i - 1) * segments + (j % segments)

    # uv table: ring rows (segments + 1 columns) and per-segment pole entries
    def tid_ring(i, j):
        return (i - 1) * (segments + 1) + j
    for i in range(1, rings):
        for j in range(segments + 1):
            uvs.append((j / segments, 1.0 - i / rings))
    top0 = len(uvs)
    for j in range(segments):
        uvs.append(((j + 0.5) / segments, 1.0))
    bot0 = len(uvs)
    for j in range(segments):
        uvs.append(((j + 0.5) / segments, 0.0))

    faces = []
    for j in range(segments):       # north fan: pole, ring1[j], ring1[j+1]  (e_theta x e_phi = outward)
        faces.append(((0, top0 + j), (pid(1, j), tid_ring(1, j)), (pid(1, j + 1), tid_ring(1, j + 1))))
    for i in range(1, rings - 1):
        for j in range(segments):
            a = (pid(i, j), tid_ring(i, j)); b = (pid(i + 1, j), tid_ring(i + 1, j))
            c = (pid(i + 1, j + 1), tid_ring(i + 1, j + 1)); d = (pid(i, j + 1), tid_ring(i, j + 1))
            faces.append((a, b, c)); faces.append((a, c, d))
    for j in range(segments):       # south fan
        faces.append(((pid(rings - 1, j), tid_ring(rings - 1, j)), (south, bot0 + j),
                      (pid(rings - 1, j + 1), tid_ring(rings - 1, j + 1))))
    return _ingest(pts, nrm, uvs, faces)


def box(half=(0.5, 0.5, 0.5), center=(0.0, 0.0, 0.0)):
    """Axis-aligned box, 12 triangles, per-face uvs; normals per position (ingest quirk) = normalised corner direction."""
    hx, hy, hz = half
    cx, cy, cz = center
    pts = [(cx + sx * hx, cy + sy * hy, cz + sz * hz) for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)]
    nrm = []
    for sz in (-1, 1):
        for sy in (-1, 1):
            for sx in (-1, 1):
                l = math.sqrt(3.0)
                nrm.append((sx / l, sy / l, sz / l))
    uvs = [(0.0, 0.0), (1.0, 0.0), (1.0, 1.0), (0.0, 1.0)]
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]  # outward CCW
    faces = []
    for q in quads:
        faces.append(((q[0], 0), (q[1], 1), (q[2], 2)))
        faces.append(((q[0], 0), (q[2], 2), (q[3], 3)))
    return _ingest(pts, nrm, uvs, faces)


def grid_plane(size=10.0, n=8, z=0.0):
    """Z-up square ground plane, n x n cells (2 n^2 triangles), uv tiling 0..1."""
    pts, nrm, uvs = [], [], []
    for i in range(n + 1):
        for j in range(n + 1):
            pts.append((-size / 2 + size * j / n, -size / 2 + size * i / n, z)); nrm.append((0.0, 0.0, 1.0))
            uvs.append((j / n, i / n))
    faces = []
    for i in range(n):
        for j in range(n):
            a = i * (n + 1) + j; b = a + 1; c = a + n + 2; d = a + n + 1
            faces.append(((a, a), (b, b), (c, c))); faces.append(((a, a), (c, c), (d, d)))
    return _ingest(pts, nrm, uvs, faces)


# ---------------------------------------------------------------- instance scatter

class PCG32:
    """PCG-XSH-RR 64/32 (O'Neill).  The engine seeds mt19937 from libc rand() per draw (ZE:592-603), which is
    platform specific, so scenes carry explicit instance arrays generated from this PRNG instead."""
    MULT = 6364136223846793005
    MASK = (1 << 64) - 1

    def __init__(self, seed=1234, seq=54):
        self.state, self.inc = 0, ((seq << 1) | 1) & self.MASK
        self.next_u32()
        self.state = (self.state + seed) & self.MASK
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & self.MASK
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def fill_u32(self, n):
        out = np.empty(n, dtype=np.uint32)
        for i in range(n):
            out[i] = self.next_u32()
        return out


def _pcg32_array(seed, n):
    """n raw u32 draws, vectorised: same stream as PCG32(seed).next_u32() (LCG jump-ahead in wrapping uint64)."""
    g = PCG32(seed)
    if n == 0:
        return np.empty(0, dtype=np.uint32)
    with np.errstate(over="ignore"):
        a = np.full(n, PCG32.MULT, dtype=np.uint64)
        a[0] = 1
        apow = np.cumprod(a, dtype=np.uint64)                       # a^k, k = 0..n-1 (mod 2^64)
        gsum = np.concatenate(([0], np.cumsum(apow, dtype=np.uint64)[:-1])).astype(np.uint64)  # sum_{i<k} a^i
        old = apow * np.uint64(g.state) + np.uint64(g.inc) * gsum   # state before the k-th draw
        xs = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        return (xs >> rot) | (xs << ((np.uint32(32) - rot) & np.uint32(31)))


def generate_instances(count, min_radius, max_radius, min_pscale, max_pscale, seed=1234):
    """XkObjectDesc::GenerateInstance (ZE:573-589) with PCG32(seed) in place of mt19937(std::rand()).

    Per instance, in draw order: angle ~ U(0,360) deg, distance ~ U(MinRadius,MaxRadius),
    pos = (sin(a) d, cos(a) d, 0); rot = (0, pi * U(0,180), 0) radians; pscale ~ U(Min,Max); tex ~ U{0..255}.
    """
    raw = _pcg32_array(seed, 5 * count).reshape(count, 5)
    u = (raw >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    f32 = np.float32
    ang = f32(0.0) + f32(360.0) * u[:, 0]
    dist = f32(min_radius) + (f32(max_radius) - f32(min_radius)) * u[:, 1]
    rad = ang * f32(0.01745329251994329576923690768489)
    inst = np.zeros(count, dtype=abi.XkInstanceData)
    inst["InstancePosition"][:, 0] = np.sin(rad).astype(np.float32) * dist
    inst["InstancePosition"][:, 1] = np.cos(rad).astype(np.float32) * dist
    inst["InstanceRotation"][:, 1] = f32(math.pi) * (f32(180.0) * u[:, 2])
    inst["InstancePScale"] = f32(min_pscale) + (f32(max_pscale) - f32(min_pscale)) * u[:, 3]
    inst["InstanceTexIndex"] = (raw[:, 4] >> np.uint32(24)).astype(np.uint8)
    return inst


# ---------------------------------------------------------------- sample world (the livelink payload)

def sample_world():
    """Rebuilds the `xkWorld` dict of Engine/ZeldaPython/ZeldaUntitled.py:28-159 field by field.

    json.dumps(sample_world()) must equal tests/golden/xkworld_untitled.json byte for byte.
    """
    def light():
        return {"Position": [20.0, 0.0, 20.0], "Type": 0, "Color": [1.0, 1.0, 1.0], "Intensity": 3.0,
                "Direction": [0.7, 0.7, 0.7], "Radius": 0.0, "ExtraData": [0.0, 0.0, 0.0, 0.0]}

    def obj(name, count, **kw):
        o = {"RenderFlags": 0, "ProfabName": name, "InstanceCount": count, "MinRadius": 0.0, "MaxRadius": 0.0,
             "MinRotYaw": 0.0, "MaxRotYaw": 0.0, "MinRotRoll": 0.0, "MaxRotRoll": 0.0, "MinRotPitch": 0.0,
             "MaxRotPitch": 0.0, "MinPScale": 0.0, "MaxPScale": 0.0}
        o.update(kw)
        return o

    world = {
        "MainCamera": {"Position": [5.0, 5.0, 5.0], "Lookat": [0.0, 0.0, 0.5], "Speed": 2.5, "FOV": 45.0,
                       "zNear": 0.1, "zFar": 45.0},
        "Skydome": {"EnableSkydome": True, "OverrideSkydome": True, "SkydomeFileName": "grassland_night.png",
                    "OverrideCubemap": True,
                    "CubemapFileNames": ["grassland_night_X0.png", "grassland_night_X1.png", "grassland_night_Y2.png",
                                         "grassland_night_Y3.png", "grassland_night_Z4.png", "grassland_night_Z5.png"]},
        "Background": {"EnableBackground": True, "OverrideBackground": True, "BackgroundFileName": "background.png"},
        "DirectionalLights": [], "PointLights": [], "SpotLights": [], "Objects": [],
    }
    world["Objects"] += [
        obj("terrain", 1), obj("rock_01", 1),
        obj("rock_02", 64, MinRadius=1.0, MaxRadius=5.0, MinPScale=0.2, MaxPScale=0.5),
        obj("grass_01", 10000, MinRadius=2.0, MaxRadius=8.0, MinPScale=0.1, MaxPScale=0.5),
        obj("grass_02", 10000, MinRadius=1.0, MaxRadius=9.0, MinPScale=0.1, MaxPScale=0.5),
    ]
    moon = light()
    moon.update({"Position": [20.0, 0.0, 20.0], "Type": 0, "Color": [0.0, 0.1, 0.6], "Intensity": 15.0,
                 "Radius": 0.0, "ExtraData": [0.0, 0.0, 0.0, 0.0]})
    moon["Direction"] = moon["Position"]
    world["DirectionalLights"].append(moon)
    world["PointLights"] += sample_point_lights(16)
    return world


def sample_point_lights(n):
    """The point-light pattern of ZeldaUntitled.py:140-159 for any n (config 5 uses n = 256)."""
    out = []
    for i in range(n):
        rng = random.Random(i)       # == random.seed(i) then module-level draws
        radians = rng.uniform(0.0, 360.0)
        distance = rng.uniform(0.1, 0.6)
        x = math.sin(math.radians(radians)) * distance
        y = math.cos(math.radians(radians)) * distance
        r = rng.uniform(0.5, 0.75)
        g = rng.uniform(0.25, 0.5)
        out.append({"Position": [x, y, 1.0], "Type": 1, "Color": [r, g, 0.0], "Intensity": 10.0,
                    "Direction": [0.0, 0.0, 1.0], "Radius": 1.5, "ExtraData": [0.0, 0.0, 0.0, 0.0]})
    return out


def lights_from_world(world):
    """XkLight(const XkLightDesc&) for each light array (ZE:781-787)."""
    def conv(arr):
        out = np.zeros(len(arr), dtype=abi.XkLight)
        for i, l in enumerate(arr):
            out[i] = abi.make_light(l["Position"], l["Type"], l["Color"], l["Intensity"], l["Direction"], l["Radius"],
                                    l["ExtraData"])
        return out
    return conv(world["DirectionalLights"]), conv(world["PointLights"]), conv(world["SpotLights"])


# ---------------------------------------------------------------- bench / parity configurations

def synthetic_cubemap(dim=64):
    """Deterministic sky gradient, 6 RGBA8 faces in +X,-X,+Y,-Y,+Z,-Z order (no texture files ship to the GPU box)."""
    faces = []
    t = (np.arange(dim, dtype=np.float32) + 0.5) / dim
    u, v = np.meshgrid(t, t)
    for f in range(6):
        img = np.zeros((dim, dim, 4), dtype=np.uint8)
        img[..., 0] = (40 + 150 * u * (0.5 + 0.08 * f)).astype(np.uint8)
        img[..., 1] = (60 + 120 * v).astype(np.uint8)
        img[..., 2] = (90 + 25 * f + 60 * (1.0 - v)).astype(np.uint8)
        img[..., 3] = 255
        faces.append(img)
    return faces


def sky_dome(radius=20.48, segments=32, rings=16):
    """Inward-facing sphere standing in for Content/Models/skydome.obj (radius 20.48 about the origin, 3968 triangles there)."""
    v, idx = uv_sphere(segments, rings, radius)
    return v, idx.reshape(-1, 3)[:, ::-1].reshape(-1).copy()


def synthetic_sky_image(width=256, height=128):
    """Deterministic gradient + bands, RGBA8 (sampled as sRGB)."""
    x = np.arange(width, dtype=np.float32)[None, :] / width
    y = np.arange(height, dtype=np.float32)[:, None] / height
    img = np.zeros((height, width, 4), dtype=np.uint8)
    img[..., 0] = (40 + 120 * x + 30 * np.sin(20 * y)).clip(0, 255).astype(np.uint8)
    img[..., 1] = (70 + 150 * y).astype(np.uint8) + np.zeros_like(x, dtype=np.uint8)
    img[..., 2] = (200 - 60 * x * y).astype(np.uint8)
    img[..., 3] = 255
    return img


def config2():
    """Single 960-triangle sphere, 512x512, 0 directional + 1 point light."""
    v, idx = uv_sphere()
    point = np.zeros(1, dtype=abi.XkLight)
    point[0] = abi.make_light((0, 0, 0), 1, (1, 1, 1), 10.0, (0, 0, 1), 10.0)
    return {"width": 512, "height": 512, "camera": abi.make_camera(),
            "objects": [{"mesh": (v, idx), "instances": None}],
            "dir": np.zeros(0, dtype=abi.XkLight), "point": point, "spot": np.zeros(0, dtype=abi.XkLight),
            "cubemap": synthetic_cubemap(64)}


def synthetic_material(dim=512):
    """Seven NON-constant RGBA8 images (bc, m, r, n, ao, ev, ms) for the sampled-material path (trilinear, anisotropic, sRGB slot 0):
    deterministic bands / checkers / ripples, so every slot really goes through the filter."""
    t = (np.arange(dim, dtype=np.float32) + 0.5) / dim
    u, v = np.meshgrid(t, t)
    chk = ((np.floor(u * 16) + np.floor(v * 16)) % 2).astype(np.float32)
    rip = 0.5 + 0.5 * np.sin(40.0 * u) * np.cos(36.0 * v)

    def img(r, g, b, a=None):
        out = np.zeros((dim, dim, 4), dtype=np.uint8)
        for k, ch in enumerate((r, g, b, a if a is not None else np.ones_like(u))):
            out[..., k] = np.clip(np.asarray(ch, dtype=np.float32) * 255.0 + 0.5, 0, 255).astype(np.uint8)
        return out
    one = np.ones_like(u)
    return [img(0.25 + 0.6 * chk, 0.3 + 0.5 * u, 0.2 + 0.6 * v),                       # base colour (sRGB)
            img(0.1 + 0.8 * rip, 0 * one, 0 * one),                                      # metallic
            img(0.25 + 0.6 * (1.0 - chk) * v, 0 * one, 0 * one),                         # roughness
            img(0.5 + 0.08 * np.sin(60.0 * u), 0.5 + 0.08 * np.cos(50.0 * v), one),      # normal map
            img(0.6 + 0.4 * rip, one, one),                                              # ambient occlusion
            img(0.05 * chk, 0.02 * one, 0.1 * (1.0 - chk)),                              # emissive
            img(one, 0.5 + 0.5 * chk, v)]                                                # mask: r = 1 (lit everywhere), g / b vary so that
    #                                                                                      the slot is an image, not a constant


def config3(n_instances=10000, width=1920, height=1080, max_radius=8.0, n_point=16, seed=1234, cube_dim=64, textured=False):
    """n instanced 960-tri spheres (~10 meshlets each), 1 directional + 16 point lights, 1024^2 PCF shadow.

    cube_dim: edge of the synthetic cubemap; the engine always holds 6 x 1024^2 faces / 11 mips (ZE:5908-6150, ZE:4308), which is
    what bench.py passes; the parity tests keep 64 (7 mips) so that the oracle's mip chain builds in a moment."""
    v, idx = uv_sphere()
    world = sample_world()
    d, _, s = lights_from_world(world)
    world["PointLights"] = sample_point_lights(n_point)
    _, p, _ = lights_from_world(world)
    inst = generate_instances(n_instances, 2.0, max_radius, 0.1, 0.5, seed)
    obj = {"mesh": (v, idx), "instances": inst}
    cfg = {"width": width, "height": height, "camera": abi.make_camera(), "objects": [obj],
           "dir": d, "point": p, "spot": s, "cubemap": synthetic_cubemap(cube_dim)}
    if textured:
        mat, keep = abi.make_material(synthetic_material(512))
        obj["material"] = mat
        cfg["_keepalive"] = keep
    return cfg


def config4(n_instances=1000000, n_point=16, cube_dim=64):
    """1M instances / ~5M meshlet-instances at 3840x2160 (8-GPU configuration); config 5 uses n_point = 256."""
    return config3(n_instances, 3840, 2160, 60.0, n_point, cube_dim=cube_dim)
