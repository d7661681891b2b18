"""An independent float64 restatement of the texture sampling rules the oracle follows (test infrastructure; numpy only).

Written from the Vulkan specification's wording, not from oracle/zo_oracle.c, and organised differently (whole-array numpy for the
mip chains, scalar float64 for one sample), so that an error in the oracle's float32 code is not repeated here:

* "Image Sample Operations" chapter of the specification:
  - sRGB decode / encode                         ("sRGB EOTF" of the Khronos Data Format specification)
  - unnormalised coordinates u = s * width, linear filtering takes texels i0 = floor(u - 1/2), i1 = i0 + 1 with weight
    frac(u - 1/2); REPEAT wraps i mod size; CLAMP_TO_EDGE clamps i to [0, size - 1]
  - mip level selection: lambda from the footprint, d_lo = floor(lambda), d_hi = d_lo + 1, weight frac(lambda)
  - "Texel Anisotropic Filtering": N = min(ceil(Pmax / Pmin), maxAniso), lambda = log2(Pmax / N), N samples spread along the
    major axis at (i / (N + 1) - 1/2), averaged
  - cube map face selection table (major axis, sc, tc, ma; ties: z over y over x), s = 1/2 (sc / |ma|) + 1/2
* vkCmdBlitImage with VK_FILTER_LINEAR (RHIGenerateMipmaps, ZE:6348-6433): destination texel centre (x + 1/2) maps to
  u = (x + 1/2) * srcW / dstW in the source level, which is then filtered linearly with CLAMP_TO_EDGE.

Choices the specification leaves open and the oracle fixes (oracle/zo_oracle.c header): maxAnisotropy 16; cube faces are filtered
with CLAMP_TO_EDGE inside the selected face (no seam filtering); cubemap levels halve by a 2x2 box, which is what the LINEAR blit of
an even-sized level is.
"""
import math

import numpy as np

F = np.float64
MAX_ANISO = 16


def srgb_decode(c8):
    c = np.asarray(c8, F) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def srgb_encode_real(l):
    """linear [0, 1] -> the real-valued 8-bit code before rounding"""
    l = np.clip(np.asarray(l, F), 0.0, 1.0)
    return 255.0 * np.where(l <= 0.0031308, 12.92 * l, 1.055 * l ** (1.0 / 2.4) - 0.055)


def decode(img8, srgb):
    """(h, w, 4) uint8 -> float64 RGBA as the sampler sees it (alpha is never sRGB)"""
    out = np.asarray(img8, F) / 255.0
    if srgb:
        out[..., :3] = srgb_decode(np.asarray(img8)[..., :3])
    return out


def blit_half_real(level8, srgb):
    """One vkCmdBlitImage(LINEAR) step to the next mip: returns the REAL-valued 8-bit codes (h', w', 4) before rounding."""
    src = decode(level8, srgb)
    sh, sw = src.shape[:2]
    dh, dw = max(1, sh // 2), max(1, sw // 2)

    def taps(dn, sn):
        u = (np.arange(dn, dtype=F) + 0.5) * (sn / dn) - 0.5
        i0 = np.floor(u)
        w = u - i0
        return np.clip(i0, 0, sn - 1).astype(int), np.clip(i0 + 1, 0, sn - 1).astype(int), w
    x0, x1, wx = taps(dw, sw)
    y0, y1, wy = taps(dh, sh)
    rows0 = src[y0][:, x0] * (1 - wx)[None, :, None] + src[y0][:, x1] * wx[None, :, None]
    rows1 = src[y1][:, x0] * (1 - wx)[None, :, None] + src[y1][:, x1] * wx[None, :, None]
    lin = rows0 * (1 - wy)[:, None, None] + rows1 * wy[:, None, None]
    out = lin * 255.0
    if srgb:
        out[..., :3] = srgb_encode_real(lin[..., :3])
    return out


def wrap_repeat(i, n):
    return int(i) % n


def bilinear_repeat(level, u, v):
    """level: decoded float64 (h, w, 4); normalised (u, v); REPEAT addressing"""
    h, w = level.shape[:2]
    x, y = u * w - 0.5, v * h - 0.5
    i0, j0 = math.floor(x), math.floor(y)
    a, b = x - i0, y - j0
    t = lambda i, j: level[wrap_repeat(j, h), wrap_repeat(i, w)]
    return (t(i0, j0) * (1 - a) + t(i0 + 1, j0) * a) * (1 - b) + (t(i0, j0 + 1) * (1 - a) + t(i0 + 1, j0 + 1) * a) * b


def trilinear(levels, lam, u, v):
    d = int(math.floor(lam))
    hi = min(d + 1, len(levels) - 1)
    f = lam - d
    return bilinear_repeat(levels[d], u, v) * (1 - f) + bilinear_repeat(levels[hi], u, v) * f


def aniso_parameters(w, h, duv, n_levels):
    """-> N, lambda, major axis derivative (du, dv), margin: the distance of Pmax / Pmin from the nearest integer below 16 (a sample
    whose ratio sits on such a boundary may legitimately take N or N + 1 taps in float32)"""
    dudx, dvdx, dudy, dvdy = [float(x) for x in duv]
    px, py = math.hypot(dudx * w, dvdx * h), math.hypot(dudy * w, dvdy * h)
    pmax, pmin = max(px, py), min(px, py)
    if pmax == 0.0:
        return 1, 0.0, (0.0, 0.0), 1.0
    ratio = pmax / pmin if pmin > 0 else math.inf
    n = MAX_ANISO if ratio >= MAX_ANISO else max(1, math.ceil(ratio))
    margin = 1.0 if (ratio >= MAX_ANISO + 0.5 or ratio == 1.0) else abs(ratio - round(ratio))
    lam = math.log2(pmax / n)
    lam = min(max(lam, 0.0), n_levels - 1.0)
    major = (dudx, dvdx) if px >= py else (dudy, dvdy)
    return n, lam, major, margin


def sample_2d(levels8, srgb, uv, duv):
    """texture(sampler2D, uv) with explicit derivatives; levels8: the mip chain as uint8 arrays"""
    levels = [decode(l, srgb) for l in levels8]
    h, w = levels[0].shape[:2]
    n, lam, (du, dv), margin = aniso_parameters(w, h, duv, len(levels))
    acc = np.zeros(4, F)
    for i in range(1, n + 1):
        off = i / (n + 1) - 0.5
        acc += trilinear(levels, lam, uv[0] + du * off, uv[1] + dv * off)
    return acc / n, n, lam, margin


# Cube Map Face Selection (the specification's table): face index, sc, tc, ma
def cube_face(r):
    rx, ry, rz = [float(x) for x in r]
    ax, ay, az = abs(rx), abs(ry), abs(rz)
    if az >= ax and az >= ay:
        return (4, rx, -ry, az) if rz >= 0 else (5, -rx, -ry, az)
    if ay >= ax:
        return (2, rx, rz, ay) if ry >= 0 else (3, rx, -rz, ay)
    return (0, -rz, -ry, ax) if rx >= 0 else (1, rz, -ry, ax)


def bilinear_clamp(face, s, t):
    d = face.shape[0]
    x, y = s * d - 0.5, t * d - 0.5
    i0, j0 = math.floor(x), math.floor(y)
    a, b = x - i0, y - j0
    cl = lambda i: min(max(i, 0), d - 1)
    g = lambda i, j: face[cl(j), cl(i)]
    return (g(i0, j0) * (1 - a) + g(i0 + 1, j0) * a) * (1 - b) + (g(i0, j0 + 1) * (1 - a) + g(i0 + 1, j0 + 1) * a) * b


def sample_cube(levels8, r, lod):
    """textureLod(samplerCube, r, lod).rgb; levels8[l]: (6, d, d, 4) uint8, sRGB"""
    f, sc, tc, ma = cube_face(r)
    s, t = 0.5 * sc / ma + 0.5, 0.5 * tc / ma + 0.5
    lam = min(max(float(lod), 0.0), len(levels8) - 1.0)
    d = int(math.floor(lam))
    hi = min(d + 1, len(levels8) - 1)
    w = lam - d
    lo_c = bilinear_clamp(srgb_decode(levels8[d][f][..., :3]), s, t)
    hi_c = bilinear_clamp(srgb_decode(levels8[hi][f][..., :3]), s, t)
    return lo_c * (1 - w) + hi_c * w


def cube_half_real(level8):
    """next cubemap level as real-valued codes: 2x2 box in linear light (colour), plain mean (alpha)"""
    lin = srgb_decode(level8[..., :3])
    d = level8.shape[1]
    nd = max(1, d // 2)
    ix = np.minimum(np.arange(nd) * 2 + 1, d - 1)
    i0 = np.arange(nd) * 2
    box = lambda a: (a[:, i0][:, :, i0] + a[:, i0][:, :, ix] + a[:, ix][:, :, i0] + a[:, ix][:, :, ix]) / 4.0
    return srgb_encode_real(box(lin)), box(np.asarray(level8[..., 3], F))
