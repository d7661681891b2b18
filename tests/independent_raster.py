"""An exact-arithmetic statement of the rasterisation rules the engine's pipeline state asks Vulkan for, written from the rules themselves
(integers and fractions only: no float, no code, trick or table shared with oracle/zo_oracle.c or the kernels).

What is evaluated, for triangles whose vertices already sit on the 1/256-pixel grid (the sub-pixel snap itself is the oracle's stated
choice and is NOT re-derived here):

* coverage of the sample at (x + 1/2, y + 1/2): strictly inside, or exactly on an edge that is a TOP edge (exactly horizontal, the
  triangle below it: y grows downwards) or a LEFT edge (not horizontal, the triangle to its right) - the rule that makes two triangles
  sharing an edge cover each sample of it exactly once.  The edges are classified geometrically (where is the third vertex?), not by
  the sign tricks a rasteriser uses;
* facing: a = -1/2 sum(x_i y_(i+1) - x_(i+1) y_i) in framebuffer coordinates, front = COUNTER_CLOCKWISE = a > 0 (ZE:5113-5123: cullMode
  BACK in the deferred-scene pass, NONE in the shadow pass); zero-area triangles produce no fragments;
* depth: the plane through the three vertices at the sample, exactly; depth test LESS in draw order (the first of equal depths stays),
  fragments outside 0 <= z <= 1 are clipped (depthClampEnable FALSE); the deferred-scene clear is 1.0, so z = 1 never passes LESS;
* shadow pass: depth bias o = 7.5 * m + 1.25 * r (vkCmdSetDepthBias(1.25, 0, 7.5), ZE:3280-3287) with m = max(|dz/dx|, |dz/dy|) of the
  plane and r = 2^(e - 23), e the exponent of the largest |z| of the triangle's vertices (D32_SFLOAT), test LESS_OR_EQUAL, result clamped
  to [0, 1].
"""
from fractions import Fraction
import math

SUB = 256      # sub-pixel units per pixel


def _edge_kind(a, b, c):
    """Edge a -> b of triangle (a, b, c), y down: 'top' (horizontal, c below), 'left' (not horizontal, c to the right), or None."""
    if a[1] == b[1]:
        return "top" if c[1] > a[1] else None
    # x of the edge's line at c's height
    t = Fraction(c[1] - a[1], b[1] - a[1])
    x_on = a[0] + t * (b[0] - a[0])
    return "left" if c[0] > x_on else None


def _orient(a, b, p):
    return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])


def covers(tri, sx, sy):
    """tri: three (X, Y) integer sub-pixel vertices; (sx, sy) the sample in sub-pixel units."""
    a, b, c = tri
    area2 = _orient(a, b, c)
    if area2 == 0:
        return False
    s = 1 if area2 > 0 else -1
    p = (sx, sy)
    for u, v, w in ((a, b, c), (b, c, a), (c, a, b)):
        e = s * _orient(u, v, p)
        if e < 0:
            return False
        if e == 0 and _edge_kind(u, v, w) is None:
            return False
    return True


def front_facing(tri):
    (x0, y0), (x1, y1), (x2, y2) = tri
    twice_a = -((x0 * y1 - x1 * y0) + (x1 * y2 - x2 * y1) + (x2 * y0 - x0 * y2))
    return twice_a > 0


def plane(tri, z):
    """-> (z at a sub-pixel point as a function, dz/dx and dz/dy per PIXEL), exact."""
    (x0, y0), (x1, y1), (x2, y2) = tri
    z0, z1, z2 = (Fraction(v) for v in z)
    det = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
    gx = ((z1 - z0) * (y2 - y0) - (z2 - z0) * (y1 - y0)) / det
    gy = ((z2 - z0) * (x1 - x0) - (z1 - z0) * (x2 - x0)) / det
    return (lambda sx, sy: z0 + gx * (sx - x0) + gy * (sy - y0)), gx * SUB, gy * SUB


def render(tris, zs, W, H, shadow=False):
    """tris[i]: three (X, Y) ints; zs[i]: three float32-exact depths.  -> (winner id or None per pixel, exact depth per pixel or None,
    ambiguous[y][x] = the two nearest depths are closer than 1e-6: a float evaluation may order them either way)."""
    win = [[None] * W for _ in range(H)]
    dep = [[None] * W for _ in range(H)]
    amb = [[False] * W for _ in range(H)]
    for i, (tri, z) in enumerate(zip(tris, zs)):
        if _orient(*tri) == 0:
            continue
        if not shadow and not front_facing(tri):
            continue
        zat, dzdx, dzdy = plane(tri, z)
        bias = Fraction(0)
        if shadow:
            zmax = max(abs(float(v)) for v in z)
            r = Fraction(0)
            if zmax > 0.0:
                e = math.frexp(zmax)[1] - 1                         # zmax = f * 2^(e+1), 0.5 <= f < 1
                r = Fraction(2) ** (e - 23)
            bias = Fraction(15, 2) * max(abs(dzdx), abs(dzdy)) + Fraction(5, 4) * r
        xs = [v[0] for v in tri]; ys = [v[1] for v in tri]
        for py in range(max(0, min(ys) // SUB - 1), min(H, max(ys) // SUB + 2)):
            for px in range(max(0, min(xs) // SUB - 1), min(W, max(xs) // SUB + 2)):
                sx, sy = px * SUB + SUB // 2, py * SUB + SUB // 2
                if not covers(tri, sx, sy):
                    continue
                zf = zat(sx, sy)
                if zf < 0 or zf > 1:
                    continue
                if shadow:
                    zf = min(max(zf + bias, Fraction(0)), Fraction(1))
                    passes = dep[py][px] is None or zf <= dep[py][px]     # LESS_OR_EQUAL (depth only: who wins does not matter)
                    if dep[py][px] is None:
                        passes = zf <= 1
                else:
                    ref = dep[py][px] if dep[py][px] is not None else Fraction(1)
                    passes = zf < ref
                    same = win[py][px] is not None and tris[win[py][px]] == tri and list(zs[win[py][px]]) == list(z)
                    if abs(zf - ref) < Fraction(1, 10 ** 6) and not same:      # (a repeated triangle ties EXACTLY in any arithmetic)
                        amb[py][px] = True
                if passes:
                    win[py][px] = i; dep[py][px] = zf
    return win, dep, amb


# ---------------------------------------------------------------------------------------------------------------- clipping
def covers_clip_space(tri_clip, X, Y):
    """Exact coverage of the NDC point (X, Y) by a triangle given in CLIP space (x, y, z, w as Fractions), for triangles that may cross
    the eye plane (w <= 0 at some vertices) - the case a rasteriser needs its clipper for.  A point of the triangle is a convex
    combination sum(mu_k v_k); it projects onto (X, Y) iff sum(mu_k (x_k, y_k, w_k)) = t (X, Y, 1) with t = w > 0, i.e. iff the solution
    lambda = mu / t of M lambda = (X, Y, 1), M = columns (x_k, y_k, w_k), has no negative component (2D homogeneous rasterisation; no
    division by any w_k, so nothing special happens at w = 0).  Returns None when not covered or degenerate, else (ndc depth, lambdas):
    the NDC depth of that point is sum(mu_k z_k) / t = sum(lambda_k z_k)."""
    (x0, y0, z0, w0), (x1, y1, z1, w1), (x2, y2, z2, w2) = tri_clip
    det = x0 * (y1 * w2 - y2 * w1) - x1 * (y0 * w2 - y2 * w0) + x2 * (y0 * w1 - y1 * w0)
    if det == 0:
        return None
    # Cramer's rule for M lambda = (X, Y, 1)
    l0 = (X * (y1 * w2 - y2 * w1) - x1 * (Y * w2 - y2) + x2 * (Y * w1 - y1)) / det
    l1 = (x0 * (Y * w2 - y2) - X * (y0 * w2 - y2 * w0) + x2 * (y0 - Y * w0)) / det
    l2 = (x0 * (y1 - Y * w1) - x1 * (y0 - Y * w0) + X * (y0 * w1 - y1 * w0)) / det
    if l0 < 0 or l1 < 0 or l2 < 0:
        return None
    return l0 * z0 + l1 * z1 + l2 * z2, (l0, l1, l2)


def orientation_clip_space(tri_clip):
    """sign of det(columns (x, y, w)): the orientation of the triangle's visible part on the screen (+1: positive signed area in NDC)."""
    (x0, y0, _, w0), (x1, y1, _, w1), (x2, y2, _, w2) = tri_clip
    det = x0 * (y1 * w2 - y2 * w1) - x1 * (y0 * w2 - y2 * w0) + x2 * (y0 * w1 - y1 * w0)
    return (det > 0) - (det < 0)
