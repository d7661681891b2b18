// zr_raster.h — the rasteriser proper: exact integer edge functions with the top-left rule over a tile's LDS keys (raster_sub), the
// near-plane / guard-band clipper (raster_clipped) and the kernel that draws a round's slow triangles (k_tile_slow).  Shared by the
// shadow pass (zr_shadow.hip) and the camera pass (zr_camera.hip).  Raster state: ZE:5094-5201, 3247-3287; rules: DESIGN.md §4.
#pragma once
#include "zr_dev.h"

struct TileCtx {
    int px0, py0;                 // tile origin in pixels
    int W, H;                     // target extent
};

// Cheap per-triangle rejection, identical in effect to the early-outs of raster_sub: degenerate or back-facing
// (GBUFFER only), or no pixel centre of this tile inside the snapped bounding box.
// HIZ (camera pass, round 2): hz[] holds the tile's 4 x 4 pyramid texels (max depth per 8 x 8 pixel block after round 1); a
// triangle whose least vertex depth lies behind every block its clipped box touches cannot win a pixel (fragment depths
// are clamped to the vertex depths).
template <int MODE, bool HIZ = false>
__device__ __forceinline__ bool tri_prefilter(int X0, int Y0, int X1, int Y1, int X2, int Y2, const TileCtx& T,
                                              float zmin = 0.0f, const float* __restrict__ hz = nullptr)
{
    if (MODE == ZR_MODE_GBUFFER) {      // the shadow pass is two-sided: its (rare) degenerate triangles are left to raster_sub
        const long long A = (long long)(X1 - X0) * (Y2 - Y0) - (long long)(X2 - X0) * (Y1 - Y0);
        if (A >= 0) return false;
    }
    const int x0 = max((imin3(X0, X1, X2) - 128 + 255) >> 8, T.px0), x1 = min((imax3(X0, X1, X2) - 128) >> 8, min(T.px0 + SPAN(MODE) - 1, T.W - 1));
    const int y0 = max((imin3(Y0, Y1, Y2) - 128 + 255) >> 8, T.py0), y1 = min((imax3(Y0, Y1, Y2) - 128) >> 8, min(T.py0 + SPAN(MODE) - 1, T.H - 1));
    if (!(x0 <= x1 && y0 <= y1)) return false;
    if (HIZ) {
        const int bx0 = (x0 - T.px0) >> 3, bx1 = (x1 - T.px0) >> 3, by0 = (y0 - T.py0) >> 3, by1 = (y1 - T.py0) >> 3;
        float h = 0.0f;
        for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx) h = __builtin_fmaxf(h, hz[by * (TILE / 8) + bx]);
        if (zmin > h) return false;
    }
    return true;
}

template <int MODE>
__device__ __forceinline__ void shade_key(int x, int y, float fy, const SV& v0, float gx, float gy, float zlo, float zhi,
                                          float bias, uint32_t prim, const TileCtx& T, unsigned long long* __restrict__ keys64, uint32_t* __restrict__ keys32)
{
    const float fx = (float)(x * 256 + 128 - v0.X);
    float z = __builtin_fmaf(gy, fy, __builtin_fmaf(gx, fx, v0.z));
    z = __builtin_fminf(__builtin_fmaxf(z, zlo), zhi);      // fragments stay within their vertices' depths (Hi-Z relies on it)
    z = z + 0.0f;
    const int li = (y - T.py0) * SPAN(MODE) + (x - T.px0);
    if (MODE == ZR_MODE_GBUFFER) {
        if (z >= 0.0f && z < 1.0f)   // depth clip (depthClampEnable FALSE) + LESS against the 1.0 clear
            atomicMin(&keys64[li], (unsigned long long)zr_f2u(z) << 32 | prim);
    } else {
        if (z >= 0.0f && z <= 1.0f) {
            const float zb = __builtin_fminf(__builtin_fmaxf(z + bias, 0.0f), 1.0f);
            atomicMin(&keys32[li], zr_f2u(zb));
        }
    }
}

// Rasterise one snapped triangle into the tile's LDS keys.
//   GBUFFER: key = depth_bits << 32 | prim, ds_min_u64  == depth test LESS in draw order (ties: lower prim wins)
//   SHADOW : key = biased depth bits,        ds_min_u32  == depth test LESS_OR_EQUAL, depth write only
// Coverage is exact integer arithmetic (edge functions of the snapped vertices, top-left rule as a -1 bias); the
// 32-bit loop is taken when every edge value met while walking the clipped bounding box fits, and is bit-identical.
// SMALL: every lane's triangle is small (every edge component below 2^14 sub-pixel units = 64 px) and given in TILE-RELATIVE
// coordinates.  The whole setup then stays in 32 bits: the clipped box lies in the tile, so |P - v| < 2^14 + 2^13 for every corner P of
// the walk and vertex v, an edge value is a difference of two products below 1.5 * 2^28 (|E| < 2^29.6 at the box origin), and the walk
// adds at most 31 steps of 256 |e| < 2^22 per axis (< 2^28): everything stays below 2^31; the area is a difference of two products
// below 2^28.  The tile kernels instantiate ONLY this form in their hot loop - bigger triangles take the clipper's route
// (raster_clipped, a call), which holds the general form - so the loop's register budget carries no 64-bit edge state.  Both forms
// produce the same integers.
#define ZR_SMALL_EDGE (1 << 14)
// every component of every EDGE below ZR_SMALL_EDGE  <=>  the snapped box is narrower than that on both axes (the widest edge
// component along an axis IS the box's extent along it).  All three edges: a triangle whose two edges at vertex 0 are short can still
// have a long third one (found by tests/test_gpu_fuzz.py: such a triangle overflowed the tile-relative 16-bit coordinates).
__device__ __forceinline__ bool tri_is_small(int x0, int y0, int x1, int y1, int x2, int y2)
{
    return imax3(x0, x1, x2) - imin3(x0, x1, x2) < ZR_SMALL_EDGE && imax3(y0, y1, y2) - imin3(y0, y1, y2) < ZR_SMALL_EDGE;
}
// BOXED: the caller hands over the clipped box it already formed with these very expressions (k_tile sorts its records by it).
template <int MODE, bool SMALL, bool BOXED = false>
__device__ __forceinline__ void raster_sub(const SV& v0, const SV& v1, const SV& v2, uint32_t prim, const TileCtx& T,
                                           unsigned long long* __restrict__ keys64, uint32_t* __restrict__ keys32, uint32_t box = 0u)
{
    const int dX1 = v1.X - v0.X, dY1 = v1.Y - v0.Y, dX2 = v2.X - v0.X, dY2 = v2.Y - v0.Y;
    constexpr bool all_small = SMALL;
    long long A;
    if (all_small) A = (long long)(dX1 * dY2 - dX2 * dY1);
    else A = (long long)dX1 * dY2 - (long long)dX2 * dY1;
    if (A == 0) return;
    // Vulkan facing: a = -A/2 in framebuffer coordinates; COUNTER_CLOCKWISE front  <=>  A < 0 (ZE:5113-5123)
    if (MODE == ZR_MODE_GBUFFER && A > 0) return;
    int x0, y0, x1, y1;
    if (BOXED) { x0 = (int)(box & 255u); y0 = (int)((box >> 8) & 255u); x1 = (int)((box >> 16) & 255u); y1 = (int)(box >> 24); }
    else {
        x0 = (imin3(v0.X, v1.X, v2.X) - 128 + 255) >> 8; x1 = (imax3(v0.X, v1.X, v2.X) - 128) >> 8;
        y0 = (imin3(v0.Y, v1.Y, v2.Y) - 128 + 255) >> 8; y1 = (imax3(v0.Y, v1.Y, v2.Y) - 128) >> 8;
        x0 = max(x0, T.px0); y0 = max(y0, T.py0);
        x1 = min(x1, min(T.px0 + SPAN(MODE) - 1, T.W - 1)); y1 = min(y1, min(T.py0 + SPAN(MODE) - 1, T.H - 1));
    }
    if (x0 > x1 || y0 > y1) return;

    const int sgn = A > 0 ? 1 : -1;
    // oriented edges (inside positive): e0 = v1->v2, e1 = v2->v0, e2 = v0->v1; top-left rule folded in as a bias
    const int ex0 = sgn * (v2.X - v1.X), ey0 = sgn * (v2.Y - v1.Y);
    const int ex1 = sgn * (v0.X - v2.X), ey1 = sgn * (v0.Y - v2.Y);
    const int ex2 = sgn * (v1.X - v0.X), ey2 = sgn * (v1.Y - v0.Y);
    const int Px0 = x0 * 256 + 128, Py0 = y0 * 256 + 128;
    const int tl0 = ((ey0 < 0) || (ey0 == 0 && ex0 > 0)) ? 0 : 1, tl1 = ((ey1 < 0) || (ey1 == 0 && ex1 > 0)) ? 0 : 1;
    const int tl2 = ((ey2 < 0) || (ey2 == 0 && ex2 > 0)) ? 0 : 1;

    // depth plane anchored at vertex 0, gradients per sub-pixel unit
    const float invA = 1.0f / (float)A;
    const float a1 = (float)(v2.Y - v0.Y) * invA, b1 = (float)(v0.X - v2.X) * invA;
    const float a2 = (float)(v0.Y - v1.Y) * invA, b2 = (float)(v1.X - v0.X) * invA;
    const float dz1 = v1.z - v0.z, dz2 = v2.z - v0.z;
    const float gx = __builtin_fmaf(a2, dz2, a1 * dz1), gy = __builtin_fmaf(b2, dz2, b1 * dz1);
    const float zlo = __builtin_fminf(__builtin_fminf(v0.z, v1.z), v2.z), zhi = __builtin_fmaxf(__builtin_fmaxf(v0.z, v1.z), v2.z);
    float bias = 0.0f;
    if (MODE == ZR_MODE_SHADOW) {
        // vkCmdSetDepthBias(1.25, 0, 7.5) on D32 (ZE:3280-3287): o = m * slope + r * constant, r = 2^(e - 23)
        const float m = __builtin_fmaxf(__builtin_fabsf(gx), __builtin_fabsf(gy)) * 256.0f;
        const float zm = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v0.z), __builtin_fabsf(v1.z)), __builtin_fabsf(v2.z));
        const uint32_t e = zr_f2u(zm) & 0x7F800000u;
        const float r = (e > (23u << 23) && e < 0x7F800000u) ? zr_u2f(e - (23u << 23)) : 0.0f;
        bias = __builtin_fmaf(m, 7.5f, r * 1.25f);
    }
    int e0 = 0, e1 = 0, e2 = 0;
    long long E0 = 0, E1 = 0, E2 = 0;
    bool fits;
    if (all_small) {
        e0 = ex0 * (Py0 - v1.Y) - ey0 * (Px0 - v1.X) - tl0;
        e1 = ex1 * (Py0 - v2.Y) - ey1 * (Px0 - v2.X) - tl1;
        e2 = ex2 * (Py0 - v0.Y) - ey2 * (Px0 - v0.X) - tl2;
        fits = true;
    } else {
        E0 = (long long)ex0 * (Py0 - v1.Y) - (long long)ey0 * (Px0 - v1.X) - tl0;
        E1 = (long long)ex1 * (Py0 - v2.Y) - (long long)ey1 * (Px0 - v2.X) - tl1;
        E2 = (long long)ex2 * (Py0 - v0.Y) - (long long)ey2 * (Px0 - v0.X) - tl2;
        // |E| anywhere in the walk <= |E at the origin| + nx*|sx| + ny*|sy|
        const long long nx = x1 - x0 + 1, ny = y1 - y0 + 1;
        const long long lim = 0x3FFFFFFFll;
        const long long m0 = (E0 < 0 ? -E0 : E0) + 256 * (nx * (ey0 < 0 ? -(long long)ey0 : ey0) + ny * (ex0 < 0 ? -(long long)ex0 : ex0));
        const long long m1 = (E1 < 0 ? -E1 : E1) + 256 * (nx * (ey1 < 0 ? -(long long)ey1 : ey1) + ny * (ex1 < 0 ? -(long long)ex1 : ex1));
        const long long m2 = (E2 < 0 ? -E2 : E2) + 256 * (nx * (ey2 < 0 ? -(long long)ey2 : ey2) + ny * (ex2 < 0 ? -(long long)ex2 : ex2));
        fits = m0 < lim && m1 < lim && m2 < lim;
        if (fits) { e0 = (int)E0; e1 = (int)E1; e2 = (int)E2; }
    }
    if (fits) {
        const int sx0 = -ey0 * 256, sx1 = -ey1 * 256, sx2 = -ey2 * 256, sy0 = ex0 * 256, sy1 = ex1 * 256, sy2 = ex2 * 256;
        for (int y = y0; y <= y1; ++y) {
            int r0 = e0, r1 = e1, r2 = e2;
            const float fy = (float)(y * 256 + 128 - v0.Y);
            for (int x = x0; x <= x1; ++x) {
                if ((r0 | r1 | r2) >= 0) shade_key<MODE>(x, y, fy, v0, gx, gy, zlo, zhi, bias, prim, T, keys64, keys32);
                r0 += sx0; r1 += sx1; r2 += sx2;
            }
            e0 += sy0; e1 += sy1; e2 += sy2;
        }
    } else {
        const long long sx0 = -(long long)ey0 * 256, sx1 = -(long long)ey1 * 256, sx2 = -(long long)ey2 * 256;
        const long long sy0 = (long long)ex0 * 256, sy1 = (long long)ex1 * 256, sy2 = (long long)ex2 * 256;
        for (int y = y0; y <= y1; ++y) {
            long long r0 = E0, r1 = E1, r2 = E2;
            const float fy = (float)(y * 256 + 128 - v0.Y);
            for (int x = x0; x <= x1; ++x) {
                if ((r0 | r1 | r2) >= 0) shade_key<MODE>(x, y, fy, v0, gx, gy, zlo, zhi, bias, prim, T, keys64, keys32);
                r0 += sx0; r1 += sx1; r2 += sx2;
            }
            E0 += sy0; E1 += sy1; E2 += sy2;
        }
    }
}

// Sutherland-Hodgman against near (z >= 0) and the 4x guard band; intersections always run inside -> outside.
__device__ __forceinline__ float plane_dist(zf4 c, int plane)
{
    switch (plane) {
    case 0: return c.z;
    case 1: return __builtin_fmaf(ZR_GUARD, c.w, c.x);
    case 2: return __builtin_fmaf(ZR_GUARD, c.w, -c.x);
    case 3: return __builtin_fmaf(ZR_GUARD, c.w, c.y);
    default: return __builtin_fmaf(ZR_GUARD, c.w, -c.y);
    }
}
__device__ __forceinline__ zf4 lerp4(zf4 in, zf4 out, float t)
{
    zf4 r;
    r.x = __builtin_fmaf(t, out.x - in.x, in.x); r.y = __builtin_fmaf(t, out.y - in.y, in.y);
    r.z = __builtin_fmaf(t, out.z - in.z, in.z); r.w = __builtin_fmaf(t, out.w - in.w, in.w);
    return r;
}
// T and the keys are tile-relative (see k_raster_chunks): (ox, oy) = the tile's origin in sub-pixel units is taken off after projecting.
template <int MODE>
__device__ __forceinline__ void raster_clipped_body(zf4 c0, zf4 c1, zf4 c2, uint32_t prim, TileCtx T, float hw, float hh, int ox, int oy,
                                                    unsigned long long* keys64, uint32_t* keys32)
{
    zf4 a[10], b[10];
    int na = 3;
    a[0] = c0; a[1] = c1; a[2] = c2;
    for (int plane = 0; plane < 5; ++plane) {
        int nb = 0;
        for (int i = 0; i < na; ++i) {
            const zf4 p = a[i], q = a[(i + 1 == na) ? 0 : i + 1];
            const float dp = plane_dist(p, plane), dq = plane_dist(q, plane);
            const bool ip = dp >= 0.0f, iq = dq >= 0.0f;
            if (ip) b[nb++] = p;
            if (ip != iq) b[nb++] = ip ? lerp4(p, q, dp / (dp - dq)) : lerp4(q, p, dq / (dq - dp));
        }
        na = nb;
        if (na < 3) return;
        for (int i = 0; i < na; ++i) a[i] = b[i];
    }
    for (int i = 0; i < na; ++i) if (!(a[i].w > 0.0f)) return;
    SV s0 = project(a[0], hw, hh); s0.X -= ox; s0.Y -= oy;
    SV sp = project(a[1], hw, hh); sp.X -= ox; sp.Y -= oy;
    for (int i = 2; i < na; ++i) {
        SV sn = project(a[i], hw, hh); sn.X -= ox; sn.Y -= oy;
        raster_sub<MODE, false>(s0, sp, sn, prim, T, keys64, keys32);
        sp = sn;
    }
}
// (a call in the rasterisers' loops - their register budget must not carry the clipper's; inlined in k_tile_slow, whose own budget is set
// so that it finds room beside the shadow rasteriser)
template <int MODE>
__device__ __noinline__ void raster_clipped(zf4 c0, zf4 c1, zf4 c2, uint32_t prim, TileCtx T, float hw, float hh, int ox, int oy,
                                            unsigned long long* keys64, uint32_t* keys32)
{
    raster_clipped_body<MODE>(c0, c1, c2, prim, T, hw, hh, ox, oy, keys64, keys32);
}

// The slow triangles of the round (they need the clipper, or have an edge of 64 px or more): every owned tile tries every one of
// them through raster_clipped.  A SMALL persistent grid (ZR_SLOW_BLOCKS workgroups stride over the owned tiles): the usual round has no
// slow triangle, and this launch sits on the camera lane's critical path - as one workgroup per owned tile (2 040 at 1080p, 8 KB of LDS
// each) it cost 35 us beside the shadow rasteriser just to find room and return; a few dozen workgroups come and go like k_scan_tri's one.
#ifndef ZR_SLOW_BLOCKS
#define ZR_SLOW_BLOCKS 256u
#endif
template <int MODE, bool BY_TILE>
__global__ __launch_bounds__(256) void k_tile_slow(ZrPass P, const uint32_t* __restrict__ owned_tiles, uint32_t n_owned, const uint4* __restrict__ slow,
                                                   uint32_t slow_cap, ZrDevStats* __restrict__ stats, int slot,
                                                   unsigned long long* __restrict__ vis64, uint32_t* __restrict__ shadow_bits,
                                                   const uint32_t* __restrict__ wave_culled, uint32_t n_waves)
{
    __shared__ unsigned long long keys64[MODE == ZR_MODE_GBUFFER ? TILE_PIX : 1];
    __shared__ uint32_t keys32[MODE == ZR_MODE_SHADOW ? SPAN_PIX(MODE) : 1];
    // camera pass: round 1's triangles sit in the first half of the list, round 2's in the second; one launch after round 2 draws both
    // (slot = 2), or round 1's alone in a one-round frame (slot = 1)
    const uint32_t half = BY_TILE ? slow_cap : slow_cap / 2u;
    const uint32_t n_a = min(stats->n_slow[BY_TILE ? slot : 1], half), n_b = (!BY_TILE && slot == 2) ? min(stats->n_slow[2], half) : 0u;
    if (!BY_TILE && slot == 2 && wave_culled) {
        // the meshlets round 2's k_geom dropped behind the pyramid: the per-wave counts, strided over this grid (one atomic per wave of
        // THAT kernel on one address would queue up for ~10 ns apiece and hold its end)
        uint32_t nc = 0;
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_waves; i += gridDim.x * 256u) nc += wave_culled[i];
        nc = (uint32_t)wave_sum((int)nc);
        if ((threadIdx.x & 63u) == 0u && nc) { atomicAdd(&stats->hiz_culled, nc); atomicAdd(&stats->hiz_culled_geom, nc); atomicSub(&stats->survivors[2], nc); }
    }
    if (n_a + n_b == 0u) return;
    const uint32_t tid = threadIdx.x;
    for (uint32_t ti = blockIdx.x; ti < n_owned; ti += gridDim.x) {
        const uint32_t tile = owned_tiles[ti];
        for (uint32_t i = tid; i < (uint32_t)SPAN_PIX(MODE); i += 256u) {
            if (MODE == ZR_MODE_GBUFFER) keys64[i] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
            else keys32[i] = 0x3F800000u;
        }
        __syncthreads();
        const int tpx0 = (int)(tile % P.tiles_x) * TILE, tpy0 = (int)(tile / P.tiles_x) * TILE;
        TileCtx T;
        T.px0 = 0; T.py0 = 0; T.W = (int)P.W - tpx0; T.H = (int)P.H - tpy0;
        for (uint32_t jj = tid; jj < n_a + n_b; jj += 256u) {
            const uint32_t j = jj < n_a ? jj : half + (jj - n_a);
            const uint4 q3 = slow[4u * j + 3u];
            if (BY_TILE && q3.y != tile) continue;          // (the meshlet-binned rasteriser lists a triangle once per tile of its meshlet)
            if (!BY_TILE) {                                 // camera pass: the tiles the triangle's snapped box reaches (k_geom), or all of them
                const uint32_t ttx = tile % P.tiles_x, tty = tile / P.tiles_x;
                if (ttx < (q3.y & 255u) || tty < ((q3.y >> 8) & 255u) || ttx > ((q3.y >> 16) & 255u) || tty > (q3.y >> 24)) continue;
            }
            const uint4 q0 = slow[4u * j], q1 = slow[4u * j + 1u], q2 = slow[4u * j + 2u];
            zf4 c0, c1, c2;
            c0.x = zr_u2f(q0.x); c0.y = zr_u2f(q0.y); c0.z = zr_u2f(q0.z); c0.w = zr_u2f(q0.w);
            c1.x = zr_u2f(q1.x); c1.y = zr_u2f(q1.y); c1.z = zr_u2f(q1.z); c1.w = zr_u2f(q1.w);
            c2.x = zr_u2f(q2.x); c2.y = zr_u2f(q2.y); c2.z = zr_u2f(q2.z); c2.w = zr_u2f(q2.w);
            raster_clipped<MODE>(c0, c1, c2, q3.x, T, P.hw, P.hh, tpx0 * 256, tpy0 * 256, keys64, keys32);
        }
        __syncthreads();
        for (uint32_t i = tid; i < (uint32_t)SPAN_PIX(MODE); i += 256u) {
            const int px = tpx0 + (int)(i % (uint32_t)SPAN(MODE)), py = tpy0 + (int)(i / (uint32_t)SPAN(MODE));
            if (px >= (int)P.W || py >= (int)P.H) continue;
            const size_t p = (size_t)py * P.W + (size_t)px;
            if (MODE == ZR_MODE_GBUFFER) {
                const unsigned long long k = keys64[i];
                if ((uint32_t)k != ZR_EMPTY_PRIM && k < vis64[p]) atomicMin(&vis64[p], k);
            } else {
                const uint32_t k = keys32[i];
                if (k != 0x3F800000u && k < shadow_bits[p]) atomicMin(&shadow_bits[p], k);
            }
        }
        __syncthreads();      // the keys are cleared again for the next tile
    }
}
