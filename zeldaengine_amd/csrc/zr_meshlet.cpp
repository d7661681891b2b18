// zr_meshlet.cpp — the library's own meshlet clusteriser and bounds (host, load time).
//
// Replaces the ZeldaMeshlet tool's BuildMeshlets (Engine/ZeldaMeshlet/ZeldaMeshlet.cpp:132-172), which calls
// meshoptimizer (absent from the reference tree; submodule pin unrecoverable).  The partition itself never
// reaches a pixel (culling is conservative, the depth/visibility key carries the draw-order triangle id), so
// this is NOT a restatement of meshopt_buildMeshlets; it is a greedy adjacency clusteriser honouring the same
// contract: <= max_vertices unique vertices, <= max_triangles triangles, every triangle in exactly one
// meshlet, bounds that enclose, and a normal cone per meshoptimizer's published definition
// (cutoff = sqrt(1 - mindp^2), 1 when the cone is wider than ~84 degrees).
#include "zr_meshlet.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

struct V3 { double x, y, z; };
inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator*(V3 a, double s) { return { a.x * s, a.y * s, a.z * s }; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline V3 pos(const XkVertex& v) { return { v.Position[0], v.Position[1], v.Position[2] }; }

}  // namespace

void zr_meshlet_bounds(const XkVertex* verts, const uint32_t* mv, uint32_t nv, const uint8_t* mt, uint32_t nt, XkMeshlet* out)
{
    // bounding sphere: Ritter's two-pass sphere over the meshlet's vertices, then inflated to enclose exactly
    V3 c = { 0, 0, 0 };
    double r = 0;
    if (nv) {
        V3 p0 = pos(verts[mv[0]]);
        uint32_t a = 0, b = 0; double best = -1;
        for (uint32_t i = 0; i < nv; ++i) { V3 d = pos(verts[mv[i]]) - p0; double l = dot(d, d); if (l > best) { best = l; a = i; } }
        V3 pa = pos(verts[mv[a]]); best = -1;
        for (uint32_t i = 0; i < nv; ++i) { V3 d = pos(verts[mv[i]]) - pa; double l = dot(d, d); if (l > best) { best = l; b = i; } }
        V3 pb = pos(verts[mv[b]]);
        c = (pa + pb) * 0.5; r = std::sqrt(best) * 0.5;
        for (uint32_t i = 0; i < nv; ++i) {
            V3 d = pos(verts[mv[i]]) - c; double l = std::sqrt(dot(d, d));
            if (l > r) { double nr = (r + l) * 0.5; c = c + d * ((nr - r) / l); r = nr; }
        }
        float cf[3] = { (float)c.x, (float)c.y, (float)c.z };
        double rr = 0;
        for (uint32_t i = 0; i < nv; ++i) {
            const float* p = verts[mv[i]].Position;
            double dx = (double)p[0] - cf[0], dy = (double)p[1] - cf[1], dz = (double)p[2] - cf[2];
            rr = std::max(rr, std::sqrt(dx * dx + dy * dy + dz * dz));
        }
        r = rr;
    }
    out->BoundsCenter[0] = (float)c.x; out->BoundsCenter[1] = (float)c.y; out->BoundsCenter[2] = (float)c.z;
    out->BoundsRadius = std::nextafter((float)(r * (1.0 + 1e-6)), INFINITY);

    // normal cone
    V3 axis = { 0, 0, 0 };
    std::vector<V3> n(nt);
    for (uint32_t t = 0; t < nt; ++t) {
        V3 a = pos(verts[mv[mt[3 * t]]]), b = pos(verts[mv[mt[3 * t + 1]]]), d = pos(verts[mv[mt[3 * t + 2]]]);
        V3 nn = cross(b - a, d - a);
        double l = std::sqrt(dot(nn, nn));
        n[t] = l > 0 ? nn * (1.0 / l) : V3{ 0, 0, 0 };
        axis = axis + n[t];
    }
    double al = std::sqrt(dot(axis, axis));
    double mindp = 1.0;
    if (al > 0) {
        axis = axis * (1.0 / al);
        for (uint32_t t = 0; t < nt; ++t) mindp = std::min(mindp, dot(n[t], axis));
    } else { axis = { 1, 0, 0 }; mindp = -1.0; }
    out->ConeAxis[0] = (float)axis.x; out->ConeAxis[1] = (float)axis.y; out->ConeAxis[2] = (float)axis.z;
    // degenerate cluster (cone wider than a hemisphere, or nearly so): cutoff 1 = never culled
    out->ConeCutoff = (mindp <= 0.1) ? 1.0f : std::nextafter((float)std::sqrt(1.0 - mindp * mindp), 2.0f);
    // apex: meshoptimizer backs the centre off along the axis far enough to see every triangle's back side
    double maxt = 0;
    if (mindp > 0.1)
        for (uint32_t t = 0; t < nt; ++t) {
            V3 a = pos(verts[mv[mt[3 * t]]]);
            double dc = dot(c - a, n[t]), dn = dot(axis, n[t]);
            if (dn > 1e-12) maxt = std::max(maxt, dc / dn);
        }
    V3 apex = c - axis * maxt;
    out->ConeApex[0] = (float)apex.x; out->ConeApex[1] = (float)apex.y; out->ConeApex[2] = (float)apex.z;
}

void zr_build_meshlets(const XkVertex* verts, uint32_t nv, const uint32_t* idx, uint32_t ni,
                       uint32_t max_vertices, uint32_t max_triangles, float cone_weight, ZrMeshletSet* out)
{
    const uint32_t nt = ni / 3;
    out->meshlets.clear(); out->mverts.clear(); out->mtris.clear(); out->tri_order.clear();
    if (nt == 0) return;

    // triangle centroid + unit normal
    std::vector<V3> cen(nt), nrm(nt);
    for (uint32_t t = 0; t < nt; ++t) {
        V3 a = pos(verts[idx[3 * t]]), b = pos(verts[idx[3 * t + 1]]), c = pos(verts[idx[3 * t + 2]]);
        cen[t] = (a + b + c) * (1.0 / 3.0);
        V3 n = cross(b - a, c - a); double l = std::sqrt(dot(n, n));
        nrm[t] = l > 0 ? n * (1.0 / l) : V3{ 0, 0, 0 };
    }
    // vertex -> triangle adjacency (CSR)
    std::vector<uint32_t> vstart(nv + 1, 0), vtri(ni);
    for (uint32_t i = 0; i < ni; ++i) vstart[idx[i] + 1]++;
    for (uint32_t v = 0; v < nv; ++v) vstart[v + 1] += vstart[v];
    { std::vector<uint32_t> fill(vstart.begin(), vstart.end() - 1);
      for (uint32_t i = 0; i < ni; ++i) vtri[fill[idx[i]]++] = i / 3; }

    std::vector<uint8_t> used(nt, 0);
    std::vector<uint8_t> vlocal(nv, 0xFF);
    std::vector<uint32_t> stamp(nt, 0xFFFFFFFFu);
    std::vector<uint32_t> cur_v; cur_v.reserve(max_vertices);
    std::vector<uint32_t> cur_t; cur_t.reserve(max_triangles);
    V3 csum = { 0, 0, 0 }, nsum = { 0, 0, 0 };
    uint32_t seed = 0, done = 0, tri_base = 0;
    (void)0;

    auto flush = [&]() {
        if (cur_t.empty()) return;
        XkMeshlet m; std::memset(&m, 0, sizeof m);
        m.VertexOffset = (uint32_t)out->mverts.size(); m.VertexCount = (uint32_t)cur_v.size();
        m.TriangleOffset = (uint32_t)out->mtris.size(); m.TriangleCount = (uint32_t)cur_t.size();
        m.BindlessContext = tri_base;
        for (uint32_t v : cur_v) out->mverts.push_back(v);
        for (uint32_t t : cur_t) {
            for (int k = 0; k < 3; ++k) out->mtris.push_back(vlocal[idx[3 * t + k]]);
            out->tri_order.push_back(t);
        }
        zr_meshlet_bounds(verts, out->mverts.data() + m.VertexOffset, m.VertexCount,
                          out->mtris.data() + m.TriangleOffset, m.TriangleCount, &m);
        out->meshlets.push_back(m);
        tri_base += m.TriangleCount;
        for (uint32_t v : cur_v) vlocal[v] = 0xFF;
        cur_v.clear(); cur_t.clear(); csum = { 0, 0, 0 }; nsum = { 0, 0, 0 };
    };
    auto extra_of = [&](uint32_t t) {
        uint32_t a = idx[3 * t], b = idx[3 * t + 1], c = idx[3 * t + 2];
        uint32_t e = (vlocal[a] == 0xFF);
        if (b != a) e += (vlocal[b] == 0xFF);
        if (c != a && c != b) e += (vlocal[c] == 0xFF);
        return e;
    };
    auto add = [&](uint32_t t) {
        for (int k = 0; k < 3; ++k) {
            uint32_t v = idx[3 * t + k];
            if (vlocal[v] == 0xFF) { vlocal[v] = (uint8_t)cur_v.size(); cur_v.push_back(v); }
        }
        cur_t.push_back(t); used[t] = 1; done++;
        csum = csum + cen[t]; nsum = nsum + nrm[t];
    };

    // mean squared triangle "size" (centroid-to-vertex), used to price a new vertex against distance from the centroid
    double size2 = 0;
    for (uint32_t t = 0; t < nt; ++t) { V3 d = pos(verts[idx[3 * t]]) - cen[t]; size2 += dot(d, d); }
    size2 /= (double)nt;

    while (done < nt) {
        uint32_t best = 0xFFFFFFFFu; double best_score = 0;
        if (!cur_t.empty()) {
            const double inv = 1.0 / (double)cur_t.size();
            const V3 cc = csum * inv;
            double nl = std::sqrt(dot(nsum, nsum));
            const V3 na = nl > 0 ? nsum * (1.0 / nl) : V3{ 0, 0, 0 };
            const uint32_t mark = done;   // unique per step
            for (uint32_t v : cur_v)
                for (uint32_t k = vstart[v]; k < vstart[v + 1]; ++k) {
                    const uint32_t t = vtri[k];
                    if (used[t] || stamp[t] == mark) continue;
                    stamp[t] = mark;
                    // compact, cone-coherent growth: distance to the running centroid, widened by the normal deviation,
                    // plus a price per vertex the triangle would add (triangles closing a fan are nearly free)
                    const V3 d = cen[t] - cc;
                    const double score = dot(d, d) * (1.0 + (double)cone_weight * (1.0 - dot(nrm[t], na))) + size2 * (double)extra_of(t);
                    if (best == 0xFFFFFFFFu || score < best_score) { best = t; best_score = score; }
                }
        }
        if (best == 0xFFFFFFFFu) {          // no neighbour left: close the cluster and reseed
            flush();
            while (used[seed]) seed++;
            best = seed;
        } else if (cur_t.size() + 1 > max_triangles || cur_v.size() + extra_of(best) > max_vertices) {
            // full: close it and reseed next to it, at the frontier triangle with the fewest unused neighbours (keeps the
            // uncovered region convex-ish and avoids stranding islands of a few triangles)
            uint32_t pick = 0xFFFFFFFFu; uint32_t pick_live = 0xFFFFFFFFu;
            const uint32_t mark = done | 0x80000000u;
            for (uint32_t v : cur_v)
                for (uint32_t k = vstart[v]; k < vstart[v + 1]; ++k) {
                    const uint32_t t = vtri[k];
                    if (used[t] || stamp[t] == mark) continue;
                    stamp[t] = mark;
                    uint32_t live = 0;
                    for (int c3 = 0; c3 < 3; ++c3) { const uint32_t vv = idx[3 * t + c3]; for (uint32_t kk = vstart[vv]; kk < vstart[vv + 1]; ++kk) live += !used[vtri[kk]]; }
                    if (live < pick_live) { pick_live = live; pick = t; }
                }
            flush();
            best = pick != 0xFFFFFFFFu ? pick : best;
        }
        add(best);
    }
    flush();
}
