// zr_surface.h — from the key buffer's winner to a shaded-ready surface: perspective-correct interpolation with fine derivatives over the
// 2 x 2 quad (pixel_geom), ComputeNormal (SH/Common.glsl:113-127) and BaseScene.frag:26-48 with its render-target packing (resolve_pixel).
#pragma once
#include "zr_dev.h"
#include "zr_texture.h"

struct Bary { float b0, b1, b2; };
struct FastSetup { int X0, Y0; float a1, b1, a2, b2, rw0, rw1, rw2; };

__device__ __forceinline__ Bary bary_screen(const FastSetup& s, int px, int py)
{
    const float fx = (float)(px * 256 + 128 - s.X0), fy = (float)(py * 256 + 128 - s.Y0);
    const float l1 = __builtin_fmaf(s.b1, fy, s.a1 * fx), l2 = __builtin_fmaf(s.b2, fy, s.a2 * fx);
    const float l0 = (1.0f - l1) - l2;
    const float q0 = l0 * s.rw0, q1 = l1 * s.rw1, q2 = l2 * s.rw2;
    const float inv = 1.0f / ((q0 + q1) + q2);
    Bary r; r.b0 = q0 * inv; r.b1 = q1 * inv; r.b2 = q2 * inv;
    return r;
}
// clipped triangles: 2D-homogeneous interpolation from the unclipped clip-space vertices
__device__ __forceinline__ Bary bary_homog(const zf4* c, float hw, float hh, int px, int py)
{
    const float u = (((float)px + 0.5f) - hw) / hw, v = (((float)py + 0.5f) - hh) / hh;
    float k[3];
    for (int i = 0; i < 3; ++i) {
        const zf4 p = c[(i + 1) % 3], q = c[(i + 2) % 3];
        const float kx = __builtin_fmaf(p.y, q.w, -(q.y * p.w));
        const float ky = __builtin_fmaf(q.x, p.w, -(p.x * q.w));
        const float kz = __builtin_fmaf(p.x, q.y, -(q.x * p.y));
        k[i] = __builtin_fmaf(kx, u, __builtin_fmaf(ky, v, kz));
    }
    const float inv = 1.0f / ((k[0] + k[1]) + k[2]);
    Bary r; r.b0 = k[0] * inv; r.b1 = k[1] * inv; r.b2 = k[2] * inv;
    return r;
}
__device__ __forceinline__ float interp1(Bary b, float a0, float a1, float a2)
{
    return __builtin_fmaf(b.b2, a2, __builtin_fmaf(b.b1, a1, b.b0 * a0));
}
__device__ __forceinline__ zf3 interp3(Bary b, zf3 a0, zf3 a1, zf3 a2)
{
    return zr3(interp1(b, a0.x, a1.x, a2.x), interp1(b, a0.y, a1.y, a2.y), interp1(b, a0.z, a1.z, a2.z));
}

// ComputeNormal(fragPosition, fragTexCoord, fragNormal, texNormal), SH/Common.glsl:113-127; ts = zr_tangent_space_normal(texNormal)
__device__ __forceinline__ zf3 compute_normal(zf3 pos_dx, zf3 pos_dy, float s1, float t1, float s2, float t2, zf3 fragN, zf3 ts)
{
    // (vec3 / scalar: one IEEE reciprocal, three multiplies - DESIGN.md section 4)
    const float rdet = 1.0f / __builtin_fmaf(s1, t2, -(s2 * t1));
    zf3 T = zr3(__builtin_fmaf(t2, pos_dx.x, -(t1 * pos_dy.x)) * rdet,
                __builtin_fmaf(t2, pos_dx.y, -(t1 * pos_dy.y)) * rdet,
                __builtin_fmaf(t2, pos_dx.z, -(t1 * pos_dy.z)) * rdet);
    const zf3 N = zr_normalize(fragN);
    T = zr_normalize(T - N * zr_dot(N, T));
    const zf3 B = zr_normalize(zr_cross(N, T));
    const zf3 w = zr3(__builtin_fmaf(N.x, ts.z, __builtin_fmaf(B.x, ts.y, T.x * ts.x)),
                      __builtin_fmaf(N.y, ts.z, __builtin_fmaf(B.y, ts.y, T.y * ts.x)),
                      __builtin_fmaf(N.z, ts.z, __builtin_fmaf(B.z, ts.y, T.z * ts.x)));
    return zr_normalize(w);
}

// fp32 -> fp16 with the conversion unit (round to nearest even, denormals kept, overflow to inf: what zr_f32_to_f16 spells out
// in integer arithmetic for the host); NaN is canonicalised as there.
__device__ __forceinline__ uint32_t f32_to_f16_hw(float f)
{
    const _Float16 h = (_Float16)f;
    uint16_t b; __builtin_memcpy(&b, &h, 2);
    return (f != f) ? (((zr_f2u(f) >> 16) & 0x8000u) | 0x7E00u) : (uint32_t)b;
}
__device__ __forceinline__ float f16_to_f32_hw(uint32_t h)      // exact (every fp16 value is an fp32 value); quiet NaNs map as in zr_f16_to_f32
{
    const uint16_t b = (uint16_t)h;
    _Float16 v; __builtin_memcpy(&v, &b, 2);
    return (float)v;
}
// M * vec4(p, 1) when M may be the identity: for finite p every product with a zero entry is +-0, the sum is p (or a zero of either
// sign) and the final "+ M[12]" with M[12] = +0 turns -0 into +0 - which is exactly what p + 0.0f does.
__device__ __forceinline__ zf3 model_point(const ZrPass& P, zf3 p)
{
    if (P.m_identity) return zr3(p.x + 0.0f, p.y + 0.0f, p.z + 0.0f);
    const zf4 w = zr_mat4_point(P.M, p);
    return zr3(w.x, w.y, w.z);
}

// The fragment of primitive `prim` at pixel (px, py): what the rasteriser and the vertex stage hand a fragment shader (Base.vert /
// BaseInstanced.vert outputs interpolated perspective-correctly, and their fine derivatives over the pixel's 2 x 2 quad).
struct PixGeom {
    const ZrObject* O; uint32_t tri, inst_i;
    zf3 P0, N0, pos_dx, pos_dy;            // fragPosition, fragNormal, dFdx / dFdy(fragPosition)
    float u0, v0, s1, t1, s2, t2;          // fragTexCoord, dFdx(uv) = (s1, t1), dFdy(uv) = (s2, t2)
    Bary b0;                               // the pixel's own barycentrics (for whatever else is interpolated)
};
__device__ __forceinline__ PixGeom pixel_geom(const ZrPass& P, const ZrObject* __restrict__ objs, uint32_t prim, int px, int py,
                                              uint8_t* __restrict__ vis_now = nullptr, uint32_t vis_mark = 1u)
{
    PixGeom g;
    const ZrObject* __restrict__ O = objs + find_object_prim(objs, (int)P.n_objects, prim);
    const uint32_t local = prim - O->prim_base;
    const uint32_t inst_i = local / O->n_tris, tri = local - inst_i * O->n_tris;
    const bool instanced = O->instanced != 0;
    const ZrInstance I = ld_record(O->inst + inst_i);
    g.O = O; g.tri = tri; g.inst_i = inst_i;
    // visibility history for next frame's round 1: this meshlet-instance owns a pixel
    if (vis_now) vis_now[O->work_base + inst_i * O->n_meshlets + ld_global(O->tri_meshlet + tri)] = (uint8_t)vis_mark;
    zf4 clip[3]; zf3 WP[3], WN[3]; float U[3], V[3]; uint32_t fl[3];
    for (int k = 0; k < 3; ++k) {
        const float4* __restrict__ rv = (const float4*)(O->rtris + 3u * tri + (uint32_t)k);       // (= rverts[indices[3 tri + k]])
        const float4 q0 = ld_global(rv), q1 = ld_global(rv + 1);       // position.xyz u | normalize(normal).xyz v
        const zf3 pos = vs_position(zr3(q0.x, q0.y, q0.z), I, instanced);
        clip[k] = zr_mat4_point(P.PVM, pos);
        WP[k] = model_point(P, pos);
        // outNormal = (M * vec4(normalize(n), 1)).xyz [* mat3(rotMat)], Base.vert:29 / BaseInstanced.vert:73
        const zf3 mn = model_point(P, zr3(q1.x, q1.y, q1.z));
        WN[k] = instanced ? zr_rowvec_mat3(mn, I.R) : mn;
        U[k] = q0.w; V[k] = q1.w;
        fl[k] = vertex_flags(clip[k]);
    }
    const int cls = classify(fl[0], fl[1], fl[2]);
    const int qx = px ^ 1, qy = py ^ 1;           // 2x2 quad partners (fine derivatives, helper lanes extrapolate)
    Bary b0, bh, bv;
    if (cls == 1) {
        const SV s0 = project(clip[0], P.hw, P.hh), s1 = project(clip[1], P.hw, P.hh), s2 = project(clip[2], P.hw, P.hh);
        const long long A = (long long)(s1.X - s0.X) * (s2.Y - s0.Y) - (long long)(s2.X - s0.X) * (s1.Y - s0.Y);
        const float invA = 1.0f / (float)A;
        FastSetup fs;
        fs.X0 = s0.X; fs.Y0 = s0.Y;
        fs.a1 = (float)(s2.Y - s0.Y) * invA; fs.b1 = (float)(s0.X - s2.X) * invA;
        fs.a2 = (float)(s0.Y - s1.Y) * invA; fs.b2 = (float)(s1.X - s0.X) * invA;
        fs.rw0 = s0.rw; fs.rw1 = s1.rw; fs.rw2 = s2.rw;
        b0 = bary_screen(fs, px, py); bh = bary_screen(fs, qx, py); bv = bary_screen(fs, px, qy);
    } else {
        b0 = bary_homog(clip, P.hw, P.hh, px, py); bh = bary_homog(clip, P.hw, P.hh, qx, py); bv = bary_homog(clip, P.hw, P.hh, px, qy);
    }
    const zf3 P0 = interp3(b0, WP[0], WP[1], WP[2]), Ph = interp3(bh, WP[0], WP[1], WP[2]), Pv = interp3(bv, WP[0], WP[1], WP[2]);
    g.P0 = P0; g.b0 = b0;
    g.N0 = interp3(b0, WN[0], WN[1], WN[2]);
    const float u0 = interp1(b0, U[0], U[1], U[2]), uh = interp1(bh, U[0], U[1], U[2]), uv_ = interp1(bv, U[0], U[1], U[2]);
    const float v0 = interp1(b0, V[0], V[1], V[2]), vh = interp1(bh, V[0], V[1], V[2]), vv = interp1(bv, V[0], V[1], V[2]);
    const float sx = (px & 1) ? 1.0f : -1.0f, sy = (py & 1) ? 1.0f : -1.0f;
    g.pos_dx = (P0 - Ph) * sx; g.pos_dy = (P0 - Pv) * sy;
    g.u0 = u0; g.v0 = v0;
    g.s1 = (u0 - uh) * sx; g.t1 = (v0 - vh) * sx; g.s2 = (u0 - uv_) * sy; g.t2 = (v0 - vv) * sy;
    return g;
}

// BaseScene.frag:26-48 for the pixel (px, py) whose winning primitive is `prim`
// returns true when the pixel holds scene geometry (not empty, not sky)
template <int IMAGES>
__device__ __forceinline__ bool resolve_pixel(const ZrPass& P, const ZrObject* __restrict__ objs, uint32_t prim, float depth,
                                              int px, int py, const GBufferPtrs& G, const float* __restrict__ lut,
                                              uint8_t* __restrict__ vis_now, uint32_t vis_mark = 1u)
{
    const size_t p = (size_t)py * P.W + (size_t)px;
    if (prim == ZR_EMPTY_PRIM) {   // clears, ZE:3427-3433
        G.depth[p] = 1.0f; G.scene_color[p] = 0xFF000000u; G.gA[p] = 0u; G.gB[p] = 0xFF000000u; G.gC[p] = 0xFF000000u;
        G.gD[p] = make_uint2(0u, 0x3C000000u);
        if (P.write_overlay) G.overlay[p] = 0u;
        return false;
    }
    const PixGeom g = pixel_geom(P, objs, prim, px, py, vis_now, vis_mark);
    const ZrObject* __restrict__ O = g.O;
    const zf3 P0 = g.P0, N0 = g.N0, pos_dx = g.pos_dx, pos_dy = g.pos_dy;
    const float u0 = g.u0, v0 = g.v0, s1 = g.s1, t1 = g.t1, s2 = g.s2, t2 = g.t2;

    if (O->flags & ZR_OBJ_SKY) {    // Skydome.frag: texture(skydomeSampler, uv).rgb, gamma - colour only, into the overlay plane: the pass
        // is drawn after the lighting quad (ZE:3681-3691) and no GBuffer attachment is written by it
        const zf4 sk = tex_sample<IMAGES>(O->tex[0], O->texc[0], true, lut, u0, v0, s1, t1, s2, t2);
        G.overlay[p] = zr_unorm(zr_pow(sk.x, 0.4545f), 255.0f) | zr_unorm(zr_pow(sk.y, 0.4545f), 255.0f) << 8 |
                       zr_unorm(zr_pow(sk.z, 0.4545f), 255.0f) << 16 | 255u << 24;
        return false;
    }
    if (P.write_overlay) G.overlay[p] = 0u;
    // texture(samplerN, fragTexCoord), BaseScene.frag:30-36; slot 0 (base colour) is R8G8B8A8_SRGB (ZE:5878).  Targets whose slots are
    // all constant were packed on the host (same zr_unorm), and so was the tangent-space normal of a constant normal map.
    uint32_t w_sc, w_gB, w_gC;
    zf3 ts;
    if (!IMAGES) {
        w_sc = O->c_scene_color; w_gB = O->c_gB; w_gC = O->c_gC;
        ts = zr3(O->ts_const[0], O->ts_const[1], O->ts_const[2]);
    } else {
        zf4 tb, tme, tro, tno, tao, tem, tms;
        if (IMAGES == 1) {                                 // every material of the scene that has images has them packed
            tb.x = O->texc[0][0]; tb.y = O->texc[0][1]; tb.z = O->texc[0][2]; tme.x = O->texc[1][0]; tro.x = O->texc[2][0];
            tno.x = O->texc[3][0]; tno.y = O->texc[3][1]; tno.z = O->texc[3][2]; tao.x = O->texc[4][0];
            tem.x = O->texc[5][0]; tem.y = O->texc[5][1]; tem.z = O->texc[5][2]; tms.x = O->texc[6][0];
            if (O->packed.data != nullptr) {
                float pk[ZR_PK_CHANNELS];
                tex_sample_packed(O->packed, lut, u0, v0, s1, t1, s2, t2, pk);
                const uint32_t cs = O->const_slots;         // a constant slot stays the constant (the oracle does not filter it)
                if (!(cs & 1u)) { tb.x = pk[ZR_PK_BC]; tb.y = pk[ZR_PK_BC + 1]; tb.z = pk[ZR_PK_BC + 2]; }
                if (!(cs & 2u)) tme.x = pk[ZR_PK_ME];
                if (!(cs & 4u)) tro.x = pk[ZR_PK_RO];
                if (!(cs & 8u)) { tno.x = pk[ZR_PK_NO]; tno.y = pk[ZR_PK_NO + 1]; tno.z = pk[ZR_PK_NO + 2]; }
                if (!(cs & 16u)) tao.x = pk[ZR_PK_AO];
                if (!(cs & 32u)) { tem.x = pk[ZR_PK_EM]; tem.y = pk[ZR_PK_EM + 1]; tem.z = pk[ZR_PK_EM + 2]; }
                if (!(cs & 64u)) tms.x = pk[ZR_PK_MS];
            }
        } else {
            zf4 ms[ZR_MATERIAL_SLOTS];
            tex_sample_material(O, lut, u0, v0, s1, t1, s2, t2, ms);
            tb = ms[0]; tme = ms[1]; tro = ms[2]; tno = ms[3]; tao = ms[4]; tem = ms[5]; tms = ms[6];
        }
        const float Rough = __builtin_fmaxf(0.01f, tro.x);
        w_sc = zr_unorm(tem.x, 255.0f) | zr_unorm(tem.y, 255.0f) << 8 | zr_unorm(tem.z, 255.0f) << 16 | zr_unorm(tms.x, 255.0f) << 24;
        w_gB = zr_unorm(tme.x, 255.0f) | zr_unorm(1.0f, 255.0f) << 8 | zr_unorm(Rough, 255.0f) << 16 | 255u << 24;
        w_gC = zr_unorm(tb.x, 255.0f) | zr_unorm(tb.y, 255.0f) << 8 | zr_unorm(tb.z, 255.0f) << 16 | zr_unorm(tao.x, 255.0f) << 24;
        ts = (O->const_slots & 8u) ? zr3(O->ts_const[0], O->ts_const[1], O->ts_const[2]) : zr_tangent_space_normal(zr3(tno.x, tno.y, tno.z));
    }
    const zf3 Nw = compute_normal(pos_dx, pos_dy, s1, t1, s2, t2, N0, ts);
    const zf3 Nn = zr_normalize(Nw);
    const zf3 NP = zr3((Nn.x + 1.0f) / 2.0f, (Nn.y + 1.0f) / 2.0f, (Nn.z + 1.0f) / 2.0f);
    G.depth[p] = depth;
    G.scene_color[p] = w_sc;
    G.gA[p] = zr_unorm(NP.z, 1023.0f) | zr_unorm(NP.y, 1023.0f) << 10 | zr_unorm(NP.x, 1023.0f) << 20 | 3u << 30;
    G.gB[p] = w_gB;
    G.gC[p] = w_gC;
    G.gD[p] = make_uint2(f32_to_f16_hw(P0.x) | f32_to_f16_hw(P0.y) << 16, f32_to_f16_hw(P0.z) | 0x3C000000u);
    return true;
}
