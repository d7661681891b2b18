// zr_forward.hip — the forward variant of the scene shading (zr_set_shading(ZR_SHADING_FORWARD)): k_forward = Base.frag per covered pixel.
#include "zr_dev.h"
#include "zr_surface.h"
#include "zr_shade.h"

// ------------------------------------------------------------------------------------------------ forward variant
// Base.frag:46-144 - the engine built with ENABLE_DEFERRED_SHADING false (ZE:93): the main render pass clears colour (0,0,0,1) and depth
// (ZE:3517-3519, 2366-2373) and Base.frag shades every fragment that passes LESS straight into the swapchain image (pipelines
// ZE:2749-2801, draws ZE:3544-3680).  Here the winner of the depth test is already known per pixel (the resolve keeps its primitive id in
// G.prim when the context shades forward), so the shader runs once per covered pixel, as with an early depth test and no overdraw.
// Against the deferred pair: the fetched material and ComputeNormal()'s result are used as floats (no render-target format in between),
// AO is not saturated, there is no Mask, every view multiplies FinalColor by ShadowFactor AFTER the gamma (:114-121), and the debug table
// is Base.frag's own (:123-143: base colour without gamma, AmbientOcclution.rgb, the interpolated vertex colour, no GBufferVis).
// The skydome and the background follow in the same render pass (view 0 only, ZE:3681-3699) exactly as in the deferred frame.
template <bool IMAGES>
__global__ __launch_bounds__(256) void k_forward(ZrPass P, ZrLightParams L, const XkView* __restrict__ view, const ZrObject* __restrict__ objs,
                                                 const uint32_t* __restrict__ owned_tiles, GBufferPtrs G, const float* __restrict__ shadowmap,
                                                 CubeDesc C, const float* __restrict__ srgb_lut, const float* __restrict__ unorm_lut,
                                                 uint32_t* __restrict__ out)
{
    __shared__ float tl[512];            // [0, 256) sRGB decode, [256, 512) c / 255: tex_decode's layout
    for (uint32_t i = threadIdx.x; i < 256u; i += 256u) { tl[i] = srgb_lut[i]; tl[256u + i] = unorm_lut[i]; }
    __syncthreads();
    const uint32_t tile_slot = blockIdx.x / (TILE_PIX / 256u), i = threadIdx.x + (blockIdx.x % (TILE_PIX / 256u)) * 256u;
    const uint32_t tile = owned_tiles[tile_slot];
    const int px = (int)(tile % L.tiles_x) * TILE + (int)(i & (TILE - 1)), py = (int)(tile / L.tiles_x) * TILE + (int)(i / TILE);
    if (px >= (int)L.W || py >= (int)L.H) return;
    const size_t p = (size_t)py * L.W + (size_t)px;
    const uint32_t prim = G.prim[p];
    uint32_t rgba = 0xFF000000u;         // clearValues[0].color, ZE:3517
    if (prim != ZR_EMPTY_PRIM) {
        const PixGeom g = pixel_geom(P, objs, prim, px, py);
        const ZrObject* __restrict__ O = g.O;
        // texture(samplerN, fragTexCoord), Base.frag:50-54 (emissive and mask are bound but not fetched)
        zf4 ms[ZR_MATERIAL_SLOTS];
        if (IMAGES) tex_sample_material(O, tl, g.u0, g.v0, g.s1, g.t1, g.s2, g.t2, ms);
        else for (int k = 0; k < ZR_MATERIAL_SLOTS; ++k) { ms[k].x = O->texc[k][0]; ms[k].y = O->texc[k][1]; ms[k].z = O->texc[k][2]; ms[k].w = O->texc[k][3]; }
        const zf3 BaseColor = zr3(ms[0].x, ms[0].y, ms[0].z);
        const float Metallic = zr_saturate(ms[1].x);
        const float Roughness = __builtin_fmaxf(0.01f, zr_saturate(ms[2].x));
        const zf3 ts = (O->const_slots & 8u) ? zr3(O->ts_const[0], O->ts_const[1], O->ts_const[2]) : zr_tangent_space_normal(zr3(ms[3].x, ms[3].y, ms[3].z));
        const zf3 Normal = compute_normal(g.pos_dx, g.pos_dy, g.s1, g.t1, g.s2, g.t2, g.N0, ts);
        const zf3 AmbientOcclution = zr3(ms[4].x, ms[4].y, ms[4].z);
        const zf3 cam = zr3(view->CameraInfo[0], view->CameraInfo[1], view->CameraInfo[2]);
        zf3 Direct, Indirect, RefC; float ShadowFactor;
        shade_surface<false>(L, view, shadowmap, C, tl, nullptr, (uint32_t)view->LightsCount[0], (uint32_t)view->LightsCount[1],
                             (float)(uint32_t)view->LightsCount[3], 1.5f * 1.0f / (float)L.SD, cam,
                             BaseColor, Metallic, Roughness, Normal, AmbientOcclution.x, g.P0, Direct, Indirect, RefC, ShadowFactor);
        zf3 Final = (Direct + Indirect) + RefC;
        Final = zr3(zr_pow(Final.x, 0.4545f), zr_pow(Final.y, 0.4545f), zr_pow(Final.z, 0.4545f));
        zf3 o;
        switch (L.debug_view) {
        case 1: o = BaseColor; break;
        case 2: o = zr3(Metallic, Metallic, Metallic); break;
        case 3: o = zr3(Roughness, Roughness, Roughness); break;
        case 4: o = Normal; break;
        case 5: o = AmbientOcclution; break;
        case 6: {   // fragColor = inColor (Base.vert:28), interpolated like every other varying
            const uint32_t* __restrict__ ix = O->indices + 3u * g.tri;
            const XkVertex* __restrict__ v0 = O->verts + ld_global(ix), * __restrict__ v1 = O->verts + ld_global(ix + 1), * __restrict__ v2 = O->verts + ld_global(ix + 2);
            o = interp3(g.b0, zr3(v0->Color[0], v0->Color[1], v0->Color[2]), zr3(v1->Color[0], v1->Color[1], v1->Color[2]),
                        zr3(v2->Color[0], v2->Color[1], v2->Color[2]));
            break;
        }
        case 7: o = RefC; break;
        case 8: o = zr3(ShadowFactor, ShadowFactor, ShadowFactor); break;
        default: o = Final * ShadowFactor; break;      // cases 0, 9 and default
        }
        rgba = zr_unorm(o.x, 255.0f) | zr_unorm(o.y, 255.0f) << 8 | zr_unorm(o.z, 255.0f) << 16 | 255u << 24;
    }
    if (L.debug_view == 0u) {            // skydome, then the background quad at depth 1 (ZE:3681-3699)
        const uint32_t ov = L.has_overlay ? G.overlay[p] : 0u;
        if (ov) rgba = ov;
        else if (L.bg_enabled && 1.0f <= G.depth[p]) {
            const float u = ((float)px + 0.5f) / (float)L.W, v = ((float)py + 0.5f) / (float)L.H;
            const float one4[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
            const zf4 bgc = tex_sample<2>(L.bg, one4, true, tl, u, v, 1.0f / (float)L.W, 0.0f, 0.0f, 1.0f / (float)L.H);
            rgba = zr_unorm(zr_pow(bgc.x, 0.4545f), 255.0f) | zr_unorm(zr_pow(bgc.y, 0.4545f), 255.0f) << 8 |
                   zr_unorm(zr_pow(bgc.z, 0.4545f), 255.0f) << 16 | 255u << 24;
        }
    }
    if (L.packed_out) out[(size_t)tile_slot * TILE_PIX + i] = rgba;
    else out[p] = rgba;
}

// ------------------------------------------------------------------------------------------------ launcher (C++ linkage, used by zr_host.cpp)

void zr_launch_forward(const ZrPass& P, const ZrLightParams& L, const XkView* view, const ZrObject* objs, const uint32_t* owned_tiles, uint32_t n_owned,
                       const GBufferPtrs& G, const float* shadowmap, const CubeDesc& C, const float* lut, const float* unorm_lut, uint32_t* out, hipStream_t s)
{
    if (n_owned == 0) return;
    if (P.images) hipLaunchKernelGGL((k_forward<true>), dim3(n_owned * (TILE_PIX / 256u)), dim3(256), 0, s, P, L, view, objs, owned_tiles, G, shadowmap, C, lut, unorm_lut, out);
    else hipLaunchKernelGGL((k_forward<false>), dim3(n_owned * (TILE_PIX / 256u)), dim3(256), 0, s, P, L, view, objs, owned_tiles, G, shadowmap, C, lut, unorm_lut, out);
}
