// zr_lighting.hip — the deferred-lighting pass (ZE:3531-3540): k_lighting = BaseLighting.frag per pixel (PCF 5x5, per-tile light list,
// ambient, cubemap IBL, gamma, debug views 0-8), k_gbuffer_vis = the GBufferVis mosaic of view 9.
#include "zr_dev.h"
#include "zr_surface.h"
#include "zr_shade.h"

// BaseLighting.frag:147-254 for every pixel of the owned tiles (the full-screen quad of ZE:3531-3540)
template <bool LIGHT_LIST, bool BACKGROUND, int TB, int PPT>      // PPT: pixels per thread, 4 or 1 (as in k_resolve_gbuffer)
// (compiled for exactly ZR_LIGHT_WAVES = 5 waves per SIMD: 96 VGPRs and 24 bytes of scratch.  Left to itself the allocator takes 127 registers.
// Rounds 4 - 5 ran it at 4 waves (97 VGPRs): beside a camera lane that filled the whole period, a fifth wave cost that lane more than it gave
// this pass.  Since the camera lane lost its scans and index passes (round 6) the frame's period is THIS lane's, and the fifth wave pays:
// 4 / 5 / 6 waves: 5 548 / 5 670 / 5 420 Mpixel/s)
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(ZR_LIGHT_WAVES, ZR_LIGHT_WAVES))) void k_lighting(ZrLightParams L, const XkView* __restrict__ view,
                                                  const uint32_t* __restrict__ owned_tiles, GBufferPtrs G,
                                                  const float* __restrict__ shadowmap, CubeDesc C,
                                                  const float* __restrict__ srgb_lut, const float* __restrict__ unorm_lut,
                                                  uint32_t* __restrict__ out)
{
    // UNORM loads are IEEE quotients c / 255 and c / 1023: 14 per pixel, served from an LDS copy of the host-built table
    __shared__ float tl[512];            // [0, 256) sRGB decode (24 cubemap fetches per pixel), [256, 512) c / 255: tex_decode's layout
    __shared__ float u10[1024];
    float* const slut = tl; float* const u8 = tl + 256;
    constexpr uint32_t T = TILE_PIX / (uint32_t)PPT, PARTS = T / (uint32_t)TB, WAVES = (uint32_t)TB / 64u;
    const uint32_t tid = threadIdx.x + (blockIdx.x % PARTS) * (uint32_t)TB;      // the thread's place among the tile's T
    const uint32_t tile_slot = blockIdx.x / PARTS;
    for (uint32_t i = threadIdx.x; i < 256u; i += (uint32_t)TB) { u8[i] = unorm_lut[i]; slut[i] = srgb_lut[i]; }
    for (uint32_t i = threadIdx.x; i < 1024u; i += (uint32_t)TB) u10[i] = unorm_lut[256u + i];
    __syncthreads();
    if (L.clear_next) {      // the clear of the next frame's shadow pass (depth 1.0, ZE:3248), a slice per workgroup: saves a launch
        const uint32_t per = (L.clear_n + gridDim.x - 1u) / gridDim.x, b = blockIdx.x * per;
        for (uint32_t i = threadIdx.x; i < per && b + i < L.clear_n; i += (uint32_t)TB) L.clear_next[b + i] = 0x3F800000u;
    }
    const uint32_t tile = owned_tiles[tile_slot];
    const int tx0 = (int)(tile % L.tiles_x) * TILE, ty0 = (int)(tile / L.tiles_x) * TILE;
    const zf3 cam = zr3(view->CameraInfo[0], view->CameraInfo[1], view->CameraInfo[2]);
    const uint32_t nDir = (uint32_t)view->LightsCount[0], nPoint = (uint32_t)view->LightsCount[1];
    const float maxmips = (float)(uint32_t)view->LightsCount[3];
    const float dxy = 1.5f * 1.0f / (float)L.SD;

    // Tile light list (LIGHT_LIST: four or more point lights): a light whose sphere of influence misses the bounding box of the tile's world
    // positions would be skipped by every pixel's own exact test below (|lp - P| >= the box distance per axis, and the squared
    // sums are monotonic), so it is dropped for the whole tile.  The list is a bitmask, walked in ascending order: the
    // accumulation order over lights is unchanged.  Pixels with Mask = 0 do not count: their colour is (...) * 0 -> stored 0.
    __shared__ float bbp[LIGHT_LIST ? WAVES : 1][6];
    __shared__ uint32_t lmask[LIGHT_LIST ? XK_MAX_POINT_LIGHTS_NUM / 32 : 1];
    constexpr bool use_mask = LIGHT_LIST;
    if constexpr (LIGHT_LIST) {
        float lo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, hi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
        bool odd = false;                    // a non-finite position: keep every light
        for (uint32_t i = tid; i < TILE_PIX; i += T) {
            const int px = tx0 + (int)(i & (TILE - 1)), py = ty0 + (int)(i / TILE);
            if (px >= (int)L.W || py >= (int)L.H) continue;
            const size_t p = (size_t)py * L.W + (size_t)px;
            if ((G.scene_color[p] >> 24) == 0u) continue;
            const uint2 D = G.gD[p];
            const float q[3] = { f16_to_f32_hw(D.x & 0xFFFFu), f16_to_f32_hw(D.x >> 16), f16_to_f32_hw(D.y & 0xFFFFu) };
            for (int a = 0; a < 3; ++a) {
                if (!(__builtin_fabsf(q[a]) <= 3.402823466e38f)) odd = true;
                lo[a] = __builtin_fminf(lo[a], q[a]); hi[a] = __builtin_fmaxf(hi[a], q[a]);
            }
        }
        for (int a = 0; a < 3; ++a) { lo[a] = wave_fmin(lo[a]); hi[a] = wave_fmax(hi[a]); }
        const bool wodd = __ballot(odd) != 0ull;
        if ((threadIdx.x & 63u) == 0u) {
            float* o = bbp[threadIdx.x >> 6];
            o[0] = wodd ? -__builtin_inff() : lo[0]; o[1] = wodd ? -__builtin_inff() : lo[1]; o[2] = wodd ? -__builtin_inff() : lo[2];
            o[3] = wodd ? __builtin_inff() : hi[0]; o[4] = wodd ? __builtin_inff() : hi[1]; o[5] = wodd ? __builtin_inff() : hi[2];
        }
        for (uint32_t i = threadIdx.x; i < XK_MAX_POINT_LIGHTS_NUM / 32; i += (uint32_t)TB) lmask[i] = 0u;
        __syncthreads();
        float blo[3], bhi[3];
        for (int a = 0; a < 3; ++a) {
            blo[a] = bbp[0][a]; bhi[a] = bbp[0][3 + a];
            for (uint32_t w = 1; w < WAVES; ++w) { blo[a] = __builtin_fminf(blo[a], bbp[w][a]); bhi[a] = __builtin_fmaxf(bhi[a], bbp[w][3 + a]); }
        }
        for (uint32_t li = threadIdx.x; li < nPoint; li += (uint32_t)TB) {
            const XkLight* __restrict__ Lt = &view->PointLights[li];
            const bool lfinite = __builtin_fabsf(Lt->Color[0]) <= 3.402823466e38f && __builtin_fabsf(Lt->Color[1]) <= 3.402823466e38f &&
                                 __builtin_fabsf(Lt->Color[2]) <= 3.402823466e38f && __builtin_fabsf(Lt->Color[3]) <= 3.402823466e38f;
            const float falloff = Lt->Direction[3];
            bool keep = true;
            if (lfinite && falloff > 0.0f) {
                zf3 e;      // per axis: how far the light lies outside the box (0 inside); |lp - P| is at least that for every P in it
                e.x = __builtin_fmaxf(0.0f, __builtin_fmaxf(blo[0] - Lt->Position[0], Lt->Position[0] - bhi[0]));
                e.y = __builtin_fmaxf(0.0f, __builtin_fmaxf(blo[1] - Lt->Position[1], Lt->Position[1] - bhi[1]));
                e.z = __builtin_fmaxf(0.0f, __builtin_fmaxf(blo[2] - Lt->Position[2], Lt->Position[2] - bhi[2]));
                if (zr_dot(e, e) > (falloff * falloff) * 1.000001f) keep = false;
            }
            if (keep) atomicOr(&lmask[li >> 5], 1u << (li & 31u));
        }
        __syncthreads();
    }

    for (uint32_t i = tid; i < TILE_PIX; i += T) {
        const int px = tx0 + (int)(i & (TILE - 1)), py = ty0 + (int)(i / TILE);
        if (px >= (int)L.W || py >= (int)L.H) continue;
        const size_t p = (size_t)py * L.W + (size_t)px;
        const uint32_t sc = G.scene_color[p], A = G.gA[p], B = G.gB[p], Cc = G.gC[p];
        const uint2 D = G.gD[p];
        // what every path ends with: the skydome / background drawn over the lit quad in view 0 (ZE:3681-3699), then the store
        auto emit = [&](uint32_t rgba) {
            if (L.debug_view == 0u) {
                const uint32_t ov = L.has_overlay ? G.overlay[p] : 0u;
                if (ov) rgba = ov;
                else if (BACKGROUND && L.bg_enabled && 1.0f <= G.depth[p]) {
                    const float u = ((float)px + 0.5f) / (float)L.W, v = ((float)py + 0.5f) / (float)L.H;
                    const float one4[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
                    const zf4 bgc = tex_sample<2>(L.bg, one4, true, tl, u, v, 1.0f / (float)L.W, 0.0f, 0.0f, 1.0f / (float)L.H);
                    rgba = zr_unorm(zr_pow(bgc.x, 0.4545f), 255.0f) | zr_unorm(zr_pow(bgc.y, 0.4545f), 255.0f) << 8 |
                           zr_unorm(zr_pow(bgc.z, 0.4545f), 255.0f) << 16 | 255u << 24;
                }
            }
            if (L.packed_out) out[(size_t)tile_slot * TILE_PIX + i] = rgba;
            else out[p] = rgba;
        };
        // A pixel nothing was drawn to holds the clear values of every target (ZE:3427-3433), so the shader computes the same
        // colour for all of them: it was computed once (zr_launch_lighting's one-pixel pre-launch of this very kernel).
        if (L.empty_rgba != nullptr && sc == 0xFF000000u && A == 0u && B == 0xFF000000u && Cc == 0xFF000000u && D.x == 0u && D.y == 0x3C000000u) {
            emit(*L.empty_rgba);
            continue;
        }
        const zf3 BaseColor = zr3(u8[Cc & 255u], u8[(Cc >> 8) & 255u], u8[(Cc >> 16) & 255u]);
        const float Metallic = zr_saturate(u8[B & 255u]);
        float Roughness = zr_saturate(u8[(B >> 16) & 255u]);
        const zf3 Normal = zr3(__builtin_fmaf(u10[(A >> 20) & 1023u], 2.0f, -1.0f), __builtin_fmaf(u10[(A >> 10) & 1023u], 2.0f, -1.0f),
                               __builtin_fmaf(u10[A & 1023u], 2.0f, -1.0f));
        const float AO = zr_saturate(u8[Cc >> 24]);
        const float Mask = u8[sc >> 24];
        Roughness = __builtin_fmaxf(0.01f, Roughness);
        const zf3 N = zr_normalize(Normal);
        const zf3 Pw = zr3(f16_to_f32_hw(D.x & 0xFFFFu), f16_to_f32_hw(D.x >> 16), f16_to_f32_hw(D.y & 0xFFFFu));
        zf3 Direct, Indirect, RefC; float ShadowFactor;
        shade_surface<use_mask>(L, view, shadowmap, C, slut, lmask, nDir, nPoint, maxmips, dxy, cam, BaseColor, Metallic, Roughness, N, AO, Pw,
                                Direct, Indirect, RefC, ShadowFactor);

        zf3 Final = ((Direct + Indirect) + RefC) * Mask;
        Final = zr3(zr_pow(Final.x, 0.4545f), zr_pow(Final.y, 0.4545f), zr_pow(Final.z, 0.4545f));
        zf3 o;
        switch (L.debug_view) {
        case 0: o = Final; break;
        case 1: o = zr3(zr_pow(BaseColor.x, 0.4545f), zr_pow(BaseColor.y, 0.4545f), zr_pow(BaseColor.z, 0.4545f)); break;
        case 2: o = zr3(Metallic, Metallic, Metallic); break;
        case 3: o = zr3(Roughness, Roughness, Roughness); break;
        case 4: o = Normal; break;
        case 5: o = zr3(AO, AO, AO); break;
        case 7: o = RefC; break;
        case 8: o = zr3(ShadowFactor, ShadowFactor, ShadowFactor); break;
        case 9: o = Final; break;       // GBufferVis: k_gbuffer_vis then overwrites the eight mosaic cells
        case 6: {   // fragColor of the full-screen quad: Background.vert:10-17 vertex colours over its two triangles
            const float u = ((float)px + 0.5f) / (float)L.W, v = ((float)py + 0.5f) / (float)L.H;
            o = v >= u ? zr3(1.0f - v, u, v - u) : zr3(1.0f - u, v, u - v);
            break;
        }
        default: o = Final * ShadowFactor; break;
        }
        emit(zr_unorm(o.x, 255.0f) | zr_unorm(o.y, 255.0f) << 8 | zr_unorm(o.z, 255.0f) << 16 | 255u << 24);
    }
}

// GBufferVis (SH/BaseLighting.frag:42-145, SPEC_CONSTANTS 9).  Runs after k_lighting has written FinalColor everywhere: the
// lighting quad samples every GBuffer target again at UV = fragTexCoord * 3 / (1 - EmptyRatio) through LINEAR / REPEAT samplers
// (ZE:2811-2847) and shows a 3 x 3 mosaic; the centre cell and whatever lies outside the cells keep FinalColor.  Bilinear
// weights are snapped to 8 fractional bits (a stated choice, like sampling hardware): with EmptyRatio = 0 every sample is
// exactly texel (3x + 1, 3y + 1) mod (W, H).
struct GTexel { float v[20]; };     // SceneColor, GBufferA, B, C, D as five vec4
__device__ __forceinline__ GTexel gbuffer_texel(const GBufferPtrs& G, uint32_t W, int x, int y)
{
    const size_t p = (size_t)y * W + (size_t)x;
    const uint32_t sc = G.scene_color[p], A = G.gA[p], B = G.gB[p], C = G.gC[p];
    const uint2 D = G.gD[p];
    GTexel t;
    for (int k = 0; k < 4; ++k) {
        t.v[k] = (float)((sc >> (8 * k)) & 255u) / 255.0f;
        t.v[8 + k] = (float)((B >> (8 * k)) & 255u) / 255.0f;
        t.v[12 + k] = (float)((C >> (8 * k)) & 255u) / 255.0f;
    }
    t.v[4] = (float)((A >> 20) & 1023u) / 1023.0f; t.v[5] = (float)((A >> 10) & 1023u) / 1023.0f;
    t.v[6] = (float)(A & 1023u) / 1023.0f; t.v[7] = (float)(A >> 30) / 3.0f;
    t.v[16] = zr_f16_to_f32(D.x & 0xFFFFu); t.v[17] = zr_f16_to_f32(D.x >> 16);
    t.v[18] = zr_f16_to_f32(D.y & 0xFFFFu); t.v[19] = zr_f16_to_f32(D.y >> 16);
    return t;
}
__device__ __forceinline__ int wrap_index(int i, int n) { const int m = i % n; return m < 0 ? m + n : m; }

__global__ __launch_bounds__(256) void k_gbuffer_vis(ZrLightParams L, const XkView* __restrict__ view, GBufferPtrs G,
                                                     const float* __restrict__ shadowmap, CubeDesc C,
                                                     const float* __restrict__ srgb_lut, uint32_t* __restrict__ out)
{
    const uint32_t px = blockIdx.x * 16u + (threadIdx.x & 15u), py = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (px >= L.W || py >= L.H) return;
    const float ERx = view->ViewportInfo[2] / view->ViewportInfo[0], ERy = view->ViewportInfo[3] / view->ViewportInfo[1];
    const float tx = ((float)px + 0.5f) / (float)L.W, ty = ((float)py + 0.5f) / (float)L.H;        // fragTexCoord
    const float UVx = (tx * 3.0f) / (1.0f - ERx), UVy = (ty * 3.0f) / (1.0f - ERy);
    const float Sx = (1.0f - ERx) / 3.0f, Sy = (1.0f - ERy) / 3.0f;                                 // Step
    int cell = -1; float bx = 0.0f, by = 0.0f;
    if (tx < Sx && ty < Sy) { cell = 0; bx = 1.0f; by = 1.0f; }
    else if (tx < Sx * 2.0f && ty < Sy) { cell = 1; bx = 2.0f; by = 1.0f; }
    else if (tx < Sx * 3.0f && ty < Sy) { cell = 2; bx = 3.0f; by = 1.0f; }
    else if (tx < Sx && ty < Sy * 2.0f) { cell = 3; bx = 1.0f; by = 2.0f; }
    else if (tx < 1.0f && ty < Sy * 2.0f && tx > Sx * 2.0f) { cell = 4; bx = 3.0f; by = 2.0f; }
    else if (tx < Sx && ty < Sx * 3.0f) { cell = 5; bx = 1.0f; by = 3.0f; }                         // Step.x * 3: as the shader has it
    else if (tx < Sx * 2.0f && tx > Sx && ty < Sy * 3.0f && ty > Sy * 2.0f) { cell = 6; bx = 2.0f; by = 3.0f; }
    else if (tx < Sx * 3.0f && tx > Sx * 2.0f && ty < Sy * 3.0f && ty > Sy * 2.0f) { cell = 7; bx = 3.0f; by = 3.0f; }
    if (cell < 0) return;                                                                           // FinalColor stays
    zf3 o;
    if (tx > Sx * (bx - ERx) || ty > Sy * (by - ERy)) o = zr3(1.0f, 1.0f, 1.0f);                      // the cells' white frames
    else {
        // texture(sampler2D, UV): one mip level, LINEAR, REPEAT
        float x = __builtin_fmaf(UVx, (float)L.W, -0.5f), y = __builtin_fmaf(UVy, (float)L.H, -0.5f);
        if (!(__builtin_fabsf(x) < 1.0e9f)) x = 0.0f;
        if (!(__builtin_fabsf(y) < 1.0e9f)) y = 0.0f;
        const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
        const float ax = __builtin_floorf(__builtin_fmaf(x - fx, 256.0f, 0.5f)) / 256.0f;
        const float ay = __builtin_floorf(__builtin_fmaf(y - fy, 256.0f, 0.5f)) / 256.0f;
        const int x0 = wrap_index((int)fx, (int)L.W), x1 = wrap_index((int)fx + 1, (int)L.W);
        const int y0 = wrap_index((int)fy, (int)L.H), y1 = wrap_index((int)fy + 1, (int)L.H);
        const GTexel t00 = gbuffer_texel(G, L.W, x0, y0), t10 = gbuffer_texel(G, L.W, x1, y0);
        const GTexel t01 = gbuffer_texel(G, L.W, x0, y1), t11 = gbuffer_texel(G, L.W, x1, y1);
        float g[20];
        for (int k = 0; k < 20; ++k) {
            const float top = __builtin_fmaf(ax, t10.v[k] - t00.v[k], t00.v[k]), bot = __builtin_fmaf(ax, t11.v[k] - t01.v[k], t01.v[k]);
            g[k] = __builtin_fmaf(ay, bot - top, top);
        }
        const zf3 BaseColor = zr3(g[12], g[13], g[14]);
        const float Metallic = zr_saturate(g[8]);
        const float Roughness = __builtin_fmaxf(0.01f, zr_saturate(g[10]));
        const zf3 Normal = zr3(__builtin_fmaf(g[4], 2.0f, -1.0f), __builtin_fmaf(g[5], 2.0f, -1.0f), __builtin_fmaf(g[6], 2.0f, -1.0f));
        const float AO = zr_saturate(g[15]);
        const zf3 N = zr_normalize(Normal);
        const zf3 Pw = zr3(g[16], g[17], g[18]);
        switch (cell) {
        case 0: o = zr3(zr_pow(BaseColor.x, 0.4545f), zr_pow(BaseColor.y, 0.4545f), zr_pow(BaseColor.z, 0.4545f)); break;
        case 1: o = zr3(Metallic, Metallic, Metallic); break;
        case 2: o = zr3(Roughness, Roughness, Roughness); break;
        case 3: o = N; break;
        case 4: o = zr3(AO, AO, AO); break;
        case 5: o = zr3(0.0f, 0.0f, 0.0f); break;
        case 6: {
            const zf3 cam = zr3(view->CameraInfo[0], view->CameraInfo[1], view->CameraInfo[2]);
            const zf3 Vv = zr_normalize(cam - Pw), Nn = zr_normalize(N);
            const float eta = 1.00f / 1.52f;
            const float dNI = zr_dot(Nn, Vv);
            const float kk = __builtin_fmaf(-(eta * eta), __builtin_fmaf(-dNI, dNI, 1.0f), 1.0f);
            zf3 R;
            if (kk < 0.0f) R = zr3(0.0f, 0.0f, 0.0f);
            else {
                const float q = __builtin_fmaf(eta, dNI, __builtin_sqrtf(kk));
                R = zr3(__builtin_fmaf(eta, Vv.x, -(q * Nn.x)), __builtin_fmaf(eta, Vv.y, -(q * Nn.y)), __builtin_fmaf(eta, Vv.z, -(q * Nn.z)));
            }
            o = cube_sample(C, srgb_lut, L.cube_dim, (int)L.cube_levels, R, 0.0f) * 10.0f;
            break;
        }
        default: {
            const zf4 s4 = zr_mat4_point(L.SB, Pw);
            const float sx = s4.x / s4.w, sy = s4.y / s4.w, sz = s4.z / s4.w, sw = s4.w / s4.w;
            const float dxy = 1.5f * 1.0f / (float)L.SD;
            float sum = 0.0f;
            for (int xo = -2; xo <= 2; ++xo)
                for (int yo = -2; yo <= 2; ++yo) sum += shadow_tap(shadowmap, (int)L.SD, sx, sy, sz, sw, dxy * (float)xo, dxy * (float)yo);
            const float sf = sum * 0.04f;
            o = zr3(sf, sf, sf);
            break;
        }
        }
    }
    out[(size_t)py * L.W + px] = zr_unorm(o.x, 255.0f) | zr_unorm(o.y, 255.0f) << 8 | zr_unorm(o.z, 255.0f) << 16 | 255u << 24;
}

// ------------------------------------------------------------------------------------------------ launchers (C++ linkage, used by zr_host.cpp)

void zr_launch_lighting(const ZrLightParams& L, const XkView* view, const uint32_t* owned_tiles, uint32_t n_owned,
                        const GBufferPtrs& G, const float* shadowmap, const CubeDesc& C, const float* lut, const float* unorm_lut,
                        uint32_t* out, hipStream_t s)
{
    if (n_owned == 0) return;
    // with several point lights each tile first builds its light list (L.light_list: decided on the host from the light count)
#define ZR_LAUNCH_LIGHTING(LL, BG) hipLaunchKernelGGL((k_lighting<LL, BG, ZR_LIGHT_TB, ZR_PIXELS_PER_THREAD>), dim3(n_owned * (TILE_PIX / ZR_PIXELS_PER_THREAD / ZR_LIGHT_TB)), dim3(ZR_LIGHT_TB), 0, s, L, view, owned_tiles, G, shadowmap, C, lut, unorm_lut, out)
    if (L.light_list) { if (L.bg_enabled) ZR_LAUNCH_LIGHTING(true, true); else ZR_LAUNCH_LIGHTING(true, false); }
    else { if (L.bg_enabled) ZR_LAUNCH_LIGHTING(false, true); else ZR_LAUNCH_LIGHTING(false, false); }
#undef ZR_LAUNCH_LIGHTING
}
void zr_launch_gbuffer_vis(const ZrLightParams& L, const XkView* view, const GBufferPtrs& G, const float* shadowmap, const CubeDesc& C,
                           const float* lut, uint32_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_gbuffer_vis, dim3((L.W + 15) / 16, (L.H + 15) / 16), dim3(256), 0, s, L, view, G, shadowmap, C, lut, out);
}
