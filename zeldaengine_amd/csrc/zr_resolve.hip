// zr_resolve.hip — k_resolve_gbuffer: BaseScene.frag per pixel from the frame's key buffer into the SoA GBuffer planes (28 B / px,
// formats ZE:2807-2843), coalesced row stores; marks the meshlet-instances that own a pixel (next frame's round 1).
#include "zr_dev.h"
#include "zr_surface.h"

// BaseScene.frag for every pixel of the owned tiles, from the frame's key buffer; resets the keys for the next frame.
// IMAGES 0: no material images in the scene; 1: every material with images has the packed form; 2: per-slot sampling.
// TB = threads per workgroup.  A tile is 256 threads x 4 pixels either way; without images that is one workgroup.  The sampled variants
// run as four independent single-wave workgroups per tile: their waves differ a lot in length (tap counts 1..16 at silhouettes) and
// hold 177+ registers, so a four-wave workgroup that waits for one slot on EVERY SIMD and retires with its slowest wave left the
// SIMDs at 1.46 resident waves of the 2 that fit.
#ifndef ZR_RESOLVE_IMG_WAVES
#define ZR_RESOLVE_IMG_WAVES 3
#endif
// PPT = pixels per thread (ZR_PIXELS_PER_THREAD; the note above k_lighting says why it is 1).
template <int IMAGES, int TB, int PPT>
__global__ __launch_bounds__(TB, TB == 64 ? ZR_RESOLVE_IMG_WAVES : 1) void k_resolve_gbuffer(ZrPass P, const ZrObject* __restrict__ objs,
                                                        const uint32_t* __restrict__ owned_tiles,
                                                        unsigned long long* __restrict__ vis64, GBufferPtrs G,
                                                        const float* __restrict__ srgb_lut, const float* __restrict__ unorm_lut,
                                                        uint8_t* __restrict__ vis_now, ZrDevStats* __restrict__ stats, uint32_t vis_mark)
{
    __shared__ uint32_t covered_s;
    __shared__ float tlut[IMAGES ? 512 : 1];       // texel decode tables of the sampler (see tex_decode)
    constexpr uint32_t T = TILE_PIX / (uint32_t)PPT, PARTS = T / (uint32_t)TB;          // threads / workgroups per tile
    const uint32_t tid = threadIdx.x + (blockIdx.x % PARTS) * (uint32_t)TB;             // the thread's place among the tile's T
    if (IMAGES) for (uint32_t i = threadIdx.x; i < 256u; i += (uint32_t)TB) { tlut[i] = srgb_lut[i]; tlut[256u + i] = unorm_lut[i]; }   // (the barrier below orders it)
    const float* __restrict__ dlut = IMAGES ? tlut : srgb_lut;
    const uint32_t tile = owned_tiles[blockIdx.x / PARTS];
    const int tx0 = (int)(tile % P.tiles_x) * TILE, ty0 = (int)(tile / P.tiles_x) * TILE;
    if (threadIdx.x == 0) covered_s = 0;
    __syncthreads();
    uint32_t ncov = 0;
    // row-major within the tile -> 128 B (256 B for GBufferD / keys) contiguous row segments per wave.  The thread's four keys are fetched
    // (and reset) together: four independent loads in flight instead of one at the head of each pixel's chain of dependent loads.
    unsigned long long keys[PPT];
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)PPT; ++q) {
        const uint32_t i = tid + q * T;
        const int px = tx0 + (int)(i & (TILE - 1)), py = ty0 + (int)(i / TILE);
        keys[q] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
        if (px < (int)P.W && py < (int)P.H) {
            const size_t p = (size_t)py * P.W + (size_t)px;
            keys[q] = vis64[p];
            vis64[p] = (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM;
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)PPT; ++q) {
        const uint32_t i = tid + q * T;
        const int px = tx0 + (int)(i & (TILE - 1)), py = ty0 + (int)(i / TILE);
        if (px >= (int)P.W || py >= (int)P.H) continue;
        const unsigned long long k = keys[q];
        ncov += resolve_pixel<IMAGES>(P, objs, (uint32_t)k, zr_u2f((uint32_t)(k >> 32)), px, py, G, dlut, vis_now, vis_mark) ? 1u : 0u;
        if (G.prim) G.prim[(size_t)py * P.W + (size_t)px] = (uint32_t)k;      // forward variant (k_forward): the depth test's winner
        if (IMAGES != 0 && P.sky_keys != nullptr) {
            // The skydome (ZE:3681-3691: drawn last, depth test LESS against the scene's depth, colour only).  Its triangles were
            // resolved among themselves into a key plane of their own; the dome shows where that depth is less than the scene's.
            // (The scene pixel above cleared the overlay word; the GBuffer keeps what the scene pass wrote, hidden or not.)
            const unsigned long long ks = P.sky_keys[(size_t)py * P.W + (size_t)px];
            if ((uint32_t)ks != ZR_EMPTY_PRIM && zr_u2f((uint32_t)(ks >> 32)) < zr_u2f((uint32_t)(k >> 32)))
                resolve_pixel<IMAGES>(P, objs, objs[P.sky_object].prim_base + (uint32_t)ks, zr_u2f((uint32_t)(ks >> 32)), px, py, G, dlut, nullptr);
        }
    }
    if (ncov) atomicAdd(&covered_s, ncov);
    __syncthreads();
    // (one add per workgroup; spread over 32 words - 32 000 single-wave workgroups on ONE address would queue for 10 ns apiece)
    if (threadIdx.x == 0 && covered_s) atomicAdd(&stats->covered_part[blockIdx.x & 31u], covered_s);
}

// ------------------------------------------------------------------------------------------------ launcher (C++ linkage, used by zr_host.cpp)

void zr_launch_resolve_gbuffer(const ZrPass& P, const ZrObject* objs, const uint32_t* owned_tiles, uint32_t n_owned,
                               unsigned long long* vis64, const GBufferPtrs& G, const float* srgb_lut, const float* unorm_lut, uint8_t* vis_now,
                               ZrDevStats* stats, hipStream_t s, uint32_t vis_mark)
{
    if (n_owned == 0) return;
    // one pixel per thread (ZR_PIXELS_PER_THREAD): see the note above k_lighting
#define ZR_LAUNCH_RESOLVE(IM, TB, PPT) hipLaunchKernelGGL((k_resolve_gbuffer<IM, TB, PPT>), dim3(n_owned * (TILE_PIX / (PPT) / (TB))), dim3(TB), 0, s, P, objs, owned_tiles, vis64, G, srgb_lut, unorm_lut, vis_now, stats, vis_mark)
    if (P.images == 1u) ZR_LAUNCH_RESOLVE(1, 64, ZR_PIXELS_PER_THREAD);
    else if (P.images) ZR_LAUNCH_RESOLVE(2, 64, ZR_PIXELS_PER_THREAD);
    else ZR_LAUNCH_RESOLVE(0, ZR_RESOLVE_TB, ZR_PIXELS_PER_THREAD);
#undef ZR_LAUNCH_RESOLVE
}
