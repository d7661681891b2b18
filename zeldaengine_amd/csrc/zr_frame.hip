// zr_frame.hip — the frame's bookkeeping kernels: k_frame_begin (statistics reset + XkView upload), fills, and the multi-GPU composite
// (k_pack_tiles / k_untile around the all-gather).
#include "zr_dev.h"

// First launch of a frame: zero the frame statistics (all but the sticky overflow latch) and, when the uniforms changed, copy
// XkView from the pinned host ring slot into this frame's device copy.  (The runtime's own hipMemcpyAsync / hipMemsetAsync
// paths cost two extra launches per frame, and the copy path stalls the host for milliseconds the first times it is used.)
// (256-thread workgroups: beside the other lane's persistent kernels a 1024-thread workgroup waits until a whole CU's worth of wave
// slots is free - the launch was seen taking 5 to 40 us at the head of the camera lane)
__global__ __launch_bounds__(256) void k_frame_begin(uint32_t* __restrict__ stats, uint32_t n_stats, const uint32_t* __restrict__ view_src,
                                                     uint32_t* __restrict__ view_dst, uint32_t n_view,
                                                     uint32_t* __restrict__ n_vis_camera, uint32_t rebuild_lists)
{
    const uint32_t i0 = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x == 0) for (uint32_t i = threadIdx.x; i < n_stats; i += 256u) stats[i] = 0u;
    // the passes' work lists (k_cull_instances) stand while camera / light matrices and scene do: only a list about to be rebuilt starts from 0
    // (the shadow pass's length sits in the shadow pipeline's own block and is reset on that pipeline's stream, see shadow_pass)
    if (blockIdx.x == 0 && threadIdx.x == 1u && (rebuild_lists & 2u)) *n_vis_camera = 0u;
    if (view_src) for (uint32_t i = i0; i < n_view; i += gridDim.x * 256u) view_dst[i] = view_src[i];
}

__global__ void k_fill32(uint32_t* __restrict__ p, uint32_t v, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_fill64(unsigned long long* __restrict__ p, unsigned long long v, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// Multi-GPU composite: gathered[rank][slot][TILE_PIX] -> frame; tile_map[t] = owner * slots_per_rank + slot of tile t
__global__ __launch_bounds__(256) void k_untile(const uint32_t* __restrict__ gathered, const uint32_t* __restrict__ tile_map,
                                                uint32_t* __restrict__ frame, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles)
{
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t* src = gathered + (size_t)tile_map[tile] * TILE_PIX;
    const uint32_t tx0 = (tile % tiles_x) * TILE, ty0 = (tile / tiles_x) * TILE;
    for (uint32_t i = threadIdx.x; i < TILE_PIX; i += 256u) {
        const uint32_t px = tx0 + (i & (TILE - 1)), py = ty0 + i / TILE;
        if (px < W && py < H) frame[(size_t)py * W + px] = src[i];
    }
}

// The other direction, for a plane that is NOT written tile-major by its producer (the shadow map): plane -> packed[slot][TILE_PIX] for
// the tiles in `tiles` (slot = place in the list); texels beyond the plane's edge are filled with `pad`.
__global__ __launch_bounds__(256) void k_pack_tiles(const uint32_t* __restrict__ plane, const uint32_t* __restrict__ tiles, uint32_t* __restrict__ packed,
                                                    uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t pad)
{
    const uint32_t tile = tiles[blockIdx.x];
    uint32_t* dst = packed + (size_t)blockIdx.x * TILE_PIX;
    const uint32_t tx0 = (tile % tiles_x) * TILE, ty0 = (tile / tiles_x) * TILE;
    for (uint32_t i = threadIdx.x; i < TILE_PIX; i += 256u) {
        const uint32_t px = tx0 + (i & (TILE - 1)), py = ty0 + i / TILE;
        dst[i] = (px < W && py < H) ? plane[(size_t)py * W + px] : pad;
    }
}

// ------------------------------------------------------------------------------------------------ launchers (C++ linkage, used by zr_host.cpp)

void zr_launch_frame_begin(ZrDevStats* stats, const XkView* view_src_pinned, XkView* view_dst, uint32_t rebuild_lists, hipStream_t s)
{
    static_assert(sizeof(XkView) % 4 == 0 && offsetof(ZrDevStats, overflow_sticky) % 4 == 0, "dword copies");
    hipLaunchKernelGGL(k_frame_begin, dim3(view_src_pinned ? 16 : 1), dim3(256), 0, s, (uint32_t*)stats, (uint32_t)(offsetof(ZrDevStats, overflow_sticky) / 4),
                       (const uint32_t*)view_src_pinned, (uint32_t*)view_dst, (uint32_t)(sizeof(XkView) / 4), &stats->n_vis_work[1], rebuild_lists);
}
void zr_launch_fill32(uint32_t* p, uint32_t v, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_fill32, dim3(1024), dim3(256), 0, s, p, v, n);
}
void zr_launch_fill64(unsigned long long* p, unsigned long long v, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_fill64, dim3(1024), dim3(256), 0, s, p, v, n);
}
void zr_launch_untile(const uint32_t* gathered, const uint32_t* tile_map, uint32_t* frame, uint32_t W, uint32_t H, uint32_t tiles_x,
                      uint32_t n_tiles, hipStream_t s)
{
    hipLaunchKernelGGL(k_untile, dim3(n_tiles), dim3(256), 0, s, gathered, tile_map, frame, W, H, tiles_x, n_tiles);
}
void zr_launch_pack_tiles(const uint32_t* plane, const uint32_t* tiles, uint32_t n_tiles, uint32_t* packed, uint32_t W, uint32_t H, uint32_t tiles_x,
                          uint32_t pad, hipStream_t s)
{
    if (n_tiles) hipLaunchKernelGGL(k_pack_tiles, dim3(n_tiles), dim3(256), 0, s, plane, tiles, packed, W, H, tiles_x, pad);
}
