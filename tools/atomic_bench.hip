// Micro-benchmark: what do scattered 64-bit atomicMin operations on a frame-sized key plane cost on this chip?
// (the question behind "small triangles straight from k_geom into vis64": 1.5 M triangle records of ~1 pixel each)
//   hipcc --offload-arch=gfx950 -O3 -o tools/atomic_bench tools/atomic_bench.hip && tools/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// a wave = one meshlet: its lanes' triangles fall into a box x box pixel window at a random place of a W x H plane; `per` pixels per lane
template <int MODE>     // 0: test + atomicMin, 1: atomicMin always, 2: plain store, 3: test only (load)
__global__ __launch_bounds__(256) void k_scatter(unsigned long long* __restrict__ plane, uint32_t W, uint32_t H, uint32_t n_waves, uint32_t box, uint32_t per, uint32_t seed, uint32_t* sink)
{
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (wave >= n_waves) return;
    const uint32_t h = hash(wave * 2654435761u + seed);
    const uint32_t ox = h % (W - box), oy = (h >> 12) % (H - box);
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t g = hash(h + lane * 97u + i * 7919u);
        const uint32_t x = ox + g % box, y = oy + (g >> 8) % box;
        const unsigned long long k = (unsigned long long)(0x3F000000u + (g >> 10)) << 32 | (wave * 64u + lane);
        unsigned long long* p = &plane[(size_t)y * W + x];
        if (MODE == 0) { if (k < *p) atomicMin(p, k); }
        else if (MODE == 1) atomicMin(p, k);
        else if (MODE == 2) *p = k;
        else acc += (uint32_t)(*p >> 32);
    }
    if (MODE == 3 && acc == 0x12345u) *sink = acc;
}

int main()
{
    const uint32_t W = 1920, H = 1080;
    unsigned long long* plane; uint32_t* sink;
    CHK(hipMalloc(&plane, (size_t)W * H * 8)); CHK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char* names[4] = { "test+atomicMin", "atomicMin", "store", "load" };
    for (uint32_t box : { 8u, 12u, 24u })
    for (uint32_t n_waves : { 24000u, 48000u })       // x 64 lanes = 1.5 M / 3 M triangles
    for (uint32_t per : { 1u, 2u })
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CHK(hipMemset(plane, 0x3F, (size_t)W * H * 8));       // "far" keys: every first touch wins
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, 0));
            const dim3 g((n_waves * 64u + 255u) / 256u), b(256);
            if (mode == 0) hipLaunchKernelGGL(k_scatter<0>, g, b, 0, 0, plane, W, H, n_waves, box, per, 17u + rep, sink);
            if (mode == 1) hipLaunchKernelGGL(k_scatter<1>, g, b, 0, 0, plane, W, H, n_waves, box, per, 17u + rep, sink);
            if (mode == 2) hipLaunchKernelGGL(k_scatter<2>, g, b, 0, 0, plane, W, H, n_waves, box, per, 17u + rep, sink);
            if (mode == 3) hipLaunchKernelGGL(k_scatter<3>, g, b, 0, 0, plane, W, H, n_waves, box, per, 17u + rep, sink);
            CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("box %2u  waves %5u  px/lane %u  %-15s %8.1f us  (%.2f G lane-ops/s)\n", box, n_waves, per, names[mode], best * 1e3f, n_waves * 64.0 * per / (best * 1e-3) / 1e9);
    }
    return 0;
}
