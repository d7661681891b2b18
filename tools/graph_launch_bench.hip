#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float a[140]; };
__global__ void k(Big b, float* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += b.a[3]; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    float* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Big b; for (int i = 0; i < 140; ++i) b.a[i] = 1.0f;
    const int NK = 14, IT = 2000;
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, b, d);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < IT; ++it) { for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, b, d); if ((it & 31) == 31) CK(hipStreamSynchronize(s)); }
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    printf("direct: %.2f us host per %d launches (%.2f each); wall %.2f us per batch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / IT, NK,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / IT / NK, std::chrono::duration<double, std::micro>(t2 - t0).count() / IT);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, b, d);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < IT; ++it) { CK(hipGraphLaunch(ge, s)); if ((it & 31) == 31) CK(hipStreamSynchronize(s)); }
    t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    t2 = std::chrono::steady_clock::now();
    printf("graph : %.2f us host per graph of %d kernels; wall %.2f us per graph\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / IT, NK,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / IT);
    return 0;
}
