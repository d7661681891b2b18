// valu_calib.hip — what does one SIMD of an MI355X (gfx950) issue per cycle?  Calibrates the vector-ALU peak bench.py quotes.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/valu_calib tools/valu_calib.hip && tools/valu_calib > profiles/<tag>_valu_calib.json
//
// Every wave runs ITERS x UNROLL instructions of one kind on CHAINS independent registers (inline asm, so the compiler can neither
// fuse nor drop them), 1 / 2 / 4 / 8 waves per SIMD (256-thread workgroups, one wave per SIMD each, 1 / 2 / 4 / 8 workgroups per CU on a
// grid of 256 x that many workgroups).  Reported per (instruction, waves per SIMD):
//   cyc_per_inst_wave        shader cycles (s_memtime) between two instructions of ONE wave
//   cyc_per_inst_simd_wall   shader cycles a SIMD spends per wave64 instruction, from the first wave's start to the last wave's end
//                            (s_memrealtime) - what the chip sustains; resident_waves_per_simd says how many waves really ran side by side
//   cyc_per_inst_simd        the same from wave cycles / (instructions x nominal waves per SIMD): equal when all waves are co-resident
//   ginst_per_s              wave64 instructions per second over the chip, over that span
//   clock_ghz           s_memtime ticks / s_memrealtime ticks x 100 MHz inside the kernel
// "div" / "sqrt" are the IEEE sequences the compiler emits for a / b and sqrtf under -fno-fast-math (what zr_math.h's / and
// zr_sqrt compile to); their figure is cycles per DIVISION, not per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { OP_FMA, OP_FMA_1V, OP_MUL_F32, OP_ADD_U32, OP_MUL_LO, OP_MAD_U64, OP_RCP, OP_SQRT_HW, OP_DIV, OP_SQRT, OP_FMA_DEP, OP_CNDMASK, OP_CVT_I2F, OP_LDS_MIN64, OP_PK_FMA, OP_PK_MUL, OP_COUNT };
static const char* kOpName[OP_COUNT] = { "v_fma_f32 (3 VGPR sources)", "v_fma_f32 (1 VGPR source)", "v_mul_f32", "v_add_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_rcp_f32", "v_sqrt_f32", "ieee_div_f32",
                                         "ieee_sqrt_f32", "v_fma_f32_dependent", "v_cndmask_b32", "v_cvt_f32_i32", "ds_min_u64",
                                         "v_pk_fma_f32 (two fp32 lanes per instruction)", "v_pk_mul_f32 (two fp32 lanes per instruction)" };

#define UNROLL 8

template <int OP>
__global__ __launch_bounds__(256) void k_calib(float* __restrict__ out, unsigned long long* __restrict__ cyc, unsigned long long* __restrict__ rt,
                                              int iters, float seed)
{
    __shared__ unsigned long long keys[1024];
    typedef float f2 __attribute__((ext_vector_type(2)));
    float a[UNROLL]; unsigned u[UNROLL]; unsigned long long w[UNROLL]; f2 p[UNROLL];
    const float b = 1.0000001f + seed, c = 1e-9f;
    for (int i = 0; i < UNROLL; ++i) { a[i] = 1.0f + 0.001f * (float)(threadIdx.x + i); u[i] = threadIdx.x * 2654435761u + (unsigned)i; w[i] = u[i]; }
    for (int i = 0; i < UNROLL; ++i) { p[i].x = a[i]; p[i].y = a[i] + 0.5f; }
    const f2 pb = { b, b }, pc = { c, c };
    if (OP == OP_LDS_MIN64) { for (int i = threadIdx.x; i < 1024; i += 256) keys[i] = ~0ull; __syncthreads(); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            else if (OP == OP_FMA_1V) asm volatile("v_fma_f32 %0, %0, 1.0, %1" : "+v"(a[i]) : "s"(c));      // no VGPR bank pressure: one vector source
            else if (OP == OP_MUL_F32) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(b));
            else if (OP == OP_FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
            else if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            else if (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(0x9E3779B1u));
            else if (OP == OP_MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(u[i]), "v"(0x9E3779B1u) : "vcc");
            else if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            else if (OP == OP_SQRT_HW) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            else if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]) : );
            else if (OP == OP_CVT_I2F) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            else if (OP == OP_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            else if (OP == OP_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            else if (OP == OP_DIV) { a[i] = b / a[i]; asm volatile("" : "+v"(a[i])); }
            else if (OP == OP_SQRT) { a[i] = __builtin_sqrtf(a[i]) + 1.0f; asm volatile("" : "+v"(a[i])); }
            else if (OP == OP_LDS_MIN64) { u[i] = u[i] * 1664525u + 1013904223u; atomicMin(&keys[u[i] >> 22], (unsigned long long)u[i] << 32 | (unsigned)i); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f; unsigned su = 0;
    for (int i = 0; i < UNROLL; ++i) { s += a[i] + p[i].x + p[i].y; su += u[i] + (unsigned)w[i]; }
    if (OP == OP_LDS_MIN64) { __syncthreads(); su += (unsigned)keys[threadIdx.x]; }
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)su;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w_ = blockIdx.x * 4 + (threadIdx.x >> 6);
        cyc[w_] = t1 - t0; rt[3 * w_] = r1 - r0; rt[3 * w_ + 1] = r0; rt[3 * w_ + 2] = r1;
    }
}

template <int OP>
static void run(int wps, int n_cu, float* d_out, unsigned long long* d_cyc, unsigned long long* d_rt, std::string& js, bool& first_row)
{
    const int blocks = n_cu * wps, iters = (OP == OP_DIV || OP == OP_SQRT) ? 4000 : (OP == OP_LDS_MIN64 ? 4000 : 20000);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_calib<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, d_rt, iters, 0.0f);      // warm clocks
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_calib<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, d_rt, iters, 0.0f);
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc((size_t)blocks * 4), rt((size_t)blocks * 12);
    CHK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(rt.data(), d_rt, rt.size() * 8, hipMemcpyDeviceToHost));
    double sc = 0, sr = 0;
    unsigned long long first = ~0ull, last = 0;
    for (size_t i = 0; i < cyc.size(); ++i) {
        sc += (double)cyc[i]; sr += (double)rt[3 * i];
        if (rt[3 * i + 1] < first) first = rt[3 * i + 1];
        if (rt[3 * i + 2] > last) last = rt[3 * i + 2];
    }
    const double span_s = (double)(last - first) * 1e-8;              // first wave's start to last wave's end (100 MHz ticks)
    const double resident = sr * 1e-8 / span_s / ((double)n_cu * 4.0);  // mean waves per SIMD actually running over that span
    const double n_inst = (double)iters * UNROLL;                       // per wave
    const double cyc_wave = sc / (double)cyc.size();
    const double clock_ghz = sc / sr * 0.1;                             // s_memrealtime ticks at 100 MHz
    const double total = n_inst * (double)blocks * 4.0;
    char buf[512];
    snprintf(buf, sizeof buf, "%s\n  {\"op\": \"%s\", \"waves_per_simd\": %d, \"workgroups\": %d, \"cyc_per_inst_simd\": %.3f, \"cyc_per_inst_wave\": %.3f, "
             "\"ginst_per_s\": %.1f, \"kernel_ms\": %.4f, \"clock_ghz\": %.3f, \"resident_waves_per_simd\": %.2f, \"cyc_per_inst_simd_wall\": %.3f}",
             first_row ? "" : ",", kOpName[OP], wps, blocks,
             cyc_wave / (n_inst * wps), cyc_wave / n_inst, total / span_s / 1e9, ms, clock_ghz, resident,
             span_s * clock_ghz * 1e9 * (double)n_cu * 4.0 / total);
    js += buf; first_row = false;
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

template <int OP>
static void run_all(int n_cu, float* d_out, unsigned long long* d_cyc, unsigned long long* d_rt, std::string& js, bool& first_row)
{
    const int wps[4] = { 1, 2, 4, 8 };
    for (int w : wps) run<OP>(w, n_cu, d_out, d_cyc, d_rt, js, first_row);
}

int main()
{
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    float* d_out; unsigned long long *d_cyc, *d_rt;
    CHK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4)); CHK(hipMalloc(&d_cyc, (size_t)n_cu * 8 * 4 * 8)); CHK(hipMalloc(&d_rt, (size_t)n_cu * 8 * 4 * 8 * 3));
    std::string js; bool first = true;
    run_all<OP_FMA>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_FMA_1V>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_MUL_F32>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_FMA_DEP>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_ADD_U32>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_CNDMASK>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_CVT_I2F>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_MUL_LO>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_MAD_U64>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_RCP>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_SQRT_HW>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_DIV>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_SQRT>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_LDS_MIN64>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_PK_FMA>(n_cu, d_out, d_cyc, d_rt, js, first);
    run_all<OP_PK_MUL>(n_cu, d_out, d_cyc, d_rt, js, first);
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"simds\": %d, \"clock_mhz_max\": %d, \"unroll\": %d,\n \"what\": \"cyc_per_inst_simd = shader cycles a SIMD spends per wave64 "
           "instruction (per division / square root for the ieee_* rows, per 64-lane ds_min_u64 on random keys of a 1024-key tile for the last row)\",\n \"rows\": [%s\n]}\n",
           prop.gcnArchName, n_cu, n_cu * 4, prop.clockRate / 1000, UNROLL, js.c_str());
    return 0;
}
