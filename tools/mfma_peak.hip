// Sustained-rate probe for the MFMA pipes of gfx950: how many TFLOP/s does a loop of NOTHING BUT
// independent v_mfma instructions reach on every CU, and at what shader clock?  The roofline in bench.py is
// quoted against the nominal peak (256 CUs x 2.4 GHz); this probe says how much of that a power-limited
// part actually sustains, which is the honest ceiling for the conv kernels.
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
//   run  : tools/mfma_peak            (GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

// NACC independent accumulators per wave, iters x NACC x 8 MFMAs per wave
template <int NACC>
__global__ __launch_bounds__(256) void mfma_f32_loop(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_bf16_loop(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(threadIdx.x * 1e-3f);
        b[i] = (__bf16)(blockIdx.x * 1e-3f);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}

template <typename K>
static void run(const char* name, K kernel, int nacc, double flops_per_mfma, int blocks_per_cu, int iters, int reps) {
    float* out;
    unsigned long long* clk;
    CHECK(hipMalloc(&out, 4));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, iters, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, iters, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double mfmas = (double)grid * 4 * iters * 8 * nacc * reps;
    const double tf = mfmas * flops_per_mfma / (ms * 1e-3) / 1e12;
    // s_memrealtime ticks at 100 MHz
    const double mhz = h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0;
    printf("%-22s acc/wave %d  waves/SIMD %d  reps %3d : %8.2f TFLOP/s   %7.3f ms/launch   memtime/realtime -> %.0f MHz\n", name, nacc,
           blocks_per_cu, reps, tf, ms / reps, mhz);
    CHECK(hipFree(out));
    CHECK(hipFree(clk));
}

int main() {
    const double F32 = 32.0 * 32 * 2 * 2, BF16 = 32.0 * 32 * 16 * 2;
    // short launches (~50 us, like one conv layer) and long ones (several ms: sustained, power-limited)
    for (int reps : {1, 50}) {
        for (int iters : {64, 4096}) {
            printf("-- iters %d (x8 x acc MFMAs per wave), %d launch(es) back to back\n", iters, reps);
            run("f32 32x32x2", mfma_f32_loop<1>, 1, F32, 1, iters, reps);
            run("f32 32x32x2", mfma_f32_loop<2>, 2, F32, 1, iters, reps);
            run("f32 32x32x2", mfma_f32_loop<2>, 2, F32, 2, iters, reps);
            run("f32 32x32x2", mfma_f32_loop<4>, 4, F32, 2, iters, reps);
            run("bf16 32x32x16", mfma_bf16_loop<2>, 2, BF16, 2, iters, reps);
            run("bf16 32x32x16", mfma_bf16_loop<4>, 4, BF16, 2, iters, reps);
        }
    }
    return 0;
}
