// Operand-path probe for gfx950: how many bytes per clock per CU can a block stage from (L2-resident) global memory
// into LDS, (a) with `buffer_load_dwordx4 ... lds` (LDS-DMA, what the conv kernels use) and (b) through registers
// (`buffer_load_dwordx4` to VGPRs + `ds_write_b128`)?  The K loops of csrc/igemm_f32.hip / igemm_wino.hip stage
// 24 KiB per 2048 MFMA cycles (128x64 tile) / 64 KiB per 4096 (Winograd 64x64x4): if the LDS-DMA rate per CU is close
// to 12 B/clk, THAT is their bound, not the matrix pipe and not L2.
//   build: hipcc -O3 --offload-arch=gfx950 tools/dma_rate.hip -o tools/dma_rate ;  run on the GPU box: tools/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

// MODE 0: LDS-DMA from a per-block 64 KiB window (L2 resident after the first pass); 1: LDS-DMA, every offset out of range
// (hardware zero fill, no memory access); 2: global -> VGPR -> ds_write_b128; 3: MODE 0 with MFMAs between the loads.
// PER: loads per thread between two waits (each load = 16 B per lane = 1 KiB per wave).
template <int MODE, int PER, int NW, int MF = 4>
__global__ __launch_bounds__(64 * NW) void stage_loop(const float* src, float* out, int iters, unsigned long long* clk) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const float* base = src + (size_t)blockIdx.x * 16384;          // 64 KiB window per block
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 65536u, 0x00020000);
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc = {0};
    float a = tid * 1e-3f, b = 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 r[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const unsigned off = MODE == 1 ? 0x80000000u : (unsigned)(((i * NW * 64 + tid) * 16) & 65535);
            float* dst = lds + ((i * NW + wave) * 64) * 4;            // 1 KiB slot per (load, wave)
            if (MODE == 2 || MODE == 4) {
                r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            } else if (MODE != 5) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, off, 0, 0, 0);
            }
            if (MODE == 3 || MODE == 4 || MODE == 5) {
#pragma unroll
                for (int k = 0; k < MF; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
        if (MODE == 4) {          // register path under MFMAs: the ds_writes follow, again MF MFMAs each
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                *reinterpret_cast<f32x4*>(lds + ((i * NW + wave) * 64 + (tid & 63)) * 4) = r[i];
#pragma unroll
                for (int k = 0; k < MF; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < PER; ++i) *reinterpret_cast<f32x4*>(lds + ((i * NW + wave) * 64 + (tid & 63)) * 4) = r[i];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = lds[tid] + acc[0];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
#endif
}

template <int MODE, int PER, int NW, int MF = 4>
static void run(const char* what, const float* src, float* out, unsigned long long* clk, int blocks_per_cu) {
    const int iters = 2000;
    const size_t lds = (size_t)PER * NW * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stage_loop<MODE, PER, NW, MF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((stage_loop<MODE, PER, NW, MF>), dim3(grid), dim3(64 * NW), lds, 0, src, out, 10, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((stage_loop<MODE, PER, NW, MF>), dim3(grid), dim3(64 * NW), lds, 0, src, out, iters, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c;
    CHECK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const double bytes_cu = (double)iters * PER * NW * 1024 * blocks_per_cu;       // per CU
    printf("%-44s %d waves x %2d loads, %d block(s)/CU: %7.1f GB/s per CU  %6.1f B/clk/CU (memtime clk: %5.1f B/tick/block)  chip %5.2f TB/s\n",
           what, NW, PER, blocks_per_cu, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu / (ms * 1e-3) / 2.4e9,
           (double)iters * PER * NW * 1024 / (double)c, bytes_cu * 256 / (ms * 1e-3) / 1e12);
}

int main() {
    float *src, *out;
    unsigned long long* clk;
    CHECK(hipMalloc(&src, (size_t)256 * 4 * 65536));
    CHECK(hipMemset(src, 0, (size_t)256 * 4 * 65536));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMalloc(&clk, 64));
    run<0, 8, 4>("LDS-DMA, L2-resident source", src, out, clk, 1);
    run<0, 16, 4>("LDS-DMA, L2-resident source", src, out, clk, 1);
    run<0, 8, 8>("LDS-DMA, L2-resident source", src, out, clk, 1);
    run<0, 8, 4>("LDS-DMA, L2-resident source", src, out, clk, 3);
    run<1, 8, 4>("LDS-DMA, out-of-range (zero fill)", src, out, clk, 1);
    run<1, 8, 4>("LDS-DMA, out-of-range (zero fill)", src, out, clk, 3);
    run<2, 8, 4>("global -> VGPR -> ds_write_b128", src, out, clk, 1);
    run<2, 16, 4>("global -> VGPR -> ds_write_b128", src, out, clk, 1);
    run<2, 8, 8>("global -> VGPR -> ds_write_b128", src, out, clk, 1);
    run<2, 8, 4>("global -> VGPR -> ds_write_b128", src, out, clk, 3);
    run<5, 8, 4>("no loads, 4 MFMA per slot (MFMA bound)", src, out, clk, 1);
    run<3, 8, 4>("LDS-DMA + 4 MFMA per load", src, out, clk, 1);
    run<3, 8, 4>("LDS-DMA + 4 MFMA per load", src, out, clk, 3);
    run<3, 8, 8>("LDS-DMA + 4 MFMA per load", src, out, clk, 1);
    run<3, 8, 4, 8>("LDS-DMA + 8 MFMA per load", src, out, clk, 1);
    run<4, 8, 4, 2>("VGPR path: load + 2 MFMA, ds_write + 2 MFMA", src, out, clk, 1);
    run<4, 8, 4, 2>("VGPR path: load + 2 MFMA, ds_write + 2 MFMA", src, out, clk, 3);
    run<4, 8, 4, 4>("VGPR path: load + 4 MFMA, ds_write + 4 MFMA", src, out, clk, 1);
    return 0;
}
