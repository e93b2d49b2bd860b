// How much does the conv epilogue's store pattern cost?  The MFMA C layout gives a lane one output row and
// 4 consecutive channels per register group, so one store instruction writes 32 B (two lanes) to each of 32
// different rows; a 128-byte row segment is completed by 4 instructions.  This probe writes the same [M][N]
// fp32 matrix (a) in that pattern and (b) with 8 consecutive lanes covering 128 contiguous bytes of a row.
//   build: hipcc -O3 --offload-arch=gfx950 tools/store_pattern.hip -o tools/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// block = 256 threads = 4 waves; wave w of block b owns rows [b*128 + w*32, +32) x all N columns (N % 32 == 0)
template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* out, int M, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long m0 = (long)blockIdx.x * 128 + wave * 32;
    if (m0 >= M) return;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int n0 = 0; n0 < N; n0 += 32) {
        if (MODE == 0) {            // MFMA C layout: row = lane & 31, 16 B at channel 8g + 4*(lane>>5)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(out + (m0 + (lane & 31)) * N + n0 + 8 * g + 4 * (lane >> 5)) = v;
        } else {                    // transposed through LDS first: 8 lanes x 16 B = one 128-byte row segment
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(out + (m0 + g * 8 + (lane >> 3)) * N + n0 + 4 * (lane & 7)) = v;
        }
    }
}

template <int MODE>
static void run(const char* name, int M, int N) {
    float* out;
    hipMalloc(&out, (size_t)M * N * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = (M + 127) / 128;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, M, N);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, M, N);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s M %7d N %4d : %8.1f us  %7.1f GB/s\n", name, M, N, ms / reps * 1e3, (double)M * N * 4 / (ms / reps * 1e-3) / 1e9);
    hipFree(out);
}

int main() {
    for (int N : {32, 64, 256}) {
        const int M = 262144;
        run<0>("MFMA C layout (32 B pieces)", M, N);
        run<1>("row-contiguous 128 B", M, N);
    }
    return 0;
}
