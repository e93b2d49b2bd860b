// Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* on gfx950 for the access patterns of this library's LDS-DMA loaders
// ("FETCH_SIZE calibration for 64-B-segment LDS-DMA reads").  A 2 GiB buffer (far beyond L2 + MALL) is read once:
//   contig : every wave instruction (buffer_load_dwordx4 ... lds) covers 1 KiB of consecutive bytes (the bf16 / fp32 direct tiles)
//   seg64  : every wave instruction covers 16 segments of 64 B, one per 128-B line; the OTHER half of each line is read by a second
//            launch much later (the F(4,3) tile: 16 floats of a 32-float pixel row per sub-chunk, the next sub-chunk a superchunk later)
//   seg32  : 32-B segments, one per 128-B line, four launches (the 2-D halo tile: 16 bf16 channels of a 48 ... 384-channel pixel)
// Known bytes per pattern = 2 GiB.  Run under rocprofv3 --pmc (tools/fetch_calib.sh) and compare.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ab/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// seg = contiguous bytes per 128-B line (128 = contig, 64, 32); phase = which part of the line this launch reads
template <int SEG>
__global__ __launch_bounds__(256) void read_kernel(const char* base, size_t bytes, int phase) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPS = SEG / 16;                       // lanes per segment
    constexpr int LINES = 64 / LPS;                     // 128-B lines one wave instruction touches
    const size_t lines_total = bytes / 128;
    const size_t waves_total = (size_t)gridDim.x * 4;
    const size_t wid = (size_t)blockIdx.x * 4 + wave;
    // the buffer is walked in 1 GiB windows (32-bit buffer offsets)
    for (size_t line0 = wid * LINES; line0 < lines_total; line0 += waves_total * LINES) {
        const size_t win = (line0 * 128) >> 30;
        const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (win << 30)), 0, 0x40000000u, 0x00020000);
        const unsigned off = (unsigned)((line0 * 128) & 0x3FFFFFFFu) + (unsigned)(lane / LPS) * 128u + (unsigned)phase * SEG + (unsigned)(lane % LPS) * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + wave * 1024), 16, off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
    const size_t bytes = 2ull << 30;
    char* buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](const char* name, auto launch, int launches) {
        hipEventRecord(e0);
        for (int p = 0; p < launches; ++p) launch(p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-8s %d launch(es) over 2 GiB: %.1f us, %.2f TB/s of requested bytes\n", name, launches, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    };
    for (int rep = 0; rep < 2; ++rep) {
        timed("contig", [&](int p) { hipLaunchKernelGGL(read_kernel<128>, dim3(2048), dim3(256), 0, 0, buf, bytes, p); }, 1);
        timed("seg64", [&](int p) { hipLaunchKernelGGL(read_kernel<64>, dim3(2048), dim3(256), 0, 0, buf, bytes, p); }, 2);
        timed("seg32", [&](int p) { hipLaunchKernelGGL(read_kernel<32>, dim3(2048), dim3(256), 0, 0, buf, bytes, p); }, 4);
    }
    hipFree(buf);
    return 0;
}
