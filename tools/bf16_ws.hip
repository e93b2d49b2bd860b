// EXPERIMENT harness for the "2-D halo" bf16 3x3 / stride-1 conv tile that csrc/igemm_bf16_ws.hip ships (same device code, included
// below): stand-alone build with its own direct-conv check and per-shape timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I contextaware-poseformer_amd/csrc -o tools/ab/bf16_ws tools/bf16_ws.hip && tools/ab/bf16_ws
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "igemm_bf16_ws_tile.h"

using namespace capf;

static inline unsigned short f2bf_host(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf2f_host(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void direct_ref(const unsigned short* x, const unsigned short* w, const float* bias, const unsigned short* res, float* y,
                           int B, int H, int W, int C, int N, int relu) {     // w: [3][3][C][N] bf16
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * H * W * N) return;
    const int n = i % N;
    const long px = i / N;
    const int wc = px % W, h = (px / W) % H, b = px / ((long)W * H);
    float s = 0.f;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            const int hh = h + kh - 1, ww = wc + kw - 1;
            if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
            const unsigned short* xp = x + (((long)b * H + hh) * W + ww) * C;
            const unsigned short* wp = w + ((long)(kh * 3 + kw) * C) * N + n;
            for (int c = 0; c < C; ++c) s += __uint_as_float((unsigned)xp[c] << 16) * __uint_as_float((unsigned)wp[(long)c * N] << 16);
        }
    s += bias[n];
    if (res) s += __uint_as_float((unsigned)res[i] << 16);
    if (relu) s = fmaxf(s, 0.f);
    y[i] = s;
}

__device__ unsigned long long g_clk[4];      // shader clock / 100 MHz wall clock at the start and end of block 8 (actual frequency under load)

template <int TN>
__global__ __launch_bounds__(256, 2) void ws_kernel(WsProblem p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (blockIdx.x == 8 && threadIdx.x == 0) { g_clk[0] = clock64(); g_clk[1] = wall_clock64(); }
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);      // XCD-contiguous tile order
    igemm_bf16_ws_tile<TN>(p, bid, lds);
    if (blockIdx.x == 8 && threadIdx.x == 0) { g_clk[2] = clock64(); g_clk[3] = wall_clock64(); }
#endif
}

static double run(int B, int H, int W, int C, int N, bool with_res, bool check, int reps = 20) {
    const long nx = (long)B * H * W * C, ny = (long)B * H * W * N, nw = 9L * C * N;
    std::vector<unsigned short> hx(nx), hw(nw), hr(ny);
    std::vector<float> hb(N);
    srand(1);
    for (auto& v : hx) v = f2bf_host((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hw) v = f2bf_host((rand() % 2001 - 1000) / (30.f * C));
    for (auto& v : hr) v = f2bf_host((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hb) v = (rand() % 2001 - 1000) / 2000.f;
    WsProblem p{};
    if (!ws_plan(B, H, W, C, N, &p)) { printf("B=%d %dx%d %d->%d: not eligible\n", B, H, W, C, N); return 0; }
    const int TN = p.NS / 32, NSL = p.NSL, NCC = C / 16;
    // packed weights [slice][cc][tap][n][quad position][8]
    std::vector<unsigned short> hp((size_t)NSL * NCC * 9 * p.NS * 16, 0);
    for (int sl = 0; sl < NSL; ++sl)
        for (int cc = 0; cc < NCC; ++cc)
            for (int tap = 0; tap < 9; ++tap)
                for (int n = 0; n < p.NS; ++n)
                    for (int h = 0; h < 2; ++h)
                        for (int e = 0; e < 8; ++e) {
                            const int ng = sl * p.NS + n, c = cc * 16 + h * 8 + e, qp = h ^ ((n >> 3) & 1);
                            hp[((((size_t)(sl * NCC + cc) * 9 + tap) * p.NS + n) * 2 + qp) * 8 + e] = ng < N ? hw[((long)tap * C + c) * N + ng] : 0;
                        }
    unsigned short *dx, *dw, *dp, *dy, *dres;
    float *dr, *db;
    hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&dp, hp.size() * 2); hipMalloc(&dy, ny * 2); hipMalloc(&dres, ny * 2);
    hipMalloc(&dr, ny * 4); hipMalloc(&db, N * 4);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dres, hr.data(), ny * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(dy, 0, ny * 2);
    p.x = dx; p.wp = dp; p.bias = db; p.res = with_res ? dres : nullptr; p.y = dy; p.relu = 1;
    const size_t lds_bytes = 2 * ws_stage_bytes(p.NS);
    const int grid = p.tiles_m * NSL;
    auto launch = [&]() {
        if (TN == 3) { hipFuncSetAttribute(reinterpret_cast<const void*>(&ws_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); hipLaunchKernelGGL(ws_kernel<3>, dim3(grid), dim3(256), lds_bytes, 0, p); }
        else if (TN == 2) { hipFuncSetAttribute(reinterpret_cast<const void*>(&ws_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); hipLaunchKernelGGL(ws_kernel<2>, dim3(grid), dim3(256), lds_bytes, 0, p); }
        else { hipFuncSetAttribute(reinterpret_cast<const void*>(&ws_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); hipLaunchKernelGGL(ws_kernel<1>, dim3(grid), dim3(256), lds_bytes, 0, p); }
    };
    launch();
    { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("  kernel error: %s\n", hipGetErrorString(e)); exit(1); } }
    if (check) {
        hipLaunchKernelGGL(direct_ref, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, 0, dx, dw, db, with_res ? dres : nullptr, dr, B, H, W, C, N, 1);
        std::vector<unsigned short> a(ny);
        std::vector<float> rf(ny);
        hipMemcpy(a.data(), dy, ny * 2, hipMemcpyDeviceToHost);
        hipMemcpy(rf.data(), dr, ny * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0; long bad = 0, first = -1;
        for (long i = 0; i < ny; ++i) {
            const double d = fabs((double)bf2f_host(a[i]) - rf[i]);
            worst = fmax(worst, d); scale = fmax(scale, fabs((double)rf[i]));
            if (d > 1e-2 * fmax(1.0, fabs((double)rf[i]))) { ++bad; if (first < 0) first = i; }
        }
        printf("  check B=%d %dx%d %d->%d res=%d (RH %d G %d P %d PP %d NS %d x %d, %d tiles): max |ws - direct| = %.3e (max |direct| %.3f) bad %ld first %ld %s\n",
               B, H, W, C, N, (int)with_res, p.RH, p.G, p.P, p.PP, p.NS, NSL, grid, worst, scale, bad, first, bad == 0 ? "OK" : "MISMATCH");
    }
    if (reps <= 0) return 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    {
        unsigned long long hc[4];
        hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc));
        printf("   [block 8 of the last launch: %.0f shader cycles in %.2f us = %.0f MHz]\n", (double)(hc[2] - hc[0]), (hc[3] - hc[1]) / 100.0,
               (double)(hc[2] - hc[0]) / ((hc[3] - hc[1]) / 100.0));
    }
    const double us = ms * 1e3 / reps, gf = 2.0 * B * H * W * (double)N * 9 * C / 1e9, mb = ((double)nx + ny * (with_res ? 2 : 1)) * 2 / 1e6;
    printf("B=%d %dx%d %d->%d res=%d bf16 ws: %8.1f us  %7.1f TFLOP/s  %6.2f TB/s (alg)  grid %d\n", B, H, W, C, N, (int)with_res, us, gf / us * 1e3, mb / us, grid);
    hipFree(dx); hipFree(dw); hipFree(dp); hipFree(dy); hipFree(dres); hipFree(dr); hipFree(db);
    return us;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "check");
    // correctness: ragged tiles, every TN, segments across images, odd widths
    run(2, 8, 8, 32, 32, true, true, 0);
    run(3, 16, 16, 48, 48, false, true, 0);
    run(5, 8, 8, 96, 96, true, true, 0);
    run(3, 64, 64, 48, 48, true, true, 0);
    run(2, 32, 32, 96, 96, true, true, 0);
    run(3, 16, 16, 192, 192, true, true, 0);
    run(7, 8, 8, 384, 384, false, true, 0);
    run(2, 24, 18, 64, 64, true, true, 0);
    run(2, 12, 9, 128, 128, true, true, 0);
    run(1, 96, 72, 64, 64, false, true, 0);
    run(2, 64, 48, 32, 32, true, true, 0);
    run(3, 8, 6, 256, 256, true, true, 0);
    if (quick) return 0;
    double sum = 0;
    for (int res = 0; res < 2; ++res) {
        sum = 0;
        sum += run(256, 64, 64, 48, 48, res, false);
        sum += run(256, 32, 32, 96, 96, res, false);
        sum += run(256, 16, 16, 192, 192, res, false);
        sum += run(256, 8, 8, 384, 384, res, false);
        printf("== HRNet-48 level at batch 256, res=%d: sum of the four branches %.1f us = %.1f TFLOP/s\n", res, sum, 4 * 43.5e3 / sum * 1.0);
    }
    run(256, 64, 64, 32, 32, true, false);
    run(256, 32, 32, 64, 64, true, false);
    run(256, 16, 16, 128, 128, true, false);
    run(256, 8, 8, 256, 256, true, false);
    run(128, 96, 72, 64, 64, false, false);
    run(128, 48, 36, 128, 128, false, false);
    run(128, 24, 18, 256, 256, false, false);
    run(128, 12, 9, 512, 512, false, false);
    return 0;
}
