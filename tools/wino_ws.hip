// EXPERIMENT harness for the fp32 Winograd F(4,3) tile on the 2-D halo loop structure (csrc/igemm_wino_ws_tile.h): stand-alone build with its
// own direct-conv check and per-shape timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I contextaware-poseformer_amd/csrc -I tools -o tools/ab/wino_ws tools/wino_ws.hip && tools/ab/wino_ws
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wino_ws_tile.h"

using namespace capf;

__global__ void direct_ref(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W, int C, int N, int relu) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;          // w: [N][C][3][3]
    if (i >= (long)B * H * W * N) return;
    const int n = i % N;
    const long px = i / N;
    const int wc = px % W, h = (px / W) % H, b = px / ((long)W * H);
    double s = 0.0;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            const int hh = h + kh - 1, ww = wc + kw - 1;
            if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
            const float* xp = x + (((long)b * H + hh) * W + ww) * C;
            for (int c = 0; c < C; ++c) s += (double)xp[c] * (double)w[(((long)n * C + c) * 3 + kh) * 3 + kw];
        }
    s += bias[n];
    if (res) s += res[i];
    if (relu) s = s > 0 ? s : 0;
    y[i] = (float)s;
}

__global__ __launch_bounds__(256, 2) void ww_kernel(WwProblem p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    igemm_wino_ws_tile(p, bid, lds);
#endif
}

static double run(int B, int H, int W, int C, int N, bool with_res, bool check, int reps = 20) {
    const long nx = (long)B * H * W * C, ny = (long)B * H * W * N, nw = 9L * C * N;
    std::vector<float> hx(nx), hw(nw), hr(ny), hb(N);
    srand(1);
    for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) / (30.f * sqrtf((float)C));
    for (auto& v : hr) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hb) v = (rand() % 2001 - 1000) / 2000.f;
    WwProblem p{};
    if (!ww_plan(B, H, W, C, N, &p)) { printf("B=%d %dx%d %d->%d: not eligible\n", B, H, W, C, N); return 0; }
    const int NSL = p.NSL, NCC = C / 8;
    std::vector<float> hp((size_t)NSL * NCC * 18 * 32 * 8, 0.f);
    for (int sl = 0; sl < NSL; ++sl)
        for (int cc = 0; cc < NCC; ++cc)
            for (int kh = 0; kh < 3; ++kh)
                for (int pq = 0; pq < 6; ++pq)
                    for (int n = 0; n < 32; ++n)
                        for (int h = 0; h < 2; ++h)
                            for (int e = 0; e < 4; ++e) {
                                const int ng = sl * 32 + n, c = cc * 8 + h * 4 + e, qp = h ^ ((n >> 3) & 1);
                                float u = 0.f;
                                if (ng < N) {
                                    const float* g = &hw[(((long)ng * C + c) * 3 + kh) * 3];
                                    const float g0 = g[0], g1 = g[1], g2 = g[2];
                                    u = pq == 0 ? 0.25f * g0 : pq == 1 ? -((g0 + g1) + g2) / 6.0f : pq == 2 ? -((g0 - g1) + g2) / 6.0f
                                        : pq == 3 ? (g0 / 24.0f + g1 / 12.0f) + g2 / 6.0f : pq == 4 ? (g0 / 24.0f - g1 / 12.0f) + g2 / 6.0f : g2;
                                }
                                hp[(((((size_t)(sl * NCC + cc) * 3 + kh) * 6 + pq) * 32 + n) * 2 + qp) * 4 + e] = u;
                            }
    float *dx, *dw, *dp, *dy, *dres, *dr, *db;
    hipMalloc(&dx, nx * 4); hipMalloc(&dw, nw * 4); hipMalloc(&dp, hp.size() * 4); hipMalloc(&dy, ny * 4); hipMalloc(&dres, ny * 4);
    hipMalloc(&dr, ny * 4); hipMalloc(&db, N * 4);
    hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dres, hr.data(), ny * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(dy, 0, ny * 4);
    p.x = dx; p.wp = dp; p.bias = db; p.res = with_res ? dres : nullptr; p.y = dy; p.relu = 1;
    const size_t lds_bytes = 2 * WW_STAGE;
    const int grid = p.tiles_m * NSL;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ww_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    auto launch = [&]() { hipLaunchKernelGGL(ww_kernel, dim3(grid), dim3(256), lds_bytes, 0, p); };
    launch();
    { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("  kernel error: %s\n", hipGetErrorString(e)); exit(1); } }
    if (check) {
        hipLaunchKernelGGL(direct_ref, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, 0, dx, dw, db, with_res ? dres : nullptr, dr, B, H, W, C, N, 1);
        std::vector<float> a(ny), rf(ny);
        hipMemcpy(a.data(), dy, ny * 4, hipMemcpyDeviceToHost);
        hipMemcpy(rf.data(), dr, ny * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0; long bad = 0, first = -1;
        for (long i = 0; i < ny; ++i) {
            const double d = fabs((double)a[i] - rf[i]);
            worst = fmax(worst, d); scale = fmax(scale, fabs((double)rf[i]));
            if (d > 2e-4) { ++bad; if (first < 0) first = i; }
        }
        printf("  check B=%d %dx%d %d->%d res=%d (RH %d G %d tiles %d SL %d x %d slices, %d blocks): max |ww - direct| = %.3e (max |direct| %.3f) bad %ld first %ld %s\n",
               B, H, W, C, N, (int)with_res, p.RH, p.G, p.tiles, p.SL, NSL, grid, worst, scale, bad, first, bad == 0 ? "OK" : "MISMATCH");
    }
    double us = 0;
    if (reps > 0) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        us = ms * 1e3 / reps;
        const double gf = 2.0 * B * H * W * (double)N * 9 * C / 1e9;
        printf("B=%d %dx%d %d->%d res=%d fp32 F(4,3) ws: %8.1f us  %7.1f TFLOP/s algorithmic  %6.1f executed  grid %d\n", B, H, W, C, N, (int)with_res, us,
               gf / us * 1e3, gf / us * 1e3 / 2, grid);
    }
    hipFree(dx); hipFree(dw); hipFree(dp); hipFree(dy); hipFree(dres); hipFree(dr); hipFree(db);
    return us;
}

int main(int argc, char** argv) {
    run(2, 8, 8, 32, 32, true, true, 0);
    run(3, 16, 16, 16, 48, false, true, 0);
    run(5, 8, 8, 64, 64, true, true, 0);
    run(2, 64, 64, 32, 32, true, true, 0);
    run(2, 32, 32, 64, 64, true, true, 0);
    run(3, 16, 16, 128, 128, true, true, 0);
    run(7, 8, 8, 256, 256, false, true, 0);
    run(2, 24, 12, 64, 40, true, true, 0);
    run(1, 64, 48, 32, 32, true, true, 0);
    if (argc > 1 && !strcmp(argv[1], "check")) return 0;
    for (int B : {64, 512}) {
        double sum = 0;
        sum += run(B, 64, 64, 32, 32, true, false);
        sum += run(B, 32, 32, 64, 64, true, false);
        sum += run(B, 16, 16, 128, 128, true, false);
        sum += run(B, 8, 8, 256, 256, true, false);
        printf("== HRNet-32 level at batch %d: sum of the four branches %.1f us = %.1f TFLOP/s algorithmic\n", B, sum, B * 0.2539e3 / sum);
        run(B, 64, 64, 64, 64, false, false);
        run(B, 64, 64, 256, 32, false, false);
    }
    return 0;
}
