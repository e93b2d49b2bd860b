// EXPERIMENT harness for the split-fp32 ("two fp16 pieces, three products") 3x3 / stride-1 conv tile of csrc/igemm_f32h2_ws_tile.h:
// stand-alone build, distance to an fp64 direct convolution in units of the sum of |terms|, per-shape timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I contextaware-poseformer_amd/csrc -I tools -o tools/ab/f32h2_ws tools/f32h2_ws.hip && tools/ab/f32h2_ws
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <utility>

#include "igemm_f32h2_ws_tile.h"
#include "f32h2_pp_tile.h"

using namespace capf;

static inline unsigned short f2h_host(float f) { const _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static inline float h2f_host(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

__global__ void direct_ref(const float* x, const float* w, const float* bias, const float* res, double* y, double* mass,
                           int B, int H, int W, int C, int N, int relu) {     // w: [3][3][C][N]
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * H * W * N) return;
    const int n = i % N;
    const long px = i / N;
    const int wc = px % W, h = (px / W) % H, b = px / ((long)W * H);
    double s = 0., m = 0.;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            const int hh = h + kh - 1, ww = wc + kw - 1;
            if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
            const float* xp = x + (((long)b * H + hh) * W + ww) * C;
            const float* wp = w + ((long)(kh * 3 + kw) * C) * N + n;
            for (int c = 0; c < C; ++c) { const double t = (double)xp[c] * (double)wp[(long)c * N]; s += t; m += fabs(t); }
        }
    s += bias[n]; m += fabs((double)bias[n]);
    if (res) { s += res[i]; m += fabs((double)res[i]); }
    if (relu) s = fmax(s, 0.);
    y[i] = s; mass[i] = m;
}

__device__ unsigned long long g_clk[4];      // shader clock / 100 MHz wall clock at the start and end of block 0 (actual frequency under load)

template <int TN>
__global__ __launch_bounds__(256, TN == 1 ? 3 : 2) void x3_kernel(H2Problem p, int nt, int tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (blockIdx.x == 8 && threadIdx.x == 0) { g_clk[0] = clock64(); g_clk[1] = wall_clock64(); }
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);      // XCD-contiguous tile order
    if (bid < tiles) igemm_f32h2_ws_tile<TN>(p, bid, lds);
    if (blockIdx.x == 8 && threadIdx.x == 0) { g_clk[2] = clock64(); g_clk[3] = wall_clock64(); }
#endif
}

template <int TN>
__global__ __launch_bounds__(256, 1) void x3_kernel_pp(H2Problem p, int nt, int tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (blockIdx.x == 8 && threadIdx.x == 0) { g_clk[0] = clock64(); g_clk[1] = wall_clock64(); }
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    if (bid < tiles) igemm_f32h2_ws_tile_pp<TN>(p, bid, lds);
    if (blockIdx.x == 8 && threadIdx.x == 0) { g_clk[2] = clock64(); g_clk[3] = wall_clock64(); }
#endif
}

static int g_pp = 0;                       // H2_PP=1: the two-stage (software-pipelined) form of the tile, one block per CU
static int g_ns = 32, g_nt = 1, g_wide = 0;     // g_wide: 1 = values spread over 2^+-20 (+ per-channel weight scales), 2 = x 1e20, 3 = x 1e-20

static double run(int B, int H, int W, int C, int N, bool with_res, bool check, int reps = 20) {
    const long nx = (long)B * H * W * C, ny = (long)B * H * W * N, nw = 9L * C * N;
    std::vector<float> hx(nx), hw(nw), hr(ny), hb(N);
    srand(1);
    auto rnd = [] { return ((rand() & 0xFFFF) * 65536.0 + (rand() & 0xFFFF)) / 4294967296.0 * 2.0 - 1.0; };   // full-mantissa values
    for (auto& v : hx) v = (float)rnd();
    for (auto& v : hw) v = (float)(rnd() * 30.0 / C);
    if (g_wide == 1) {
        for (auto& v : hx) v *= exp2f((float)(rand() % 41 - 20));
        for (long k = 0; k < nw; ++k) hw[k] *= exp2f((float)((k % N) * 7 % 13 - 6));
    } else if (g_wide == 2) { for (auto& v : hx) v *= 1e20f; }
    else if (g_wide == 3) { for (auto& v : hx) v *= 1e-20f; }
    for (auto& v : hr) v = (float)rnd();
    for (auto& v : hb) v = (float)(rnd() * 0.5);
    H2Problem p{};
    if (!h2_plan(B, H, W, C, N, g_ns, &p)) { printf("B=%d %dx%d %d->%d: not eligible\n", B, H, W, C, N); return 0; }
    const int NS = p.g.NS, TN = NS / 32, NSL = p.g.NSL, NCC = C / 16;
    std::vector<unsigned short> hp((size_t)h2_pack_elems(N, C), 0);
    float* hinv = reinterpret_cast<float*>(hp.data() + h2_piece_elems(N, C));
    const int NSL32 = (N + 31) / 32;
    std::vector<float> tsc(NSL32 * 32, 1.f);
    for (int ng = 0; ng < NSL32 * 32; ++ng) {
        float m = 0.f;
        if (ng < N) for (long k = 0; k < 9L * C; ++k) m = fmaxf(m, fabsf(hw[k * N + ng]));
        int mb; memcpy(&mb, &m, 4);
        const int sb = h2_scale_exp(mb), ib = (254 - sb) << 23, tb = sb << 23;
        memcpy(&tsc[ng], &tb, 4);
        memcpy(&hinv[ng], &ib, 4);
    }
    for (int sl = 0; sl < NSL32; ++sl)
        for (int cc = 0; cc < NCC; ++cc)
            for (int tap = 0; tap < 9; ++tap)
                for (int n = 0; n < 32; ++n)
                    for (int h = 0; h < 2; ++h)
                        for (int e = 0; e < 8; ++e) {
                            const int ng = sl * 32 + n, c = cc * 16 + h * 8 + e, qp = h ^ ((n >> 3) & 1);
                            float v = ng < N ? hw[((long)tap * C + c) * N + ng] * tsc[ng] : 0.f;
                            for (int pc = 0; pc < 2; ++pc) {
                                const unsigned short b16 = f2h_host(v);
                                hp[(((((size_t)(sl * NCC + cc) * 2 + pc) * 9 + tap) * 32 + n) * 2 + qp) * 8 + e] = b16;
                                v -= h2f_host(b16);
                            }
                        }
    float *dx, *dw, *dy, *dres, *db;
    unsigned short* dp;
    double *dr, *dm;
    hipMalloc(&dx, nx * 4); hipMalloc(&dw, nw * 4); hipMalloc(&dp, hp.size() * 2); hipMalloc(&dy, ny * 4); hipMalloc(&dres, ny * 4);
    hipMalloc(&db, N * 4);
    hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dres, hr.data(), ny * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(dy, 0, ny * 4);
    p.x = dx; p.g.wp = dp; p.winv = reinterpret_cast<const float*>(dp + h2_piece_elems(N, C)); p.g.bias = db; p.res = with_res ? dres : nullptr; p.y = dy; p.g.relu = 1;
    const size_t lds_bytes = g_pp ? h2p_lds_bytes(NS) : h2_lds_bytes(NS);
    const int tiles = p.g.tiles_m * NSL, grid = tiles;
    auto launch = [&]() {
        if (g_pp && TN == 2) { hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_kernel_pp<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); hipLaunchKernelGGL(x3_kernel_pp<2>, dim3(grid), dim3(256), lds_bytes, 0, p, g_nt, tiles); }
        else if (g_pp) { hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_kernel_pp<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); hipLaunchKernelGGL(x3_kernel_pp<1>, dim3(grid), dim3(256), lds_bytes, 0, p, g_nt, tiles); }
        else if (TN == 2) { hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); hipLaunchKernelGGL(x3_kernel<2>, dim3(grid), dim3(256), lds_bytes, 0, p, g_nt, tiles); }
        else { hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); hipLaunchKernelGGL(x3_kernel<1>, dim3(grid), dim3(256), lds_bytes, 0, p, g_nt, tiles); }
    };
    launch();
    { int nb = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, TN == 2 ? reinterpret_cast<const void*>(&x3_kernel<2>) : reinterpret_cast<const void*>(&x3_kernel<1>), 256, lds_bytes); static int once[3] = {0, 0, 0}; if (!once[TN]++) printf("   [occupancy: %d blocks of 256 threads per CU with %zu B of LDS]\n", nb, lds_bytes); }
    { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("  kernel error: %s\n", hipGetErrorString(e)); exit(1); } }
    if (check) {
        hipMalloc(&dr, ny * 8); hipMalloc(&dm, ny * 8);
        hipLaunchKernelGGL(direct_ref, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, 0, dx, dw, db, with_res ? dres : nullptr, dr, dm, B, H, W, C, N, 1);
        std::vector<float> a(ny);
        std::vector<double> rf(ny), ms(ny);
        hipMemcpy(a.data(), dy, ny * 4, hipMemcpyDeviceToHost);
        hipMemcpy(rf.data(), dr, ny * 8, hipMemcpyDeviceToHost);
        hipMemcpy(ms.data(), dm, ny * 8, hipMemcpyDeviceToHost);
        double worst = 0, rms = 0; long bad = 0, first = -1;
        for (long i = 0; i < ny; ++i) {
            const double d = fabs((double)a[i] - rf[i]) / ms[i];
            worst = fmax(worst, d); rms += d * d;
            if (!(d <= 2e-6)) { ++bad; if (first < 0) first = i; }
        }
        printf("  check B=%d %dx%d %d->%d res=%d (RH %d G %d P %d PP %d NS %d x %d, %d tiles): |h2 - fp64| / sum|terms|: max %.3e rms %.3e  bad %ld first %ld %s\n",
               B, H, W, C, N, (int)with_res, p.g.RH, p.g.G, p.g.P, p.g.PP, NS, NSL, grid, worst, sqrt(rms / ny), bad, first, bad == 0 ? "OK" : "MISMATCH");
        hipFree(dr); hipFree(dm);
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
        unsigned long long hc[4];
        hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc));
#ifdef H2_KNOCK
        if ((H2_KNOCK & 32) && g_pp) {
            long long sg[8];
            hipMemcpyFromSymbol(sg, HIP_SYMBOL(capf::h2_seg_dbg), sizeof(sg));
            const double n = sg[5] > 1 ? (double)(sg[5] - 1) : 1.0;
            printf("   [wave 0 of block 8, cycles per chunk: issue %.0f  taps %.0f  rescale+maximum %.0f  wait %.0f  barrier %.0f]\n", sg[0] / (n + 1), sg[1] / (n + 1), sg[2] / n, sg[3] / n, sg[4] / n);
        }
#endif
        printf("   [block 8 of the last launch: %.0f shader cycles in %.2f us = %.0f MHz]\n", (double)(hc[2] - hc[0]), (hc[3] - hc[1]) / 100.0,
               (double)(hc[2] - hc[0]) / ((hc[3] - hc[1]) / 100.0));
        const double gf = 2.0 * B * H * W * (double)N * 9 * C / 1e9, mb = ((double)nx + ny * (with_res ? 2 : 1)) * 4 / 1e6;
        printf("B=%d %dx%d %d->%d res=%d f32h2 ws NS %d nt %d: %8.1f us  %7.1f TFLOP/s (fp32-equivalent; %.2f of the f16 pipe)  %6.2f TB/s (alg)  grid %d\n", B, H, W, C, N,
               (int)with_res, NS, g_nt, us, gf / us * 1e3, 3 * gf / us * 1e3 / 2500.0, mb / us, grid);
    }
    hipFree(dx); hipFree(dw); hipFree(dp); hipFree(dy); hipFree(dres); hipFree(db);
    return us;
}

// ---- one HRNet-32 dependency level (the four branch convs) as ONE grid, the way the product launches it -- with the block order and the wave
// priorities as knobs (timing only: random operands, no check):  level B policy prio_mask [reps]
//   policy 0: longest K loop first (the product's order)     1: every problem at the same relative pace (proportional interleave)
//          2: 256- and 128-channel tiles first, then the 64- and 32-channel tiles interleaved     3: ... then all 32-channel tiles, then the 64-channel ones
struct LvlArgs { H2Problem g[8]; };
__global__ __launch_bounds__(256, 3) void lvl_kernel(LvlArgs ga, const int2* __restrict__ map, int prio_mask) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int2 m = map[blockIdx.x];
    if (m.x < 0) return;
    if ((prio_mask >> m.x) & 1) __builtin_amdgcn_s_setprio(2);
    igemm_f32h2_ws_tile<1>(ga.g[m.x], m.y, lds);
#endif
}

static void level(int B, int policy, int prio_mask, int reps) {
    LvlArgs ga{};
    int tiles[8];
    // LVL_SPLITK: bit i = problem i as TWO problems of half the K depth (what a split-K = 2 of its tiles would cost in this grid, without
    // the hand-over); LVL_DROP: bit i = leave problem i out
    const int splitk = getenv("LVL_SPLITK") ? atoi(getenv("LVL_SPLITK")) : 0, drop = getenv("LVL_DROP") ? atoi(getenv("LVL_DROP")) : 0;
    struct Pd { int R, C, N; };
    std::vector<Pd> pd;
    for (int i = 0; i < 4; ++i) {
        if ((drop >> i) & 1) continue;
        const int C = 256 >> i, R = 8 << i;
        if ((splitk >> i) & 1) { pd.push_back({R, C / 2, C}); pd.push_back({R, C / 2, C}); }
        else pd.push_back({R, C, C});
    }
    const int NP = (int)pd.size();
    srand(2);
    auto rnd = [] { return ((rand() & 0xFFFF) * 65536.0 + (rand() & 0xFFFF)) / 4294967296.0 * 2.0 - 1.0; };
    double gf = 0, mb = 0;
    for (int i = 0; i < NP; ++i) {                          // problem 0 = 256 ch 8x8 (longest K) .. problem 3 = 32 ch 64x64
        const int C = pd[i].C, R = pd[i].R, N = pd[i].N;
        H2Problem& p = ga.g[i];
        if (!h2_plan(B, R, R, C, N, 32, &p)) { printf("not eligible\n"); return; }
        const long nx = (long)B * R * R * C, ny = (long)B * R * R * N;
        std::vector<float> hx(nx), hr(ny);
        for (auto& v : hx) v = (float)rnd();
        for (auto& v : hr) v = (float)rnd();
        std::vector<unsigned short> hp((size_t)h2_pack_elems(N, C));
        const long np = h2_piece_elems(N, C);
        for (long k = 0; k < np; ++k) hp[k] = f2h_host((float)(rnd() * ((k / (9 * 32 * 16)) & 1 ? 8.0 : 16384.0)));      // piece 0 large, piece 1 small
        float* hinv = reinterpret_cast<float*>(hp.data() + np);
        for (int n = 0; n < N; ++n) hinv[n] = 1.f / 16384.f / 64.f;
        std::vector<float> hb(N, 0.1f);
        float *dx, *dres, *dy, *db; unsigned short* dp;
        hipMalloc(&dx, nx * 4); hipMalloc(&dres, ny * 4); hipMalloc(&dy, ny * 4); hipMalloc(&db, N * 4); hipMalloc(&dp, hp.size() * 2);
        hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(dres, hr.data(), ny * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
        p.x = dx; p.res = dres; p.y = dy; p.g.bias = db; p.g.wp = dp; p.winv = reinterpret_cast<const float*>(dp + np); p.g.relu = 1;
        tiles[i] = p.g.tiles_m * p.g.NSL;
        gf += 2.0 * B * R * R * (double)N * 9 * C / 1e9; mb += (nx + 2.0 * ny) * 4 / 1e6;
    }
    // per XCD x: its contiguous eighth of every problem's tiles, in the policy's order; block b = k * 8 + x
    std::vector<std::vector<int2>> seq(8);
    for (int x = 0; x < 8; ++x) {
        std::vector<std::vector<int2>> cls(8);
        for (int i = 0; i < NP; ++i) {
            const int per = (tiles[i] + 7) / 8;
            for (int k = 0; k < per; ++k) { const int bid = x * per + k; if (bid < tiles[i]) cls[i].push_back(int2{i, bid}); }
        }
        auto interleave = [&](std::vector<int> which) {     // proportional pace over the classes in `which`
            std::vector<std::pair<double, int2>> all;
            for (int i : which) for (size_t k = 0; k < cls[i].size(); ++k) all.push_back({(k + 0.5) / cls[i].size() + i * 1e-6, cls[i][k]});
            std::stable_sort(all.begin(), all.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            for (auto& e : all) seq[x].push_back(e.second);
        };
        if (policy == 0) { for (int i = 0; i < NP; ++i) for (auto& e : cls[i]) seq[x].push_back(e); }
        else if (policy == 1) interleave({0, 1, 2, 3});
        else if (policy == 2) { for (int i = 0; i < 2; ++i) for (auto& e : cls[i]) seq[x].push_back(e); interleave({2, 3}); }
        else if (policy == 3) { for (int i : {0, 1, 3, 2}) for (auto& e : cls[i]) seq[x].push_back(e); }
        else if (policy == 4) { for (auto& e : cls[0]) seq[x].push_back(e); interleave({1, 2, 3}); }
        else if (policy == 5) { for (int i : {3, 2, 1, 0}) for (auto& e : cls[i]) seq[x].push_back(e); }
    }
    size_t longest = 0;
    for (auto& v : seq) longest = std::max(longest, v.size());
    std::vector<int2> map(longest * 8, int2{-1, 0});
    for (int x = 0; x < 8; ++x) for (size_t k = 0; k < seq[x].size(); ++k) map[k * 8 + x] = seq[x][k];
    int2* dmap; hipMalloc(&dmap, map.size() * sizeof(int2));
    hipMemcpy(dmap, map.data(), map.size() * sizeof(int2), hipMemcpyHostToDevice);
    const size_t lds_bytes = h2_lds_bytes(32);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&lvl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    auto launch = [&]() { hipLaunchKernelGGL(lvl_kernel, dim3((unsigned)map.size()), dim3(256), lds_bytes, 0, ga, dmap, prio_mask); };
    for (int i = 0; i < 3; ++i) launch();
    { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("  kernel error: %s\n", hipGetErrorString(e)); exit(1); } }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("level B=%d policy %d prio mask %d: %7.1f us per launch  %6.1f TFLOP/s (fp32-equivalent; %.3f of the f16 pipe)  %5.2f TB/s (alg)  grid %zu\n", B, policy, prio_mask, us,
           gf / us * 1e3, 3 * gf / us * 1e3 / 2500.0, mb / us, map.size());
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "check");
    if (getenv("X3_NT")) g_nt = atoi(getenv("X3_NT"));
    if (getenv("H2_PP")) g_pp = atoi(getenv("H2_PP"));
    printf("== form: %s\n", g_pp ? "two stages, software-pipelined, one block per CU" : "one stage, two / three blocks per CU");
    if (argc > 4 && !strcmp(argv[1], "level")) { level(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc > 5 ? atoi(argv[5]) : 50); return 0; }
    if (argc > 7 && !strcmp(argv[1], "one")) {            // one B H W C N NS [reps]: a single shape (PMC passes)
        g_ns = atoi(argv[7]);
        run(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), true, false, argc > 8 ? atoi(argv[8]) : 5);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "time")) {
        g_ns = 32;
        for (int B : {64, 512}) {
            run(B, 64, 64, 32, 32, true, false);
            run(B, 32, 32, 64, 64, true, false);
            run(B, 16, 16, 128, 128, true, false);
            run(B, 8, 8, 256, 256, true, false);
        }
        return 0;
    }
    for (int ns = 32; ns <= 64; ns += 32) {
        g_ns = ns;
        run(2, 8, 8, 32, 32, true, true, 0);
        run(3, 16, 16, 48, 48, false, true, 0);
        run(5, 8, 8, 96, 92, true, true, 0);
        run(3, 64, 64, 32, 32, true, true, 0);
        run(2, 32, 32, 64, 64, true, true, 0);
        run(3, 16, 16, 128, 128, true, true, 0);
        run(7, 8, 8, 256, 256, false, true, 0);
        run(2, 24, 18, 64, 64, true, true, 0);
        run(2, 12, 9, 128, 128, true, true, 0);
        run(1, 96, 72, 64, 64, false, true, 0);
        run(2, 64, 48, 32, 32, true, true, 0);
    }
    for (int wide = 1; wide <= 3; ++wide) {
        g_wide = wide; g_ns = 32;
        printf("-- wide mode %d\n", wide);
        run(2, 32, 32, 64, 64, true, true, 0);
        run(2, 8, 8, 96, 92, wide == 1, true, 0);
        run(2, 12, 9, 128, 128, false, true, 0);
    }
    g_wide = 0;
    if (quick) return 0;
    for (int ns = 32; ns <= 64; ns += 32) {
        g_ns = ns;
        for (int B : {64, 512}) {
            double sum = 0;
            sum += run(B, 64, 64, 32, 32, true, false);
            sum += run(B, 32, 32, 64, 64, true, false);
            sum += run(B, 16, 16, 128, 128, true, false);
            sum += run(B, 8, 8, 256, 256, true, false);
            printf("== HRNet-32 level at batch %d, NS %d: sum of the four branches %.1f us = %.1f TFLOP/s (fp32-equivalent)\n", B, ns, sum,
                   4 * 2.0 * B * 4096 * 9 * 1024 / 1e6 / sum);
        }
    }
    return 0;
}
