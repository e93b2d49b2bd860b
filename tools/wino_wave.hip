// EXPERIMENT (not part of libcapf): a WAVE-OWNED Winograd F(4,3)-along-W fp32 3x3 conv -- the structure DESIGN 6 "Next (1)" names
// for the dominant kernel, built stand-alone to measure what it reaches before it replaces anything.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ab/wino_wave tools/wino_wave.hip && tools/ab/wino_wave
// A wave owns 32 tiles (4 output pixels each) x one 32-channel N-tile x all SIX Winograd positions (96 accumulators):
//   * raw pixels: wave-PRIVATE LDS stage, double buffered, filled by LDS-DMA (6 sub-chunks j = 0..5 of 32 tile rows x 16
//     channels per superchunk), no block barrier anywhere -- a wave waits for its own loads (vmcnt) only;
//   * weights: pre-transformed (U = G g) and pre-packed in MFMA fragment order, one contiguous 1 KiB buffer_load_dwordx4 per
//     (kh, k-step, position, N-tile), prefetched one k-step ahead in registers;
//   * 24 MFMAs per k-step (6 positions x 4), scheduler fenced per position.
// Epilogue here: output transform + plain 16-byte stores in the accumulator layout (no bias / residual / ReLU: an experiment).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct Prob {
    const float* x;      // [B][H][W][C]
    const float* wf;     // packed fragments: [kh][C/8][6][N/32][64 lanes][4]
    float* y;            // [B][H][W][N]
    int B, H, W, C, N;
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256, 1) void wino_wave_kernel(Prob p, int n_items) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int STAGE = 6 * 32 * 16;                 // floats per wave stage: 6 sub-chunks x 32 tiles x 16 channels (12 KiB)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* mine = lds + wave * (2 * STAGE);
    const int WT = p.W / 4, tiles = p.B * p.H * WT, tgroups = (tiles + 31) / 32, NT = p.N / 32, KS = p.C / 8, NSC = p.C / 16;
    const int frow = lane & 31, fhalf = lane >> 5, fsw = (frow >> 2) & 3;
    // descriptor base one image row and one pixel BEFORE the tensor: every (kh, j) offset is then non-negative and a pure scalar
    // (nothing is read there: halo lanes get an out-of-range offset)
    const rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - (long)(p.W + 1) * p.C), 0, 0x7FFFFF00u, 0x00020000);
    const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wf, 0, (unsigned)((long)18 * p.C * p.N * 4), 0x00020000);

    for (int item = blockIdx.x * 4 + wave; item < n_items; item += gridDim.x * 4) {
        const int nt = item / tgroups, tg = item - nt * tgroups;      // consecutive waves: same N-tile (its weight fragments hit in L1)
        // ---- DMA addressing: sub-chunk j, rows = 32 tiles, 4 quads (16 channels) per row: 128 lanes -> 2 instructions per sub-chunk.
        // lane -> (row r = (i * 64 + lane) / 4, logical quad q = lane & 3); LDS is written linearly, so the GLOBAL quad fetched into
        // LDS slot (r, qs) is q = qs ^ ((r >> 2) & 3).
        unsigned voff[2], bad[2];                      // byte offset of (tile row, quad) from the shifted base; bit kh * 6 + j set = halo / no tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (i * 64 + lane) >> 2, qs = lane & 3, q = qs ^ ((r >> 2) & 3);
            const int t = tg * 32 + r;
            const int tt = t < tiles ? t : 0;
            const int row = tt / WT, wt = tt - row * WT, h = row % p.H;     // row = b * H + h
            voff[i] = (unsigned)(((long)row * p.W + 4 * wt) * p.C + q * 4) * 4u;
            unsigned m = 0;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int wcol = 4 * wt - 1 + j, hrow = h + kh - 1;
                    if (!(t < tiles && wcol >= 0 && wcol < p.W && hrow >= 0 && hrow < p.H)) m |= 1u << (kh * 6 + j);
                }
            bad[i] = m;
        }
        f32x16 acc[6];
#pragma unroll
        for (int pp = 0; pp < 6; ++pp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pp][r] = 0.f;
        // weights of k-step (kh, s): 6 fragments, 1 KiB each, contiguous
        // ---- K loop: superchunk u = (kh, 16 channels), two k-steps each, two superchunks per trip (stage / register slots are
        // compile-time).  At the top of superchunk u the DMA and the 12 weight fragments of superchunk u + 1 are requested; the A
        // fragments of every k-step are read and transformed one k-step ahead, a piece behind each position's four MFMAs.  All
        // offsets advance by scalar adds (no division, no per-lane address arithmetic in the loop).
        const int total_sc = 3 * NSC;                  // (C >= 32: an even number)
        f32x4 wb[2][2][6];                             // [superchunk parity][k-step][position]
        f32x4 d[6], va[6], vb[6];
        const unsigned w_lane = (unsigned)lane * 16u;
        const unsigned w_step = (unsigned)(6 * NT) * 1024u, w_pos = (unsigned)NT * 1024u;      // bytes per k-step / per position
        const unsigned cstep = (unsigned)p.C * 4u;
        unsigned w_so = (unsigned)nt * 1024u;          // weights: byte offset of (k-step, position 0, this N-tile); k-steps are consecutive
        unsigned x_so = 0;                             // pixels: byte offset of (kh, sc) = (kh * W * C + sc * 16) * 4
        int x_sc = 0, x_kh = 0;                        // ... of the NEXT superchunk to request
        auto req_w_piece = [&](auto par, int ks, int pp) {          // one weight fragment of the next superchunk
            constexpr int P = decltype(par)::value;
            wb[P][ks][pp] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, w_so + (unsigned)ks * w_step + (unsigned)pp * w_pos, 0));
        };
        auto req_x_piece = [&](auto par, bool live, int j) {        // the two DMA instructions of sub-chunk j of the next superchunk
            constexpr int P = decltype(par)::value;
            float* st = mine + P * STAGE;
            const int sh = x_kh * 6;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned dead = live ? ((bad[i] >> (sh + j)) & 1u) : 1u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(st + j * 512 + i * 256), 16, voff[i] | (dead << 31), x_so + (unsigned)j * cstep, 0, 0);
            }
        };
        auto adv_x = [&]() { if (++x_sc == NSC) { x_sc = 0; ++x_kh; x_so = (unsigned)(x_kh * p.W) * cstep; } else x_so += 64u; };
        auto read_d = [&](auto par, int ks) {
            constexpr int P = decltype(par)::value;
            const float* st = mine + P * STAGE;
#pragma unroll
            for (int j = 0; j < 6; ++j) d[j] = *reinterpret_cast<const f32x4*>(&st[j * 512 + frow * 16 + (((ks * 2 + fhalf) ^ fsw) * 4)]);
        };
        auto transform = [&](f32x4 (&o)[6], int lo, int hi) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = d[0][e], x1 = d[1][e], x2 = d[2][e], x3 = d[3][e], x4 = d[4][e], x5 = d[5][e];
                const float a = x3 - x1, b = x4 - x2;
                if (lo <= 0 && 0 < hi) o[0][e] = fmaf(-5.f, x2, fmaf(4.f, x0, x4));
                if (lo <= 1 && 1 < hi) o[1][e] = fmaf(-4.f, x1 + x2, x3 + x4);
                if (lo <= 2 && 2 < hi) o[2][e] = fmaf(4.f, x1 - x2, x4 - x3);
                if (lo <= 3 && 3 < hi) o[3][e] = fmaf(2.f, a, b);
                if (lo <= 4 && 4 < hi) o[4][e] = fmaf(-2.f, a, b);
                if (lo <= 5 && 5 < hi) o[5][e] = fmaf(-5.f, x3, fmaf(4.f, x1, x5));
            }
        };
        // k-step: 6 positions x 4 MFMAs; behind each position's MFMAs a piece of the A path of the next k-step and a piece of the
        // requests for the next superchunk (first k-step: its 12 DMA instructions, second k-step: its 12 weight fragments)
        auto kstep = [&](const f32x4 (&w)[6], const f32x4 (&v)[6], f32x4 (&vnext)[6], auto next_stage, int next_ks, auto req_par, bool more,
                         int which) {
#pragma unroll
            for (int pp = 0; pp < 6; ++pp) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[pp][e], v[pp][e], acc[pp], 0, 0, 0);
                if (pp == 0) read_d(next_stage, next_ks);
                if (pp == 2) transform(vnext, 0, 2);
                if (pp == 3) transform(vnext, 2, 4);
                if (pp == 4) transform(vnext, 4, 6);
                if (which == 0) req_x_piece(req_par, more, pp);
                else { req_w_piece(req_par, 0, pp); req_w_piece(req_par, 1, pp); }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto superchunk = [&](auto par, bool more) {   // superchunk in stage / slots P; `more`: another one follows
            constexpr int P = decltype(par)::value;
            // in flight here, oldest first: W(P) 12 (requested during the previous superchunk's second k-step).  Its pixels landed
            // before that k-step read them.
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // first 6 of W(P): fragments are requested (k-step 0, 1) interleaved -> all 12
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            kstep(wb[P][0], va, vb, par, 1, std::integral_constant<int, P ^ 1>{}, more, 0);     // + DMA of the next superchunk
            adv_x();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the next superchunk's pixels (read during the k-step below)
            __builtin_amdgcn_sched_barrier(0);
            kstep(wb[P][1], vb, va, std::integral_constant<int, P ^ 1>{}, 0, std::integral_constant<int, P ^ 1>{}, more, 1);   // + its weights
            w_so += 2 * w_step;
        };
        // prologue: pixels and weights of superchunk 0, then its first k-step's fragments
#pragma unroll
        for (int j = 0; j < 6; ++j) req_x_piece(std::integral_constant<int, 0>{}, true, j);
        adv_x();
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) { req_w_piece(std::integral_constant<int, 0>{}, 0, pp); req_w_piece(std::integral_constant<int, 0>{}, 1, pp); }
        w_so += 2 * w_step;
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        read_d(std::integral_constant<int, 0>{}, 0);
        transform(va, 0, 6);
        for (int u = 0; u < total_sc; u += 2) {
            superchunk(std::integral_constant<int, 0>{}, true);
            superchunk(std::integral_constant<int, 1>{}, u + 2 < total_sc);
        }
        // ---- output transform (A^T: 4 x 6) and stores in the accumulator layout: lane = tile, registers 4 g + e = channel 8 g + 4 fhalf + e
        const int t = tg * 32 + frow;
        if (t < tiles) {
            const int row = t / WT, wt = t - row * WT;
            float* yb = p.y + ((long)row * p.W + 4 * wt) * p.N + nt * 32 + 4 * fhalf;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o0, o1, o2, o3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float m0 = acc[0][4 * g + e], m1 = acc[1][4 * g + e], m2 = acc[2][4 * g + e], m3 = acc[3][4 * g + e],
                                m4 = acc[4][4 * g + e], m5 = acc[5][4 * g + e];
                    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                    o0[e] = m0 + s12 + s34;
                    o1[e] = d12 + 2.f * d34;
                    o2[e] = s12 + 4.f * s34;
                    o3[e] = d12 + 8.f * d34 + m5;
                }
                *reinterpret_cast<f32x4*>(yb + 0 * p.N + 8 * g) = o0;
                *reinterpret_cast<f32x4*>(yb + 1 * p.N + 8 * g) = o1;
                *reinterpret_cast<f32x4*>(yb + 2 * p.N + 8 * g) = o2;
                *reinterpret_cast<f32x4*>(yb + 3 * p.N + 8 * g) = o3;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

__global__ void direct_ref(const float* x, const float* w, float* y, int B, int H, int W, int C, int N) {   // w: [3][3][C][N]
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
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
            const float* wp = w + ((long)(kh * 3 + kw) * C) * N + n;
            for (int c = 0; c < C; ++c) s += (double)xp[c] * wp[(long)c * N];
        }
    y[i] = (float)s;
}

static void run(int B, int H, int W, int C, int N, bool check) {
    const long nx = (long)B * H * W * C, ny = (long)B * H * W * N, nw = 9L * C * N;
    std::vector<float> hx(nx), hw(nw), hf(18L * C * N);
    srand(1);
    for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 3000.f;
    static const double G[6][3] = {{0.25, 0, 0}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6}, {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0, 0, 1}};
    const int KS = C / 8, NT = N / 32;
    for (int kh = 0; kh < 3; ++kh)
        for (int s = 0; s < KS; ++s)
            for (int pp = 0; pp < 6; ++pp)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int n = nt * 32 + (lane & 31), c = s * 8 + 4 * (lane >> 5) + e;
                            double u = 0;
                            for (int kw = 0; kw < 3; ++kw) u += G[pp][kw] * hw[((long)(kh * 3 + kw) * C + c) * N + n];
                            hf[((((long)(kh * KS + s) * 6 + pp) * NT + nt) * 64 + lane) * 4 + e] = (float)u;
                        }
    float *dx, *dw, *df, *dy, *dr;
    hipMalloc(&dx, nx * 4); hipMalloc(&dw, nw * 4); hipMalloc(&df, hf.size() * 4); hipMalloc(&dy, ny * 4); hipMalloc(&dr, ny * 4);
    hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemcpy(df, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dy, 0, ny * 4);
    Prob p{dx, df, dy, B, H, W, C, N};
    const int tiles = B * H * (W / 4), n_items = ((tiles + 31) / 32) * NT;
    const size_t lds_bytes = 4 * 2 * 6 * 32 * 16 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    const int grid = n_items / 4 < 256 ? (n_items + 3) / 4 : 256;
    hipLaunchKernelGGL(wino_wave_kernel, dim3(grid), dim3(256), lds_bytes, 0, p, n_items);
    hipDeviceSynchronize();
    if (check) {
        hipLaunchKernelGGL(direct_ref, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, 0, dx, dw, dr, B, H, W, C, N);
        std::vector<float> a(ny), r(ny);
        hipMemcpy(a.data(), dy, ny * 4, hipMemcpyDeviceToHost);
        hipMemcpy(r.data(), dr, ny * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (long i = 0; i < ny; ++i) { worst = fmax(worst, fabs((double)a[i] - r[i])); scale = fmax(scale, fabs((double)r[i])); }
        printf("  check B=%d %dx%d C=%d N=%d: max |wino - direct| = %.3e (max |direct| %.3f) %s\n", B, H, W, C, N, worst, scale,
               worst <= 3e-5 * scale + 1e-6 ? "OK" : "MISMATCH");
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wino_wave_kernel, dim3(grid), dim3(256), lds_bytes, 0, p, n_items);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wino_wave_kernel, dim3(grid), dim3(256), lds_bytes, 0, p, n_items);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, gf = 2.0 * B * H * W * (double)N * 9 * C / 1e9;
    const double alg_tf = gf / us * 1e3;             // GFLOP / us = 1000 TFLOP/s
    printf("B=%d %dx%d %d->%d: %8.1f us  %7.1f TFLOP/s algorithmic  %6.1f executed (%.2f of the fp32 MFMA peak)  grid %d, %d items\n", B, H, W, C, N, us,
           alg_tf, alg_tf / 2, alg_tf / 2 / 157.3, grid, n_items);
    hipFree(dx); hipFree(dw); hipFree(df); hipFree(dy); hipFree(dr);
}

int main() {
    run(2, 16, 16, 32, 32, true);
    run(3, 8, 8, 64, 64, true);
    run(2, 32, 32, 16, 32, true);
    run(64, 64, 64, 32, 32, false);
    run(64, 32, 32, 64, 64, false);
    run(64, 16, 16, 128, 128, false);
    run(64, 8, 8, 256, 256, false);
    return 0;
}
