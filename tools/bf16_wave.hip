// EXPERIMENT (not part of libcapf): a wave-owned, WEIGHT-IN-REGISTERS bf16 3x3 / stride-1 conv for the narrow HRNet-48 branch
// (48 -> 48 channels at 64 x 64), stand-alone with its own direct-conv check.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ab/bf16_wave tools/bf16_wave.hip && tools/ab/bf16_wave
// The row-halo tile of the library spends 1.5 ds_read_b128 per MFMA and re-stages its weights for every tile; at 48 channels the
// whole filter is 9 x 48 x 64 (padded) bf16 = 54 MFMA B-fragments = 216 VGPRs -- a lone wave per SIMD can simply KEEP them.
// A wave owns 32 consecutive output pixels of an image row x all 48 (64) channels: three input rows x 34 pixels x 96 B go through a
// wave-private double-buffered LDS stage (10 LDS-DMA instructions per tile, the next tile's in flight under this tile's 54
// MFMAs, one request behind every 2-3 MFMA steps), 27 A-fragment reads (a ring four steps deep) + 54 MFMAs per tile, stores straight from
// the accumulator layout.  Measured at batch 256 (64 x 64): 67.7 us = 643 TFLOP/s, 3.0 TB/s -- the library's row-halo tile alone: 106 us =
// 411 TFLOP/s; per tile 2750 cycles of MFMAs + DMA issue (the 54 MFMAs are 1728; a DMA instruction costs the wave ~100 cycles) and 390 of
// epilogue.  Not in the library: inside a grouped HRNet level the narrow branches overlap the wide ones' MFMA time, and a kernel with
// one 345-register wave per SIMD and 80 KiB of LDS leaves room for only one row-halo block per CU beside it.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

static inline unsigned short f2bf_host(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf2f_host(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const bf2 r = __builtin_convertvector(f2{lo, hi}, bf2);
    return __builtin_bit_cast(unsigned, r);
}

struct Prob {
    const unsigned short* x;   // [B][H][W][48] bf16
    const unsigned short* wf;  // fragments: [kh][kw][s = 0..2][jn = 0..1][64 lanes][8] bf16 (1 KiB each), N padded to 64
    unsigned short* y;         // [B][H][W][48] bf16
    int B, H, W;
    unsigned long long* dbg;
};

constexpr int C = 48, PXB = 96;                         // channels, bytes per pixel
constexpr int STAGE_B = 3 * 34 * PXB;                   // 9792 B: three rows of 34 pixels
constexpr int STAGE_PAD = 10240;                        // 10 DMA instructions of 1 KiB
constexpr int EPS = 36;

__global__ __launch_bounds__(256, 1) void bf16_wave_kernel(Prob p, int ntiles) {
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* mine = lds + wave * (2 * STAGE_PAD + 32 * EPS * 4);
    float* ep = reinterpret_cast<float*>(mine + 2 * STAGE_PAD);
    const int frow = lane & 31, fhalf = lane >> 5;
    const int TPR = p.W / 32;                           // tiles per image row
    // descriptor one row + one pixel before the tensor: all halo offsets non-negative
    const rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - (long)(p.W + 1) * C), 0, 0x7FFFFF00u, 0x00020000);
    const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wf, 0, 54u * 1024u, 0x00020000);

    // ---- the whole filter, once: 54 fragments
    u32x4 wr[27][2];
#pragma unroll
    for (int f = 0; f < 27; ++f)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) wr[f][jn] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)lane * 16u, (unsigned)(f * 2 + jn) * 1024u, 0);

    // ---- per-lane DMA geometry (tile independent): instruction i, lane -> linear quad Q = 64 i + lane of the stage:
    // stage row pr = Q / 6 (0..101: input row r = pr / 34, pixel c = pr % 34 = image column w0 - 1 + c), slot qs = Q % 6 holds quad qs ^ ((pr >> 3) & 1)
    unsigned q_off[10];                                 // byte offset inside (r * W + c) * 96 + quad * 16 relative to the tile's (row - 1, w0 - 1) pixel... from the shifted base
    int q_r[10], q_c[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int Q = i * 64 + lane, pr = Q / 6, qs = Q - pr * 6, q = qs ^ ((pr >> 3) & 1);
        const int r = pr / 34, c = pr - r * 34;
        q_r[i] = pr < 102 ? r : 99;                     // (beyond the stage: never valid)
        q_c[i] = c;
        q_off[i] = (unsigned)((r * p.W + c) * PXB + q * 16);
    }
    // scalar state of the tile being requested
    unsigned rq_so = 0; int rq_h = 0, rq_w0 = 0; bool rq_live = false;
    auto request_begin = [&](int tile) {
        rq_live = tile < ntiles;
        const int tt = rq_live ? tile : 0;
        const int row = tt / TPR;
        rq_w0 = (tt - row * TPR) * 32;
        rq_h = row % p.H;
        rq_so = __builtin_amdgcn_readfirstlane((unsigned)((row * p.W + rq_w0) * PXB));
    };
    auto request_piece = [&](int i, int buf) {        // DMA instruction i (0..9) of the tile set up by request_begin
        const int hh = rq_h - 1 + q_r[i], ww = rq_w0 - 1 + q_c[i];
        const bool ok = rq_live && q_r[i] < 3 && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(mine + buf * STAGE_PAD + i * 1024), 16, ok ? q_off[i] : OOB, rq_so, 0, 0);
    };
    auto request = [&](int tile, int buf) {
        request_begin(tile);
#pragma unroll
        for (int i = 0; i < 10; ++i) request_piece(i, buf);
    };

    int tile = blockIdx.x * 4 + wave;
    const int step = gridDim.x * 4;
    request(tile, 0);
    {   // three dropped stores: the loop is entered with "10 DMA, then 3 stores" in flight like its back edge leaves
        const rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0u, 0x00020000);
#pragma unroll
        for (int k = 0; k < 6; ++k) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rs_none, OOB + 16u * k, 0, 0);
    }
    int buf = 0;
    unsigned long long ph[4] = {0, 0, 0, 0}, t0, t1;
    for (; tile < ntiles; tile += step, buf ^= 1) {
        t0 = __builtin_amdgcn_s_memtime();
        request_begin(tile + step);
        t1 = __builtin_amdgcn_s_memtime(); ph[0] += t1 - t0; t0 = t1;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");            // this tile's 10 DMA landed (requested under the previous tile's MFMAs; its 3 stores may fly)
        t1 = __builtin_amdgcn_s_memtime(); ph[1] += t1 - t0; t0 = t1;
        const unsigned char* st = mine + buf * STAGE_PAD;
        f32x16 acc[2];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jn][r] = 0.f;
        auto a_frag = [&](int kh, int kw, int s) {
            const int pr = kh * 34 + frow + kw;
            return *reinterpret_cast<const u32x4*>(st + pr * PXB + (((2 * s + fhalf) ^ ((pr >> 3) & 1)) * 16));
        };
        // fragments are read FOUR steps ahead: a step is two bf16 MFMAs = 64 cycles, an LDS read takes ~130
        auto frag_of = [&](int f) { const int ff = f < 27 ? f : 26; return a_frag(ff / 9, (ff / 3) % 3, ff % 3); };
        u32x4 ring[4] = {frag_of(0), frag_of(1), frag_of(2), frag_of(3)};
#pragma unroll
        for (int f = 0; f < 27; ++f) {
            const u32x4 af = ring[f & 3];
            ring[f & 3] = frag_of(f + 4);
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wr[f][jn]), __builtin_bit_cast(bf16x8, af), acc[jn], 0, 0, 0);
            // the next tile's ten DMA instructions, one behind steps 1, 3, 6, 8, 11, 13, 16, 18, 21, 23 (issuing one costs the wave ~130
            // cycles: two bf16 MFMAs keep the matrix pipe busy for 64 of them)
            if (f % 5 == 1 && f < 25) request_piece(2 * (f / 5), buf ^ 1);
            if (f % 5 == 3 && f < 25) request_piece(2 * (f / 5) + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 0" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime(); ph[2] += t1 - t0; t0 = t1;
        // ---- epilogue: transposed accumulator (lane = pixel, register 4 g + e = channel 8 g + 4 fhalf + e of N-tile jn) -> LDS -> 16-byte stores;
        // the 32 pixels x 96 B of a tile are contiguous in memory: quad Q' = 64 i + lane (i = 0..2) -> pixel Q' / 6, channels 8 (Q' % 6)
        const int row = tile / TPR, w0 = (tile - row * TPR) * 32;
        unsigned short* yb = p.y + ((long)row * p.W + w0) * C;
        // direct stores in the accumulator layout: lane = pixel, 4 consecutive channels (8 B) per register group; the two halves of a
        // wave complete 16 B per pixel and instruction (six instructions per tile; a transpose through LDS cost 1360 cycles per tile)
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int g = 0; g < (jn == 0 ? 4 : 2); ++g)
                *reinterpret_cast<u32x2*>(yb + frow * C + jn * 32 + 8 * g + 4 * fhalf) =
                    u32x2{pack2(acc[jn][4 * g], acc[jn][4 * g + 1]), pack2(acc[jn][4 * g + 2], acc[jn][4 * g + 3])};
        t1 = __builtin_amdgcn_s_memtime(); ph[3] += t1 - t0;
    }
    if (p.dbg && blockIdx.x == 7 && tid == 0) { for (int k = 0; k < 4; ++k) p.dbg[k] = ph[k]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void direct_ref(const unsigned short* x, const unsigned short* w, float* y, int B, int H, int W) {     // w: [3][3][48][48] bf16
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * H * W * C) return;
    const int n = i % C;
    const long px = i / C;
    const int wc = px % W, h = (px / W) % H, b = px / ((long)W * H);
    float s = 0.f;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            const int hh = h + kh - 1, ww = wc + kw - 1;
            if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
            const unsigned short* xp = x + (((long)b * H + hh) * W + ww) * C;
            const unsigned short* wp = w + ((long)(kh * 3 + kw) * C) * C + n;
            for (int c = 0; c < C; ++c) s += __uint_as_float((unsigned)xp[c] << 16) * __uint_as_float((unsigned)wp[(long)c * C] << 16);
        }
    y[i] = s;
}

static void run(int B, int H, int W, bool check) {
    const long nx = (long)B * H * W * C, nw = 9L * C * C;
    std::vector<unsigned short> hx(nx), hw(nw), hf(54 * 512, 0);
    srand(1);
    for (auto& v : hx) v = f2bf_host((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hw) v = f2bf_host((rand() % 2001 - 1000) / 3000.f);
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw)
            for (int s = 0; s < 3; ++s)
                for (int jn = 0; jn < 2; ++jn)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int n = jn * 32 + (lane & 31), c = s * 16 + 8 * (lane >> 5) + e;
                            const int f = (kh * 3 + kw) * 3 + s;
                            hf[((long)(f * 2 + jn) * 64 + lane) * 8 + e] = n < C ? hw[((long)(kh * 3 + kw) * C + c) * C + n] : 0;
                        }
    unsigned short *dx, *dw, *df, *dy;
    float* dr;
    hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&df, hf.size() * 2); hipMalloc(&dy, nx * 2); hipMalloc(&dr, nx * 4);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemcpy(df, hf.data(), hf.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dy, 0, nx * 2);
    unsigned long long* ddbg; hipMalloc(&ddbg, 64); hipMemset(ddbg, 0, 64);
    Prob p{dx, df, dy, B, H, W, ddbg};
    const int ntiles = B * H * (W / 32);
    const size_t lds_bytes = 4 * (2 * STAGE_PAD + 32 * EPS * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bf16_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    const int grid = ntiles / 4 < 256 ? (ntiles + 3) / 4 : 256;
    hipLaunchKernelGGL(bf16_wave_kernel, dim3(grid), dim3(256), lds_bytes, 0, p, ntiles);
    { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) printf("  kernel error: %s\n", hipGetErrorString(e)); }
    if (check) {
        hipLaunchKernelGGL(direct_ref, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, 0, dx, dw, dr, B, H, W);
        std::vector<unsigned short> a(nx);
        std::vector<float> r(nx);
        hipMemcpy(a.data(), dy, nx * 2, hipMemcpyDeviceToHost);
        hipMemcpy(r.data(), dr, nx * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (long i = 0; i < nx; ++i) { worst = fmax(worst, fabs((double)bf2f_host(a[i]) - r[i])); scale = fmax(scale, fabs((double)r[i])); }
        printf("  check B=%d %dx%d: max |wave - direct| = %.3e (max |direct| %.3f) %s\n", B, H, W, worst, scale, worst <= 1e-2 * scale ? "OK" : "MISMATCH");
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(bf16_wave_kernel, dim3(grid), dim3(256), lds_bytes, 0, p, ntiles);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(bf16_wave_kernel, dim3(grid), dim3(256), lds_bytes, 0, p, ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, gf = 2.0 * B * H * W * (double)C * 9 * C / 1e9, mb = 2.0 * nx * 2 / 1e6;
    { unsigned long long h[4]; hipMemcpy(h, ddbg, 32, hipMemcpyDeviceToHost); const int tiles_pw = (ntiles + grid * 4 - 1) / (grid * 4);
      printf("  cycles per tile (s_memtime, one wave): request setup %llu  wait %llu  MFMAs + interleaved DMA issue %llu  epilogue %llu  (tiles per wave %d)\n", h[0] / tiles_pw, h[1] / tiles_pw, h[2] / tiles_pw, h[3] / tiles_pw, tiles_pw); }
    printf("B=%d %dx%d 48->48 bf16: %8.1f us  %7.1f TFLOP/s  %6.2f TB/s (in + out)  grid %d, %d tiles\n", B, H, W, us, gf / us * 1e3, mb / us, grid, ntiles);
    hipFree(dx); hipFree(dw); hipFree(df); hipFree(dy); hipFree(dr);
}

int main() {
    run(2, 8, 64, true);
    run(3, 64, 64, true);
    run(24, 64, 64, true);
    run(256, 64, 64, false);
    return 0;
}
