// Two pointwise (1x1, stride 1) fp32 convs back to back in ONE kernel -- the end of an HRNet `layer1` bottleneck and the start
// of the next one (pose_hrnet.py:98-136):
//     y  = relu(W3 . t + b3 + res)      64 -> 256 channels, stored (it is the next block's residual)
//     t' = relu(W1 . y + b1)            256 -> 64 channels, stored
// As two launches y (268 MB at batch 64) is written by the first and read back by the second, and both are short-K kernels
// that live on prologue and epilogue (igemm_f32_pw.hip): 130 + 99 us at batch 64 against 119 us of matrix time and 146 us of
// compulsory HBM time for the pair.  This kernel: 177.5 us (without the residual operand it runs at 150 TFLOP/s = 0.96 of the
// fp32 MFMA peak: what is left is waiting for residual rows).
//
// Here a WAVE owns 32 pixels through both convs and never meets another wave:
//   * both weight matrices (64 KiB each) are staged into LDS once per persistent block, in the 32-float sub-chunk layout with
//     the quad XOR swizzle of the other fp32 kernels (conflict-free ds_read_b128 fragments);
//   * the A operand of the first conv is read from global memory straight into fragment registers (a lane needs 16
//     consecutive bytes of its row per k-step: eight loads per 32-row tile), one tile ahead;
//   * the first conv's accumulators ARE the second conv's A operand: with D = mfma(B-fragment, A-fragment) the accumulator
//     register 4 g + e of N-tile j holds channel 32 j + 8 g + 4 (lane >> 5) + e of the lane's row -- exactly the k-pair
//     (k, k + 4) layout a ds_read_b128-fed k-step 4 j + g consumes -- so y goes through bias / residual / ReLU in registers,
//     is stored, and feeds the second K loop without touching LDS;
//   * residual rows are requested as a trickle under the MFMAs -- half of a tile's during the second K loop of the PREVIOUS tile,
//     half during its own first k-steps (all of it that early does not fit: y and the residual both live in the 256
//     architectural VGPRs) -- in a coalesced layout (8 rows x 128 B per instruction) and meet the accumulators through a
//     4.5 KiB per-wave LDS transpose; y and t' leave the same way.
// No barrier after the weight staging, no LDS traffic but the B fragments (one ds_read_b128 per 4 MFMAs), every global access
// a raw buffer access with an out-of-range offset instead of a branch.  One wave per SIMD (471 registers, no spill; the
// scheduler is fenced per MFMA group -- left alone it hoists dozens of fragment reads and spills 500 registers).
// Same K order and the same ((acc + bias) + res) epilogue as igemm_f32_pw / igemm_f32: bit-identical results.
#include "kernels.h"

namespace capf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int K1, int N1, int N2>
__global__ __launch_bounds__(256, 1) void igemm_f32_pwchain_kernel(GemmArgs p3, GemmArgs p1, int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT1 = N1 / 32, NT2 = N2 / 32, ST1 = K1 / 8;     // N-tiles of the two convs, k-steps of the first
    constexpr int W3F = N1 * K1, W1F = N2 * N1;                    // floats of the two weight matrices
    constexpr unsigned OOB = 0x80000000u;
    static_assert(K1 % 32 == 0 && N1 % 32 == 0 && N2 == 64, "whole 32-float sub-chunks; eight MFMA groups per sub-chunk of the second conv");
    constexpr int EPS = 36;                                        // padded row of a wave's 32 x 32 transpose scratch
    extern __shared__ __attribute__((aligned(16))) float lds[];   // W3 | W1 | b3 | b1 | 4 x transpose scratch
    float* W3s = lds;
    float* W1s = lds + W3F;
    float* B3s = W1s + W1F;
    float* B1s = B3s + N1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weights and biases -> LDS, once.  [N][K] row-major global -> sub-chunk s = k / 32: [N][32] with quad q ^ ((n >> 1) & 7)
    auto stage = [&](const float* W, float* dst, int N, int K) {
        const int qpr = K / 4;
        for (int i = tid; i < N * qpr; i += 256) {
            const int n = i / qpr, q = i - n * qpr;
            const int sub = q >> 3, qq = q & 7;
            *reinterpret_cast<f32x4*>(dst + (size_t)sub * N * 32 + n * 32 + ((qq ^ ((n >> 1) & 7)) * 4)) =
                *reinterpret_cast<const f32x4*>(W + (size_t)n * K + q * 4);
        }
    };
    stage(p3.Wp, W3s, N1, K1);
    stage(p1.Wp, W1s, N2, N1);
    for (int i = tid; i < N1; i += 256) B3s[i] = p3.bias[i];
    for (int i = tid; i < N2; i += 256) B1s[i] = p1.bias[i];
    __syncthreads();

    float* ep = B1s + N2 + wave * (32 * EPS);
    const int frow = lane & 31, fhalf = lane >> 5, fsw = (frow >> 1) & 7;
    const int er = lane >> 3, ec = (lane & 7) * 4;                 // coalesced layout: 8 lanes x 16 B = the 128 B of one row of an N-tile
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;      // this wave among all waves: tiles gw, gw + nw, ...
    const long s_a = p3.K, s_r = p3.rmap.S1, s_y = p3.omap.S1, s_t = p1.omap.S1;

    f32x4 a1[ST1];                       // A fragments of the first conv: k = 8 step + 4 fhalf .. + 3 of row frow
    f32x4 res[NT1][4];                   // residual, COALESCED layout: rows 8 h + er, channels 32 j + ec .. + 3 (transposed through LDS
                                         // in the epilogue: in the accumulator layout an instruction touches 32 rows x 32 B, and
                                         // the pair ran at 3.4 TB/s -- 83 of its 197 us were these loads)
    auto request_a = [&](int tile) {
        const long m0 = (long)tile * 32;
        const bool ok = tile < ntiles && m0 + frow < p3.M;
        const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p3.A + (tile < ntiles ? m0 : 0) * s_a), 0, 0x7FFFFF00u, 0x00020000);
#pragma unroll
        for (int st = 0; st < ST1; ++st)
            a1[st] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rs, ok ? (unsigned)(frow * (int)s_a + st * 8 + fhalf * 4) * 4u : OOB, 0, 0));
    };
    auto request_res = [&](int tile, int j) {        // the four 8-row pieces of N-tile j
        const long m0 = (long)tile * 32;
        const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p3.res + (tile < ntiles ? m0 : 0) * s_r + p3.rmap.off), 0, 0x7FFFFF00u,
                                                            0x00020000);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const bool ok = tile < ntiles && m0 + 8 * h + er < p3.M;
            res[j][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rs, ok ? (unsigned)((8 * h + er) * (int)s_r + 32 * j + ec) * 4u : OOB, 0, 0));
        }
    };

    int tile = gw;
    // The 1024 waves start together and do identical work: left alone they all store y at the same moment and all sit in a K loop
    // (no memory traffic) at the same moment.  Wave w of a block starts w x 6400 cycles (a quarter of a K loop) late: 185 -> 177.5 us
    // per pair at batch 64, 674 -> 662 us at batch 256.
    for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(100);
    request_a(tile);
#pragma unroll
    for (int j = 0; j < NT1 / 2; ++j) request_res(tile, j);
    for (; tile < ntiles; tile += nw) {
        const long m0 = (long)tile * 32;
        // ---- first conv: 32 rows x N1 channels, K1 deep
        f32x16 y[NT1];
#pragma unroll
        for (int j = 0; j < NT1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[j][r] = 0.f;
        // (fragments are read one MFMA group ahead and the scheduler is fenced per group: left alone it hoists dozens of
        // ds_read_b128 to the top of the unrolled loop and spills them)
        auto w3_frag = [&](int st, int j) {
            return *reinterpret_cast<const f32x4*>(&W3s[(st >> 2) * (N1 * 32) + (32 * j + frow) * 32 + (((st & 3) * 2 + fhalf) ^ fsw) * 4]);
        };
        {
            f32x4 bf = w3_frag(0, 0);
#pragma unroll
            for (int st = 0; st < ST1; ++st)
#pragma unroll
                for (int j = 0; j < NT1; ++j) {
                    const int jn = j + 1 < NT1 ? j + 1 : 0, sn = j + 1 < NT1 ? st : (st + 1 < ST1 ? st + 1 : st);
                    const f32x4 nx = w3_frag(sn, jn);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[e], a1[st][e], y[j], 0, 0, 0);
                    bf = nx;
                    if (j == NT1 - 1 && st < NT1 / 2) request_res(tile, NT1 / 2 + st);    // second half of this tile's residual
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        // ---- y = relu((acc + b3) + res), stored, and kept as the second conv's A operand
        const rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(p3.out + m0 * s_y + p3.omap.off), 0, 0x7FFFFF00u, 0x00020000);
        // The epilogue of N-tile j in seven pieces; N-tile 0 runs here, N-tile j + 1 rides behind the eight MFMA groups of the second
        // conv's sub-chunk j (which only needs N-tiles <= j): the VALU / LDS / store work of 7/8 of y's epilogue issues under MFMAs.
        auto epi_piece = [&](int j, int piece) {
            if (piece == 0) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int h = 0; h < 4; ++h) *reinterpret_cast<f32x4*>(&ep[(8 * h + er) * EPS + ec]) = res[j][h];
                __builtin_amdgcn_wave_barrier();
            } else if (piece <= 4) {
                const int g = piece - 1;
                float* cell = &ep[frow * EPS + 8 * g + 4 * fhalf];          // this lane's 4 channels, accumulator layout
                const f32x4 rv = *reinterpret_cast<const f32x4*>(cell);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&B3s[32 * j + 8 * g + 4 * fhalf]);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = (y[j][4 * g + e] + bv[e]) + rv[e];
                    t = fmaxf(t, 0.f);                      // (both convs end in ReLU: gemm_f32_pwchain_ok)
                    y[j][4 * g + e] = t;
                    v[e] = t;
                }
                *reinterpret_cast<f32x4*>(cell) = v;                        // (same lane, same cell: in place)
                if (piece == 4) __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int h = (piece - 5) * 2 + hh;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&ep[(8 * h + er) * EPS + ec]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y,
                                                           m0 + 8 * h + er < p3.M ? (unsigned)((8 * h + er) * (int)s_y + 32 * j + ec) * 4u : OOB, 0, 0);
                }
            }
        };
#pragma unroll
        for (int piece = 0; piece < 7; ++piece) epi_piece(0, piece);
        // ---- second conv: K = N1 from the registers above, N2 channels
        f32x16 z[NT2];
#pragma unroll
        for (int jn = 0; jn < NT2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) z[jn][r] = 0.f;
        auto w1_frag = [&](int j, int g, int jn) {  // k-step 4 j + g (sub-chunk j), N-tile jn
            return *reinterpret_cast<const f32x4*>(&W1s[j * (N2 * 32) + (32 * jn + frow) * 32 + ((g * 2 + fhalf) ^ fsw) * 4]);
        };
        {
            f32x4 bf = w1_frag(0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT1; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int jn = 0; jn < NT2; ++jn) {
                        const bool last_n = jn + 1 == NT2, last_g = g == 3;
                        const int n2 = last_n ? 0 : jn + 1, g2 = last_n ? (last_g ? 0 : g + 1) : g,
                                  j2 = last_n && last_g ? (j + 1 < NT1 ? j + 1 : j) : j;
                        const f32x4 nx = w1_frag(j2, g2, n2);
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[e], y[j][4 * g + e], z[jn], 0, 0, 0);
                        bf = nx;
                        if (j + 1 < NT1 && g * NT2 + jn < 7) epi_piece(j + 1, g * NT2 + jn);       // (NT2 == 2: eight groups per sub-chunk)
                        if (j == NT1 / 2 && g == 0 && jn == 0) request_a(tile + nw);     // (a1 is free since the first conv; requested here, where
                                                                                     // half of the residual registers are free again)
                        // the first half of the NEXT tile's residual, one N-tile per sub-chunk of the second half of this loop (the
                        // other half follows during the next tile's first k-steps): a trickle under the MFMAs, far ahead of its
                        // use.  All of it this early does not fit: y and the residual both live in the 256 architectural VGPRs.
                        if (g == 3 && last_n && j >= NT1 / 2) request_res(tile + nw, j - NT1 / 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
        const rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void*)(p1.out + m0 * s_t + p1.omap.off), 0, 0x7FFFFF00u, 0x00020000);
#pragma unroll
        for (int jn = 0; jn < NT2; ++jn) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&B1s[32 * jn + 8 * g + 4 * fhalf]);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(z[jn][4 * g + e] + bv[e], 0.f);
                *reinterpret_cast<f32x4*>(&ep[frow * EPS + 8 * g + 4 * fhalf]) = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&ep[(8 * h + er) * EPS + ec]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_t,
                                                       m0 + 8 * h + er < p3.M ? (unsigned)((8 * h + er) * (int)s_t + 32 * jn + ec) * 4u : OOB, 0, 0);
            }
        }
    }
#endif
}

// conv `a` (64 -> 256 + residual) directly followed by conv `b` (256 -> 64) on a's output, both plain NHWC fp32 pointwise convs
bool gemm_f32_pwchain_ok(const GemmArgs& a, const GemmArgs& b) {
    static const int on = [] { const char* e = diag_env("CAPF_PWCHAIN"); return e ? atoi(e) : 1; }();        // A/B runs only
    auto plain = [](const GemmArgs& g) {
        return g.conv && g.ks == 1 && g.stride == 1 && g.pad == 0 && g.K == g.Cin && g.Kpad == g.K && g.omap.G == 1 && (g.omap.S1 & 3) == 0 &&
               (g.omap.off & 3) == 0 && !g.rscale && !g.ln_g && g.splits <= 1 && g.act != ACT_GELU && !g.out_bf16;
    };
    if (!on || !plain(a) || !plain(b)) return false;
    if (a.K != 64 || a.N != 256 || b.K != 256 || b.N != 64 || b.res || a.M != b.M) return false;
    if (!a.res || !a.bias || !b.bias || a.act != ACT_RELU || b.act != ACT_RELU) return false;      // the bottleneck's shape, hard-wired
    if (b.A != a.out + a.omap.off || a.omap.S1 != b.K) return false;          // b reads exactly what a writes, rows dense
    if (a.rmap.G != 1 || (a.rmap.S1 & 3) || (a.rmap.off & 3)) return false;
    if ((long)a.omap.S1 * 32 * 4 >= (1L << 31) || (long)a.rmap.S1 * 32 * 4 >= (1L << 31)) return false;
    {   // As two launches the second conv's output may alias the first one's residual or input (both are dead by then and the plan's
        // allocator reuses their memory); in ONE persistent kernel a wave stores t' rows while other waves still read those operands
        // for later tiles.  Reject the chain unless b.out is disjoint from a.res and a.A.
        auto overlaps = [](const float* p0, long n0, const float* p1, long n1) { return p0 < p1 + n1 && p1 < p0 + n0; };
        const float* bo = b.out + b.omap.off;
        const long bn = (long)(b.M - 1) * b.omap.S1 + b.N;
        if (overlaps(bo, bn, a.res + a.rmap.off, (long)(a.M - 1) * a.rmap.S1 + a.N)) return false;
        if (overlaps(bo, bn, a.A, (long)a.M * a.K)) return false;
    }
    return a.M >= 32 * 1024 * 4;                                               // >= 4 tiles per wave of a full grid
}

const char* gemm_f32_pwchain_kernel_name() { return "igemm_f32_pwchain<64,256,64>"; }

hipError_t launch_gemm_f32_pwchain(const GemmArgs& a, const GemmArgs& b, hipStream_t s) {
    if (!gemm_f32_pwchain_ok(a, b)) return hipErrorInvalidValue;
    const int ntiles = (a.M + 31) / 32;
    constexpr size_t lds_bytes = (size_t)(256 * 64 + 64 * 256 + 256 + 64 + 4 * 32 * 36) * 4;
    static DynLdsAttr attr_once;
    const hipError_t attr = attr_once.ensure(reinterpret_cast<const void*>(&igemm_f32_pwchain_kernel<64, 256, 64>), (int)lds_bytes);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((igemm_f32_pwchain_kernel<64, 256, 64>), dim3(256), dim3(256), lds_bytes, s, a, b, ntiles);
    return hipGetLastError();
}

}  // namespace capf
