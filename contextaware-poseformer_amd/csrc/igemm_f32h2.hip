// fp32 implicit GEMM on the 16-bit matrix pipe for everything that is NOT a 3x3 / stride-1 conv: the 1x1 and stride-2 convs of the HRNet
// fuse / transition layers (pose_hrnet.py:225-303), lone backbone convs, and the lifter's nn.Linear layers (pose_dformer.py:15-59) from
// batch 5 up.  Same arithmetic as the two-fp16-piece conv tile (igemm_f32h2_ws_tile.h): an fp32 operand a travels as a1 = fp16(s a),
// a2 = fp16(s a - a1) under an exact power-of-two scale s, |s a - a1 - a2| <= 2^-23 |s a|, and a1 w1 + a1 w2 + a2 w1 is accumulated in fp32
// (dropped: a2 w2 <= 2^-22 |a w|) -- three v_mfma_f32_32x32x16_f16 for what igemm_f32.hip issues as eight v_mfma_f32_32x32x2_f32 at 1/16
// of the rate.  What differs from the conv tile is WHERE the activation is split:
//   * the A tile goes global -> LDS as plain fp32 by LDS-DMA, exactly as in igemm_f32.hip (same [rows][32 floats] image, same XOR swizzle,
//     same out-of-range-offset zero fill for padding taps and ragged edges): no register round trip on the load path;
//   * a wave splits the fragments it is about to use, in registers, between the ds_read and the MFMA: per 32-deep chunk and 32-row
//     fragment block 16 values per lane -> maximum over the wave (v_max3, four DPP steps, four readlanes) -> scale -> two fp16
//     fragments per 16-deep step.  The scale is per wave, 32 rows and chunk; the accumulators move to a new scale (exact) when it changes;
//   * weights are split once at pack time, one power-of-two scale per output channel, and packed so that a row's chunk is the SAME
//     128 bytes the fp32 pack has: [piece 0: 32 x fp16 | piece 1: 32 x fp16] -- the weight DMA and its swizzle are igemm_f32.hip's.
// Tiles 64 x 64 (three stages) and 128 x 64 (two), 48 KiB of LDS, three blocks per CU; grouped launches as igemm_f32_group_kernel.
// These problems are HBM- / latency-bound once they leave the fp32 pipe; the loop is therefore the plain one (wait, barrier, fire the
// DMA of chunk c + S - 1, read, split, multiply) and relies on the three resident blocks for overlap.
#include <algorithm>

#include "igemm_f32h2_ws_tile.h"
#include "kernels.h"

namespace capf {

typedef float g2_f32x4 __attribute__((ext_vector_type(4)));
typedef float g2_f32x16 __attribute__((ext_vector_type(16)));
static constexpr int G2_BK = 32;
static constexpr int G2_LDS_FLOATS = 2 * (128 + 64) * G2_BK;        // the ring, 48 KiB: three blocks per CU for every tile
[[maybe_unused]] static constexpr int G2_WIDE_FLOATS = 2 * (128 + 128) * G2_BK;      // the 128 x 128 tile's ring, 64 KiB: two blocks per CU
static constexpr long G2_WIDE_MIN_TILES = 960;
static constexpr int G2_LNK = 256;                         // largest LayerNorm'ed width (gamma, beta, 2 x 128 row statistics: 3 KiB behind the 48 KiB ring, still three blocks per CU)

long f32h2_gemm_pack_elems(int N, int Kpad) { return (long)N * Kpad + ((N + 3) & ~3); }      // floats: pieces, then 1 / channel scale

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) void* g2_lptr_t;
typedef __amdgpu_buffer_rsrc_t g2_rsrc_t;

__device__ __forceinline__ float g2_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ long g2_rowmap(const RowMap& r, int m) {
    if (r.G == 1) return (long)m * r.S1 + r.off;
    const int q = m / r.G;
    return (long)q * r.S1 + (long)(m - q * r.G) * r.S2 + r.off;
}
__device__ __forceinline__ int g2_div(int n, FastDiv d) { return (int)((__umulhi((unsigned)n, d.mul) + (unsigned)n) >> d.shift); }

// one output tile (logical id bid) with the calling 256-thread block; lds: S * (BM + BN) * 32 floats
// LNA (rows mode): LayerNorm of the A rows over their K <= G2_LNK columns on the way from LDS to the split, as igemm_f32.hip's LNA path has it --
// row statistics in a prologue (two passes over the row in global memory, while the first chunks' DMA flies), gamma / beta in LDS behind the
// ring (zeros beyond K), x_hat = ((x - mean) * rstd) * gamma + beta in the same order -- so that the fp32 LayerNorm'ed operand is what gets
// split: the res / context blocks' qkv and fc1 projections (pose_dformer.py:62-79, 115-138) leave the fp32 matrix pipe too.
template <int BM, int BN, int WM, int WN, int S, int CONV, bool PLAIN, bool LNA = false>
__device__ __forceinline__ void igemm_h2_tile(const GemmArgs& p, const int bid, float* __restrict__ lds) {
    constexpr int BK = G2_BK;
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPR = 32;                               // tile rows covered by one DMA round of the block (256 threads, 8 per row)
    constexpr int RA = BM / RPR, RB = BN / RPR;
    constexpr int NLOAD = RA + RB;
    constexpr int STAGE = (BM + BN) * BK;
    static_assert((BM / WM) * (BN / WN) == 4, "wave grid");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = (p.N + BN - 1) / BN;
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int srow = tid >> 3;
    const int kq = (((tid & 7) ^ ((srow >> 1) & 7))) * 4;   // source quad of this thread's staging slot (igemm_f32.hip's swizzle)
    const int nchunks = p.Kpad / BK;

    constexpr unsigned OOB_A = CONV ? 0x80000000u : 0xFFFFFFFFu;
    constexpr unsigned NREC_A = CONV ? 0x7FFFFF00u : 0xFFFFFF00u;
    long a_base = 0;
    if (CONV) {
        const int b = g2_div(m0, p.fd_hw), rem = m0 - b * p.Ho * p.Wo;
        const int ho = g2_div(rem, p.fd_wo), wo = rem - ho * p.Wo;
        a_base = ((long)b * p.H * p.W + (long)(ho * p.stride - p.pad) * p.W + (wo * p.stride - p.pad)) * p.Cin;
    }
    const g2_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + a_base), 0, NREC_A, 0x00020000);
    const g2_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wp + (long)n0 * p.Kpad), 0,
                                                               (unsigned)(p.N - n0) * (unsigned)p.Kpad * 4u, 0x00020000);
    unsigned a_rel[RA], a_mask[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + srow + RPR * i;
        a_rel[i] = 0; a_mask[i] = 0u;
        if (m < p.M) {
            if (!CONV) {
                a_rel[i] = (unsigned)(g2_rowmap(p.amap, m) + kq) * 4u;
                a_mask[i] = 1u;
            } else {
                const int b = g2_div(m, p.fd_hw), rem = m - b * p.Ho * p.Wo;
                const int ho = g2_div(rem, p.fd_wo), wo = rem - ho * p.Wo;
                const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
                const long off = ((long)b * p.H * p.W + (long)h0 * p.W + w0) * p.Cin;
                a_rel[i] = (unsigned)(off - a_base + kq) * 4u;
                const int kw_lo = max(0, -w0), kw_hi = min(p.ks, p.W - w0);
                const int kh_lo = max(0, -h0), kh_hi = min(p.ks, p.H - h0);
                if (kw_hi > kw_lo && kh_hi > kh_lo) {
                    const unsigned wbits = ((1u << kw_hi) - 1) & ~((1u << kw_lo) - 1);
                    const unsigned below_hi = kh_hi * p.ks >= 32 ? ~0u : ((1u << (kh_hi * p.ks)) - 1);
                    const unsigned below_lo = (1u << (kh_lo * p.ks)) - 1;
                    a_mask[i] = (wbits * (unsigned)p.spread) & below_hi & ~below_lo;
                }
            }
        }
    }
    unsigned w_off[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) w_off[i] = (unsigned)((srow + RPR * i) * p.Kpad + kq) * 4u;

    const bool uni = CONV && (p.Cin & (BK - 1)) == 0;       // a chunk lies inside one tap: block-uniform walk, tap offset in the scalar offset
    int u_tap = 0, u_ci = 0, u_kh = 0, u_kw = 0;
    int tap = 0, ci = kq, kh = 0, kw = 0;
    if (CONV && !uni) { tap = kq / p.Cin; ci = kq - tap * p.Cin; kh = tap / p.ks; kw = tap - kh * p.ks; }
    unsigned voff[NLOAD];
    unsigned soff_a = 0;
    auto prepare = [&](int c) {
        if (!CONV) {
            const bool k_ok = c * BK + kq < p.K;
#pragma unroll
            for (int i = 0; i < RA; ++i) voff[i] = (k_ok && a_mask[i]) ? a_rel[i] + (unsigned)(c * BK) * 4u : OOB_A;
        } else if (uni) {
            const unsigned bit = u_tap < 32 ? (1u << u_tap) : 0u;
            soff_a = __builtin_amdgcn_readfirstlane((unsigned)((u_kh * p.W + u_kw) * p.Cin + u_ci) * 4u);
#pragma unroll
            for (int i = 0; i < RA; ++i) voff[i] = (a_mask[i] & bit) ? a_rel[i] : OOB_A;
            u_ci += BK;
            if (u_ci >= p.Cin) { u_ci = 0; ++u_tap; if (++u_kw == p.ks) { u_kw = 0; ++u_kh; } }
        } else {
            const unsigned bit = tap < 32 ? (1u << tap) : 0u;
            const unsigned t = (unsigned)((kh * p.W + kw) * p.Cin + ci - kq) * 4u;
#pragma unroll
            for (int i = 0; i < RA; ++i) voff[i] = (a_mask[i] & bit) ? a_rel[i] + t : OOB_A;
            ci += BK;
            while (ci >= p.Cin) { ci -= p.Cin; ++tap; if (++kw == p.ks) { kw = 0; ++kh; } }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) { voff[RA + i] = w_off[i]; w_off[i] += BK * 4u; }
    };
    auto fire_all = [&](int stage) {
        float* As = lds + stage * STAGE;
#pragma unroll
        for (int idx = 0; idx < NLOAD; ++idx) {
            if (idx < RA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (g2_lptr_t)(As + (idx * RPR + wave * 8) * BK), 16, voff[idx], soff_a, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (g2_lptr_t)(As + BM * BK + ((idx - RA) * RPR + wave * 8) * BK), 16, voff[idx], 0, 0, 0);
        }
    };

    g2_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int frow = lane & 31, fsw = (frow >> 1) & 7, fhalf = lane >> 5;
    int sb[TM];                                             // biased exponent of the scale the accumulators of row block i are in
    int smin[TM];                                           // ... of the smallest scale (largest chunk maximum) so far
#pragma unroll
    for (int i = 0; i < TM; ++i) { sb[i] = 127; smin[i] = 190; }

#pragma unroll
    for (int s = 0; s < S - 1; ++s) {
        if (s < nchunks) { prepare(s); fire_all(s); }
    }
    float* const ln_g = lds + ((BM == 128 && BN == 128) ? G2_WIDE_FLOATS : G2_LDS_FLOATS);   // [G2_LNK] | beta [G2_LNK] | mean [128] | rstd [128]
    float* const ln_b = ln_g + G2_LNK;
    float mu_f[TM], rs_f[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { mu_f[i] = 0.f; rs_f[i] = 1.f; }
    if (LNA) {
        float* const st_mu = ln_b + G2_LNK;
        float* const st_rs = st_mu + 128;
        for (int k = tid * 4; k < p.Kpad; k += 256 * 4) {
            const bool in = k < p.K;
            *reinterpret_cast<g2_f32x4*>(ln_g + k) = in ? *reinterpret_cast<const g2_f32x4*>(p.ln_g + k) : g2_f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<g2_f32x4*>(ln_b + k) = in ? *reinterpret_cast<const g2_f32x4*>(p.ln_b + k) : g2_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        constexpr int TPR = 256 / BM;                       // threads per row (adjacent lanes)
        const int r = tid / TPR, part = tid % TPR;
        const int m = m0 + r;
        const bool ok = m < p.M;
        const float* x = p.A + (ok ? g2_rowmap(p.amap, m) : 0);
        float sum = 0.f;
        if (ok)
            for (int k = part * 4; k < p.K; k += TPR * 4) {
                const g2_f32x4 v = *reinterpret_cast<const g2_f32x4*>(x + k);
                sum += (v[0] + v[1]) + (v[2] + v[3]);
            }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum / (float)p.K;
        float sq = 0.f;
        if (ok)
            for (int k = part * 4; k < p.K; k += TPR * 4) {
                const g2_f32x4 v = *reinterpret_cast<const g2_f32x4*>(x + k);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) sq += __shfl_xor(sq, o, 64);
        if (part == 0) {
            st_mu[r] = ok ? mean : 0.f;
            st_rs[r] = ok ? 1.0f / sqrtf(sq / (float)p.K + p.ln_eps) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            mu_f[i] = st_mu[wm0 + i * 32 + frow];
            rs_f[i] = st_rs[wm0 + i * 32 + frow];
        }
    }
    int st_read = 0, st_fill = S - 1;
    for (int c = 0; c < nchunks; ++c) {
        // chunk c has landed (the S - 2 youngest chunks may still fly), and every wave is done with the stage chunk c + S - 1 goes into
        if (c + S - 1 <= nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * NLOAD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c + S - 1 < nchunks) { prepare(c + S - 1); fire_all(st_fill); }
        const float* As = lds + st_read * STAGE;
        const float* Bs = As + BM * BK;
        // fragments of the chunk's two 16-deep steps: lane (frow, fhalf) holds k = 16 step + 8 fhalf + 0 .. 7 of its row
        g2_f32x4 ar[TM][2][2];
        ws_f16x8 bw[TN][2][2];                              // [piece][step]
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    ar[i][st][h] = *reinterpret_cast<const g2_f32x4*>(&As[(wm0 + i * 32 + frow) * BK + (((4 * st + 2 * fhalf + h) ^ fsw) * 4)]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    bw[j][pc][st] = __builtin_bit_cast(ws_f16x8, *reinterpret_cast<const g2_f32x4*>(&Bs[(wn0 + j * 32 + frow) * BK + (((4 * pc + 2 * st + fhalf) ^ fsw) * 4)]));
        }
        if (LNA) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int kq = c * BK + 16 * st + 8 * fhalf + 4 * h;
                    const g2_f32x4 gq = *reinterpret_cast<const g2_f32x4*>(ln_g + kq), bq = *reinterpret_cast<const g2_f32x4*>(ln_b + kq);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) ar[i][st][h][e] = ((ar[i][st][h][e] - mu_f[i]) * rs_f[i]) * gq[e] + bq[e];
                }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float m = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(ar[i][st][h][e]));
            // wave-uniform.  A scale never rises more than 2^80 above the smallest one this row block has used: the accumulators hold up
            // to 2^35 in units of that scale, and following a chunk of zeros (or of values 2^-90 below an earlier chunk) all the way
            // up would overflow them -- such a chunk's values are below anything the fp32 sum keeps either way
            const int sn = min(h2_scale_exp(h2_wave_max(m)), (c > 0 ? smin[i] : 190) + 80);
            smin[i] = c > 0 ? min(smin[i], sn) : sn;
            if (sn != sb[i]) {
                if (c > 0) {
                    const float f = __int_as_float((127 + sn - sb[i]) << 23);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
                }
                sb[i] = sn;
            }
            const float sc = __int_as_float(sn << 23);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                ws_u32x4 u1, u2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned s1, s2;
                    h2_split2(ar[i][st][k >> 1][2 * (k & 1)], ar[i][st][k >> 1][2 * (k & 1) + 1], sc, s1, s2);
                    u1[k] = s1; u2[k] = s2;
                }
                const ws_f16x8 a1 = __builtin_bit_cast(ws_f16x8, u1), a2 = __builtin_bit_cast(ws_f16x8, u2);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[j][0][st], a2, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[j][1][st], a1, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[j][0][st], a1, acc[i][j], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        st_read = (st_read + 1 == S) ? 0 : st_read + 1;
        st_fill = (st_fill + 1 == S) ? 0 : st_fill + 1;
    }

    // ---- epilogue (igemm_f32.hip's: lane = row lane & 31, register group g = four consecutive channels 8 g + 4 (lane >> 5) + 0 .. 3):
    // out = act(acc / (row block's scale * channel's weight scale) + bias (+ residual))
    const float* winv = p.Wp + (long)p.N * p.Kpad;
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    long o_row[TM], r_row[TM];
    bool m_ok[TM];
    float rs[TM];                                           // per-row scale of the branch output (DropPath keep mask / keep_prob), 1 without
    if constexpr (TN > 2) {                                 // (the wide tile: one column block at a time -- all of bv / wv / rv at once is 192 registers)
        bool mk[TM];
        long orow[TM], rrow[TM];
        float rsc[TM], inv_s[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + i * 32 + (lane & 31);
            mk[i] = full || m < p.M;
            orow[i] = 0; rrow[i] = 0; rsc[i] = 1.0f;
            inv_s[i] = __int_as_float((254 - sb[i]) << 23);
            if (mk[i]) {
                if (p.rscale) rsc[i] = p.rscale[m / p.rs_div];
                orow[i] = PLAIN ? (long)m * p.omap.S1 + p.omap.off : g2_rowmap(p.omap, m);
                if (p.res) rrow[i] = PLAIN ? (long)m * p.rmap.S1 + p.rmap.off : g2_rowmap(p.rmap, m);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            g2_f32x4 b4[4], w4[4], r4[TM][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                const bool nk = full || n < p.N;
                b4[g] = g2_f32x4{0.f, 0.f, 0.f, 0.f};
                w4[g] = g2_f32x4{0.f, 0.f, 0.f, 0.f};
                if (nk) {
                    w4[g] = *reinterpret_cast<const g2_f32x4*>(winv + n);
                    if (p.bias) b4[g] = *reinterpret_cast<const g2_f32x4*>(p.bias + n);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    r4[i][g] = g2_f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p.res && mk[i] && nk) r4[i][g] = *reinterpret_cast<const g2_f32x4*>(p.res + rrow[i] + n);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                    g2_f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = fmaf(fmaf(acc[i][j][4 * g + e], w4[g][e] * inv_s[i], b4[g][e]), rsc[i], r4[i][g][e]);
                        if (p.act == ACT_GELU) t = g2_gelu(t);
                        if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
                        v[e] = t;
                    }
                    if (mk[i] && (full || n < p.N)) *reinterpret_cast<g2_f32x4*>(p.out + orow[i] + n) = v;
                }
        }
        return;
    }
    g2_f32x4 bv[TN][4], wv[TN][4], rv[TM][TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
            bv[j][g] = g2_f32x4{0.f, 0.f, 0.f, 0.f};
            wv[j][g] = g2_f32x4{0.f, 0.f, 0.f, 0.f};
            if (full || n < p.N) {
                wv[j][g] = *reinterpret_cast<const g2_f32x4*>(winv + n);
                if (p.bias) bv[j][g] = *reinterpret_cast<const g2_f32x4*>(p.bias + n);
            }
        }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + i * 32 + (lane & 31);
        m_ok[i] = full || m < p.M;
        o_row[i] = 0; r_row[i] = 0; rs[i] = 1.0f;
        if (m_ok[i]) {
            if (p.rscale) rs[i] = p.rscale[m / p.rs_div];
            o_row[i] = PLAIN ? (long)m * p.omap.S1 + p.omap.off : g2_rowmap(p.omap, m);
            if (p.res) r_row[i] = PLAIN ? (long)m * p.rmap.S1 + p.rmap.off : g2_rowmap(p.rmap, m);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                rv[i][j][g] = g2_f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.res && m_ok[i] && (full || n < p.N)) rv[i][j][g] = *reinterpret_cast<const g2_f32x4*>(p.res + r_row[i] + n);
            }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const float inv_s = __int_as_float((254 - sb[i]) << 23);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                g2_f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = fmaf(fmaf(acc[i][j][4 * g + e], wv[j][g][e] * inv_s, bv[j][g][e]), rs[i], rv[i][j][g][e]);
                    if (p.act == ACT_GELU) t = g2_gelu(t);
                    if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
                    v[e] = t;
                }
                if (m_ok[i] && (full || n < p.N)) *reinterpret_cast<g2_f32x4*>(p.out + o_row[i] + n) = v;
            }
    }
}

__device__ __forceinline__ int g2_xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
#endif

static_assert(3 * (64 + 64) * G2_BK <= G2_LDS_FLOATS, "h2 gemm LDS");

// cfg 0: 128 x 64 tile, two stages, four waves ALONG M (32 rows x 64 columns each: a row block is split by one wave, not by the two of a
// 2 x 2 grid -- the split's VALU work, not the matrix pipe, bounds this kernel: cfg3 57.6 -> 56.9 ms on one box, same bits);  1: 64 x 64, three
// stages, 2 x 2 waves;  2: 128 x 32
template <int CONV, bool PLAIN, bool LNA = false>
__global__ __launch_bounds__(256, 3) void igemm_f32h2g_kernel(GemmArgs p, int cfg) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float lds[G2_LDS_FLOATS + (LNA ? 2 * G2_LNK + 256 : 0)];
    const int bid = g2_xcd_remap(blockIdx.x, gridDim.x);
    if (cfg == 0) igemm_h2_tile<128, 64, 32, 64, 2, CONV, PLAIN, LNA>(p, bid, lds);
    else if (cfg == 2) igemm_h2_tile<128, 32, 32, 32, 2, CONV, PLAIN, LNA>(p, bid, lds);
    else igemm_h2_tile<64, 64, 32, 32, 3, CONV, PLAIN, LNA>(p, bid, lds);
#endif
}

// cfg 3: 128 x 128 tile, two stages, four waves along M with 32 rows x 128 columns each -- a split row block feeds FOUR column blocks (24 MFMAs
// per chunk and wave behind the same 16 values per lane), 64 KiB of LDS, two blocks per CU: the wide projections of the training batch
// (qkv / fc1 of the joint blocks at batch 512), where the split's VALU work bounds the narrower tiles (profiles/r05_pmc_f32h2g_linear.txt)
template <int CONV, bool PLAIN, bool LNA = false>
__global__ __launch_bounds__(256, 2) void igemm_f32h2g_wide_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float lds[G2_WIDE_FLOATS + (LNA ? 2 * G2_LNK + 256 : 0)];
    igemm_h2_tile<128, 128, 32, 128, 2, CONV, PLAIN, LNA>(p, g2_xcd_remap(blockIdx.x, gridDim.x), lds);
#endif
}

struct G2GroupArgs {
    GemmArgs g[MAXG];
    int start[MAXG + 1];
    int tiles[MAXG];
    int cfg[MAXG];
    int n;
};

__global__ __launch_bounds__(256, 3) void igemm_f32h2g_group_kernel(G2GroupArgs ga) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float lds[G2_LDS_FLOATS];
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < ga.n && b >= ga.start[pi + 1]) ++pi;
    const int l = b - ga.start[pi];
    const int per_xcd = (ga.start[pi + 1] - ga.start[pi]) >> 3;
    const int bid = (l & 7) * per_xcd + (l >> 3);
    if (bid >= ga.tiles[pi]) return;
    if (ga.cfg[pi] == 0) igemm_h2_tile<128, 64, 32, 64, 2, 1, true>(ga.g[pi], bid, lds);
    else if (ga.cfg[pi] == 2) igemm_h2_tile<128, 32, 32, 32, 2, 1, true>(ga.g[pi], bid, lds);
    else igemm_h2_tile<64, 64, 32, 32, 3, 1, true>(ga.g[pi], bid, lds);
#endif
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------
static bool g2_plain(const GemmArgs& a) { return a.omap.G == 1 && (!a.res || a.rmap.G == 1); }

// What this kernel takes (a function of the problem alone): fp32 conv (any ks <= 5 / stride / pad, NHWC) or rows-mode GEMM with
// K % 32 == 0 staged rows, N % 4 == 0, 16-byte aligned output / residual rows, LayerNorm fold up to 256 columns (plain rows), no split-K
bool gemm_f32h2g_ok(const GemmArgs& a) {
    if (!a.Wh2 || a.out_bf16 || a.splits > 1 || a.M <= 0 || a.N <= 0 || (a.N & 3) || a.Kpad % G2_BK != 0) return false;
    if (a.act == ACT_GELU && a.conv) return false;
    if (a.ln_g && (a.conv || !a.ln_b || a.K > G2_LNK || a.Kpad > G2_LNK || (a.K & 3) || a.amap.G < 1 || !g2_plain(a))) return false;   // (LNA: plain output rows)
    if (a.conv) {
        if (a.ks < 1 || a.ks > 5 || a.Cin % 4 != 0 || a.K != a.ks * a.ks * a.Cin || a.Ho <= 0 || a.Wo <= 0) return false;
        if ((double)a.H * a.W * a.Cin * 4.0 * 3.0 >= 2.0e9) return false;    // (input offsets count from the tile's first pixel: one tile spans < 2 frames)
        if (a.omap.G != 1 || (a.res && a.rmap.G != 1)) return false;
    } else {
        if (a.K % 4 != 0 || a.amap.G < 1 || (a.amap.S1 & 3) || (a.amap.S2 & 3) || (a.amap.off & 3)) return false;
        const double g = a.amap.G;
        if ((((double)a.M / g + 1.0) * (double)a.amap.S1 + g * (double)a.amap.S2 + (double)a.amap.off + a.Kpad) * 4.0 >= 4.0e9) return false;
    }
    if (a.omap.G == 1 && (double)a.M * (double)a.omap.S1 >= 4.0e9) return false;
    if ((a.omap.S1 & 3) || (a.omap.S2 & 3) || (a.omap.off & 3)) return false;
    if (a.res && ((a.rmap.S1 & 3) || (a.rmap.S2 & 3) || (a.rmap.off & 3))) return false;
    return true;
}

static int g2_cfg(const GemmArgs& a, bool grouped = false) {                      // 128 x 64 tiles once they give every CU two blocks (512; transition1.1.0.0 at batch 64: 123 -> 116 us)
    // at most 32 output columns (the fuse layers' convs into the 32-channel branch): 128 x 32 -- four waves along M instead of two waves
    // multiplying columns that do not exist; the split blocks (32 rows x chunk) are those of the other tiles: same bits
    if (a.N <= 32 && a.M >= 128) return 2;
    const long big = (long)((a.M + 127) / 128) * ((a.N + 63) / 64);
    // 128 x 128 from two full rounds of that tile (two blocks per CU) with no ragged last column block.  Measured at batch 512 (8704 rows, K 640, GELU):
    // N 1920 (1020 tiles) 107.6 -> 93.1 us; N 1280 (680 tiles: 1.33 rounds) 74.8 -> 76.0; N 640 the same -- EXPERIMENTS R6.7
    if (!grouped && a.N % 128 == 0 && (long)((a.M + 127) / 128) * (a.N / 128) >= G2_WIDE_MIN_TILES) return 3;
    return big >= 512 ? 0 : 1;
}
static int g2_tiles(const GemmArgs& a, int cfg) {
    if (cfg == 2) return (a.M + 127) / 128;
    if (cfg == 3) return ((a.M + 127) / 128) * ((a.N + 127) / 128);
    return ((a.M + (cfg == 0 ? 127 : 63)) / (cfg == 0 ? 128 : 64)) * ((a.N + 63) / 64);
}

static void g2_fill(GemmArgs& a) {
    if (a.conv) {
        a.fd_hw = make_fastdiv((unsigned)(a.Ho * a.Wo));
        a.fd_wo = make_fastdiv((unsigned)a.Wo);
        unsigned long long sp = 0;
        for (int kh = 0; kh < a.ks; ++kh) sp |= 1ull << (kh * a.ks);
        a.spread = sp;
    }
}

hipError_t launch_gemm_f32h2g(const GemmArgs& a_in, hipStream_t s) {
    if (!gemm_f32h2g_ok(a_in)) return hipErrorInvalidValue;
    GemmArgs a = a_in;
    a.Wp = a.Wh2;
    if (a.rs_div <= 0) a.rs_div = 1;
    g2_fill(a);
    const int cfg = g2_cfg(a), tiles = g2_tiles(a, cfg);
    const bool plain = g2_plain(a);
    if (cfg == 3) {
        if (a.conv) {
            if (!plain) return hipErrorInvalidValue;
            hipLaunchKernelGGL((igemm_f32h2g_wide_kernel<1, true>), dim3(tiles), dim3(256), 0, s, a);
        } else if (a.ln_g) {
            hipLaunchKernelGGL((igemm_f32h2g_wide_kernel<0, true, true>), dim3(tiles), dim3(256), 0, s, a);
        } else if (plain) {
            hipLaunchKernelGGL((igemm_f32h2g_wide_kernel<0, true>), dim3(tiles), dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL((igemm_f32h2g_wide_kernel<0, false>), dim3(tiles), dim3(256), 0, s, a);
        }
        return hipGetLastError();
    }
    if (a.conv) {
        if (!plain) return hipErrorInvalidValue;
        hipLaunchKernelGGL((igemm_f32h2g_kernel<1, true>), dim3(tiles), dim3(256), 0, s, a, cfg);
    } else if (a.ln_g) {
        hipLaunchKernelGGL((igemm_f32h2g_kernel<0, true, true>), dim3(tiles), dim3(256), 0, s, a, cfg);
    } else if (plain) {
        hipLaunchKernelGGL((igemm_f32h2g_kernel<0, true>), dim3(tiles), dim3(256), 0, s, a, cfg);
    } else {
        hipLaunchKernelGGL((igemm_f32h2g_kernel<0, false>), dim3(tiles), dim3(256), 0, s, a, cfg);
    }
    return hipGetLastError();
}

hipError_t launch_gemm_f32h2g_group(const GemmArgs* list, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > MAXG) return hipErrorInvalidValue;
    struct Item { GemmArgs a; int cfg, tiles, cost; };
    Item it[MAXG];
    for (int i = 0; i < n; ++i) {
        if (!gemm_f32h2g_ok(list[i]) || !list[i].conv || !g2_plain(list[i])) return hipErrorInvalidValue;
        it[i].a = list[i];
        it[i].a.Wp = list[i].Wh2;
        g2_fill(it[i].a);
        it[i].cfg = g2_cfg(it[i].a, true);
        it[i].tiles = g2_tiles(it[i].a, it[i].cfg);
        it[i].cost = it[i].a.Kpad;
    }
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && it[j].cost > it[j - 1].cost; --j) { Item t = it[j]; it[j] = it[j - 1]; it[j - 1] = t; }
    G2GroupArgs ga;
    ga.n = n;
    int start = 0;
    for (int i = 0; i < n; ++i) {
        ga.g[i] = it[i].a; ga.cfg[i] = it[i].cfg; ga.tiles[i] = it[i].tiles;
        ga.start[i] = start;
        start += (it[i].tiles + 7) & ~7;
    }
    ga.start[n] = start;
    for (int i = n; i < MAXG; ++i) { ga.start[i + 1] = start; ga.tiles[i] = 0; ga.cfg[i] = 1; ga.g[i] = ga.g[0]; }
    hipLaunchKernelGGL(igemm_f32h2g_group_kernel, dim3(start), dim3(256), 0, s, ga);
    return hipGetLastError();
}

const char* gemm_f32h2g_kernel_name(const GemmArgs& a, bool grouped) {
    if (grouped) return "igemm_f32h2g_group";
    const int cfg = g2_cfg(a);
    if (cfg == 3) return a.conv ? "igemm_f32h2g<128x128,conv>" : "igemm_f32h2g<128x128,rows>";
    return a.conv ? (cfg == 0 ? "igemm_f32h2g<128x64,conv>" : (cfg == 2 ? "igemm_f32h2g<128x32,conv>" : "igemm_f32h2g<64x64,conv>"))
                  : (cfg == 0 ? "igemm_f32h2g<128x64,rows>" : (cfg == 2 ? "igemm_f32h2g<128x32,rows>" : "igemm_f32h2g<64x64,rows>"));
}

// ---- pack: fold (conv: BatchNorm as launch_pack_conv; linear: none) -> one power-of-two scale per output channel -> two fp16 pieces;
// Wp[n][chunk][piece][32] fp16 over the fp32 pack's geometry ([N][Kpad] floats, k = (kh, kw, ci) for convs), then [N] fp32 inverse scales
__global__ __launch_bounds__(256) void g2_wscale_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                                        float eps, float* __restrict__ winv, int N, int K) {
    __shared__ float red[256];
    const int n = blockIdx.x;
    const float sc = gamma ? gamma[n] / sqrtf(var[n] + eps) : 1.f;
    float m = 0.f;
    for (int i = threadIdx.x; i < K; i += 256) m = fmaxf(m, fabsf(w[(long)n * K + i] * sc));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) winv[n] = __int_as_float((254 - h2_scale_exp(__float_as_int(red[0]))) << 23);
}

__global__ void g2_pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps, unsigned short* __restrict__ Wp,
                               const float* __restrict__ winv, float* __restrict__ bias, int N, int Cin, int ks, int K, int Kpad, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {     // i = (n, k) of the padded matrix
        const int n = (int)(i / Kpad), k = (int)(i - (long)n * Kpad);
        float v = 0.f;
        if (k < K) {
            const float sc = gamma ? gamma[n] / sqrtf(var[n] + eps) : 1.f;
            long src;
            if (ks > 0) {                                   // conv: k = (kh, kw, ci) of an OIHW filter
                const int t = k / Cin, c = k - t * Cin;
                src = (((long)n * Cin + c) * ks + t / ks) * ks + t % ks;
            } else {
                src = (long)n * K + k;
            }
            v = w[src] * sc * __uint_as_float(0x7F000000u - __float_as_uint(winv[n]));
        }
        if (bias && k == 0 && gamma) bias[n] = beta[n] - mean[n] * (gamma[n] / sqrtf(var[n] + eps));
        const _Float16 p0 = (_Float16)v;
        const _Float16 p1 = (_Float16)(v - (float)p0);
        const long base = ((long)n * (Kpad / 32) + k / 32) * 64 + (k & 31);
        Wp[base] = __builtin_bit_cast(unsigned short, p0);
        Wp[base + 32] = __builtin_bit_cast(unsigned short, p1);
    }
}

// conv: w OIHW [N][Cin][ks][ks] with BatchNorm (gamma may be null: no fold, bias untouched); linear: ks = 0, w [N][K]
hipError_t launch_pack_f32h2_gemm(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                  float* Wp, float* bias, int N, int Cin, int ks, int K, int Kpad, hipStream_t s) {
    if (N <= 0 || Kpad % 32 != 0 || K > Kpad) return hipErrorInvalidValue;
    float* winv = Wp + (long)N * Kpad;
    hipLaunchKernelGGL(g2_wscale_kernel, dim3(N), dim3(256), 0, s, w, gamma, var, eps, winv, N, K);
    const long total = (long)N * Kpad;
    const long want = (total + 255) / 256;
    hipLaunchKernelGGL(g2_pack_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, s, w, gamma, beta, mean, var, eps,
                       reinterpret_cast<unsigned short*>(Wp), winv, bias, N, Cin, ks, K, Kpad, total);
    return hipGetLastError();
}

hipError_t launch_pack_f32h2_gemm_rows(const float* w, float* Wp, int n0, int n, int Ntot, int K, int Kpad, hipStream_t s) {
    if (n <= 0 || Kpad % 32 != 0 || K > Kpad || n0 < 0 || n0 + n > Ntot) return hipErrorInvalidValue;
    float* winv = Wp + (long)Ntot * Kpad + n0;
    hipLaunchKernelGGL(g2_wscale_kernel, dim3(n), dim3(256), 0, s, w, (const float*)nullptr, (const float*)nullptr, 0.f, winv, n, K);
    const long total = (long)n * Kpad;
    const long want = (total + 255) / 256;
    hipLaunchKernelGGL(g2_pack_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, s, w, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, 0.f, reinterpret_cast<unsigned short*>(Wp + (long)n0 * Kpad), winv,
                       (float*)nullptr, n, 0, 0, K, Kpad, total);
    return hipGetLastError();
}

// ---- the training step's weights (train.cpp): every nn.Linear of the lifter as W (forward) AND as W^T (input gradient), all matrices of
// the step in three launches off one table.  A block = one 32 x 32 tile of one W: pass 1 leaves the row and column maxima (bit patterns
// of non-negative floats order like integers: atomicMax, order-independent), pass 2 scales by the row's / column's power of two, splits
// and writes the tile into the forward pack and -- across 4 KiB of LDS -- into the transposed one, 64 contiguous bytes per row and piece.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ const H2TrainW& g2_find(const H2TrainW* __restrict__ tab, int n, int tile) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile_start <= tile) lo = mid; else hi = mid - 1;
    }
    return tab[lo];
}
#endif

template <int PASS>
__global__ __launch_bounds__(256) void g2_train_pack_kernel(const H2TrainW* __restrict__ tab, int n, float* __restrict__ base, int* __restrict__ maxima) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float tile[32][33];
    const H2TrainW e = g2_find(tab, n, (int)blockIdx.x);
    const int t = (int)blockIdx.x - e.tile_start, tk_n = (e.K + 31) >> 5;
    const int tn = t / tk_n, tk = t - tn * tk_n;
    const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
    const int nn = tn * 32 + r, k0 = tk * 32 + q * 4;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (nn < e.N && k0 + i < e.K) ? e.w[(long)nn * e.ld + k0 + i] : 0.f;
    int* rowmax = maxima + e.max_off;
    int* colmax = rowmax + e.N;
    if (PASS == 0) {
        float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (q == 0 && nn < e.N) atomicMax(&rowmax[nn], __float_as_int(m));
        if (e.bwd_off >= 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tile[r][q * 4 + i] = fabsf(v[i]);
            __syncthreads();
            if (threadIdx.x < 32) {
                float c = 0.f;
#pragma unroll
                for (int rr = 0; rr < 32; ++rr) c = fmaxf(c, tile[rr][threadIdx.x]);
                const int k = tk * 32 + (int)threadIdx.x;
                if (k < e.K) atomicMax(&colmax[k], __float_as_int(c));
            }
        }
    } else {
        auto split_store = [](unsigned short* dst, const float* x, float sc) {       // four values -> 4 + 4 fp16 (piece 1 sits 32 halves on)
            unsigned short p0[4], p1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y = x[i] * sc;
                const _Float16 a = (_Float16)y;
                const _Float16 b = (_Float16)(y - (float)a);
                p0[i] = __builtin_bit_cast(unsigned short, a);
                p1[i] = __builtin_bit_cast(unsigned short, b);
            }
            *reinterpret_cast<uint2*>(dst) = uint2{(unsigned)p0[0] | ((unsigned)p0[1] << 16), (unsigned)p0[2] | ((unsigned)p0[3] << 16)};
            *reinterpret_cast<uint2*>(dst + 32) = uint2{(unsigned)p1[0] | ((unsigned)p1[1] << 16), (unsigned)p1[2] | ((unsigned)p1[3] << 16)};
        };
        if (e.fwd_off >= 0 && nn < e.N) {
            const int Kpad = tk_n * 32;
            const int sb = h2_scale_exp(rowmax[nn]);
            float* fw = base + e.fwd_off;
            split_store(reinterpret_cast<unsigned short*>(fw) + ((long)nn * tk_n + tk) * 64 + q * 4, v, __int_as_float(sb << 23));
            if (tk == 0 && q == 0) fw[(long)e.N * Kpad + nn] = __int_as_float((254 - sb) << 23);
        }
        if (e.bwd_off >= 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tile[r][q * 4 + i] = v[i];
            __syncthreads();
            const int k = tk * 32 + r;                       // this thread: row k of W^T, its columns (n) tn * 32 + 4 q ..
            if (k < e.K) {
                const int tn_n = (e.N + 31) >> 5, Np = tn_n * 32;
                const int sb = h2_scale_exp(colmax[k]);
                float x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = tile[q * 4 + i][r];
                float* bw = base + e.bwd_off;
                split_store(reinterpret_cast<unsigned short*>(bw) + ((long)k * tn_n + tn) * 64 + q * 4, x, __int_as_float(sb << 23));
                if (tn == 0 && q == 0) bw[(long)e.K * Np + k] = __int_as_float((254 - sb) << 23);
            }
        }
    }
#endif
}

hipError_t launch_pack_f32h2_train(const H2TrainW* tab_dev, int n, int tiles, float* base, int* maxima, long maxima_elems, hipStream_t s) {
    if (n <= 0 || tiles <= 0) return hipSuccess;
    hipError_t r = hipMemsetAsync(maxima, 0, sizeof(int) * (size_t)maxima_elems, s);
    if (r != hipSuccess) return r;
    hipLaunchKernelGGL(g2_train_pack_kernel<0>, dim3(tiles), dim3(256), 0, s, tab_dev, n, base, maxima);
    hipLaunchKernelGGL(g2_train_pack_kernel<1>, dim3(tiles), dim3(256), 0, s, tab_dev, n, base, maxima);
    return hipGetLastError();
}

}  // namespace capf
