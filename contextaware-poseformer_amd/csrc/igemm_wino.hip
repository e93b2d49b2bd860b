// fp32 3x3 / stride-1 convolution with the Winograd F(2,3) transform applied ALONG W, fused into one MFMA kernel
// (v_mfma_f32_32x32x2_f32).  Covers the BasicBlock / Bottleneck 3x3 convs of HRNet and CPN (pose_hrnet.py:66-136,
// networks/resnet.py:58-93): 77 % of the path's FLOPs.
//
//   direct:   y[h, w, n]  = sum_{kh, kw, c} x[h + kh - 1, w + kw - 1, c] * g[n, c, kh, kw]               9 MACs per (c, n, pixel)
//   F(2,3):   a "tile" t is the output pair (w = 2 wt, 2 wt + 1) of a row; with d_j = x[h + kh - 1, 2 wt - 1 + j, c], j = 0..3,
//             v_0 = d_0 - d_2,  v_1 = d_1 + d_2,  v_2 = d_2 - d_1,  v_3 = d_1 - d_3                    (input transform  B^T d)
//             u_0 = g_0,  u_1 = (g_0 + g_1 + g_2) / 2,  u_2 = (g_0 - g_1 + g_2) / 2,  u_3 = g_2        (weights, packed   G g)
//             m_p = sum_{kh, c} v_p * u_p          (FOUR independent GEMMs over K = 3 C, M = tiles)      12 MACs per (c, n, pair) = 6 per pixel
//             y_0 = m_0 + m_1 + m_2,   y_1 = m_1 - m_2 - m_3                                           (output transform A^T m)
//   i.e. 1.5x fewer MFMAs than the implicit GEMM of igemm_f32.hip for the same result (to fp32 roundoff: the transform
//   coefficients are 0, +-1, 1/2).  The 2-D F(2x2,3x3) variant would need 16 accumulator sets per output tile (a whole
//   AGPR file for one 32x32 MFMA tile) and 2.7x the LDS per staged chunk; the 1-D form needs 4 and reuses the direct
//   kernel's operand path unchanged:
//     * the raw pixels d_0..d_3 are the im2col of a "3 x 4 kernel, stride (1, 2)" convolution, so they are staged by the
//       same `buffer_load_dwordx4 ... lds` loader (block-uniform descriptor, per-row tap mask, hardware zero fill for
//       padding) as four consecutive 32-channel sub-chunks j = 0..3 of one (kh, channel chunk) SUPERCHUNK; the packed
//       weights u_0..u_3 are laid out in the same (kh, chunk, p, c) order, so the W loader is the direct kernel's too;
//     * the input transform happens between the LDS read and the MFMA: 4 ds_read_b128 of raw pixels + 16 v_add/v_sub give
//       the four A fragments of an 8-deep k-step (4 x 4 MFMAs); the output transform is 6 adds per element in the epilogue;
//     * one barrier per superchunk (64 MFMAs per wave) instead of one per 32-deep chunk (32): the superchunk's 16 DMA
//       instructions per thread are spread over the 4096 MFMA cycles of the previous one.
// Block = 4 waves, 64 tiles (128 output pixels) x 64 channels, wave tile 32 tiles x 32 channels x 4 positions (64
// accumulator registers); LDS = 2 superstages x 4 sub-stages x (64 + 64) rows x 128 B = 128 KiB (one block per CU).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace capf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

static constexpr int WBK = 32;                 // channels per sub-chunk (128 B rows)
static constexpr int WBT = 64, WBN = 64;       // block tile: 64 tiles x 64 output channels
static constexpr int WSUB = (WBT + WBN) * WBK; // floats per sub-stage
static constexpr int WLDS = 2 * 4 * WSUB;      // 2 superstages x 4 sub-stages

#ifdef CAPF_DIAG   // (diagnosis build) per-block stamps: {t_entry, t_prologue_done, t_loop_done, t_exit, realtime_entry, hw_id, xcc_id, realtime_exit}
__device__ unsigned long long capf_wino_timeline[8192 * 8];
__device__ unsigned long long capf_wino_phases[8192 * 8];     // wino43s_tile: cycles per phase of the K loop, summed over the tile
#define WINO_STAMP(var) var = __builtin_amdgcn_s_memtime()
#define WINO_PHASE(k) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); dbg_ph[k] += _t - dbg_last; dbg_last = _t; } while (0)
#else
#define WINO_STAMP(var)
#define WINO_PHASE(k)
#endif

__device__ __forceinline__ int fast_div_w(int n, FastDiv d) {
    return (int)((__umulhi((unsigned)n, d.mul) + (unsigned)n) >> d.shift);
}

#if defined(__HIP_DEVICE_COMPILE__)
// One output tile (logical id `bid`) of problem p.  p.M = number of tiles (B * H * W / 2), p.Ho = H, p.Wo = W / 2 (tile grid),
// p.Kpad = 12 * Cin, p.fd_hw / p.fd_wo divide by H * (W/2) and W/2.
// PP = false: two superstages (128 KiB), the next superchunk's DMA instructions ride in the MFMA slots of the current one.
// PP = true ("ping-pong"): ONE superstage (64 KiB, two blocks per CU).  A block alternates a compute phase (64 MFMAs per wave,
//   fragments + transforms of the next k-step in the slots) with a load phase (16 DMA instructions per thread, wait, barrier)
//   and relies on the co-resident block — naturally out of phase, on the same SIMDs — to use the matrix pipe meanwhile; the
//   partner also covers the ~10 us a tile spends outside its K loop (first-load wait, residual loads, output stores), which
//   with one block per CU were fully exposed: 2048 tiles of the 64x64 layer-1 conv took 166 us for 82 us of MFMA time.
template <bool PP>
__device__ __forceinline__ void wino_tile(const GemmArgs& p, const int bid, float* __restrict__ lds) {
#ifdef CAPF_DIAG
    unsigned long long dbg_t0 = 0, dbg_t1 = 0, dbg_t2 = 0;
    const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    WINO_STAMP(dbg_t0);
    constexpr int NT = 256;
    constexpr int RPR = NT / 8;                          // 32 tile rows per DMA round
    constexpr int RA = WBT / RPR, RB = WBN / RPR;        // 2 + 2 DMA instructions per thread per sub-chunk
    constexpr int NSUBLOAD = RA + RB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nbn = (p.N + WBN - 1) / WBN;
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * WBT, n0 = tile_n * WBN;      // m0: first tile row

    const int srow = tid >> 3;
    const int kq = (((tid & 7) ^ ((srow >> 1) & 7))) * 4; // source-side XOR swizzle (see igemm_f32.hip)

    const int CC = p.Cin / WBK;                          // channel chunks
    const int nsc = 3 * CC;                              // superchunks

    constexpr unsigned OOB_A = 0x80000000u;
    long a_base;
    {
        const int b = fast_div_w(m0, p.fd_hw), rem = m0 - b * p.Ho * p.Wo;
        const int h = fast_div_w(rem, p.fd_wo), wt = rem - h * p.Wo;
        a_base = ((long)b * p.H * p.W + (long)(h - 1) * p.W + (2 * wt - 1)) * p.Cin;
    }
    const rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + a_base), 0, 0x7FFFFF00u, 0x00020000);
    const rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wp + (long)n0 * p.Kpad), 0,
                                                            (unsigned)(p.N - n0) * (unsigned)p.Kpad * 4u, 0x00020000);
    unsigned a_rel[RA], a_mask[RA];                      // mask bit kh * 4 + j: raw pixel j of input row kh is inside the image
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int t = m0 + srow + RPR * i;
        a_rel[i] = 0;
        a_mask[i] = 0u;
        if (t < p.M) {
            const int b = fast_div_w(t, p.fd_hw), rem = t - b * p.Ho * p.Wo;
            const int h = fast_div_w(rem, p.fd_wo), wt = rem - h * p.Wo;
            const int h0 = h - 1, w0 = 2 * wt - 1;
            const long off = ((long)b * p.H * p.W + (long)h0 * p.W + w0) * p.Cin;
            a_rel[i] = (unsigned)(off - a_base + kq) * 4u;
            const int j_lo = max(0, -w0), j_hi = min(4, p.W - w0);
            const int kh_lo = max(0, -h0), kh_hi = min(3, p.H - h0);
            if (j_hi > j_lo && kh_hi > kh_lo) {
                const unsigned wbits = ((1u << j_hi) - 1) & ~((1u << j_lo) - 1);
                const unsigned below_hi = (1u << (kh_hi * 4)) - 1, below_lo = (1u << (kh_lo * 4)) - 1;
                a_mask[i] = (wbits * 0x111u) & below_hi & ~below_lo;
            }
        }
    }
    unsigned w_off[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) w_off[i] = (unsigned)((srow + RPR * i) * p.Kpad + kq) * 4u;

    // walk of the sub-chunk being prepared: (kh, channel chunk, j), j fastest
    int u_kh = 0, u_cc = 0, u_j = 0;
    unsigned voff[NSUBLOAD];
    unsigned soff_a = 0;
    auto prepare = [&]() {
        const unsigned bit = u_kh < 3 ? (1u << (u_kh * 4 + u_j)) : 0u;
        soff_a = __builtin_amdgcn_readfirstlane((unsigned)((u_kh * p.W + u_j) * p.Cin + u_cc * WBK) * 4u);
#pragma unroll
        for (int i = 0; i < RA; ++i) voff[i] = (a_mask[i] & bit) ? a_rel[i] : OOB_A;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            voff[RA + i] = w_off[i];
            w_off[i] += WBK * 4u;
        }
        if (++u_j == 4) {
            u_j = 0;
            if (++u_cc == CC) { u_cc = 0; ++u_kh; }
        }
    };
    // fire load #idx of the prepared sub-chunk into sub-stage `sub` (0..7)
    auto fire = [&](int idx, int sub) {
        float* As = lds + sub * WSUB;
        if (idx < RA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(As + (idx * RPR + wave * 8) * WBK), 16, voff[idx], soff_a, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(As + WBT * WBK + ((idx - RA) * RPR + wave * 8) * WBK), 16,
                                                     voff[idx], 0, 0, 0);
    };

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int wm0 = (wave >> 1) * 32;          // tile rows of this wave inside the block tile
    const int wn0 = (wave & 1) * 32;           // channels of this wave
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    const int fhalf = lane >> 5;

    // prologue: superchunk 0 -> superstage 0
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        prepare();
#pragma unroll
        for (int i = 0; i < NSUBLOAD; ++i) fire(i, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // K loop.  Per 8-deep k-step a wave issues 16 MFMAs (slot i = 4 e + p: position p, k sub-step e) and, between them, the
    // step's other work, ONE piece per 64-cycle MFMA slot so that nothing but an LDS-DMA instruction (60-100 cycles of
    // issue, tools/dma_rate.hip) ever overflows its slot:
    //   slots 0-3    the next k-step's fragments, two ds_read_b128 each: (d2, d1) (d0, d3) (u0, u1) (u2, u3)
    //   slots 4-14   even: one DMA instruction of the NEXT superchunk (16 per superchunk: 6 + 6 + 4 over steps 0-2, each
    //                sub-chunk's offsets prepared just before its first load); odd 7, 9, 11, 13: the input transform of the
    //                next fragments, 4 VALU each, in the order their operands were read (v1, v2, v0, v3)
    //   step 3       no loads before slot 8; there: vmcnt(0) + lgkmcnt(0) + s_barrier (the next superchunk has landed, every
    //                wave has issued its last read of this one); slots 8-11 read the next superchunk's first fragments,
    //                slots 12-15 transform them.  The second half of step 3 runs from registers only.
    f32x4 dn[4];                               // raw pixels of the next k-step
    f32x4 v[2][4], uf[2][4];                   // transformed activations / weights of the current and next k-step
    const float* const a_ptr = lds + (wm0 + frow) * WBK;
    const float* const b_ptr = lds + WBT * WBK + (wn0 + frow) * WBK;
    auto rd_a = [&](int ss, int q, int j) { dn[j] = *reinterpret_cast<const f32x4*>(a_ptr + (ss * 4 + j) * WSUB + q * 4); };
    auto rd_b = [&](int ss, int q, int j, int buf) { uf[buf][j] = *reinterpret_cast<const f32x4*>(b_ptr + (ss * 4 + j) * WSUB + q * 4); };
    auto xform = [&](int pq, int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (pq == 0) v[buf][0][e] = dn[0][e] - dn[2][e];
            else if (pq == 1) v[buf][1][e] = dn[1][e] + dn[2][e];
            else if (pq == 2) v[buf][2][e] = dn[2][e] - dn[1][e];
            else v[buf][3][e] = dn[1][e] - dn[3][e];
        }
    };
    {
        const int q0 = fhalf ^ fsw;
#pragma unroll
        for (int j = 0; j < 4; ++j) { rd_a(0, q0, j); rd_b(0, q0, j, 0); }
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) xform(pq, 0);
    }
    WINO_STAMP(dbg_t1);
    // epilogue operands (bias, residual rows of the two output pixels): transposed accumulator -> this lane owns tile row
    // t = m0 + wm0 + (lane & 31) and, per register group g, four consecutive channels n = 8 g + 4 (lane >> 5) + e
    const int t = m0 + wm0 + (lane & 31);
    const bool t_ok = t < p.M;
    const long o_row = (long)(2 * t) * p.omap.S1 + p.omap.off;          // output pixel 2 t (and 2 t + 1: the next NHWC row)
    const long r_row = (long)(2 * t) * p.rmap.S1 + p.rmap.off;
    f32x4 bv[4], r0[4], r1[4];
    auto load_epilogue_operands = [&]() {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn0 + 4 * (lane >> 5) + 8 * g;
            bv[g] = r0[g] = r1[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (n < p.N) {
                if (p.bias) bv[g] = *reinterpret_cast<const f32x4*>(p.bias + n);
                if (p.res && t_ok) {
                    r0[g] = *reinterpret_cast<const f32x4*>(p.res + r_row + n);
                    r1[g] = *reinterpret_cast<const f32x4*>(p.res + r_row + p.rmap.S1 + n);
                }
            }
        }
    };
    if (PP) {
        for (int sc = 0; sc < nsc; ++sc) {
            // the last compute phase has no load phase behind it: its 64 MFMAs hide the latency of the residual rows
            if (sc == nsc - 1) load_epilogue_operands();
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                const int fb = step & 1, nb = fb ^ 1;
                const int q_next = ((step + 1) * 2 + fhalf) ^ fsw;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int pq = i & 3, e = i >> 2;
                    acc[pq] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[fb][pq][e], v[fb][pq][e], acc[pq], 0, 0, 0);
                    if (step < 3) {
                        if (i == 0) { rd_a(0, q_next, 2); rd_a(0, q_next, 1); }
                        else if (i == 1) { rd_a(0, q_next, 0); rd_a(0, q_next, 3); }
                        else if (i == 2) { rd_b(0, q_next, 0, nb); rd_b(0, q_next, 1, nb); }
                        else if (i == 3) { rd_b(0, q_next, 2, nb); rd_b(0, q_next, 3, nb); }
                        else if (i == 6) xform(1, nb);
                        else if (i == 8) xform(2, nb);
                        else if (i == 10) xform(0, nb);
                        else if (i == 12) xform(3, nb);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (sc + 1 < nsc) {
                // load phase: every wave has issued its last read of the superstage -> overwrite it with the next superchunk
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    prepare();
#pragma unroll
                    for (int i = 0; i < NSUBLOAD; ++i) fire(i, j);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const int q0 = fhalf ^ fsw;
#pragma unroll
                for (int j = 0; j < 4; ++j) { rd_a(0, q0, j); rd_b(0, q0, j, 0); }
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) xform(pq, 0);
            }
        }
    } else
    for (int sc = 0; sc < nsc; ++sc) {
        const int ss = sc & 1, sn = ss ^ 1;
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int fb = step & 1, nb = fb ^ 1;
            const int rs = step < 3 ? ss : sn;                                 // superstage the next fragments come from
            const int q_next = (((step + 1) & 3) * 2 + fhalf) ^ fsw;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int pq = i & 3, e = i >> 2;
                acc[pq] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[fb][pq][e], v[fb][pq][e], acc[pq], 0, 0, 0);
                if (step == 3 && i == 7) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                const int ri = step < 3 ? i : i - 8;                           // read / transform slots shift behind the barrier in step 3
                if (ri == 0) { rd_a(rs, q_next, 2); rd_a(rs, q_next, 1); }
                else if (ri == 1) { rd_a(rs, q_next, 0); rd_a(rs, q_next, 3); }
                else if (ri == 2) { rd_b(rs, q_next, 0, nb); rd_b(rs, q_next, 1, nb); }
                else if (ri == 3) { rd_b(rs, q_next, 2, nb); rd_b(rs, q_next, 3, nb); }
                if (step < 3) {
                    if (i >= 4 && (i & 1) == 0) {                              // DMA slots 4, 6, ..., 14
                        const int k = step * 6 + (i - 4) / 2;                  // DMA instruction 0..15 of the next superchunk
                        if (k < 16) {
                            if ((k & 3) == 0) prepare();
                            fire(k & 3, sn * 4 + (k >> 2));
                        }
                    }
                    if (i == 7) xform(1, nb);
                    else if (i == 9) xform(2, nb);
                    else if (i == 11) xform(0, nb);
                    else if (i == 13) xform(3, nb);
                } else {
                    if (i == 12) xform(1, nb);
                    else if (i == 13) xform(2, nb);
                    else if (i == 14) xform(0, nb);
                    else if (i == 15) xform(3, nb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WINO_STAMP(dbg_t2);

    // ---- epilogue: output transform + bias (+ residual) (+ ReLU)
    if (!PP) load_epilogue_operands();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn0 + 4 * (lane >> 5) + 8 * g;
        f32x4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = acc[0][4 * g + e], a1 = acc[1][4 * g + e], a2 = acc[2][4 * g + e], a3 = acc[3][4 * g + e];
            float s0 = ((a0 + a1) + a2) + bv[g][e] + r0[g][e];
            float s1 = ((a1 - a2) - a3) + bv[g][e] + r1[g][e];
            if (p.act == ACT_RELU) { s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f); }
            y0[e] = s0; y1[e] = s1;
        }
        if (t_ok && n < p.N) {
            *reinterpret_cast<f32x4*>(p.out + o_row + n) = y0;
            *reinterpret_cast<f32x4*>(p.out + o_row + p.omap.S1 + n) = y1;
        }
    }
#ifdef CAPF_DIAG
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long* d = capf_wino_timeline + (size_t)blockIdx.x * 8;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[3] = __builtin_amdgcn_s_memtime();
        d[4] = dbg_r0; d[5] = hw; d[6] = xcc; d[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// Half-size block tiles for the shapes the 64 x 64 tile does not fit: HBT x HBN = 64 tiles x 32 channels (the 32-channel
// high-resolution branch: a 64-wide tile would multiply zeros in half of its MFMAs) or 32 tiles x 64 channels (the 8 x 8
// branch: twice the blocks, so that every CU gets its ping-pong pair at batch 64).  The four waves are 2 sub-tiles (32 tiles x
// 32 channels each, split along M or N) x 2 POSITION PAIRS: wave (pp, sub) accumulates positions p = 2 pp, 2 pp + 1 — per
// 8-deep k-step 3 raw-pixel reads (d_pp .. d_pp+2) + 2 weight reads + 8 VALU feed 8 MFMAs — and the two pairs of a sub-tile
// meet in the epilogue through LDS (each wave finishes two of the four 8-channel register groups).  One superstage
// (4 x (HBT + HBN) x 128 B = 48 KiB), ping-pong schedule as wino_tile<true>.
template <int HBT, int HBN>
__device__ __forceinline__ void wino_tile_h(const GemmArgs& p, const int bid, float* __restrict__ lds) {
    constexpr int RPR = 32;
    constexpr int RA = HBT / RPR, RB = HBN / RPR, NSUBLOAD = RA + RB;
    constexpr int HSUB = (HBT + HBN) * WBK;
    static_assert((HBT == 64 && HBN == 32) || (HBT == 32 && HBN == 64), "half tiles");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nbn = (p.N + HBN - 1) / HBN;
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * HBT, n0 = tile_n * HBN;

    const int srow = tid >> 3;
    const int kq = (((tid & 7) ^ ((srow >> 1) & 7))) * 4;
    const int CC = p.Cin / WBK;
    const int nsc = 3 * CC;

    constexpr unsigned OOB_A = 0x80000000u;
    long a_base;
    {
        const int b = fast_div_w(m0, p.fd_hw), rem = m0 - b * p.Ho * p.Wo;
        const int h = fast_div_w(rem, p.fd_wo), wt = rem - h * p.Wo;
        a_base = ((long)b * p.H * p.W + (long)(h - 1) * p.W + (2 * wt - 1)) * p.Cin;
    }
    const rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + a_base), 0, 0x7FFFFF00u, 0x00020000);
    const rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wp + (long)n0 * p.Kpad), 0,
                                                            (unsigned)(p.N - n0) * (unsigned)p.Kpad * 4u, 0x00020000);
    unsigned a_rel[RA], a_mask[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int t = m0 + srow + RPR * i;
        a_rel[i] = 0;
        a_mask[i] = 0u;
        if (t < p.M) {
            const int b = fast_div_w(t, p.fd_hw), rem = t - b * p.Ho * p.Wo;
            const int h = fast_div_w(rem, p.fd_wo), wt = rem - h * p.Wo;
            const int h0 = h - 1, w0 = 2 * wt - 1;
            const long off = ((long)b * p.H * p.W + (long)h0 * p.W + w0) * p.Cin;
            a_rel[i] = (unsigned)(off - a_base + kq) * 4u;
            const int j_lo = max(0, -w0), j_hi = min(4, p.W - w0);
            const int kh_lo = max(0, -h0), kh_hi = min(3, p.H - h0);
            if (j_hi > j_lo && kh_hi > kh_lo) {
                const unsigned wbits = ((1u << j_hi) - 1) & ~((1u << j_lo) - 1);
                const unsigned below_hi = (1u << (kh_hi * 4)) - 1, below_lo = (1u << (kh_lo * 4)) - 1;
                a_mask[i] = (wbits * 0x111u) & below_hi & ~below_lo;
            }
        }
    }
    unsigned w_off[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) w_off[i] = (unsigned)((srow + RPR * i) * p.Kpad + kq) * 4u;

    int u_kh = 0, u_cc = 0, u_j = 0;
    unsigned voff[NSUBLOAD];
    unsigned soff_a = 0;
    auto prepare = [&]() {
        const unsigned bit = u_kh < 3 ? (1u << (u_kh * 4 + u_j)) : 0u;
        soff_a = __builtin_amdgcn_readfirstlane((unsigned)((u_kh * p.W + u_j) * p.Cin + u_cc * WBK) * 4u);
#pragma unroll
        for (int i = 0; i < RA; ++i) voff[i] = (a_mask[i] & bit) ? a_rel[i] : OOB_A;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            voff[RA + i] = w_off[i];
            w_off[i] += WBK * 4u;
        }
        if (++u_j == 4) {
            u_j = 0;
            if (++u_cc == CC) { u_cc = 0; ++u_kh; }
        }
    };
    auto fire = [&](int idx, int sub) {
        float* As = lds + sub * HSUB;
        if (idx < RA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(As + (idx * RPR + wave * 8) * WBK), 16, voff[idx], soff_a, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(As + HBT * WBK + ((idx - RA) * RPR + wave * 8) * WBK), 16,
                                                     voff[idx], 0, 0, 0);
    };
    auto load_superchunk = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            prepare();
#pragma unroll
            for (int i = 0; i < NSUBLOAD; ++i) fire(i, j);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int pp = wave >> 1;                  // position pair
    const int sub = wave & 1;                  // sub-tile
    const int wm0 = HBT == 64 ? sub * 32 : 0;
    const int wn0 = HBT == 64 ? 0 : sub * 32;
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    const int fhalf = lane >> 5;

    load_superchunk();

    f32x4 dn[3];                               // raw pixels d_pp, d_pp+1, d_pp+2 of the next k-step
    f32x4 v[2][2], uf[2][2];
    const float* const a_ptr = lds + (wm0 + frow) * WBK + pp * HSUB;
    const float* const b_ptr = lds + HBT * WBK + (wn0 + frow) * WBK + 2 * pp * HSUB;
    auto rd_a = [&](int q, int j) { dn[j] = *reinterpret_cast<const f32x4*>(a_ptr + j * HSUB + q * 4); };
    auto rd_b = [&](int q, int j, int buf) { uf[buf][j] = *reinterpret_cast<const f32x4*>(b_ptr + j * HSUB + q * 4); };
    // B^T d for this wave's two positions: pp = 0: (d0 - d2, d1 + d2); pp = 1 (dn = d1, d2, d3): (d2 - d1, d1 - d3)
    float cf[2][3];                            // B^T rows of this wave's pair as wave-uniform scalars (no per-lane selects)
    cf[0][0] = pp == 0 ? 1.f : -1.f; cf[0][1] = pp == 0 ? 0.f : 1.f; cf[0][2] = pp == 0 ? -1.f : 0.f;      // d0 - d2 | d2 - d1
    cf[1][0] = pp == 0 ? 0.f : 1.f;  cf[1][1] = pp == 0 ? 1.f : 0.f; cf[1][2] = pp == 0 ? 1.f : -1.f;      // d1 + d2 | d1 - d3
    auto xform = [&](int which, int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[buf][which][e] = (cf[which][0] * dn[0][e] + cf[which][1] * dn[1][e]) + cf[which][2] * dn[2][e];
    };
    auto first_frags = [&]() {
        const int q0 = fhalf ^ fsw;
#pragma unroll
        for (int j = 0; j < 3; ++j) rd_a(q0, j);
        rd_b(q0, 0, 0); rd_b(q0, 1, 0);
        xform(0, 0); xform(1, 0);
    };
    first_frags();

    // epilogue operands of the two register groups this wave finishes (g = 2 pp, 2 pp + 1)
    const int t = m0 + wm0 + (lane & 31);
    const bool t_ok = t < p.M;
    const long o_row = (long)(2 * t) * p.omap.S1 + p.omap.off;
    const long r_row = (long)(2 * t) * p.rmap.S1 + p.rmap.off;
    f32x4 bv[2], r0[2], r1[2];
    auto load_epilogue_operands = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = n0 + wn0 + 4 * (lane >> 5) + 8 * (2 * pp + k);
            bv[k] = r0[k] = r1[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (n < p.N) {
                if (p.bias) bv[k] = *reinterpret_cast<const f32x4*>(p.bias + n);
                if (p.res && t_ok) {
                    r0[k] = *reinterpret_cast<const f32x4*>(p.res + r_row + n);
                    r1[k] = *reinterpret_cast<const f32x4*>(p.res + r_row + p.rmap.S1 + n);
                }
            }
        }
    };

    for (int sc = 0; sc < nsc; ++sc) {
        if (sc == nsc - 1) load_epilogue_operands();
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int fb = step & 1, nb = fb ^ 1;
            const int q_next = ((step + 1) * 2 + fhalf) ^ fsw;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pq = i & 1, e = i >> 1;
                acc[pq] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[fb][pq][e], v[fb][pq][e], acc[pq], 0, 0, 0);
                if (step < 3) {
                    if (i == 0) { rd_a(q_next, 0); rd_a(q_next, 1); }
                    else if (i == 1) { rd_a(q_next, 2); rd_b(q_next, 0, nb); }
                    else if (i == 2) rd_b(q_next, 1, nb);
                    else if (i == 5) xform(0, nb);
                    else if (i == 6) xform(1, nb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (sc + 1 < nsc) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            load_superchunk();
            first_frags();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // every wave is done with the superstage: it becomes the exchange buffer

    // partial output transform:  pp = 0: (s0, s1) = (m0 + m1, m1);  pp = 1: (m2, -(m2 + m3));  y0 = s0 + s0', y1 = s1 + s1'
    const float q00 = 1.f, q01 = pp == 0 ? 1.f : 0.f, q10 = pp == 0 ? 0.f : -1.f, q11 = pp == 0 ? 1.f : -1.f;   // wave-uniform
    float* const xch = lds;                    // [sub][writer pp][4 slots][64 lanes] f32x4 = 16 KiB
    {
        float* dst = xch + (((sub * 2 + pp) * 4) * 64 + lane) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if ((g >> 1) == pp) continue;      // (wave-uniform) a group this wave finishes itself
            f32x4 s0, s1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = acc[0][4 * g + e], y = acc[1][4 * g + e];
                s0[e] = q00 * x + q01 * y;
                s1[e] = q10 * x + q11 * y;
            }
            *reinterpret_cast<f32x4*>(dst + (2 * (g & 1) + 0) * 64 * 4) = s0;
            *reinterpret_cast<f32x4*>(dst + (2 * (g & 1) + 1) * 64 * 4) = s1;
        }
    }
    __syncthreads();
    const float* src = xch + (((sub * 2 + (pp ^ 1)) * 4) * 64 + lane) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if ((g >> 1) != pp) continue;
        const int k = g & 1;
        const int n = n0 + wn0 + 4 * (lane >> 5) + 8 * g;
        const f32x4 o0 = *reinterpret_cast<const f32x4*>(src + (2 * k + 0) * 64 * 4);
        const f32x4 o1 = *reinterpret_cast<const f32x4*>(src + (2 * k + 1) * 64 * 4);
        f32x4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = acc[0][4 * g + e], y = acc[1][4 * g + e];
            const float m_s0 = q00 * x + q01 * y, m_s1 = q10 * x + q11 * y;
            // fixed order whichever wave finishes the group: (pair 0 partial) + (pair 1 partial)
            float s0 = (pp == 0 ? m_s0 + o0[e] : o0[e] + m_s0) + bv[k][e] + r0[k][e];
            float s1 = (pp == 0 ? m_s1 + o1[e] : o1[e] + m_s1) + bv[k][e] + r1[k][e];
            if (p.act == ACT_RELU) { s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f); }
            y0[e] = s0; y1[e] = s1;
        }
        if (t_ok && n < p.N) {
            *reinterpret_cast<f32x4*>(p.out + o_row + n) = y0;
            *reinterpret_cast<f32x4*>(p.out + o_row + p.omap.S1 + n) = y1;
        }
    }
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// F(4,3) along W: a tile is FOUR output pixels (w = 4 wt .. 4 wt + 3) computed from six raw pixels d_0..d_5
// (w_in = 4 wt - 1 + j) through SIX positions:
//   v = B^T d:  v0 = 4 d0 - 5 d2 + d4          v1 = -4 (d1 + d2) + (d3 + d4)      v2 = 4 (d1 - d2) + (d4 - d3)
//               v3 = 2 (d3 - d1) + (d4 - d2)   v4 = 2 (d1 - d3) + (d4 - d2)        v5 = 4 d1 - 5 d3 + d5
//   u = G g:    u0 = g0 / 4   u1 = -(g0 + g1 + g2) / 6   u2 = -(g0 - g1 + g2) / 6   u3 = g0 / 24 + g1 / 12 + g2 / 6
//               u4 = g0 / 24 - g1 / 12 + g2 / 6   u5 = g2
//   y = A^T m:  y0 = m0 + m1 + m2 + m3 + m4      y1 = (m1 - m2) + 2 (m3 - m4)
//               y2 = (m1 + m2) + 4 (m3 + m4)     y3 = (m1 - m2) + 8 (m3 - m4) + m5
// 18 MACs per four outputs = 4.5 per pixel: HALF the MFMAs of the direct conv (F(2,3): two thirds).  fp32 error against
// an fp64 conv: 4e-6 .. 9e-6 absolute on O(4) outputs (direct: 1e-6 .. 2e-6) — three orders below the 1e-3 bar.
// Launched on its own it is no faster than F(2,3) (batch 64: 64x64 64->64 162 vs 144 us, the 32x32 / 16x16 / 8x8 branches 39 /
// 48 / 90 vs 38 / 38 / 48 us, the 32-channel branch 45 vs 48 us): with 48 instead of 64 MFMAs per superchunk the load phase
// (18 DMA instructions per thread + 8 fragment reads + 60 VALU of first-fragment transforms) outlasts the compute phase — the
// timeline shows 5.35 us per superchunk per block against 3.25 us of MFMA time, and offsetting the second resident block of a
// CU by one compute phase at start (LDS_ALLOC base != 0; tried) does not move it.  Inside the grouped launch of an HRNet
// level, where blocks of four different shapes share the CUs, the saved MFMAs do show: a 4-branch level 157 -> 146 us and
// 5143 -> 5307 frames/s end to end (selecting it only for the large maps: 5183-5272), so the plan uses it wherever W % 4 == 0.
// Block = 4 waves = 2 sub-tiles (32 tiles x 32 channels) x 2 position TRIPLES: triple 0 (p0..p2) reads d0..d4, triple 1
// (p3..p5) reads d1..d5; per 8-deep k-step 5 raw reads + 3 weight reads + 48 VALU feed 12 MFMAs; the triples meet in the
// epilogue through LDS.  One superstage of six sub-chunks (6 x 96 rows x 128 B = 72 KiB, two blocks per CU), ping-pong
// schedule as wino_tile<true>.  Block tiles 64 tiles x 32 channels or 32 tiles x 64 channels (= 128 output pixels x 64 / 256 x 32).
template <int HBT, int HBN>
__device__ __forceinline__ void wino43_tile(const GemmArgs& p, const int bid, float* __restrict__ lds) {
#ifdef CAPF_DIAG
    unsigned long long dbg_t0 = 0, dbg_t1 = 0, dbg_t2 = 0;
    const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    WINO_STAMP(dbg_t0);
    constexpr int RPR = 32;
    constexpr int RA = HBT / RPR, RB = HBN / RPR, NSUBLOAD = RA + RB;
    constexpr int HSUB = (HBT + HBN) * WBK;
    static_assert((HBT == 64 && HBN == 32) || (HBT == 32 && HBN == 64), "F(4,3) tiles");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nbn = (p.N + HBN - 1) / HBN;
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * HBT, n0 = tile_n * HBN;

    const int srow = tid >> 3;
    const int kq = (((tid & 7) ^ ((srow >> 1) & 7))) * 4;
    const int CC = p.Cin / WBK;
    const int nsc = 3 * CC;

    constexpr unsigned OOB_A = 0x80000000u;
    long a_base;
    {
        const int b = fast_div_w(m0, p.fd_hw), rem = m0 - b * p.Ho * p.Wo;
        const int h = fast_div_w(rem, p.fd_wo), wt = rem - h * p.Wo;
        a_base = ((long)b * p.H * p.W + (long)(h - 1) * p.W + (4 * wt - 1)) * p.Cin;
    }
    const rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + a_base), 0, 0x7FFFFF00u, 0x00020000);
    const rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wp + (long)n0 * p.Kpad), 0,
                                                            (unsigned)(p.N - n0) * (unsigned)p.Kpad * 4u, 0x00020000);
    unsigned a_rel[RA], a_mask[RA];                      // mask bit kh * 8 + j
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int t = m0 + srow + RPR * i;
        a_rel[i] = 0;
        a_mask[i] = 0u;
        if (t < p.M) {
            const int b = fast_div_w(t, p.fd_hw), rem = t - b * p.Ho * p.Wo;
            const int h = fast_div_w(rem, p.fd_wo), wt = rem - h * p.Wo;
            const int h0 = h - 1, w0 = 4 * wt - 1;
            const long off = ((long)b * p.H * p.W + (long)h0 * p.W + w0) * p.Cin;
            a_rel[i] = (unsigned)(off - a_base + kq) * 4u;
            const int j_lo = max(0, -w0), j_hi = min(6, p.W - w0);
            const int kh_lo = max(0, -h0), kh_hi = min(3, p.H - h0);
            if (j_hi > j_lo && kh_hi > kh_lo) {
                const unsigned wbits = ((1u << j_hi) - 1) & ~((1u << j_lo) - 1);
                const unsigned below_hi = (1u << (kh_hi * 8)) - 1, below_lo = (1u << (kh_lo * 8)) - 1;
                a_mask[i] = (wbits * 0x10101u) & below_hi & ~below_lo;
            }
        }
    }
    unsigned w_off[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) w_off[i] = (unsigned)((srow + RPR * i) * p.Kpad + kq) * 4u;

    int u_kh = 0, u_cc = 0, u_j = 0;
    unsigned voff[NSUBLOAD];
    unsigned soff_a = 0;
    auto prepare = [&]() {
        const unsigned bit = u_kh < 3 ? (1u << (u_kh * 8 + u_j)) : 0u;
        soff_a = __builtin_amdgcn_readfirstlane((unsigned)((u_kh * p.W + u_j) * p.Cin + u_cc * WBK) * 4u);
#pragma unroll
        for (int i = 0; i < RA; ++i) voff[i] = (a_mask[i] & bit) ? a_rel[i] : OOB_A;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            voff[RA + i] = w_off[i];
            w_off[i] += WBK * 4u;
        }
        if (++u_j == 6) {
            u_j = 0;
            if (++u_cc == CC) { u_cc = 0; ++u_kh; }
        }
    };
    auto fire = [&](int idx, int sub) {
        float* As = lds + sub * HSUB;
        if (idx < RA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(As + (idx * RPR + wave * 8) * WBK), 16, voff[idx], soff_a, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(As + HBT * WBK + ((idx - RA) * RPR + wave * 8) * WBK), 16,
                                                     voff[idx], 0, 0, 0);
    };
    auto load_superchunk = [&]() {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            prepare();
#pragma unroll
            for (int i = 0; i < NSUBLOAD; ++i) fire(i, j);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int pp = wave >> 1;                  // position triple
    const int sub = wave & 1;                  // sub-tile
    const int wm0 = HBT == 64 ? sub * 32 : 0;
    const int wn0 = HBT == 64 ? 0 : sub * 32;
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    const int fhalf = lane >> 5;

    load_superchunk();

    f32x4 dn[5];                               // raw pixels d_pp .. d_pp+4 of the next k-step
    f32x4 v[2][3], uf[2][3];
    const float* const a_ptr = lds + (wm0 + frow) * WBK + pp * HSUB;
    const float* const b_ptr = lds + HBT * WBK + (wn0 + frow) * WBK + 3 * pp * HSUB;
    auto rd_a = [&](int q, int j) { dn[j] = *reinterpret_cast<const f32x4*>(a_ptr + j * HSUB + q * 4); };
    auto rd_b = [&](int q, int j, int buf) { uf[buf][j] = *reinterpret_cast<const f32x4*>(b_ptr + j * HSUB + q * 4); };
    // one (position, k sub-step) unit of the input transform: which = 0, 1, 2 = this wave's first / second / third position
    // (triple 0: p0 p1 p2 on d0..d4; triple 1: p3 p4 p5 on d1..d5), 5 VALU each — small enough to ride in an MFMA shadow
    // B^T rows of this wave's triple as wave-uniform scalars (SGPRs): no per-lane select between the two triples' formulas
    float cf[3][5];
    {
        const float t0[3][5] = {{4.f, 0.f, -5.f, 0.f, 1.f}, {0.f, -4.f, -4.f, 1.f, 1.f}, {0.f, 4.f, -4.f, -1.f, 1.f}};     // p0 p1 p2 on d0..d4
        const float t1[3][5] = {{-2.f, -1.f, 2.f, 1.f, 0.f}, {2.f, -1.f, -2.f, 1.f, 0.f}, {4.f, 0.f, -5.f, 0.f, 1.f}};     // p3 p4 p5 on d1..d5
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) cf[a][b] = pp == 0 ? t0[a][b] : t1[a][b];
    }
    auto xform1 = [&](int which, int e, int buf) {
        const float x0 = dn[0][e], x1 = dn[1][e], x2 = dn[2][e], x3 = dn[3][e], x4 = dn[4][e];
        v[buf][which][e] = ((cf[which][0] * x0 + cf[which][1] * x1) + (cf[which][2] * x2 + cf[which][3] * x3)) + cf[which][4] * x4;
    };
    auto xform = [&](int which, int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xform1(which, e, buf);
    };
    auto first_frags = [&]() {
        const int q0 = fhalf ^ fsw;
#pragma unroll
        for (int j = 0; j < 5; ++j) rd_a(q0, j);
#pragma unroll
        for (int j = 0; j < 3; ++j) rd_b(q0, j, 0);
        xform(0, 0); xform(1, 0); xform(2, 0);
    };
    first_frags();
    WINO_STAMP(dbg_t1);

    // epilogue operands of the two register groups this wave finishes (g = 2 pp, 2 pp + 1), four output pixels each
    const int t = m0 + wm0 + (lane & 31);
    const bool t_ok = t < p.M;
    const long o_row = (long)(4 * t) * p.omap.S1 + p.omap.off;
    const long r_row = (long)(4 * t) * p.rmap.S1 + p.rmap.off;
    f32x4 bv[2], rr[2][4];
    auto load_epilogue_operands = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = n0 + wn0 + 4 * (lane >> 5) + 8 * (2 * pp + k);
            bv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < 4; ++o) rr[k][o] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (n < p.N) {
                if (p.bias) bv[k] = *reinterpret_cast<const f32x4*>(p.bias + n);
                if (p.res && t_ok) {
#pragma unroll
                    for (int o = 0; o < 4; ++o) rr[k][o] = *reinterpret_cast<const f32x4*>(p.res + r_row + (long)o * p.rmap.S1 + n);
                }
            }
        }
    };

    for (int sc = 0; sc < nsc; ++sc) {
        if (sc == nsc - 1) load_epilogue_operands();
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int fb = step & 1, nb = fb ^ 1;
            const int q_next = ((step + 1) * 2 + fhalf) ^ fsw;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int pq = i % 3, e = i / 3;
                acc[pq] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[fb][pq][e], v[fb][pq][e], acc[pq], 0, 0, 0);
                if (step < 3) {
                    if (i == 0) { rd_a(q_next, 0); rd_a(q_next, 1); }
                    else if (i == 1) { rd_a(q_next, 2); rd_a(q_next, 3); }
                    else if (i == 2) { rd_a(q_next, 4); rd_b(q_next, 0, nb); }
                    else if (i == 3) { rd_b(q_next, 1, nb); rd_b(q_next, 2, nb); }
                    else if (i >= 6) {                 // 12 transform units over slots 6..11, two per slot
                        const int u0 = (i - 6) * 2, u1 = u0 + 1;
                        xform1(u0 / 4, u0 % 4, nb);
                        xform1(u1 / 4, u1 % 4, nb);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (sc + 1 < nsc) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            load_superchunk();
            first_frags();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WINO_STAMP(dbg_t2);
    __builtin_amdgcn_s_barrier();              // every wave is done with the superstage: it becomes the exchange buffer

    // partial output transform of this wave's triple (s[o] = contribution to output pixel o):
    //   triple 0 (m0 m1 m2): m0 + (m1 + m2),  m1 - m2,        m1 + m2,        m1 - m2
    //   triple 1 (m3 m4 m5): m3 + m4,         2 (m3 - m4),    4 (m3 + m4),    8 (m3 - m4) + m5
    float qf[4][3];                            // A^T columns of this wave's triple, wave-uniform
    {
        const float t0[4][3] = {{1.f, 1.f, 1.f}, {0.f, 1.f, -1.f}, {0.f, 1.f, 1.f}, {0.f, 1.f, -1.f}};
        const float t1[4][3] = {{1.f, 1.f, 0.f}, {2.f, -2.f, 0.f}, {4.f, 4.f, 0.f}, {8.f, -8.f, 1.f}};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) qf[a][b] = pp == 0 ? t0[a][b] : t1[a][b];
    }
    auto partial = [&](int g, int o, int e) -> float {      // g, o, e are compile-time after unrolling: static register indices
        return (qf[o][0] * acc[0][4 * g + e] + qf[o][1] * acc[1][4 * g + e]) + qf[o][2] * acc[2][4 * g + e];
    };
    float* const xch = lds;                    // [sub][writer pp][8 slots][64 lanes] f32x4 = 32 KiB
    {
        float* dst = xch + (((sub * 2 + pp) * 8) * 64 + lane) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if ((g >> 1) == pp) continue;      // (wave-uniform) a group this wave finishes itself
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                f32x4 sv;
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[e] = partial(g, o, e);
                *reinterpret_cast<f32x4*>(dst + (4 * (g & 1) + o) * 64 * 4) = sv;
            }
        }
    }
    __syncthreads();
    const float* src = xch + (((sub * 2 + (pp ^ 1)) * 8) * 64 + lane) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if ((g >> 1) != pp) continue;
        const int k = g & 1;
        const int n = n0 + wn0 + 4 * (lane >> 5) + 8 * g;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const f32x4 other = *reinterpret_cast<const f32x4*>(src + (4 * k + o) * 64 * 4);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float mine = partial(g, o, e);
                // fixed order whichever wave finishes the group: (triple 0 partial) + (triple 1 partial)
                float sv = (pp == 0 ? mine + other[e] : other[e] + mine) + bv[k][e] + rr[k][o][e];
                if (p.act == ACT_RELU) sv = fmaxf(sv, 0.f);
                y[e] = sv;
            }
            if (t_ok && n < p.N) *reinterpret_cast<f32x4*>(p.out + o_row + (long)o * p.omap.S1 + n) = y;
        }
    }
#ifdef CAPF_DIAG
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long* d = capf_wino_timeline + (size_t)blockIdx.x * 8;
        d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[3] = __builtin_amdgcn_s_memtime();
        d[4] = dbg_r0; d[5] = 0; d[6] = 0; d[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// F(4,3) with 16-channel sub-chunks ("short" superchunks): the same maths, tile shapes, wave roles and epilogue as wino43_tile,
// but a superchunk is (kh, 16 channels, 6 raw pixels): 6 x 96 rows x 64 B = 36 KiB of LDS instead of 72, and the kernel that
// hosts it is built for THREE resident blocks per CU (<= 168 registers).  Why: a ping-pong block alternates a compute phase
// and a load phase of about the same length and cannot overlap them itself, so a launch lasts (sum of the blocks' serial
// times) / (resident blocks) unless the matrix pipe saturates first.  With two residents at ~48 % duty each the pipe idles
// whenever both load and stalls whenever both compute (measured: 0.45 of the nominal MFMA rate, 1.8 blocks resident on
// average); a third resident fills those holes.  Per superchunk a wave issues 24 MFMAs (2 k-steps of 12) and 9 LDS-DMA
// instructions (3 per sub-chunk pair).
//   LDS image of a superchunk: raw-pixel sub-chunks A_0..A_5 (HBT rows x 64 B each), then weight positions W_0..W_5 (HBN rows):
//   576 rows = 9 DMA rounds of 64 rows (4 lanes per row); 16-byte quad q of row r lives at position q ^ ((r >> 2) & 3):
//   conflict-free ds_read_b128.
// The packed weights are wino43_tile's ([N][(kh, 32-channel chunk, p, c)], K'' = 18 Cin): both kernels share one copy.
// DB = true: TWO superstages (72 KiB, two resident blocks): the 9 DMA instructions of superchunk s + 1 ride in MFMA slots of
// superchunk s, so a block's own load latency hides behind its own MFMAs and only [wait, barrier, first fragments] stays
// exposed between compute phases (the partner block covers that).  Measured motive (tools/wino_level_timeline.py, HRNet-32
// level at batch 64, DB = false): 2.4 - 3.9 us per superchunk per block against 0.79 us of MFMA time -- the launch lasts as
// long as the 48-superchunk tiles of the 8 x 8 branch take to crawl through their load phases (115 of 133 us).
template <int HBT, int HBN, bool DB>
__device__ __forceinline__ void wino43s_tile(const GemmArgs& p, const int bid, float* __restrict__ lds) {
#ifdef CAPF_DIAG
    unsigned long long dbg_t0 = 0, dbg_t1 = 0, dbg_t2 = 0, dbg_last = 0;
    unsigned long long dbg_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 compute, 1 drain + barrier, 2 DMA issue, 3 data wait, 4 barrier, 5 first fragments
    const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    WINO_STAMP(dbg_t0);
    constexpr int BK = 16;
    constexpr int PSUB_A = HBT * BK, PSUB_W = HBN * BK;          // floats per sub-chunk of each operand
    constexpr int NRA = 6 * HBT / 64, NRW = 6 * HBN / 64;        // DMA rounds (64 rows x 64 B) per superchunk: (6, 3) or (3, 6)
    static_assert((HBT == 64 && HBN == 32) || (HBT == 32 && HBN == 64), "F(4,3) tiles");
    static_assert(NRA + NRW == 9, "nine DMA instructions per thread and superchunk");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nbn = (p.N + HBN - 1) / HBN;
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * HBT, n0 = tile_n * HBN;

    const int srow = tid >> 2;                                   // row of this thread inside a 64-row DMA round
    const int kq = ((tid & 3) ^ ((srow >> 2) & 3)) * 4;          // source-side XOR swizzle of the 16-byte quad
    const int arow = HBT == 64 ? srow : (srow & 31);             // tile row / weight row this thread stages (fixed for the tile)
    const int asub = HBT == 64 ? 0 : (srow >> 5);                // ... and, where one round spans two sub-chunks, which of them
    const int wrow = HBN == 64 ? srow : (srow & 31);
    const int wsub = HBN == 64 ? 0 : (srow >> 5);
    const int CC = p.Cin / BK;                                   // 16-channel chunks
    const int nsc = 3 * CC;

    constexpr unsigned OOB_A = 0x80000000u;
    long a_base;
    {
        const int b = fast_div_w(m0, p.fd_hw), rem = m0 - b * p.Ho * p.Wo;
        const int h = fast_div_w(rem, p.fd_wo), wt = rem - h * p.Wo;
        a_base = ((long)b * p.H * p.W + (long)(h - 1) * p.W + (4 * wt - 1)) * p.Cin;
    }
    const rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + a_base), 0, 0x7FFFFF00u, 0x00020000);
    const rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wp + (long)n0 * p.Kpad), 0,
                                                            (unsigned)(p.N - n0) * (unsigned)p.Kpad * 4u, 0x00020000);
    unsigned a_rel = 0, a_mask = 0u;                             // mask bit kh * 8 + j (already shifted by this thread's asub)
    {
        const int t = m0 + arow;
        if (t < p.M) {
            const int b = fast_div_w(t, p.fd_hw), rem = t - b * p.Ho * p.Wo;
            const int h = fast_div_w(rem, p.fd_wo), wt = rem - h * p.Wo;
            const int h0 = h - 1, w0 = 4 * wt - 1;
            const long off = ((long)b * p.H * p.W + (long)h0 * p.W + w0) * p.Cin;
            a_rel = (unsigned)(off - a_base + kq + asub * p.Cin) * 4u;
            const int j_lo = max(0, -w0), j_hi = min(6, p.W - w0);
            const int kh_lo = max(0, -h0), kh_hi = min(3, p.H - h0);
            if (j_hi > j_lo && kh_hi > kh_lo) {
                const unsigned wbits = ((1u << j_hi) - 1) & ~((1u << j_lo) - 1);
                const unsigned below_hi = (1u << (kh_hi * 8)) - 1, below_lo = (1u << (kh_lo * 8)) - 1;
                a_mask = ((wbits * 0x10101u) & below_hi & ~below_lo) >> asub;
            }
        }
    }
    // weights: row wrow of the block's N range; position p = round's first position + wsub; 16-channel half (cc & 1) of chunk cc >> 1
    const unsigned w_rel = (unsigned)(wrow * p.Kpad + kq + wsub * 32) * 4u;

    // LDS image of a superchunk: A_j (HBT rows x 64 B) at j * PSUB_A, j = 0..5, then W_p (HBN rows) at 6 PSUB_A + p * PSUB_W
    constexpr int STAGE = 6 * (PSUB_A + PSUB_W);                 // floats per superstage (9216 = 36 KiB)
    int u_kh = 0, u_cc = 0;                                      // walk of the superchunk being staged: chunk fastest, then kh
    unsigned soff_a = 0, soff_w = 0, row_mask = 0;
    auto prepare = [&]() {                                       // offsets of the next superchunk to stage
        soff_a = __builtin_amdgcn_readfirstlane((unsigned)(u_kh * p.W * p.Cin + u_cc * BK) * 4u);
        soff_w = __builtin_amdgcn_readfirstlane((unsigned)((u_kh * (CC >> 1) + (u_cc >> 1)) * 6 * 32 + (u_cc & 1) * BK) * 4u);
        row_mask = (a_mask >> (u_kh * 8)) & 0xffu;               // bit j: raw pixel j (+ asub) of this input row is inside the image
        if (++u_cc == CC) { u_cc = 0; ++u_kh; }
    };
    auto fire = [&](int k, float* stage) {                       // DMA instruction k = 0..8 of the prepared superchunk
        if (k < NRA) {                                           // round k: HBT == 64: raw pixel j = k; HBT == 32: j = 2 k + asub
            const int j = HBT == 64 ? k : 2 * k;
            const unsigned vo = (row_mask & (1u << j)) ? a_rel : OOB_A;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(stage + (k * 64 + wave * 16) * BK), 16, vo,
                                                     soff_a + (unsigned)(j * p.Cin * 4), 0, 0);
        } else {                                                 // round i: HBN == 64: position i; HBN == 32: 2 i + wsub
            const int i = k - NRA, pos = HBN == 64 ? i : 2 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(stage + 6 * PSUB_A + (i * 64 + wave * 16) * BK), 16, w_rel,
                                                     soff_w + (unsigned)(pos * 32 * 4), 0, 0);
        }
    };
    auto load_superchunk = [&](float* stage) {
        prepare();
#pragma unroll
        for (int k = 0; k < 9; ++k) fire(k, stage);
        WINO_PHASE(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WINO_PHASE(3);
        __builtin_amdgcn_s_barrier();
        WINO_PHASE(4);
    };

    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int pp = wave >> 1;                  // position triple
    const int sub = wave & 1;                  // sub-tile
    const int wm0 = HBT == 64 ? sub * 32 : 0;
    const int wn0 = HBT == 64 ? 0 : sub * 32;
    const int frow = lane & 31;
    const int fsw = (frow >> 2) & 3;
    const int fhalf = lane >> 5;

    load_superchunk(lds);

    f32x4 dn[5];                               // raw pixels d_pp .. d_pp+4 of the next k-step
    f32x4 v[2][3], uf[2][3];
    const float* const a_ptr = lds + (wm0 + frow) * BK + pp * PSUB_A;              // raw pixel j = pp + jj
    const float* const b_ptr = lds + 6 * PSUB_A + (wn0 + frow) * BK + 3 * pp * PSUB_W; // weight position 3 pp + k
    int so = 0;                                // float offset of the superstage being consumed (DB: 0 / STAGE)
    auto rd_a = [&](int q, int jj) { dn[jj] = *reinterpret_cast<const f32x4*>(a_ptr + so + jj * PSUB_A + q * 4); };
    auto rd_b = [&](int q, int k, int buf) { uf[buf][k] = *reinterpret_cast<const f32x4*>(b_ptr + so + k * PSUB_W + q * 4); };
    // Input transform of one k sub-step e for this wave's triple, SPECIALISED per triple (a wave-uniform branch picks the loop
    // body): with the B^T rows as run-time scalars every unit cost 6 VALU (two of five coefficients are 0 or +-1 in every row
    // but the compiler cannot know), 72 per k-step; written out they are 8 (triple 0) / 6 (triple 1) per sub-step = 32 / 24:
    //   triple 0 (d0..d4):  p0 = 4 d0 - 5 d2 + d4     p1 = -4 (d1 + d2) + (d3 + d4)     p2 = 4 (d1 - d2) + (d4 - d3)
    //   triple 1 (d1..d5):  p3 = 2 (d3 - d1) + (d4 - d2)     p4 = -2 (d3 - d1) + (d4 - d2)     p5 = 4 d1 - 5 d3 + d5
    auto xform_e = [&](auto ppc, int e, int buf) {
        const float x0 = dn[0][e], x1 = dn[1][e], x2 = dn[2][e], x3 = dn[3][e], x4 = dn[4][e];
        if constexpr (decltype(ppc)::value == 0) {
            v[buf][0][e] = __builtin_fmaf(-5.f, x2, __builtin_fmaf(4.f, x0, x4));
            v[buf][1][e] = __builtin_fmaf(-4.f, x1 + x2, x3 + x4);
            v[buf][2][e] = __builtin_fmaf(4.f, x1 - x2, x4 - x3);
        } else {
            const float a = x2 - x0, b = x3 - x1;
            v[buf][0][e] = __builtin_fmaf(2.f, a, b);
            v[buf][1][e] = __builtin_fmaf(-2.f, a, b);
            v[buf][2][e] = __builtin_fmaf(-5.f, x2, __builtin_fmaf(4.f, x0, x4));
        }
    };
    auto first_frags = [&](auto ppc) {
        const int q0 = fhalf ^ fsw;
#pragma unroll
        for (int j = 0; j < 5; ++j) rd_a(q0, j);
#pragma unroll
        for (int j = 0; j < 3; ++j) rd_b(q0, j, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) xform_e(ppc, e, 0);
    };
    // Output tile of the block in LDS order: PXT output pixels x NQ 16-byte quads (32 KiB for both tile shapes).  The epilogue
    // assembles it in LDS and every thread finishes EIGHT row-contiguous quads (piece i * 256 + tid): residual loads and stores are
    // 16 B per lane with consecutive lanes on consecutive addresses (a whole 128 / 256 B pixel row per 8 / 16 lanes).  In the
    // accumulator layout a store instruction touched 32 different cache lines for 32 B each (lanes = tiles, 4 pixels apart):
    // 4x the line touches for the same bytes on the one address path the LDS-DMA loads of the co-resident blocks also need.
    // The residual quads are requested right behind the K loop: the output transform, its two barriers and 16-32 LDS accesses
    // per lane cover their latency, and no register is held through the K loop for them (prefetching them before the last
    // compute phase needs 32 more registers there: 58 spilled at the three-resident budget).
    constexpr int NQ = HBN / 4, PXT = 4 * HBT;
    static_assert(PXT * NQ == 2048, "32 KiB output tile");
    // Every epilogue access is a raw buffer load / store on a block-local descriptor: an invalid piece (ragged M / N) gets an
    // out-of-range offset instead of a branch -- with divergent branches around them the compiler cannot count vmcnt and put
    // `s_waitcnt vmcnt(0)` behind every bias load and in front of every store (each store then waited for the previous one to
    // COMPLETE: 7-11 us of epilogue per block).  A null residual / bias is a descriptor of zero records: it reads as zeros.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned OOB_E = 0x80000000u;
    const rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        p.res ? (void*)(p.res + (long)(4 * m0) * p.rmap.S1 + p.rmap.off + n0) : (void*)p.out, 0, p.res ? 0x7FFFFF00u : 0u, 0x00020000);
    const rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (long)(4 * m0) * p.omap.S1 + p.omap.off + n0), 0,
                                                            0x7FFFFF00u, 0x00020000);
    const rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)(p.bias + n0) : (void*)p.out, 0,
                                                             p.bias ? (unsigned)(p.N - n0) * 4u : 0u, 0x00020000);
    f32x4 rr[8];
    auto piece_off = [&](int i, long S1) -> unsigned {     // byte offset of piece i * 256 + tid inside the block's output tile
        const int piece = i * 256 + tid;
        const int r = piece / NQ;                           // pixel row of the tile, channel quad (undoing the LDS swizzle)
        const int q = (piece - r * NQ) ^ ((r >> 2) & (NQ - 1));
        const bool ok = (m0 + (r >> 2)) < p.M && (n0 + 4 * q) < p.N;
        return ok ? (unsigned)(r * (int)S1 + 4 * q) * 4u : OOB_E;
    };
    auto load_epilogue_operands = [&](int half) {          // residual quads 4 half .. 4 half + 3 of this thread
#pragma unroll
        for (int i = 4 * half; i < 4 * half + 4; ++i)
            rr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, piece_off(i, p.rmap.S1), 0, 0));
    };
    auto kloop = [&](auto ppc) {
    first_frags(ppc);
    WINO_STAMP(dbg_t1);
#ifdef CAPF_DIAG
    dbg_last = dbg_t1;
    for (int k = 0; k < 8; ++k) dbg_ph[k] = 0;
#endif

    auto superchunk = [&](auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        float* const next_stage = lds + (DB ? STAGE - so : 0);
        if (DB && more) prepare();
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            const int fb = step & 1, nb = fb ^ 1;
            const int q_next = (2 + fhalf) ^ fsw;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int pq = i % 3, e = i / 3;
                acc[pq] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[fb][pq][e], v[fb][pq][e], acc[pq], 0, 0, 0);
                if (step == 0) {
                    if (i == 0) { rd_a(q_next, 0); rd_a(q_next, 1); }
                    else if (i == 1) { rd_a(q_next, 2); rd_a(q_next, 3); }
                    else if (i == 2) { rd_a(q_next, 4); rd_b(q_next, 0, nb); }
                    else if (i == 3) { rd_b(q_next, 1, nb); rd_b(q_next, 2, nb); }
                    else if (i >= 6 && i < 10) xform_e(ppc, i - 6, nb);      // the next k-step's transforms, one sub-step per slot
                }
                if (DB && more) {                      // the next superchunk's nine DMA instructions, one per MFMA slot
                    const int k = step == 0 ? i - 4 : i + 8;        // step 0 slots 4..11 -> 0..7, step 1 slot 0 -> 8
                    if (k >= 0 && k < 9) fire(k, next_stage);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        WINO_PHASE(0);
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (DB) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WINO_PHASE(3);
                __builtin_amdgcn_s_barrier();          // the next superchunk has landed; every wave is done reading this one
                WINO_PHASE(4);
                so = STAGE - so;
            } else {
                __builtin_amdgcn_s_barrier();
                WINO_PHASE(1);
                load_superchunk(lds);
            }
            first_frags(ppc);
            WINO_PHASE(5);
        }
    };
    for (int sc = 0; sc + 1 < nsc; ++sc) superchunk(std::true_type{});
    // last superchunk peeled: HALF of the residual quads are requested in front of it (its 24 MFMAs and the co-residents hide
    // their latency; 16 registers that are live only here, not around the loop -- all eight would need 182 registers, two
    // residents), the other half right behind it, covered by the output transform's two barriers and LDS passes
    load_epilogue_operands(0);
    superchunk(std::false_type{});
    };
    if (pp == 0) kloop(std::integral_constant<int, 0>{});
    else kloop(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WINO_STAMP(dbg_t2);
    // bias first (triple-0 waves need it right behind the barrier): vmcnt retires in order, so a bias load issued BEHIND the
    // eight residual loads would make its consumer wait for all of them
    f32x4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)                // (n >= N falls outside the descriptor: zeros)
        bq[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, (unsigned)(wn0 + 8 * g + 4 * fhalf) * 4u, 0, 0));
    load_epilogue_operands(1);
    __builtin_amdgcn_s_barrier();              // every wave is done with the superstage: it becomes the output tile

    // ---- epilogue: output transform of this wave's triple (A^T columns written out per triple, like the input transform)
    //   triple 0 (m0 m1 m2): y0 += m0 + (m1 + m2)   y1 += m1 - m2         y2 += m1 + m2         y3 += m1 - m2
    //   triple 1 (m3 m4 m5): y0 += m3 + m4          y1 += 2 (m3 - m4)     y2 += 4 (m3 + m4)     y3 += 8 (m3 - m4) + m5
    // Triple 0 writes (its part + bias) into the LDS output tile, triple 1 adds its part in place (fixed order), then every
    // thread finishes its eight row-contiguous quads: + residual, ReLU, store.  Quad q of pixel row r lives at position
    // q ^ ((r >> 2) & (NQ - 1)) of the row: the 16 lanes of a ds_write_b128 group are 16 different tiles.
    float* const T = lds;
    const int tl = wm0 + frow;                                   // this lane's tile inside the block tile
    auto quad_addr = [&](int g, int o) {
        const int q = (wn0 >> 2) + 2 * g + fhalf;
        return T + ((4 * tl + o) * NQ + (q ^ (tl & (NQ - 1)))) * 4;
    };
    auto part = [&](int g, int o, int e) -> float {              // g, o, e compile-time after unrolling
        const float a0 = acc[0][4 * g + e], a1 = acc[1][4 * g + e], a2 = acc[2][4 * g + e];
        if (pp == 0) return o == 0 ? a0 + (a1 + a2) : (o == 2 ? a1 + a2 : a1 - a2);
        return o == 0 ? a0 + a1 : (o == 1 ? 2.f * (a0 - a1) : (o == 2 ? 4.f * (a0 + a1) : __builtin_fmaf(8.f, a0 - a1, a2)));
    };
    if (pp == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                f32x4 sv;
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[e] = part(g, o, e) + bq[g][e];
                *reinterpret_cast<f32x4*>(quad_addr(g, o)) = sv;
            }
        }
    }
    __syncthreads();
    if (pp == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float* q = quad_addr(g, o);
                f32x4 sv = *reinterpret_cast<const f32x4*>(q);
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[e] += part(g, o, e);
                *reinterpret_cast<f32x4*>(q) = sv;
            }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x4 y = *reinterpret_cast<const f32x4*>(T + (i * 256 + tid) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] += rr[i][e];
            if (p.act == ACT_RELU) y[e] = fmaxf(y[e], 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rs_out, piece_off(i, p.omap.S1), 0, 0);
    }
#ifdef CAPF_DIAG
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long* d = capf_wino_timeline + (size_t)blockIdx.x * 8;
        d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[3] = __builtin_amdgcn_s_memtime();
        d[4] = dbg_r0; d[5] = (unsigned long long)p.Cin; d[6] = (unsigned long long)nsc; d[7] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* ph = capf_wino_phases + (size_t)blockIdx.x * 8;
        for (int k = 0; k < 8; ++k) ph[k] = dbg_ph[k];
    }
#endif
}
#endif

__device__ __forceinline__ int xcd_remap_w(int b, int nblk) {   // see igemm_f32.hip :: xcd_remap
    const int q = nblk >> 3, r = nblk & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// tile configuration of a problem: F(2,3): 0 = 64 tiles x 64 channels, 1 = 64 x 32, 2 = 32 x 64; F(4,3): 3 = 64 x 32, 4 = 32 x 64
#if defined(__HIP_DEVICE_COMPILE__)
template <bool PP>
__device__ __forceinline__ void wino_dispatch(const GemmArgs& p, int cfg, int bid, float* lds) {
    if (cfg == 1) wino_tile_h<64, 32>(p, bid, lds);
    else if (cfg == 2) wino_tile_h<32, 64>(p, bid, lds);
    else if (cfg == 3) wino43_tile<64, 32>(p, bid, lds);
    else if (cfg == 4) wino43_tile<32, 64>(p, bid, lds);
    else wino_tile<PP>(p, bid, lds);
}
#endif

template <bool PP>
__global__ __launch_bounds__(256, 2) void igemm_wino_kernel(GemmArgs p, int cfg) {      // 2 blocks per CU: <= 256 registers per wave
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    wino_dispatch<PP>(p, cfg, xcd_remap_w(blockIdx.x, gridDim.x), wlds);
#endif
}

// grouped launch: up to MAXG Winograd problems in one grid (the 3x3 convs of an HRNet level), longest K first
struct WinoGroupArgs {
    GemmArgs g[MAXG];
    int start[MAXG + 1];
    int tiles[MAXG];
    int cfg[MAXG];
    int n;
};

template <bool PP>
__global__ __launch_bounds__(256, 2) void igemm_wino_group_kernel(WinoGroupArgs ga) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < ga.n && b >= ga.start[pi + 1]) ++pi;
    const int l = b - ga.start[pi];
    const int per_xcd = (ga.start[pi + 1] - ga.start[pi]) >> 3;
    const int bid = (l & 7) * per_xcd + (l >> 3);
    if (bid >= ga.tiles[pi]) return;
    wino_dispatch<PP>(ga.g[pi], ga.cfg[pi], bid, wlds);
#endif
}

// F(4,3) problems only, 16-channel sub-chunks, THREE blocks per CU (<= 168 registers, 36 KiB of LDS): see wino43s_tile
struct Wino43GroupArgs {
    GemmArgs g[MAXG];
    int start[MAXG + 1];
    int tiles[MAXG];
    int cfg[MAXG];                             // 3: 64 tiles x 32 channels, 4: 32 tiles x 64 channels
    int prio[MAXG];                            // s_setprio level of the problem's waves (see launch_wino43_group)
    int n;
};

[[maybe_unused]] static constexpr int W43S_LDS = 6 * (64 + 32) * 16;              // floats per superstage (36 KiB); device code only
template <bool DB>
__device__ __forceinline__ void wino43_group_body(const Wino43GroupArgs& ga, float* wlds) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < ga.n && b >= ga.start[pi + 1]) ++pi;
    const int l = b - ga.start[pi];
    const int per_xcd = (ga.start[pi + 1] - ga.start[pi]) >> 3;
    const int bid = (l & 7) * per_xcd + (l >> 3);
    if (bid >= ga.tiles[pi]) return;
    // issue priority by K length (measured: no effect on the level either way; kept switchable in the diagnosis build)
    const int prio = ga.prio[pi];
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    if (ga.cfg[pi] == 3) wino43s_tile<64, 32, DB>(ga.g[pi], bid, wlds);
    else wino43s_tile<32, 64, DB>(ga.g[pi], bid, wlds);
#endif
}

// (the per-triple transform needs no coefficient registers: 128 registers without the residual prefetch, i.e. FOUR residents
// would fit -- measured: no faster than three (124.7 vs 124.6 us per level, 6122 vs 6177 frames/s): every resident added
// lengthens the others' phases, the level is bound by the shared matrix pipe + LDS-DMA path, not by occupancy)
__global__ __launch_bounds__(256, 3) void igemm_wino43_group_kernel(Wino43GroupArgs ga) {          // ping-pong, three residents
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float wlds[W43S_LDS];
    wino43_group_body<false>(ga, wlds);
#endif
}

__global__ __launch_bounds__(256, 2) void igemm_wino43_group_db_kernel(Wino43GroupArgs ga) {       // double-buffered, two residents
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float wlds[2 * W43S_LDS];
    wino43_group_body<true>(ga, wlds);
#endif
}

// ---- weights: BN fold + G transform, packed [Cout][(kh, chunk, p, c)] (K'' = 12 Cin) ----------------------------------
__global__ void pack_conv_wino_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                      float* __restrict__ Wp, float* __restrict__ bias, int Cout, int Cin, int NP) {
    const int Kw = 3 * NP * Cin, CC = Cin / WBK;          // NP = 4: F(2,3), 6: F(4,3)
    const long total = (long)Cout * Kw;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kw), k = (int)(i - (long)n * Kw);
        const int cl = k % WBK, pq = (k / WBK) % NP, cc = (k / (NP * WBK)) % CC, kh = k / (NP * WBK * CC);
        const int c = cc * WBK + cl;
        const float sc = gamma ? gamma[n] / sqrtf(var[n] + eps) : 1.f;
        const float* g = w + (((long)n * Cin + c) * 3 + kh) * 3;
        const float g0 = g[0] * sc, g1 = g[1] * sc, g2 = g[2] * sc;
        float u;
        if (NP == 4) {
            if (pq == 0) u = g0;
            else if (pq == 1) u = 0.5f * ((g0 + g1) + g2);
            else if (pq == 2) u = 0.5f * ((g0 - g1) + g2);
            else u = g2;
        } else {
            if (pq == 0) u = 0.25f * g0;
            else if (pq == 1) u = -((g0 + g1) + g2) / 6.0f;
            else if (pq == 2) u = -((g0 - g1) + g2) / 6.0f;
            else if (pq == 3) u = (g0 / 24.0f + g1 / 12.0f) + g2 / 6.0f;
            else if (pq == 4) u = (g0 / 24.0f - g1 / 12.0f) + g2 / 6.0f;
            else u = g2;
        }
        Wp[i] = u;
        if (k == 0 && bias) bias[n] = gamma ? beta[n] - mean[n] * sc : 0.f;
    }
}

// variant 23: F(2,3), Wp [Cout][12 Cin]; 43: F(4,3), Wp [Cout][18 Cin]
hipError_t launch_pack_conv_wino(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                                 float eps, float* Wp, float* bias, int Cout, int Cin, hipStream_t s, int variant) {
    if (Cin % WBK != 0 || (variant != 23 && variant != 43)) return hipErrorInvalidValue;
    const int NP = variant == 43 ? 6 : 4;
    const long total = (long)Cout * 3 * NP * Cin;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_conv_wino_kernel, dim3(blocks), dim3(256), 0, s, w, gamma, beta, mean, var, eps, Wp, bias, Cout, Cin, NP);
    return hipGetLastError();
}

// ---- host side --------------------------------------------------------------------------------------------------------
// the variant rides in the packed-weight pitch: Kpad == 18 Cin -> F(4,3) (needs W % 4 == 0), otherwise F(2,3) (even W)
static bool is43(const GemmArgs& a) { return a.Kpad == 18 * a.Cin; }

bool gemm_wino_ok(const GemmArgs& a) {
    return a.conv && a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Cin % WBK == 0 && (a.W % (is43(a) ? 4 : 2)) == 0 && (a.N & 3) == 0 &&
           a.omap.G == 1 && (!a.res || a.rmap.G == 1) && !a.rscale && a.act != ACT_GELU && !a.out_bf16 && a.H == a.Ho && a.W == a.Wo &&
           (a.Kpad == 12 * a.Cin || a.Kpad == 18 * a.Cin);
}

// rewrite a direct-conv problem description into the tile-grid form the tile functions expect; false if out of range
static bool wino_prepare(GemmArgs& a) {
    if (!a.Wp || !gemm_wino_ok(a)) return false;        // (no Winograd pack: the engine skips layouts its plan cannot reach)
    const int px = is43(a) ? 4 : 2;
    const long tiles = (long)(a.M / (a.Ho * a.Wo)) * a.H * (a.W / px);
    if ((double)a.M * (double)a.omap.S1 >= 4.0e9 || tiles > 0x7fffffffL) return false;
    a.Wo = a.W / px;                        // tile grid: H x W/px
    a.M = (int)tiles;
    a.fd_hw = make_fastdiv((unsigned)(a.Ho * a.Wo));
    a.fd_wo = make_fastdiv((unsigned)a.Wo);
    return true;
}

static hipError_t wino_attr() {          // (per device: DynLdsAttr)
    static DynLdsAttr big[2], half[2];
    const int big_bytes = WLDS * (int)sizeof(float), half_bytes = 6 * (64 + 32) * WBK * (int)sizeof(float);
    hipError_t r = big[0].ensure(reinterpret_cast<const void*>(igemm_wino_kernel<false>), big_bytes);
    if (r == hipSuccess) r = big[1].ensure(reinterpret_cast<const void*>(igemm_wino_group_kernel<false>), big_bytes);
    if (r == hipSuccess) r = half[0].ensure(reinterpret_cast<const void*>(igemm_wino_kernel<true>), half_bytes);
    if (r == hipSuccess) r = half[1].ensure(reinterpret_cast<const void*>(igemm_wino_group_kernel<true>), half_bytes);
    return r;
}

// CAPF_WINO_MODE (A/B runs only): 0 (default) ping-pong, 64 KiB, two blocks per CU; 1: double-buffered 64 x 64 tile, 128 KiB
static int wino_mode() {
    static const int m = [] { const char* e = diag_env("CAPF_WINO_MODE"); return e ? atoi(e) : 0; }();
    return m;
}

static const int kWT[5] = {64, 64, 32, 64, 32}, kWN[5] = {64, 32, 64, 32, 64};
static const int kWLDS[5] = {WLDS / 2, 4 * (64 + 32) * WBK, 4 * (64 + 32) * WBK, 6 * (64 + 32) * WBK, 6 * (64 + 32) * WBK};   // floats

// tile configuration for a prepared problem (a.M = tiles).  F(2,3): narrow outputs -> 64 x 32; few tiles -> 32 x 64 (twice the
// blocks).  F(4,3): 32 tiles x 64 channels, or 64 x 32 for outputs that are not a multiple of 64 wide.
static int wino_cfg(const GemmArgs& a) {
    static const int forced = [] { const char* e = diag_env("CAPF_WINO_CFG"); return e ? atoi(e) : -1; }();   // tuning only
    if (is43(a)) return a.N % 64 != 0 ? 3 : 4;
    if (forced == 1 || (forced == 2 && a.N % 64 == 0)) return forced;
    if (a.N % 64 != 0) return 1;
    const long blocks = (long)((a.M + 63) / 64) * (a.N / 64);
    return blocks < 512 ? 2 : 0;
}

static int wino_tiles(const GemmArgs& a, int cfg) { return ((a.M + kWT[cfg] - 1) / kWT[cfg]) * ((a.N + kWN[cfg] - 1) / kWN[cfg]); }

// F(4,3) problems take the three-resident kernel (16-channel sub-chunks need Cin % 32 == 0 for the shared weight layout: always)
static int wino43_short() {
    // 0: the 72 KiB ping-pong tiles of igemm_wino_group_kernel; 1: 16-channel superchunks, ping-pong, three residents;
    // 2: 16-channel superchunks, double-buffered, two residents
    static const int v = [] { const char* e = diag_env("CAPF_WINO43_SHORT"); return e ? atoi(e) : 1; }();     // A/B runs only
    return v;
}

static int wino43_prio() {
    static const int v = [] { const char* e = diag_env("CAPF_WINO43_PRIO"); return e ? atoi(e) : 0; }();       // A/B runs only
    return v;
}

static hipError_t launch_wino43_group(const GemmArgs* prep, const int* cfgs, int n, hipStream_t s) {
    struct Item { int idx, tiles; double cost; };
    Item it[MAXG];
    for (int i = 0; i < n; ++i)
        it[i] = Item{i, wino_tiles(prep[i], cfgs[i]), (double)prep[i].Cin};          // both tile shapes do the same work per block
    for (int i = 1; i < n; ++i)                  // longest tile first
        for (int j = i; j > 0 && it[j].cost > it[j - 1].cost; --j) { Item t = it[j]; it[j] = it[j - 1]; it[j - 1] = t; }
    if (const char* ord = diag_env("CAPF_WINO43_ORDER")) {       // A/B runs only: dispatch order as a permutation of the cost ranks, e.g. "0312"
        Item tmp[MAXG];
        int m = 0;
        for (const char* c = ord; *c && m < n; ++c)
            if (*c >= '0' && *c - '0' < n) tmp[m++] = it[*c - '0'];
        if (m == n) for (int i = 0; i < n; ++i) it[i] = tmp[i];
    }
    Wino43GroupArgs ga;
    ga.n = n;
    int start = 0;
    for (int i = 0; i < n; ++i) {
        ga.g[i] = prep[it[i].idx];
        ga.cfg[i] = cfgs[it[i].idx];
        ga.tiles[i] = it[i].tiles;
        ga.start[i] = start;
        start += (it[i].tiles + 7) & ~7;
        // longest K -> priority 3, next distinct K length 2, ...; problems of the shortest K (and lone problems) stay at 0
        int longer = 0;
        for (int j = 0; j < n; ++j)
            if (it[j].cost > it[i].cost && (j == 0 || it[j].cost != it[j - 1].cost)) ++longer;
        int shorter = 0;
        for (int j = 0; j < n; ++j) shorter += it[j].cost < it[i].cost;
        ga.prio[i] = (shorter && wino43_prio()) ? std::max(1, 3 - longer) : 0;
    }
    ga.start[n] = start;
    for (int i = n; i < MAXG; ++i) { ga.start[i + 1] = start; ga.tiles[i] = 0; ga.cfg[i] = 4; ga.prio[i] = 0; }
    if (wino43_short() == 2) hipLaunchKernelGGL(igemm_wino43_group_db_kernel, dim3(start), dim3(256), 0, s, ga);
    else hipLaunchKernelGGL(igemm_wino43_group_kernel, dim3(start), dim3(256), 0, s, ga);
    return hipGetLastError();
}

hipError_t launch_gemm_wino(const GemmArgs& a_in, hipStream_t s) {
    if (gemm_f32x3_wanted(a_in)) return launch_gemm_f32x3(a_in, s);       // (large launches: the split-fp32 tile, igemm_f32x3_ws.hip)
    GemmArgs a = a_in;
    if (!wino_prepare(a)) return hipErrorInvalidValue;
    hipError_t r = wino_attr();
    if (r != hipSuccess) return r;
    const int cfg = wino_cfg(a);
    if (cfg >= 3 && wino43_short()) return launch_wino43_group(&a, &cfg, 1, s);
    const int nb = wino_tiles(a, cfg);
    if (wino_mode() == 1) hipLaunchKernelGGL(igemm_wino_kernel<false>, dim3(nb), dim3(256), WLDS * sizeof(float), s, a, cfg);
    else hipLaunchKernelGGL(igemm_wino_kernel<true>, dim3(nb), dim3(256), kWLDS[cfg] * sizeof(float), s, a, cfg);
    return hipGetLastError();
}

// what rocprofv3 will call the launch: F(4,3) problems (alone or grouped) run igemm_wino43_group_kernel
const char* gemm_wino_kernel_name(const GemmArgs& a) {
    if (gemm_f32x3_wanted(a)) return gemm_f32x3_kernel_name(a);
    return (a.Kpad == 18 * a.Cin && wino43_short()) ? "igemm_wino43_group" : "igemm_wino<w4,F(2,3)>";
}

hipError_t launch_gemm_wino_group(const GemmArgs* list, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n == 1) return launch_gemm_wino(list[0], s);
    if (n > MAXG) return hipErrorInvalidValue;
    {   // the problems the split-fp32 tile wants (a per-problem rule: igemm_f32x3_ws.hip) leave in one launch of their own
        GemmArgs x3[MAXG], rest[MAXG];
        int nx = 0, nrest = 0;
        for (int i = 0; i < n; ++i) {
            if (gemm_f32x3_wanted(list[i])) x3[nx++] = list[i];
            else rest[nrest++] = list[i];
        }
        if (nx) {
            const hipError_t rx = launch_gemm_f32x3_group(x3, nx, s);
            if (rx != hipSuccess || nrest == 0) return rx;
            return launch_gemm_wino_group(rest, nrest, s);
        }
    }
    hipError_t r = wino_attr();
    if (r != hipSuccess) return r;
    struct Item { int idx, cfg, tiles; double cost; };
    Item it[MAXG];
    GemmArgs prep[MAXG];
    int lds_floats = 0;
    if (wino43_short()) {                        // F(4,3) problems -> their own grid (three resident blocks per CU); the rest below
        GemmArgs p43[MAXG], rest[MAXG];
        int c43[MAXG], n43 = 0, nrest = 0;
        for (int i = 0; i < n; ++i) {
            GemmArgs a = list[i];
            if (!wino_prepare(a)) return hipErrorInvalidValue;
            const int cfg = wino_cfg(a);
            if (cfg >= 3) { p43[n43] = a; c43[n43++] = cfg; }
            else rest[nrest++] = list[i];
        }
        if (n43) {
            r = launch_wino43_group(p43, c43, n43, s);
            if (r != hipSuccess || nrest == 0) return r;
            if (nrest == 1) return launch_gemm_wino(rest[0], s);
            list = rest;                         // (local copies live until the launch below returns)
            n = nrest;
            return launch_gemm_wino_group(list, n, s);
        }
    }
    for (int i = 0; i < n; ++i) {
        prep[i] = list[i];
        if (!wino_prepare(prep[i])) return hipErrorInvalidValue;
        const int cfg = wino_cfg(prep[i]);
        lds_floats = std::max(lds_floats, kWLDS[cfg]);
        it[i] = Item{i, cfg, wino_tiles(prep[i], cfg), (double)prep[i].Cin * kWT[cfg] * kWN[cfg] * (cfg >= 3 ? 1.5 : 1.0)};
    }
    for (int i = 1; i < n; ++i)                  // longest tile first
        for (int j = i; j > 0 && it[j].cost > it[j - 1].cost; --j) { Item t = it[j]; it[j] = it[j - 1]; it[j - 1] = t; }
    WinoGroupArgs ga;
    ga.n = n;
    int start = 0;
    for (int i = 0; i < n; ++i) {
        ga.g[i] = prep[it[i].idx];
        ga.cfg[i] = it[i].cfg;
        ga.tiles[i] = it[i].tiles;
        ga.start[i] = start;
        start += (it[i].tiles + 7) & ~7;
    }
    ga.start[n] = start;
    for (int i = n; i < MAXG; ++i) { ga.start[i + 1] = start; ga.tiles[i] = 0; ga.cfg[i] = 0; }
    if (wino_mode() == 1) hipLaunchKernelGGL(igemm_wino_group_kernel<false>, dim3(start), dim3(256), WLDS * sizeof(float), s, ga);
    else hipLaunchKernelGGL(igemm_wino_group_kernel<true>, dim3(start), dim3(256), lds_floats * sizeof(float), s, ga);
    return hipGetLastError();
}

}  // namespace capf

#ifdef CAPF_DIAG
extern "C" __attribute__((visibility("default"))) int capf_debug_wino_timeline(unsigned long long* dst, int blocks) {
    if (blocks > 8192) blocks = 8192;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(capf::capf_wino_timeline), (size_t)blocks * 64);
}
extern "C" __attribute__((visibility("default"))) int capf_debug_wino_phases(unsigned long long* dst, int blocks) {
    if (blocks > 8192) blocks = 8192;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(capf::capf_wino_phases), (size_t)blocks * 64);
}
#endif
