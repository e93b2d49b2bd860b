// bf16 implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x16_bf16 (BASELINE.json configs[2] / [4]:
// "bf16 inference ... MFMA bf16 path").  bf16 operands, fp32 accumulate, bf16 NHWC activations in HBM.
//
//   out[m, n] = act( sum_k A[m, k] * Wp[n, k] + bias[n] + res[m, n] ),  m = (b, ho, wo), k = (kh, kw, ci)
//
// Same machinery as igemm_f32.hip, re-dimensioned for 2-byte elements:
//   * a tile row in LDS is still 128 B = 64 bf16 (BK = 64), staged by global_load_lds_dwordx4 with the same
//     source-side XOR swizzle; ONE ds_read_b128 (8 bf16) is exactly one MFMA operand: lanes 0-31 carry
//     k = 16*step + 0..7, lanes 32-63 carry k = 16*step + 8..15.
//   * a 64-deep chunk is only 4 MFMAs (32 cycles each) per 32x32 tile, so this kernel lives on the load
//     path, not on the matrix pipe (the bf16 peak is 16x the fp32 one): what matters is bytes, and every
//     activation byte is half of what the fp32 path moves.
//   * the accumulator is kept transposed (weights as the MFMA A operand) so each lane owns 4 consecutive
//     output channels; the bf16 epilogue transposes 32x32 blocks through the idle stage so that residual loads (prefetched
//     with the last chunk) and stores are 16 B per lane, 64 B contiguous per row.
//   * bf16 rounding is round-to-nearest-even, done once, in the epilogue (v_cvt_pk_bf16_f32).
// Kernels in this file (DESIGN.md 4.1b has the measurements behind each):
//   igemm_bf16_tile      the general tile; S >= 2: ring of S stages (launches below 2048 tiles, the lifter's projections),
//                        S == 1: ping-pong schedule (one stage, load phase / compute phase, 5 blocks per CU)
//   igemm_bf16_rh_tile   3x3 / stride-1 "row-halo" tile: one staged activation tile for the three kw taps
//   igemm_bf16_group_kernel / _pp_kernel / _rh_kernel   up to 8 independent convs in one grid (ring / ping-pong / ping-pong
//                        with row-halo tiles); igemm_bf16_kernel / igemm_bf16_rh_kernel: one conv
//   igemm_bf16_stem_kernel (+ igemm_bf16_smallc_kernel)   Cin = 3 stem under compute_dtype = bf16
#include <stdlib.h>

#include "kernels.h"

namespace capf {
#ifdef CAPF_DIAG   // (diagnosis build) per-block stamps {t_entry, t_prologue_done, t_loop_done, t_stores_issued, t_exit, realtime_entry, load_wait_ticks, realtime_exit}
__device__ unsigned long long capf_bf16_timeline[8192 * 8];
#define B16_STAMP(var) var = __builtin_amdgcn_s_memtime()
#else
#define B16_STAMP(var)
#endif


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

static constexpr int BKH = 64;     // K chunk in bf16 elements = 128 B per tile row

__device__ __forceinline__ int fast_div_b(int n, FastDiv d) {
    return (int)((__umulhi((unsigned)n, d.mul) + (unsigned)n) >> d.shift);
}
__device__ __forceinline__ unsigned short f2bf(float f) { return to_bf16(f); }
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt_b() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 4 waves per block, block tile BM x BN, wave tile WM x WN, S LDS stages.
// OUTF32 / GELU select the epilogue of the lifter's projections (pose_dformer.py:15-59, run as 1x1 "convolutions" over an
// [M, 1] image of K channels): bf16 operands and fp32 accumulation like the convs, but the residual stream stays fp32 —
//   OUTF32: out / res are fp32, addressed through the (G, S1, S2) row maps (qkv, proj + residual, fc2 + residual);
//   GELU:   exact-erf GELU before the bf16 store (fc1, whose output is only ever the bf16 operand of fc2).
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type and builtins exist on the device side only
// one output tile (logical id `bid`) with the calling block; body of the single and the grouped kernel
__device__ __forceinline__ long rowmap_b(const RowMap& r, int m) {
    if (r.G == 1) return (long)m * r.S1 + r.off;
    const int q = m / r.G;
    return (long)q * r.S1 + (long)(m - q * r.G) * r.S2 + r.off;
}

template <int BM, int BN, int WM, int WN, int S, bool OUTF32 = false, bool GELU = false, bool UPADD = false>
__device__ __forceinline__ void igemm_bf16_tile(const GemmArgs& p, const int bid, unsigned short* __restrict__ lds) {
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int NLOAD = RA + RB;
    constexpr int STAGE = (BM + BN) * BKH;                // bf16 elements per stage
    static_assert((BM / WM) * (BN / WN) == 4, "wave grid");

    const unsigned short* A = reinterpret_cast<const unsigned short*>(p.A);
    const unsigned short* Wp = reinterpret_cast<const unsigned short*>(p.Wp);
    const unsigned short* Rs = reinterpret_cast<const unsigned short*>(p.res);
    unsigned short* Out = reinterpret_cast<unsigned short*>(p.out);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CAPF_DIAG
    unsigned long long dbg_t0 = 0, dbg_t1 = 0, dbg_t2 = 0, dbg_t3 = 0, dbg_w = 0, dbg_a = 0, dbg_b = 0;
    const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime();
    auto dbg_flush = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0 && blockIdx.x < 8192) {
            unsigned long long* d = capf_bf16_timeline + (size_t)blockIdx.x * 8;
            d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[3] = dbg_t3; d[4] = __builtin_amdgcn_s_memtime();
            d[5] = dbg_r0; d[6] = dbg_w; d[7] = __builtin_amdgcn_s_memrealtime();
        }
    };
#endif
    B16_STAMP(dbg_t0);

    const int nbn = (p.N + BN - 1) / BN;
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int srow = tid >> 3;
    const int kq = (((tid & 7) ^ ((srow >> 1) & 7))) * 8;      // logical k offset (bf16 elements) of this lane's quad

    // Operand fetch = `buffer_load_dwordx4 ... lds` with block-uniform descriptors and 32-bit per-lane byte
    // offsets; out-of-range offsets make the hardware write zeros (see igemm_f32.hip for the scheme).
    const int nchunks = p.Kpad / BKH;
    constexpr unsigned OOB_A = 0x80000000u;
    long a_base;
    {
        const int b = fast_div_b(m0, p.fd_hw), rem = m0 - b * p.Ho * p.Wo;
        const int ho = fast_div_b(rem, p.fd_wo), wo = rem - ho * p.Wo;
        a_base = ((long)b * p.H * p.W + (long)(ho * p.stride - p.pad) * p.W + (wo * p.stride - p.pad)) * p.Cin;
    }
    const rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + a_base), 0, 0x7FFFFF00u, 0x00020000);
    const rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(Wp + (long)n0 * p.Kpad), 0,
                                                            (unsigned)(p.N - n0) * (unsigned)p.Kpad * 2u, 0x00020000);
    unsigned a_rel[RA];                // byte offset of (row, tap 0, channel kq) from the descriptor base
    unsigned a_mask[RA];               // bit t set <=> tap t of this row reads real data
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + srow + 32 * i;
        a_rel[i] = 0;
        a_mask[i] = 0u;
        if (m < p.M) {
            const int b = fast_div_b(m, p.fd_hw), rem = m - b * p.Ho * p.Wo;
            const int ho = fast_div_b(rem, p.fd_wo), wo = rem - ho * p.Wo;
            const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
            const long off = ((long)b * p.H * p.W + (long)h0 * p.W + w0) * p.Cin;
            a_rel[i] = (unsigned)(off - a_base + kq) * 2u;
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
    unsigned w_off[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) w_off[i] = (unsigned)((srow + 32 * i) * p.Kpad + kq) * 2u;

    // Cin % 64 == 0: a chunk lies inside one tap -> block-uniform walk, tap offset in the scalar offset
    const bool uni = (p.Cin & (BKH - 1)) == 0;
    int u_tap = 0, u_ci = 0, u_kh = 0, u_kw = 0;
    int tap = 0, ci = kq, kh = 0, kw = 0;
    if (!uni) {
        tap = kq / p.Cin;
        ci = kq - tap * p.Cin;
        kh = tap / p.ks;
        kw = tap - kh * p.ks;
    }

    unsigned voff[NLOAD];
    unsigned soff_a = 0;
    auto prepare = [&](int c) {
        if (uni) {
            const unsigned bit = u_tap < 32 ? (1u << u_tap) : 0u;
            soff_a = __builtin_amdgcn_readfirstlane((unsigned)((u_kh * p.W + u_kw) * p.Cin + u_ci) * 2u);
#pragma unroll
            for (int i = 0; i < RA; ++i) voff[i] = (a_mask[i] & bit) ? a_rel[i] : OOB_A;
            u_ci += BKH;
            if (u_ci >= p.Cin) {
                u_ci = 0;
                ++u_tap;
                if (++u_kw == p.ks) { u_kw = 0; ++u_kh; }
            }
        } else {
            const unsigned bit = tap < 32 ? (1u << tap) : 0u;
            const unsigned t = (unsigned)((kh * p.W + kw) * p.Cin + ci - kq) * 2u;
#pragma unroll
            for (int i = 0; i < RA; ++i) voff[i] = (a_mask[i] & bit) ? a_rel[i] + t : OOB_A;
            ci += BKH;
            while (ci >= p.Cin) {
                ci -= p.Cin;
                ++tap;
                if (++kw == p.ks) { kw = 0; ++kh; }
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            voff[RA + i] = w_off[i];
            w_off[i] += BKH * 2u;
        }
    };
    auto fire = [&](int idx, int stage) {
        unsigned short* As = lds + stage * STAGE;
        if (idx < RA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(As + (idx * 32 + wave * 8) * BKH), 16, voff[idx],
                                                     soff_a, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc_w, (lptr_t)(As + BM * BKH + ((idx - RA) * 32 + wave * 8) * BKH), 16, voff[idx], 0, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    const int fhalf = lane >> 5;

    // Epilogue operands (residual rows, bias) in the coalesced epilogue's layout -- lane = 8 channels of one row, see below --
    // are requested when the LAST chunk is: their latency (2-3 us under load, which used to be a quarter of a block's life at
    // K = 9*48) hides behind that chunk's.
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    const bool vec_ok = !OUTF32 && (p.N & 7) == 0 && (p.omap.S1 & 7) == 0 && (p.omap.off & 7) == 0 &&
                        (!Rs || ((p.rmap.S1 & 7) == 0 && (p.rmap.off & 7) == 0));
    const int er = lane >> 2, ec = (lane & 3) * 8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rr[TM][TN][2];
    f32x4 bb[TN][2];
    // Every access of the vector epilogue is a raw buffer load / store on a block-local descriptor; a piece outside a ragged
    // M / N edge gets an out-of-range offset (reads zeros, store dropped) instead of a branch.  With divergent branches around
    // them the compiler cannot count vmcnt: it waited `vmcnt(0)` in front of every residual use, i.e. for the PREVIOUS STORE
    // to complete -- the 2 TM TN stores of a block were serialised round trips.  A null residual / bias is a descriptor of
    // zero records.
    constexpr unsigned OOB_E = 0x80000000u;
    const rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        (Rs && vec_ok) ? (void*)(Rs + (long)m0 * p.rmap.S1 + p.rmap.off + n0) : (void*)Out, 0, (Rs && vec_ok) ? 0x7FFFFF00u : 0u,
        0x00020000);
    const rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(Out + (long)m0 * p.omap.S1 + p.omap.off + n0), 0,
                                                            vec_ok ? 0x7FFFFF00u : 0u, 0x00020000);
    const rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)(p.bias + n0) : (void*)Out, 0,
                                                             p.bias ? (unsigned)(p.N - n0) * 4u : 0u, 0x00020000);
    auto piece_off = [&](int i, int j, int h, int S1) -> unsigned {   // byte offset of this lane's 8 channels inside the block's tile
        const int ml = wm0 + i * 32 + h * 16 + er, nl = wn0 + j * 32 + ec;
        return (m0 + ml < p.M && n0 + nl < p.N) ? (unsigned)(ml * S1 + nl) * 2u : OOB_E;
    };
    auto prefetch_epilogue = [&]() {
        if (!vec_ok) return;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned nb = (unsigned)(wn0 + j * 32 + ec) * 4u;
            bb[j][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, nb, 0, 0));
            bb[j][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, nb + 16u, 0, 0));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    rr[i][j][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, piece_off(i, j, h, (int)p.rmap.S1), 0, 0);
        }
    };

    constexpr int PER_STEP = (NLOAD + 1) / 2;
    if constexpr (S == 1) {
        // "ping-pong": ONE stage per block (24-32 KiB), so that 5-6 blocks are resident per CU.  A block alternates a load
        // phase (all DMA instructions of a chunk, wait, barrier) and a compute phase (the chunk's MFMAs); the other resident
        // blocks use the matrix pipe meanwhile.  At bf16 MFMA speed a 64-deep chunk is only 256 matrix cycles per wave
        // against ~2500 cycles of load latency under traffic: what matters is how many chunks a CU has in flight (5 of 6
        // slots here; 3 of 6 with the two-stage ring and three blocks).
        bf16x8 af[2][TM], bfr[2][TN];
        auto read_frags = [&](int step, int buf) {
            const unsigned short* As = lds;
            const unsigned short* Bs = As + BM * BKH;
            const int q = ((step * 2) + fhalf) ^ fsw;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[buf][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&As[(wm0 + i * 32 + frow) * BKH + q * 8]));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[buf][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&Bs[(wn0 + j * 32 + frow) * BKH + q * 8]));
        };
        B16_STAMP(dbg_t1);
        auto chunk = [&](int c) {
            B16_STAMP(dbg_a);
            prepare(c);
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) fire(i, 0);
            wait_vmcnt_b<0>();
            __builtin_amdgcn_s_barrier();
#ifdef CAPF_DIAG
            B16_STAMP(dbg_b);
            dbg_w += dbg_b - dbg_a;
#endif
            read_frags(0, 0);
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                if (step < 3) read_frags(step + 1, (step + 1) & 1);
                const int fb = step & 1;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[fb][j], af[fb][i], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // every wave has read the stage: the next chunk may overwrite it
        };
        for (int c = 0; c < nchunks - 1; ++c) chunk(c);
        prefetch_epilogue();                           // (the last chunk is peeled so that these registers are not live in the loop)
        if (nchunks > 0) chunk(nchunks - 1);
    } else {
#pragma unroll
    for (int s = 0; s < S - 1; ++s) {
        prepare(s);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) fire(i, s);
    }
    int st_read = 0, st_fill = S - 1;
    prepare(S - 1);
    // fragments are double buffered: the reads of k-step s+1 (or, at the chunk boundary, of the next chunk's
    // step 0) are issued before the MFMAs of step s
    bf16x8 af[2][TM], bfr[2][TN];
    auto read_frags = [&](int stage, int step, int buf) {
        const unsigned short* As = lds + stage * STAGE;
        const unsigned short* Bs = As + BM * BKH;
        const int q = ((step * 2) + fhalf) ^ fsw;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[buf][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&As[(wm0 + i * 32 + frow) * BKH + q * 8]));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bfr[buf][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&Bs[(wn0 + j * 32 + frow) * BKH + q * 8]));
    };
    wait_vmcnt_b<(S - 2) * NLOAD>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0);
    auto ring_iter = [&](int c) {
        const int st_next = (st_read + 1 == S) ? 0 : st_read + 1;
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            if (step == 2) prepare(c + S);
            if (step < 3) {
                read_frags(st_read, step + 1, (step + 1) & 1);
            } else {
                wait_vmcnt_b<(S - 2) * NLOAD>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                read_frags(st_next, 0, 0);
            }
            const int fb = step & 1;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[fb][j], af[fb][i], acc[i][j], 0, 0, 0);
            if (step < 2) {
#pragma unroll
                for (int f = 0; f < PER_STEP; ++f) {
                    const int idx = step * PER_STEP + f;
                    if (idx < NLOAD) fire(idx, st_fill);
                }
            }
        }
        st_read = st_next;
        st_fill = (st_fill + 1 == S) ? 0 : st_fill + 1;
    };
    for (int c = 0; c < nchunks - 1; ++c) ring_iter(c);
    prefetch_epilogue();
    if (nchunks > 0) ring_iter(nchunks - 1);
    }
    wait_vmcnt_b<0>();
    B16_STAMP(dbg_t2);

    // ---- epilogue.  Accumulator layout: lane = one row m (lane & 31), register group g = 4 consecutive channels
    // (8 g + 4 (lane >> 5)).
    if constexpr (OUTF32) {
        // fp32 result + fp32 residual through the strided token maps (the lifter's proj / fc2): 16-byte stores per lane.
        long o_row[TM];
        bool m_ok[TM];
        f32x4 bv[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                bv[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias && (full || n < p.N)) bv[j][g] = *reinterpret_cast<const f32x4*>(p.bias + n);
            }
        f32x4 rf[TM][TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + i * 32 + (lane & 31);
            m_ok[i] = full || m < p.M;
            o_row[i] = m_ok[i] ? rowmap_b(p.omap, m) : 0;
            const long r_row = (p.res && m_ok[i]) ? rowmap_b(p.rmap, m) : 0;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                    rf[i][j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p.res && m_ok[i] && (full || n < p.N)) rf[i][j][g] = *reinterpret_cast<const f32x4*>(p.res + r_row + n);
                }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][4 * g + e] + bv[j][g][e]) + rf[i][j][g][e];
                    if (m_ok[i] && (full || n < p.N)) *reinterpret_cast<f32x4*>(p.out + o_row[i] + n) = v;
                }
    } else {
        // bf16 result, coalesced: in the accumulator layout a lane owns 4 channels of a row, i.e. 8-byte accesses at a row
        // stride (16 B contiguous per row and instruction).  With K = 9*48 .. 9*384 a tile's K loop is only 7-54 chunks of
        // 256 matrix cycles, and that epilogue -- with the residual's load latency in it -- was a quarter of a block's life
        // (tools/bf16_timeline.py).  Instead each wave transposes its 32x32 fp32 blocks through its own 4.5 KiB of the (now
        // idle) stage memory and finishes 8 channels of a row per lane: residual loads (prefetched above) and stores are 16 B
        // per lane, 64 B contiguous per row, 16 rows per instruction.  Shapes the vector path cannot take (N % 8, unaligned
        // strides) finish the same 8 channels element by element.
        constexpr int EPS = 36;                            // padded row stride (floats): b128 accesses stay conflict-free
        if (S != 1) __syncthreads();                       // ring schedule: other waves may still be reading the last stage
        float* ep = reinterpret_cast<float*>(lds) + wave * (32 * EPS);
        auto finish = [&](float t) {
            if (GELU) t = 0.5f * t * (1.0f + erff(t * 0.70710678118654752440f));
            if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
            return t;
        };
        auto transpose_block = [&](int i, int j) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(&ep[(lane & 31) * EPS + 8 * g + 4 * (lane >> 5)]) =
                    f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            __builtin_amdgcn_wave_barrier();
        };
        if (vec_ok) {
            // UPADD: + bilinear_upsample(p.up) AFTER the activation (CPN globalNet.py:66, feature = lateral + upsampled path): the four
            // corners of the row's pixel in the low-resolution map, 8 channels (16 B) per corner and 32-column block, requested for both
            // rows of a 32-row block before its transposes; ATen's align_corners = True arithmetic, the expression of bilinear_kernel
            // (elementwise.hip), so that the sum equals what the resize-add launch it replaces computed from the same operands -- except
            // that the lateral term is no longer rounded to bf16 before the add
            [[maybe_unused]] rsrc_t rs_up;
            if constexpr (UPADD) rs_up = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.up), 0, 0x7FFFFF00u, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                [[maybe_unused]] u32x4 uq[2][TN][4];
                [[maybe_unused]] float ulh[2], ulw[2];
                if constexpr (UPADD) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int m = m0 + wm0 + i * 32 + h * 16 + er;
                        const bool ok = m < p.M;
                        const int b = fast_div_b(ok ? m : 0, p.fd_hw), rem = (ok ? m : 0) - b * p.Ho * p.Wo;
                        const int ho = fast_div_b(rem, p.fd_wo), wo = rem - ho * p.Wo;
                        const float fh = p.up_sh * ho, fw = p.up_sw * wo;
                        const int h0 = (int)fh, w0 = (int)fw;
                        const int h1 = h0 + (h0 < p.up_H - 1), w1 = w0 + (w0 < p.up_W - 1);
                        ulh[h] = fh - h0; ulw[h] = fw - w0;
                        const int rb = b * p.up_H;
                        const unsigned o00 = (unsigned)(((rb + h0) * p.up_W + w0) * p.N) * 2u, o01 = (unsigned)(((rb + h0) * p.up_W + w1) * p.N) * 2u;
                        const unsigned o10 = (unsigned)(((rb + h1) * p.up_W + w0) * p.N) * 2u, o11 = (unsigned)(((rb + h1) * p.up_W + w1) * p.N) * 2u;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int n = n0 + wn0 + j * 32 + ec;
                            const unsigned cb = (ok && n < p.N) ? (unsigned)n * 2u : OOB_E;
                            uq[h][j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_up, cb + o00, 0, 0);
                            uq[h][j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_up, cb + o01, 0, 0);
                            uq[h][j][2] = __builtin_amdgcn_raw_buffer_load_b128(rs_up, cb + o10, 0, 0);
                            uq[h][j][3] = __builtin_amdgcn_raw_buffer_load_b128(rs_up, cb + o11, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    transpose_block(i, j);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = h * 16 + er;
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec]);
                        const f32x4 x1 = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec + 4]);
                        u32x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned rw = rr[i][j][h][q];
                            const float xa = q < 2 ? x0[2 * q] : x1[2 * q - 4], xb = q < 2 ? x0[2 * q + 1] : x1[2 * q - 3];
                            const float ba = q < 2 ? bb[j][0][2 * q] : bb[j][1][2 * q - 4];
                            const float bc = q < 2 ? bb[j][0][2 * q + 1] : bb[j][1][2 * q - 3];
                            float va = finish(xa + ba + __uint_as_float(rw << 16)), vb = finish(xb + bc + __uint_as_float(rw & 0xFFFF0000u));
                            if constexpr (UPADD) {
                                const float lh1 = ulh[h], lw1 = ulw[h], lh0 = 1.f - lh1, lw0 = 1.f - lw1;
                                const unsigned q00 = uq[h][j][0][q], q01 = uq[h][j][1][q], q10 = uq[h][j][2][q], q11 = uq[h][j][3][q];
                                va += lh0 * (lw0 * __uint_as_float(q00 << 16) + lw1 * __uint_as_float(q01 << 16)) +
                                      lh1 * (lw0 * __uint_as_float(q10 << 16) + lw1 * __uint_as_float(q11 << 16));
                                vb += lh0 * (lw0 * __uint_as_float(q00 & 0xFFFF0000u) + lw1 * __uint_as_float(q01 & 0xFFFF0000u)) +
                                      lh1 * (lw0 * __uint_as_float(q10 & 0xFFFF0000u) + lw1 * __uint_as_float(q11 & 0xFFFF0000u));
                            }
                            o[q] = pack_bf16x2(va, vb);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, piece_off(i, j, h, (int)p.omap.S1), 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn0 + j * 32 + ec;
                    transpose_block(i, j);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = h * 16 + er, m = m0 + wm0 + i * 32 + row;
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec]);
                        const f32x4 x1 = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec + 4]);
                        if (m >= p.M) continue;
                        for (int e = 0; e < 8 && n + e < p.N; ++e) {
                            const float x = e < 4 ? x0[e & 3] : x1[e & 3];
                            const float bsv = p.bias ? p.bias[n + e] : 0.f;
                            const float rsv = Rs ? bf2f(Rs[(long)m * p.rmap.S1 + p.rmap.off + n + e]) : 0.f;
                            Out[(long)m * p.omap.S1 + p.omap.off + n + e] = f2bf(finish(x + bsv + rsv));
                        }
                    }
                }
        }
    }
#ifdef CAPF_DIAG
    B16_STAMP(dbg_t3);
    dbg_flush();
#endif
}
#endif

// =====================================================================================================
// "Row-halo" tile for 3x3 / stride 1 / pad 1 convs (the BasicBlock convs: >90 % of the bf16 FLOPs of HRNet).
// The implicit GEMM above stages a fresh [BM x 64] A chunk for every tap, so an activation travels L2 -> LDS nine times;
// at bf16 MFMA speed that operand path, not the matrix pipe, is the bound.  Here the K order is
// (kh, channel chunk, kw, c) and ONE staged A tile serves the three kw taps of a (kh, channel chunk): the tile is 128
// consecutive flat pixels m0-1 .. m0+126 of the row shifted by kh-1, output pixel m0+i reads staged rows i, i+1, i+2, and
// the two taps that would wrap around an image row (w = 0 with kw = 0, w = W-1 with kw = 2) are zeroed per lane in the
// fragment registers.  126 outputs per tile (1.6 % idle MFMA rows) keep the staged tile at exactly 128 rows = whole DMA
// rounds.  Per 192-deep "superchunk" a block stages 128 x CW + 3 x 64 x CW halves instead of 3 x (128 + 64) x 64: 44 % fewer
// bytes and DMA instructions, a third of the load phases and barriers, and no K padding (9 * 48 = 432 is used as is).
// CW = channels per chunk: 64, 48 or 32 (Cin % CW == 0); weights packed by launch_pack_conv_bf16_rh as
// [N][kh][Cin / CW][kw][CW].  Ping-pong schedule only (one stage of 320 * CW halves: 40 / 30 / 20 KiB).
// =====================================================================================================
#if defined(__HIP_DEVICE_COMPILE__)
// TN = 32-column blocks per tile (tile = 126 output pixels x 32 TN channels); the four waves split the 128 staged rows, each
// computing 32 rows x 32 TN columns (1 A + TN B fragment reads per TN MFMAs).  LDS: (128 + 96 TN) x CW halves, at least the
// epilogue's 18 KiB.
template <int CW, int TM, int TN>
__device__ __forceinline__ void igemm_bf16_rh_tile(const GemmArgs& p, const int bid, unsigned short* __restrict__ lds) {
    constexpr int BMS = 128 * TM, BMO = BMS - 2, BN = 32 * TN;     // staged rows, output pixels, output channels per tile
    constexpr int QPR = CW / 8;                            // 16-byte quads per LDS row
    constexpr int KS = CW / 16;                            // MFMA k-steps per tap
    constexpr int RA = TM * QPR / 2;                       // DMA rounds (256 quads each) of the BMS x CW A tile
    constexpr int NBQ = 3 * BN * QPR;                      // quads of the 3 x BN x CW weight tile
    constexpr int RB = (NBQ + 255) / 256;
    constexpr int BOFF = BMS * CW;                         // halves
    static_assert(CW == 64 || CW == 48 || CW == 32, "chunk width");

    const unsigned short* A = reinterpret_cast<const unsigned short*>(p.A);
    const unsigned short* Wp = reinterpret_cast<const unsigned short*>(p.Wp);
    const unsigned short* Rs = reinterpret_cast<const unsigned short*>(p.res);
    unsigned short* Out = reinterpret_cast<unsigned short*>(p.out);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CAPF_DIAG
    unsigned long long dbg_t0 = 0, dbg_t1 = 0, dbg_t2 = 0, dbg_t3 = 0, dbg_w = 0, dbg_a = 0, dbg_b = 0;
    const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    B16_STAMP(dbg_t0);
    const int nbn = (p.N + BN - 1) / BN;
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * BMO, n0 = tile_n * BN;
    const int HW = p.H * p.W, Kp = 9 * p.Cin, ncc = p.Cin / CW;
    // XOR swizzle of the 16-byte quad inside an LDS row, chosen so that the 16 lanes of a ds_read_b128 group (rows covering every
    // residue mod 16, all reading the same logical quad) hit 16 different bank quads.  128-B rows: (row >> 1) & 7; 64-B rows:
    // (row >> 2) & 3; 96-B rows (CW = 48, HRNet-48's 48- and 96-channel branches): rows r and r + 8 start on the same bank
    // (8 x 96 B = 3 x 256 B), so they swap adjacent quads: (row >> 3) & 1 (quad ^ 1 stays inside 0..5).  Without it every
    // fragment read of those branches was a 2-way conflict: SQ_LDS_BANK_CONFLICT = 33 % of SQ_LDS_IDX_ACTIVE on the grouped
    // HRNet-48 level (profiles/r03_pmc_bf16_rh.txt).
    auto swz = [](int row) { return QPR == 8 ? (row >> 1) & 7 : (QPR == 4 ? (row >> 2) & 3 : (row >> 3) & 1); };

    // ---- operand fetch: LDS-DMA with block-uniform descriptors (see igemm_bf16_tile); the descriptor of A starts one
    // pixel and one image row before the tile so that every (row, kh) offset is non-negative
    constexpr unsigned OOB = 0x80000000u;
    const rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + ((long)m0 - 1 - p.W) * p.Cin), 0, 0x7FFFFF00u, 0x00020000);
    const rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(Wp + (long)n0 * Kp), 0,
                                                            (unsigned)(p.N - n0) * (unsigned)Kp * 2u, 0x00020000);
    unsigned a_voff[RA], a_ok[RA];                         // a_ok bit kh: staged row (pixel, shifted by kh-1) is real data
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int q = r * 256 + tid, j = q / QPR, pq = q - j * QPR;
        const int mj = m0 - 1 + j;
        a_voff[r] = (unsigned)(j * p.Cin + ((pq ^ swz(j)) * 8)) * 2u;
        a_ok[r] = 0u;
        if (mj >= 0 && mj < p.M) {
            const int b = fast_div_b(mj, p.fd_hw), rem = mj - b * HW;
            const int h = fast_div_b(rem, p.fd_wo);
            a_ok[r] = (h >= 1 ? 1u : 0u) | 2u | (h + 1 < p.H ? 4u : 0u);
        }
    }
    unsigned b_voff[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int q = r * 256 + tid;
        const int kw = q / (BN * QPR), rem = q - kw * (BN * QPR);
        const int n = rem / QPR, pq = rem - n * QPR;
        b_voff[r] = (q < NBQ) ? (unsigned)(n * Kp + kw * CW + ((pq ^ swz(n)) * 8)) * 2u : OOB;
    }
    auto fire = [&](int kh, int cc) {
        const unsigned soff_a = __builtin_amdgcn_readfirstlane((unsigned)(kh * p.W * p.Cin + cc * CW) * 2u);
        const unsigned soff_b = __builtin_amdgcn_readfirstlane((unsigned)((kh * ncc + cc) * 3 * CW) * 2u);
#pragma unroll
        for (int r = 0; r < RA; ++r)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(lds + (r * 256 + wave * 64) * 8), 16,
                                                     ((a_ok[r] >> kh) & 1u) ? a_voff[r] : OOB, soff_a, 0, 0);
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (r * 256 + wave * 64 < NBQ)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(lds + BOFF + (r * 256 + wave * 64) * 8), 16,
                                                         b_voff[r], soff_b, 0, 0);
    };
    // the accumulators start at the bias (transposed layout: register 4 g + e of block j = channel n0 + 32 j + 8 g + 4 fhalf + e),
    // so the epilogue needs no bias operand; the loads retire behind the first superchunk's
    const int wm0 = wave * 32 * TM;
    const int frow = lane & 31, fhalf = lane >> 5;
    const rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)(p.bias + n0) : (void*)Out, 0,
                                                             p.bias ? (unsigned)(p.N - n0) * 4u : 0u, 0x00020000);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                           rs_bias, (unsigned)(j * 32 + 8 * g + 4 * fhalf) * 4u, 0, 0));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = bv[e];
        }
    bool zl[TM], zr[TM];                                   // this lane's output pixel sits on the left / right image border
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + i * 32 + frow;
        const int w = m - fast_div_b(m, p.fd_wo) * p.W;    // (fd_wo divides by W: Wo == W here)
        zl[i] = w == 0;
        zr[i] = w == p.W - 1;
    }
    const int b_sw = swz(frow);                            // (weight rows j * 32 + frow: the swizzle only looks at frow's bits)

    // residual rows in the coalesced epilogue's layout, requested with the last superchunk (see igemm_bf16_tile)
    const bool vec_ok = (p.N & 7) == 0 && (p.omap.S1 & 7) == 0 && (p.omap.off & 7) == 0 &&
                        (!Rs || ((p.rmap.S1 & 7) == 0 && (p.rmap.off & 7) == 0));
    const int er = lane >> 2, ec = (lane & 3) * 8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rr[TM][TN][2];
    // (raw buffer accesses on block-local descriptors, out-of-range offsets instead of branches: see igemm_bf16_tile)
    const rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        (Rs && vec_ok) ? (void*)(Rs + (long)m0 * p.rmap.S1 + p.rmap.off + n0) : (void*)Out, 0, (Rs && vec_ok) ? 0x7FFFFF00u : 0u,
        0x00020000);
    const rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(Out + (long)m0 * p.omap.S1 + p.omap.off + n0), 0,
                                                            vec_ok ? 0x7FFFFF00u : 0u, 0x00020000);
    auto row_ok = [&](int i_local, int m) { return i_local < BMO && m < p.M; };
    auto piece_off = [&](int i, int j, int h, int S1) -> unsigned {
        const int il = wm0 + i * 32 + h * 16 + er, nl = j * 32 + ec;
        return (row_ok(il, m0 + il) && n0 + nl < p.N) ? (unsigned)(il * S1 + nl) * 2u : OOB;
    };
    auto prefetch_epilogue = [&]() {
        if (!vec_ok) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    rr[i][j][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, piece_off(i, j, h, (int)p.rmap.S1), 0, 0);
    };

    bf16x8 af[2][TM], bfr[2][TN];
    auto read_frags = [&](int u, int buf) {                // u = kw * KS + k-step
        const int kw = u / KS, st = u - kw * KS;
        const int lq = st * 2 + fhalf;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm0 + i * 32 + frow + kw;
            f32x4 v = *reinterpret_cast<const f32x4*>(&lds[(r * QPR + (lq ^ swz(r))) * 8]);
            if ((kw == 0 && zl[i]) || (kw == 2 && zr[i])) v = f32x4{0.f, 0.f, 0.f, 0.f};
            af[buf][i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bfr[buf][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(
                                                         &lds[BOFF + ((kw * BN + j * 32 + frow) * QPR + (lq ^ b_sw)) * 8]));
    };
    B16_STAMP(dbg_t1);
    auto superchunk = [&](int kh, int cc) {
        B16_STAMP(dbg_a);
        fire(kh, cc);
        wait_vmcnt_b<0>();
        __builtin_amdgcn_s_barrier();
#ifdef CAPF_DIAG
        B16_STAMP(dbg_b);
        dbg_w += dbg_b - dbg_a;
#endif
        read_frags(0, 0);
#pragma unroll
        for (int u = 0; u < 3 * KS; ++u) {
            if (u + 1 < 3 * KS) read_frags(u + 1, (u + 1) & 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[u & 1][j], af[u & 1][i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // every wave has read the stage: the next superchunk may overwrite it
    };
    const int nsc = 3 * ncc;
    int kh = 0, cc = 0;
    for (int sc = 0; sc < nsc - 1; ++sc) {
        superchunk(kh, cc);
        if (++cc == ncc) { cc = 0; ++kh; }
    }
    prefetch_epilogue();                                   // (last superchunk peeled: these registers are not live in the loop)
    superchunk(kh, cc);

    B16_STAMP(dbg_t2);
    // ---- epilogue: the coalesced bf16 epilogue of igemm_bf16_tile (32x32 blocks transposed through the idle stage)
    constexpr int EPS = 36;
    float* ep = reinterpret_cast<float*>(lds) + wave * (32 * EPS);
    auto finish = [&](float t) { return p.act == ACT_RELU ? fmaxf(t, 0.f) : t; };
    auto transpose_block = [&](int i, int j) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(&ep[(lane & 31) * EPS + 8 * g + 4 * (lane >> 5)]) =
                f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        __builtin_amdgcn_wave_barrier();
    };
    if (vec_ok) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                transpose_block(i, j);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = h * 16 + er;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec]);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec + 4]);
                    u32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned rw = rr[i][j][h][q];
                        const float xa = q < 2 ? x0[2 * q] : x1[2 * q - 4], xb = q < 2 ? x0[2 * q + 1] : x1[2 * q - 3];
                        o[q] = pack_bf16x2(finish(xa + __uint_as_float(rw << 16)), finish(xb + __uint_as_float(rw & 0xFFFF0000u)));
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, piece_off(i, j, h, (int)p.omap.S1), 0, 0);
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + j * 32 + ec;
                transpose_block(i, j);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = h * 16 + er, il = wm0 + i * 32 + row, m = m0 + il;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec]);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec + 4]);
                    if (!row_ok(il, m)) continue;
                    for (int e = 0; e < 8 && n + e < p.N; ++e) {
                        const float x = e < 4 ? x0[e & 3] : x1[e & 3];
                        const float rsv = Rs ? bf2f(Rs[(long)m * p.rmap.S1 + p.rmap.off + n + e]) : 0.f;
                        Out[(long)m * p.omap.S1 + p.omap.off + n + e] = f2bf(finish(x + rsv));
                    }
                }
            }
    }
#ifdef CAPF_DIAG
    B16_STAMP(dbg_t3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long* d = capf_bf16_timeline + (size_t)blockIdx.x * 8;
        d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[3] = dbg_t3; d[4] = __builtin_amdgcn_s_memtime();
        d[5] = dbg_r0; d[6] = dbg_w; d[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}
#endif

__device__ __forceinline__ int xcd_remap_b(int b, int nblk) {   // see igemm_f32.hip :: xcd_remap
    const int q = nblk >> 3, r = nblk & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <int BM, int BN, int WM, int WN, int S, bool OUTF32 = false, bool GELU = false, bool UPADD = false>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int HALVES = S * (BM + BN) * BKH < 4 * 32 * 36 * 2 ? 4 * 32 * 36 * 2 : S * (BM + BN) * BKH;   // >= the epilogue's 18 KiB
    __shared__ __attribute__((aligned(16))) unsigned short lds[HALVES];
    igemm_bf16_tile<BM, BN, WM, WN, S, OUTF32, GELU, UPADD>(p, xcd_remap_b(blockIdx.x, gridDim.x), lds);
#endif
}

// =====================================================================================================
// Small-Cin stem (Cin = 3, k = 3 or 7) under compute_dtype = bf16: the fp32 image is gathered element-wise (K order
// (kh, kw, c), like the fp32 igemm_f32_smallc kernel whose packed fp32 weights it shares), rounded to bf16 on the way into
// a padded LDS tile, and multiplied on v_mfma_f32_32x32x16_bf16: 1/8 of the matrix cycles of the fp32 stem, which was
// 1.1 ms of CPN's 9.7 ms at batch 128 (384x288) -- the gather is what remains.  Output bf16 NHWC.
// =====================================================================================================
[[maybe_unused]] static constexpr int SPITCH = 40;          // halves per LDS row: 32 k-values + 8 pad (80 B: 16-byte aligned fragment reads)

__global__ __launch_bounds__(256) void igemm_bf16_smallc_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, BN = 64, WM = 64, WN = 32, SBK = 32;
    constexpr int WAVES_N = BN / WN, TM = WM / 32, TN = WN / 32, RA = BM / 32, RB = BN / 32;
    __shared__ __attribute__((aligned(16))) unsigned short lds[(BM + BN) * SPITCH];
    unsigned short* As = lds;
    unsigned short* Bs = lds + BM * SPITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbn = (p.N + BN - 1) / BN;
    const int bid = xcd_remap_b(blockIdx.x, gridDim.x);
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int srow = tid >> 3, kq = (tid & 7) * 4;

    long a_base[RA];
    int a_h0[RA], a_w0[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + srow + 32 * i;
        a_base[i] = 0; a_h0[i] = -(1 << 20); a_w0[i] = 0;
        if (m < p.M) {
            const int b = fast_div_b(m, p.fd_hw), rem = m - b * p.Ho * p.Wo;
            const int ho = fast_div_b(rem, p.fd_wo), wo = rem - ho * p.Wo;
            a_base[i] = (long)b * p.H * p.W * p.Cin;
            a_h0[i] = ho * p.stride - p.pad;
            a_w0[i] = wo * p.stride - p.pad;
        }
    }
    float a_reg[RA][4];
    f32x4 b_reg[RB];
    auto load_chunk = [&](int c) {
        int dh[4], dw[4], doff[4];
        bool kok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {          // the (kh, kw, c) of this thread's four k-values: the same for every row
            const int ke = c * SBK + kq + e;
            const int t = ke / p.Cin, c1 = ke - t * p.Cin;
            dh[e] = t / p.ks;
            dw[e] = t - dh[e] * p.ks;
            doff[e] = (dh[e] * p.W + dw[e]) * p.Cin + c1;
            kok[e] = ke < p.K;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const long row = a_base[i] + ((long)a_h0[i] * p.W + a_w0[i]) * p.Cin;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int hi = a_h0[i] + dh[e], wi = a_w0[i] + dw[e];
                a_reg[i][e] = (kok[e] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) ? p.A[row + doff[e]] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = n0 + srow + 32 * i;
            b_reg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (n < p.N) b_reg[i] = *reinterpret_cast<const f32x4*>(p.Wp + (long)n * p.Kpad + c * SBK + kq);
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int nchunks = p.Kpad / SBK;          // the fp32 pack pads K to a multiple of 32
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int i = 0; i < RA; ++i)
            *reinterpret_cast<u32x2*>(&As[(srow + 32 * i) * SPITCH + kq]) =
                u32x2{pack_bf16x2(a_reg[i][0], a_reg[i][1]), pack_bf16x2(a_reg[i][2], a_reg[i][3])};
#pragma unroll
        for (int i = 0; i < RB; ++i)
            *reinterpret_cast<u32x2*>(&Bs[(srow + 32 * i) * SPITCH + kq]) =
                u32x2{pack_bf16x2(b_reg[i][0], b_reg[i][1]), pack_bf16x2(b_reg[i][2], b_reg[i][3])};
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
        for (int step = 0; step < SBK / 16; ++step) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&As[(wm0 + i * 32 + frow) * SPITCH + step * 16 + fhalf * 8]));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&Bs[(wn0 + j * 32 + frow) * SPITCH + step * 16 + fhalf * 8]));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // transposed accumulator (lane = row m, register group g = channels 8 g + 4 (lane >> 5) .. + 3): 8-byte stores
    unsigned short* Out = reinterpret_cast<unsigned short*>(p.out);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + i * 32 + frow;
        if (m >= p.M) continue;
        const long o_row = (long)m * p.omap.S1 + p.omap.off;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + j * 32 + 4 * fhalf + 8 * g;
                if (n >= p.N) continue;
                u16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][j][4 * g + e] + (p.bias ? p.bias[n + e] : 0.f);
                    if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
                    v[e] = f2bf(t);
                }
                *reinterpret_cast<u16x4*>(Out + o_row + n) = v;
            }
    }
#endif
}

// Stem, second version (Cin = 3, k = 7 or 3): for a fixed (output pixel, kh) the 3 k input values are CONTIGUOUS in the NHWC
// image (and in the (kh, kw, c)-ordered weights), so a thread fetches one such run with 16-byte loads (dword-aligned
// addresses) instead of 21 / 9 scalar gathers, zeroes what lies outside the image, and writes it as bf16 into an LDS tile
// whose K axis is (kh, 24 | 16): the whole K of a 128 x 64 tile is staged once (no chunk loop), then 11 / 3 MFMA k-steps.
template <int KS, int BM>
__global__ __launch_bounds__(256) void igemm_bf16_stem_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BN = 64, WM = BM / 2, WN = 32, TM = WM / 32;
    constexpr int RUN = KS * 3;                      // contiguous floats per (pixel, kh)
    constexpr int NX4 = (RUN + 3) / 4;               // 16-byte loads per run
    constexpr int PKH = (RUN + 7) / 8 * 8;           // halves per kh in LDS (24 / 16)
    constexpr int KP = (KS * PKH + 15) / 16 * 16;    // staged K (176 / 48)
    constexpr int PITCH = KP + 8;                    // halves per LDS row: 16-byte aligned, conflict-free b128 reads
    __shared__ __attribute__((aligned(16))) unsigned short lds[(BM + BN) * PITCH];
    unsigned short* As = lds;
    unsigned short* Bs = lds + BM * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbn = (p.N + BN - 1) / BN;
    const int bid = xcd_remap_b(blockIdx.x, gridDim.x);
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long total = (long)(p.M / (p.Ho * p.Wo)) * p.H * p.W * 3;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    auto put_run = [&](unsigned short* dst, const float (&v)[NX4 * 4], int kh) {     // RUN values (+ zero padding) -> LDS as bf16
        unsigned pk[PKH / 2];
#pragma unroll
        for (int e = 0; e < PKH / 2; ++e)
            pk[e] = pack_bf16x2(2 * e < NX4 * 4 ? v[2 * e] : 0.f, 2 * e + 1 < NX4 * 4 ? v[2 * e + 1] : 0.f);
#pragma unroll
        for (int q = 0; q < PKH / 8; ++q)
            *reinterpret_cast<u32x4*>(dst + kh * PKH + q * 8) = u32x4{pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]};
        if (KP > KS * PKH && kh == KS - 1)                                         // the tail of the last k-step
            *reinterpret_cast<u32x4*>(dst + KS * PKH) = u32x4{0u, 0u, 0u, 0u};
    };
    for (int t = tid; t < BM * KS; t += 256) {               // activation runs: consecutive lanes = consecutive pixels
        const int kh = t / BM, row = t - kh * BM;
        const int m = m0 + row;
        float v[NX4 * 4];
#pragma unroll
        for (int e = 0; e < NX4 * 4; ++e) v[e] = 0.f;
        if (m < p.M) {
            const int b = fast_div_b(m, p.fd_hw), rem = m - b * p.Ho * p.Wo;
            const int ho = fast_div_b(rem, p.fd_wo), wo = rem - ho * p.Wo;
            const int hi = ho * p.stride - p.pad + kh, wi0 = wo * p.stride - p.pad;
            if ((unsigned)hi < (unsigned)p.H) {
                const long src = (((long)b * p.H + hi) * p.W + wi0) * 3;
#pragma unroll
                for (int x = 0; x < NX4; ++x) {
                    const long o = src + 4 * x;
                    if (o >= 0 && o + 4 <= total) {
                        __builtin_memcpy(&v[4 * x], p.A + o, 16);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (o + e >= 0 && o + e < total) v[4 * x + e] = p.A[o + e];
                    }
                }
#pragma unroll
                for (int e = 0; e < NX4 * 4; ++e)
                    if (e >= RUN || (unsigned)(wi0 + e / 3) >= (unsigned)p.W) v[e] = 0.f;
            }
        }
        put_run(As + row * PITCH, v, kh);
    }
    for (int t = tid; t < BN * KS; t += 256) {               // weight runs from the fp32 pack [N][Kpad], K order (kh, kw, c)
        const int kh = t / BN, n = t - kh * BN;
        float v[NX4 * 4];
#pragma unroll
        for (int e = 0; e < NX4 * 4; ++e) v[e] = 0.f;
        if (n0 + n < p.N) {
            const float* src = p.Wp + (long)(n0 + n) * p.Kpad + kh * RUN;          // (kh * RUN + 4 NX4 <= Kpad: launcher checks)
#pragma unroll
            for (int x = 0; x < NX4; ++x) __builtin_memcpy(&v[4 * x], src + 4 * x, 16);
#pragma unroll
            for (int e = RUN; e < NX4 * 4; ++e) v[e] = 0.f;
        }
        put_run(Bs + n * PITCH, v, kh);
    }
    __syncthreads();

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
    for (int step = 0; step < KP / 16; ++step) {
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&Bs[(wn0 + frow) * PITCH + step * 16 + fhalf * 8]));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&As[(wm0 + i * 32 + frow) * PITCH + step * 16 + fhalf * 8]));
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, af, acc[i], 0, 0, 0);
        }
    }
    // transposed accumulator (lane = row m, register group g = channels 8 g + 4 fhalf .. + 3): 8-byte stores
    unsigned short* Out = reinterpret_cast<unsigned short*>(p.out);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn0 + 4 * fhalf + 8 * g;
        if (n >= p.N) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + i * 32 + frow;
            if (m >= p.M) continue;
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t[e] = acc[i][4 * g + e] + bv[e];
                if (p.act == ACT_RELU) t[e] = fmaxf(t[e], 0.f);
            }
            *reinterpret_cast<u32x2*>(Out + (long)m * p.omap.S1 + p.omap.off + n) = u32x2{pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2], t[3])};
        }
    }
#endif
}

// Stem, third version: PERSISTENT blocks with the weights staged once and the next tile's runs in flight while the current
// tile is multiplied and stored.  The second version above is one latency chain per block -- gather (6 x 16-byte loads per run
// behind per-quad branches, i.e. one L2 round trip after the other), LDS, 11 MFMAs, 8-byte strided stores -- with only two
// 70 KiB blocks per CU to hide it: 488 us for CPN's 7x7 at batch 128 (1.3 TB/s; 453 MB out + 170 MB in would take 125 us at
// 5 TB/s).  Here: 64-pixel tiles (47 KiB of operands + 18 KiB of epilogue scratch: two blocks per CU), every run fetched with
// raw buffer loads on ONE descriptor over the whole image tensor (a quad outside it, or of a row above / below the image,
// gets an out-of-range offset; a quad only PARTLY beyond the tensor's end returns its in-range dwords -- range checking is
// per dword, tools/buffer_oob.hip), so all of a thread's 12 / 3 loads are in flight at once and stay in flight across the
// MFMAs and the stores of the previous tile; the epilogue transposes through LDS and stores 16 bytes per lane, 64 B
// contiguous per row.  XCD-aware tile order: each XCD walks its own contiguous eighth of the tiles, so the 7 / 3 input rows
// that vertically adjacent tiles share are fetched into ONE L2.  Only the handful of runs at the very start of the tensor (negative
// offsets cannot be expressed) are patched element-wise by the blocks that own those pixels.
template <int KS>
__global__ __launch_bounds__(256, 2) void igemm_bf16_stem_stream_kernel(GemmArgs p, int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 64, BN = 64;
    constexpr int RUN = KS * 3;                      // contiguous floats per (pixel, kh)
    constexpr int NX4 = RUN / 4;                     // 16-byte loads per run (5 / 2) ...
    static_assert(RUN % 4 == 1, "... plus ONE dword: a 16-byte load whose upper dwords nobody reads leaves dead destination "
                                "registers that the allocator reuses -- and a vmcnt(0) in front of that reuse");
    constexpr int PKH = (RUN + 7) / 8 * 8;           // halves per kh in LDS (24 / 16)
    constexpr int KP = (KS * PKH + 15) / 16 * 16;    // staged K (176 / 48)
    constexpr int PITCH = KP + 8;                    // halves per LDS row: 16-byte aligned, conflict-free b128 reads
    constexpr int NT = (BM * KS + 255) / 256;        // runs per thread and tile (2 / 1)
    constexpr int EPS = 36;
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned short lds[(BM + BN) * PITCH + 4 * 32 * EPS * 2];
    unsigned short* As = lds;
    unsigned short* Bs = lds + BM * PITCH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* ep = reinterpret_cast<float*>(lds + (BM + BN) * PITCH) + wave * (32 * EPS);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned short* Out = reinterpret_cast<unsigned short*>(p.out);
    const long total = (long)(p.M / (p.Ho * p.Wo)) * p.H * p.W * 3;              // floats of the image tensor (< 2^29: launcher)
    const rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)(total * 4), 0x00020000);
    const rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)p.bias : (void*)Out, 0,
                                                             p.bias ? (unsigned)p.N * 4u : 0u, 0x00020000);

    auto put_run = [&](unsigned short* dst, const f32x4 (&q)[NX4], float last, unsigned mask, int kh) {   // RUN values (+ zeros) -> LDS
        auto val = [&](int e) { return e < NX4 * 4 ? q[e >> 2][e & 3] : (e == NX4 * 4 ? last : 0.f); };
        unsigned pk[PKH / 2];
        if (mask == (1u << RUN) - 1u) {              // interior pixel (all but ~1 run in 70): no per-element selects
#pragma unroll
            for (int e = 0; e < PKH / 2; ++e) pk[e] = pack_bf16x2(2 * e < RUN ? val(2 * e) : 0.f, 2 * e + 1 < RUN ? val(2 * e + 1) : 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < PKH / 2; ++e) {
                const float a = 2 * e < RUN && ((mask >> (2 * e)) & 1u) ? val(2 * e) : 0.f;
                const float b = 2 * e + 1 < RUN && ((mask >> (2 * e + 1)) & 1u) ? val(2 * e + 1) : 0.f;
                pk[e] = pack_bf16x2(a, b);
            }
        }
#pragma unroll
        for (int g = 0; g < PKH / 8; ++g)
            *reinterpret_cast<u32x4*>(dst + kh * PKH + g * 8) = u32x4{pk[4 * g], pk[4 * g + 1], pk[4 * g + 2], pk[4 * g + 3]};
        if (KP > KS * PKH && kh == KS - 1)                                         // the tail of the last k-step
            *reinterpret_cast<u32x4*>(dst + KS * PKH) = u32x4{0u, 0u, 0u, 0u};
    };

    // ---- weights, once per block: runs of the fp32 pack [N][Kpad], K order (kh, kw, c)
    {
        const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (unsigned)p.N * (unsigned)p.Kpad * 4u, 0x00020000);
        for (int t = tid; t < BN * KS; t += 256) {
            const int kh = t / BN, n = t - kh * BN;
            f32x4 q[NX4];
            const unsigned wo = n < p.N ? (unsigned)(n * p.Kpad + kh * RUN) * 4u : OOB;
#pragma unroll
            for (int x = 0; x < NX4; ++x)
                q[x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, n < p.N ? wo + 16u * x : OOB, 0, 0));
            const float last = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w, n < p.N ? wo + 16u * NX4 : OOB, 0, 0));
            put_run(Bs + n * PITCH, q, last, (1u << RUN) - 1u, kh);
        }
    }
    // XCD-aware persistent walk: XCD x owns tiles [x * per_xcd, (x + 1) * per_xcd); its blocks take them round-robin
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;        // (gridDim.x % 8 == 0: launcher)
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_end = min(ntiles, (xcd + 1) * per_xcd);
    int tile = xcd * per_xcd + slot;

    f32x4 q[NT][NX4];
    float qlast[NT];
    unsigned mask[NT];
    // run i of this thread; a thread without an i-th run repeats another one (same loads, the same LDS bytes written twice) --
    // a branch around put_run would leave "maybe pending" loads behind it and cost a vmcnt(0) in front of the next requests
    auto run_of = [&](int i) { return (tid + 256 * i) % (BM * KS); };
    auto issue = [&](int tl) {                       // request the runs of tile tl (tl >= t_end: every offset out of range)
        const int m0 = tl * BM;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = run_of(i);
            const int kh = t / BM, row = t - kh * BM;
            const int m = m0 + row;
            bool ok = tl < t_end && m < p.M;
            const int mm = ok ? m : 0;
            const int b = fast_div_b(mm, p.fd_hw), rem = mm - b * p.Ho * p.Wo;
            const int ho = fast_div_b(rem, p.fd_wo), wo = rem - ho * p.Wo;
            const int hi = ho * p.stride - p.pad + kh, wi0 = wo * p.stride - p.pad;
            ok = ok && (unsigned)hi < (unsigned)p.H;
            const int o = ((b * p.H + hi) * p.W + wi0) * 3;                      // floats from the tensor start (may be < 0 at its start)
            // elements 3 lo .. 3 hn - 1 of the run lie inside the image row
            const int lo = max(0, -wi0), hn = max(lo, min(KS, p.W - wi0));
            const unsigned mk = ((1u << (3 * hn)) - 1u) & ~((1u << (3 * lo)) - 1u);
            mask[i] = ok ? mk : 0u;
#pragma unroll
            for (int x = 0; x < NX4; ++x)
                q[i][x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                        rs_in, (ok && o + 4 * x >= 0) ? (unsigned)(o + 4 * x) * 4u : OOB, 0, 0));
            qlast[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     rs_in, (ok && o + 4 * NX4 >= 0) ? (unsigned)(o + 4 * NX4) * 4u : OOB, 0, 0));
        }
    };
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int er = lane >> 2, ec = (lane & 3) * 8;
    const f32x4 b0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, (unsigned)(wn0 + ec) * 4u, 0, 0));
    const f32x4 b1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, (unsigned)(wn0 + ec + 4) * 4u, 0, 0));

    issue(tile);
    {   // two dropped stores: the loop is entered with the same "loads, then two stores" in flight as its back edge leaves, so the
        // wait in front of the LDS writes is a counted vmcnt(2) -- not a vmcnt(0) that would sit out the previous tile's stores
        const rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc((void*)Out, 0, 0u, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rs_none, OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rs_none, OOB + 16u, 0, 0);   // (distinct: identical stores are merged)
    }
    for (; tile < t_end; tile += nslots) {
        const int m0 = tile * BM;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = run_of(i);
            const int kh = t / BM, row = t - kh * BM;
            put_run(As + row * PITCH, q[i], qlast[i], mask[i], kh);
        }
        const int ho_max = p.pad / p.stride;           // output rows (of frame 0) with a window row on input row 0
        if (m0 < (ho_max + 1) * p.Wo) {
            // Frame 0, input row 0, window starting left of the image: the run begins at a NEGATIVE offset from the tensor start,
            // which a buffer offset cannot express -- its first quads were requested out of range above.  The few such runs
            // (output rows ho = (pad - kh) / stride, columns wo * stride < pad) are rewritten here element by element.
            __syncthreads();
            const int nw = min(p.Wo, (p.pad + p.stride - 1) / p.stride);
            for (int idx = tid; idx < (ho_max + 1) * nw * PKH; idx += 256) {
                const int e = idx % PKH, r = idx / PKH, wo = r % nw, ho = r / nw;
                const int kh = p.pad - ho * p.stride, m = ho * p.Wo + wo;
                if (kh >= 0 && kh < KS && m >= m0 && m < m0 + BM) {
                    const int wi = wo * p.stride - p.pad + e / 3;
                    float v = 0.f;
                    if (e < RUN && (unsigned)wi < (unsigned)p.W) v = p.A[wi * 3 + e % 3];
                    As[(m - m0) * PITCH + kh * PKH + e] = f2bf(v);
                }
            }
        }
        __syncthreads();
        issue(tile + nslots);                         // in flight across this tile's MFMAs and stores

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int step = 0; step < KP / 16; ++step) {
            const bf16x8 bfr = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&Bs[(wn0 + frow) * PITCH + step * 16 + fhalf * 8]));
            const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&As[(wm0 + frow) * PITCH + step * 16 + fhalf * 8]));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, af, acc, 0, 0, 0);
        }
        // transposed accumulator (lane = row, register group g = channels 8 g + 4 fhalf ..) -> LDS -> 8 channels of a row per lane
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(&ep[frow * EPS + 8 * g + 4 * fhalf]) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        __builtin_amdgcn_wave_barrier();
        const rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(Out + (long)m0 * p.omap.S1 + p.omap.off), 0, 0x7FFFFF00u, 0x00020000);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = h * 16 + er, ml = wm0 + row, nl = wn0 + ec;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec]);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec + 4]);
            float t[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { t[e] = x0[e] + b0[e]; t[4 + e] = x1[e] + b1[e]; }
            if (p.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = fmaxf(t[e], 0.f);
            }
            const u32x4 o = u32x4{pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2], t[3]), pack_bf16x2(t[4], t[5]), pack_bf16x2(t[6], t[7])};
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, (m0 + ml < p.M && nl < p.N) ? (unsigned)(ml * (int)p.omap.S1 + nl) * 2u : OOB, 0, 0);
        }
        __syncthreads();                              // every wave has read As: the next tile's runs may overwrite it
    }
#endif
}

// fp32 NHWC image (Cin % 4 != 0), fp32 packed weights [N][Kpad32], bf16 NHWC result; no residual.
bool gemm_bf16_smallc_ok(const GemmArgs& a) {
    return a.conv && a.out_bf16 && !a.res && a.omap.G == 1 && a.N % 4 == 0 && a.omap.S1 % 4 == 0 && a.omap.off % 4 == 0 &&
           a.Kpad % 32 == 0 && a.act != ACT_GELU;
}

static int stem_runs_ks(const GemmArgs& a) {     // 7 / 3: the run-based stem kernel applies; 0: the element-wise gather kernel
    static const int v2 = [] { const char* e = diag_env("CAPF_STEM_V2"); return e ? atoi(e) : 1; }();     // A/B runs only
    if (!v2 || a.Cin != 3 || a.pad != a.ks / 2 || a.K != a.ks * a.ks * 3) return 0;
    if (a.ks == 7 && a.Kpad >= 6 * 21 + 24) return 7;
    if (a.ks == 3 && a.Kpad >= 2 * 9 + 12) return 3;
    return 0;
}

// the persistent streaming stem: one 64-column tile, 8-channel vector stores, 32-bit float offsets into the image tensor
static bool stem_stream_ok(const GemmArgs& a) {
    static const int on = [] { const char* e = diag_env("CAPF_STEM_STREAM"); return e ? atoi(e) : 1; }();        // A/B runs only
    const long total = (long)(a.M / (a.Ho * a.Wo)) * a.H * a.W * 3;
    return on && a.N <= 64 && a.N % 8 == 0 && a.omap.S1 % 8 == 0 && a.omap.off % 8 == 0 && total < (1L << 29) && a.Wo >= 4;
}

const char* gemm_bf16_smallc_kernel_name(const GemmArgs& a) {
    if (!stem_runs_ks(a)) return "igemm_bf16_smallc<w4,128x64>";
    return stem_stream_ok(a) ? "igemm_bf16_stem_stream<w4,64x64>" : "igemm_bf16_stem<w4,128x64>";
}

hipError_t launch_gemm_bf16_smallc(const GemmArgs& a_in, hipStream_t s) {
    if (!gemm_bf16_smallc_ok(a_in)) return hipErrorInvalidValue;
    GemmArgs a = a_in;
    a.fd_hw = make_fastdiv((unsigned)(a.Ho * a.Wo));
    a.fd_wo = make_fastdiv((unsigned)a.Wo);
    const int ks = stem_runs_ks(a);
    if (ks && stem_stream_ok(a)) {
        const int ntiles = (a.M + 63) / 64;
        int blocks = 512;                              // two per CU
        while (blocks > 8 && blocks / 2 >= ntiles) blocks /= 2;
        if (ks == 7) hipLaunchKernelGGL((igemm_bf16_stem_stream_kernel<7>), dim3(blocks), dim3(256), 0, s, a, ntiles);
        else hipLaunchKernelGGL((igemm_bf16_stem_stream_kernel<3>), dim3(blocks), dim3(256), 0, s, a, ntiles);
        return hipGetLastError();
    }
    const dim3 grid(((a.M + 127) / 128) * ((a.N + 63) / 64));
    // (64-pixel tiles -- 47 KiB, three blocks per CU for the 7x7 -- measured slower: 0.50 -> 0.59 ms for CPN, 0.30 -> 0.35 for HRNet)
    if (ks == 7) hipLaunchKernelGGL((igemm_bf16_stem_kernel<7, 128>), grid, dim3(256), 0, s, a);
    else if (ks == 3) hipLaunchKernelGGL((igemm_bf16_stem_kernel<3, 128>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(igemm_bf16_smallc_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

static constexpr int rh_lds_halves(int cw, int tm, int tn) { return (128 * tm + 96 * tn) * cw < 9216 ? 9216 : (128 * tm + 96 * tn) * cw; }

template <int CW, int TM, int TN>
__global__ __launch_bounds__(256, TM == 1 ? 4 : 3) void igemm_bf16_rh_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) unsigned short lds[rh_lds_halves(CW, TM, TN)];
    igemm_bf16_rh_tile<CW, TM, TN>(p, xcd_remap_b(blockIdx.x, gridDim.x), lds);
#endif
}

// 32-column blocks per row-halo tile for N output channels at chunk width cw: 32 channels -> 1, multiples of 96 -> 3 (no padded
// columns; the 48-wide chunks keep 3 x 96 x 48 weights at 27 KiB), else 2
static int rh_tn(int N, int cw) { return N <= 32 ? 1 : ((N % 96 == 0 && cw == 48) ? 3 : 2); }

// chunk width of the row-halo kernel for Cin input channels (a multiple of 64, 48 or 32), 0 = not supported
int bf16_rh_width(int Cin) {
    static const int force = [] { const char* e = diag_env("CAPF_BF16_RH_CW"); return e ? atoi(e) : 0; }();   // tuning only
    if (force && Cin % force == 0 && (force == 64 || force == 48 || force == 32)) return force;
    return Cin % 64 == 0 ? 64 : (Cin % 48 == 0 ? 48 : (Cin % 32 == 0 ? 32 : 0));
}

// Measured against the ping-pong direct kernel at >= 2048 tiles (tools/bf16_timeline.py, TFLOP/s direct -> row-halo, final tile
// shapes): 64 ch 598 -> 666, 128 ch 756 -> 830, 192 ch 582 -> 671, 256 ch 772 -> 776, 48 ch 358 -> 411, 96 ch 469 -> 546, 32 ch
// 342 -> 472.  Inside a grouped launch every eligible problem takes the row-halo tile: mixing the two tile families in one grid
// measured slower (HRNet-48 batch 256: 10730 vs 11200 frames/s) than moving all of them.

// ... for a conv: 0 unless it is 3x3 / stride 1 / pad 1 with plain row maps
int gemm_bf16_rh_cw(const GemmArgs& a) {
    if (!a.conv || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W || a.N % 4 != 0 || a.act == ACT_GELU ||
        a.omap.G != 1 || (a.res && a.rmap.G != 1) || a.rscale || a.M <= 0 || (double)a.M * a.Cin * 2.0 >= 2.0e9)
        return 0;
    return bf16_rh_width(a.Cin);
}

static void prep_rh(GemmArgs& a) {
    a.fd_hw = make_fastdiv((unsigned)(a.H * a.W));
    a.fd_wo = make_fastdiv((unsigned)a.W);
}

// Wp = weights packed by launch_pack_conv_bf16_rh for the SAME chunk width gemm_bf16_rh_cw(a) returns
hipError_t launch_gemm_bf16_rh(const GemmArgs& a_in, hipStream_t s) {
    const int cw = gemm_bf16_rh_cw(a_in);
    if (!cw) return hipErrorInvalidValue;
    GemmArgs a = a_in;
    prep_rh(a);
    const int tn = rh_tn(a.N, cw);
    // TM = 2 (254-pixel tiles, wave tile 64 x 32 TN: a third less LDS traffic per MFMA, 2-3 blocks per CU) measured equal or slower
    // on every shape (48 ch 411 -> 398 TFLOP/s, 64 ch 666 -> 527 at CW 64 / 627 at CW 32, 128 ch 806 -> 793): not instantiated
    const dim3 grid(((a.M + 125) / 126) * ((a.N + 32 * tn - 1) / (32 * tn)));
#define RH_CASE(CW_, TN_) case CW_ * 10 + TN_: hipLaunchKernelGGL((igemm_bf16_rh_kernel<CW_, 1, TN_>), grid, dim3(256), 0, s, a); break;
    switch (cw * 10 + tn) {
        RH_CASE(64, 1) RH_CASE(64, 2) RH_CASE(48, 1) RH_CASE(48, 2) RH_CASE(48, 3) RH_CASE(32, 1) RH_CASE(32, 2)
        default: return hipErrorInvalidValue;
    }
#undef RH_CASE
    return hipGetLastError();
}

// BN fold + re-layout of a 3x3 conv weight for the row-halo kernel: Wp[n][((kh * (Cin / CW) + cc) * 3 + kw) * CW + c] =
// bf16(w[n][cc * CW + c][kh][kw] * gamma[n] / sqrt(var[n] + eps));  bias as launch_pack_conv
__global__ void pack_conv_bf16_rh_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                         unsigned short* __restrict__ Wp, float* __restrict__ bias, int Cout, int Cin, int CW) {
    const long total = (long)Cout * 9 * Cin;
    const int ncc = Cin / CW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / (9 * Cin));
        int k = (int)(i - (long)n * 9 * Cin);
        const bool first = k == 0;
        const int c = k % CW; k /= CW;
        const int kw = k % 3; k /= 3;
        const int cc = k % ncc, kh = k / ncc;
        const float sc = gamma ? gamma[n] / sqrtf(var[n] + eps) : 1.f;
        Wp[i] = f2bf(w[(((long)n * Cin + cc * CW + c) * 3 + kh) * 3 + kw] * sc);
        if (first && bias) bias[n] = gamma ? beta[n] - mean[n] * sc : 0.f;
    }
}

hipError_t launch_pack_conv_bf16_rh(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                                    float eps, void* Wp_bf16, float* bias, int Cout, int Cin, int CW, hipStream_t s) {
    if ((CW != 64 && CW != 48 && CW != 32) || Cin % CW != 0) return hipErrorInvalidValue;
    const long total = (long)Cout * 9 * Cin;
    const long want = (total + 255) / 256;
    hipLaunchKernelGGL(pack_conv_bf16_rh_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, s, w, gamma, beta, mean, var,
                       eps, static_cast<unsigned short*>(Wp_bf16), bias, Cout, Cin, CW);
    return hipGetLastError();
}

// Grouped launch (see igemm_f32.hip "Grouped launch"): up to MAXG independent bf16 convs in one grid.
struct GroupArgsB {
    GemmArgs g[MAXG];
    int start[MAXG + 1];
    int tiles[MAXG];
    int cfg[MAXG];         // 0: 128x64 (S=2), 1: 64x64 (S=3), 2: 128x32 (S=2)
    int n;
};
#ifndef CAPF_BF16_GROUP_STAGES
#define CAPF_BF16_GROUP_STAGES 2
#endif
// 48 KiB at 2 stages = 3 blocks per CU.  Measured alternatives (profiles/README.md, round 2): 3 stages / 72 KiB / 2 blocks per CU
// -6.5 % (HRNet-48 B=256), -2 % (CPN), -8 % (HRNet-32 B=64); 64x64 + 128x32 tiles only in 40 KiB / 4 blocks per CU -9 %, -1 %, 0 %.
#define GROUP_LDS_HALVES (CAPF_BF16_GROUP_STAGES * (128 + 64) * BKH)

__global__ __launch_bounds__(256) void igemm_bf16_group_kernel(GroupArgsB ga) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) unsigned short lds[GROUP_LDS_HALVES];
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < ga.n && b >= ga.start[pi + 1]) ++pi;
    const int l = b - ga.start[pi];
    const int per_xcd = (ga.start[pi + 1] - ga.start[pi]) >> 3;
    const int bid = (l & 7) * per_xcd + (l >> 3);
    if (bid >= ga.tiles[pi]) return;
    const GemmArgs& p = ga.g[pi];
    switch (ga.cfg[pi]) {
        case 0: igemm_bf16_tile<128, 64, 64, 32, CAPF_BF16_GROUP_STAGES>(p, bid, lds); break;
        case 1: igemm_bf16_tile<64, 64, 32, 32, CAPF_BF16_GROUP_STAGES + 1>(p, bid, lds); break;
        default: igemm_bf16_tile<128, 32, 32, 32, CAPF_BF16_GROUP_STAGES>(p, bid, lds); break;
    }
#endif
}

// ping-pong variant of the grouped kernel: one stage per block, 24 KiB, up to 5 blocks per CU (register cap 102)
__global__ __launch_bounds__(256, 5) void igemm_bf16_group_pp_kernel(GroupArgsB ga) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) unsigned short lds[(128 + 64) * BKH];
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < ga.n && b >= ga.start[pi + 1]) ++pi;
    const int l = b - ga.start[pi];
    const int per_xcd = (ga.start[pi + 1] - ga.start[pi]) >> 3;
    const int bid = (l & 7) * per_xcd + (l >> 3);
    if (bid >= ga.tiles[pi]) return;
    const GemmArgs& p = ga.g[pi];
    switch (ga.cfg[pi]) {
        case 0: igemm_bf16_tile<128, 64, 64, 32, 1>(p, bid, lds); break;
        case 1: igemm_bf16_tile<64, 64, 32, 32, 1>(p, bid, lds); break;
        default: igemm_bf16_tile<128, 32, 32, 32, 1>(p, bid, lds); break;
    }
#endif
}

// ping-pong grouped kernel for launches that contain row-halo problems (cfg 3.. = (chunk width, 32-column blocks) (64,2) (64,1)
// (48,2) (48,3) (48,1) (32,2) (32,1)); the stage size is the largest any problem of the launch needs (dynamic LDS, 18-40 KiB)
__global__ __launch_bounds__(256, 4) void igemm_bf16_group_rh_kernel(GroupArgsB ga) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_dyn[];
    unsigned short* lds = lds_dyn;
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < ga.n && b >= ga.start[pi + 1]) ++pi;
    const int l = b - ga.start[pi];
    const int per_xcd = (ga.start[pi + 1] - ga.start[pi]) >> 3;
    const int bid = (l & 7) * per_xcd + (l >> 3);
    if (bid >= ga.tiles[pi]) return;
    const GemmArgs& p = ga.g[pi];
    switch (ga.cfg[pi]) {
        case 0: igemm_bf16_tile<128, 64, 64, 32, 1>(p, bid, lds); break;
        case 1: igemm_bf16_tile<64, 64, 32, 32, 1>(p, bid, lds); break;
        case 2: igemm_bf16_tile<128, 32, 32, 32, 1>(p, bid, lds); break;
        case 3: igemm_bf16_rh_tile<64, 1, 2>(p, bid, lds); break;
        case 4: igemm_bf16_rh_tile<64, 1, 1>(p, bid, lds); break;
        case 5: igemm_bf16_rh_tile<48, 1, 2>(p, bid, lds); break;
        case 6: igemm_bf16_rh_tile<48, 1, 3>(p, bid, lds); break;
        case 7: igemm_bf16_rh_tile<48, 1, 1>(p, bid, lds); break;
        case 8: igemm_bf16_rh_tile<32, 1, 2>(p, bid, lds); break;
        default: igemm_bf16_rh_tile<32, 1, 1>(p, bid, lds); break;
    }
#endif
}

// The ping-pong schedule needs other resident blocks to cover a block's load phase: it is used from this many tiles per
// launch (8 per CU) and the ring schedule below.  Measured, HRNet-32 bf16 forward: ping-pong everywhere is 30 % slower at
// batch 1 / 8; thresholds 512 / 1024 / 2048 / 4096 / never give 9489 / 9599 / 10144 / 10137 / 10066 frames/s at batch 32 and
// 13699 / 13716 / 13713 / 13195 / 13052 at batch 64.
static int pp_min_tiles() {
    static const int v = [] { const char* e = diag_env("CAPF_BF16_PP_MIN_TILES"); return e ? atoi(e) : 2048; }();
    return v;
}

template <int BM, int BN, int WM, int WN, int S>
static hipError_t launch_cfg_b(const GemmArgs& a, hipStream_t s) {
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    if (a.up) {          // (+ bilinear_upsample(up) behind the activation: the two tile shapes a 256-channel lateral conv can get)
        if constexpr ((BM == 128 && BN == 128) || (BM == 64 && BN == 64))
            hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WM, WN, S, false, false, true>), dim3(nbm * nbn), dim3(256), 0, s, a);
        else
            return hipErrorInvalidValue;
        return hipGetLastError();
    }
    hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WM, WN, S>), dim3(nbm * nbn), dim3(256), 0, s, a);
    return hipGetLastError();
}

// the post-activation upsampled add needs the vector epilogue (8-channel pieces) and 32-bit offsets into the low-resolution map
bool gemm_bf16_upadd_ok(const GemmArgs& a) {
    return a.conv && a.up && a.N > 64 && a.N % 8 == 0 && a.omap.G == 1 && (a.omap.S1 & 7) == 0 && (a.omap.off & 7) == 0 && !a.rscale &&
           (!a.res || (a.rmap.G == 1 && (a.rmap.S1 & 7) == 0 && (a.rmap.off & 7) == 0)) && a.up_H > 0 && a.up_W > 0 && a.Ho * a.Wo > 0 &&
           (double)(a.M / (a.Ho * a.Wo)) * a.up_H * a.up_W * a.N * 2.0 < 2.0e9;
}

const char* gemm_bf16_kernel_name(const GemmArgs& a) {
    if (gemm_bf16_ws_wanted(a)) return gemm_bf16_ws_kernel_name(a);
    if (a.Wp2 && gemm_bf16_rh_cw(a) && (long)((a.M + 127) / 128) * ((a.N + 63) / 64) >= pp_min_tiles()) return "igemm_bf16_rh<w4,126x64,conv>";
    if (a.N <= 32) return "igemm_bf16<w4,128x32,conv>";
    if (a.N <= 64) return ((long)a.M >= 128L * 512) ? "igemm_bf16<w4,128x64,conv>" : "igemm_bf16<w4,64x64,conv>";
    if ((long)((a.M + 127) / 128) * ((a.N + 127) / 128) >= 512) return "igemm_bf16<w4,128x128,conv>";
    return "igemm_bf16<w4,64x64,conv>";
}

// bf16 NHWC conv: A / res / out are bf16, Wp bf16 [N][Kpad] (Kpad % 64 == 0), bias fp32.  Cin % 8 == 0, N % 4 == 0.
static bool bf16_ok(const GemmArgs& a) {
    return a.conv && a.Kpad % BKH == 0 && a.Cin % 8 == 0 && a.N % 4 == 0 && a.act != ACT_GELU && a.ks * a.ks <= 32 &&
           a.M > 0 && a.N > 0;
}

hipError_t launch_gemm_bf16(const GemmArgs& a_in, hipStream_t s);

static void prep_conv_b(GemmArgs& a) {
    a.fd_hw = make_fastdiv((unsigned)(a.Ho * a.Wo));
    a.fd_wo = make_fastdiv((unsigned)a.Wo);
    a.spread = 0ull;
    for (int kh = 0; kh < a.ks && kh * a.ks < 64; ++kh) a.spread |= 1ull << (kh * a.ks);
}

bool gemm_bf16_groupable(const GemmArgs& a) { return bf16_ok(a) && a.omap.G == 1 && (!a.res || a.rmap.G == 1) && !a.rscale && !a.up; }

hipError_t launch_gemm_bf16_group(const GemmArgs* list, int n, hipStream_t s, int* variant) {
    if (variant) *variant = -1;
    if (n <= 0) return hipSuccess;
    if (n > MAXG) return hipErrorInvalidValue;
    {   // the 3x3 stride-1 problems the 2-D halo tile wants (a per-problem rule, igemm_bf16_ws.hip) share one grid of that kernel;
        // whatever else the level holds follows as a second launch on the kernels below
        GemmArgs wsl[MAXG], rest[MAXG];
        int nws = 0, nrest = 0;
        for (int i = 0; i < n; ++i) {
            if (gemm_bf16_ws_wanted(list[i])) wsl[nws++] = list[i];
            else rest[nrest++] = list[i];
        }
        if (nws) {
            const hipError_t e = launch_gemm_bf16_ws_group(wsl, nws, s);
            if (e != hipSuccess) return e;
            if (variant) *variant = 3;
            int v2 = -1;
            return nrest ? launch_gemm_bf16_group(rest, nrest, s, &v2) : hipSuccess;
        }
    }
    if (n == 1) return launch_gemm_bf16(list[0], s);
    static const int BMs[3] = {128, 64, 128}, BNs[3] = {64, 64, 32};
    double total = 0.0;
    for (int i = 0; i < n; ++i) {
        if (!gemm_bf16_groupable(list[i])) return hipErrorInvalidValue;
        total += (double)list[i].M * list[i].N * (list[i].Kpad / BKH) / 4096.0;
    }
    const double per_cu = total / 256.0;
    struct Item { int idx, cfg, tiles; double cost; };
    Item it[MAXG];
    for (int i = 0; i < n; ++i) {
        const GemmArgs& a = list[i];
        const int chunks = a.Kpad / BKH;
        int cfg;
        if (a.N <= 32) cfg = 2;
        else cfg = (chunks * 2.0 * 3.0 <= 0.8 * per_cu && a.M >= 128) ? 0 : 1;
        it[i] = Item{i, cfg, ((a.M + BMs[cfg] - 1) / BMs[cfg]) * ((a.N + BNs[cfg] - 1) / BNs[cfg]),
                     chunks * (BMs[cfg] * BNs[cfg] / 4096.0)};
    }
    // ping-pong launches (>= pp_min_tiles tiles at the tile sizes above): 3x3 stride-1 problems that carry the row-halo
    // weight layout move to that tile (126 x 64 outputs)
    int tiles_small = 0, nrh = 0, lds_halves = (128 + 64) * BKH;
    for (int i = 0; i < n; ++i) tiles_small += (it[i].tiles + 7) & ~7;
    if (tiles_small >= pp_min_tiles())
        for (int i = 0; i < n; ++i) {
            const GemmArgs& a = list[it[i].idx];
            const int cw = a.Wp2 ? gemm_bf16_rh_cw(a) : 0;
            if (!cw) continue;
            const int tn = rh_tn(a.N, cw);
            it[i].cfg = cw == 64 ? (tn == 2 ? 3 : 4) : (cw == 48 ? (tn == 2 ? 5 : (tn == 3 ? 6 : 7)) : (tn == 2 ? 8 : 9));
            it[i].tiles = ((a.M + 125) / 126) * ((a.N + 32 * tn - 1) / (32 * tn));
            it[i].cost = (9.0 * a.Cin / BKH) * tn;
            if (rh_lds_halves(cw, 1, tn) > lds_halves) lds_halves = rh_lds_halves(cw, 1, tn);
            ++nrh;
        }
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && it[j].cost > it[j - 1].cost; --j) { Item t = it[j]; it[j] = it[j - 1]; it[j - 1] = t; }
    GroupArgsB ga;
    ga.n = n;
    int start = 0;
    for (int i = 0; i < n; ++i) {
        GemmArgs a = list[it[i].idx];
        prep_conv_b(a);
        if (it[i].cfg >= 3) { a.Wp = a.Wp2; a.Kpad = 9 * a.Cin; }       // (row-halo tile)
        ga.g[i] = a;
        ga.cfg[i] = it[i].cfg;
        ga.tiles[i] = it[i].tiles;
        ga.start[i] = start;
        start += (it[i].tiles + 7) & ~7;
    }
    ga.start[n] = start;
    for (int i = n; i < MAXG; ++i) { ga.start[i + 1] = start; ga.tiles[i] = 0; ga.cfg[i] = 0; }
    if (variant) *variant = nrh ? 2 : (start >= pp_min_tiles() ? 1 : 0);
    if (nrh) hipLaunchKernelGGL(igemm_bf16_group_rh_kernel, dim3(start), dim3(256), (size_t)lds_halves * 2, s, ga);
    else if (start >= pp_min_tiles()) hipLaunchKernelGGL(igemm_bf16_group_pp_kernel, dim3(start), dim3(256), 0, s, ga);
    else hipLaunchKernelGGL(igemm_bf16_group_kernel, dim3(start), dim3(256), 0, s, ga);
    return hipGetLastError();
}

hipError_t launch_gemm_bf16(const GemmArgs& a_in, hipStream_t s) {
    if (a_in.M <= 0 || a_in.N <= 0) return hipSuccess;
    if (!a_in.conv || a_in.Kpad % BKH != 0 || a_in.Cin % 8 != 0 || a_in.N % 4 != 0 || a_in.act == ACT_GELU ||
        a_in.ks * a_in.ks > 32)   // 32-bit tap masks
        return hipErrorInvalidValue;
    GemmArgs a = a_in;
    a.fd_hw = make_fastdiv((unsigned)(a.Ho * a.Wo));
    a.fd_wo = make_fastdiv((unsigned)a.Wo);
    a.spread = 0ull;
    for (int kh = 0; kh < a.ks && kh * a.ks < 64; ++kh) a.spread |= 1ull << (kh * a.ks);
    if (a.up && !gemm_bf16_upadd_ok(a)) return hipErrorInvalidValue;
    if (gemm_bf16_ws_wanted(a)) return launch_gemm_bf16_ws(a, s);
    const long tiles128 = (long)((a.M + 127) / 128) * ((a.N + 63) / 64);
    if (tiles128 >= pp_min_tiles()) {
        if (a.Wp2 && gemm_bf16_rh_cw(a)) {
            a.Wp = a.Wp2;
            a.Kpad = 9 * a.Cin;
            return launch_gemm_bf16_rh(a, s);
        }
        if (a.N <= 32) return launch_cfg_b<128, 32, 32, 32, 1>(a, s);
        if (a.N <= 64) return ((long)a.M >= 128L * 512) ? launch_cfg_b<128, 64, 64, 32, 1>(a, s) : launch_cfg_b<64, 64, 32, 32, 1>(a, s);
        if ((long)((a.M + 127) / 128) * ((a.N + 127) / 128) >= 512) return launch_cfg_b<128, 128, 64, 64, 1>(a, s);
        return launch_cfg_b<64, 64, 32, 32, 1>(a, s);
    }
    if (a.N <= 32) return launch_cfg_b<128, 32, 32, 32, 3>(a, s);
    if (a.N <= 64) return ((long)a.M >= 128L * 512) ? launch_cfg_b<128, 64, 64, 32, 3>(a, s) : launch_cfg_b<64, 64, 32, 32, 3>(a, s);
    if ((long)((a.M + 127) / 128) * ((a.N + 127) / 128) >= 512) return launch_cfg_b<128, 128, 64, 64, 2>(a, s);
    return launch_cfg_b<64, 64, 32, 32, 3>(a, s);
}

// The lifter's projections on the bf16 MFMA path: out[omap(m) + n] = f(A[m, :K] . W[n, :K] + bias[n]) with A bf16 [M][K]
// contiguous and W bf16 [N][Kpad].  mode 0: fp32 out (+ fp32 residual through rmap); mode 1: GELU, bf16 out [M][N].
template <int BM, int BN, int WM, int WN, int S>
static hipError_t launch_rows_b(const GemmArgs& a, int mode, hipStream_t s) {
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    if (mode == 0) hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WM, WN, S, true, false>), dim3(nbm * nbn), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WM, WN, S, false, true>), dim3(nbm * nbn), dim3(256), 0, s, a);
    return hipGetLastError();
}

static bool rows_b_big(int M, int N) { return (long)((M + 127) / 128) * ((N + 127) / 128) >= 384; }

const char* gemm_bf16_rows_kernel_name(int M, int N) { return rows_b_big(M, N) ? "igemm_bf16<w4,128x128,rows>" : "igemm_bf16<w4,64x64,rows>"; }

hipError_t launch_gemm_bf16_rows(const void* A_bf16, const void* W_bf16, const float* bias, int M, int N, int K, int Kpad,
                                 float* out, RowMap omap, const float* res, RowMap rmap, int gelu_bf16_out, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if (Kpad % BKH != 0 || K % 8 != 0 || N % 4 != 0 || (double)M * K * 2.0 >= 2.0e9) return hipErrorInvalidValue;
    if (gelu_bf16_out && (res || omap.G != 1)) return hipErrorInvalidValue;
    GemmArgs a{};
    a.A = static_cast<const float*>(A_bf16); a.Wp = static_cast<const float*>(W_bf16); a.bias = bias; a.res = res; a.out = out;
    a.M = M; a.N = N; a.K = K; a.Kpad = Kpad;
    a.conv = 1; a.Cin = K; a.H = M; a.W = 1; a.Ho = M; a.Wo = 1; a.ks = 1; a.stride = 1; a.pad = 0;
    a.omap = omap; a.rmap = rmap; a.act = ACT_NONE;
    prep_conv_b(a);
    if (rows_b_big(M, N)) return launch_rows_b<128, 128, 64, 64, 2>(a, gelu_bf16_out, s);
    return launch_rows_b<64, 64, 32, 32, 3>(a, gelu_bf16_out, s);
}

}  // namespace capf

#ifdef CAPF_DIAG
extern "C" __attribute__((visibility("default"))) int capf_debug_bf16_timeline(unsigned long long* dst, int blocks) {
    if (blocks > 8192) blocks = 8192;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(capf::capf_bf16_timeline), (size_t)blocks * 64);
}
#endif
