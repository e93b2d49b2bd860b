// fp32 implicit-GEMM for gfx950 (CDNA4) on v_mfma_f32_32x32x2_f32.
//
//   out[m, n] = act( sum_k A[m, k] * Wp[n, k] + bias[n] + res[m, n] )
//
// Covers every dense contraction on the hot path: the HRNet / CPN 3x3, 1x1 and 7x7 convolutions with
// folded BatchNorm (+ReLU, +residual) — pose_hrnet.py:66-136, networks/resnet.py:58-93 — and the
// lifter's nn.Linear layers (+bias, +GELU, +residual) — pose_dformer.py:15-59.
//
// Design (MI355X-first, not a translation of a warp-32 tiling):
//   * activations are NHWC, weights are pre-packed [N][Kpad] with k = (kh, kw, ci): both MFMA operands
//     are "row-major with K contiguous".  One ds_read_b128 feeds FOUR 32x32x2 MFMAs: lanes 0-31 hold
//     k = kk+j, lanes 32-63 hold k = kk+4+j (the K order inside an MFMA is free as long as A and B agree).
//   * tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip, no ds_write).  The
//     LDS image of such a load is lane-linear (8 rows x 128 B per wave instruction), so the bank-conflict
//     fix is an XOR swizzle applied to the SOURCE k-quad and to the ds_read_b128 address.
//   * addressing is a block-uniform buffer descriptor + a 32-bit per-lane byte offset (+ a scalar tap
//     offset when Cin % 32 == 0).  Every tap's validity (zero padding, ragged M) is one bit of a per-row
//     mask computed ONCE before the K loop; an invalid element gets an out-of-range offset and the
//     hardware writes zeros to LDS without a memory access: the K loop has no branch, no zero page and
//     3 VALU instructions per staged KiB (the 64-bit-pointer version of this loader cost ~12 and was
//     6 % slower end to end).
//   * S-stage LDS ring, counted s_waitcnt vmcnt, ONE raw s_barrier per 32-deep K chunk; the DMA
//     instructions of chunk c+S-1 are issued in the 64-cycle shadows of the MFMAs of chunk c.
//   * 4 wave64 per block share the A/B tiles (fewer L2->LDS bytes per FLOP than private tiles).
//   * blockIdx is remapped so that consecutive tiles (which share halo rows / the same weights) land on
//     the same XCD and hit its private L2.
//   * f32 MFMA is an exact fmaf chain (1/16 of the bf16 rate): results match an fp32 reference to
//     accumulation-order roundoff, which is what the 1e-3 parity bar of BASELINE.json needs.
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"

namespace capf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BK = 32;      // K chunk (floats) = 128 B per tile row

enum { AMODE_ROWS = 0, AMODE_CONV = 1 };

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ long rowmap(const RowMap& r, int m) {
    if (r.G == 1) return (long)m * r.S1 + r.off;
    int q = m / r.G;
    return (long)q * r.S1 + (long)(m - q * r.G) * r.S2 + r.off;
}

// n / d for n < 2^31 with a host-computed (mul, shift): q = (umulhi(n, mul) + n) >> shift
__device__ __forceinline__ int fast_div(int n, FastDiv d) {
    return (int)((__umulhi((unsigned)n, d.mul) + (unsigned)n) >> d.shift);
}

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NW waves per block (1 or 4), block tile BM x BN, wave tile WM x WN, S LDS stages.
// PLAIN: out / res rows are addressed with a plain leading dimension (every conv, most linears);
// otherwise the (G, S1, S2) row maps of the lifter's strided token views are evaluated per row.
// ABL: ablation for diagnosis only (0 = product kernel; 1 = no DMA inside the K loop; 2 = no MFMA)
template <int NW, int BM, int BN, int WM, int WN, int S, int AMODE, bool GELU, bool PLAIN, int ABL = 0>
__global__ __launch_bounds__(64 * NW) void igemm_f32_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type and builtins exist on the device side only
    constexpr int NT = 64 * NW;
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPR = NT / 8;                           // tile rows covered by one DMA round of the block
    constexpr int RA = BM / RPR, RB = BN / RPR;           // DMA instructions per thread per chunk
    constexpr int NLOAD = RA + RB;
    constexpr int STAGE = (BM + BN) * BK;                 // floats per stage
    constexpr int KEYSH = (NW == 4) ? 1 : 0;              // swizzle key = (row >> KEYSH) & 7 (see below)
    static_assert((BM / WM) * (BN / WN) == NW, "wave grid");
    static_assert(BM % RPR == 0 && BN % RPR == 0, "tile rows per DMA round");

    __shared__ __attribute__((aligned(16))) float lds[S * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware tile order: physical block b runs on XCD b % 8; give each XCD a contiguous range of
    // logical tiles (bijective for any grid size).
    const int nblk = gridDim.x;
    int bid;
    {
        const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int nbn = (p.N + BN - 1) / BN;
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // Staging assignment: DMA round i, thread tid -> tile row i*RPR + tid/8, physical 16-byte quad tid%8.
    // The quad holds logical k-quad (tid%8) ^ key(row); key must not depend on i so that a thread walks
    // ONE k sequence: with 32 rows per round key = (row>>1)&7 (conflict-free ds_read_b128), with 8 rows
    // per round key = row&7 (2-way conflict: 8 instead of 4 LDS cycles per read, irrelevant next to
    // 4 x 64-cycle MFMAs per read).
    const int srow = tid >> 3;
    const int kq = (((tid & 7) ^ ((srow >> KEYSH) & 7))) * 4;

    const int nchunks = p.Kpad / BK;
    // split-K (rows mode, weight gradients): grid.y slices the chunk range, each slice writes its own slab
    const int c_begin = blockIdx.y * p.cps;
    const int c_end = min(nchunks, c_begin + p.cps);

    // Both operands are fetched with `buffer_load_dwordx4 ... lds`: a block-uniform resource descriptor
    // (SGPRs) plus a 32-bit per-lane byte offset.  An offset past num_records makes the hardware write
    // ZEROS into LDS without touching memory, so padding taps, rows >= M, k >= K and weight rows >= N
    // need no zero page and no 64-bit per-lane pointers: the per-chunk address work is one select per load.
    //   A, conv mode: base = address of (first row of the tile, tap 0, channel 0) -- row offsets are
    //     monotonic in m, so every real tap of every row of the tile is at a small non-negative offset;
    //     OOB_A is far beyond num_records even after the scalar tap offset is added.
    //   A, rows mode: base = A, absolute 32-bit offsets (the launcher rejects operands >= 4 GiB).
    //   W: base = first weight row of the tile, num_records = bytes up to the end of the packed matrix
    //     (rows >= N and chunks past Kpad fall off the end by themselves).
    constexpr unsigned OOB_A = AMODE == AMODE_CONV ? 0x80000000u : 0xFFFFFFFFu;
    constexpr unsigned NREC_A = AMODE == AMODE_CONV ? 0x7FFFFF00u : 0xFFFFFF00u;
    long a_base = 0;                   // element offset of the descriptor base (block-uniform)
    if (AMODE == AMODE_CONV) {
        const int b = fast_div(m0, p.fd_hw), rem = m0 - b * p.Ho * p.Wo;
        const int ho = fast_div(rem, p.fd_wo), wo = rem - ho * p.Wo;
        a_base = ((long)b * p.H * p.W + (long)(ho * p.stride - p.pad) * p.W + (wo * p.stride - p.pad)) * p.Cin;
    }
    const rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + a_base), 0, NREC_A, 0x00020000);
    const rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wp + (long)n0 * p.Kpad), 0,
                                                            (unsigned)(p.N - n0) * (unsigned)p.Kpad * 4u, 0x00020000);

    unsigned a_rel[RA];                // byte offset of (row, tap 0, channel kq) from the descriptor base
    unsigned a_mask[RA];               // bit t set <=> tap t of this row reads real data (ks <= 5)
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + srow + RPR * i;
        a_rel[i] = 0;
        a_mask[i] = 0u;
        if (m < p.M) {
            if (AMODE == AMODE_ROWS) {
                a_rel[i] = (unsigned)(rowmap(p.amap, m) + kq) * 4u;
                a_mask[i] = 1u;
            } else {
                const int b = fast_div(m, p.fd_hw), rem = m - b * p.Ho * p.Wo;
                const int ho = fast_div(rem, p.fd_wo), wo = rem - ho * p.Wo;
                const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
                const long off = ((long)b * p.H * p.W + (long)h0 * p.W + w0) * p.Cin;
                a_rel[i] = (unsigned)(off - a_base + kq) * 4u;
                // valid taps: kw in [max(0,-w0), min(ks, W-w0)), kh likewise; mask = wbits * spread(hbits)
                // (bit kh*ks+kw; no carries because wbits < 2^ks and spread has its bits ks apart)
                const int kw_lo = max(0, -w0), kw_hi = min(p.ks, p.W - w0);
                const int kh_lo = max(0, -h0), kh_hi = min(p.ks, p.H - h0);
                if (kw_hi > kw_lo && kh_hi > kh_lo) {
                    const unsigned wbits = ((1u << kw_hi) - 1) & ~((1u << kw_lo) - 1);
                    const unsigned below_hi = kh_hi * p.ks >= 32 ? ~0u : ((1u << (kh_hi * p.ks)) - 1);
                    const unsigned below_lo = (1u << (kh_lo * p.ks)) - 1;
                    a_mask[i] = (wbits * (unsigned)p.spread) & below_hi & ~below_lo;   // rows [kh_lo, kh_hi) only
                }
            }
        }
    }
    unsigned w_off[RB];                // running byte offset of this thread's weight quad (advances 128 B / chunk)
#pragma unroll
    for (int i = 0; i < RB; ++i) w_off[i] = (unsigned)((srow + RPR * i) * p.Kpad + c_begin * BK + kq) * 4u;

    // k decomposition of the chunk being prepared.  Cin % 32 == 0 (every HRNet-32 / CPN conv): a chunk
    // lies inside ONE tap, the decomposition is block-uniform (SGPRs) and the tap offset rides in the
    // load's scalar offset.  Otherwise each thread walks its own (tap, ci) sequence.
    const bool uni = AMODE == AMODE_CONV && (p.Cin & (BK - 1)) == 0;
    int u_tap = 0, u_ci = 0, u_kh = 0, u_kw = 0;  // uniform walk (u_ci = first channel of the chunk)
    int tap = 0, ci = kq, kh = 0, kw = 0;         // per-thread walk (ci = this thread's channel)
    if (AMODE == AMODE_CONV && !uni) {
        tap = kq / p.Cin;
        ci = kq - tap * p.Cin;
        kh = tap / p.ks;
        kw = tap - kh * p.ks;
    }

    // offsets of the chunk being staged (computed once per chunk, fired between MFMAs)
    unsigned voff[NLOAD];
    unsigned soff_a = 0;
    auto prepare = [&](int c) {
        if (AMODE == AMODE_ROWS) {
            const int k = c * BK + kq;
            const bool k_ok = k < p.K;
#pragma unroll
            for (int i = 0; i < RA; ++i) voff[i] = (k_ok && a_mask[i]) ? a_rel[i] + (unsigned)(c * BK) * 4u : OOB_A;
        } else if (uni) {
            const unsigned bit = u_tap < 32 ? (1u << u_tap) : 0u;
            soff_a = __builtin_amdgcn_readfirstlane((unsigned)((u_kh * p.W + u_kw) * p.Cin + u_ci) * 4u);
#pragma unroll
            for (int i = 0; i < RA; ++i) voff[i] = (a_mask[i] & bit) ? a_rel[i] : OOB_A;
            u_ci += BK;
            if (u_ci >= p.Cin) {
                u_ci = 0;
                ++u_tap;
                if (++u_kw == p.ks) { u_kw = 0; ++u_kh; }
            }
        } else {
            const unsigned bit = tap < 32 ? (1u << tap) : 0u;
            const unsigned t = (unsigned)((kh * p.W + kw) * p.Cin + ci - kq) * 4u;
#pragma unroll
            for (int i = 0; i < RA; ++i) voff[i] = (a_mask[i] & bit) ? a_rel[i] + t : OOB_A;
            ci += BK;
            while (ci >= p.Cin) {
                ci -= p.Cin;
                ++tap;
                if (++kw == p.ks) { kw = 0; ++kh; }
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            voff[RA + i] = w_off[i];
            w_off[i] += BK * 4u;
        }
    };
    // fire load #idx of the prepared chunk into `stage` (LDS image: 8 rows x 128 B per wave instruction)
    auto fire = [&](int idx, int stage) {
        float* As = lds + stage * STAGE;
        if (idx < RA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(As + (idx * RPR + wave * 8) * BK), 16, voff[idx],
                                                     soff_a, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc_w, (lptr_t)(As + BM * BK + ((idx - RA) * RPR + wave * 8) * BK), 16, voff[idx], 0, 0, 0);
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
    const int fsw = (frow >> KEYSH) & 7;      // swizzle key of this lane's fragment row (tile offsets are multiples of 32)
    const int fhalf = lane >> 5;              // which 4-wide k half of an 8-wide step this lane feeds

    // Pipeline.  Chunks c+1 .. c+S-2 are in flight at the top of iteration c; chunk c+S-1 is fired into
    // the stage that iteration c-1 read, AFTER this iteration's barrier (every wave of the block has then
    // finished reading it), between the MFMAs of the first two k-steps.  Single-wave blocks need no
    // barrier: the wave's own counted vmcnt orders its DMA against its ds_reads.
    constexpr int PER_STEP = (NLOAD + 1) / 2;
#pragma unroll
    for (int s = 0; s < S - 1; ++s) {
        prepare(c_begin + s);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) fire(i, s);
    }
    int st_read = 0, st_fill = S - 1;
    prepare(c_begin + S - 1);    // sources of the chunk fired in iteration 0
    for (int c = c_begin; c < c_end; ++c) {
        wait_vmcnt<(S - 2) * NLOAD>();
        if (NW > 1) __builtin_amdgcn_s_barrier();
        const float* As = lds + st_read * STAGE;
        const float* Bs = As + BM * BK;
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            // the address arithmetic of the NEXT iteration's loads runs in the MFMA shadows of k-step 2
            // (this iteration's loads were all fired in steps 0-1); past the last chunk the offsets are out of
            // range or point at data nobody reads: branch-free, harmless
            if (step == 2) prepare(c + S);
            const int q = ((step * 2) + fhalf) ^ fsw;          // physical quad of logical quad 2*step + half
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(&As[(wm0 + i * 32 + frow) * BK + q * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn0 + j * 32 + frow) * BK + q * 4]);
            int fired = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (ABL != 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);
                        else { acc[i][j][e] += af[i][e] * bf[j][e]; }   // (ablation only)
                        if (step < 2 && fired < PER_STEP) {
                            const int idx = step * PER_STEP + fired;
                            if (idx < NLOAD && ABL != 1) fire(idx, st_fill);
                            ++fired;
                        }
                    }
            if (step < 2) {   // tiles with fewer MFMAs per step than loads: fire the rest here
#pragma unroll
                for (int f = TM * TN * 4; f < PER_STEP; ++f) {
                    const int idx = step * PER_STEP + f;
                    if (idx < NLOAD && ABL != 1) fire(idx, st_fill);
                }
            }
        }
        st_read = (st_read + 1 == S) ? 0 : st_read + 1;
        st_fill = (st_fill + 1 == S) ? 0 : st_fill + 1;
    }
    wait_vmcnt<0>();

    // ---- epilogue.  The MFMAs were issued with the WEIGHTS as the A operand, so the accumulator holds the
    // transposed tile: C/D map of the 32x32 MFMA gives this lane ONE output row m = lane & 31 and, per
    // register group g = r >> 2, FOUR CONSECUTIVE channels n = 8g + 4*(lane>>5) + (r & 3).  NHWC output
    // therefore goes out as 16-byte stores (4 per 32x32 tile instead of 16 dword stores: the store tail of
    // a short-K tile is issue-bound), bias and residual come in as 16-byte loads, and a row's address is
    // computed once per lane.  Residuals are loaded for the whole tile before anything is stored (an
    // in-place residual aliases `out`).
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    const bool vec_ok = (p.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + i * 32 + (lane & 31);
        const bool m_ok = full || m < p.M;
        long o_row = 0, r_row = 0;
        float rs = 1.0f;            // per-row scale of the branch output (DropPath keep mask / keep_prob)
        if (m_ok) {
            if (p.rscale) rs = p.rscale[m / p.rs_div];
            o_row = (PLAIN ? (long)m * p.omap.S1 + p.omap.off : rowmap(p.omap, m)) + (long)blockIdx.y * p.split_stride;
            if (p.res) r_row = PLAIN ? (long)m * p.rmap.S1 + p.rmap.off : rowmap(p.rmap, m);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + wn0 + j * 32 + 4 * (lane >> 5);
            f32x4 rv[4], bv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb + 8 * g;
                const bool ok = m_ok && (full || n < p.N);
                rv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                bv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (vec_ok) {
                    if (ok && p.bias) bv[g] = *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (ok && p.res) rv[g] = *reinterpret_cast<const f32x4*>(p.res + r_row + n);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (m_ok && n + e < p.N) {
                            if (p.bias) bv[g][e] = p.bias[n + e];
                            if (p.res) rv[g][e] = p.res[r_row + n + e];
                        }
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb + 8 * g;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = (acc[i][j][4 * g + e] + bv[g][e]) * rs + rv[g][e];
                    if (GELU) { if (p.act == ACT_GELU) t = gelu_erf(t); }
                    if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
                    v[e] = t;
                }
                if (vec_ok) {
                    if (m_ok && (full || n < p.N)) *reinterpret_cast<f32x4*>(p.out + o_row + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m_ok && n + e < p.N) p.out[o_row + n + e] = v[e];
                }
            }
        }
    }
#endif
}

// =====================================================================================================
// Small-Cin (stem: Cin = 3, k = 3 or 7) variant: K is gathered element-wise into registers, staged with
// ds_write_b128 into a padded LDS tile (pitch 36 floats: conflict-free b128 without a swizzle).  0.3 % of
// the FLOPs of the path; kept simple.
// =====================================================================================================
static constexpr int PITCH = 36;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_f32_smallc_kernel(GemmArgs p) {
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RA = BM / 32, RB = BN / 32;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * PITCH];
    float* As = lds;
    float* Bs = lds + BM * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbn = (p.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / nbn, tile_n = blockIdx.x - tile_m * nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int srow = tid >> 3, kq = (tid & 7) * 4;

    long a_base[RA];
    int a_h0[RA], a_w0[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + srow + 32 * i;
        if (m < p.M) {
            const int b = fast_div(m, p.fd_hw), rem = m - b * p.Ho * p.Wo;
            const int ho = fast_div(rem, p.fd_wo), wo = rem - ho * p.Wo;
            a_base[i] = (long)b * p.H * p.W * p.Cin;
            a_h0[i] = ho * p.stride - p.pad;
            a_w0[i] = wo * p.stride - p.pad;
        } else {
            a_base[i] = 0;
            a_h0[i] = -(1 << 20);
            a_w0[i] = 0;
        }
    }
    f32x4 a_reg[RA], b_reg[RB];
    auto load_chunk = [&](int c) {
        const int k = c * BK + kq;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ke = k + e;
                const int t = ke / p.Cin, c1 = ke - t * p.Cin;
                const int kh = t / p.ks, kw = t - kh * p.ks;
                const int hi = a_h0[i] + kh, wi = a_w0[i] + kw;
                if (ke < p.K && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    v[e] = p.A[a_base[i] + ((long)hi * p.W + wi) * p.Cin + c1];
            }
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = n0 + srow + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < p.N) v = *reinterpret_cast<const f32x4*>(p.Wp + (long)n * p.Kpad + c * BK + kq);
            b_reg[i] = v;
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
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    const int nchunks = p.Kpad / BK;
    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(&As[(srow + 32 * i) * PITCH + kq]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(&Bs[(srow + 32 * i) * PITCH + kq]) = b_reg[i];
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 8) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(&As[(wm0 + i * 32 + frow) * PITCH + kk + fk]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn0 + j * 32 + frow) * PITCH + kk + fk]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const bool n_ok = n < p.N;
        const float bv = (p.bias && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n_ok && m < p.M) {
                    float v = acc[i][j][r] + bv;
                    if (p.res) v += p.res[rowmap(p.rmap, m) + n];
                    if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                    if (p.out_bf16) {
                        unsigned u = __float_as_uint(v);
                        u += 0x7FFFu + ((u >> 16) & 1u);
                        reinterpret_cast<unsigned short*>(p.out)[rowmap(p.omap, m) + n] = (unsigned short)(u >> 16);
                    } else
                    p.out[rowmap(p.omap, m) + n] = v;
                }
            }
        }
    }
}

// =====================================================================================================
// host side: tile selection + launch
// =====================================================================================================
FastDiv make_fastdiv(unsigned d) {
    FastDiv f{0u, 0u};
    if (d <= 1) return f;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.shift = s;
    f.mul = (unsigned)((((1ull << s) - d) << 32) / d + 1);
    return f;
}

enum TileCfg { W4_128x64 = 0, W4_64x64, W4_128x128, W4_256x32, N_TILES };
static const char* kTileNames[N_TILES] = {"w4,128x64", "w4,64x64", "w4,128x128", "w4,256x32"};

// N decides the column tile, M how tall it can be while the grid still fills 256 CUs several times over.
// CAPF_TILE=<index> forces a tile (micro-benchmark tuning only).
static TileCfg pick_tile(const GemmArgs& a) {
    static const int forced = [] { const char* e = getenv("CAPF_TILE"); return e ? atoi(e) : -1; }();
    if (forced >= 0 && forced < N_TILES) return (TileCfg)forced;
    // measured on MI355X (tools/bench_conv.py, batch 64): N <= 32 -> 256x32 (69 TF on 32->32@64^2 vs 45-55 for
    // the others); N <= 64 -> 128x64 for big M (95 TF on 64->64@64^2), 64x64 otherwise; single-wave
    // blocks (private LDS ring, no barrier) lost 15-25 % to their extra L2->LDS traffic and were dropped.
    if (a.N <= 32) return ((long)a.M >= 256L * 512) ? W4_256x32 : W4_64x64;
    if (a.N <= 64) return ((long)a.M >= 128L * 512) ? W4_128x64 : W4_64x64;
    const long tiles128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (tiles128 >= 512) return W4_128x128;
    if ((long)((a.M + 127) / 128) * ((a.N + 63) / 64) >= 256) return W4_128x64;
    return W4_64x64;
}

const char* gemm_f32_kernel_name(const GemmArgs& a) {
    static char buf[N_TILES][2][48];
    static bool init = false;
    if (!init) {
        const char* modes[2] = {"rows", "conv"};
        for (int t = 0; t < N_TILES; ++t)
            for (int m = 0; m < 2; ++m) snprintf(buf[t][m], sizeof(buf[t][m]), "igemm_f32<%s,%s>", kTileNames[t], modes[m]);
        init = true;
    }
    if (a.conv && a.Cin % 4 != 0) return "igemm_f32_smallc<w4,128x64>";
    return buf[pick_tile(a)][a.conv ? 1 : 0];
}

template <int NW, int BM, int BN, int WM, int WN, int S>
static hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    dim3 grid(nbm * nbn, a.splits > 1 ? a.splits : 1), block(64 * NW);
    const bool plain = a.omap.G == 1 && (!a.res || a.rmap.G == 1);
    if (a.conv) {
        if (!plain) return hipErrorInvalidValue;
        static const int abl = [] { const char* e = getenv("CAPF_ABLATE"); return e ? atoi(e) : 0; }();
        if (abl == 1)
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true, 1>), grid, block, 0, s, a);
        else if (abl == 2)
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true, 2>), grid, block, 0, s, a);
        else
        hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true>), grid, block, 0, s, a);
    } else if (a.act == ACT_GELU) {
        if (!plain) return hipErrorInvalidValue;
        hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_ROWS, true, true>), grid, block, 0, s, a);
    } else if (plain) {
        hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_ROWS, false, true>), grid, block, 0, s, a);
    } else {
        hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_ROWS, false, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_gemm_f32(const GemmArgs& a_in, hipStream_t s) {
    if (a_in.M <= 0 || a_in.N <= 0) return hipSuccess;
    if (a_in.Kpad % BK != 0) return hipErrorInvalidValue;
    GemmArgs a = a_in;
    if (a.splits <= 1) { a.splits = 1; a.cps = a.Kpad / BK; a.split_stride = 0; }
    else if (a.conv || a.bias || a.res) return hipErrorInvalidValue;     // split-K slabs are raw partial sums
    if (a.rs_div <= 0) a.rs_div = 1;
    // the epilogue's fast path addresses out / res with 32-bit element offsets from their bases
    if (a.omap.G == 1 && (double)a.M * (double)a.omap.S1 >= 4.0e9) return hipErrorInvalidValue;
    if (a.res && a.rmap.G == 1 && (double)a.M * (double)a.rmap.S1 >= 4.0e9) return hipErrorInvalidValue;
    if (a.conv) {
        a.fd_hw = make_fastdiv((unsigned)(a.Ho * a.Wo));
        a.fd_wo = make_fastdiv((unsigned)a.Wo);
        a.spread = 0ull;
        for (int kh = 0; kh < a.ks && kh * a.ks < 64; ++kh) a.spread |= 1ull << (kh * a.ks);
        if (a.act == ACT_GELU) return hipErrorInvalidValue;
        if (a.Cin % 4 == 0 && a.ks * a.ks > 32) return hipErrorInvalidValue;   // 32-bit tap masks (ks <= 5)
        if (a.Cin % 4 != 0) {
            dim3 grid(((a.M + 127) / 128) * ((a.N + 63) / 64)), block(256);
            hipLaunchKernelGGL((igemm_f32_smallc_kernel<128, 64, 64, 32>), grid, block, 0, s, a);
            return hipGetLastError();
        }
    }
    switch (pick_tile(a)) {
        case W4_128x64: return launch_cfg<4, 128, 64, 64, 32, 2>(a, s);
        case W4_64x64: return launch_cfg<4, 64, 64, 32, 32, 3>(a, s);
        case W4_128x128: return launch_cfg<4, 128, 128, 64, 64, 2>(a, s);
        case W4_256x32: return launch_cfg<4, 256, 32, 64, 32, 2>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace capf
