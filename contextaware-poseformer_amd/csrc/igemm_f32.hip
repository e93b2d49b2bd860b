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
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"

namespace capf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BK = 32;      // K chunk (floats) = 128 B per tile row
static constexpr int LNK = 1280;   // largest K of a LayerNorm'ed-A launch (gamma, beta: 2 * LNK floats of LDS behind the ring)

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

// (diagnosis, CAPF_ABLATE=7) per-block timeline: 8 x u64 per block = {t_entry, t_prologue_done, t_loop_done,
// t_exit, realtime_entry, hw_id, xcc_id, realtime_exit}; read back with capf_debug_timeline()
static constexpr int DBG_BLOCKS = 8192;
__device__ unsigned long long capf_dbg_timeline[DBG_BLOCKS * 8];

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NW waves per block (4), block tile BM x BN, wave tile WM x WN, S LDS stages.
// PLAIN: out / res rows are addressed with a plain leading dimension (every conv, most linears);
// otherwise the (G, S1, S2) row maps of the lifter's strided token views are evaluated per row.
// ABL: ablation for diagnosis only, instantiated by `make DIAG=1` (0 = product kernel; 1 = no DMA inside the K loop;
// 2 = no MFMA; 3 = no LDS fragment reads; 4 = no vmcnt / barrier; 5 = no epilogue; 6 = 1+3+4; 7 = block timeline)
//
// igemm_tile computes ONE output tile (logical tile id `bid`, split-K slice `ky`) with the calling block;
// `lds` is the block's ring (S stages).  It is the body of both the one-problem kernel and the grouped kernel.
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type and builtins exist on the device side only
// LNA (rows mode): the A operand is LayerNorm'ed on the fly — out = act(LN(A) . W^T + bias (+ res)) with
// LN(x) = (x - mean) * rstd * gamma + beta over the K columns of a row (pose_dformer.py:77-78: the norm1 -> qkv and
// norm2 -> fc1 pairs).  Row statistics are computed by the block for its own BM rows before the pipeline starts
// (two passes over rows that are about to be streamed anyway), gamma / beta sit in LDS behind the ring, and the
// normalisation is applied to the A fragments between the LDS read and the MFMA: 12 VALU instructions per
// fragment, no normalised copy of the activations in HBM and no LayerNorm launch.
template <int NW, int BM, int BN, int WM, int WN, int S, int AMODE, bool GELU, bool PLAIN, int ABL, bool LNA = false, bool SPLITK = false>
__device__ __forceinline__ void igemm_tile(const GemmArgs& p, const int bid, const int ky, float* __restrict__ lds,
                                           const int dbg_block) {
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

    unsigned long long dbg_t0 = 0, dbg_r0 = 0, dbg_t1 = 0, dbg_t2 = 0;
    if (ABL == 7) { dbg_t0 = __builtin_amdgcn_s_memtime(); dbg_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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
    const int c_begin = ky * p.cps;
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
    if (SPLITK && AMODE == AMODE_CONV && uni && c_begin > 0) {          // split-K slice: start the walk at chunk c_begin
        u_tap = (c_begin * BK) / p.Cin;
        u_ci = c_begin * BK - u_tap * p.Cin;
        u_kh = u_tap / p.ks;
        u_kw = u_tap - u_kh * p.ks;
    }
    if (AMODE == AMODE_CONV && !uni) {
        tap = (c_begin * BK + kq) / p.Cin;
        ci = c_begin * BK + kq - tap * p.Cin;
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

    // ---- LNA prologue: gamma / beta -> LDS, mean / rstd of the tile's rows -> LDS -> registers
    float* const ln_g = lds + S * STAGE;           // [LNK]
    float* const ln_b = ln_g + LNK;                // [LNK]
    float mu_f[TM], rs_f[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { mu_f[i] = 0.f; rs_f[i] = 1.f; }
    if (LNA) {
        float* const st_mu = ln_b + LNK;           // [BM]
        float* const st_rs = st_mu + BM;           // [BM]
        for (int k = tid * 4; k < p.K; k += NT * 4) {
            *reinterpret_cast<f32x4*>(ln_g + k) = *reinterpret_cast<const f32x4*>(p.ln_g + k);
            *reinterpret_cast<f32x4*>(ln_b + k) = *reinterpret_cast<const f32x4*>(p.ln_b + k);
        }
        constexpr int TPR = NT / BM >= 1 ? NT / BM : 1;          // threads per row (adjacent lanes)
        constexpr int RPP = BM / (NT / TPR);                      // row passes (1 unless BM > NT)
#pragma unroll
        for (int rp = 0; rp < RPP; ++rp) {
            const int r = rp * (NT / TPR) + tid / TPR, part = tid % TPR;
            const int m = m0 + r;
            const bool ok = m < p.M;
            const float* x = p.A + (ok ? rowmap(p.amap, m) : 0);
            float sum = 0.f;
            if (ok)
                for (int k = part * 4; k < p.K; k += TPR * 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(x + k);
                    sum += (v[0] + v[1]) + (v[2] + v[3]);
                }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o, 64);
            const float mean = sum / (float)p.K;
            float sq = 0.f;
            if (ok)
                for (int k = part * 4; k < p.K; k += TPR * 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(x + k);
                    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                    sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) sq += __shfl_xor(sq, o, 64);
            if (part == 0) {
                st_mu[r] = ok ? mean : 0.f;
                st_rs[r] = ok ? 1.0f / sqrtf(sq / (float)p.K + p.ln_eps) : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            mu_f[i] = st_mu[wm0 + i * 32 + frow];
            rs_f[i] = st_rs[wm0 + i * 32 + frow];
        }
    }
    f32x4 gq[2], bq[2];                            // gamma / beta quads of the fragments in af[0], af[1]
    gq[0] = gq[1] = bq[0] = bq[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Pipeline.  Chunks c+1 .. c+S-2 are in flight at the top of iteration c; chunk c+S-1 is fired into
    // the stage that iteration c-1 read, AFTER the barrier that closed iteration c-1 (every wave of the
    // block has then finished reading it), between the MFMAs of the first two k-steps.
    constexpr int PER_STEP = (NLOAD + 1) / 2;
#pragma unroll
    for (int s = 0; s < S - 1; ++s) {
        prepare(c_begin + s);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) fire(i, s);
    }
    int st_read = 0, st_fill = S - 1;
    prepare(c_begin + S - 1);    // sources of the chunk fired in iteration 0

    // Fragment registers are double buffered: the ds_read_b128 of k-step s+1 are issued BEFORE the MFMAs
    // of k-step s, so the ~130-cycle LDS latency hides under 8+ x 64 MFMA cycles instead of stalling the
    // wave four times per chunk.  The step-0 fragments of chunk c+1 are read during step 3 of chunk c,
    // right after the chunk-boundary wait + barrier.
    f32x4 af[2][TM], bf[2][TN];
    auto read_frags = [&](int stage, int step, int buf) {
        const float* As = lds + stage * STAGE;
        const float* Bs = As + BM * BK;
        const int q = ((step * 2) + fhalf) ^ fsw;              // physical quad of logical quad 2*step + half
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[buf][i] = *reinterpret_cast<const f32x4*>(&As[(wm0 + i * 32 + frow) * BK + q * 4]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bf[buf][j] = *reinterpret_cast<const f32x4*>(&Bs[(wn0 + j * 32 + frow) * BK + q * 4]);
    };
    wait_vmcnt<(S - 2) * NLOAD>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0);
    if (LNA) {
        gq[0] = *reinterpret_cast<const f32x4*>(ln_g + c_begin * BK + fhalf * 4);
        bq[0] = *reinterpret_cast<const f32x4*>(ln_b + c_begin * BK + fhalf * 4);
    }
    if (ABL == 7) dbg_t1 = __builtin_amdgcn_s_memtime();
    for (int c = c_begin; c < c_end; ++c) {
        const int st_next = (st_read + 1 == S) ? 0 : st_read + 1;
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            // the address arithmetic of the NEXT iteration's loads runs in the MFMA shadows of k-step 2
            // (this iteration's loads were all fired in steps 0-1); past the last chunk the offsets are out of
            // range or point at data nobody reads: branch-free, harmless
            if (step == 2) prepare(c + S);
            constexpr bool NO_DMA = ABL == 1 || ABL == 6, NO_LDS = ABL == 3 || ABL == 6, NO_SYNC = ABL == 4 || ABL == 6;
            // The memory instructions of this step -- the TM + TN fragment reads of the NEXT k-step (double
            // buffered registers) and, in steps 0-1, this step's share of the chunk's DMA loads -- are spread
            // EVENLY between the step's MFMAs, one per 64-cycle MFMA slot, and pinned there with scheduling
            // barriers: left alone the compiler batches them (3 ds_read + 3 buffer_load in a row), and with one
            // wave per SIMD the matrix pipe then idles behind their issue cycles.
            const int rd_stage = step < 3 ? st_read : st_next;
            const int rd_step = step < 3 ? step + 1 : 0, rd_buf = (step + 1) & 1;
            const float* rAs = lds + rd_stage * STAGE;
            const float* rBs = rAs + BM * BK;
            const int rq = ((rd_step * 2) + fhalf) ^ fsw;
            constexpr int NM = 4 * TM * TN;                                   // MFMAs of a step
            constexpr int NLN = LNA ? 2 : 0;                                  // gamma / beta quads of the next fragments
            const int n_mem = (TM + TN + NLN) + (step < 2 ? PER_STEP : 0);
            const int ln_k = ((step < 3 ? c : c + 1) * BK) + (rd_step * 2 + fhalf) * 4;   // k of the fragments being read
            auto mem_op = [&](int k) {
                if (k < TM) {
                    if (!NO_LDS) af[rd_buf][k] = *reinterpret_cast<const f32x4*>(&rAs[(wm0 + k * 32 + frow) * BK + rq * 4]);
                } else if (k < TM + TN) {
                    if (!NO_LDS) bf[rd_buf][k - TM] = *reinterpret_cast<const f32x4*>(&rBs[(wn0 + (k - TM) * 32 + frow) * BK + rq * 4]);
                } else if (k < TM + TN + NLN) {
                    if (k == TM + TN) gq[rd_buf] = *reinterpret_cast<const f32x4*>(ln_g + ln_k);
                    else bq[rd_buf] = *reinterpret_cast<const f32x4*>(ln_b + ln_k);
                } else {
                    const int idx = step * PER_STEP + (k - TM - TN - NLN);
                    if (idx < NLOAD && !NO_DMA) fire(idx, st_fill);
                }
            };
            // Chunk boundary (k-step 3): chunk c+1 has landed (all but the S-2 youngest chunks) and every wave's
            // fragment reads of chunk c have returned (lgkmcnt) -> after the barrier, stage st_read may be
            // overwritten by the loads fired in iteration c+1.  It sits after the first half of the step's MFMAs
            // (more time for the DMA to land); the next chunk's first fragments are read in the slots after it.
            constexpr int BSLOT = NM / 2;
            const int fb = step & 1;
            if (LNA) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) af[fb][i][e] = ((af[fb][i][e] - mu_f[i]) * rs_f[i]) * gq[fb][e] + bq[fb][e];
            }
            int issued = 0, slot = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (ABL != 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[fb][j][e], af[fb][i][e], acc[i][j], 0, 0, 0);
                        else { acc[i][j][e] += af[fb][i][e] * bf[fb][j][e]; }   // (ablation only)
                        ++slot;
                        if (step == 3 && slot == BSLOT && !NO_SYNC) {
                            wait_vmcnt<(S - 2) * NLOAD>();
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                        }
                        // after MFMA #slot: the next memory op -- fragment reads first (they are needed at the next
                        // k-step), then the DMA loads; the last slot takes whatever is left
                        bool took = false;                     // one per slot (the last slot takes the rest)
#pragma unroll
                        for (int k = 0; k < TM + TN + NLN + PER_STEP; ++k)
                            if (k == issued && k < n_mem && ((!took && (step < 3 || slot >= BSLOT)) || slot == NM)) {
                                mem_op(k);
                                ++issued;
                                took = true;
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
            for (int k = 0; k < TM + TN + NLN + PER_STEP; ++k)      // (never more memory ops than slots x their share; safety)
                if (k >= issued && k < n_mem) mem_op(k);
        }
        st_read = st_next;
        st_fill = (st_fill + 1 == S) ? 0 : st_fill + 1;
    }
    wait_vmcnt<0>();
    if (ABL == 7) dbg_t2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue.  The MFMAs were issued with the WEIGHTS as the A operand, so the accumulator holds the
    // transposed tile: C/D map of the 32x32 MFMA gives this lane ONE output row m = lane & 31 and, per
    // register group g = r >> 2, FOUR CONSECUTIVE channels n = 8g + 4*(lane>>5) + (r & 3).  NHWC output
    // therefore goes out as 16-byte stores (4 per 32x32 tile instead of 16 dword stores: the store tail of
    // a short-K tile is issue-bound), bias and residual come in as 16-byte loads, and a row's address is
    // computed once per lane.
    if (ABL == 5 && p.M > 0) return;     // (ablation) no epilogue; the condition is opaque to the compiler, the MFMAs stay
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    const bool vec_ok = (p.N & 3) == 0;
    if (SPLITK && PLAIN && p.split_cnt) {
        // ---- split-K with an in-launch reduction (small-batch convs; rows-mode GEMMs of the batch-1 lifter): park the raw partial tile,
        // count, and let the last slice of the tile reduce + finish
        float* slab = p.split_ws + (long)ky * p.split_stride;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + i * 32 + (lane & 31);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                    if (m < p.M && n < p.N)
                        *reinterpret_cast<f32x4*>(slab + (long)m * p.N + n) =
                            f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                }
        }
        // Hand-off (cdna_hip_programming.md, in-launch split-K reduction): ONE agent-scope release per slice and ONE agent-scope acquire per
        // tile, both by lane 0 behind a block barrier -- not a __threadfence() (L2 write-back + invalidate) in every thread of every
        // slice, which cost ~15 us per launch.  Order matters: every wave drains its stores, barrier, release fence, drained again (the
        // compiler may drop the wait behind the write-back), THEN the ticket.  The "I am last" flag lives in the ring (its last stage
        // was read before the barrier): a second __shared__ object would make the compiler drain the DMA pipeline at every k-step.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* s_last = reinterpret_cast<int*>(lds);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ticket = __hip_atomic_fetch_add(p.split_cnt + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == p.splits - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (drops this CU's L1: the other slices' slabs are read from L2)
                p.split_cnt[bid] = 0;                 // self-resetting: the next launch finds zeros
            }
            *s_last = last;
        }
        __syncthreads();
        if (!*s_last) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int sl = 0; sl < p.splits; ++sl) {       // fixed order: the sum does not depend on which slice came last
            const float* src = p.split_ws + (long)sl * p.split_stride;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm0 + i * 32 + (lane & 31);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                        if (m < p.M && n < p.N) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)m * p.N + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += v[e];
                        }
                    }
            }
        }
    }
    // Pass 1: EVERY bias / residual load of the tile is issued before the first store.  vmcnt retires in
    // order, so a load issued after a store would make its consumer wait for that store's completion too
    // (~1.5 us per 32x32 sub-tile when loads and stores alternate); it is also what an in-place residual
    // (res aliases out) needs.
    long o_row[TM], r_row[TM];
    bool m_ok[TM];
    float rs[TM];                     // per-row scale of the branch output (DropPath keep mask / keep_prob)
    f32x4 bv[TN][4], rv[TM][TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
            bv[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                if (vec_ok) {
                    if (full || n < p.N) bv[j][g] = *reinterpret_cast<const f32x4*>(p.bias + n);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) bv[j][g][e] = p.bias[n + e];
                }
            }
        }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + i * 32 + (lane & 31);
        m_ok[i] = full || m < p.M;
        o_row[i] = 0; r_row[i] = 0; rs[i] = 1.0f;
        if (m_ok[i]) {
            if (p.rscale) rs[i] = p.rscale[m / p.rs_div];
            o_row[i] = (PLAIN ? (long)m * p.omap.S1 + p.omap.off : rowmap(p.omap, m)) + (p.split_cnt ? 0 : (long)ky * p.split_stride);
            if (p.res) r_row[i] = PLAIN ? (long)m * p.rmap.S1 + p.rmap.off : rowmap(p.rmap, m);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                rv[i][j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.res && m_ok[i]) {
                    if (vec_ok) {
                        if (full || n < p.N) rv[i][j][g] = *reinterpret_cast<const f32x4*>(p.res + r_row[i] + n);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) rv[i][j][g][e] = p.res[r_row[i] + n + e];
                    }
                }
            }
    }
    // Pass 2: combine and store.
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + j * 32 + 4 * (lane >> 5) + 8 * g;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = (acc[i][j][4 * g + e] + bv[j][g][e]) * rs[i] + rv[i][j][g][e];
                    if (GELU) { if (p.act == ACT_GELU) t = gelu_erf(t); }
                    if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
                    v[e] = t;
                }
                if (vec_ok) {
                    if (m_ok[i] && (full || n < p.N)) *reinterpret_cast<f32x4*>(p.out + o_row[i] + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m_ok[i] && n + e < p.N) p.out[o_row[i] + n + e] = v[e];
                }
            }
    if (ABL == 7 && tid == 0 && dbg_block < DBG_BLOCKS && ky == 0) {
        const unsigned long long t3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();   // stores still in flight
        unsigned long long* d = capf_dbg_timeline + (size_t)dbg_block * 8;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[3] = t3;
        d[4] = dbg_r0; d[5] = hw; d[6] = xcc; d[7] = r3;
    }
}
#endif

// XCD-aware tile order: physical block b runs on XCD b % 8; give each XCD a contiguous range of logical
// tiles (bijective for any grid size).
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <int NW, int BM, int BN, int WM, int WN, int S, int AMODE, bool GELU, bool PLAIN, int ABL = 0, bool LNA = false>
__global__ __launch_bounds__(64 * NW) void igemm_f32_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float lds[S * (BM + BN) * BK + (LNA ? 2 * LNK + 2 * BM : 0)];
    igemm_tile<NW, BM, BN, WM, WN, S, AMODE, GELU, PLAIN, ABL, LNA>(p, xcd_remap(blockIdx.x, gridDim.x), blockIdx.y, lds,
                                                                     blockIdx.x);
#endif
}

// Rows-mode GEMM split along K with the in-launch reduction of igemm_tile (batch 1-2: the joint blocks' 17- / 34-row GEMMs are 10-30 tiles
// with a 20- / 40-chunk K loop each -- pose_dformer.py:15-59 at B p (l c) -- on 256 CUs): block b = slice b % splits of tile b / splits.
template <bool GELU>
__global__ __launch_bounds__(256) void igemm_f32_rows_splitk_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float lds[3 * (64 + 64) * BK];
    const int tile = blockIdx.x / p.splits, ky = blockIdx.x - tile * p.splits;
    igemm_tile<4, 64, 64, 32, 32, 3, AMODE_ROWS, GELU, true, 0, false, true>(p, tile, ky, lds, blockIdx.x);
#endif
}

// =====================================================================================================
// Grouped launch: up to MAXG independent convolutions (the branches of an HRNet module at the same depth,
// the 1x1 / stride-2 convs of a fuse layer) share ONE grid.  A kernel per conv leaves the chip idle while
// its last tiles drain and the next kernel's first tiles wait for their first loads (5-10 us out of ~50),
// and separate streams overlap almost nothing because each kernel fills every CU; inside one grid the
// block scheduler back-fills every freed slot at once and the problems' prologue / epilogue bursts
// interleave.  Problems are ordered longest-K first (the 256-channel 8x8 branch has 8x the K loop of the
// 32-channel 64x64 one); each problem's tile range is padded to a multiple of 8 blocks so that the
// XCD-contiguous tile order of the single-problem kernel holds per problem.
struct GroupArgs {
    GemmArgs g[MAXG];
    int start[MAXG + 1];   // first physical block of problem i (multiples of 8)
    int tiles[MAXG];       // real tiles of problem i
    int cfg[MAXG];         // 0: 128x64 (S=2), 1: 64x64 (S=3), 2: 128x32 (S=2)
    int splits[MAXG];      // split-K slices per tile (1 = none); tiles[] counts blocks = tiles x splits
    int n;
};
static constexpr int GROUP_LDS_FLOATS = 2 * (128 + 64) * BK;   // 48 KiB: 3 blocks per CU for every configuration
static_assert(3 * (64 + 64) * BK <= GROUP_LDS_FLOATS && 2 * (128 + 32) * BK <= GROUP_LDS_FLOATS, "group LDS");

// SPLITK: the variant with the conv split-K path compiled in (small batches only: the extra code costs the plain kernel
// 2-40 % through the instruction cache, so it is a separate instantiation)
template <int ABL, bool SPLITK = false>
__global__ __launch_bounds__(256) void igemm_f32_group_kernel(GroupArgs ga) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float lds[GROUP_LDS_FLOATS];
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < ga.n && b >= ga.start[pi + 1]) ++pi;          // block-uniform
    const int l = b - ga.start[pi];
    const int per_xcd = (ga.start[pi + 1] - ga.start[pi]) >> 3;
    const int bid = (l & 7) * per_xcd + (l >> 3);                 // XCD-contiguous tile order inside the problem
    if (bid >= ga.tiles[pi]) return;                              // padding block
    const GemmArgs& p = ga.g[pi];
    const int sp = SPLITK ? ga.splits[pi] : 1;
    const int tile = sp > 1 ? bid / sp : bid, ky = sp > 1 ? bid - tile * sp : 0;
    switch (ga.cfg[pi]) {
        case 0: igemm_tile<4, 128, 64, 64, 32, 2, AMODE_CONV, false, true, ABL, false, SPLITK>(p, tile, ky, lds, b); break;
        case 1: igemm_tile<4, 64, 64, 32, 32, 3, AMODE_CONV, false, true, ABL, false, SPLITK>(p, tile, ky, lds, b); break;
        default: igemm_tile<4, 128, 32, 32, 32, 2, AMODE_CONV, false, true, ABL, false, SPLITK>(p, tile, ky, lds, b); break;
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
                        reinterpret_cast<unsigned short*>(p.out)[rowmap(p.omap, m) + n] = to_bf16(v);
                    } else
                    p.out[rowmap(p.omap, m) + n] = v;
                }
            }
        }
    }
}

// =====================================================================================================
// The 3x3 / stride-2 fp32 stem as a persistent streaming kernel -- the fp32 twin of igemm_bf16_stem_stream_kernel
// (igemm_bf16.hip, where the scheme is described): weights staged once per block, the next 64-pixel tile's runs (for a fixed
// (pixel, kh) the 9 input floats are contiguous in the NHWC image: two 16-byte loads + one dword, raw buffer loads with
// out-of-range offsets instead of branches) in flight across the MFMAs and stores of the current tile, LDS-transposed 16-byte
// stores (128 B contiguous per row), XCD-contiguous tile walk.  K in LDS = (kh, 12): 9 values + 3 zeros per kh, 40 staged
// floats per row.  The element-wise kernel above moved 2.0 TB/s at batch 64 (155 us for 268 MB out + 50 MB in).
// =====================================================================================================
template <int KS>
__global__ __launch_bounds__(256, 2) void igemm_f32_stem_stream_kernel(GemmArgs p, int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 64, BN = 64;
    constexpr int RUN = KS * 3, NX4 = RUN / 4;      // 9 floats = 2 quads + 1 dword
    static_assert(RUN % 4 == 1, "one dword behind the quads (a 16-byte load with dead upper dwords costs a vmcnt(0))");
    constexpr int PKH = (RUN + 3) / 4 * 4;          // floats per kh in LDS (12)
    constexpr int KP = (KS * PKH + 7) / 8 * 8;      // staged K (40)
    constexpr int SP = KP + 4;                      // floats per LDS row: conflict-free b128 reads
    constexpr int EPS = 36;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(BM * KS <= 256, "one run per thread");
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * SP + 4 * 32 * EPS];
    float* As = lds;
    float* Bs = lds + BM * SP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* ep = lds + (BM + BN) * SP + wave * (32 * EPS);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const long total = (long)(p.M / (p.Ho * p.Wo)) * p.H * p.W * 3;              // floats of the image tensor (< 2^29: launcher)
    const rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)(total * 4), 0x00020000);
    const rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)p.bias : (void*)p.out, 0,
                                                             p.bias ? (unsigned)p.N * 4u : 0u, 0x00020000);
    auto ldq = [&](rsrc_t r, unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); };
    auto ldd = [&](rsrc_t r, unsigned off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); };
    auto put_run = [&](float* dst, const f32x4 (&q)[NX4], float last, unsigned mask, int kh) {
        auto val = [&](int e) { return e < NX4 * 4 ? q[e >> 2][e & 3] : (e == NX4 * 4 ? last : 0.f); };
#pragma unroll
        for (int g = 0; g < PKH / 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (4 * g + e < RUN && ((mask >> (4 * g + e)) & 1u)) ? val(4 * g + e) : 0.f;
            *reinterpret_cast<f32x4*>(dst + kh * PKH + 4 * g) = v;
        }
        if (KP > KS * PKH && kh == KS - 1) *reinterpret_cast<f32x4*>(dst + KS * PKH) = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    {   // weights, once per block: runs of the fp32 pack [N][Kpad], K order (kh, kw, c)
        const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (unsigned)p.N * (unsigned)p.Kpad * 4u, 0x00020000);
        for (int t = tid; t < BN * KS; t += 256) {
            const int kh = t / BN, n = t - kh * BN;
            const unsigned wo = n < p.N ? (unsigned)(n * p.Kpad + kh * RUN) * 4u : OOB;
            f32x4 q[NX4];
#pragma unroll
            for (int x = 0; x < NX4; ++x) q[x] = ldq(rs_w, n < p.N ? wo + 16u * x : OOB);
            const float last = ldd(rs_w, n < p.N ? wo + 16u * NX4 : OOB);
            put_run(Bs + n * SP, q, last, (1u << RUN) - 1u, kh);
        }
    }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_end = min(ntiles, (xcd + 1) * per_xcd);
    int tile = xcd * per_xcd + slot;

    f32x4 q[NX4];
    float qlast;
    unsigned mask;
    const int t_run = tid % (BM * KS);              // (threads beyond the 192 runs repeat one: same loads, same LDS bytes)
    const int r_kh = t_run / BM, r_row = t_run - r_kh * BM;
    auto issue = [&](int tl) {
        const int m = tl * BM + r_row;
        bool ok = tl < t_end && m < p.M;
        const int mm = ok ? m : 0;
        const int b = fast_div(mm, p.fd_hw), rem = mm - b * p.Ho * p.Wo;
        const int ho = fast_div(rem, p.fd_wo), wo = rem - ho * p.Wo;
        const int hi = ho * p.stride - p.pad + r_kh, wi0 = wo * p.stride - p.pad;
        ok = ok && (unsigned)hi < (unsigned)p.H;
        const int o = ((b * p.H + hi) * p.W + wi0) * 3;
        const int lo = max(0, -wi0), hn = max(lo, min(KS, p.W - wi0));
        mask = ok ? (((1u << (3 * hn)) - 1u) & ~((1u << (3 * lo)) - 1u)) : 0u;
#pragma unroll
        for (int x = 0; x < NX4; ++x) q[x] = ldq(rs_in, (ok && o + 4 * x >= 0) ? (unsigned)(o + 4 * x) * 4u : OOB);
        qlast = ldd(rs_in, (ok && o + 4 * NX4 >= 0) ? (unsigned)(o + 4 * NX4) * 4u : OOB);
    };
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int frow = lane & 31, fhalf = lane >> 5, fk = fhalf * 4;
    const int er = lane >> 3, ec = (lane & 7) * 4;          // epilogue: 8 lanes x 16 B = one 128-B row of the wave's 32 channels
    const f32x4 bv = ldq(rs_bias, (unsigned)(wn0 + ec) * 4u);

    issue(tile);
    {   // four dropped stores: the loop is entered with the same "loads, then four stores" in flight as its back edge leaves
        const rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0u, 0x00020000);
#pragma unroll
        for (int k = 0; k < 4; ++k)                 // (distinct offsets: identical stores would be merged into one)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rs_none, OOB + 16u * k, 0, 0);
    }
    for (; tile < t_end; tile += nslots) {
        const int m0 = tile * BM;
        put_run(As + r_row * SP, q, qlast, mask, r_kh);
        const int ho_max = p.pad / p.stride;
        if (m0 < (ho_max + 1) * p.Wo) {             // frame 0, input row 0, window left of the image: negative tensor offsets
            __syncthreads();
            const int nw = min(p.Wo, (p.pad + p.stride - 1) / p.stride);
            for (int idx = tid; idx < (ho_max + 1) * nw * PKH; idx += 256) {
                const int e = idx % PKH, r = idx / PKH, wo = r % nw, ho = r / nw;
                const int kh = p.pad - ho * p.stride, m = ho * p.Wo + wo;
                if (kh >= 0 && kh < KS && m >= m0 && m < m0 + BM) {
                    const int wi = wo * p.stride - p.pad + e / 3;
                    float v = 0.f;
                    if (e < RUN && (unsigned)wi < (unsigned)p.W) v = p.A[wi * 3 + e % 3];
                    As[(m - m0) * SP + kh * PKH + e] = v;
                }
            }
        }
        __syncthreads();
        issue(tile + nslots);

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KP; kk += 8) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(&As[(wm0 + frow) * SP + kk + fk]);
            const f32x4 bf = *reinterpret_cast<const f32x4*>(&Bs[(wn0 + frow) * SP + kk + fk]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[e], af[e], acc, 0, 0, 0);
        }
        // transposed accumulator: lane = row m (lane & 31), register 4 g + e = channel 8 g + 4 fhalf + e
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(&ep[frow * EPS + 8 * g + 4 * fhalf]) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        __builtin_amdgcn_wave_barrier();
        const rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (long)m0 * p.omap.S1 + p.omap.off), 0, 0x7FFFFF00u, 0x00020000);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int row = h * 8 + er, ml = wm0 + row, nl = wn0 + ec;
            f32x4 x = *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] += bv[e];
                if (p.act == ACT_RELU) x[e] = fmaxf(x[e], 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), rs_out,
                                                   (m0 + ml < p.M && nl < p.N) ? (unsigned)(ml * (int)p.omap.S1 + nl) * 4u : OOB, 0, 0);
        }
        __syncthreads();
    }
#endif
}

static bool stem_stream_f32_ok(const GemmArgs& a) {
    static const int on = [] { const char* e = diag_env("CAPF_STEM_STREAM"); return e ? atoi(e) : 1; }();        // A/B runs only
    const long total = (long)(a.M / (a.Ho * a.Wo)) * a.H * a.W * 3;
    return on && a.conv && a.Cin == 3 && a.ks == 3 && a.pad == 1 && a.K == 27 && a.Kpad >= 2 * 9 + 12 && !a.res && !a.out_bf16 &&
           !a.rscale && a.act != ACT_GELU && a.omap.G == 1 && a.rmap.G <= 1 && a.N <= 64 && a.N % 4 == 0 && a.omap.S1 % 4 == 0 &&
           a.omap.off % 4 == 0 && total < (1L << 29) && a.Wo >= 4;
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

// Tile choice by a small cost model fitted to the per-block timelines (tools/timeline.py), in units of one
// 64x64x32 tile-chunk (1024 MFMA cycles of a CU): a CU gets tpc = ceil(tiles / 256) tiles, up to `occ` of them
// co-resident; its K loops run at 78 / 83 / 88 % of the MFMA rate with 1 / 2 / 3 co-resident blocks, and every
// round of co-resident blocks pays ~14 units of exposed first-load wait + store drain.  What this captures
// that "biggest tile that still gives >= 256 blocks" does not: 270 tiles of 128x64 take TWO rounds on 256 CUs
// (the lifter's M = 17 * 64 GEMMs), 510 tiles of 64x64 take one.  CAPF_TILE=<index> forces a tile (tuning only).
static TileCfg pick_tile(const GemmArgs& a) {
    static const int forced = [] { const char* e = diag_env("CAPF_TILE"); return e ? atoi(e) : -1; }();
    if (forced >= 0 && forced < N_TILES) return (TileCfg)forced;
    struct Cand { TileCfg cfg; int bm, bn, unit, occ; };
    static const Cand cands[4] = {{W4_256x32, 256, 32, 2, 2}, {W4_128x128, 128, 128, 4, 2}, {W4_128x64, 128, 64, 2, 3},
                                  {W4_64x64, 64, 64, 1, 3}};
    const int chunks = a.splits > 1 ? a.cps : a.Kpad / BK;
    const int ksplit = a.splits > 1 ? a.splits : 1;
    double best = 0.0;
    TileCfg pick = W4_64x64;
    for (const Cand& c : cands) {
        if (c.cfg == W4_256x32 && a.N > 32) continue;          // tall tile: only for one narrow column tile
        if (c.cfg == W4_128x128 && a.N <= 64) continue;
        if (c.cfg == W4_128x64 && a.N <= 32) continue;         // would multiply zero columns half of the time
        const long tiles = (long)((a.M + c.bm - 1) / c.bm) * ((a.N + c.bn - 1) / c.bn) * ksplit;
        const long tpc = (tiles + 255) / 256;
        const long res = tpc < c.occ ? tpc : c.occ;
        const double eff = res >= 3 ? 0.88 : (res == 2 ? 0.83 : 0.78);
        const double cost = (double)tpc * c.unit * chunks / eff + 14.0 * (double)((tpc + c.occ - 1) / c.occ);
        if (best == 0.0 || cost < best) { best = cost; pick = c.cfg; }
    }
    return pick;
}

// compute_dtype = bf16 (the op writes bf16): the stem runs on the bf16 MFMA too (igemm_bf16.hip); CAPF_STEM_BF16=0 keeps it fp32
static bool stem_on_bf16(const GemmArgs& a) {
    static const int on = [] { const char* e = diag_env("CAPF_STEM_BF16"); return e ? atoi(e) : 1; }();
    return on && gemm_bf16_smallc_ok(a);
}

// The pointwise kernel or the two-fp16-piece GEMM for a 1x1 conv that both take?  A function of the conv alone (every schedule must pick
// the same kernel): the pointwise kernel keeps the channel EXPANSIONS (N >= 4 K: layer1's 64 -> 256, bound by their writes -- 2.6-3.1 TB/s on
// either kernel, tools/bench_pw_h2.py), everything else with a two-piece pack leaves the fp32 pipe (256 -> 64: 106 -> 75 us at batch 64,
// 687 -> 524 at 512; the fuse layers' 1x1 convs at batch 512, which as a GROUP never reached the pointwise kernel anyway)
static bool pw_preferred(const GemmArgs& a) { return gemm_f32_pw_ok(a) && (!gemm_f32h2g_ok(a) || a.N >= 4 * a.K); }
bool gemm_f32_on_h2g(const GemmArgs& a) { return gemm_f32h2g_ok(a) && !pw_preferred(a); }

const char* gemm_f32_kernel_name(const GemmArgs& a) {
    static char buf[N_TILES][2][48];
    static bool init = false;
    if (!init) {
        const char* modes[2] = {"rows", "conv"};
        for (int t = 0; t < N_TILES; ++t)
            for (int m = 0; m < 2; ++m) snprintf(buf[t][m], sizeof(buf[t][m]), "igemm_f32<%s,%s>", kTileNames[t], modes[m]);
        init = true;
    }
    if (a.conv && a.Cin % 4 != 0)
        return stem_on_bf16(a) ? gemm_bf16_smallc_kernel_name(a) : (stem_stream_f32_ok(a) ? "igemm_f32_stem_stream<w4,64x64>" : "igemm_f32_smallc<w4,128x64>");
    if (pw_preferred(a)) return gemm_f32_pw_kernel_name();
    if (gemm_f32h2g_ok(a)) return gemm_f32h2g_kernel_name(a, false);
    if (gemm_f32_rows_splitk(a)) return "igemm_f32_rows_splitk";
    return buf[pick_tile(a)][a.conv ? 1 : 0];
}

template <int NW, int BM, int BN, int WM, int WN, int S>
static hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    dim3 grid(nbm * nbn, a.splits > 1 ? a.splits : 1), block(64 * NW);
    const bool plain = a.omap.G == 1 && (!a.res || a.rmap.G == 1);
    if (a.conv) {
        if (!plain) return hipErrorInvalidValue;
#ifdef CAPF_DIAG      // ablation / timeline instantiations (make DIAG=1; tools/ablate.sh, tools/timeline.py)
        static const int abl = [] { const char* e = diag_env("CAPF_ABLATE"); return e ? atoi(e) : 0; }();
        if (abl == 1)
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true, 1>), grid, block, 0, s, a);
        else if (abl == 3)
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true, 3>), grid, block, 0, s, a);
        else if (abl == 4)
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true, 4>), grid, block, 0, s, a);
        else if (abl == 5)
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true, 5>), grid, block, 0, s, a);
        else if (abl == 6)
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true, 6>), grid, block, 0, s, a);
        else if (abl == 7)
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true, 7>), grid, block, 0, s, a);
        else if (abl == 2)
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true, 2>), grid, block, 0, s, a);
        else
#endif
        hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_CONV, false, true>), grid, block, 0, s, a);
    } else if (a.ln_g) {
        // LayerNorm'ed A operand: plain output, K columns == the normalised width, no split-K
        if (!plain || a.Kpad != a.K || a.K > LNK || (a.K & 3) || a.splits > 1 || !a.ln_b) return hipErrorInvalidValue;
        if (a.act == ACT_GELU)
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_ROWS, true, true, 0, true>), grid, block, 0, s, a);
        else
            hipLaunchKernelGGL((igemm_f32_kernel<NW, BM, BN, WM, WN, S, AMODE_ROWS, false, true, 0, true>), grid, block, 0, s, a);
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

hipError_t launch_gemm_f32(const GemmArgs& a_in, hipStream_t s);

// conv mode: index-math constants; false if the shape is outside what the kernels address
static bool prep_conv(GemmArgs& a) {
    a.fd_hw = make_fastdiv((unsigned)(a.Ho * a.Wo));
    a.fd_wo = make_fastdiv((unsigned)a.Wo);
    a.spread = 0ull;
    for (int kh = 0; kh < a.ks && kh * a.ks < 64; ++kh) a.spread |= 1ull << (kh * a.ks);
    if (a.act == ACT_GELU) return false;
    if (a.Cin % 4 == 0 && a.ks * a.ks > 32) return false;   // 32-bit tap masks (ks <= 5)
    return true;
}

bool gemm_f32_groupable(const GemmArgs& a) {
    return a.conv && a.Cin % 4 == 0 && a.ks * a.ks <= 32 && a.splits <= 1 && !a.rscale && a.act != ACT_GELU &&
           a.omap.G == 1 && (!a.res || a.rmap.G == 1) && a.M > 0 && a.N > 0 && a.Kpad % BK == 0 && !a.out_bf16;
}

// tiles of a conv problem under the grouped kernel's tile choice at small sizes (64x64, or 128x32 for N <= 32)
static long group_tiles_small(const GemmArgs& a) {
    return a.N <= 32 ? (long)((a.M + 127) / 128) : (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
}
// does a lone conv gain from the split-K path of the grouped kernel? (few tiles, long K, scratch available)
static bool worth_splitting(const GemmArgs& a) {
    // (a.N & 3) == 0: the slab stores / reloads are 16-byte accesses at slab + m * N + n
    return a.conv && a.split_ws && a.split_cnt && gemm_f32_groupable(a) && (a.N & 3) == 0 && a.Kpad / BK >= 16 && group_tiles_small(a) <= 16 &&
           (long)a.M * a.N * 2 <= a.split_ws_elems;
}

// the same question for a rows-mode GEMM (the lifter's projections at batch 1-2): at most 32 tiles of 64 x 64, at least 16 chunks, plain row
// pitches, no LayerNorm fold, scratch lent by the caller.  Measured (ms per forward at batch 1 / 2 / 4): without 2.26 / 2.37 / 2.92; this
// rule 2.17 / 2.26 / 2.88; up to 64 tiles 2.22 / 2.26 / 2.89; from 8 chunks in slices of 4 2.20 / 2.29 / 2.88
bool gemm_f32_rows_splitk(const GemmArgs& a) {
    if (a.conv || a.ln_g || a.splits > 1 || !a.split_ws || !a.split_cnt || a.rscale || (a.N & 3) || a.Kpad % BK != 0) return false;
    if (a.amap.G != 1 || a.omap.G != 1 || (a.res && a.rmap.G != 1)) return false;
    const long tiles = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
    const int chunks = a.Kpad / BK;
    if (tiles > 32 || chunks < 16) return false;
    const int sp = std::min(8, chunks / 6);
    return sp > 1 && (long)a.M * a.N * sp <= a.split_ws_elems && tiles <= a.split_cnt_elems && (double)a.M * (double)a.omap.S1 < 4.0e9;
}

hipError_t launch_gemm_f32_group(const GemmArgs* list, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n == 1 && !worth_splitting(list[0])) return launch_gemm_f32(list[0], s);
    if (n > MAXG) return hipErrorInvalidValue;
    {   // problems that carry the two-fp16-piece pack (GemmArgs::Wh2, igemm_f32h2.hip) go out as their own grid; the HBM-bound pointwise
        // convs keep their kernel (launch_gemm_f32 routes them)
        GemmArgs h2[MAXG], rest[MAXG];
        int nh = 0, nr = 0;
        for (int i = 0; i < n; ++i) {
            if (gemm_f32_on_h2g(list[i])) h2[nh++] = list[i];
            else rest[nr++] = list[i];
        }
        if (nh) {
            const hipError_t e = nh == 1 ? launch_gemm_f32h2g(h2[0], s) : launch_gemm_f32h2g_group(h2, nh, s);
            if (e != hipSuccess) return e;
            for (int i = 0; i < nr; ++i) rest[i].Wh2 = nullptr;
            return nr ? launch_gemm_f32_group(rest, nr, s) : hipSuccess;
        }
    }
    static const int BMs[3] = {128, 64, 128}, BNs[3] = {64, 64, 32};
    // work in units of one 64x64x32 tile-chunk (1024 MFMA cycles of a CU), per CU
    double total = 0.0;
    for (int i = 0; i < n; ++i) {
        if (!gemm_f32_groupable(list[i])) return hipErrorInvalidValue;
        total += (double)list[i].M * list[i].N * (list[i].Kpad / BK) / 4096.0;
    }
    const double per_cu = total / 256.0;
    struct Item { int idx, cfg, tiles; double cost; };
    Item it[MAXG];
    for (int i = 0; i < n; ++i) {
        const GemmArgs& a = list[i];
        const int chunks = a.Kpad / BK;
        int cfg;
        if (a.N <= 32) cfg = 2;
        else {
            // a tile shares its CU with two others: the big tile must not outlast the whole launch
            const double big = chunks * 2.0 * 3.0;
            cfg = (big <= 0.8 * per_cu && a.M >= 128) ? 0 : 1;
        }
        it[i] = Item{i, cfg, ((a.M + BMs[cfg] - 1) / BMs[cfg]) * ((a.N + BNs[cfg] - 1) / BNs[cfg]),
                     chunks * (BMs[cfg] * BNs[cfg] / 4096.0)};
    }
    for (int i = 1; i < n; ++i)                  // longest tile first (insertion sort, n <= 8)
        for (int j = i; j > 0 && it[j].cost > it[j - 1].cost; --j) { Item t = it[j]; it[j] = it[j - 1]; it[j - 1] = t; }
    // Small batches: a level is a few dozen tiles and its duration is the K loop of the longest problem (72 chunks for the
    // 256-channel 8x8 branch against 9 for the 32-channel one).  A problem with a long K and a handful of tiles is split
    // along K; the slices of a tile meet through the scratch the caller lent (GemmArgs::split_ws / split_cnt), reduced in
    // fixed order by the last slice to finish (see igemm_tile).  The decision is a function of the problem ALONE
    // (worth_splitting), so a conv sums in the same order whether it is launched on its own or inside a group.
    // Measured (HRNet-32 256x256, frames/s with / without): batch 1 362 / 302, batch 2 678 / 600, batch 4 1171 / 1172 -- with the
    // __threadfence() hand-off of rounds 2-3; with the agent-scope release / acquire by lane 0 (igemm_tile) the same rule gives 430 / 840 /
    // 1353, and splitting from 16 chunks into slices of >= 6 (was: from 32, >= 12) 442 / 856 / 1354; finer (slices of 4, or problems of up to
    // 64 tiles) loses again (411 / 755 / 1306; 444 / 800 / 1202).  A
    // group-wide rule (split everything to the shortest problem's length whenever the level had < 192 tiles) lost 13 % /
    // 10 % at batch 2 / 4: the device-scope release / acquire around the counter (an L2 write-back on this multi-XCD
    // part) and the second pass over the slabs cost ~15 us, which only a long loop on a handful of tiles repays.
    long ws_used = 0, cnt_used = 0;
    GroupArgs ga;
    ga.n = n;
    int start = 0;
    for (int i = 0; i < n; ++i) {
        GemmArgs a = list[it[i].idx];
        const int chunks = a.Kpad / BK;
        a.splits = 1; a.cps = chunks; a.split_stride = 0;
        float* ws = a.split_ws; int* cnt = a.split_cnt;
        a.split_ws = nullptr; a.split_cnt = nullptr;
        if (ws && cnt && worth_splitting(list[it[i].idx])) {
            int sp = std::min(8, chunks / 6);
            const long slab = (long)a.M * a.N;
            if ((ws_used + slab * sp) > list[it[i].idx].split_ws_elems || cnt_used + it[i].tiles > list[it[i].idx].split_cnt_elems) sp = 1;
            if (sp > 1) {
                a.cps = (chunks + sp - 1) / sp;
                sp = (chunks + a.cps - 1) / a.cps;
                a.splits = sp;
                a.split_stride = slab;
                a.split_ws = ws + ws_used;
                a.split_cnt = cnt + cnt_used;
                ws_used += slab * sp;
                cnt_used += it[i].tiles;
            }
        }
        if (a.rs_div <= 0) a.rs_div = 1;
        if ((double)a.M * (double)a.omap.S1 >= 4.0e9 || (a.res && (double)a.M * (double)a.rmap.S1 >= 4.0e9))
            return hipErrorInvalidValue;
        if (!prep_conv(a)) return hipErrorInvalidValue;
        ga.g[i] = a;
        ga.cfg[i] = it[i].cfg;
        ga.splits[i] = a.splits;
        ga.tiles[i] = it[i].tiles * a.splits;
        ga.start[i] = start;
        start += (ga.tiles[i] + 7) & ~7;
    }
    ga.start[n] = start;
    for (int i = n; i < MAXG; ++i) { ga.start[i + 1] = start; ga.tiles[i] = 0; ga.cfg[i] = 0; ga.splits[i] = 1; }
    bool any_split = false;
    for (int i = 0; i < n; ++i) any_split |= ga.splits[i] > 1;
#ifdef CAPF_DIAG
    static const int abl = [] { const char* e = diag_env("CAPF_ABLATE"); return e ? atoi(e) : 0; }();
    if (abl == 7 && !any_split) {
        hipLaunchKernelGGL((igemm_f32_group_kernel<7>), dim3(start), dim3(256), 0, s, ga);
        return hipGetLastError();
    }
#endif
    if (any_split) hipLaunchKernelGGL((igemm_f32_group_kernel<0, true>), dim3(start), dim3(256), 0, s, ga);
    else hipLaunchKernelGGL((igemm_f32_group_kernel<0>), dim3(start), dim3(256), 0, s, ga);
    return hipGetLastError();
}

hipError_t launch_gemm_f32(const GemmArgs& a_in, hipStream_t s) {
    if (a_in.M <= 0 || a_in.N <= 0) return hipSuccess;
    if (a_in.Kpad % BK != 0) return hipErrorInvalidValue;
    if (pw_preferred(a_in)) return launch_gemm_f32_pw(a_in, s);
    if (gemm_f32h2g_ok(a_in)) return launch_gemm_f32h2g(a_in, s);
    if (a_in.splits <= 1 && worth_splitting(a_in)) return launch_gemm_f32_group(&a_in, 1, s);
    GemmArgs a = a_in;
    if (gemm_f32_rows_splitk(a_in)) {                                      // a handful of tiles with a long K loop: slices + in-launch reduction
        const int chunks = a.Kpad / BK, tiles = ((a.M + 63) / 64) * ((a.N + 63) / 64);
        int sp = std::min(8, chunks / 6);
        a.cps = (chunks + sp - 1) / sp;
        a.splits = (chunks + a.cps - 1) / a.cps;
        a.split_stride = (long)a.M * a.N;
        if (a.rs_div <= 0) a.rs_div = 1;
        if (a.act == ACT_GELU) hipLaunchKernelGGL(igemm_f32_rows_splitk_kernel<true>, dim3(tiles * a.splits), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(igemm_f32_rows_splitk_kernel<false>, dim3(tiles * a.splits), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    a.split_ws = nullptr; a.split_cnt = nullptr;                           // (the in-kernel reduction belongs to the grouped / split kernels)
    if (a.splits <= 1) { a.splits = 1; a.cps = a.Kpad / BK; a.split_stride = 0; }
    else if (a.conv || a.bias || a.res) return hipErrorInvalidValue;     // split-K slabs are raw partial sums
    if (a.rs_div <= 0) a.rs_div = 1;
    // the epilogue's fast path addresses out / res with 32-bit element offsets from their bases
    if (a.omap.G == 1 && (double)a.M * (double)a.omap.S1 >= 4.0e9) return hipErrorInvalidValue;
    if (a.res && a.rmap.G == 1 && (double)a.M * (double)a.rmap.S1 >= 4.0e9) return hipErrorInvalidValue;
    if (!a.conv) {   // rows mode addresses A with absolute 32-bit byte offsets from its base (buffer descriptor)
        const double g = a.amap.G > 0 ? a.amap.G : 1;
        const double span = ((double)a.M / g + 1.0) * (double)a.amap.S1 + g * (double)a.amap.S2 + (double)a.amap.off + a.Kpad;
        if (span * 4.0 >= 4.0e9) return hipErrorInvalidValue;
    }
    if (a.conv) {
        if (!prep_conv(a)) return hipErrorInvalidValue;
        if (a.Cin % 4 != 0) {
            if (stem_on_bf16(a)) return launch_gemm_bf16_smallc(a, s);
            if (stem_stream_f32_ok(a)) {
                const int ntiles = (a.M + 63) / 64;
                int blocks = 512;                          // two per CU
                while (blocks > 8 && blocks / 2 >= ntiles) blocks /= 2;
                hipLaunchKernelGGL((igemm_f32_stem_stream_kernel<3>), dim3(blocks), dim3(256), 0, s, a, ntiles);
                return hipGetLastError();
            }
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

#ifdef CAPF_DIAG
// (diagnosis build only, not part of include/capf.h) copy the CAPF_ABLATE=7 block timeline to the host
extern "C" __attribute__((visibility("default"))) int capf_debug_timeline(unsigned long long* dst, int blocks) {
    if (blocks > capf::DBG_BLOCKS) blocks = capf::DBG_BLOCKS;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(capf::capf_dbg_timeline), (size_t)blocks * 64);
}
#endif
