// All `res_blocks` of the lifter as ONE launch (PoseTransformer.forward pose_dformer.py:231-234: Block x depth over the (levels + 1) tokens
// of each joint; Block :62-79, Attention :34-59, Mlp :15-31), compute_dtype = fp32.
//
// A res block is five dependent launches (LayerNorm-folded qkv, attention, proj, LayerNorm-folded fc1 + GELU, fc2) on [B 17 5, 128] rows --
// 2.9 % of the FLOPs of a forward in launches that are all latency: 20 of them for the four blocks, ~8 us each at batch 64, ~12 us each at
// batch 1.  A joint's 5 tokens only ever meet each other in these blocks, so a workgroup can own 12 joints (60 rows,
// two 32-row MFMA blocks per wave: what bounds the launch is the weight stream, 54 MB per block at batch 64, so every B fragment feeds both) and take them through ALL blocks without leaving the CU: the residual stream X, the normalised rows, q | k | v and
// the hidden layer live in LDS (130 KiB), the weights stream from L2 as MFMA B fragments (0.59 MB per block and workgroup, the two-fp16-piece
// packs the per-op route uses: [N][K / 32][piece 0: 32 fp16 | piece 1: 32 fp16], then 1 / channel scale), and every wave multiplies the
// tile's 32 rows by its quarter of a projection's columns.
//
// Arithmetic: igemm_f32h2.hip's, at every batch -- an activation a travels as a1 = fp16(s a), a2 = fp16(s a - a1) under a power-of-two
// scale s, a1 w1 + a1 w2 + a2 w1 accumulated in fp32 (include/capf.h, THE BOUND) -- with the split done ONCE, by the phase that PRODUCES the
// operand (LayerNorm, attention, GELU: they hold the values in registers anyway), under one scale per 32 rows and projection (the maximum
// crosses 8 bytes of LDS under the barrier the phase hand-over needs): the four waves read ready fp16 fragments, no wave re-splits what
// another one split, and the K loop is fragment reads and MFMAs only (VERDICT r5 item 4's "split the A tile once per block").  LayerNorm,
// softmax, GELU (exact erf) and the residual adds in fp32.  Same expressions per output as the per-op kernels (layernorm / attention_kernel /
// igemm_f32h2g epilogues), so the two routes agree to fp32 summation order (tests/test_gpu_lifter_chain.py; CAPF_PLAN_NO_FUSED_LIFTER
// keeps the per-op route).
#include "igemm_f32h2_ws_tile.h"
#include "kernels.h"
#ifndef RC_STAMP
#define RC_STAMP(i)          // (tools/res_chain_bench.hip: per-phase clock stamps)
#endif

namespace capf {

typedef float rc_f32x4 __attribute__((ext_vector_type(4)));
typedef float rc_f32x16 __attribute__((ext_vector_type(16)));

static constexpr int RC_C = 128;                 // token width
static constexpr int RC_ROWS = 64;               // rows per workgroup: two 32-row MFMA blocks per wave (every B fragment feeds both)
[[maybe_unused]] static constexpr int RC_MB = RC_ROWS / 32;
static constexpr int RC_LDX = RC_C + 4;          // padded LDS rows: 16-byte accesses of 32 consecutive rows spread over the banks
// a work row (388 floats): [q -> attention output | k | v] in fp32 while the attention runs; otherwise columns 128 .. hold the fp16 PLANES of
// the operand of the next projection -- piece 0 of the row's K values, then piece 1 (K = 128: 2 x 256 B; the hidden layer, K = 256: 2 x 512 B)
static constexpr int RC_LDQ = 3 * RC_C + 4;
static constexpr int RC_LDS_FLOATS = RC_ROWS * (RC_LDX + RC_LDQ) + 8 + 4 * 64;      // X | work rows | the row blocks' maxima | prefetch landing pads: 131 KiB

#if defined(__HIP_DEVICE_COMPILE__)

__device__ __forceinline__ float rc_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float rc_sum8(float v) {      // sum over the 8 lanes of a row (aligned groups of 8), every lane gets it
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));     // row_half_mirror: the other quad's sum
    return v;
}

// Pull a weight pack towards this CU's L2 BEFORE the projection that reads it: inside a forward the lifter's weights are cold (the backbone
// has been through every cache since the last step), and a fragment load one chunk ahead then waits out an HBM round trip per chunk
// (175 us per launch inside the engine against 99 in a loop that keeps the weights hot).  One dword per 128-byte line and lane through the
// LDS-DMA path (no destination registers to keep alive; the bytes land on a per-wave pad nobody reads), the waves taking turns.
__device__ __forceinline__ void rc_prefetch(const float* __restrict__ pack, const int floats, float* __restrict__ pad, const int wave, const int lane) {
    const ws_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pack), 0, (unsigned)floats * 4u, 0x00020000);
    const int lines = (floats * 4 + 127) >> 7;
    for (int l0 = wave * 64; l0 < lines; l0 += 4 * 64)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (ws_lptr_t)(pad + wave * 64), 4, (unsigned)(l0 + lane) * 128u, 0, 0, 0);
}

// four consecutive values of a row -> their two fp16 pieces under scale sc, stored at piece 0 / piece 1 of the row's planes (k = column)
template <int K>
__device__ __forceinline__ void rc_store_planes(float* __restrict__ wrow, const int k, const rc_f32x4 v, const float sc) {
    unsigned a0, a1, b0, b1;
    h2_split2(v[0], v[1], sc, a0, b0);
    h2_split2(v[2], v[3], sc, a1, b1);
    unsigned short* p = reinterpret_cast<unsigned short*>(wrow + RC_C);
    *reinterpret_cast<ws_u32x2*>(p + k) = ws_u32x2{a0, a1};
    *reinterpret_cast<ws_u32x2*>(p + K + k) = ws_u32x2{b0, b1};
}

// acc[mb][j] (rows 32 mb .., columns n0 + 32 j ..) = A[64 rows][K] . W[n][K]^T on the two-piece arithmetic.  A: the fp16 planes of the work
// rows (already split, one scale per 32 rows).  W: the projection's two-piece pack RE-LAID OUT in fragment order (res_chain_repack_kernel):
// [32-row block][chunk][piece][step][lane 64][8 fp16], so that a wave's B fragment load is ONE KiB of contiguous memory -- straight from
// the [N][K] pack a wave instruction touched 32 cache lines for 32 bytes each, and the address path, not L2, bounded the loop (measured
// 3.3 k cycles per 48 KiB chunk of the qkv projection: 14 B per cycle and CU).  Loaded from global memory (L2) one chunk ahead.
template <int NT, int K>
__device__ __forceinline__ void rc_gemm(const float* __restrict__ Ws, const float* __restrict__ Wp, const int n0, const int lane,
                                        rc_f32x16 (&acc)[RC_MB][NT]) {
    const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
    for (int mb = 0; mb < RC_MB; ++mb)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][j][r] = 0.f;
    constexpr int nchunks = K >> 5;
    // narrow projections (one or two column blocks per wave) keep the two 16-deep steps of a chunk in SEPARATE accumulators, summed at the
    // end: with the three piece products issued product-major over every (row block, column block, step) a dependent MFMA is always 2 NT SP
    // instructions away from its predecessor -- back to back on one accumulator each one waits out the previous one's latency (measured:
    // ~100 cycles per MFMA on a lone wave per SIMD against 32 of issue)
    constexpr int SP = NT <= 2 ? 2 : 1;
    rc_f32x16 part[RC_MB][NT];                             // (SP == 2: the odd steps' partial sums)
    if constexpr (SP == 2) {
#pragma unroll
        for (int mb = 0; mb < RC_MB; ++mb)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) part[mb][j][r] = 0.f;
    }
    ws_u32x4 bw[2][NT][2][2];
    auto load_b = [&](int c, int buf) {                    // (fragment-order pack: 1 KiB contiguous per wave instruction)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float* blkp = Wp + ((size_t)((n0 >> 5) + j) * nchunks + c) * (4 * 64 * 4) + lane * 4;
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                for (int st = 0; st < 2; ++st)
                    bw[buf][j][pc][st] = *reinterpret_cast<const ws_u32x4*>(blkp + (2 * pc + st) * (64 * 4));
        }
    };
    load_b(0, 0);
#pragma unroll
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunks) load_b(c + 1, cur ^ 1);
        ws_f16x8 a1[RC_MB][2], a2[RC_MB][2];
#pragma unroll
        for (int mb = 0; mb < RC_MB; ++mb) {
            const unsigned short* p = reinterpret_cast<const unsigned short*>(Ws + (32 * mb + frow) * RC_LDQ + RC_C);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int k = c * 32 + 16 * st + 8 * fhalf;
                a1[mb][st] = __builtin_bit_cast(ws_f16x8, *reinterpret_cast<const ws_u32x4*>(p + k));
                a2[mb][st] = __builtin_bit_cast(ws_f16x8, *reinterpret_cast<const ws_u32x4*>(p + K + k));
            }
        }
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)                     // (weight piece, activation piece) = (0, 2nd), (1, 1st), (0, 1st): smallest first
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int mb = 0; mb < RC_MB; ++mb)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const ws_f16x8 wv = __builtin_bit_cast(ws_f16x8, bw[cur][j][pr == 1 ? 1 : 0][st]);
                        const ws_f16x8 av = pr == 0 ? a2[mb][st] : a1[mb][st];
                        if (SP == 2 && st == 1) part[mb][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv, av, part[mb][j], 0, 0, 0);
                        else acc[mb][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv, av, acc[mb][j], 0, 0, 0);
                    }
    }
    if constexpr (SP == 2) {
#pragma unroll
        for (int mb = 0; mb < RC_MB; ++mb)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][j][r] += part[mb][j][r];
    }
}

#endif

struct ResChainArgs {
    float* X;                    // fp32 rows of 128, updated in place: row m at X + (m / rG) * rS1 + (m % rG) * rS2 + roff
    int rows, heads, nblk;
    float eps;
    int rG;
    long rS1, rS2, roff;
    ResBlockW blk[8];
};

// ATTN = false: only the second half of a block -- x + fc2(gelu(fc1(norm2(x)))) -- on arbitrary rows: the MLP half of a deformable context
// block (pose_dformer.py:137-138) behind ctx_attn_kernel, one launch instead of two
template <int T, bool ATTN>      // tokens per group
__global__ __launch_bounds__(256, 1) void res_chain_kernel(ResChainArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float rc_lds[];
    float* const Xs = rc_lds;
    float* const Ws = Xs + RC_ROWS * RC_LDX;               // work rows (see RC_LDQ)
    int* const smax = reinterpret_cast<int*>(Ws + RC_ROWS * RC_LDQ);      // [phase parity][row block]: bit pattern of the block's largest |value|
    float* const pad = reinterpret_cast<float*>(smax + 8);
    constexpr int P_QKV = 3 * RC_C * RC_C + 3 * RC_C, P_PROJ = RC_C * RC_C + RC_C, P_FC1 = 2 * RC_C * RC_C + 2 * RC_C, P_FC2 = 2 * RC_C * RC_C + RC_C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    constexpr int G = RC_ROWS / T, R = ATTN ? G * T : RC_ROWS;      // whole token groups per tile: 12 x 5 = 60 of the 64 rows
    auto xrow = [&](int m) -> float* { return a.X + (size_t)(m / a.rG) * a.rS1 + (size_t)(m % a.rG) * a.rS2 + a.roff; };
    constexpr int HD = 16;                                 // (8 heads: res_chain_ok)
    const int m0 = blockIdx.x * R;
    const int nrows = min(R, a.rows - m0);
    if (ATTN) { rc_prefetch(a.blk[0].wqkv, P_QKV, pad, wave, lane); rc_prefetch(a.blk[0].wproj, P_PROJ, pad, wave, lane); }
    else { rc_prefetch(a.blk[0].wfc1, P_FC1, pad, wave, lane); rc_prefetch(a.blk[0].wfc2, P_FC2, pad, wave, lane); }
    RC_STAMP(0);
    // ---- X tile in (rows beyond the tile's groups: zeros -- they ride along through every phase and are never stored)
    for (int i = tid; i < RC_ROWS * (RC_C / 4); i += 256) {
        const int r = i / (RC_C / 4), c = (i - r * (RC_C / 4)) * 4;
        rc_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < nrows) v = *reinterpret_cast<const rc_f32x4*>(xrow(m0 + r) + c);
        *reinterpret_cast<rc_f32x4*>(&Xs[r * RC_LDX + c]) = v;
    }
    if (tid < 8) smax[tid] = 0;
    __syncthreads();
    RC_STAMP(1);
    int par = 0;                                           // which pair of maxima the current producer phase uses (the other pair is being reset)
    // a producer phase has left the maxima of its values: the scale of row block mb, and its inverse for the epilogue
    auto scale_exp = [&](int mb) -> int { return h2_scale_exp(smax[2 * par + mb]); };
    auto next_phase = [&]() { par ^= 1; if (tid < 2) smax[2 * (par ^ 1) + tid] = 0; };     // (after the barrier behind the readers of the old pair)
    // y = acc / (row block's scale * channel's weight scale) + bias for accumulator group g (columns n .. n + 3)
    auto scaled = [&](const rc_f32x16& acc, int g, const float* winv, const float* bias, int n, float inv_s) -> rc_f32x4 {
        const rc_f32x4 wv = *reinterpret_cast<const rc_f32x4*>(winv + n), bv = *reinterpret_cast<const rc_f32x4*>(bias + n);
        rc_f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(acc[4 * g + e], wv[e] * inv_s, bv[e]);
        return o;
    };
    // LayerNorm of the tile's rows -> planes: 8 lanes per row, two passes over the row like layernorm_kernel; a pass covers one row block
    auto layernorm = [&](const float* g, const float* bta) {
        rc_f32x4 o[RC_ROWS / 32][4];
#pragma unroll
        for (int pass = 0; pass < RC_ROWS / 32; ++pass) {
            const int row = 32 * pass + (tid >> 3), part = tid & 7;
            rc_f32x4 v[4];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = *reinterpret_cast<const rc_f32x4*>(&Xs[row * RC_LDX + 4 * (part + 8 * i)]);
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
            const float mean = rc_sum8(s) / (float)RC_C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
            const float rstd = 1.0f / sqrtf(rc_sum8(q) / (float)RC_C + a.eps);
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * (part + 8 * i);
                const rc_f32x4 gv = *reinterpret_cast<const rc_f32x4*>(g + c), bv = *reinterpret_cast<const rc_f32x4*>(bta + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[pass][i][e] = (v[i][e] - mean) * rstd * gv[e] + bv[e]; m = fmaxf(m, fabsf(o[pass][i][e])); }
            }
            const int wm = h2_wave_max(m);
            if (lane == 0) atomicMax(&smax[2 * par + pass], wm);
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < RC_ROWS / 32; ++pass) {
            const int row = 32 * pass + (tid >> 3), part = tid & 7;
            const float sc = __int_as_float(scale_exp(pass) << 23);
#pragma unroll
            for (int i = 0; i < 4; ++i) rc_store_planes<RC_C>(&Ws[row * RC_LDQ], 4 * (part + 8 * i), o[pass][i], sc);
        }
    };
    for (int b = 0; b < a.nblk; ++b) {
        const ResBlockW& w = a.blk[b];
        if constexpr (ATTN) {
        // ---- x + proj(attn(norm1(x)))
        layernorm(w.ln1_g, w.ln1_b);
        __syncthreads();
        if (b == 0) RC_STAMP(2);
        {
            rc_f32x16 acc[RC_MB][3];
            const int n0 = wave * 96;
            rc_prefetch(w.wfc1, P_FC1, pad, wave, lane);                     // (two projections ahead: proj was requested a projection ago)
            rc_gemm<3, RC_C>(Ws, w.wqkv, n0, lane, acc);
            if (b == 0) RC_STAMP(3);
            const int se[RC_MB] = {scale_exp(0), scale_exp(1)};
            __syncthreads();                               // (q | k | v overwrite the planes every wave has just read)
            next_phase();
            const float* winv = w.wqkv + (size_t)3 * RC_C * RC_C;
#pragma unroll
            for (int mb = 0; mb < RC_MB; ++mb) {
                const float inv_s = __int_as_float((254 - se[mb]) << 23);
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + 32 * j + 8 * g + 4 * fhalf;
                        *reinterpret_cast<rc_f32x4*>(&Ws[(32 * mb + frow) * RC_LDQ + n]) = scaled(acc[mb][j], g, winv, w.bqkv, n, inv_s);
                    }
            }
        }
        __syncthreads();
        if (b == 0) RC_STAMP(4);
        // attention over the T tokens of a group: one thread per (group, head, token), attention_kernel's expressions; the outputs wait in
        // registers for the row blocks' maxima, then go out as planes over the dead k columns
        {
            constexpr int NI = (G * 8 * T + 255) / 256;    // items per thread (heads <= 8)
            rc_f32x4 o[NI][4];                             // (head_dim <= 16)
            float m[RC_MB] = {0.f, 0.f};
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int t = tid + 256 * it;
                const bool on = t < G * a.heads * T;
                const int tt = on ? t : 0;
                const int i = tt % T, h = (tt / T) % a.heads, g = tt / (T * a.heads);
                const float scale = 1.0f / sqrtf((float)HD);
                const float* q = &Ws[(g * T + i) * RC_LDQ + h * HD];
                float sc[T];
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    const float* k = &Ws[(g * T + j) * RC_LDQ + RC_C + h * HD];
                    float s = 0.f;
#pragma unroll
                    for (int c = 0; c < HD; c += 4) {
                        const rc_f32x4 qa = *reinterpret_cast<const rc_f32x4*>(q + c), ka = *reinterpret_cast<const rc_f32x4*>(k + c);
                        s += qa[0] * ka[0] + qa[1] * ka[1] + qa[2] * ka[2] + qa[3] * ka[3];
                    }
                    sc[j] = s * scale;
                    mx = fmaxf(mx, sc[j]);
                }
                float den = 0.f;
#pragma unroll
                for (int j = 0; j < T; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
                const float inv = 1.0f / den;
                float mm = 0.f;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    o[it][c4] = rc_f32x4{0.f, 0.f, 0.f, 0.f};
                    if (4 * c4 < HD) {
#pragma unroll
                        for (int j = 0; j < T; ++j)
                            o[it][c4] += *reinterpret_cast<const rc_f32x4*>(&Ws[(g * T + j) * RC_LDQ + 2 * RC_C + h * HD + 4 * c4]) * (sc[j] * inv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) mm = fmaxf(mm, fabsf(o[it][c4][e]));
                    }
                }
                if (on) { if (g * T + i < 32) m[0] = fmaxf(m[0], mm); else m[1] = fmaxf(m[1], mm); }
            }
#pragma unroll
            for (int mb = 0; mb < RC_MB; ++mb) {
                const int wm = h2_wave_max(m[mb]);
                if (lane == 0) atomicMax(&smax[2 * par + mb], wm);
            }
            __syncthreads();                               // every k / v read is done; the maxima are in place
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int t = tid + 256 * it;
                if (t < G * a.heads * T) {
                    const int i = t % T, h = (t / T) % a.heads, g = t / (T * a.heads);
                    const int row = g * T + i;
                    const float sc = __int_as_float(scale_exp(row >> 5) << 23);
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4)
                        if (4 * c4 < HD) rc_store_planes<RC_C>(&Ws[row * RC_LDQ], h * HD + 4 * c4, o[it][c4], sc);
                }
            }
            if (tid < RC_ROWS - R) {                       // (the rows beyond the tile's groups: zero planes)
                unsigned* p = reinterpret_cast<unsigned*>(&Ws[(R + tid) * RC_LDQ + RC_C]);
                for (int k = 0; k < RC_C; ++k) p[k] = 0u;
            }
        }
        __syncthreads();
        if (b == 0) RC_STAMP(5);
        {
            rc_f32x16 acc[RC_MB][1];
            const int n0 = wave * 32;
            rc_prefetch(w.wfc2, P_FC2, pad, wave, lane);
            rc_gemm<1, RC_C>(Ws, w.wproj, n0, lane, acc);
            const float* winv = w.wproj + (size_t)RC_C * RC_C;
#pragma unroll
            for (int mb = 0; mb < RC_MB; ++mb) {
                const float inv_s = __int_as_float((254 - scale_exp(mb)) << 23);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + 8 * g + 4 * fhalf;
                    float* xp = &Xs[(32 * mb + frow) * RC_LDX + n];
                    *reinterpret_cast<rc_f32x4*>(xp) = scaled(acc[mb][0], g, winv, w.bproj, n, inv_s) + *reinterpret_cast<const rc_f32x4*>(xp);
                }
            }
        }
        __syncthreads();
        next_phase();
        if (b == 0) RC_STAMP(6);
        }
        // ---- x + fc2(gelu(fc1(norm2(x))))
        layernorm(w.ln2_g, w.ln2_b);
        __syncthreads();
        if (b == 0) RC_STAMP(7);
        {
            rc_f32x16 acc[RC_MB][2];
            const int n0 = wave * 64;
            if (ATTN && b + 1 < a.nblk) rc_prefetch(a.blk[b + 1].wqkv, P_QKV, pad, wave, lane);
            rc_gemm<2, RC_C>(Ws, w.wfc1, n0, lane, acc);
            const float* winv = w.wfc1 + (size_t)2 * RC_C * RC_C;
            rc_f32x4 hv[RC_MB][2][4];
            int wm[RC_MB];
#pragma unroll
            for (int mb = 0; mb < RC_MB; ++mb) {
                const float inv_s = __int_as_float((254 - scale_exp(mb)) << 23);
                float m = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        hv[mb][j][g] = scaled(acc[mb][j], g, winv, w.bfc1, n0 + 32 * j + 8 * g + 4 * fhalf, inv_s);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { hv[mb][j][g][e] = rc_gelu(hv[mb][j][g][e]); m = fmaxf(m, fabsf(hv[mb][j][g][e])); }
                    }
                wm[mb] = h2_wave_max(m);
            }
            __syncthreads();                               // (the hidden planes overwrite the planes every wave has just read; the old maxima are read)
            next_phase();
            if (lane == 0) { atomicMax(&smax[2 * par + 0], wm[0]); atomicMax(&smax[2 * par + 1], wm[1]); }
            __syncthreads();
#pragma unroll
            for (int mb = 0; mb < RC_MB; ++mb) {
                const float sc = __int_as_float(scale_exp(mb) << 23);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        rc_store_planes<2 * RC_C>(&Ws[(32 * mb + frow) * RC_LDQ], n0 + 32 * j + 8 * g + 4 * fhalf, hv[mb][j][g], sc);
            }
        }
        __syncthreads();
        if (b == 0) RC_STAMP(8);
        {
            rc_f32x16 acc[RC_MB][1];
            const int n0 = wave * 32;
            if (ATTN && b + 1 < a.nblk) rc_prefetch(a.blk[b + 1].wproj, P_PROJ, pad, wave, lane);
            rc_gemm<1, 2 * RC_C>(Ws, w.wfc2, n0, lane, acc);
            const float* winv = w.wfc2 + (size_t)RC_C * 2 * RC_C;
#pragma unroll
            for (int mb = 0; mb < RC_MB; ++mb) {
                const float inv_s = __int_as_float((254 - scale_exp(mb)) << 23);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + 8 * g + 4 * fhalf;
                    float* xp = &Xs[(32 * mb + frow) * RC_LDX + n];
                    *reinterpret_cast<rc_f32x4*>(xp) = scaled(acc[mb][0], g, winv, w.bfc2, n, inv_s) + *reinterpret_cast<const rc_f32x4*>(xp);
                }
            }
        }
        __syncthreads();
        next_phase();
        if (b == 0) RC_STAMP(9);
    }
    RC_STAMP(40);
    for (int i = tid; i < RC_ROWS * (RC_C / 4); i += 256) {
        const int r = i / (RC_C / 4), c = (i - r * (RC_C / 4)) * 4;
        if (r < nrows) *reinterpret_cast<rc_f32x4*>(xrow(m0 + r) + c) = *reinterpret_cast<const rc_f32x4*>(&Xs[r * RC_LDX + c]);
    }
#endif
}

// two-piece pack [N][K floats] (128-byte chunks {piece 0: 32 fp16 | piece 1: 32 fp16}, then [N] inverse scales) -> the same bits in fragment
// order [N / 32][K / 32][piece][step][lane][4 floats], then the inverse scales; one thread per 16-byte fragment piece
__global__ void res_chain_repack_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int K) {
    const int nchunks = K >> 5;
    const long total = (long)N * K / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total + (N + 3) / 4; i += (long)gridDim.x * blockDim.x) {
        if (i >= total) {                                  // the inverse channel scales ride behind
            const long e = (i - total) * 4;
            for (int k = 0; k < 4 && e + k < N; ++k) dst[(long)N * K + e + k] = src[(long)N * K + e + k];
            continue;
        }
        long t = i;
        const int lane = (int)(t & 63); t >>= 6;
        const int st = (int)(t & 1); t >>= 1;
        const int pc = (int)(t & 1); t >>= 1;
        const int c = (int)(t % nchunks);
        const int nt = (int)(t / nchunks);
        const int n = nt * 32 + (lane & 31), fhalf = lane >> 5;
        const rc_f32x4 v = *reinterpret_cast<const rc_f32x4*>(src + (size_t)n * K + c * 32 + (4 * pc + 2 * st + fhalf) * 4);
        *reinterpret_cast<rc_f32x4*>(dst + i * 4) = v;
    }
}

hipError_t launch_res_chain_repack(const float* h2g_pack, float* chain_pack, int N, int K, hipStream_t s) {
    if (N % 32 != 0 || K % 32 != 0) return hipErrorInvalidValue;
    const long total = (long)N * K / 4 + (N + 3) / 4;
    hipLaunchKernelGGL(res_chain_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h2g_pack, chain_pack, N, K);
    return hipGetLastError();
}

bool res_chain_ok(int dim, int tokens, int heads, int nblk) {
#ifdef CAPF_NO_CHAIN                 // (A/B builds: tools/ab_libs.sh)
    return false;
#endif
    return dim == RC_C && tokens == 5 && heads == 8 && nblk >= 1 && nblk <= 8;
}

template <bool ATTN>
static hipError_t rc_launch(const ResChainArgs& a, int R, hipStream_t s) {
    static DynLdsAttr attr;
    const size_t lds = (size_t)RC_LDS_FLOATS * sizeof(float);
    const hipError_t e = attr.ensure(reinterpret_cast<const void*>(&res_chain_kernel<5, ATTN>), (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((res_chain_kernel<5, ATTN>), dim3((unsigned)((a.rows + R - 1) / R)), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_res_chain(float* X, int rows, int tokens, int heads, float eps, const ResBlockW* blk, int nblk, hipStream_t s) {
    if (!res_chain_ok(RC_C, tokens, heads, nblk) || rows <= 0 || rows % tokens != 0) return hipErrorInvalidValue;
    ResChainArgs a{};
    a.X = X; a.rows = rows; a.heads = heads; a.nblk = nblk; a.eps = eps;
    a.rG = 1; a.rS1 = RC_C; a.rS2 = 0; a.roff = 0;
    for (int i = 0; i < nblk; ++i) a.blk[i] = blk[i];
    return rc_launch<true>(a, (RC_ROWS / tokens) * tokens, s);
}

// x + fc2(gelu(fc1(LayerNorm(x)))) on the rows a RowMap names (blk.ln2_*, wfc1, bfc1, wfc2, bfc2; the attention half's fields unused)
hipError_t launch_mlp_chain(float* X, RowMap rows_map, int rows, float eps, const ResBlockW& blk, hipStream_t s) {
    if (rows <= 0 || rows_map.G <= 0 || ((rows_map.S1 | rows_map.S2 | rows_map.off) & 3)) return hipErrorInvalidValue;
    ResChainArgs a{};
    a.X = X; a.rows = rows; a.heads = 8; a.nblk = 1; a.eps = eps;
    a.rG = rows_map.G; a.rS1 = rows_map.S1; a.rS2 = rows_map.S2; a.roff = rows_map.off;
    a.blk[0] = blk;
    return rc_launch<false>(a, RC_ROWS, s);
}

}  // namespace capf
