// fp32 3x3 / stride-1 / pad-1 convolution on the 16-bit matrix pipe with HALF the MFMAs of igemm_f32x3_ws_tile.h: every fp32 operand is
// carried as TWO fp16 pieces under an exact power-of-two block scale, three piece products per fp32 MAC (BasicBlock convs of HRNet,
// pose_hrnet.py:66-95, under compute_dtype = fp32).
//
// Why: the three-bf16-piece tile runs against the board's power limit -- the only lever left is fewer MFMAs per result.
// With s a power of two such that |s a| < 2^15:  a1 = fp16(s a),  a2 = fp16(s a - a1)  (the remainder is exact in fp32), and
//     |s a - a1 - a2| <= 2^-23 |s a|          (11 + 11 mantissa bits and the sign of the remainder; fp32 carries 24),
// for every |s a| >= 2^-3; below that the second piece is an fp16 denormal and the error is at most 2^-25 ABSOLUTE, i.e. 2^-39 of
// the block's largest value.  Products of fp16 numbers are exact in the pipe's fp32 accumulator.  Of the four piece products
//     a1 w1 + (a1 w2 + a2 w1)                  dropped: a2 w2 <= 2^-22 |a w|
// so one term is off by at most 2^-21 of itself (rms ~1e-7), which over the K = 9 Cin terms of a dot product is BELOW the fp32
// accumulation error any fp32 GEMM has: on random operands 6e-9 of the sum of |terms| rms before accumulation (a pairwise fp32
// sum of exact products: 8e-9; the three-bf16 tile: 5e-10; both drown in the MFMA's own fp32 accumulation, ~1e-7).  Not exact-
// operand arithmetic like the three-piece tile (which stays in the library: CAPF_PLAN_F32X3_EXACT), but the same measured distance
// to an fp64 evaluation as this library's direct fp32 MFMA kernel (tests/test_gpu_ops.py::test_f32h2_*).
//
// The scales.  Weights: one power of two per output channel, chosen at pack time (max |w| of the channel's folded filter ->
// [2^14, 2^15)); its inverse comes back in the epilogue.  Pixels: one power of two per BLOCK AND CHUNK -- the 16 channels x (halo
// tile) values a block stages at a time -- from their maximum: every lane takes the max of what it loaded, DPP + readlane reduce it
// per wave, the four wave maxima cross through 16 bytes of LDS under the barrier the stage hand-over needs anyway; the accumulators
// are rescaled (exact: a power of two) whenever the scale changes between chunks, and 1 / scale of the last chunk goes into the
// epilogue's multiplier.  Nothing outside the kernel knows about any of this: tensors in HBM are plain fp32 NHWC.
//
// Tile: igemm_f32x3_ws_tile.h's (256 output pixels x 32 TN channels, accumulators resident for the whole K, 16-channel chunks with
// a 1-pixel halo staged once for all nine taps, pixels through registers, weights by LDS-DMA, half-plane LDS image with the
// conflict-free pixel -> MFMA column deal), with two planes instead of three: 26 KiB of pixels + 18 TN KiB of weights --
// THREE blocks per CU at TN = 1 (the three-piece tile: two), so that a block's split phase between its two barriers is covered by
// two others' MFMAs; 6 VALU instructions per pair of values instead of 11; 4 + 2 TN fragment reads feed 6 TN MFMAs per tap.
#pragma once
#ifndef H2_EXPERIMENT
#define H2_EXPERIMENT 0      // (tools/f32h2_ws.hip knock-outs: timing only)
#endif
#include <algorithm>

#include "igemm_bf16_ws_tile.h"

namespace capf {

static constexpr int H2_HP = WS_MAX_PP * 16 + 64;           // one half-plane (16 B per staged pixel) + 64 B: the two halves of a pixel 16 banks apart
static constexpr int H2_A_BYTES = 4 * H2_HP;
static constexpr int H2_AUX_BYTES = 16 + 2 * 64 * 4 + 4 * 64 * 4 + 16;   // the four wave maxima of the chunk being split; the slice's inverse weight scales and
                                                                   // biases; per wave, which tile pixel each of its 2 x 32 MFMA columns is (epilogue)
inline constexpr int h2_w_bytes(int NS) { return 2 * 9 * NS * 32; }
inline constexpr int h2_lds_bytes(int NS) { return H2_A_BYTES + h2_w_bytes(NS) + H2_AUX_BYTES; }

struct H2Problem {
    WsProblem g;                  // geometry (g.x / g.res / g.y unused; g.wp = packed pieces, g.bias)
    const float* x;               // [B][H][W][C] fp32
    const float* res;             // [M][ldr] fp32 or nullptr
    float* y;                     // [M][ldy] fp32
    const float* winv;            // [ceil(N / 32) * 32] 1 / (the channel's weight scale)
    // PLANES (round 6): a tensor that only ever travels from one tile conv to the next (a BasicBlock's conv1 -> conv2, pose_hrnet.py:78-88)
    // is written by the producer ALREADY SPLIT -- per pixel and 16-channel chunk [piece 0: 16 fp16 | piece 1: 16 fp16], the same 64 bytes
    // and the same addresses as the fp32 values -- under one power-of-two scale per (producer tile, chunk) whose biased exponent goes to
    // eout[tile * (N / 16) + chunk]; the consumer stages the pieces as they are (no maximum, no split, no scale exchange) and only moves the
    // halo rows that came from the neighbouring tiles onto the smallest of the three scales (exact: a power of two).  Same tiling on both sides.
    const int* ein;               // x holds planes: [tiles_m][C / 16] scale exponents (nullptr: x is fp32)
    int* eout;                    // y is written as planes: [tiles_m][N / 16] (nullptr: fp32)
    // UNIT TABLE (round 6, optional): which bytes of x each lane stages is a function of the map's geometry alone -- [H2_NAU][256] words
    // from h2_unit_table(): a lane's unit j = (byte offset from its tile's first row of pixels, a multiple of 16) | bit 0 the column exists,
    // bit 1 / bit 2 the unit lies in the halo row above / below the tile's rows.  The kernel's prologue then loads seven words per lane where
    // it otherwise runs three magic-number divisions and a validity chain per unit (~250 of its ~620 VALU instructions; H2_EXPERIMENT 64: -6 %
    // of a batch-64 level).  nullptr, planes in, or a ragged last tile of several images: the computed path.  Same addresses, same bits.
    const unsigned* utab;
};
static constexpr int H2_NAU = 7;

// packed weights: [32-channel slice][C / 16][piece 2][tap 9][32][2 swizzled halves][8] fp16, then [slices * 32] fp32 inverse scales.  One
// layout for both tile widths: a 64-channel tile stages two neighbouring slices side by side
inline long h2_piece_elems(int N, int C) { return (long)((N + 31) / 32) * (C / 16) * 2 * 9 * 32 * 16; }
inline long h2_pack_elems(int N, int C) { return h2_piece_elems(N, C) + 2L * ((N + 31) / 32) * 32; }

inline bool h2_plan(int B, int H, int W, int C, int N, int NS, H2Problem* q) {
    if (B <= 0 || H <= 0 || W <= 0 || N % 4 != 0 || (NS != 32 && NS != 64)) return false;
    // the tile addresses its pixels from the first frame it touches and its output rows from its first pixel: tensors of any size (the
    // 32-bit offsets span one tile -- at most WS_MAX_PP staged pixels of at most WS_MAX_P / (H W) + 1 frames); pixel COUNTS stay ints
    if ((double)H * W * C * 4.0 * (WS_MAX_P / (H * W) + 2) >= 2.0e9 || (double)WS_MAX_P * N * 4.0 >= 2.0e9) return false;
    if (!ws_plan(B, H, W, C, (N + 7) & ~7, &q->g, true)) return false;
    q->g.N = N;
    q->g.ldy = q->g.ldr = N;
    q->g.NS = NS;
    q->g.NSL = (N + NS - 1) / NS;
    q->utab = nullptr;
    return true;
}

// the unit table of a map geometry (h2_plan's, for any batch and channel counts N): false = no table form (neither one segment per tile nor
// whole images per tile); out: H2_NAU * 256 words
inline bool h2_unit_table(int H, int W, int C, unsigned* out) {
    H2Problem q;
    if (!h2_plan(1, H, W, C, 32, 32, &q)) return false;      // (the tile geometry does not depend on the batch)
    const WsProblem& p = q.g;
    if (p.G != 1 && p.RGPI != 1) return false;
    const int n_units = 4 * p.PP;
    for (int j = 0; j < H2_NAU; ++j)
        for (int tid = 0; tid < 256; ++tid) {
            const int qi = std::min(j * 256 + tid, n_units - 1);
            const int px = qi >> 2, qt = qi & 3;
            const int g = px / p.SEGP, rem = px - g * p.SEGP;
            const int rr = rem / p.PW, ww = rem - rr * p.PW;
            const int col = ww - 1;
            const long off = ((((long)g * p.H + rr - 1) * p.W + col) * p.C + qt * 4) * 4;      // (g = 0 unless a tile holds whole images)
            if (off >= (1L << 31) || off < -(1L << 31)) return false;
            out[j * 256 + tid] = ((unsigned)(int)off & ~15u) | (col >= 0 && col < p.W ? 1u : 0u) | (rr == 0 ? 2u : 0u) | (rr == p.RH + 1 ? 4u : 0u);
        }
    return true;
}

// biased exponent of the power-of-two scale for a block maximum with bit pattern `mbits` (>= 0): max * scale in [2^14, 2^15).  Kept within
// 2^+-63, so that the scale, its inverse, the ratio of two scales and (1 / pixel scale) * (1 / weight scale) are all normal numbers:
// maxima in [2^-49, 2^77) -- 1.8e-15 .. 1.5e23 -- get their exact scale; smaller ones keep 2^63 (precision relative to the maximum
// degrades below 2^-49; everything below 2^-87 is zero), larger ones overflow fp16 (Inf, then NaN -- include/capf.h)
inline __host__ __device__ int h2_scale_exp(int mbits) {
    const int sb = 268 - (mbits >> 23);
    return sb < 64 ? 64 : (sb > 190 ? 190 : sb);
}

#if defined(__HIP_DEVICE_COMPILE__)

typedef _Float16 ws_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ws_u32x2 __attribute__((ext_vector_type(2)));

// a pair of fp32 values, scaled by s (a power of two) -> the pair's two packed fp16 pieces
__device__ __forceinline__ void h2_split2(float x, float y, float s, unsigned& p1, unsigned& p2) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = f2_t{x, y} * s;
    const h2_t a = __builtin_convertvector(v, h2_t);
    const f2_t r = v - __builtin_convertvector(a, f2_t);
    const h2_t b = __builtin_convertvector(r, h2_t);
    p1 = __builtin_bit_cast(unsigned, a);
    p2 = __builtin_bit_cast(unsigned, b);
}

// two packed fp16 values times two packed fp16 values (a power of two in both halves here: exact, denormal results included)
__device__ __forceinline__ unsigned h2_pk_mul(unsigned a, unsigned b) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t r = __builtin_bit_cast(h2_t, a) * __builtin_bit_cast(h2_t, b);
    return __builtin_bit_cast(unsigned, r);
}

// maximum of a non-negative value over the wave (bit patterns of non-negative floats order like integers), wave-uniform
__device__ __forceinline__ int h2_wave_max(float v) {
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));     // row_half_mirror
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));     // row_mirror
    return max(max(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)),
               max(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
}

// one tile (logical id bid = pixel tile * NSL + channel slice) with the calling 256-thread block; lds: h2_lds_bytes(32 TN) bytes
// PIN: x holds planes (q.ein);  POUT: y is written as planes (q.eout; no residual)
template <int TN, bool PIN = false, bool POUT = false>
__device__ __forceinline__ void igemm_f32h2_ws_tile(const H2Problem& q, const int bid, unsigned char* __restrict__ lds) {
    constexpr int NS = 32 * TN;
    constexpr int WP_BYTES = 9 * 32 * 32;                  // one piece of a 32-channel slice's chunk
    constexpr int WS_BYTES = 2 * WP_BYTES;                 // a slice's chunk: 18 LDS-DMA instructions
    constexpr int W2_BYTES = TN * WS_BYTES;                // the tile's chunk: TN slices side by side
    constexpr int NWI = W2_BYTES / 1024;                   // weight DMA instructions per chunk: 18 TN
    constexpr int NWS = (NWI + 3) / 4;                     // ... per wave
    constexpr int NAU = H2_NAU;                            // quarter-pixel units (4 channels = 16 B of fp32) per lane and chunk (1664 at most in all)
    constexpr unsigned OOB = 0x80000000u;
    const WsProblem& p = q.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    const int NCC = p.C >> 4;
    const int tm = bid / p.NSL, slice = bid - tm * p.NSL;

    const int b_first = ws_div(tm * p.G, p.d_rgpi);         // first frame this tile touches: pixel offsets count from there
    const ws_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(q.x + (size_t)b_first * p.H * p.W * p.C), 0, 0x7FFFFF00u, 0x00020000);
    // ---- pixel units of this lane.  LDS image of a piece = two HALF-PLANES (channels 0-7 / 8-15 of the chunk), 16 B per staged pixel,
    // pixel-linear (igemm_f32x3_ws_tile.h).  A unit is a QUARTER of a staged pixel's chunk -- 4 channels, one 16-byte load -- and four
    // consecutive lanes take the four quarters of one pixel: a wave's load instruction is 16 requests of 64 contiguous bytes (eight
    // consecutive pixels of one half per lane group, as the three-piece tile has it, is 64 requests of 16 bytes: the L1's tag rate,
    // not its bandwidth, then bounds three resident blocks -- 2048 cycles per chunk and block against 1728 of MFMAs)
    constexpr int HP = H2_HP;                                // piece pc, half hf at (2 pc + hf) HP
    unsigned a_voff[NAU], a_lds[NAU];
    [[maybe_unused]] unsigned a_cls = 0;
    const int n_units = 4 * p.PP;
    // (block-uniform) the table form: one segment per tile, or whole images per tile and every one of this tile's images exists
    const bool tab_ok = !(H2_EXPERIMENT & 128) && !PIN && q.utab != nullptr && (p.G == 1 || (p.RGPI == 1 && (tm + 1) * p.G <= p.RG));
    if (tab_ok) {
        const int k = tm * p.G - b_first * p.RGPI;           // the tile's row group inside its (first) image: 0 when a tile holds whole images
        const unsigned kill = (k == 0 ? 2u : 0u) | (k == p.RGPI - 1 ? 4u : 0u);       // halo rows that lie outside the image
        const int base = k * p.RH * p.W * p.C * 4;
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            const unsigned t = q.utab[j * 256 + tid];
            const int qi = min(j * 256 + tid, n_units - 1);
            const int px = qi >> 2, qt = qi & 3;
            a_lds[j] = (unsigned)((qt >> 1) * HP + px * 16 + (qt & 1) * 8);
            a_voff[j] = ((t & 1u) && !(t & kill)) ? (unsigned)((int)(t & ~15u) + base) : OOB;
        }
    } else {
        const int q0 = tm * p.G;
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            const int qi = min(j * 256 + tid, n_units - 1);  // (beyond the geometry: the last unit once more, same bytes same place)
            const int px = qi >> 2, qt = qi & 3;
            a_lds[j] = PIN ? (unsigned)((2 * (qt >> 1) + (qt & 1)) * HP + px * 16)       // (planes: quarter qt = piece qt >> 1, channels 8 (qt & 1) ..)
                           : (unsigned)((qt >> 1) * HP + px * 16 + (qt & 1) * 8);
#if (H2_EXPERIMENT & 64)
            a_voff[j] = (unsigned)(qi * 16 + q0);            // (timing only: no per-unit geometry)
            continue;
#endif
            const int g = ws_div(px, p.d_segp), rem = px - g * p.SEGP;
            const int rr = ws_div(rem, p.d_pw), ww = rem - rr * p.PW;
            if (PIN) a_cls |= (rr == 0 ? 1u : (rr == p.RH + 1 ? 2u : 0u)) << (2 * j);   // whose scale the unit's row carries: own tile / the one above / below
            const int sg = q0 + g;
            const int b = ws_div(sg, p.d_rgpi);
            const int h = (sg - b * p.RGPI) * p.RH + rr - 1, col = ww - 1;
            const bool ok = sg < p.RG && h >= 0 && h < p.H && col >= 0 && col < p.W;
            a_voff[j] = ok ? (unsigned)(((((b - b_first) * p.H + h) * p.W + col) * p.C + qt * 4) * 4) : OOB;
        }
    }
    ws_f32x4 ar[NAU];
    auto load_a = [&](int cc) {
#pragma unroll
        for (int j = 0; j < NAU; ++j)
            ar[j] = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, a_voff[j], (unsigned)cc * 64u, 0));
    };
    int* const aux = reinterpret_cast<int*>(lds + H2_A_BYTES + W2_BYTES);
    float* const aux_w = reinterpret_cast<float*>(aux + 4);      // [NS] 1 / weight scale of the slice's channels
    float* const aux_b = aux_w + 64;                             // [NS] their biases
    int* const aux_px = reinterpret_cast<int*>(aux_b + 64) + wave * 64;   // [2][32] this wave's column -> tile pixel table
    [[maybe_unused]] int* const aux_o = reinterpret_cast<int*>(aux_b + 64) + 4 * 64;      // [2 TN] POUT: largest |output| of the tile's 16-channel chunks
    // ---- PIN: the scales come with the planes.  A tile with in-image neighbours (one segment of RH < H rows) reads three exponents per chunk,
    // its own and those of the tiles its two halo rows came from; the chunk's scale is the smallest of them
    [[maybe_unused]] int e_nb[3] = {190, 190, 190};        // (own, above, below) of the chunk about to be staged
    [[maybe_unused]] const int* e_row[3] = {nullptr, nullptr, nullptr};
    if constexpr (PIN) {
        e_row[0] = q.ein + (size_t)tm * NCC;
        if (p.G == 1 && p.RH < p.H) {
            const int k = tm - ws_div(tm, p.d_rgpi) * p.RGPI;                 // row group of the tile inside its image
            if (k > 0) e_row[1] = q.ein + (size_t)(tm - 1) * NCC;
            if (k + 1 < p.RGPI) e_row[2] = q.ein + (size_t)(tm + 1) * NCC;
        }
    }
    [[maybe_unused]] int e_raw[3] = {190, 190, 190};       // (requested together with the chunk's pixels, a chunk ahead)
    auto load_e = [&](int cc) {
        if constexpr (PIN) {
#pragma unroll
            for (int k = 0; k < 3; ++k) e_raw[k] = e_row[k] ? e_row[k][cc] : 190;
        }
    };
    auto use_e = [&]() {
        if constexpr (PIN) {
#pragma unroll
            for (int k = 0; k < 3; ++k) e_nb[k] = __builtin_amdgcn_readfirstlane(e_raw[k]);
        }
    };
    // which of the three row classes a wave's unit j holds at all (wave-uniform bit masks, bit j): the rescale below is skipped per unit on
    // SCALAR conditions -- typically nothing needs moving (equal exponents), or only the units of a halo row do
    [[maybe_unused]] unsigned has_cls[3] = {0, 0, 0};
    if constexpr (PIN) {
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            const unsigned c = (a_cls >> (2 * j)) & 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) has_cls[k] |= (__builtin_amdgcn_ballot_w64(c == (unsigned)k) != 0 ? 1u : 0u) << j;
        }
    }
    auto store_planes = [&](int sn) {                      // the loaded units -> LDS, rows of other tiles moved onto the chunk's scale 2^(sn - 127)
        // classes whose rows are on another scale than the chunk's (a missing neighbour's rows are zeros: never)
        const unsigned moved = (e_nb[0] != sn ? has_cls[0] : 0u) | ((e_nb[1] != sn && e_nb[1] != 190) ? has_cls[1] : 0u) |
                               ((e_nb[2] != sn && e_nb[2] != 190) ? has_cls[2] : 0u);
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            const ws_u32x4 src = __builtin_bit_cast(ws_u32x4, ar[j]);
            unsigned w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
            if ((moved >> j) & 1) {                        // (wave-uniform)
                const int cls = (a_cls >> (2 * j)) & 3;
                const int en = cls == 0 ? e_nb[0] : (cls == 1 ? e_nb[1] : e_nb[2]);
                const int r = en == 190 ? 0 : sn - en;     // <= 0
                // 2^r in one or two fp16 factors: 2^max(r, -14) (normal), then 2^(r + 14) down to 2^-24 as a denormal, i.e. exact scaling down
                // to 2^-38 -- a first piece of 2^15 is still 2^-23 there; below that everything the row holds is under the fp16 grid (2^-24 in
                // the chunk's units = 2^-39 of its largest value: the bound of include/capf.h) and becomes zero
                const int r1 = r >= -14 ? r : -14, r2 = r - r1;
                const unsigned hb = (unsigned)(15 + r1) << 10, mb = hb | (hb << 16);
                w0 = h2_pk_mul(w0, mb); w1 = h2_pk_mul(w1, mb); w2 = h2_pk_mul(w2, mb); w3 = h2_pk_mul(w3, mb);
                if (__builtin_amdgcn_ballot_w64(r2 < 0)) {                      // (rows more than 2^14 below the chunk's scale: rare)
                    const unsigned gb = r2 >= -14 ? (unsigned)(15 + r2) << 10 : (r2 >= -24 ? 1u << (24 + r2) : 0u), nb = gb | (gb << 16);
                    w0 = h2_pk_mul(w0, nb); w1 = h2_pk_mul(w1, nb); w2 = h2_pk_mul(w2, nb); w3 = h2_pk_mul(w3, nb);
                }
            }
            *reinterpret_cast<ws_u32x4*>(lds + a_lds[j]) = ws_u32x4{w0, w1, w2, w3};
        }
    };
    auto publish_max = [&]() {                             // this wave's maximum of the loaded chunk -> aux[wave]
        if constexpr (PIN) return;
#if (H2_EXPERIMENT & 1)
        return;
#endif
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < NAU; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(ar[j][e]));
        const int wm = h2_wave_max(m);
        if (lane == 0) aux[wave] = wm;
    };
    auto block_scale_exp = [&]() -> int {                  // (after the barrier behind publish_max)
        if constexpr (PIN) return min(e_nb[0], min(e_nb[1], e_nb[2]));
#if (H2_EXPERIMENT & 1)
        return 127;
#endif
        const ws_u32x4 v = *reinterpret_cast<const ws_u32x4*>(aux);
        const int m = max(max((int)v[0], (int)v[1]), max((int)v[2], (int)v[3]));
        return __builtin_amdgcn_readfirstlane(h2_scale_exp(m));
    };
    auto split_a = [&](float s) {                          // the loaded chunk, scaled -> two fp16 planes
        if constexpr (PIN) { store_planes(__float_as_int(s) >> 23); return; }
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            ws_u32x2 u1, u2;
#if (H2_EXPERIMENT & 1)
            u1[0] = __float_as_uint(ar[j][0]); u1[1] = __float_as_uint(ar[j][1]); u2[0] = __float_as_uint(ar[j][2]); u2[1] = __float_as_uint(ar[j][3]);
            *reinterpret_cast<ws_u32x2*>(lds + a_lds[j]) = u1;
            *reinterpret_cast<ws_u32x2*>(lds + 2 * HP + a_lds[j]) = u2;
            continue;
#endif
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned s1, s2;
                h2_split2(ar[j][2 * k], ar[j][2 * k + 1], s, s1, s2);
                u1[k] = s1; u2[k] = s2;
            }
            *reinterpret_cast<ws_u32x2*>(lds + a_lds[j]) = u1;
            *reinterpret_cast<ws_u32x2*>(lds + 2 * HP + a_lds[j]) = u2;
        }
    };
    const unsigned w_voff = (unsigned)lane * 16u;
    const int nsl32 = (p.N + 31) >> 5;                     // (a 64-channel tile's second slice may not exist: its DMA reads zeros)
    const ws_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wp + (size_t)slice * TN * NCC * (WS_BYTES / 2)), 0,
                                                             (unsigned)(nsl32 - slice * TN) * (unsigned)NCC * (unsigned)WS_BYTES, 0x00020000);
    auto fire_w = [&](int cc) {
#pragma unroll
        for (int i = 0; i < NWS; ++i) {
            const int k = min(i * 4 + wave, NWI - 1);
            const int j = k / 18, rem = k - j * 18;        // slice j of the tile, instruction rem of its chunk
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (ws_lptr_t)(lds + H2_A_BYTES + k * 1024), 16, w_voff,
                                                     (unsigned)(j * NCC + cc) * (unsigned)WS_BYTES + (unsigned)rem * 1024u, 0, 0);
        }
    };
    load_a(0);
    load_e(0);
    if (POUT && tid < 4) aux_o[tid] = 0;
    if (tid < NS) {
        const int n = slice * NS + tid;
        aux_w[tid] = n < ((p.N + 31) & ~31) ? q.winv[n] : 1.f;
        aux_b[tid] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }

    // ---- which tile pixel a lane's MFMA column is (igemm_f32x3_ws_tile.h): ds_read_b128 is served in lane groups {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} (+ 32); a group costs one LDS cycle iff its 16 staged pixels are distinct mod 16
    int pl_i[2];
    {
        const int in_g1 = (frow >= 4 && frow < 12) || (frow >= 16 && frow < 20) || frow >= 28;
        const int pos = in_g1 ? (frow < 12 ? frow - 4 : (frow < 20 ? frow - 8 : frow - 16))
                              : (frow < 4 ? frow : (frow < 16 ? frow - 8 : frow - 12));
        if ((p.W & 15) == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) pl_i[i] = ((2 * wave + i) * 2 + in_g1) * 16 + pos;
        } else {
            unsigned short* tab = reinterpret_cast<unsigned short*>(lds);            // [16 groups][16 classes]
            unsigned short* ovf = tab + 256;
            int* cnt = reinterpret_cast<int*>(lds + 1024);                           // [16] + overflow / empty counters
            if (tid < 18) cnt[tid] = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < p.P) {
                const int g = ws_div(tid, p.d_rhw), rem = tid - g * p.RHW;
                const int r = ws_div(rem, p.d_w), w = rem - r * p.W;
                const int c = ((g * (p.RH + 2) + r) * p.PW + w) & 15;
                const int rank = atomicAdd(&cnt[c], 1);
                if (rank < 16) tab[rank * 16 + c] = (unsigned short)tid;
                else ovf[atomicAdd(&cnt[16], 1)] = (unsigned short)tid;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if ((tid >> 4) >= cnt[tid & 15]) {                                      // an empty cell: an overflowed pixel, or idle
                const int e = atomicAdd(&cnt[17], 1);
                tab[tid] = e < cnt[16] ? ovf[e] : (unsigned short)0xFFFFu;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) pl_i[i] = tab[((2 * wave + i) * 2 + in_g1) * 16 + pos];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                            // (the planes overwrite the table)
        }
    }
    fire_w(0);                                              // (behind the table's last barrier: the DMA lands in the weight area only)
    unsigned a_addr[2][3];                                  // pixel block i, filter row kh; kw and the piece are immediates
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int pl = pl_i[i];
        if (pl >= p.P) { pl = 0; pl_i[i] = 0x7FFF; }         // (idle columns of a ragged geometry: computed, never stored)
        const int g = ws_div(pl, p.d_rhw), rem = pl - g * p.RHW;
        const int r = ws_div(rem, p.d_w), w = rem - r * p.W;
        const int pix0 = (g * (p.RH + 2) + r) * p.PW + w;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) a_addr[i][kh] = (unsigned)((pix0 + kh * p.PW) * 16 + fhalf * HP);
    }
    const unsigned b_addr = (unsigned)(H2_A_BYTES + frow * 32 + ((fhalf ^ ((frow >> 3) & 1)) << 4));
    if (fhalf == 0) { aux_px[frow] = pl_i[0]; aux_px[32 + frow] = pl_i[1]; }      // (read after the barriers of the first chunk)

    const int Mi = (int)p.M;
    const int gp0 = tm * p.G * p.RHW;                      // first flat output pixel of the tile: row offsets count from there
    const ws_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(q.res ? (void*)(q.res + (size_t)gp0 * p.ldr) : (void*)q.y, 0, q.res ? 0x7FFFFF00u : 0u,
                                                               0x00020000);
    const ws_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(q.y + (size_t)gp0 * p.ldy), 0, 0x7FFFFF00u, 0x00020000);
    // (weight piece, pixel piece) of the three products, smallest first
    constexpr int PW_[3] = {0, 1, 0};
    constexpr int PA_[3] = {1, 0, 0};

    // ---- chunk 0: maximum -> scale -> split
    publish_max();                                         // (the compiler waits for the pixel loads; the weight DMA may still fly)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    use_e();
    int sb = block_scale_exp();
    int smin = sb;
    split_a(__int_as_float(sb << 23));

    ws_f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (NCC > 1) { load_a(1); load_e(1); }
    __builtin_amdgcn_s_barrier();

    // ---- epilogue addressing: lane = 4 consecutive channels (register group g) of its pixel; the residual rows are requested before
    // the last chunk's MFMAs
    // Coalesced layout of the epilogue: lane (er, ec) = 8 consecutive channels (32 B) of block row h * 16 + er, so that four lanes cover a
    // pixel's 128 contiguous bytes and a wave instruction touches 16 full lines -- as pixel-per-lane (what the accumulators are) a 16-byte
    // store / residual load is 64 requests of 16 B at a stride of ld * 4 bytes.  The accumulators cross through 4.5 KiB of per-wave LDS
    // scratch (over the dead pixel planes) after the K loop; which pixel a block row is comes from aux_px.
    const int er = lane >> 2, ec = (lane & 3) * 8;
    auto row_off = [&](int pl, int j, int q, int ld) -> unsigned {        // quad q (4 channels) of the lane's 8
        const int n = slice * NS + j * 32 + ec + 4 * q;
        const int gp = gp0 + pl;
        return (pl < p.P && gp < Mi && n < p.N) ? (unsigned)(pl * ld + n) * 4u : OOB;
    };
    ws_f32x4 rr[2][TN][4];                                  // [pixel block][channel block][h * 2 + q]
    ws_f16x8 af[2][2][2], bfr[2][2][TN];
    for (int cc = 0; cc < NCC; ++cc) {
        if (cc == NCC - 1 && !(H2_EXPERIMENT & 32) && !POUT) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pl = aux_px[i * 32 + h * 16 + er];
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            rr[i][j][h * 2 + q] = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, row_off(pl, j, q, p.ldr), 0, 0));
                }
        }
        auto read_frags = [&](int t, int buf) {            // (in the order the products below consume them)
            if ((H2_EXPERIMENT & 16) && (t > 0 || cc > 0)) return;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int pw = PW_[o], pa = PA_[o];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[buf][pw][j] = __builtin_bit_cast(ws_f16x8, *reinterpret_cast<const ws_f32x4*>(lds + b_addr + j * WS_BYTES + pw * WP_BYTES + t * 1024));
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    af[buf][pa][i] = __builtin_bit_cast(ws_f16x8, *reinterpret_cast<const ws_f32x4*>(lds + pa * 2 * HP + (t % 3) * 16 + a_addr[i][t / 3]));
            }
        };
        read_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t < 8) read_frags(t + 1, (t + 1) & 1);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (!(H2_EXPERIMENT & 8)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bfr[t & 1][PW_[k]][j], af[t & 1][PA_[k]][i], acc[i][j], 0, 0, 0);
                        else acc[i][j][0] += (float)bfr[t & 1][PW_[k]][j][0] + (float)af[t & 1][PA_[k]][i][0];
            if (t < 8) {                                   // the next tap's 4 + 2 TN fragment reads one at a time behind this tap's MFMAs
#pragma unroll
                for (int x = 0; x < 4 + 2 * TN; ++x) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (6 * TN > 4 + 2 * TN) __builtin_amdgcn_sched_group_barrier(0x008, 6 * TN - (4 + 2 * TN), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cc + 1 < NCC) {
            publish_max();                                 // chunk cc + 1's pixels were requested a chunk ago
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // everybody is done reading the stage; the four maxima are in place
            if (!(H2_EXPERIMENT & 2)) fire_w(cc + 1);
            use_e();
            // (a scale never rises more than 2^80 above the smallest one of the tile so far: the accumulators hold up to 2^35 in units of
            // that scale and must not overflow behind a chunk of zeros; values that far below an earlier chunk are below the fp32 sum)
            const int sn = min(block_scale_exp(), smin + 80);
            smin = min(smin, sn);
            if (sn != sb) {                                // (block-uniform) the accumulators move to the new scale: exact
                const float f = __int_as_float((127 + sn - sb) << 23);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] *= f;
                sb = sn;
            }
            split_a(__int_as_float(sb << 23));
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (cc + 2 < NCC && !(H2_EXPERIMENT & 4)) { load_a(cc + 2); load_e(cc + 2); }
            __builtin_amdgcn_s_barrier();
        }
    }

    if (H2_EXPERIMENT & 32) {                              // (timing only: one store per lane that keeps the accumulators alive)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        if (t == 1234.5678f) q.y[tid] = t;
        return;
    }
    // ---- epilogue: y = acc / (pixel scale * the channel's weight scale) + bias (+ residual), ReLU
    // (accumulator register 4 g + e of channel block j = channel slice * NS + 32 j + 8 g + 4 fhalf + e of the lane's pixel)
    const float inv_s = __int_as_float((254 - sb) << 23);
    if constexpr (POUT) {
        // planes out: the chunk maxima straight from the accumulators (lane = pixel, register 4 g + e = channel 8 g + 4 fhalf + e of column
        // block j: g = 0, 1 -> chunk 2 j, g = 2, 3 -> chunk 2 j + 1), so that they cross LDS under the barrier the epilogue has anyway
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float m[2] = {0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const ws_f32x4 wv = *reinterpret_cast<const ws_f32x4*>(aux_w + j * 32 + 8 * g + 4 * fhalf);
                const ws_f32x4 bv = *reinterpret_cast<const ws_f32x4*>(aux_b + j * 32 + 8 * g + 4 * fhalf);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = fmaf(acc[i][j][4 * g + e], wv[e] * inv_s, bv[e]);
                        m[g >> 1] = fmaxf(m[g >> 1], p.relu ? fmaxf(t, 0.f) : fabsf(t));
                    }
            }
            const int w0 = h2_wave_max(m[0]), w1 = h2_wave_max(m[1]);
            if (lane == 0) { atomicMax(&aux_o[2 * j], w0); atomicMax(&aux_o[2 * j + 1], w1); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // every wave is done with the planes: the scratch below overlays them
    constexpr int EPS = 36;
    float* ep = reinterpret_cast<float*>(lds) + wave * (32 * EPS);
    [[maybe_unused]] const int hi_chunk = (lane >> 1) & 1;  // (coalesced layout: the lane's 8 channels lie in chunk 2 j + hi_chunk)
    if constexpr (POUT) {
        const int nck = p.N >> 4;
        if (tid < 2 * TN && slice * 2 * TN + tid < nck) q.eout[(size_t)tm * nck + slice * 2 * TN + tid] = h2_scale_exp(aux_o[tid]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        ws_f32x4 wv[2], bv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wv[q] = *reinterpret_cast<const ws_f32x4*>(aux_w + j * 32 + ec + 4 * q);
            bv[q] = *reinterpret_cast<const ws_f32x4*>(aux_b + j * 32 + ec + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[q][e] *= inv_s;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<ws_f32x4*>(&ep[frow * EPS + 8 * g + 4 * fhalf]) =
                    ws_f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = h * 16 + er;
                const int pl = aux_px[i * 32 + row];
                [[maybe_unused]] ws_f32x4 ov[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const ws_f32x4 x = *reinterpret_cast<const ws_f32x4*>(&ep[row * EPS + ec + 4 * q]);
                    ws_f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = POUT ? fmaf(x[e], wv[q][e], bv[q][e]) : fmaf(x[e], wv[q][e], bv[q][e] + rr[i][j][h * 2 + q][e]);
                        o[e] = p.relu ? fmaxf(t, 0.f) : t;
                    }
                    if constexpr (POUT) ov[q] = o;
                    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, o), rs_out, row_off(pl, j, q, p.ldy), 0, 0);
                }
                if constexpr (POUT) {                      // the row's 8 channels -> piece 0 | piece 1 where the fp32 values would have gone
                    const float sc = __int_as_float(h2_scale_exp(aux_o[2 * j + hi_chunk]) << 23);
                    ws_u32x4 p0, p1;
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            unsigned s1, s2;
                            h2_split2(ov[qq][2 * k], ov[qq][2 * k + 1], sc, s1, s2);
                            p0[2 * qq + k] = s1; p1[2 * qq + k] = s2;
                        }
                    const int n = slice * NS + j * 32 + ec;
                    const bool ok = pl < p.P && gp0 + pl < Mi && n < p.N;
                    const unsigned off = ok ? (unsigned)(pl * p.ldy + (n & ~15)) * 4u + (unsigned)(n & 15) * 2u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(p0, rs_out, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(p1, rs_out, ok ? off + 32u : OOB, 0, 0);
                }
            }
        }
    }
}

#endif

}  // namespace capf
