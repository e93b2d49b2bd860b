// fp32 3x3 / stride-1 / pad-1 convolution on the bf16 matrix pipe: the "2-D halo" tile of igemm_bf16_ws_tile.h with every fp32 operand
// split, losslessly, into three bf16 pieces (BasicBlock convs of HRNet, pose_hrnet.py:66-95, under compute_dtype = fp32).
//
// Why: v_mfma_f32_32x32x2_f32 retires 2 k per 64 cycles, v_mfma_f32_32x32x16_bf16 16 k per 32 -- 16 x the rate.  An fp32 number is
// EXACTLY a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2) (8 + 8 + 8 mantissa bits and the signs of the
// remainders; round-to-nearest leaves nothing over), and a product of two bf16 numbers is exact in the pipe's fp32 accumulator.  Of
// the nine piece products of a * w the six of weight >= 2^-18 are kept:
//     a1 w1 + (a1 w2 + a2 w1) + (a1 w3 + a2 w2 + a3 w1)          dropped: a2 w3 + a3 w2 + a3 w3 <= 2^-25 |a w|,
// below the 2^-24 rounding of the fp32 multiply this replaces (measured on random operands: 1.6e-9 of the sum of |terms| before the
// fp32 accumulation, which is the same as in any fp32 GEMM; tools/f32x3_ws.hip prints the distance to an fp64 evaluation).
// Six bf16 MFMAs for what is 8 fp32 MFMAs' worth of k: 2.67 x the fp32 pipe, 1.33 x what F(4,3) Winograd on the fp32 pipe can reach
// (igemm_wino.hip: 0.52 busy), with a smaller error than either.
//
// Tile (geometry, LDS image and fragment addressing are WsProblem's): 256 output pixels x 32 TN channels, accumulators resident
// for the whole K, 16-channel chunks staged once for all nine taps.  What differs from the bf16 tile:
//   * pixels arrive as fp32 through registers (8 channels of a staged pixel = two 16-byte loads per unit, 4 units per lane and
//     chunk), are split there (11 VALU instructions per pair of values) and written as three bf16 planes in the bf16 tile's
//     layout; the loads of chunk c + 1 are in flight under the MFMAs of chunk c;
//   * weights are split once at pack time; a chunk's three pieces (27 TN KiB) arrive by LDS-DMA;
//   * per tap and 32-pixel block 3 + 3 fragments feed 6 TN MFMAs (bf16 tile: 1 + TN feed TN): half the LDS reads per MFMA;
//   * ONE stage (3 x 13 KiB of pixels + 27 TN KiB of weights: two blocks per CU at TN = 1), so two barriers per chunk: the split of
//     chunk c + 1 runs between them, under the OTHER resident block's MFMAs;
//   * fp32 epilogue straight from the accumulators: lane = 4 consecutive channels of one pixel per 16-byte store.
#pragma once
#include "igemm_bf16_ws_tile.h"

namespace capf {

static constexpr int X3_A_BYTES = 3 * WS_A_BYTES;
inline constexpr int x3_w_bytes(int NS) { return 3 * 9 * NS * 32; }
inline constexpr int x3_lds_bytes(int NS) { return X3_A_BYTES + x3_w_bytes(NS); }

struct X3Problem {
    WsProblem g;                  // geometry (g.x / g.res / g.y unused; g.wp = packed pieces, g.bias)
    const float* x;               // [B][H][W][C] fp32
    const float* res;             // [M][ldr] fp32 or nullptr
    float* y;                     // [M][ldy] fp32
};

// packed weights: [N slice][C / 16][piece 3][tap 9][NS][2 swizzled halves][8] bf16
inline long x3_pack_elems(int N, int C, int NS) { return (long)((N + NS - 1) / NS) * (C / 16) * 3 * 9 * NS * 16; }

inline bool x3_plan(int B, int H, int W, int C, int N, int NS, X3Problem* q) {
    if (N % 4 != 0 || (NS != 32 && NS != 64)) return false;
    if ((double)B * H * W * C * 4.0 >= 2.0e9 || (double)B * H * W * N * 4.0 >= 2.0e9) return false;
    if (!ws_plan(B, H, W, C, (N + 7) & ~7, &q->g)) return false;
    q->g.N = N;
    q->g.ldy = q->g.ldr = N;
    q->g.NS = NS;
    q->g.NSL = (N + NS - 1) / NS;
    return true;
}

#if defined(__HIP_DEVICE_COMPILE__)

// a pair of fp32 values -> the pair's three packed bf16 pieces
__device__ __forceinline__ void x3_split2(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = ws_pack2(x, y);
    const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = ws_pack2(rx, ry);
    const float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xFFFF0000u);
    p3 = ws_pack2(sx, sy);
}

// `nt` consecutive tiles (logical ids bid0 .. bid0 + nt - 1, id = pixel tile * NSL + channel slice) with the calling 256-thread block; lds:
// x3_lds_bytes(32 TN) bytes.  Between two tiles nothing drains: the next tile's first pixel chunk is requested before the current
// tile's last chunk of MFMAs, its first weight chunk right after them (the epilogue reads no LDS), so a narrow-channel tile -- two or
// four chunks -- starts with its operands already on the way.  Measured (tools/f32x3_ws.hip, X3_NT): nothing -- 32 ch 64^2 at batch 512
// 322 us with one tile per block, 330 with two, 334 with four -- because these launches are not waiting on latencies either: the
// shader clock inside that launch reads 1.2 GHz (HBM traffic and MFMAs share the board's power budget), at which a tile's 21 k
// cycles are two thirds MFMA time.  igemm_f32x3_ws.hip therefore launches one tile per block (more blocks, shorter tail).
template <int TN>
__device__ __forceinline__ void igemm_f32x3_ws_tiles(const X3Problem& q, const int bid0, const int nt, unsigned char* __restrict__ lds) {
    constexpr int NS = 32 * TN;
    constexpr int WP_BYTES = 9 * NS * 32;                  // one piece of a chunk's weights
    constexpr int W3_BYTES = 3 * WP_BYTES;
    constexpr int NWI = W3_BYTES / 1024;                   // weight DMA instructions per chunk: 27 TN
    constexpr int NWS = (NWI + 3) / 4;                     // ... per wave
    constexpr int NAU = 4;                                 // half-pixel units per lane and chunk (832 at most in all)
    constexpr unsigned OOB = 0x80000000u;
    const WsProblem& p = q.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    const int NCC = p.C >> 4;

    const ws_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)q.x, 0, 0x7FFFFF00u, 0x00020000);
    // ---- pixel units of this lane.  LDS image of a piece: two HALF-PLANES (channels 0-7 / 8-15 of the chunk), 16 B per staged pixel,
    // pixel-linear -- a fragment read's bank slot is (staged pixel + constant) mod 16, so a lane group that is conflict-free
    // for one tap is conflict-free for all nine (see the pixel permutation below).  Unit qi: lanes 8 k .. 8 k + 7 = eight
    // consecutive pixels of one half (a ds_write_b128 lane group = 128 contiguous bytes)
    constexpr int HP = WS_MAX_PP * 16;                       // one half-plane; piece pc, half hf at (2 pc + hf) HP
    static_assert(6 * HP == X3_A_BYTES, "plane layout");
    unsigned a_voff[NAU], a_lds[NAU];
    const int n_units = 2 * ((p.PP + 7) & ~7);
#pragma unroll
    for (int j = 0; j < NAU; ++j) {
        const int qi = min(j * 256 + tid, n_units - 1);      // (beyond the geometry: the last unit once more, same bytes same place)
        a_lds[j] = (unsigned)(((qi >> 3) & 1) * HP + (((qi >> 4) << 3) | (qi & 7)) * 16);
    }
    auto tile_voff = [&](int bid) {                        // global offsets of this lane's units for tile `bid`
        const int q0 = (bid / p.NSL) * p.G;
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            const int qi = min(j * 256 + tid, n_units - 1);
            const int half = (qi >> 3) & 1;
            const int px = ((qi >> 4) << 3) | (qi & 7);
            const int g = ws_div(px, p.d_segp), rem = px - g * p.SEGP;
            const int rr = ws_div(rem, p.d_pw), ww = rem - rr * p.PW;
            const int sg = q0 + g;
            const int b = ws_div(sg, p.d_rgpi);
            const int h = (sg - b * p.RGPI) * p.RH + rr - 1, col = ww - 1;
            const bool ok = px < p.PP && sg < p.RG && h >= 0 && h < p.H && col >= 0 && col < p.W;
            a_voff[j] = ok ? (unsigned)((((b * p.H + h) * p.W + col) * p.C + half * 8) * 4) : OOB;
        }
    };
    ws_f32x4 ar[NAU][2];
    auto load_a = [&](int cc) {
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            ar[j][0] = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, a_voff[j], (unsigned)cc * 64u, 0));
            ar[j][1] = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, a_voff[j] + 16u, (unsigned)cc * 64u, 0));
        }
    };
    auto split_a = [&]() {                                 // the loaded chunk -> three bf16 planes
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            ws_u32x4 u1, u2, u3;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned s1, s2, s3;
                x3_split2(ar[j][k >> 1][2 * (k & 1)], ar[j][k >> 1][2 * (k & 1) + 1], s1, s2, s3);
                u1[k] = s1; u2[k] = s2; u3[k] = s3;
            }
            *reinterpret_cast<ws_u32x4*>(lds + a_lds[j]) = u1;
            *reinterpret_cast<ws_u32x4*>(lds + 2 * HP + a_lds[j]) = u2;
            *reinterpret_cast<ws_u32x4*>(lds + 4 * HP + a_lds[j]) = u3;
        }
    };
    const unsigned w_voff = (unsigned)lane * 16u;
    auto fire_w = [&](int bid, int cc) {                   // chunk cc of tile bid's channel slice
        const int slice = bid - (bid / p.NSL) * p.NSL;
        const ws_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wp + (size_t)slice * NCC * (W3_BYTES / 2)), 0,
                                                                 (unsigned)NCC * (unsigned)W3_BYTES, 0x00020000);
#pragma unroll
        for (int i = 0; i < NWS; ++i) {
            const int k = min(i * 4 + wave, NWI - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (ws_lptr_t)(lds + X3_A_BYTES + k * 1024), 16, w_voff,
                                                     (unsigned)cc * (unsigned)W3_BYTES + (unsigned)k * 1024u, 0, 0);
        }
    };
    tile_voff(bid0);
    load_a(0);
    fire_w(bid0, 0);

    // ---- which tile pixel a lane's MFMA column is.  ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
    // (+ 32): a group costs one LDS cycle iff its 16 staged pixels are distinct mod 16.  W % 16 == 0: group k of the tile =
    // tile pixels 16 k .. 16 k + 15 (consecutive in a row).  Otherwise (8-wide rows with their 10-pixel pitch, the 72 / 36 / 18 /
    // 9-wide CPN maps) the tile's pixels are dealt out by residue class -- group `rank`, slot `class` -- with LDS atomics; which
    // pixel lands in which column is irrelevant to the arithmetic (a column is one pixel's dot products).
    int pl_i[2];
    {
        // lane -> (which of the block's two groups, position in it)
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
    const unsigned b_addr = (unsigned)(X3_A_BYTES + frow * 32 + ((fhalf ^ ((frow >> 3) & 1)) << 4));

    const int Mi = (int)p.M;
    const ws_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)p.bias : (void*)q.y, 0, p.bias ? (unsigned)p.N * 4u : 0u, 0x00020000);
    const ws_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(q.res ? (void*)q.res : (void*)q.y, 0, q.res ? 0x7FFFFF00u : 0u, 0x00020000);
    const ws_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)q.y, 0, 0x7FFFFF00u, 0x00020000);
    // (weight piece, pixel piece) of the six products, smallest first
    constexpr int PW_[6] = {0, 2, 1, 0, 1, 0};
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0};

    for (int it = 0; it < nt; ++it) {
        const int bid = bid0 + it;
        const int tm = bid / p.NSL, slice = bid - tm * p.NSL;
        const int gp0 = tm * p.G * p.RHW;                  // first flat output pixel of the tile
        const bool more = it + 1 < nt;

        // ---- accumulators start at the bias (register 4 g + e of channel block j = channel slice * NS + 32 j + 8 g + 4 fhalf + e)
        ws_f32x16 acc[2][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const ws_f32x4 bv = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                     rs_bias, (unsigned)(slice * NS + j * 32 + 8 * g + 4 * fhalf) * 4u, 0, 0));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = bv[e];
            }

        split_a();                                         // chunk 0 (requested before the previous tile's last MFMAs, or above)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (NCC > 1) load_a(1);
        else if (more) { tile_voff(bid + 1); load_a(0); }
        __builtin_amdgcn_s_barrier();

        // ---- epilogue addressing: lane = 4 consecutive channels (register group g) of its pixel; the residual rows are requested before
        // the last chunk's MFMAs
        auto piece_off = [&](int i, int j, int g, int ld) -> unsigned {
            const int pl = pl_i[i], n = slice * NS + j * 32 + 8 * g + 4 * fhalf;
            const int gp = gp0 + pl;
            return (pl < p.P && gp < Mi && n < p.N) ? (unsigned)(gp * ld + n) * 4u : OOB;
        };
        ws_f32x4 rr[2][TN][4];
        // fragments of two taps in registers, a tap's reads issued one tap ahead, one behind every MFMA.  (Two taps ahead -- three register
        // sets, 256 VGPRs -- changes nothing: 228 vs 232 us at batch 512, 128 ch 16^2.  The kernel runs against the board's power limit,
        // not against an issue or LDS limit: tools/f32x3_ws.hip reads the shader clock inside the launch -- 1.60 GHz sustained at batch
        // 512, 1.96 GHz in a 30 us launch at batch 64, nominal 2.4 -- so "0.40 of the nominal bf16 peak" is 0.60 of the pipe's cycles.)
        ws_bf16x8 af[2][3][2], bfr[2][3][TN];
        for (int cc = 0; cc < NCC; ++cc) {
            if (cc == NCC - 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            rr[i][j][g] = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, piece_off(i, j, g, p.ldr), 0, 0));
            }
            auto read_frags = [&](int t, int buf) {        // (in the order the products below consume them)
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const int pw = PW_[o], pa = PA_[o];
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bfr[buf][pw][j] = __builtin_bit_cast(ws_bf16x8, *reinterpret_cast<const ws_f32x4*>(lds + b_addr + pw * WP_BYTES + (t * NS + j * 32) * 32));
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        af[buf][pa][i] = __builtin_bit_cast(ws_bf16x8, *reinterpret_cast<const ws_f32x4*>(lds + pa * 2 * HP + (t % 3) * 16 + a_addr[i][t / 3]));
                }
            };
            read_frags(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t < 8) read_frags(t + 1, (t + 1) & 1);
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[t & 1][PW_[k]][j], af[t & 1][PA_[k]][i], acc[i][j], 0, 0, 0);
                if (t < 8) {                               // the next tap's 6 + 3 TN fragment reads one at a time behind this tap's MFMAs
#pragma unroll
                    for (int x = 0; x < 6 + 3 * TN; ++x) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 12 * TN - (6 + 3 * TN), 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (cc + 1 < NCC) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();              // everybody is done reading the stage
                fire_w(bid, cc + 1);
                split_a();                                 // (the loads of chunk cc + 1 went out a chunk ago; the DMA lands under the split)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (cc + 2 < NCC) load_a(cc + 2);
                else if (more) { tile_voff(bid + 1); load_a(0); }      // the next tile's first chunk, under this tile's last MFMAs
                __builtin_amdgcn_s_barrier();
            }
        }
        if (more) {                                        // the next tile's first weights, under the epilogue (which reads no LDS)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            fire_w(bid + 1, 0);
        }

        // ---- epilogue
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    ws_f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = acc[i][j][4 * g + e] + rr[i][j][g][e];
                        o[e] = p.relu ? fmaxf(t, 0.f) : t;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, o), rs_out, piece_off(i, j, g, p.ldy), 0, 0);
                }
        }
    }
}

template <int TN>
__device__ __forceinline__ void igemm_f32x3_ws_tile(const X3Problem& q, const int bid, unsigned char* __restrict__ lds) {
    igemm_f32x3_ws_tiles<TN>(q, bid, 1, lds);
}

#endif

}  // namespace capf
