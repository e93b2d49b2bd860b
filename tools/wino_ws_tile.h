// fp32 Winograd F(4,3)-along-W convolution (3x3 / stride 1 / pad 1) on the loop structure of the bf16 "2-D halo" tile
// (igemm_bf16_ws_tile.h): EXPERIMENT (not part of libcapf), used by the stand-alone harness tools/wino_ws.hip.  Outcome (EXPERIMENTS.md round 4): correct, its K loop alone reaches 0.65 MFMA-busy (production tile 0.53-0.58) with two blocks per CU, but the register-layout epilogue and the 8x8 / 16x16 branches (80-128 blocks of 70 us) leave it level with the production kernel.
//
//   v = B^T d, u = G g, y = A^T m as in igemm_wino.hip (same constants): a W-tile is four output pixels of a row, computed from six raw
//   pixels through six positions; position p is a GEMM over K = 3 Cin with M = W-tiles.
//   * a block owns up to 128 W-tiles (512 output pixels: G segments of RH whole image rows) x 32 output channels; each of its four
//     waves keeps the accumulators of 32 W-tiles x 32 channels x ALL SIX positions (96 registers) for the whole K, so the output
//     transform happens in registers and nothing meets through LDS;
//   * K is walked in 8-channel chunks.  A chunk stages the segments' raw pixels with their halo ONCE for the three kh taps and
//     without the 1.5x im2col duplication of the d_0 .. d_5 sub-chunks (igemm_wino.hip stages 4.5x the raw bytes), plus the chunk's
//     3 x 6 x 32 x 8 transformed weights: 0.135 LDS-DMA instructions per MFMA instead of 0.375;
//   * LDS image of the pixels: columns are de-interleaved by (column + 1) mod 4, so that raw pixel j of 32 consecutive W-tiles is 32
//     consecutive 32-byte slots (one ds_read_b128 per raw pixel and lane, halves XOR-swizzled by bit 3 of the slot);
//   * two stages, one s_barrier per chunk of 72 MFMAs per wave, two blocks per CU.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "igemm_bf16_ws_tile.h"          // WsDiv / ws_make_div / ws_div

namespace capf {

static constexpr int WW_A_SLOTS = 704;                    // 32-byte pixel slots per stage (22 LDS-DMA instructions): two stages of a block = 80 KiB,
                                                          // two blocks per CU
static constexpr int WW_A_BYTES = WW_A_SLOTS * 32;
static constexpr int WW_W_BYTES = 3 * 6 * 32 * 32;        // one chunk of transformed weights: [kh][position][n][8 channels] fp32
static constexpr int WW_STAGE = WW_A_BYTES + WW_W_BYTES;  // 40960 B

struct WwProblem {
    const float* x;        // [B][H][W][C] fp32
    const float* wp;       // packed: [N / 32 slice][C / 8][kh][position][32 n][2 swizzled halves][4] fp32 (u = G g, BatchNorm folded)
    const float* bias;     // [N] or nullptr
    const float* res;      // [M][ldr] or nullptr
    float* y;              // [M][ldy]
    int B, H, W, C, N, ldy, ldr, relu;
    int RH, G, TPR, TP;    // rows per segment, segments per tile, W-tiles per row, slots per (row, class)
    int tiles;             // W-tiles per block tile = G * RH * TPR (<= 128)
    int SL;                // staged slots = G * (RH + 2) * (W + 2): a row holds its columns by class (column + 1) mod 4 -- TP, TP, TPR, TPR slots
    int RGPI, RG, NSL, tiles_m;
    long M;
    WsDiv d_rowslots, d_tp, d_rh2, d_rgpi, d_per, d_tpr;
};

inline bool ww_plan(int B, int H, int W, int C, int N, WwProblem* p) {
    if (B <= 0 || H <= 0 || W <= 0 || (W & 3) || (C & 7) || (N & 3) || W > 512) return false;
    if ((double)B * H * W * C * 4.0 >= 2.0e9 || (double)B * H * W * N * 4.0 >= 2.0e9) return false;
    const int TPR = W / 4, TP = TPR + 1;
    auto slots = [&](int g, int rh) { return g * (rh + 2) * (W + 2); };
    const int TR = 128 / TPR;                                // rows a block's 128 W-tiles cover
    int RH = 0, G = 1;
    if (TR >= H) {
        RH = H;
        G = TR / H;
        while (G > 1 && slots(G, RH) > WW_A_SLOTS) --G;
        if (slots(1, RH) > WW_A_SLOTS) return false;
    } else {
        for (int r = TR; r >= 1; --r)
            if (H % r == 0 && slots(1, r) <= WW_A_SLOTS) { RH = r; break; }
        if (!RH) return false;
    }
    p->B = B; p->H = H; p->W = W; p->C = C; p->N = N; p->ldy = N; p->ldr = N;
    p->RH = RH; p->G = G; p->TPR = TPR; p->TP = TP;
    p->tiles = G * RH * TPR;
    p->SL = slots(G, RH);
    p->RGPI = H / RH;
    p->RG = B * p->RGPI;
    p->NSL = (N + 31) / 32;
    p->tiles_m = (p->RG + G - 1) / G;
    p->M = (long)B * H * W;
    p->d_rowslots = ws_make_div((unsigned)(W + 2));
    p->d_tp = ws_make_div((unsigned)TP);
    p->d_rh2 = ws_make_div((unsigned)(RH + 2));
    p->d_rgpi = ws_make_div((unsigned)p->RGPI);
    p->d_per = ws_make_div((unsigned)(RH * TPR));
    p->d_tpr = ws_make_div((unsigned)TPR);
    return true;
}

#if defined(__HIP_DEVICE_COMPILE__)
// one tile (logical id `bid` = pixel tile * NSL + slice) with the calling 256-thread block; lds: 2 * WW_STAGE bytes
__device__ __forceinline__ void igemm_wino_ws_tile(const WwProblem& p, const int bid, unsigned char* __restrict__ lds) {
    constexpr int NAS = 6, NWS = 5, NWI = WW_W_BYTES / 1024;      // DMA instructions per wave: pixels, weights; weight instructions in all
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    const int tm = bid / p.NSL, slice = bid - tm * p.NSL;
    const int q0 = tm * p.G;
    const int NCC = p.C >> 3;

    const ws_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7FFFFF00u, 0x00020000);
    const ws_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wp + (size_t)slice * NCC * (WW_W_BYTES / 4)), 0,
                                                             (unsigned)NCC * (unsigned)WW_W_BYTES, 0x00020000);
    // ---- pixel DMA: instruction k covers quads 64 k .. 64 k + 63 = slots 32 k .. 32 k + 31 (a piece index beyond the geometry's
    // last instruction is clamped to it: the same bytes land in the same place twice, so the loop has no branches)
    const int NAI = (2 * p.SL + 63) >> 6;
    unsigned a_voff[NAS];
    int a_k[NAS];
#pragma unroll
    for (int j = 0; j < NAS; ++j) {
        a_k[j] = min(j * 4 + wave, NAI - 1);
        const int qi = a_k[j] * 64 + lane;
        const int slot = qi >> 1;
        const int half = (qi & 1) ^ ((slot >> 3) & 1);
        const int rowi = ws_div(slot, p.d_rowslots), r2 = slot - rowi * (p.W + 2);
        const int cls = r2 < p.TP ? 0 : (r2 < 2 * p.TP ? 1 : (r2 < 2 * p.TP + p.TPR ? 2 : 3));
        const int s = r2 - (cls == 0 ? 0 : (cls == 1 ? p.TP : (cls == 2 ? 2 * p.TP : 2 * p.TP + p.TPR)));
        const int g = ws_div(rowi, p.d_rh2), rr = rowi - g * (p.RH + 2);
        const int q = q0 + g;
        const int b = ws_div(q, p.d_rgpi);
        const int h = (q - b * p.RGPI) * p.RH + rr - 1, col = 4 * s + cls - 1;
        const bool ok = slot < p.SL && q < p.RG && h >= 0 && h < p.H && col >= 0 && col < p.W;
        a_voff[j] = ok ? (unsigned)((((b * p.H + h) * p.W + col) * p.C + half * 4) * 4) : OOB;
    }
    const unsigned w_voff = (unsigned)lane * 16u;
    auto fire_piece = [&](int idx, int stage, int cc) {
        if (idx < NAS) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (ws_lptr_t)(lds + stage * WW_STAGE + a_k[idx] * 1024), 16, a_voff[idx], (unsigned)cc * 32u, 0, 0);
        } else {
            const int k = min((idx - NAS) * 4 + wave, NWI - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (ws_lptr_t)(lds + stage * WW_STAGE + WW_A_BYTES + k * 1024), 16, w_voff,
                                                     (unsigned)cc * (unsigned)WW_W_BYTES + (unsigned)k * 1024u, 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < NAS + NWS; ++i) fire_piece(i, 0, 0);

    // ---- fragment addresses: this lane's W-tile, raw pixel j = 0..5 (column 4 t - 1 + j), row shifted by kh
    int tl = 32 * wave + frow;
    const bool tile_ok = tl < p.tiles;
    if (!tile_ok) tl = 0;
    const int tg = ws_div(tl, p.d_per), trem = tl - tg * p.RH * p.TPR;
    const int tr = ws_div(trem, p.d_tpr), tt = trem - tr * p.TPR;
    unsigned a_addr[3][6];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int rowi = tg * (p.RH + 2) + tr + kh;
            const int cls = j & 3;
            const int slot = rowi * (p.W + 2) + (cls == 0 ? 0 : (cls == 1 ? p.TP : (cls == 2 ? 2 * p.TP : 2 * p.TP + p.TPR))) + tt + (j >> 2);
            a_addr[kh][j] = (unsigned)(slot * 32 + ((fhalf ^ ((slot >> 3) & 1)) << 4));
        }
    const unsigned b_addr = (unsigned)(WW_A_BYTES + frow * 32 + ((fhalf ^ ((frow >> 3) & 1)) << 4));

    ws_f32x16 acc[6];
#pragma unroll
    for (int pq = 0; pq < 6; ++pq)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pq][r] = 0.f;

    auto chunk = [&](auto SC, auto LAST, int cnext) {
        constexpr int S = decltype(SC)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* st = lds + S * WW_STAGE;
        // the raw pixels of tap kh + 1 are requested before tap kh's MFMAs (double-buffered), the weights of a tap at its start: their
        // latency hides behind the tap's input transform, which only needs the pixels
        ws_f32x4 dd[2][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) dd[0][j] = *reinterpret_cast<const ws_f32x4*>(st + a_addr[0][j]);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            ws_f32x4 wf[6];
#pragma unroll
            for (int pq = 0; pq < 6; ++pq) wf[pq] = *reinterpret_cast<const ws_f32x4*>(st + b_addr + (kh * 6 + pq) * 1024);
            if (kh < 2) {
#pragma unroll
                for (int j = 0; j < 6; ++j) dd[(kh + 1) & 1][j] = *reinterpret_cast<const ws_f32x4*>(st + a_addr[kh + 1][j]);
            }
            const ws_f32x4* d = dd[kh & 1];
            // input transform B^T d on this lane's four channels
            ws_f32x4 v[6];
            const ws_f32x4 s12 = d[1] + d[2], d12 = d[1] - d[2], s34 = d[3] + d[4], d43 = d[4] - d[3], d31 = d[3] - d[1], d42 = d[4] - d[2];
            v[0] = 4.f * d[0] - 5.f * d[2] + d[4];
            v[1] = s34 - 4.f * s12;
            v[2] = d43 + 4.f * d12;
            v[3] = d42 + 2.f * d31;
            v[4] = d42 - 2.f * d31;
            v[5] = 4.f * d[1] - 5.f * d[3] + d[5];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int pq = 0; pq < 6; ++pq) acc[pq] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[pq][jj], v[pq][jj], acc[pq], 0, 0, 0);
#if !defined(WW_ABL) || !(WW_ABL & 2)
            if constexpr (!decltype(LAST)::value) {
#pragma unroll
                for (int k = kh; k < NAS + NWS; k += 3) fire_piece(k, S ^ 1, cnext);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        int c = 0;
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        for (; c + 2 < NCC; c += 2) {
            chunk(S0{}, std::false_type{}, c + 1);
            chunk(S1{}, std::false_type{}, c + 2);
        }
        if (c + 2 == NCC) {
            chunk(S0{}, std::false_type{}, c + 1);
            chunk(S1{}, std::true_type{}, NCC);
        } else {
            chunk(S0{}, std::true_type{}, NCC);
        }
    }

    // ---- epilogue: output transform A^T m in registers, then bias / residual / ReLU and 16-byte stores: register group g of a lane
    // = channels slice * 32 + 8 g + 4 fhalf .. + 3 of its W-tile's four pixels
    const ws_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(p.res ? (void*)p.res : (void*)p.y, 0, p.res ? 0x7FFFFF00u : 0u, 0x00020000);
    const ws_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7FFFFF00u, 0x00020000);
    const ws_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)p.bias : (void*)p.y, 0, p.bias ? (unsigned)p.N * 4u : 0u, 0x00020000);
    const int gp = ((q0 + tg) * p.RH + tr) * p.W + 4 * tt;             // flat index of the tile's first output pixel
    const bool seg_ok = tile_ok && (q0 + tg) < p.RG;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = slice * 32 + 8 * g + 4 * fhalf;
        const bool ok = seg_ok && n < p.N;
        const ws_f32x4 bv = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, (unsigned)n * 4u, 0, 0));
        ws_f32x4 m[6];
#pragma unroll
        for (int pq = 0; pq < 6; ++pq) m[pq] = ws_f32x4{acc[pq][4 * g], acc[pq][4 * g + 1], acc[pq][4 * g + 2], acc[pq][4 * g + 3]};
        const ws_f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        ws_f32x4 yq[4];
        yq[0] = (m[0] + s12) + s34;
        yq[1] = d12 + 2.f * d34;
        yq[2] = s12 + 4.f * s34;
        yq[3] = (d12 + 8.f * d34) + m[5];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned off_o = ok ? (unsigned)((gp + q) * p.ldy + n) * 4u : OOB;
            const unsigned off_r = ok ? (unsigned)((gp + q) * p.ldr + n) * 4u : OOB;
#if defined(WW_ABL) && (WW_ABL & 1)
            const unsigned keep = (yq[q][0] == 12345.678f) ? 0u : OOB;           // (ablation: no epilogue traffic, results still "used")
            const ws_f32x4 rv = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, off_r | keep, 0, 0));
#else
            const unsigned keep = 0u;
            const ws_f32x4 rv = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, off_r, 0, 0));
#endif
            ws_f32x4 o = (yq[q] + bv) + rv;
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, o), rs_out, off_o | keep, 0, 0);
        }
    }
}
#endif

}  // namespace capf
