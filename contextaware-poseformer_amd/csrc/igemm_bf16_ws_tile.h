// "2-D halo" tile of the bf16 3x3 / stride-1 / pad-1 convolution (BasicBlock convs of HRNet, pose_hrnet.py:66-95; the 3x3 of the
// ResNet / refine bottlenecks, networks/resnet.py:58-93): device code shared by csrc/igemm_bf16_ws.hip and the stand-alone
// harness tools/bf16_ws.hip.
//
// What bounded the row-halo tile (profiles/r03_pmc_bf16_rh_level.txt) was not a roof but the operand path: per
// 192-deep superchunk a block re-staged 3 x 64 weight rows for only 126 pixels -- 2.4 MFMAs per KiB of LDS-DMA, and an LDS-DMA
// instruction costs the issuing wave 100-185 cycles against 32 per MFMA.  This tile turns the loops around:
//   * a block owns up to 256 output pixels (G segments of RH whole image rows, possibly of several images) x NS = 32 TN output
//     channels and keeps ALL of their accumulators in registers for the whole K = 9 Cin (TN = 3: 96 fp32 registers per lane);
//   * K is walked in 16-channel chunks.  A chunk stages the segments WITH their 1-pixel halo -- (RH + 2) x (W + 2) pixels x
//     32 B, zero filled by the hardware outside the image -- ONCE for all nine taps (the row-halo tile: once per kh), plus the
//     chunk's 9 x NS x 16 weights: 5.6 MFMAs per staged KiB, 180 matrix cycles per LDS-DMA instruction and wave;
//   * two stages, one s_barrier per chunk (54 MFMAs per wave at TN = 3), the next chunk's DMA instructions issued one per tap
//     behind the MFMAs; two blocks per CU (<= 256 registers, 2 x 40 KiB of LDS each) cover each other's prologue / epilogue;
//   * a fragment is one ds_read_b128 at `register + immediate`: the 9 x 2 tap addresses of a lane are computed once per tile.
//     LDS image: pixel pitch 32 B, the two 16-byte halves of pixel px swapped when bit 3 of px is set (conflict-free b128 reads
//     for 16 consecutive pixels); the packed weights carry the same swizzle, so their DMA is a linear copy.
//   * epilogue as in the row-halo tile: accumulators start at the bias, 32 x 32 blocks transposed through per-wave LDS scratch,
//     residual rows prefetched before the last chunk, 16-byte raw-buffer stores.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace capf {

struct WsDiv {
    unsigned mul, shift;
};
inline WsDiv ws_make_div(unsigned d) {          // n / d for 0 <= n < 2^31: (umulhi(n, mul) + n) >> shift
    WsDiv f{0u, 0u};
    if (d <= 1) return f;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.shift = s;
    f.mul = (unsigned)((((1ull << s) - d) << 32) / d + 1);
    return f;
}

static constexpr int WS_A_BYTES = 13312;        // staged pixels x 32 B: at most 416 pixels = 13 LDS-DMA instructions
static constexpr int WS_MAX_PP = WS_A_BYTES / 32;
static constexpr int WS_MAX_P = 256;            // output pixels per tile (8 MFMA pixel blocks, two per wave)
inline constexpr int ws_stage_bytes(int NS) { return WS_A_BYTES + 9 * NS * 32; }

struct WsProblem {
    const unsigned short* x;      // [B][H][W][C] bf16
    const unsigned short* wp;     // packed by ws_pack: [N slice][C / 16][tap][NS][2 swizzled halves][8] bf16
    const float* bias;            // [N] fp32 or nullptr
    const unsigned short* res;    // [M][ldr] bf16 or nullptr
    unsigned short* y;            // [M][ldy] bf16
    int B, H, W, C, N;
    int ldy, ldr;                 // row pitch (elements) of y / res
    int relu;
    int RH, G;                    // a tile = G segments of RH rows (a segment lies inside one image: RH divides H)
    int P, PP, PW, SEGP;          // output pixels per tile, staged pixels per tile, W + 2, (RH + 2) * (W + 2)
    int RHW;                      // RH * W
    int RGPI, RG;                 // segments ("row groups") per image, in total
    int NS, NSL;                  // output channels per tile (32 / 64 / 96), slices
    int tiles_m;                  // ceil(RG / G)
    long M;                       // B * H * W
    WsDiv d_segp, d_pw, d_rgpi, d_rhw, d_w;
};

// tile geometry for a conv; false = this tile cannot take it (the caller keeps the row-halo / direct kernel)
// (tile_relative: the kernel addresses x / y / res from per-tile bases -- only the pixel count has to fit 31 bits)
inline bool ws_plan(int B, int H, int W, int C, int N, WsProblem* p, bool tile_relative = false) {
    if (B <= 0 || H <= 0 || W <= 0 || C % 16 != 0 || N % 8 != 0 || W > WS_MAX_P) return false;
    if (tile_relative) {
        if ((double)B * H * W >= 2.0e9) return false;
    } else if ((double)B * H * W * C * 2.0 >= 2.0e9 || (double)B * H * W * N * 2.0 >= 2.0e9) {
        return false;
    }
    const int TR = WS_MAX_P / W;
    int RH = 0, G = 1;
    if (TR >= H) {
        RH = H;
        G = TR / H;
        while (G > 1 && G * (RH + 2) * (W + 2) > WS_MAX_PP) --G;
        if ((RH + 2) * (W + 2) > WS_MAX_PP) return false;
    } else {
        for (int r = TR; r >= 1; --r)
            if (H % r == 0 && (r + 2) * (W + 2) <= WS_MAX_PP) { RH = r; break; }
        if (!RH) return false;
    }
    p->B = B; p->H = H; p->W = W; p->C = C; p->N = N;
    p->ldy = N; p->ldr = N;
    p->RH = RH; p->G = G;
    p->P = G * RH * W;
    p->PW = W + 2;
    p->SEGP = (RH + 2) * (W + 2);
    p->PP = G * p->SEGP;
    p->RHW = RH * W;
    p->RGPI = H / RH;
    p->RG = B * p->RGPI;
    p->NS = N % 96 == 0 ? 96 : (N <= 32 ? 32 : 64);
    p->NSL = (N + p->NS - 1) / p->NS;
    p->tiles_m = (p->RG + G - 1) / G;
    p->M = (long)B * H * W;
    p->d_segp = ws_make_div((unsigned)p->SEGP);
    p->d_pw = ws_make_div((unsigned)p->PW);
    p->d_rgpi = ws_make_div((unsigned)p->RGPI);
    p->d_rhw = ws_make_div((unsigned)p->RHW);
    p->d_w = ws_make_div((unsigned)W);
    return true;
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef float ws_f32x4 __attribute__((ext_vector_type(4)));
typedef float ws_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ws_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* ws_lptr_t;
typedef __amdgpu_buffer_rsrc_t ws_rsrc_t;

__device__ __forceinline__ int ws_div(int n, WsDiv d) { return (int)((__umulhi((unsigned)n, d.mul) + (unsigned)n) >> d.shift); }
__device__ __forceinline__ unsigned ws_pack2(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{lo, hi}, bf2));
}

// one tile (logical id `bid` = pixel tile * NSL + slice) with the calling 256-thread block; lds: 2 * ws_stage_bytes(32 TN) bytes
template <int TN>
__device__ __forceinline__ void igemm_bf16_ws_tile(const WsProblem& p, const int bid, unsigned char* __restrict__ lds) {
    constexpr int NS = 32 * TN;
    constexpr int W_BYTES = 9 * NS * 32;
    constexpr int ST = WS_A_BYTES + W_BYTES;
    constexpr int NWI = W_BYTES / 1024;                    // weight DMA instructions per chunk: 27 / 18 / 9
    constexpr int NWS = (NWI + 3) / 4;                     // ... per wave
    constexpr int NAS = 4;                                 // pixel DMA instructions per wave (13 in all at most)
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    const int tm = bid / p.NSL, slice = bid - tm * p.NSL;
    const int q0 = tm * p.G;                               // first segment of the tile
    const int NCC = p.C >> 4;

    // ---- operand fetch: LDS-DMA on block-uniform descriptors, 32-bit byte offsets, out-of-range offset = hardware zero fill
    const ws_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7FFFFF00u, 0x00020000);
    const ws_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wp + (size_t)slice * NCC * (W_BYTES / 2)), 0,
                                                             (unsigned)NCC * (unsigned)W_BYTES, 0x00020000);
    // Every wave issues the same number of DMA instructions per chunk (4 pixel + NWS weight pieces); a piece index beyond the
    // geometry's last one is CLAMPED to it -- the same bytes land in the same place twice -- so the loop has no branches.
    const int NAI = (2 * p.PP + 63) >> 6;                  // pixel DMA instructions of this geometry (<= 13)
    unsigned a_voff[NAS];
    int a_k[NAS];
#pragma unroll
    for (int j = 0; j < NAS; ++j) {
        a_k[j] = min(j * 4 + wave, NAI - 1);
        const int qi = a_k[j] * 64 + lane;                // quad slot of the stage: pixel qi / 2, physical half qi & 1
        const int px = qi >> 1;
        const int half = (qi & 1) ^ ((px >> 3) & 1);
        const int g = ws_div(px, p.d_segp), rem = px - g * p.SEGP;
        const int rr = ws_div(rem, p.d_pw), ww = rem - rr * p.PW;
        const int q = q0 + g;
        const int b = ws_div(q, p.d_rgpi);
        const int h = (q - b * p.RGPI) * p.RH + rr - 1, col = ww - 1;
        const bool ok = px < p.PP && q < p.RG && h >= 0 && h < p.H && col >= 0 && col < p.W;
        a_voff[j] = ok ? (unsigned)((((b * p.H + h) * p.W + col) * p.C + half * 8) * 2) : OOB;
    }
    const unsigned w_voff = (unsigned)lane * 16u;
    auto fire_piece = [&](int idx, int stage, int cc) {    // DMA instruction idx (0 .. NAS + NWS - 1) of chunk cc into `stage`
        if (idx < NAS) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (ws_lptr_t)(lds + stage * ST + a_k[idx] * 1024), 16, a_voff[idx], (unsigned)cc * 32u, 0, 0);
        } else {
            const int k = min((idx - NAS) * 4 + wave, NWI - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (ws_lptr_t)(lds + stage * ST + WS_A_BYTES + k * 1024), 16, w_voff,
                                                     (unsigned)cc * (unsigned)W_BYTES + (unsigned)k * 1024u, 0, 0);
        }
    };
    // chunk 0 goes out before anything else is computed
#pragma unroll
    for (int i = 0; i < NAS + NWS; ++i) fire_piece(i, 0, 0);

    // ---- accumulators start at the bias (register 4 g + e of channel block j = channel slice * NS + 32 j + 8 g + 4 fhalf + e)
    const ws_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)p.bias : (void*)p.y, 0, p.bias ? (unsigned)p.N * 4u : 0u, 0x00020000);
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

    // ---- fragment addresses: pixel block i of this wave = tile pixels (2 wave + i) * 32 + frow, tap (kh, kw)
    unsigned a_addr[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int pl = (2 * wave + i) * 32 + frow;
        if (pl >= p.P) pl = 0;                             // (idle rows of a ragged geometry: computed, never stored)
        const int g = ws_div(pl, p.d_rhw), rem = pl - g * p.RHW;
        const int r = ws_div(rem, p.d_w), w = rem - r * p.W;
        const int pix0 = (g * (p.RH + 2) + r) * p.PW + w;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int pix = pix0 + (t / 3) * p.PW + (t % 3);
            a_addr[i][t] = (unsigned)(pix * 32 + ((fhalf ^ ((pix >> 3) & 1)) << 4));
        }
    }
    const unsigned b_addr = (unsigned)(WS_A_BYTES + frow * 32 + ((fhalf ^ ((frow >> 3) & 1)) << 4));

    // ---- residual rows in the coalesced epilogue's layout (lane = 8 channels of one row), requested before the last chunk.
    // `elane` is laundered through an empty asm so that the compiler cannot hoist the 24 piece offsets above the K loop and
    // keep them in registers through it.
    int elane = lane;
    auto piece_rc = [&](int& er_, int& ec_) { asm volatile("" : "+v"(elane)); er_ = elane >> 2; ec_ = (elane & 3) * 8; };
    const int gp0 = q0 * p.RHW;                            // first flat output pixel of the tile: tile pixel pl is flat pixel gp0 + pl
    const int Mi = (int)p.M;                               // (ws_plan: M * N * 2 < 2^31, so 32-bit offsets throughout)
    const ws_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(p.res ? (void*)p.res : (void*)p.y, 0, p.res ? 0x7FFFFF00u : 0u, 0x00020000);
    const ws_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7FFFFF00u, 0x00020000);
    auto piece_off = [&](int i, int j, int h, int ld, int er, int ec) -> unsigned {
        const int pl = (2 * wave + i) * 32 + h * 16 + er, n = slice * NS + j * 32 + ec;
        const int gp = gp0 + pl;
        return (pl < p.P && gp < Mi && n < p.N) ? (unsigned)(gp * ld + n) * 2u : OOB;
    };
    // (pixel block 0's rows before the last chunk, block 1's behind it: all twelve pieces that early do not fit in 256 registers)
    ws_u32x4 rr[2][TN][2];
    auto prefetch_residual = [&](int i, int j0, int j1) {   // channel blocks j0 .. j1 - 1 of pixel block i
        int er, ec;
        piece_rc(er, ec);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (j >= j0 && j < j1) rr[i][j][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, piece_off(i, j, h, p.ldr, er, ec), 0, 0);
    };
    // TN = 3 runs at the 256-register limit of two waves per SIMD: more than one channel block requested before the last chunk
    // and the compiler spills the loaded rows (load, wait, scratch store -- worse than no prefetch)
    constexpr int EARLY = TN == 3 ? 1 : TN;
    ws_bf16x8 af[2][2], bfr[2][TN];
    auto chunk = [&](auto SC, auto LAST, int cnext) {      // multiply the chunk staged in stage S; fire chunk `cnext` into the other
        constexpr int S = decltype(SC)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the chunk has landed ...
        __builtin_amdgcn_s_barrier();                      // ... everybody's has, and everybody is done reading the other stage
        if constexpr (decltype(LAST)::value) prefetch_residual(0, 0, EARLY);   // (behind the wait: its latency hides under this chunk's MFMAs)
        const unsigned char* st = lds + S * ST;
        auto read_frags = [&](int t, int buf) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[buf][i] = __builtin_bit_cast(ws_bf16x8, *reinterpret_cast<const ws_f32x4*>(st + a_addr[i][t]));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[buf][j] = __builtin_bit_cast(ws_bf16x8, *reinterpret_cast<const ws_f32x4*>(st + b_addr + (t * NS + j * 32) * 32));
        };
        read_frags(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t < 8) read_frags(t + 1, (t + 1) & 1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[t & 1][j], af[t & 1][i], acc[i][j], 0, 0, 0);
            if constexpr (!decltype(LAST)::value) {        // (the last chunk fires nothing: the idle stage becomes the epilogue's scratch)
                // one or two of the next chunk's pieces behind every tap (all of them behind the first five / three taps, so that the
                // last has longer to land before the next chunk's wait: measured equal, 210.5 / 215.9 vs 209.5 us per HRNet-48 level)
                fire_piece(t, S ^ 1, cnext);
                if (t + 9 < NAS + NWS) fire_piece(t + 9, S ^ 1, cnext);
            }
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
    prefetch_residual(0, EARLY, TN);
    if constexpr (TN < 3) prefetch_residual(1, 0, TN);     // (TN = 3: behind pixel block 0's stores, when its 48 accumulators are free)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // every wave has read the last stage: the scratch below overlays it

    // ---- epilogue: 32 x 32 fp32 blocks transposed through 4.5 KiB of per-wave scratch, 8 channels of a row per lane
    constexpr int EPS = 36;
    int er, ec;
    piece_rc(er, ec);
    float* ep = reinterpret_cast<float*>(lds) + wave * (32 * EPS);
    auto finish = [&](float t) { return p.relu ? fmaxf(t, 0.f) : t; };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if constexpr (TN == 3) { if (i == 1) prefetch_residual(1, 0, TN); }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<ws_f32x4*>(&ep[frow * EPS + 8 * g + 4 * fhalf]) =
                    ws_f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = h * 16 + er;
                const ws_f32x4 x0 = *reinterpret_cast<const ws_f32x4*>(&ep[row * EPS + ec]);
                const ws_f32x4 x1 = *reinterpret_cast<const ws_f32x4*>(&ep[row * EPS + ec + 4]);
                ws_u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned rw = rr[i][j][h][q];
                    const float xa = q < 2 ? x0[2 * q] : x1[2 * q - 4], xb = q < 2 ? x0[2 * q + 1] : x1[2 * q - 3];
                    o[q] = ws_pack2(finish(xa + __uint_as_float(rw << 16)), finish(xb + __uint_as_float(rw & 0xFFFF0000u)));
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, piece_off(i, j, h, p.ldy, er, ec), 0, 0);
            }
        }
    }
}

#endif

}  // namespace capf

