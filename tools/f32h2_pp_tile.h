// EXPERIMENT (not in the product): the software-pipelined, two-stage form of the two-fp16-piece conv tile (csrc/igemm_f32h2_ws_tile.h),
// with knock-out switches and per-segment cycle counters.  Result (EXPERIMENTS.md R5.2): bit-identical, and no faster -- a block alone on
// its CU spends ~4000 cycles per 16-channel chunk in EITHER form (1728 of them MFMA time); the knock-outs show the tap loop itself takes
// ~2200 cycles with or without its MFMAs (fragment reads with ~3 in flight at ~128 cycles each), the pixel loads + block maximum ~860.
#pragma once
#include "igemm_f32h2_ws_tile.h"

namespace capf {

inline constexpr int h2p_lds_bytes(int NS) { return 2 * (H2_A_BYTES + h2_w_bytes(NS)) + 32 + 2 * 64 * 4; }

#ifdef H2_KNOCK
__device__ long long h2_seg_dbg[8];
#endif

#if defined(__HIP_DEVICE_COMPILE__)

// ---- the software-pipelined form of the same tile, for tiles with a long K loop ----------------------------------------------------------
// The tile above has ONE stage: per 16-channel chunk every wave stops at two barriers with the weight DMA's latency, the scale exchange and
// the split between them -- ~4300 cycles per chunk measured for a block that is alone on its CU against 864 cycles of MFMAs (TN = 1), which
// three resident blocks cover only while there are blocks to spare.  A 256-channel conv at batch 64 is 128 blocks of 16 chunks: 33 us of
// such chunks back to back whatever else shares the launch.  This form runs the whole operand path one chunk ahead, under the MFMAs:
//   * two stages (pixel planes + weights: 2 x 44 KiB at TN = 1, 2 x 62 KiB at TN = 2 -- one block per CU, up to 512 registers per lane),
//     ONE barrier per chunk;
//   * during the MFMAs of chunk c: the weights of chunk c + 1 land by LDS-DMA in the other stage; the raw pixels of chunk c + 1 (in
//     registers since chunk c - 1) are scaled, split and written to the other stage's planes, one unit behind every other tap; the raw
//     pixels of chunk c + 2 are requested into the second register set, and their block maximum is published (16 bytes of LDS per
//     parity) before the barrier that ends the chunk -- so the scale of chunk c + 2 is known right after it;
//   * the accumulators move to the next chunk's scale behind the last tap (exact, and only when the scale changes).
// Same arithmetic, same K order, same scales as the single-stage form: bit-identical results (tests/test_gpu_ops.py).
template <int TN>
__device__ __forceinline__ void igemm_f32h2_ws_tile_pp(const H2Problem& q, const int bid, unsigned char* __restrict__ lds) {
    constexpr int NS = 32 * TN;
    constexpr int WP_BYTES = 9 * 32 * 32;
    constexpr int WS_BYTES = 2 * WP_BYTES;
    constexpr int W2_BYTES = TN * WS_BYTES;
    constexpr int ST = H2_A_BYTES + W2_BYTES;              // one stage
    constexpr int NWI = W2_BYTES / 1024;
    constexpr int NWS = (NWI + 3) / 4;
    constexpr int NAU = 4;
    constexpr unsigned OOB = 0x80000000u;
#ifdef H2_KNOCK                                            // tools/f32h2_ws.hip only: 1 no MFMAs, 2 no split, 4 no weight DMA, 8 no fragment reads, 16 no pixel loads / maxima
    constexpr int KN = H2_KNOCK;
#else
    constexpr int KN = 0;
#endif
    const WsProblem& p = q.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    const int NCC = p.C >> 4;
    const int tm = bid / p.NSL, slice = bid - tm * p.NSL;

    const ws_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)q.x, 0, 0x7FFFFF00u, 0x00020000);
    constexpr int HP = H2_HP;
    unsigned a_voff[NAU], a_lds[NAU];
    const int n_units = 2 * ((p.PP + 7) & ~7);
    {
        const int q0 = tm * p.G;
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            const int qi = min(j * 256 + tid, n_units - 1);
            const int half = (qi >> 3) & 1;
            const int px = ((qi >> 4) << 3) | (qi & 7);
            a_lds[j] = (unsigned)(half * HP + px * 16);
            const int g = ws_div(px, p.d_segp), rem = px - g * p.SEGP;
            const int rr = ws_div(rem, p.d_pw), ww = rem - rr * p.PW;
            const int sg = q0 + g;
            const int b = ws_div(sg, p.d_rgpi);
            const int h = (sg - b * p.RGPI) * p.RH + rr - 1, col = ww - 1;
            const bool ok = px < p.PP && sg < p.RG && h >= 0 && h < p.H && col >= 0 && col < p.W;
            a_voff[j] = ok ? (unsigned)((((b * p.H + h) * p.W + col) * p.C + half * 8) * 4) : OOB;
        }
    }
    ws_f32x4 ar[2][NAU][2];                                // two register sets of raw pixels: chunk c in set c & 1
    auto load_a = [&](auto SET, int cc) {
        constexpr int R = decltype(SET)::value;
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            ar[R][j][0] = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, a_voff[j], (unsigned)cc * 64u, 0));
            ar[R][j][1] = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, a_voff[j] + 16u, (unsigned)cc * 64u, 0));
        }
    };
    int* const aux = reinterpret_cast<int*>(lds + 2 * ST);          // [2 parities][4 waves] maxima
    float* const aux_w = reinterpret_cast<float*>(aux + 8);
    float* const aux_b = aux_w + 64;
    auto publish_max = [&](auto SET) {                     // this wave's maximum of the chunk in register set SET -> aux[SET][wave]
        constexpr int R = decltype(SET)::value;
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < NAU; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(ar[R][j][k][e]));
        const int wm = h2_wave_max(m);
        if (lane == 0) aux[R * 4 + wave] = wm;
    };
    auto block_scale_exp = [&](int parity) -> int {
        const ws_u32x4 v = *reinterpret_cast<const ws_u32x4*>(aux + parity * 4);
        const int m = max(max((int)v[0], (int)v[1]), max((int)v[2], (int)v[3]));
        return __builtin_amdgcn_readfirstlane(h2_scale_exp(m));
    };
    auto split_unit = [&](auto SET, int j, float sc, unsigned char* planes) {      // unit j of register set SET -> the two planes at `planes`
        constexpr int R = decltype(SET)::value;
        ws_u32x4 u1, u2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned s1, s2;
            h2_split2(ar[R][j][k >> 1][2 * (k & 1)], ar[R][j][k >> 1][2 * (k & 1) + 1], sc, s1, s2);
            u1[k] = s1; u2[k] = s2;
        }
        *reinterpret_cast<ws_u32x4*>(planes + a_lds[j]) = u1;
        *reinterpret_cast<ws_u32x4*>(planes + 2 * HP + a_lds[j]) = u2;
    };
    const unsigned w_voff = (unsigned)lane * 16u;
    const int nsl32 = (p.N + 31) >> 5;
    const ws_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wp + (size_t)slice * TN * NCC * (WS_BYTES / 2)), 0,
                                                             (unsigned)(nsl32 - slice * TN) * (unsigned)NCC * (unsigned)WS_BYTES, 0x00020000);
    auto fire_w = [&](int cc, int stage) {
#pragma unroll
        for (int i = 0; i < NWS; ++i) {
            const int k = min(i * 4 + wave, NWI - 1);
            const int j = k / 18, rem = k - j * 18;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (ws_lptr_t)(lds + stage * ST + H2_A_BYTES + k * 1024), 16, w_voff,
                                                     (unsigned)(j * NCC + cc) * (unsigned)WS_BYTES + (unsigned)rem * 1024u, 0, 0);
        }
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    load_a(R0{}, 0);
    if (NCC > 1) load_a(R1{}, 1);
    if (tid < NS) {
        const int n = slice * NS + tid;
        aux_w[tid] = n < ((p.N + 31) & ~31) ? q.winv[n] : 1.f;
        aux_b[tid] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }

    int pl_i[2];
    {
        const int in_g1 = (frow >= 4 && frow < 12) || (frow >= 16 && frow < 20) || frow >= 28;
        const int pos = in_g1 ? (frow < 12 ? frow - 4 : (frow < 20 ? frow - 8 : frow - 16))
                              : (frow < 4 ? frow : (frow < 16 ? frow - 8 : frow - 12));
        if ((p.W & 15) == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) pl_i[i] = ((2 * wave + i) * 2 + in_g1) * 16 + pos;
        } else {
            unsigned short* tab = reinterpret_cast<unsigned short*>(lds);
            unsigned short* ovf = tab + 256;
            int* cnt = reinterpret_cast<int*>(lds + 1024);
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
            if ((tid >> 4) >= cnt[tid & 15]) {
                const int e = atomicAdd(&cnt[17], 1);
                tab[tid] = e < cnt[16] ? ovf[e] : (unsigned short)0xFFFFu;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) pl_i[i] = tab[((2 * wave + i) * 2 + in_g1) * 16 + pos];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    fire_w(0, 0);
    unsigned a_addr[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int pl = pl_i[i];
        if (pl >= p.P) { pl = 0; pl_i[i] = 0x7FFF; }
        const int g = ws_div(pl, p.d_rhw), rem = pl - g * p.RHW;
        const int r = ws_div(rem, p.d_w), w = rem - r * p.W;
        const int pix0 = (g * (p.RH + 2) + r) * p.PW + w;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) a_addr[i][kh] = (unsigned)((pix0 + kh * p.PW) * 16 + fhalf * HP);
    }
    const unsigned b_addr = (unsigned)(H2_A_BYTES + frow * 32 + ((fhalf ^ ((frow >> 3) & 1)) << 4));

    const int Mi = (int)p.M;
    const ws_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(q.res ? (void*)q.res : (void*)q.y, 0, q.res ? 0x7FFFFF00u : 0u, 0x00020000);
    const ws_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)q.y, 0, 0x7FFFFF00u, 0x00020000);
    const int gp0 = tm * p.G * p.RHW;
    constexpr int PW_[3] = {0, 1, 0};
    constexpr int PA_[3] = {1, 0, 0};

    // ---- chunk 0: maximum -> scale -> split into stage 0; chunk 1's maximum goes out with it
    publish_max(R0{});
    if (NCC > 1) publish_max(R1{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int sb = block_scale_exp(0);
    {
        const float sc = __int_as_float(sb << 23);
#pragma unroll
        for (int j = 0; j < NAU; ++j) split_unit(R0{}, j, sc, lds);
    }
    ws_f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    auto piece_off = [&](int i, int j, int g, int ld) -> unsigned {
        const int pl = pl_i[i], n = slice * NS + j * 32 + 8 * g + 4 * fhalf;
        const int gp = gp0 + pl;
        return (pl < p.P && gp < Mi && n < p.N) ? (unsigned)(gp * ld + n) * 4u : OOB;
    };
    ws_f32x4 rr[2][TN][4];
    ws_f16x8 af[2][2][2], bfr[2][2][TN];
    // chunk cc from stage S (= cc & 1): its MFMAs, and under them everything chunk cc + 1 and cc + 2 need
#ifdef H2_KNOCK
    long long seg[6] = {0, 0, 0, 0, 0, 0};               // (KN & 32) cycles of wave 0 per segment of a chunk: issue, taps, rescale + maximum, wait, barrier, chunks
#define H2_T(k) do { if constexpr (KN & 32) { const long long t_ = clock64(); seg[k] += t_ - t_last; t_last = t_; } } while (0)
    long long t_last = clock64();
#else
#define H2_T(k) do { } while (0)
#endif
    auto chunk = [&](auto SC, auto LASTC, int cc) {
        constexpr int S = decltype(SC)::value;
        constexpr bool has1 = !decltype(LASTC)::value;      // a chunk follows
        using RS = std::integral_constant<int, S>;          // register set of chunk cc (already split), free for chunk cc + 2
        using RN = std::integral_constant<int, S ^ 1>;      // register set of chunk cc + 1
        const unsigned char* st = lds + S * ST;
        unsigned char* nx = lds + (S ^ 1) * ST;
        const bool has2 = cc + 2 < NCC;
        int sn = sb;
        float scn = 0.f;
        if constexpr (has1) {
            if constexpr (!(KN & 4)) fire_w(cc + 1, S ^ 1);
            if constexpr (!(KN & 16)) sn = block_scale_exp(S ^ 1);
            scn = __int_as_float(sn << 23);
            if constexpr (!(KN & 16)) { if (has2) load_a(RS{}, cc + 2); }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        rr[i][j][g] = __builtin_bit_cast(ws_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, piece_off(i, j, g, p.ldr), 0, 0));
        }
        auto read_frags = [&](int t, int buf) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int pw = PW_[o], pa = PA_[o];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[buf][pw][j] = __builtin_bit_cast(ws_f16x8, *reinterpret_cast<const ws_f32x4*>(st + b_addr + j * WS_BYTES + pw * WP_BYTES + t * 1024));
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    af[buf][pa][i] = __builtin_bit_cast(ws_f16x8, *reinterpret_cast<const ws_f32x4*>(st + pa * 2 * HP + (t % 3) * 16 + a_addr[i][t / 3]));
            }
        };
        H2_T(0);
        if (!(KN & 8) || cc == 0) read_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if constexpr (!(KN & 8)) { if (t < 8) read_frags(t + 1, (t + 1) & 1); }
            const bool sp = has1 && (t & 1) == 0 && t < 8 && !(KN & 2);   // taps 0, 2, 4, 6 carry the split of units 0 .. 3 (compile time: t is unrolled)
            if (sp) split_unit(RN{}, t >> 1, scn, nx);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if constexpr (KN & 1) acc[i][j][k] += (float)bfr[(KN & 8) ? 0 : (t & 1)][PW_[k]][j][0] * (float)af[(KN & 8) ? 0 : (t & 1)][PA_[k]][i][0];
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bfr[(KN & 8) ? 0 : (t & 1)][PW_[k]][j], af[(KN & 8) ? 0 : (t & 1)][PA_[k]][i], acc[i][j], 0, 0, 0);
                    }
            // issue order inside the tap: one fragment read (of the next tap) and a few of the split's VALU instructions behind every MFMA
            if (t < 8) {
#pragma unroll
                for (int x = 0; x < 6 * TN; ++x) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (x < 4 + 2 * TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (sp) __builtin_amdgcn_sched_group_barrier(0x002, (24 + 6 * TN - 1) / (6 * TN), 0);
                }
                if (sp) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        H2_T(1);
        if constexpr (has1) {
            if (sn != sb) {                                // (block-uniform) the accumulators move to chunk cc + 1's scale: exact
                const float f = __int_as_float((127 + sn - sb) << 23);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] *= f;
                sb = sn;
            }
            if constexpr (!(KN & 16)) { if (has2) publish_max(RS{}); }       // chunk cc + 2 has had this chunk's MFMAs to arrive
            H2_T(2);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            H2_T(3);
            __builtin_amdgcn_s_barrier();                  // stage S ^ 1 complete (planes, weights), stage S read by everybody
            H2_T(4);
        }
    };
    {
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        int c = 0;
        for (; c + 2 < NCC; c += 2) {
            chunk(S0{}, std::false_type{}, c);
            chunk(S1{}, std::false_type{}, c + 1);
        }
        if (c + 2 == NCC) {
            chunk(S0{}, std::false_type{}, c);
            chunk(S1{}, std::true_type{}, c + 1);
        } else {
            chunk(S0{}, std::true_type{}, c);
        }
    }

#ifdef H2_KNOCK
    if constexpr (KN & 32) {
        if (blockIdx.x == 8 && tid == 0) { for (int k = 0; k < 5; ++k) h2_seg_dbg[k] = seg[k]; h2_seg_dbg[5] = NCC; }
    }
#endif
    const float inv_s = __int_as_float((254 - sb) << 23);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ws_f32x4 wv = *reinterpret_cast<const ws_f32x4*>(aux_w + j * 32 + 8 * g + 4 * fhalf);
            const ws_f32x4 bv = *reinterpret_cast<const ws_f32x4*>(aux_b + j * 32 + 8 * g + 4 * fhalf);
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[e] *= inv_s;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ws_f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = fmaf(acc[i][j][4 * g + e], wv[e], bv[e] + rr[i][j][g][e]);
                    o[e] = p.relu ? fmaxf(t, 0.f) : t;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, o), rs_out, piece_off(i, j, g, p.ldy), 0, 0);
            }
        }
}

#undef H2_T
#endif

}  // namespace capf
