// The FIRST bottleneck of a `layer1` (ResNet-50 of CPN: networks/resnet.py:58-93 with the projection shortcut of :119-133; HRNet:
// pose_hrnet.py:98-136, 321-333) under compute_dtype = bf16 as ONE persistent kernel:
//     t1 = relu(bn1(conv1 x))        1x1,  64 ->  64
//     t2 = relu(bn2(conv2 t1))       3x3,  64 ->  64, pad 1
//     r  = bn_d(downsample x)        1x1,  64 -> 256      (rounded to bf16 like the tensor the unfused plan stores)
//     y  = relu(bn3(conv3 t2) + r)   1x1,  64 -> 256
// Unfused, the block is five launches that move 2.2 KB per pixel through the CUs (conv1 + downsample read x and write t1 and r, conv2 reads
// t1 with its halo and writes t2, conv3 reads t2 and r and writes y); here x is read once with a one-pixel halo and y written once: 0.2 + 0.5 KB
// per pixel, t1 / t2 / r never leave the CU.
//
// Tile: 8 x 8 output pixels of one frame = 64 pixels = two 32-pixel MFMA blocks; conv1 is recomputed on the 10 x 10 halo (100 pixels = four
// blocks: one per wave; K = 64 makes that cheap: 8 MFMAs per wave).  One 256-thread block per CU, persistent over the tiles of its XCD's
// contiguous eighth (neighbouring tiles share halo pixels through that XCD's L2):
//   * LDS, 156.5 KiB: W2 (72 KiB, nine [64][64] sub-chunks, one per tap), W3 and Wd (32 KiB each) staged ONCE per block in the 128-byte-row /
//     quad-XOR image of the bf16 tiles; 18 KiB of work area (t1 halo tile 12.5 KiB; t2 8 KiB over it once conv2 is done; the per-wave
//     transpose scratch of the coalesced store behind that); the four bias vectors.  W1 (8 KiB) lives in 32 registers per lane.
//   * phase A: a wave takes 32 halo pixels through conv1 -- pixel fragments straight from global memory, requested a tile ahead -- and writes
//     relu(. + b1) as bf16 into the t1 tile, ZERO where the halo pixel lies outside the image (the 3x3 conv pads t1, not x);
//   * phase B: wave (pixel block, 32-channel half) runs conv2 as 4 chunks x 9 taps of 16-deep MFMA steps from LDS; t2 crosses LDS so that
//   * phase C: wave (pixel block, 128-channel half) has all 64 channels of its pixels: conv3 from t2 and the downsample from the tile's own
//     x pixels (fragments from global memory, L1 / L2 hits) into separate accumulators (every accumulator starts at its bias, as the 2-D halo
//     tile's do); y = relu(acc3 + bf16(acc_d)), transposed
//     per 32 x 32 block through LDS so that a store instruction writes 16 rows x 64 contiguous bytes.
// TAP = true additionally stores t1 (the tile's own 64 pixels), t2 and r to the buffers the unfused ops would have written: the layer-wise
// parity test recomputes every stage from the operands the kernel itself produced, and a second test holds the product kernel's y to the
// TAP kernel's bit for bit (tests/test_gpu_bneck.py).
#include "kernels.h"
#ifndef BN_EXP
#define BN_EXP 0               // (tools/ab_define.sh knock-outs: timing only)
#endif

namespace capf {

typedef float bn_f32x4 __attribute__((ext_vector_type(4)));
typedef float bn_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned bn_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bn_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t bn_rsrc_t;

[[maybe_unused]] static constexpr int BN_W2_OFF = 0, BN_W3_OFF = 73728, BN_WD_OFF = 106496, BN_WK_OFF = 139264;
// (work area: t1 12800 B | t2 8192 B + 4 x 2560 B of scratch.  A scratch row = 64 B of bf16, its 16-byte column g kept at g ^ ((row >> 2) & 3): the
// accumulator-layout writes -- 16 lanes = 16 rows, one column -- and the coalesced reads -- a lane group = four rows x four columns -- both touch 16
// different bank quads; an 80-byte pitch served the writes and put three rows of a read group on the same banks)
[[maybe_unused]] static constexpr int BN_T2_BYTES = 8192, BN_EP_PITCH = 64, BN_EP_BYTES = 2560;
static constexpr int BN_WK_BYTES = BN_T2_BYTES + 4 * BN_EP_BYTES;
static constexpr int BN_B_OFF = BN_WK_OFF + BN_WK_BYTES;
static constexpr int BN_LDS_BYTES = BN_B_OFF + (64 + 64 + 256 + 256) * 4;
static_assert(BN_WK_BYTES >= 100 * 128 && BN_LDS_BYTES <= 160 * 1024, "bottleneck LDS");

struct Bneck0Args {
    const unsigned short* x;                  // [B][H][W][64] bf16
    unsigned short* y;                        // [B][H][W][256]
    const unsigned short *w1, *w2, *w3, *wd;  // plain packs [64][64], [64][9 * 64] (k = (kh, kw, ci)), [256][64], [256][64]
    const float *b1, *b2, *b3, *bd;
    unsigned short *t1, *t2, *r;              // TAP: [B][H][W][64], [B][H][W][64], [B][H][W][256]
    int B, H, W, tiles_x, tiles_pf, ntiles;
};

// DS = true: the first bottleneck (64 -> 64 -> 64 -> 256, projection shortcut, W1 in registers, Wd in LDS).
// DS = false: an IDENTITY bottleneck (256 -> 64 -> 64 -> 256, y = relu(conv3 + x); networks/resnet.py:58-93 without downsample, pose_hrnet.py:98-136):
// W1 (32 KiB, four [64][64] sub-chunks) takes Wd's place in LDS, conv1 reads 256-channel halo pixels (16 fragment loads per lane, a tile ahead), the
// residual rows arrive in the coalesced layout (lane = 8 channels of a row) and cross the wave's scratch into the accumulator layout.
template <bool TAP, bool DS>
__global__ __launch_bounds__(256, 1) void bneck_bf16_kernel(Bneck0Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char bn_lds[];
    constexpr unsigned OOB = 0x80000000u;
    unsigned char* const W2s = bn_lds + BN_W2_OFF;
    unsigned char* const W3s = bn_lds + BN_W3_OFF;
    unsigned char* const Wds = bn_lds + BN_WD_OFF;
    unsigned char* const wk = bn_lds + BN_WK_OFF;
    float* const B1s = reinterpret_cast<float*>(bn_lds + BN_B_OFF);
    float* const B2s = B1s + 64;
    float* const B3s = B2s + 64;
    float* const BDs = B3s + 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5, fsw = (frow >> 1) & 7;

    // ---- weights and biases -> LDS, once.  Row n of a [.][64] sub-chunk = 128 bytes = 8 quads of 8 k; quad q sits at q ^ ((n >> 1) & 7)
    for (int i = tid; i < 64 * 72; i += 256) {
        const int n = i / 72, q = i - n * 72, tap = q >> 3, qq = q & 7;
        *reinterpret_cast<bn_u32x4*>(W2s + (tap * 64 + n) * 128 + ((qq ^ ((n >> 1) & 7)) * 16)) = *reinterpret_cast<const bn_u32x4*>(a.w2 + (size_t)n * 576 + q * 8);
    }
    constexpr int CX = DS ? 64 : 256;         // channels of x
    constexpr int NST = CX / 16;              // 16-deep k-steps of conv1
    for (int i = tid; i < 256 * 8; i += 256) {
        const int n = i >> 3, q = i & 7;
        *reinterpret_cast<bn_u32x4*>(W3s + n * 128 + ((q ^ ((n >> 1) & 7)) * 16)) = *reinterpret_cast<const bn_u32x4*>(a.w3 + (size_t)n * 64 + q * 8);
        if (DS) *reinterpret_cast<bn_u32x4*>(Wds + n * 128 + ((q ^ ((n >> 1) & 7)) * 16)) = *reinterpret_cast<const bn_u32x4*>(a.wd + (size_t)n * 64 + q * 8);
    }
    if (!DS) {                                // W1 [64][256] -> sub-chunk k / 64: [64][64], in Wd's place
        for (int i = tid; i < 64 * 32; i += 256) {
            const int n = i >> 5, q = i & 31, sub = q >> 3, qq = q & 7;
            *reinterpret_cast<bn_u32x4*>(Wds + (sub * 64 + n) * 128 + ((qq ^ ((n >> 1) & 7)) * 16)) = *reinterpret_cast<const bn_u32x4*>(a.w1 + (size_t)n * 256 + q * 8);
        }
    }
    if (tid < 64) { B1s[tid] = a.b1[tid]; B2s[tid] = a.b2[tid]; }
    B3s[tid] = a.b3[tid];
    if (DS) BDs[tid] = a.bd[tid];
    [[maybe_unused]] bn_u32x4 w1f[2][4];      // DS: conv1's weights as MFMA fragments: row 32 j + frow, k = 16 st + 8 fhalf .. + 7
    if constexpr (DS) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int st = 0; st < 4; ++st) w1f[j][st] = *reinterpret_cast<const bn_u32x4*>(a.w1 + (size_t)(32 * j + frow) * 64 + st * 16 + fhalf * 8);
    }
    __syncthreads();

    // ---- this lane's pixels (the same for every tile).  Phase A: halo slot s = 32 wave + frow -> (s / 10, s % 10) of the 10 x 10 halo tile
    const int s_a = 32 * wave + frow;
    const bool slot_ok = s_a < 100;
    const int a_dh = s_a / 10 - 1, a_dw = s_a % 10 - 1;
    [[maybe_unused]] const bool a_interior = slot_ok && a_dh >= 0 && a_dh < 8 && a_dw >= 0 && a_dw < 8;
    const unsigned t1_wr = (unsigned)(s_a * 128 + fhalf * 8);
    // quad swizzle of the t1 tile: halo pixel (r, c) keeps quad q at q ^ ((c >> 1) + 4 r) -- ds_read_b128 is served in lane groups {0-3, 12-15,
    // 20-27} / {4-11, 16-19, 28-31}: with lanes = (row frow >> 3, column frow & 7) of a 4 x 8 pixel block a group is four 4-pixel runs of four
    // different halo rows, and this choice gives its 16 lanes 16 different bank quads for every tap (the linear-pixel swizzle of the GEMM tiles put
    // three of them on the same banks: SQ_LDS_BANK_CONFLICT 38 % of the LDS cycles, profiles/r06_pmc_bneck0.txt)
    const int t1_sw = ((a_dw + 1) >> 1) + 4 * (a_dh + 1);
    // phases B / C: pixel block rb = wave >> 1 (tile rows 4 rb .. + 3), lane = (row frow >> 3, column frow & 7)
    const int rb = wave >> 1, wj = wave & 1;
    const int pr = 4 * rb + (frow >> 3), pc = frow & 7;
    const int px2 = 32 * rb + frow;           // the pixel's row in the t2 tile
    const int er = lane >> 2, ec = (lane & 3) * 8;
    unsigned char* const ep = wk + BN_T2_BYTES + wave * BN_EP_BYTES;

    // ---- tiles: block b runs on XCD b % 8 and takes that XCD's contiguous eighth of the tiles, 32 (= blocks per XCD) apart
    const int per = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7;
    const int tpx = (a.ntiles + 7) >> 3;
    const int t_end = min((xcd + 1) * tpx, a.ntiles);
    int tile = xcd * tpx + ((int)blockIdx.x >> 3);

    bn_u32x4 xa[NST];                          // conv1's pixel fragments (halo slot), k = 16 st + 8 fhalf .. + 7
    [[maybe_unused]] bn_u32x4 xc[4];           // DS: the downsample's (own pixel)
    auto tile_origin = [&](int t, int& b, int& th0, int& tw0) {
        b = t / a.tiles_pf;
        const int rem = t - b * a.tiles_pf;
        const int ty = rem / a.tiles_x;
        th0 = ty * 8; tw0 = (rem - ty * a.tiles_x) * 8;
    };
    auto request_xa = [&](int t) {
        int b, th0, tw0;
        tile_origin(t < t_end ? t : 0, b, th0, tw0);
        const int h = th0 + a_dh, w = tw0 + a_dw;
        const bool ok = !(BN_EXP & 8) && t < t_end && slot_ok && h >= 0 && h < a.H && w >= 0 && w < a.W;
        const bn_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)b * a.H * a.W * CX), 0, 0x7FFFFF00u, 0x00020000);
        const unsigned off = ok ? (unsigned)((h * a.W + w) * CX + fhalf * 8) * 2u : OOB;
#pragma unroll
        for (int st = 0; st < NST; ++st) xa[st] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + st * 32u : OOB, 0, 0);
    };
    auto request_xc = [&](int t) {
        int b, th0, tw0;
        tile_origin(t < t_end ? t : 0, b, th0, tw0);
        const bn_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)b * a.H * a.W * 64), 0, 0x7FFFFF00u, 0x00020000);
        const unsigned off = (unsigned)(((th0 + pr) * a.W + tw0 + pc) * 64 + fhalf * 8) * 2u;
#pragma unroll
        for (int st = 0; st < 4; ++st) xc[st] = __builtin_amdgcn_raw_buffer_load_b128(rs, t < t_end ? off + st * 32u : OOB, 0, 0);
    };

    request_xa(tile);
    if constexpr (DS) request_xc(tile);
    for (; tile < t_end; tile += per) {
        int b, th0, tw0;
        tile_origin(tile, b, th0, tw0);
        const size_t frame_px = (size_t)b * a.H * a.W;
        // ================= phase A: conv1 on this wave's 32 halo pixels
        bn_f32x16 acc1[2];                    // (every accumulator of this kernel starts at its bias: register 4 g + e = channel 8 g + 4 fhalf + e of the block)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bn_f32x4 bv = *reinterpret_cast<const bn_f32x4*>(B1s + 32 * j + 8 * g + 4 * fhalf);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[j][4 * g + e] = bv[e];
            }
        if constexpr (DS) {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bn_bf16x8, w1f[j][st]), __builtin_bit_cast(bn_bf16x8, xa[st]), acc1[j], 0, 0, 0);
        } else {                              // W1 fragments from LDS, the ring of the other loops: step s = (k-step s / 2, channel block s % 2)
            constexpr int DEPTH = 3, NSTEP = 2 * NST;
            bn_u32x4 wfr[DEPTH + 1];
            auto fetch = [&](int s) {
                const int st = s >> 1, j = s & 1;
                wfr[s & DEPTH] = *reinterpret_cast<const bn_u32x4*>(Wds + ((st >> 2) * 64 + 32 * j + frow) * 128 + (((2 * (st & 3) + fhalf) ^ fsw) * 16));
            };
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) fetch(s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (s + DEPTH < NSTEP) fetch(s + DEPTH);
                acc1[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bn_bf16x8, wfr[s & DEPTH]), __builtin_bit_cast(bn_bf16x8, xa[s >> 1]), acc1[s & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int ha = th0 + a_dh, wa = tw0 + a_dw;
        const bool in_img = slot_ok && ha >= 0 && ha < a.H && wa >= 0 && wa < a.W;
        request_xa(tile + per);               // (xa is dead: the next tile's halo pixels fly under phases B and C)
        __builtin_amdgcn_s_barrier();         // every wave is done with the previous tile's t2 / scratch: the work area is t1 again
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = in_img ? fmaxf(acc1[j][4 * g + e], 0.f) : 0.f;
                const bn_u32x2 pk = bn_u32x2{pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2], t[3])};
                if (slot_ok) *reinterpret_cast<bn_u32x2*>(wk + t1_wr + ((((4 * j + g) ^ t1_sw) & 7) * 16)) = pk;
                if (TAP && a_interior) *reinterpret_cast<bn_u32x2*>(a.t1 + (frame_px + (size_t)ha * a.W + wa) * 64 + 32 * j + 8 * g + 4 * fhalf) = pk;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ================= phase B: conv2, wave = (pixel block rb, channels 32 wj .. + 31): 4 chunks of 16 channels x 9 taps
        bn_f32x16 acc2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bn_f32x4 bv = *reinterpret_cast<const bn_f32x4*>(B2s + 32 * wj + 8 * g + 4 * fhalf);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc2[4 * g + e] = bv[e];
        }
        if (!(BN_EXP & 1)) {
            // one wave per SIMD: nothing hides an LDS round trip but the wave's own instruction stream -- the fragments of step s + 3 are
            // requested before the MFMA of step s (a ring of four register slots; the compiler alone emits read, read, wait, multiply)
            const unsigned char* w2r = W2s + (32 * wj + frow) * 128;
            constexpr int DEPTH = 3, NSTEP = 36;
            bn_u32x4 afr[DEPTH + 1], bfr[DEPTH + 1];
            auto fetch = [&](int s) {             // step s = chunk s / 9 (16 channels), tap s % 9
                const int cc = s / 9, tap = s - cc * 9;
                const int hr = pr + tap / 3, hc = pc + tap % 3, p = hr * 10 + hc;
                afr[s & DEPTH] = *reinterpret_cast<const bn_u32x4*>(wk + p * 128 + ((((2 * cc + fhalf) ^ ((hc >> 1) + 4 * hr)) & 7) * 16));
                bfr[s & DEPTH] = *reinterpret_cast<const bn_u32x4*>(w2r + tap * (64 * 128) + (((2 * cc + fhalf) ^ fsw) * 16));
            };
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) fetch(s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (s + DEPTH < NSTEP) fetch(s + DEPTH);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bn_bf16x8, bfr[s & DEPTH]), __builtin_bit_cast(bn_bf16x8, afr[s & DEPTH]), acc2, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        bn_u32x2 t2p[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            t2p[g] = bn_u32x2{pack_bf16x2(fmaxf(acc2[4 * g], 0.f), fmaxf(acc2[4 * g + 1], 0.f)),
                              pack_bf16x2(fmaxf(acc2[4 * g + 2], 0.f), fmaxf(acc2[4 * g + 3], 0.f))};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();         // every wave is done reading t1: t2 goes over it
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<bn_u32x2*>(wk + px2 * 128 + (((4 * wj + g) ^ ((px2 >> 1) & 7)) * 16) + fhalf * 8) = t2p[g];
            if (TAP) *reinterpret_cast<bn_u32x2*>(a.t2 + (frame_px + (size_t)(th0 + pr) * a.W + tw0 + pc) * 64 + 32 * wj + 8 * g + 4 * fhalf) = t2p[g];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ================= phase C: conv3 (from t2) and the downsample (from x), wave = (pixel block rb, channels 128 wj .. + 127)
        bn_u32x4 a2[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) a2[st] = *reinterpret_cast<const bn_u32x4*>(wk + px2 * 128 + (((2 * st + fhalf) ^ ((px2 >> 1) & 7)) * 16));
        const bn_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + frame_px * 256), 0, 0x7FFFFF00u, 0x00020000);
        [[maybe_unused]] bn_u32x4 resq[4][2];  // identity block: the residual rows of this wave's four channel blocks, coalesced layout (rows 16 h + er, channels n0 + ec .. + 7)
        if constexpr (!DS) {
            const bn_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + frame_px * 256), 0, 0x7FFFFF00u, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 16 * h + er;
                    resq[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (unsigned)(((th0 + 4 * rb + (row >> 3)) * a.W + tw0 + (row & 7)) * 256 + 32 * (4 * wj + i) + ec) * 2u, 0, 0);
                }
        }
        bn_f32x16 z[4];
        [[maybe_unused]] bn_f32x16 d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bn_f32x4 b3v = *reinterpret_cast<const bn_f32x4*>(B3s + 32 * (4 * wj + i) + 8 * g + 4 * fhalf);
#pragma unroll
                for (int e = 0; e < 4; ++e) z[i][4 * g + e] = b3v[e];
                if constexpr (DS) {
                    const bn_f32x4 bdv = *reinterpret_cast<const bn_f32x4*>(BDs + 32 * (4 * wj + i) + 8 * g + 4 * fhalf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[i][4 * g + e] = bdv[e];
                }
            }
        if (!(BN_EXP & 2)) {
            constexpr int DEPTH = 3, NSTEP = 16;   // step s = (k-step s / 4, channel block s % 4): the same ring as conv2's
            bn_u32x4 f3r[DEPTH + 1];
            [[maybe_unused]] bn_u32x4 fdr[DEPTH + 1];
            auto fetch = [&](int s) {
                const int st = s >> 2, i = s & 3;
                const unsigned wo = (unsigned)((32 * (4 * wj + i) + frow) * 128 + (((2 * st + fhalf) ^ fsw) * 16));
                f3r[s & DEPTH] = *reinterpret_cast<const bn_u32x4*>(W3s + wo);
                if constexpr (DS) fdr[s & DEPTH] = *reinterpret_cast<const bn_u32x4*>(Wds + wo);
            };
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) fetch(s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int st = s >> 2, i = s & 3;
                if (s + DEPTH < NSTEP) fetch(s + DEPTH);
                z[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bn_bf16x8, f3r[s & DEPTH]), __builtin_bit_cast(bn_bf16x8, a2[st]), z[i], 0, 0, 0);
                if constexpr (DS) d[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bn_bf16x8, fdr[s & DEPTH]), __builtin_bit_cast(bn_bf16x8, xc[st]), d[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (DS) request_xc(tile + per);
        // ---- y = relu(acc3 + shortcut), shortcut = bf16(acc_d) or the residual rows; one 32 x 32 block at a time through this wave's scratch
#pragma unroll
        for (int i = 0; i < ((BN_EXP & 4) ? 1 : 4); ++i) {
            const int n0 = 32 * (4 * wj + i);
            __builtin_amdgcn_wave_barrier();
            if constexpr (!DS) {                  // the residual rows cross the scratch: coalesced layout in, accumulator layout out
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 16 * h + er;
                    *reinterpret_cast<bn_u32x4*>(ep + row * BN_EP_PITCH + (((lane & 3) ^ ((row >> 2) & 3)) * 16)) = resq[i][h];
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            bn_u32x2 rres[4];
            if constexpr (!DS) {
#pragma unroll
                for (int g = 0; g < 4; ++g) rres[g] = *reinterpret_cast<const bn_u32x2*>(ep + frow * BN_EP_PITCH + ((g ^ ((frow >> 2) & 3)) * 16) + fhalf * 8);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();  // (every lane has its residual: the scratch takes y)
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned r01, r23;
                if constexpr (DS) {
                    r01 = pack_bf16x2(d[i][4 * g], d[i][4 * g + 1]);
                    r23 = pack_bf16x2(d[i][4 * g + 2], d[i][4 * g + 3]);
                } else {
                    r01 = rres[g][0]; r23 = rres[g][1];
                }
                typedef float f2_t __attribute__((ext_vector_type(2)));
                const f2_t s01 = f2_t{z[i][4 * g], z[i][4 * g + 1]} + f2_t{__uint_as_float(r01 << 16), __uint_as_float(r01 & 0xFFFF0000u)};
                const f2_t s23 = f2_t{z[i][4 * g + 2], z[i][4 * g + 3]} + f2_t{__uint_as_float(r23 << 16), __uint_as_float(r23 & 0xFFFF0000u)};
                const float y0 = fmaxf(s01[0], 0.f), y1 = fmaxf(s01[1], 0.f), y2 = fmaxf(s23[0], 0.f), y3 = fmaxf(s23[1], 0.f);
                if (BN_EXP & 32) __builtin_amdgcn_raw_buffer_store_b64(bn_u32x2{pack_bf16x2(y0, y1), pack_bf16x2(y2, y3)}, rs_y, (unsigned)(((th0 + pr) * a.W + tw0 + pc) * 256 + n0 + 8 * g + 4 * fhalf) * 2u, 0, 0);
                else *reinterpret_cast<bn_u32x2*>(ep + frow * BN_EP_PITCH + ((g ^ ((frow >> 2) & 3)) * 16) + fhalf * 8) = bn_u32x2{pack_bf16x2(y0, y1), pack_bf16x2(y2, y3)};
                if (TAP && DS) *reinterpret_cast<bn_u32x2*>(a.r + (frame_px + (size_t)(th0 + pr) * a.W + tw0 + pc) * 256 + n0 + 8 * g + 4 * fhalf) = bn_u32x2{r01, r23};
            }
            if (BN_EXP & 32) continue;
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 16 * h + er;                      // pixel (4 rb + (row >> 3), row & 7) of the tile
                const bn_u32x4 o = *reinterpret_cast<const bn_u32x4*>(ep + row * BN_EP_PITCH + (((lane & 3) ^ ((row >> 2) & 3)) * 16));
                const unsigned off = (unsigned)(((th0 + 4 * rb + (row >> 3)) * a.W + tw0 + (row & 7)) * 256 + n0 + ec) * 2u;
                if (!(BN_EXP & 16) || (o[0] == 0x12345678u)) __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, off, 0, 0);
            }
        }
    }
#endif
}

// the four convs of a first bottleneck: c1 (1x1 64 -> 64, ReLU), c2 (3x3 pad 1 on c1's output, ReLU), ds (1x1 64 -> 256 on c1's input, no
// activation), c3 (1x1 64 -> 256 on c2's output + ds's output, ReLU); plain bf16 NHWC tensors with 8 x 8 tiles that fit the image exactly
bool bneck0_bf16_ok(const GemmArgs& c1, const GemmArgs& c2, const GemmArgs& ds, const GemmArgs& c3) {
    auto base = [](const GemmArgs& g, int ks, int pad, int cin, int n, int act) {
        return g.conv && g.ks == ks && g.stride == 1 && g.pad == pad && g.Cin == cin && g.N == n && g.K == ks * ks * cin && g.Kpad == g.K && g.act == act &&
               g.omap.G == 1 && g.omap.S1 == n && g.omap.off == 0 && !g.rscale && !g.ln_g && !g.up && g.splits <= 1 && g.Ho == g.H && g.Wo == g.W;
    };
    if (!base(c1, 1, 0, 64, 64, ACT_RELU) || !base(c2, 3, 1, 64, 64, ACT_RELU) || !base(ds, 1, 0, 64, 256, ACT_NONE) || !base(c3, 1, 0, 64, 256, ACT_RELU)) return false;
    // (who reads whom is the caller's statement -- Engine::bneck0_head compares buffer ids; the launcher compares the pointers)
    if (c1.res || c2.res || ds.res || c3.rmap.G != 1 || c3.rmap.S1 != 256 || c3.rmap.off != 0) return false;
    if (c1.H != c2.H || c1.W != c2.W || c1.H != c3.H || c1.W != c3.W || c1.H != ds.H || c1.W != ds.W || c1.M != c2.M || c1.M != c3.M || c1.M != ds.M) return false;
    if (c1.H % 8 != 0 || c1.W % 8 != 0 || c1.M % (c1.H * c1.W) != 0) return false;
    if ((double)c1.H * c1.W * 256.0 * 2.0 >= 2.0e9) return false;      // (32-bit offsets span one frame)
    return true;
}

const char* bneck0_bf16_kernel_name() { return "bneck0_bf16<8x8>"; }

hipError_t launch_bneck0_bf16(const GemmArgs& c1, const GemmArgs& c2, const GemmArgs& ds, const GemmArgs& c3, bool tap, hipStream_t s) {
    if (!bneck0_bf16_ok(c1, c2, ds, c3)) return hipErrorInvalidValue;
    for (const GemmArgs* g : {&c1, &c2, &ds, &c3})
        if (!g->A || !g->Wp || !g->bias || !g->out) return hipErrorInvalidValue;
    if (c1.A != ds.A || c2.A != c1.out || c3.A != c2.out || c3.res != ds.out) return hipErrorInvalidValue;
    typedef const unsigned short* hp;
    Bneck0Args a{};
    a.x = reinterpret_cast<hp>(c1.A);
    a.y = reinterpret_cast<unsigned short*>(c3.out);
    a.w1 = reinterpret_cast<hp>(c1.Wp); a.w2 = reinterpret_cast<hp>(c2.Wp); a.w3 = reinterpret_cast<hp>(c3.Wp); a.wd = reinterpret_cast<hp>(ds.Wp);
    a.b1 = c1.bias; a.b2 = c2.bias; a.b3 = c3.bias; a.bd = ds.bias;
    a.t1 = reinterpret_cast<unsigned short*>(c1.out); a.t2 = reinterpret_cast<unsigned short*>(c2.out); a.r = reinterpret_cast<unsigned short*>(ds.out);
    a.H = c1.H; a.W = c1.W; a.B = c1.M / (c1.H * c1.W);
    a.tiles_x = a.W / 8; a.tiles_pf = (a.H / 8) * a.tiles_x; a.ntiles = a.B * a.tiles_pf;
    static DynLdsAttr attr_p, attr_t;
    const void* k = tap ? reinterpret_cast<const void*>(&bneck_bf16_kernel<true, true>) : reinterpret_cast<const void*>(&bneck_bf16_kernel<false, true>);
    const hipError_t e = (tap ? attr_t : attr_p).ensure(k, BN_LDS_BYTES);
    if (e != hipSuccess) return e;
    if (tap) hipLaunchKernelGGL((bneck_bf16_kernel<true, true>), dim3(256), dim3(256), BN_LDS_BYTES, s, a);
    else hipLaunchKernelGGL((bneck_bf16_kernel<false, true>), dim3(256), dim3(256), BN_LDS_BYTES, s, a);
    return hipGetLastError();
}

// an identity bottleneck: c1 (1x1 256 -> 64, ReLU), c2 (3x3 pad 1 on c1's output, ReLU), c3 (1x1 64 -> 256 on c2's output + c1's INPUT, ReLU)
bool bneck1_bf16_ok(const GemmArgs& c1, const GemmArgs& c2, const GemmArgs& c3) {
    auto base = [](const GemmArgs& g, int ks, int pad, int cin, int n) {
        return g.conv && g.ks == ks && g.stride == 1 && g.pad == pad && g.Cin == cin && g.N == n && g.K == ks * ks * cin && g.Kpad == g.K && g.act == ACT_RELU &&
               g.omap.G == 1 && g.omap.S1 == n && g.omap.off == 0 && !g.rscale && !g.ln_g && !g.up && g.splits <= 1 && g.Ho == g.H && g.Wo == g.W;
    };
    if (!base(c1, 1, 0, 256, 64) || !base(c2, 3, 1, 64, 64) || !base(c3, 1, 0, 64, 256)) return false;
    if (c1.res || c2.res || c3.rmap.G != 1 || c3.rmap.S1 != 256 || c3.rmap.off != 0) return false;
    if (c1.H != c2.H || c1.W != c2.W || c1.H != c3.H || c1.W != c3.W || c1.M != c2.M || c1.M != c3.M) return false;
    if (c1.H % 8 != 0 || c1.W % 8 != 0 || c1.M % (c1.H * c1.W) != 0) return false;
    if ((double)c1.H * c1.W * 256.0 * 2.0 >= 2.0e9) return false;
    return true;
}

const char* bneck1_bf16_kernel_name() { return "bneck1_bf16<8x8>"; }

hipError_t launch_bneck1_bf16(const GemmArgs& c1, const GemmArgs& c2, const GemmArgs& c3, bool tap, hipStream_t s) {
    if (!bneck1_bf16_ok(c1, c2, c3)) return hipErrorInvalidValue;
    for (const GemmArgs* g : {&c1, &c2, &c3})
        if (!g->A || !g->Wp || !g->bias || !g->out) return hipErrorInvalidValue;
    if (c2.A != c1.out || c3.A != c2.out || c3.res != c1.A || c3.out == c1.A) return hipErrorInvalidValue;
    typedef const unsigned short* hp;
    Bneck0Args a{};
    a.x = reinterpret_cast<hp>(c1.A);
    a.y = reinterpret_cast<unsigned short*>(c3.out);
    a.w1 = reinterpret_cast<hp>(c1.Wp); a.w2 = reinterpret_cast<hp>(c2.Wp); a.w3 = reinterpret_cast<hp>(c3.Wp);
    a.b1 = c1.bias; a.b2 = c2.bias; a.b3 = c3.bias;
    a.t1 = reinterpret_cast<unsigned short*>(c1.out); a.t2 = reinterpret_cast<unsigned short*>(c2.out);
    a.H = c1.H; a.W = c1.W; a.B = c1.M / (c1.H * c1.W);
    a.tiles_x = a.W / 8; a.tiles_pf = (a.H / 8) * a.tiles_x; a.ntiles = a.B * a.tiles_pf;
    static DynLdsAttr attr_p, attr_t;
    const void* k = tap ? reinterpret_cast<const void*>(&bneck_bf16_kernel<true, false>) : reinterpret_cast<const void*>(&bneck_bf16_kernel<false, false>);
    const hipError_t e = (tap ? attr_t : attr_p).ensure(k, BN_LDS_BYTES);
    if (e != hipSuccess) return e;
    if (tap) hipLaunchKernelGGL((bneck_bf16_kernel<true, false>), dim3(256), dim3(256), BN_LDS_BYTES, s, a);
    else hipLaunchKernelGGL((bneck_bf16_kernel<false, false>), dim3(256), dim3(256), BN_LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace capf
