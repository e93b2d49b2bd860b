// bf16 twin of igemm_f32_pwchain.hip: the end of a ResNet / HRNet `layer1` bottleneck and the start of the next one
//     y  = relu(W3 . t + b3 + res)      64 -> 256 channels, stored as bf16 (it is the next block's residual)
//     t' = relu(W1 . y + b1)            256 -> 64 channels, stored as bf16
// in ONE persistent kernel in which a wave owns 32 pixels through both convs (CPN: networks/resnet.py:58-93; HRNet:
// pose_hrnet.py:98-136; compute_dtype = bf16).  At bf16 MFMA speed the pair is pure HBM streaming (2048 matrix cycles per
// 32-pixel tile against ~4700 cycles of its 40 KiB at 4.6 TB/s): what the chain removes is the write + read-back of y's second
// trip (the second conv reading it) and two prologue / epilogue-bound launches.
//   * both weight matrices (32 KiB each) stay in LDS: 128-byte rows of 64 k with the quad swizzle of the bf16 tiles;
//   * the first conv's A fragments come straight from global memory (8 consecutive bf16 of the lane's row per k-step);
//   * y goes through bias / residual / ReLU in fp32 registers, is ROUNDED TO bf16 ONCE -- the value that is stored is the value
//     the second conv multiplies -- and becomes the second conv's A operand without leaving the registers: the accumulator
//     register 4 g + e of N-tile j is channel 32 j + 8 g + 4 (lane >> 5) + e, so the eight values g = 0, 1 (and g = 2, 3) of a lane
//     are the k-slots 8 (lane >> 5) .. + 7 of one 16-deep MFMA step IF the step's k order is (0-3, 8-11 | 4-7, 12-15): the
//     second conv's weights are staged with the two middle 4-channel groups of every 16 swapped;
//   * residual rows and stores are coalesced (16 rows x 64 B per instruction) through a per-wave LDS transpose.
// The first conv is bit-identical to igemm_bf16_tile's (same operands per MFMA, same ((acc + bias) + res) epilogue); the second
// sums every 16 products in one MFMA like the plain kernel but in a permuted slot order -- equal up to the matrix core's
// internal summation order (tests: layer-wise bf16 parity of the engine's ops).
#include "kernels.h"

namespace capf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int K1, int N1, int N2>
__global__ __launch_bounds__(256, 1) void igemm_bf16_pwchain_kernel(GemmArgs p3, GemmArgs p1, int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT1 = N1 / 32, NT2 = N2 / 32, ST1 = K1 / 16;    // N-tiles of the two convs, 16-deep k-steps of the first
    constexpr int EPS = 36;                                        // padded fp32 row of a wave's 32 x 32 transpose scratch
    constexpr unsigned OOB = 0x80000000u;
    static_assert(K1 == 64 && N1 % 64 == 0 && N2 % 32 == 0, "one 64-deep sub-chunk for the first conv, whole ones for the second");
    extern __shared__ __attribute__((aligned(16))) unsigned short ldsb[];     // W3 [N1][64] | W1 N1/64 x [N2][64] | fp32: b3 | b1 | scratch
    unsigned short* W3s = ldsb;
    unsigned short* W1s = ldsb + N1 * K1;
    float* B3s = reinterpret_cast<float*>(W1s + N2 * N1);
    float* B1s = B3s + N1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* ep = B1s + N2 + wave * (32 * EPS);
    const unsigned short* W3g = reinterpret_cast<const unsigned short*>(p3.Wp);
    const unsigned short* W1g = reinterpret_cast<const unsigned short*>(p1.Wp);

    // ---- weights -> LDS, once: sub-chunk s = k / 64: [N][64] halves = 128-byte rows of 8 quads, quad q stored at q ^ ((n >> 1) & 7)
    for (int i = tid; i < N1 * (K1 / 8); i += 256) {
        const int n = i / (K1 / 8), q = i - n * (K1 / 8);
        *reinterpret_cast<u32x4*>(W3s + n * 64 + ((q ^ ((n >> 1) & 7)) * 8)) = *reinterpret_cast<const u32x4*>(W3g + (size_t)n * p3.Kpad + q * 8);
    }
    // second conv: pairs of quads (one 16-deep MFMA step); channels (0-3, 4-7 | 8-11, 12-15) -> slots (0-3, 8-11 | 4-7, 12-15)
    for (int i = tid; i < N2 * (N1 / 16); i += 256) {
        const int n = i / (N1 / 16), pr = i - n * (N1 / 16);
        const u32x4 lo = *reinterpret_cast<const u32x4*>(W1g + (size_t)n * p1.Kpad + pr * 16);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(W1g + (size_t)n * p1.Kpad + pr * 16 + 8);
        const int sub = pr >> 2, q0 = (pr & 3) * 2;
        unsigned short* row = W1s + sub * (N2 * 64) + n * 64;
        *reinterpret_cast<u32x4*>(row + ((q0 ^ ((n >> 1) & 7)) * 8)) = u32x4{lo[0], lo[1], hi[0], hi[1]};
        *reinterpret_cast<u32x4*>(row + (((q0 + 1) ^ ((n >> 1) & 7)) * 8)) = u32x4{lo[2], lo[3], hi[2], hi[3]};
    }
    for (int i = tid; i < N1; i += 256) B3s[i] = p3.bias[i];
    for (int i = tid; i < N2; i += 256) B1s[i] = p1.bias[i];
    __syncthreads();

    const int frow = lane & 31, fhalf = lane >> 5, fsw = (frow >> 1) & 7;
    const int er = lane >> 2, ec = (lane & 3) * 8;                 // coalesced layout: 4 lanes x 16 B = the 64 B of one row of an N-tile
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const long s_a = p3.K, s_r = p3.rmap.S1, s_y = p3.omap.S1, s_t = p1.omap.S1;     // (elements = halves)
    const unsigned short* Ag = reinterpret_cast<const unsigned short*>(p3.A);
    const unsigned short* Rg = reinterpret_cast<const unsigned short*>(p3.res);
    unsigned short* Yg = reinterpret_cast<unsigned short*>(p3.out);
    unsigned short* Tg = reinterpret_cast<unsigned short*>(p1.out);

    u32x4 a1[ST1];                       // A fragments of the first conv: k = 16 st + 8 fhalf .. + 7 of row frow
    u32x4 res[NT1][2];                   // residual, coalesced layout: rows 16 h + er, channels 32 j + ec .. + 7
    auto request_a = [&](int tile) {
        const long m0 = (long)tile * 32;
        const bool ok = tile < ntiles && m0 + frow < p3.M;
        const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Ag + (tile < ntiles ? m0 : 0) * s_a), 0, 0x7FFFFF00u, 0x00020000);
#pragma unroll
        for (int st = 0; st < ST1; ++st)
            a1[st] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(frow * (int)s_a + st * 16 + fhalf * 8) * 2u : OOB, 0, 0);
    };
    auto request_res = [&](int tile, int j) {
        const long m0 = (long)tile * 32;
        const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Rg + (tile < ntiles ? m0 : 0) * s_r + p3.rmap.off), 0, 0x7FFFFF00u, 0x00020000);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = tile < ntiles && m0 + 16 * h + er < p3.M;
            res[j][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)((16 * h + er) * (int)s_r + 32 * j + ec) * 2u : OOB, 0, 0);
        }
    };

    int tile = gw;
    for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(30);   // de-phase the four waves of a CU (see igemm_f32_pwchain.hip)
    request_a(tile);
#pragma unroll
    for (int j = 0; j < NT1; ++j) request_res(tile, j);
    for (; tile < ntiles; tile += nw) {
        const long m0 = (long)tile * 32;
        // ---- first conv
        f32x16 y[NT1];
#pragma unroll
        for (int j = 0; j < NT1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[j][r] = 0.f;
        auto w3_frag = [&](int st, int j) {
            return *reinterpret_cast<const u32x4*>(&W3s[(32 * j + frow) * 64 + (((st * 2 + fhalf) ^ fsw) * 8)]);
        };
        {
            u32x4 bf = w3_frag(0, 0);
#pragma unroll
            for (int st = 0; st < ST1; ++st)
#pragma unroll
                for (int j = 0; j < NT1; ++j) {
                    const int jn = j + 1 < NT1 ? j + 1 : 0, sn = j + 1 < NT1 ? st : (st + 1 < ST1 ? st + 1 : st);
                    const u32x4 nx = w3_frag(sn, jn);
                    y[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf), __builtin_bit_cast(bf16x8, a1[st]), y[j], 0, 0, 0);
                    bf = nx;
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        request_a(tile + nw);
        // ---- y = relu((acc + b3) + res): stored as bf16, and the same bf16 values packed as the second conv's A operand
        const rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(Yg + m0 * s_y + p3.omap.off), 0, 0x7FFFFF00u, 0x00020000);
        u32x4 yb[NT1][2];                // [j][s]: k-step 2 j + s of the second conv (g = 2 s, 2 s + 1)
#pragma unroll
        for (int j = 0; j < NT1; ++j) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float* dst = &ep[(16 * h + er) * EPS + ec];
                const u32x4 q = res[j][h];
                *reinterpret_cast<f32x4*>(dst) = f32x4{__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xFFFF0000u),
                                                       __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xFFFF0000u)};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{__uint_as_float(q[2] << 16), __uint_as_float(q[2] & 0xFFFF0000u),
                                                           __uint_as_float(q[3] << 16), __uint_as_float(q[3] & 0xFFFF0000u)};
            }
            __builtin_amdgcn_wave_barrier();
            unsigned pk[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float* cell = &ep[frow * EPS + 8 * g + 4 * fhalf];
                const f32x4 rv = *reinterpret_cast<const f32x4*>(cell);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&B3s[32 * j + 8 * g + 4 * fhalf]);
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = fmaxf((y[j][4 * g + e] + bv[e]) + rv[e], 0.f);
                pk[2 * g] = pack_bf16x2(t[0], t[1]);
                pk[2 * g + 1] = pack_bf16x2(t[2], t[3]);
                // (the rounded values go back to the scratch as fp32: exact, and the coalesced pass below packs them again)
                *reinterpret_cast<f32x4*>(cell) = f32x4{__uint_as_float(pk[2 * g] << 16), __uint_as_float(pk[2 * g] & 0xFFFF0000u),
                                                        __uint_as_float(pk[2 * g + 1] << 16), __uint_as_float(pk[2 * g + 1] & 0xFFFF0000u)};
            }
            yb[j][0] = u32x4{pk[0], pk[1], pk[2], pk[3]};
            yb[j][1] = u32x4{pk[4], pk[5], pk[6], pk[7]};
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* src = &ep[(16 * h + er) * EPS + ec];
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
                const u32x4 o = u32x4{pack_bf16x2(x0[0], x0[1]), pack_bf16x2(x0[2], x0[3]), pack_bf16x2(x1[0], x1[1]), pack_bf16x2(x1[2], x1[3])};
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, m0 + 16 * h + er < p3.M ? (unsigned)((16 * h + er) * (int)s_y + 32 * j + ec) * 2u : OOB, 0, 0);
            }
        }
        // ---- second conv: K = N1 from the registers above
        f32x16 z[NT2];
#pragma unroll
        for (int jn = 0; jn < NT2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) z[jn][r] = 0.f;
        auto w1_frag = [&](int j, int s, int jn) {   // k-step 2 j + s: sub-chunk (2 j + s) / 4, quads 2 ((2 j + s) % 4) + fhalf
            const int ks = 2 * j + s;
            return *reinterpret_cast<const u32x4*>(&W1s[(ks >> 2) * (N2 * 64) + (32 * jn + frow) * 64 + ((((ks & 3) * 2 + fhalf) ^ fsw) * 8)]);
        };
        {
            u32x4 bf = w1_frag(0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT1; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int jn = 0; jn < NT2; ++jn) {
                        const bool last_n = jn + 1 == NT2, last_s = s == 1;
                        const int n2 = last_n ? 0 : jn + 1, s2 = last_n ? (last_s ? 0 : 1) : s, j2 = last_n && last_s ? (j + 1 < NT1 ? j + 1 : j) : j;
                        const u32x4 nx = w1_frag(j2, s2, n2);
                        z[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf), __builtin_bit_cast(bf16x8, yb[j][s]), z[jn], 0, 0, 0);
                        bf = nx;
                        if (last_s && last_n) request_res(tile + nw, j);     // the next tile's residual, a trickle under this loop
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
        const rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void*)(Tg + m0 * s_t + p1.omap.off), 0, 0x7FFFFF00u, 0x00020000);
#pragma unroll
        for (int jn = 0; jn < NT2; ++jn) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&B1s[32 * jn + 8 * g + 4 * fhalf]);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(z[jn][4 * g + e] + bv[e], 0.f);
                *reinterpret_cast<f32x4*>(&ep[frow * EPS + 8 * g + 4 * fhalf]) = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* src = &ep[(16 * h + er) * EPS + ec];
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
                const u32x4 o = u32x4{pack_bf16x2(x0[0], x0[1]), pack_bf16x2(x0[2], x0[3]), pack_bf16x2(x1[0], x1[1]), pack_bf16x2(x1[2], x1[3])};
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_t, m0 + 16 * h + er < p3.M ? (unsigned)((16 * h + er) * (int)s_t + 32 * jn + ec) * 2u : OOB, 0, 0);
            }
        }
    }
#endif
}

// conv `a` (64 -> 256 + residual + ReLU) directly followed by conv `b` (256 -> 64 + ReLU) on a's output, plain bf16 NHWC pointwise convs
bool gemm_bf16_pwchain_ok(const GemmArgs& a, const GemmArgs& b) {
    static const int on = [] { const char* e = diag_env("CAPF_PWCHAIN"); return e ? atoi(e) : 1; }();        // A/B runs only
    auto plain = [](const GemmArgs& g) {
        return g.conv && g.ks == 1 && g.stride == 1 && g.pad == 0 && g.K == g.Cin && g.omap.G == 1 && (g.omap.S1 & 7) == 0 && (g.omap.off & 7) == 0 &&
               !g.rscale && !g.ln_g && g.splits <= 1 && g.act == ACT_RELU && g.bias;
    };
    if (!on || !plain(a) || !plain(b)) return false;
    if (a.K != 64 || a.Kpad != 64 || a.N != 256 || b.K != 256 || b.Kpad != 256 || b.N != 64 || b.res || a.M != b.M || !a.res) return false;
    if (reinterpret_cast<const unsigned short*>(b.A) != reinterpret_cast<const unsigned short*>(a.out) + a.omap.off || a.omap.S1 != b.K) return false;
    if (a.rmap.G != 1 || (a.rmap.S1 & 7) || (a.rmap.off & 7)) return false;
    {   // b.out must not alias a.res / a.A: see gemm_f32_pwchain_ok
        typedef const unsigned short* hp;
        auto overlaps = [](hp p0, long n0, hp p1, long n1) { return p0 < p1 + n1 && p1 < p0 + n0; };
        const hp bo = reinterpret_cast<hp>(b.out) + b.omap.off;
        const long bn = (long)(b.M - 1) * b.omap.S1 + b.N;
        if (overlaps(bo, bn, reinterpret_cast<hp>(a.res) + a.rmap.off, (long)(a.M - 1) * a.rmap.S1 + a.N)) return false;
        if (overlaps(bo, bn, reinterpret_cast<hp>(a.A), (long)a.M * a.K)) return false;
    }
    return a.M >= 32 * 1024 * 4;
}

const char* gemm_bf16_pwchain_kernel_name() { return "igemm_bf16_pwchain<64,256,64>"; }

hipError_t launch_gemm_bf16_pwchain(const GemmArgs& a, const GemmArgs& b, hipStream_t s) {
    if (!gemm_bf16_pwchain_ok(a, b)) return hipErrorInvalidValue;
    const int ntiles = (a.M + 31) / 32;
    constexpr size_t lds_bytes = (size_t)(256 * 64 + 64 * 256) * 2 + (size_t)(256 + 64 + 4 * 32 * 36) * 4;
    static DynLdsAttr attr_once;
    const hipError_t attr = attr_once.ensure(reinterpret_cast<const void*>(&igemm_bf16_pwchain_kernel<64, 256, 64>), (int)lds_bytes);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((igemm_bf16_pwchain_kernel<64, 256, 64>), dim3(256), dim3(256), lds_bytes, s, a, b, ntiles);
    return hipGetLastError();
}

}  // namespace capf
