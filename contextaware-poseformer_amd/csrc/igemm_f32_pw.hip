// fp32 pointwise (1x1, stride 1) conv + folded BN (+ residual) (+ ReLU) for the HBM-bound bottleneck convs of `layer1`
// (pose_hrnet.py:98-136: 256 -> 64 and 64 -> 256 + residual at 64 x 64): out[m, n] = act(A[m, :K] . W[n, :K] + bias[n] + res[m, n])
// with A the NHWC activations ([M][K], K = Cin), W the ordinary fp32 pack ([N][K]).
//
// The general implicit-GEMM tile (igemm_f32.hip) spends most of a block's life outside its K loop on these shapes (K = 64:
// two chunks; tools/timeline.py: prologue 4.5 us, loop 4.2 us, epilogue 6.6 us) and reaches 2.1-3.4 TB/s.  This kernel is
// the fp32 twin of what fixed the same problem on the bf16 path (igemm_bf16.hip):
//   * ping-pong schedule: one LDS stage of two 32-deep sub-chunks (48 KiB), load phase / compute phase, three blocks per CU;
//   * no taps, no masks, no divisions: a row of A is K contiguous floats;
//   * residual rows and bias requested with the LAST load phase, in the layout of
//   * a coalesced epilogue: every wave transposes its 32x32 accumulator blocks through 4.5 KiB of the idle stage and finishes
//     8 channels of a row per lane (two 16-byte stores, 128 B contiguous per row and four lanes).
// Same arithmetic as the general kernel's epilogue ((acc + bias) + res, ReLU); the K sum runs in the same order (k ascending in
// steps of 8 with the MFMA's (k, k+4) pairing), so results are bit-identical to igemm_f32's for these shapes.
#include "kernels.h"

namespace capf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

[[maybe_unused]] static constexpr int PBK = 32;                 // floats per sub-chunk row (128 B)

template <int NSUB>                            // sub-chunks per load phase (K % (32 NSUB) == 0)
__global__ __launch_bounds__(256, 3) void igemm_f32_pw_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, BN = 64, WM = 64, TM = 2;
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int SUB = (BM + BN) * PBK;        // floats per sub-chunk stage
    __shared__ __attribute__((aligned(16))) float lds[NSUB * SUB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = (p.N + BN - 1) / BN;
    // XCD-aware tile order (see igemm_f32.hip :: xcd_remap)
    const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, x8 = blockIdx.x & 7;
    const int bid = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (blockIdx.x >> 3);
    const int tile_m = bid / nbn, tile_n = bid - tile_m * nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int K = p.K;

    const int srow = tid >> 3;
    const int kq = ((tid & 7) ^ ((srow >> 1) & 7)) * 4;            // logical k offset (floats) of this lane's quad
    constexpr unsigned OOB = 0x80000000u;
    const rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)m0 * K), 0, 0x7FFFFF00u, 0x00020000);
    const rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wp + (long)n0 * p.Kpad), 0,
                                                            (unsigned)(p.N - n0) * (unsigned)p.Kpad * 4u, 0x00020000);
    unsigned a_off[RA], w_off[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) a_off[i] = (m0 + srow + 32 * i < p.M) ? (unsigned)((srow + 32 * i) * K + kq) * 4u : OOB;
#pragma unroll
    for (int i = 0; i < RB; ++i) w_off[i] = (unsigned)((srow + 32 * i) * p.Kpad + kq) * 4u;
    auto fire = [&](int k0) {                                       // one load phase: NSUB sub-chunks starting at column k0
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) {
            const unsigned soff = (unsigned)(k0 + sb * PBK) * 4u;
            float* st = lds + sb * SUB;
#pragma unroll
            for (int i = 0; i < RA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(st + (i * 32 + wave * 8) * PBK), 16, a_off[i], soff, 0, 0);
#pragma unroll
            for (int i = 0; i < RB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(st + BM * PBK + (i * 32 + wave * 8) * PBK), 16, w_off[i], soff, 0, 0);
        }
    };

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * 32;
    const int frow = lane & 31, fhalf = lane >> 5, fsw = (frow >> 1) & 7;

    // epilogue operands in the coalesced layout (lane = 8 channels of one row)
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    const int er = lane >> 2, ec = (lane & 3) * 8;
    f32x4 rr[TM][2][2], bb[2];
    auto prefetch_epilogue = [&]() {
        const int n = n0 + wn0 + ec;
        bb[0] = bb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && (full || n < p.N)) {
            bb[0] = *reinterpret_cast<const f32x4*>(p.bias + n);
            bb[1] = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = m0 + wm0 + i * 32 + h * 16 + er;
                rr[i][h][0] = rr[i][h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.res && (full || (m < p.M && n < p.N))) {
                    const float* src = p.res + (long)m * p.rmap.S1 + p.rmap.off + n;
                    rr[i][h][0] = *reinterpret_cast<const f32x4*>(src);
                    rr[i][h][1] = *reinterpret_cast<const f32x4*>(src + 4);
                }
            }
    };
    auto phase = [&](int k0) {
        fire(k0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) {
            const float* As = lds + sb * SUB;
            const float* Bs = As + BM * PBK;
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                const int qd = ((step * 2 + fhalf) ^ fsw) * 4;
                f32x4 af[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(&As[(wm0 + i * 32 + frow) * PBK + qd]);
                const f32x4 bf = *reinterpret_cast<const f32x4*>(&Bs[(wn0 + frow) * PBK + qd]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[e], af[i][e], acc[i], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    const int nph = K / (PBK * NSUB);
    for (int c = 0; c < nph - 1; ++c) phase(c * PBK * NSUB);
    prefetch_epilogue();                                            // (last phase peeled: these registers are not live in the loop)
    phase((nph - 1) * PBK * NSUB);

    constexpr int EPS = 36;
    float* ep = lds + wave * (32 * EPS);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int n = n0 + wn0 + ec;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(&ep[(lane & 31) * EPS + 8 * g + 4 * (lane >> 5)]) =
                f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = h * 16 + er, m = m0 + wm0 + i * 32 + row;
            f32x4 x[2] = {*reinterpret_cast<const f32x4*>(&ep[row * EPS + ec]), *reinterpret_cast<const f32x4*>(&ep[row * EPS + ec + 4])};
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = (x[q][e] + bb[q][e]) + rr[i][h][q][e];
                    if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
                    x[q][e] = t;
                }
            if (full || (m < p.M && n < p.N)) {
                float* dst = p.out + (long)m * p.omap.S1 + p.omap.off + n;
                *reinterpret_cast<f32x4*>(dst) = x[0];
                *reinterpret_cast<f32x4*>(dst + 4) = x[1];
            }
        }
    }
#endif
}

// 1x1 / stride 1 / no padding, K = Cin = Kpad a multiple of 32, N a multiple of 8, plain 16-byte aligned row maps, no
// DropPath scale / LayerNorm fold / split-K / GELU, and enough tiles for the ping-pong schedule (8 per CU)
bool gemm_f32_pw_ok(const GemmArgs& a) {
    static const int on = [] { const char* e = diag_env("CAPF_F32_PW"); return e ? atoi(e) : 1; }();     // A/B runs only
    if (!on || !a.conv || a.ks != 1 || a.stride != 1 || a.pad != 0 || a.K != a.Cin || a.Kpad != a.K || a.K % 32 != 0 || a.N % 8 != 0)
        return false;
    if (a.omap.G != 1 || (a.omap.S1 & 3) || (a.omap.off & 3) || (a.res && (a.rmap.G != 1 || (a.rmap.S1 & 3) || (a.rmap.off & 3)))) return false;
    if (a.rscale || a.ln_g || a.splits > 1 || a.act == ACT_GELU || a.out_bf16) return false;
    if ((double)a.M * a.K * 4.0 >= 2.0e9) return false;
    return (long)((a.M + 127) / 128) * ((a.N + 63) / 64) >= 2048;
}

const char* gemm_f32_pw_kernel_name() { return "igemm_f32_pw<w4,128x64>"; }

hipError_t launch_gemm_f32_pw(const GemmArgs& a, hipStream_t s) {
    if (!gemm_f32_pw_ok(a)) return hipErrorInvalidValue;
    const dim3 grid(((a.M + 127) / 128) * ((a.N + 63) / 64));
    // (one 32-deep sub-chunk per phase -- 24 KiB, four blocks per CU -- measured 5 % slower on these launches: 0.845 -> 0.89 ms)
    if (a.K % 64 == 0) hipLaunchKernelGGL(igemm_f32_pw_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(igemm_f32_pw_kernel<1>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace capf
