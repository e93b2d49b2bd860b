// Memory-bound helpers of the backbone (gfx950): weight fold/pack, HRNet fuse-sum with nearest
// upsampling, CPN max-pool and bilinear (align_corners=True) resize.  All tensors NHWC fp32; every
// kernel moves 16 bytes per lane with consecutive lanes on consecutive addresses.
#include "kernels.h"

namespace capf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// 4 consecutive channels of an NHWC tensor stored as fp32 (16 B) or bf16 (8 B); arithmetic is always fp32
__device__ __forceinline__ unsigned short f2bf_e(float f) { return to_bf16(f); }
template <bool BF>
__device__ __forceinline__ f32x4 load4(const float* base, long i4) {
    if (!BF) return reinterpret_cast<const f32x4*>(base)[i4];
    const u16x4 h = reinterpret_cast<const u16x4*>(base)[i4];
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __uint_as_float((unsigned)h[e] << 16);
    return v;
}
template <bool BF>
__device__ __forceinline__ void store4(float* base, long i4, f32x4 v) {
    if (!BF) { reinterpret_cast<f32x4*>(base)[i4] = v; return; }
    typedef unsigned u32x2_e __attribute__((ext_vector_type(2)));
    reinterpret_cast<u32x2_e*>(base)[i4] = u32x2_e{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

// ---- BN fold + re-layout of a conv weight (eval-mode BatchNorm, pose_hrnet.py:72-75 etc.) ---------
//   y = (conv(x) - mean) / sqrt(var + eps) * gamma + beta  ==  conv_{w*s}(x) + (beta - mean*s)
template <bool BF>
__global__ void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ mean,
                                 const float* __restrict__ var, float eps, float* __restrict__ Wp,
                                 float* __restrict__ bias, int Cout, int Cin, int ks, int Kpad) {
    const long total = (long)Cout * Kpad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kpad), k = (int)(i - (long)n * Kpad);
        const int K = ks * ks * Cin;
        float v = 0.f;
        const float sc = gamma ? gamma[n] / sqrtf(var[n] + eps) : 1.f;
        if (k < K) {
            const int tap = k / Cin, ci = k - tap * Cin;
            const int kh = tap / ks, kw = tap - kh * ks;
            v = w[(((long)n * Cin + ci) * ks + kh) * ks + kw] * sc;
        }
        if (BF) reinterpret_cast<unsigned short*>(Wp)[i] = f2bf_e(v);
        else Wp[i] = v;
        if (k == 0 && bias) bias[n] = gamma ? beta[n] - mean[n] * sc : 0.f;
    }
}

hipError_t launch_pack_conv(const float* w, const float* gamma, const float* beta, const float* mean,
                            const float* var, float eps, float* Wp, float* bias, int Cout, int Cin,
                            int ks, int Kpad, hipStream_t s) {
    const long total = (long)Cout * Kpad;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_conv_kernel<false>, dim3(blocks), dim3(256), 0, s, w, gamma, beta, mean, var, eps, Wp,
                       bias, Cout, Cin, ks, Kpad);
    return hipGetLastError();
}

// same fold, weights written as bf16 [Cout][Kpad] (Kpad a multiple of 64), bias stays fp32
hipError_t launch_pack_conv_bf16(const float* w, const float* gamma, const float* beta, const float* mean,
                                 const float* var, float eps, void* Wp_bf16, float* bias, int Cout, int Cin, int ks,
                                 int Kpad, hipStream_t s) {
    const long total = (long)Cout * Kpad;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_conv_kernel<true>, dim3(blocks), dim3(256), 0, s, w, gamma, beta, mean, var, eps,
                       reinterpret_cast<float*>(Wp_bf16), bias, Cout, Cin, ks, Kpad);
    return hipGetLastError();
}

__global__ void pack_linear_kernel(const float* __restrict__ w, float* __restrict__ Wp, int N, int K, int Kpad) {
    const long total = (long)N * Kpad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kpad), k = (int)(i - (long)n * Kpad);
        Wp[i] = k < K ? w[(long)n * K + k] : 0.f;
    }
}

hipError_t launch_pack_linear(const float* w, float* Wp, int N, int K, int Kpad, hipStream_t s) {
    const long total = (long)N * Kpad;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_linear_kernel, dim3(blocks), dim3(256), 0, s, w, Wp, N, K, Kpad);
    return hipGetLastError();
}

__global__ void pack_linear_quad_kernel(const float* __restrict__ w, float* __restrict__ Wq, int N, int K, int n0, int Ntot) {
    const long total = (long)N * K;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / K), k = (int)(i - (long)n * K);
        Wq[((long)(k >> 2) * Ntot + n0 + n) * 4 + (k & 3)] = w[i];
    }
}

hipError_t launch_pack_linear_quad(const float* w, float* Wq, int N, int K, int n0, int Ntot, hipStream_t s) {
    if (K % 4 != 0) return hipErrorInvalidValue;
    const long total = (long)N * K;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_linear_quad_kernel, dim3(blocks), dim3(256), 0, s, w, Wq, N, K, n0, Ntot);
    return hipGetLastError();
}

// ---- HRNet fuse: out = relu(sum_i nearest_up(in_i))  (pose_hrnet.py:294-301, nn.Upsample nearest) --
// Input i has resolution (H >> shift_i, W >> shift_i); nearest upsampling by 2^s reads (h>>s, w>>s).
// V channels per lane: 4 (fp32 16 B, bf16 8 B) or 8 (bf16, 16 B); pixel arithmetic in 32 bits (B * H * W < 2^31, launcher checks)
template <bool BF, int V>
__device__ __forceinline__ void fuse_sum_body(const FuseSumArgs& a, const long first, const long stride) {
    constexpr int Q = V / 4;                                  // 4-channel groups per lane
    const int CV = a.C / V, C4 = a.C >> 2;
    const long total = (long)a.B * a.H * a.W * CV;
    for (long i = first; i < total; i += stride) {
        const unsigned pix = (unsigned)(i / CV);
        const int cv = (int)(i - (long)pix * CV);
        const unsigned row = pix / (unsigned)a.W;
        const int w = (int)(pix - row * (unsigned)a.W);
        const int b = (int)(row / (unsigned)a.H), h = (int)(row - (unsigned)b * (unsigned)a.H);
        f32x4 acc[Q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < a.n_in) {
                const int s = a.shift[k];
                const int hs = a.H >> s, ws = a.W >> s;
                const long off = ((((long)b * hs + (h >> s)) * ws + (w >> s)) * C4 + cv * Q);
                if (V == 8) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 q = reinterpret_cast<const u32x4*>(a.in[k])[off >> 1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {               // same order as the reference: ((x0 + x1) + x2) + x3
                        const float lo = __uint_as_float(q[e] << 16), hi = __uint_as_float(q[e] & 0xFFFF0000u);
                        acc[e >> 1][(e & 1) * 2] = k == 0 ? lo : acc[e >> 1][(e & 1) * 2] + lo;
                        acc[e >> 1][(e & 1) * 2 + 1] = k == 0 ? hi : acc[e >> 1][(e & 1) * 2 + 1] + hi;
                    }
                } else {
                    const f32x4 v = load4<BF>(a.in[k], off);
                    acc[0] = k == 0 ? v : acc[0] + v;
                }
            }
        }
        if (a.relu) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                acc[q][0] = fmaxf(acc[q][0], 0.f); acc[q][1] = fmaxf(acc[q][1], 0.f);
                acc[q][2] = fmaxf(acc[q][2], 0.f); acc[q][3] = fmaxf(acc[q][3], 0.f);
            }
        }
        if (V == 8) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            reinterpret_cast<u32x4*>(a.out)[i] = u32x4{pack_bf16x2(acc[0][0], acc[0][1]), pack_bf16x2(acc[0][2], acc[0][3]),
                                                       pack_bf16x2(acc[Q - 1][0], acc[Q - 1][1]), pack_bf16x2(acc[Q - 1][2], acc[Q - 1][3])};
        } else {
            store4<BF>(a.out, i, acc[0]);
        }
    }
}

template <bool BF, int V>
__global__ void fuse_sum_kernel(FuseSumArgs a) {
    fuse_sum_body<BF, V>(a, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

struct FuseGroupArgs {
    FuseSumArgs p[4];
    int bstart[5];     // first block of problem i (bstart[n] = grid size)
    int n;
};
template <bool BF, int V>
__global__ void fuse_sum_group_kernel(FuseGroupArgs g) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < g.n && (int)blockIdx.x >= g.bstart[i]) pi = i;
    const int nb = g.bstart[pi + 1] - g.bstart[pi];
    fuse_sum_body<BF, V>(g.p[pi], (long)(blockIdx.x - g.bstart[pi]) * blockDim.x + threadIdx.x, (long)nb * blockDim.x);
}

hipError_t launch_fuse_sum(const FuseSumArgs& a, hipStream_t s) {
    if ((long)a.B * a.H * a.W >= (1L << 31)) return hipErrorInvalidValue;
    const int V = (a.bf16 && a.C % 8 == 0) ? 8 : 4;
    const long total = (long)a.B * a.H * a.W * (a.C / V);
    const long want = (total + 255) / 256;
    const int blocks = (int)(want < 16384 ? want : 16384);
    if (V == 8) hipLaunchKernelGGL((fuse_sum_kernel<true, 8>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.bf16) hipLaunchKernelGGL((fuse_sum_kernel<true, 4>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((fuse_sum_kernel<false, 4>), dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_fuse_sum_group(const FuseSumArgs* a, int n, hipStream_t s) {
    if (n < 1 || n > 4) return hipErrorInvalidValue;
    if (n == 1) return launch_fuse_sum(a[0], s);
    FuseGroupArgs g{};
    g.n = n;
    const int V = (a[0].bf16 && a[0].C % 8 == 0) ? 8 : 4;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if ((long)a[i].B * a[i].H * a[i].W >= (1L << 31) || a[i].bf16 != a[0].bf16) return hipErrorInvalidValue;
        if (((a[i].bf16 && a[i].C % 8 == 0) ? 8 : 4) != V) return hipErrorInvalidValue;
        const long total = (long)a[i].B * a[i].H * a[i].W * (a[i].C / V);
        const long want = (total + 255) / 256;
        g.p[i] = a[i];
        g.bstart[i] = blocks;
        blocks += (int)(want < 8192 ? want : 8192);
    }
    g.bstart[n] = blocks;
    for (int i = n + 1; i < 5; ++i) g.bstart[i] = blocks;
    if (V == 8) hipLaunchKernelGGL((fuse_sum_group_kernel<true, 8>), dim3(blocks), dim3(256), 0, s, g);
    else if (a[0].bf16) hipLaunchKernelGGL((fuse_sum_group_kernel<true, 4>), dim3(blocks), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((fuse_sum_group_kernel<false, 4>), dim3(blocks), dim3(256), 0, s, g);
    return hipGetLastError();
}

// ---- 3x3 stride-2 pad-1 max pool (networks/resnet.py:104, :140) ----------------------------------
template <bool BF, int V>      // V channels per lane: 4, or 8 for bf16 (16-byte accesses)
__global__ void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                               int C, int Ho, int Wo) {
    constexpr int Q = V / 4;
    const int CV = C / V, C4 = C >> 2;
    const long total = (long)B * Ho * Wo * CV;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const unsigned pix = (unsigned)(i / CV);                   // B * Ho * Wo < 2^31 (launcher checks)
        const int cv = (int)(i - (long)pix * CV);
        const unsigned row = pix / (unsigned)Wo;
        const int wo = (int)(pix - row * (unsigned)Wo);
        const int b = (int)(row / (unsigned)Ho), ho = (int)(row - (unsigned)b * (unsigned)Ho);
        f32x4 m[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) m[q] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * 2 - 1 + kh;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * 2 - 1 + kw;
                if ((unsigned)wi >= (unsigned)W) continue;
                const long off = (((long)b * H + hi) * W + wi) * C4 + cv * Q;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const f32x4 v = load4<BF>(in, off + q);
                    m[q][0] = fmaxf(m[q][0], v[0]); m[q][1] = fmaxf(m[q][1], v[1]);
                    m[q][2] = fmaxf(m[q][2], v[2]); m[q][3] = fmaxf(m[q][3], v[3]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) store4<BF>(out, i * Q + q, m[q]);
    }
}

hipError_t launch_maxpool3x3s2(const float* in, float* out, int B, int H, int W, int C, int Ho, int Wo,
                               hipStream_t s, int bf16) {
    if ((long)B * Ho * Wo >= (1L << 31)) return hipErrorInvalidValue;
    const int V = (bf16 && C % 8 == 0) ? 8 : 4;
    const long total = (long)B * Ho * Wo * (C / V);
    const long want = (total + 255) / 256;
    const dim3 grid((int)(want < 16384 ? want : 16384));
    if (V == 8) hipLaunchKernelGGL((maxpool_kernel<true, 8>), grid, dim3(256), 0, s, in, out, B, H, W, C, Ho, Wo);
    else if (bf16) hipLaunchKernelGGL((maxpool_kernel<true, 4>), grid, dim3(256), 0, s, in, out, B, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL((maxpool_kernel<false, 4>), grid, dim3(256), 0, s, in, out, B, H, W, C, Ho, Wo);
    return hipGetLastError();
}

// ---- bilinear resize, align_corners=True (globalNet.py:40, refineNet.py:61) ----------------------
// ATen upsample_bilinear2d: src = dst * (in-1)/(out-1) (0 if out == 1); i0 = (int)src, i1 = i0 + (i0 < in-1);
// l1 = src - i0, l0 = 1 - l1;  out = l0h*(l0w*v00 + l1w*v01) + l1h*(l0w*v10 + l1w*v11).
template <bool BF, int V>      // V channels per lane: 4 (fp32: 16 B; bf16: 8 B) or 8 (bf16: 16 B)
__global__ void bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                int C, int Ho, int Wo, float sh, float sw, const float* __restrict__ add) {
    const int CV = C / V;
    const long total = (long)B * Ho * Wo * CV;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const unsigned pix = (unsigned)(i / CV);                    // B * Ho * Wo < 2^31 (launcher checks)
        const int cv = (int)(i - (long)pix * CV);
        const unsigned row = pix / (unsigned)Wo;
        const int wo = (int)(pix - row * (unsigned)Wo);
        const int b = (int)(row / (unsigned)Ho), ho = (int)(row - (unsigned)b * (unsigned)Ho);
        const float fh = sh * ho, fw = sw * wo;
        const int h0 = (int)fh, w0 = (int)fw;
        const int h1 = h0 + (h0 < H - 1), w1 = w0 + (w0 < W - 1);
        const float lh1 = fh - h0, lw1 = fw - w0, lh0 = 1.f - lh1, lw0 = 1.f - lw1;
        const long sb = ((long)b * H * W * CV + cv) * (V / 4);      // in units of 4 channels
        const long o00 = sb + ((long)h0 * W + w0) * (C >> 2), o01 = sb + ((long)h0 * W + w1) * (C >> 2);
        const long o10 = sb + ((long)h1 * W + w0) * (C >> 2), o11 = sb + ((long)h1 * W + w1) * (C >> 2);
        if (V == 8) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4* src = reinterpret_cast<const u32x4*>(in);
            const u32x4 q00 = src[o00 >> 1], q01 = src[o01 >> 1], q10 = src[o10 >> 1], q11 = src[o11 >> 1];
            u32x4 qa = {0u, 0u, 0u, 0u};
            if (add) qa = reinterpret_cast<const u32x4*>(add)[i];
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo = lh0 * (lw0 * __uint_as_float(q00[e] << 16) + lw1 * __uint_as_float(q01[e] << 16)) +
                           lh1 * (lw0 * __uint_as_float(q10[e] << 16) + lw1 * __uint_as_float(q11[e] << 16));
                float hi = lh0 * (lw0 * __uint_as_float(q00[e] & 0xFFFF0000u) + lw1 * __uint_as_float(q01[e] & 0xFFFF0000u)) +
                           lh1 * (lw0 * __uint_as_float(q10[e] & 0xFFFF0000u) + lw1 * __uint_as_float(q11[e] & 0xFFFF0000u));
                lo += __uint_as_float(qa[e] << 16);
                hi += __uint_as_float(qa[e] & 0xFFFF0000u);
                o[e] = pack_bf16x2(lo, hi);
            }
            reinterpret_cast<u32x4*>(out)[i] = o;
        } else {
            const f32x4 v00 = load4<BF>(in, o00), v01 = load4<BF>(in, o01), v10 = load4<BF>(in, o10), v11 = load4<BF>(in, o11);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                r[e] = lh0 * (lw0 * v00[e] + lw1 * v01[e]) + lh1 * (lw0 * v10[e] + lw1 * v11[e]);
            if (add) {
                const f32x4 a = load4<BF>(add, i);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] += a[e];
            }
            store4<BF>(out, i, r);
        }
    }
}

// The same, one block per output ROW (b, ho): the row's two source rows and vertical weights are block-uniform (scalar), an
// item needs one division (by the channel-group count: a shift for the 256-channel CPN maps) instead of three plus 64-bit row
// arithmetic per element.  Same expression per output element as bilinear_kernel: bit-identical.
template <bool BF, int V>
__global__ void bilinear_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, int Ho, int Wo,
                                     float sh, float sw, const float* __restrict__ add, int cv_shift) {
    const int CV = C / V, C4 = C >> 2;
    const unsigned row = blockIdx.x;                                // b * Ho + ho
    const int b = (int)(row / (unsigned)Ho), ho = (int)(row - (unsigned)b * (unsigned)Ho);
    const float fh = sh * ho;
    const int h0 = (int)fh, h1 = h0 + (h0 < H - 1);
    const float lh1 = fh - h0, lh0 = 1.f - lh1;
    const long r0 = ((long)b * H + h0) * W * C4, r1 = ((long)b * H + h1) * W * C4;     // in units of 4 channels
    const long obase = (long)row * Wo * CV;
    const int items = Wo * CV;
    for (int t = threadIdx.x; t < items; t += blockDim.x) {
        const int wo = cv_shift >= 0 ? (t >> cv_shift) : t / CV;
        const int cv = t - wo * CV;
        const float fw = sw * wo;
        const int w0 = (int)fw, w1 = w0 + (w0 < W - 1);
        const float lw1 = fw - w0, lw0 = 1.f - lw1;
        const int c0 = w0 * C4 + cv * (V / 4), c1 = w1 * C4 + cv * (V / 4);
        const long i = obase + t;
        if (V == 8) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4* src = reinterpret_cast<const u32x4*>(in);
            const u32x4 q00 = src[(r0 + c0) >> 1], q01 = src[(r0 + c1) >> 1], q10 = src[(r1 + c0) >> 1], q11 = src[(r1 + c1) >> 1];
            u32x4 qa = {0u, 0u, 0u, 0u};
            if (add) qa = reinterpret_cast<const u32x4*>(add)[i];
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo = lh0 * (lw0 * __uint_as_float(q00[e] << 16) + lw1 * __uint_as_float(q01[e] << 16)) +
                           lh1 * (lw0 * __uint_as_float(q10[e] << 16) + lw1 * __uint_as_float(q11[e] << 16));
                float hi = lh0 * (lw0 * __uint_as_float(q00[e] & 0xFFFF0000u) + lw1 * __uint_as_float(q01[e] & 0xFFFF0000u)) +
                           lh1 * (lw0 * __uint_as_float(q10[e] & 0xFFFF0000u) + lw1 * __uint_as_float(q11[e] & 0xFFFF0000u));
                lo += __uint_as_float(qa[e] << 16);
                hi += __uint_as_float(qa[e] & 0xFFFF0000u);
                o[e] = pack_bf16x2(lo, hi);
            }
            reinterpret_cast<u32x4*>(out)[i] = o;
        } else {
            const f32x4 v00 = load4<BF>(in, r0 + c0), v01 = load4<BF>(in, r0 + c1), v10 = load4<BF>(in, r1 + c0), v11 = load4<BF>(in, r1 + c1);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                r[e] = lh0 * (lw0 * v00[e] + lw1 * v01[e]) + lh1 * (lw0 * v10[e] + lw1 * v11[e]);
            if (add) {
                const f32x4 a = load4<BF>(add, i);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] += a[e];
            }
            store4<BF>(out, i, r);
        }
    }
}

hipError_t launch_bilinear_resize(const float* in, float* out, int B, int H, int W, int C, int Ho, int Wo,
                                  hipStream_t s, int bf16, const float* add) {
    if ((long)B * Ho * Wo >= (1L << 31)) return hipErrorInvalidValue;
    const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int V = (bf16 && C % 8 == 0) ? 8 : 4;
    const int CV = C / V, items = Wo * CV;
    // one block per output row where the map grows more than 2x (CPN refine cascades 0 / 1: 74 -> 54 us, 73 -> 57 us at batch 128);
    // the 2x upsamples and the downsample already move 4.5-5 TB/s with the flat kernel (measured 1-7 % slower by rows)
    if (Ho > 2 * H && items >= 128 && (long)B * Ho < (1L << 31) && (long)H * W * (C >> 2) < (1L << 30)) {
        int sh2 = -1;
        for (int k = 0; k < 12; ++k) if ((1 << k) == CV) sh2 = k;
        const dim3 grid((unsigned)(B * Ho)), blk(items >= 256 ? 256 : 128);
        if (V == 8) hipLaunchKernelGGL((bilinear_rows_kernel<true, 8>), grid, blk, 0, s, in, out, H, W, C, Ho, Wo, sh, sw, add, sh2);
        else if (bf16) hipLaunchKernelGGL((bilinear_rows_kernel<true, 4>), grid, blk, 0, s, in, out, H, W, C, Ho, Wo, sh, sw, add, sh2);
        else hipLaunchKernelGGL((bilinear_rows_kernel<false, 4>), grid, blk, 0, s, in, out, H, W, C, Ho, Wo, sh, sw, add, sh2);
        return hipGetLastError();
    }
    const long total = (long)B * Ho * Wo * (C / V);
    const long want = (total + 255) / 256;
    const dim3 grid((int)(want < 16384 ? want : 16384));
    if (V == 8) hipLaunchKernelGGL((bilinear_kernel<true, 8>), grid, dim3(256), 0, s, in, out, B, H, W, C, Ho, Wo, sh, sw, add);
    else if (bf16) hipLaunchKernelGGL((bilinear_kernel<true, 4>), grid, dim3(256), 0, s, in, out, B, H, W, C, Ho, Wo, sh, sw, add);
    else hipLaunchKernelGGL((bilinear_kernel<false, 4>), grid, dim3(256), 0, s, in, out, B, H, W, C, Ho, Wo, sh, sw, add);
    return hipGetLastError();
}

// ---- many small device-to-device copies as ONE launch (the bias vectors of the lifter's packed linears: capf_lifter_params_changed issued
// 216 hipMemcpyAsync per optimizer step, 0.9 ms of a training step): block b copies segment b of a table that lives in device memory
__global__ __launch_bounds__(256) void copy_segments_kernel(const CopySegment* __restrict__ tab) {
    const CopySegment sg = tab[blockIdx.x];
    for (int i = threadIdx.x; i < sg.n; i += 256) sg.dst[i] = sg.src[i];
}

hipError_t launch_copy_segments(const CopySegment* table_dev, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(copy_segments_kernel, dim3(n), dim3(256), 0, s, table_dev);
    return hipGetLastError();
}

}  // namespace capf
