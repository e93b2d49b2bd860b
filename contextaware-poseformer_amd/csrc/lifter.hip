// Wavefront-level kernels of the lifting transformer (pose_dformer.py) for gfx950: crop-keypoint
// normalisation + coordinate embedding, bilinear joint-context sampling (both padding modes of the
// reference), LayerNorm, the deformable-sampling reduction, the tiny (5- and 17-token) attention and
// the output head.  Feature maps are NHWC, so a wave reads one pixel's channels as one coalesced run.
// Built with -ffp-contract=off: the bilinear corner indices must be bit-identical to ATen's
// (GridSampler.h:27-36: ((g + 1) / 2) * (size - 1), floor) — no FMA contraction, IEEE division.
#include "kernels.h"

namespace capf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ long rowmap(const RowMap& r, int m) {
    if (r.G == 1) return (long)m * r.S1 + r.off;
    const int q = m / r.G;
    return (long)q * r.S1 + (long)(m - q * r.G) * r.S2 + r.off;
}

// ---- conpose.py:34-35 (in place, 192x256 constants) + coord_embed pose_dformer.py:214 + pos-embed :225
// X layout [B, J, L1, C] ("b p l c"); token 0 = coordinate embedding.
__global__ void prep_embed_kernel(float* __restrict__ kcrop, const float* __restrict__ k2d,
                                  const float* __restrict__ w, const float* __restrict__ bias,
                                  const float* __restrict__ pos, float* __restrict__ X, int BJ, int J, int L1,
                                  int C) {
    const int bp = blockIdx.x;
    if (threadIdx.x == 0) {
        const float x = kcrop[bp * 2 + 0], y = kcrop[bp * 2 + 1];
        kcrop[bp * 2 + 0] = __fsub_rn(__fdiv_rn(x, 96.0f), 1.0f);    // /= 192//2 ; -= 1
        kcrop[bp * 2 + 1] = __fsub_rn(__fdiv_rn(y, 128.0f), 1.0f);   // /= 256//2 ; -= 1
    }
    const float kx = k2d[bp * 2 + 0], ky = k2d[bp * 2 + 1];
    const int p = bp % J;
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        X[((long)bp * L1) * C + c] = (kx * w[c * 2 + 0] + ky * w[c * 2 + 1]) + bias[c] + pos[(long)p * C + c];
}

hipError_t launch_prep_embed(float* kcrop, const float* k2d, const float* w, const float* bias,
                             const float* pos, float* X, int B, int J, int L1, int C, hipStream_t s) {
    hipLaunchKernelGGL(prep_embed_kernel, dim3(B * J), dim3(128), 0, s, kcrop, k2d, w, bias, pos, X, B * J, J,
                       L1, C);
    return hipGetLastError();
}

// ---- bilinear corner computation shared by both sampling sites ------------------------------------
typedef BilinearCorner Corner;      // kernels.h: shared with lifter_fused.hip and the op-level entry point below
template <bool BORDER>
__device__ __forceinline__ Corner corner_of(float gx, float gy, int H, int W) { return bilinear_corner<BORDER>(gx, gy, H, W); }

__global__ void bilinear_corners_kernel(const float* __restrict__ grid, int n, int H, int W, int border, int* __restrict__ idx,
                                        float* __restrict__ frac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gx = grid[i * 2 + 0], gy = grid[i * 2 + 1];
    const Corner c = border ? corner_of<true>(gx, gy, H, W) : corner_of<false>(gx, gy, H, W);
    idx[i * 2 + 0] = c.x0; idx[i * 2 + 1] = c.y0;
    frac[i * 2 + 0] = c.wx1; frac[i * 2 + 1] = c.wy1;
}

hipError_t launch_bilinear_corners(const float* grid, int n, int H, int W, int border, int* idx, float* frac, hipStream_t s) {
    hipLaunchKernelGGL(bilinear_corners_kernel, dim3((n + 255) / 256), dim3(256), 0, s, grid, n, H, W, border, idx, frac);
    return hipGetLastError();
}

// element c of an NHWC pixel stored as fp32 or bf16
template <bool BF>
__device__ __forceinline__ float ldf(const float* pix, int c) {
    if (!BF) return pix[c];
    return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(pix)[c] << 16);
}
// pointer to pixel `pixel_index` (C channels) of a map stored as fp32 or bf16
template <bool BF>
__device__ __forceinline__ const float* pixptr(const float* base, long pixel_index, int C) {
    if (!BF) return base + pixel_index * C;
    return reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(base) + pixel_index * C);
}

// F.grid_sample(features, ref[B,17,1,2], bilinear, zeros, align_corners=True), pose_dformer.py:216-218.
// One wave per (b, p); lanes stride over channels.
template <bool BF>
__global__ void sample_ref_kernel(const float* __restrict__ feat, const float* __restrict__ ref,
                                  float* __restrict__ S, int* __restrict__ idx, int BJ, int J, int H, int W,
                                  int C) {
    const int bp = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (bp >= BJ) return;
    const int lane = threadIdx.x & 63;
    const int b = bp / J;
    const Corner k = corner_of<false>(ref[bp * 2 + 0], ref[bp * 2 + 1], H, W);
    if (idx && lane == 0) {
        idx[bp * 2 + 0] = k.x0;
        idx[bp * 2 + 1] = k.y0;
    }
    const bool vx0 = (unsigned)k.x0 < (unsigned)W, vx1 = (unsigned)(k.x0 + 1) < (unsigned)W;
    const bool vy0 = (unsigned)k.y0 < (unsigned)H, vy1 = (unsigned)(k.y0 + 1) < (unsigned)H;
    const float wx0 = 1.0f - k.wx1, wy0 = 1.0f - k.wy1;
    const float w00 = (vx0 && vy0) ? wx0 * wy0 : 0.f, w01 = (vx1 && vy0) ? k.wx1 * wy0 : 0.f;
    const float w10 = (vx0 && vy1) ? wx0 * k.wy1 : 0.f, w11 = (vx1 && vy1) ? k.wx1 * k.wy1 : 0.f;
    const int xa = min(max(k.x0, 0), W - 1), xb = min(max(k.x0 + 1, 0), W - 1);
    const int ya = min(max(k.y0, 0), H - 1), yb = min(max(k.y0 + 1, 0), H - 1);
    const long ib = (long)b * H * W;
    const float* p00 = pixptr<BF>(feat, ib + (long)ya * W + xa, C);
    const float* p01 = pixptr<BF>(feat, ib + (long)ya * W + xb, C);
    const float* p10 = pixptr<BF>(feat, ib + (long)yb * W + xa, C);
    const float* p11 = pixptr<BF>(feat, ib + (long)yb * W + xb, C);
    for (int c = lane; c < C; c += 64)
        S[(long)bp * C + c] = ((ldf<BF>(p00, c) * w00 + ldf<BF>(p01, c) * w01) + ldf<BF>(p10, c) * w10) + ldf<BF>(p11, c) * w11;
}

hipError_t launch_sample_ref(const float* feat, const float* ref, float* S, int* idx, int B, int J, int H,
                             int W, int C, hipStream_t s, int feat_bf16) {
    const int BJ = B * J;
    if (feat_bf16)
        hipLaunchKernelGGL(sample_ref_kernel<true>, dim3((BJ + 3) / 4), dim3(256), 0, s, feat, ref, S, idx, BJ, J, H, W, C);
    else
    hipLaunchKernelGGL(sample_ref_kernel<false>, dim3((BJ + 3) / 4), dim3(256), 0, s, feat, ref, S, idx, BJ, J, H, W, C);
    return hipGetLastError();
}

// ---- LayerNorm, one wave per row (two-pass, like ATen's CPU kernel) -------------------------------
__device__ __forceinline__ unsigned short f2bf_l(float f) { return to_bf16(f); }    // round-to-nearest-even, like every bf16 store of the path
// OB: the normalised rows are written as bf16 (the A operand of a bf16 MFMA projection, compute_dtype = bf16)
template <int MAXV, bool OB = false>
__global__ void layernorm_kernel(const float* __restrict__ in, RowMap imap, const float* __restrict__ add,
                                 RowMap amap, const float* __restrict__ g, const float* __restrict__ b,
                                 float eps, float* __restrict__ out, int rows, int C) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* x = in + rowmap(imap, r);
    const float* a = add ? add + rowmap(amap, r) : nullptr;
    float v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        float t = 0.f;
        if (c < C) {
            t = x[c];
            if (a) t += a[c];
        }
        v[i] = t;
        s += t;
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        const float d = (c < C) ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
            const float y = (v[i] - mean) * rstd * g[c] + b[c];
            if (OB) reinterpret_cast<unsigned short*>(out)[(long)r * C + c] = f2bf_l(y);
            else out[(long)r * C + c] = y;
        }
    }
}

hipError_t launch_layernorm(const float* in, RowMap imap, const float* add, RowMap amap, const float* g,
                            const float* b, float eps, float* out, int rows, int C, hipStream_t s, int out_bf16) {
    dim3 grid((rows + 3) / 4), block(256);
    if (out_bf16) {
        if (C <= 128)
            hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, block, 0, s, in, imap, add, amap, g, b, eps, out, rows, C);
        else if (C <= 640)
            hipLaunchKernelGGL((layernorm_kernel<10, true>), grid, block, 0, s, in, imap, add, amap, g, b, eps, out, rows, C);
        else if (C <= 1536)
            hipLaunchKernelGGL((layernorm_kernel<24, true>), grid, block, 0, s, in, imap, add, amap, g, b, eps, out, rows, C);
        else
            return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (C <= 128)
        hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, in, imap, add, amap, g, b, eps, out, rows, C);
    else if (C <= 640)
        hipLaunchKernelGGL(layernorm_kernel<10>, grid, block, 0, s, in, imap, add, amap, g, b, eps, out, rows, C);
    else if (C <= 1536)
        hipLaunchKernelGGL(layernorm_kernel<24>, grid, block, 0, s, in, imap, add, amap, g, b, eps, out, rows, C);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- deformable sampling (DeformableBlock.forward pose_dformer.py:122-135) ------------------------
// softmax over the NS samples of a head, tanh offsets, pos = off + ref, border-padded bilinear gather,
// weighted sum over samples.  Because embed_proj is linear and the softmax weights sum to 1,
//   sum_s w_s (W v_s + b) = W (sum_s w_s v_s) + b,
// so only U = sum_s w_s v_s  [(b,p,h), C_l] is produced here and the projection is a GEMM with
// M = B*17*4 rows instead of B*17*16 (the [B,17,16,C_l] tensor is never materialised).
// One block per (b, p); wave w handles level w; lanes stride over channels.
template <bool BF>
__device__ __forceinline__ f32x4 ldq(const float* pix, int q) {       // channels 4 q .. 4 q + 3 of a pixel
    if (!BF) return *reinterpret_cast<const f32x4*>(pix + 4 * q);
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(pix) + 4 * q);
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}

template <int NS, bool BF>
__global__ void deform_sample_kernel(DeformArgs a) {
    const int bp = blockIdx.x;
    const int l = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    if (l >= a.L) return;
    const int b = bp / a.J;
    const int H = a.H[l], W = a.W[l], C = a.C[l];
    const float* feat = pixptr<BF>(a.feat[l], (long)b * H * W, C);
    const int nk = a.NH * NS;
    const float* ao = a.AO + ((long)bp * a.L + l) * (a.ld_ao ? a.ld_ao : 3 * nk);
    const float rx = a.ref[bp * 2 + 0], ry = a.ref[bp * 2 + 1];
    float* U = a.U[l] + (long)bp * a.NH * C;
    // G lanes per head, one channel QUAD each (16-byte corner loads; G = the largest power of two <= min(64, C / 4)), 64 / G heads at a
    // time: at 32 channels all four heads' 64 corner loads are in flight at once.  Per channel the same expression as before, same bits.
    const int Q = C >> 2;
    int G = 64;
    while (G > Q) G >>= 1;
    const int grp = lane / G, ql = lane - grp * G, PP = 64 / G;
    for (int h0 = 0; h0 < a.NH; h0 += PP) {
        const bool live = h0 + grp < a.NH;
        const int h = live ? h0 + grp : a.NH - 1;
        float lg[NS], mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) { lg[s] = ao[h * NS + s]; mx = fmaxf(mx, lg[s]); }
        float den = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) { lg[s] = expf(lg[s] - mx); den += lg[s]; }
        const float* p00[NS]; const float* p01[NS]; const float* p10[NS]; const float* p11[NS];
        float w00[NS], w01[NS], w10[NS], w11[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float ws = lg[s] / den;
            const int k = h * NS + s;
            const float px = tanhf(ao[nk + 2 * k + 0]) + rx;
            const float py = tanhf(ao[nk + 2 * k + 1]) + ry;
            const Corner q = corner_of<true>(px, py, H, W);
            if (a.cidx && live && ql == 0) {                // debug taps (capf_set_debug): positions and NW corners
                const long t = ((((long)bp * a.L + l) * nk) + k) * 2;
                a.cpos[t] = px; a.cpos[t + 1] = py;
                a.cidx[t] = q.x0; a.cidx[t + 1] = q.y0;
            }
            // border mode: coordinates are already clipped; the +1 corner can only fall outside when its
            // weight is exactly 0, so clamping its index is equivalent to ATen's masked load.
            const int xb = min(q.x0 + 1, W - 1), yb = min(q.y0 + 1, H - 1);
            const float wx0 = 1.0f - q.wx1, wy0 = 1.0f - q.wy1;
            w00[s] = ws * (wx0 * wy0); w01[s] = ws * (q.wx1 * wy0);
            w10[s] = ws * (wx0 * q.wy1); w11[s] = ws * (q.wx1 * q.wy1);
            p00[s] = pixptr<BF>(feat, (long)q.y0 * W + q.x0, C); p01[s] = pixptr<BF>(feat, (long)q.y0 * W + xb, C);
            p10[s] = pixptr<BF>(feat, (long)yb * W + q.x0, C);   p11[s] = pixptr<BF>(feat, (long)yb * W + xb, C);
        }
        if (!live) continue;
        for (int cq = ql; cq < Q; cq += G) {
            f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const f32x4 f00 = ldq<BF>(p00[s], cq), f01 = ldq<BF>(p01[s], cq), f10 = ldq<BF>(p10[s], cq), f11 = ldq<BF>(p11[s], cq);
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] += ((f00[e] * w00[s] + f01[e] * w01[s]) + f10[e] * w10[s]) + f11[e] * w11[s];
            }
            *reinterpret_cast<f32x4*>(U + (long)h * C + 4 * cq) = u;
        }
    }
}

hipError_t launch_deform_sample(const DeformArgs& a, hipStream_t s) {
    if (a.NS != 4 || a.L > 4) return hipErrorInvalidValue;
    for (int l = 0; l < a.L; ++l)
        if (a.C[l] < 4 || (a.C[l] & 3)) return hipErrorInvalidValue;
    if (a.feat_bf16) hipLaunchKernelGGL((deform_sample_kernel<4, true>), dim3(a.B * a.J), dim3(64 * a.L), 0, s, a);
    else hipLaunchKernelGGL((deform_sample_kernel<4, false>), dim3(a.B * a.J), dim3(64 * a.L), 0, s, a);
    return hipGetLastError();
}

// store 4 consecutive attention outputs as fp32 or (OB) bf16
template <bool OB>
__device__ __forceinline__ void st4(float* out, long idx, f32x4 v) {
    if (!OB) { *reinterpret_cast<f32x4*>(out + idx) = v; return; }
    unsigned short* o = reinterpret_cast<unsigned short*>(out) + idx;
    uint2 pk;
    pk.x = pack_bf16x2(v[0], v[1]);
    pk.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(o) = pk;
}

// ---- tiny attention (Attention.forward pose_dformer.py:46-59): 5 or 17 tokens per group -----------
// qkv row layout [3][heads][d] (the reshape at :49).  One thread per (group, head, query).
template <int NMAX, bool OB = false>
__global__ void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int groups, int N,
                                 int heads, int d, float scale) {
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long total = (long)groups * heads * N;
    if (t >= total) return;
    const int i = (int)(t % N);
    const int h = (int)((t / N) % heads);
    const long g = t / ((long)N * heads);
    const int Cq = 3 * heads * d;
    const float* q = qkv + (g * N + i) * Cq + h * d;
    const float* kbase = qkv + (g * N) * Cq + heads * d + h * d;
    const float* vbase = qkv + (g * N) * Cq + 2 * heads * d + h * d;
    float sc[NMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        float s = 0.f;
        if (j < N) {
            const float* k = kbase + (long)j * Cq;
            for (int c = 0; c < d; c += 4) {
                const f32x4 qa = *reinterpret_cast<const f32x4*>(q + c);
                const f32x4 ka = *reinterpret_cast<const f32x4*>(k + c);
                s += qa[0] * ka[0] + qa[1] * ka[1] + qa[2] * ka[2] + qa[3] * ka[3];
            }
            s *= scale;
            mx = fmaxf(mx, s);
        }
        sc[j] = s;
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        sc[j] = (j < N) ? expf(sc[j] - mx) : 0.f;
        den += sc[j];
    }
    const float inv = 1.0f / den;
    const long o = (g * N + i) * (long)(heads * d) + h * d;
    for (int c = 0; c < d; c += 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            if (j < N) {
                const f32x4 va = *reinterpret_cast<const f32x4*>(vbase + (long)j * Cq + c);
                acc += va * (sc[j] * inv);
            }
        }
        st4<OB>(out, o + c, acc);
    }
}

// Same computation with PARTS lanes per query: each lane owns d/PARTS of the head dimension (partial
// q.k dot products are combined with two shuffles, every lane then accumulates its own slice of P.V).
// 4x the threads of the kernel above for the 17-token joint attention, where B*8*17 threads cannot fill
// 256 CUs.
template <int NMAX, int PARTS, bool OB = false>
__global__ void attention_split_kernel(const float* __restrict__ qkv, float* __restrict__ out, int groups, int N,
                                       int heads, int d, float scale) {
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long total = (long)groups * heads * N * PARTS;
    const bool live = t < total;
    const long tq = live ? t / PARTS : 0;
    const int part = (int)(t % PARTS);
    const int i = (int)(tq % N);
    const int h = (int)((tq / N) % heads);
    const long g = tq / ((long)N * heads);
    const int Cq = 3 * heads * d;
    const int dp = d / PARTS, c0 = part * dp;
    const float* q = qkv + (g * N + i) * Cq + h * d + c0;
    const float* kbase = qkv + (g * N) * Cq + heads * d + h * d + c0;
    const float* vbase = qkv + (g * N) * Cq + 2 * heads * d + h * d + c0;
    float sc[NMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        float s = 0.f;
        if (j < N) {
            const float* k = kbase + (long)j * Cq;
            for (int c = 0; c < dp; c += 4) {
                const f32x4 qa = *reinterpret_cast<const f32x4*>(q + c);
                const f32x4 ka = *reinterpret_cast<const f32x4*>(k + c);
                s += qa[0] * ka[0] + qa[1] * ka[1] + qa[2] * ka[2] + qa[3] * ka[3];
            }
#pragma unroll
            for (int o = 1; o < PARTS; o <<= 1) s += __shfl_xor(s, o, 64);
            s *= scale;
            mx = fmaxf(mx, s);
        }
        sc[j] = s;
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        sc[j] = (j < N) ? expf(sc[j] - mx) : 0.f;
        den += sc[j];
    }
    const float inv = 1.0f / den;
    if (!live) return;
    const long o = (g * N + i) * (long)(heads * d) + h * d + c0;
    for (int c = 0; c < dp; c += 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            if (j < N) {
                const f32x4 va = *reinterpret_cast<const f32x4*>(vbase + (long)j * Cq + c);
                acc += va * (sc[j] * inv);
            }
        }
        st4<OB>(out, o + c, acc);
    }
}

// The 17-token joint attention (B * 8 (group, head) pairs of 17 x 80 floats): one wave per pair, q / k / v
// staged in LDS with coalesced reads, 17 x 17 scores, one softmax row per lane, P.V with d-contiguous stores.
// (The per-query kernels above read k / v rows straight from global memory: 36 us per block at B = 64.)
template <int NMAX, bool OB = false>
__global__ __launch_bounds__(256) void attention_lds_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N,
                                                           int heads, int d, float scale) {
    extern __shared__ float sm[];
    const int ld = d + 1;
    float* q = sm;
    float* k = q + NMAX * ld;
    float* v = k + NMAX * ld;
    float* P = v + NMAX * ld;                  // [NMAX][NMAX + 1]
    const int lane = threadIdx.x;
    const long g = blockIdx.x / heads;
    const int h = blockIdx.x - (int)g * heads;
    const int Cq = 3 * heads * d;
    const float* qb = qkv + (g * N) * Cq + h * d;
    const int nd = N * d;
    for (int idx = lane; idx < nd; idx += blockDim.x) {
        const int t = idx / d, c = idx - t * d;
        const float* row = qb + (long)t * Cq + c;
        q[t * ld + c] = row[0];
        k[t * ld + c] = row[heads * d];
        v[t * ld + c] = row[2 * heads * d];
    }
    __syncthreads();
    for (int idx = lane; idx < N * N; idx += blockDim.x) {
        const int i = idx / N, j = idx - i * N;
        float s = 0.f;
        for (int c = 0; c < d; ++c) s += q[i * ld + c] * k[j * ld + c];
        P[i * (NMAX + 1) + j] = s * scale;
    }
    __syncthreads();
    if (lane < N) {
        float* p = P + lane * (NMAX + 1);
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, p[j]);
        float den = 0.f;
        for (int j = 0; j < N; ++j) { p[j] = expf(p[j] - mx); den += p[j]; }
        const float inv = 1.0f / den;
        for (int j = 0; j < N; ++j) p[j] *= inv;
    }
    __syncthreads();
    const long ob = (g * N) * (long)(heads * d) + h * d;
    for (int idx = lane; idx < nd; idx += blockDim.x) {
        const int t = idx / d, c = idx - t * d;
        float acc = 0.f;
        for (int j = 0; j < N; ++j) acc += P[t * (NMAX + 1) + j] * v[j * ld + c];
        if (OB) reinterpret_cast<unsigned short*>(out)[ob + (long)t * (heads * d) + c] = f2bf_l(acc);
        else out[ob + (long)t * (heads * d) + c] = acc;
    }
}

hipError_t launch_attention(const float* qkv, float* out, int groups, int N, int heads, int d, hipStream_t s, int out_bf16) {
    if (d % 4 != 0) return hipErrorInvalidValue;
    const float scale = 1.0f / sqrtf((float)d);
    if (N > 5 && N <= 17 && (size_t)(3 * 17 * (d + 1) + 17 * 18) * sizeof(float) <= 64 * 1024 && (long)groups * heads <= 0x7fffffffL) {
        const size_t lds = (size_t)(3 * 17 * (d + 1) + 17 * 18) * sizeof(float);
        const dim3 grid((unsigned)((long)groups * heads)), block(256);
        if (out_bf16) hipLaunchKernelGGL((attention_lds_kernel<17, true>), grid, block, lds, s, qkv, out, N, heads, d, scale);
        else hipLaunchKernelGGL((attention_lds_kernel<17, false>), grid, block, lds, s, qkv, out, N, heads, d, scale);
        return hipGetLastError();
    }
    if (d % 16 == 0 && N <= 17) {
        const long total = (long)groups * heads * N * 4;
        dim3 grid((unsigned)((total + 255) / 256)), block(256);
        if (N <= 5) {
            if (out_bf16) hipLaunchKernelGGL((attention_split_kernel<5, 4, true>), grid, block, 0, s, qkv, out, groups, N, heads, d, scale);
            else hipLaunchKernelGGL((attention_split_kernel<5, 4, false>), grid, block, 0, s, qkv, out, groups, N, heads, d, scale);
        } else {
            if (out_bf16) hipLaunchKernelGGL((attention_split_kernel<17, 4, true>), grid, block, 0, s, qkv, out, groups, N, heads, d, scale);
            else hipLaunchKernelGGL((attention_split_kernel<17, 4, false>), grid, block, 0, s, qkv, out, groups, N, heads, d, scale);
        }
        return hipGetLastError();
    }
    const long total = (long)groups * heads * N;
    dim3 grid((unsigned)((total + 127) / 128)), block(128);
    if (N <= 5) {
        if (out_bf16) hipLaunchKernelGGL((attention_kernel<5, true>), grid, block, 0, s, qkv, out, groups, N, heads, d, scale);
        else hipLaunchKernelGGL((attention_kernel<5, false>), grid, block, 0, s, qkv, out, groups, N, heads, d, scale);
    } else if (N <= 17) {
        if (out_bf16) hipLaunchKernelGGL((attention_kernel<17, true>), grid, block, 0, s, qkv, out, groups, N, heads, d, scale);
        else hipLaunchKernelGGL((attention_kernel<17, false>), grid, block, 0, s, qkv, out, groups, N, heads, d, scale);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- head: LayerNorm(eps 1e-5) + Linear(C -> NO) (pose_dformer.py:205-208, :240), one wave per row --
template <int MAXV>
__global__ void head_kernel(const float* __restrict__ X, const float* __restrict__ g, const float* __restrict__ b,
                            float eps, const float* __restrict__ w, const float* __restrict__ wb,
                            float* __restrict__ out, int rows, int C, int NO) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* x = X + (long)r * C;
    float v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < C) ? x[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        const float dlt = (c < C) ? v[i] - mean : 0.f;
        q += dlt * dlt;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < C) ? (v[i] - mean) * rstd * g[c] + b[c] : 0.f;
    }
    for (int o = 0; o < NO; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < C) acc += v[i] * w[(long)o * C + c];
        }
        acc = wave_sum(acc);
        if (lane == 0) out[(long)r * NO + o] = acc + wb[o];
    }
}

hipError_t launch_head(const float* X, const float* g, const float* b, float eps, const float* w,
                       const float* wb, float* out, int rows, int C, int NO, hipStream_t s) {
    dim3 grid((rows + 3) / 4), block(256);
    if (C <= 640)
        hipLaunchKernelGGL(head_kernel<10>, grid, block, 0, s, X, g, b, eps, w, wb, out, rows, C, NO);
    else if (C <= 1536)
        hipLaunchKernelGGL(head_kernel<24>, grid, block, 0, s, X, g, b, eps, w, wb, out, rows, C, NO);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace capf
