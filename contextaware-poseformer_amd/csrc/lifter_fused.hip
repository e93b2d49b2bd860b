// Fused front half of the lifting transformer (inference plan): the chains of tiny kernels around the two
// bilinear sampling sites of pose_dformer.py collapse into two kernels, so a forward issues 2 launches where the
// unfused plan issued 9 + 8 per context block (prep_embed, 4 x sample_ref, 4 x feat_embed GEMM | LayerNorm,
// attention/offset GEMM, deformable sampling, 4 x embed_proj GEMM, + the residual add).
//
//   embed_kernel        conpose.py:34-35 (crop keypoints -> ref, IN PLACE) + coord_embed (:214) + grid_sample with
//                       padding zeros at the reference points (:216-218) + feat_embed[l] (:220-221) + pos-embed (:225)
//   ctx_attn_kernel     DeformableBlock.forward pose_dformer.py:115-135: LayerNorm(x_l + x_0), attention_weights /
//                       sampling_offsets, softmax-4 / tanh, border-padded bilinear gather, per-head weighted sum,
//                       embed_proj[l] and the residual add  (the MLP half, :137-138, stays on the MFMA GEMM)
//
// The contractions here have 1-4 rows per (frame, joint): they run as FMA dot products over weights read through
// L1 / L2 (G (frame, joint) pairs share every weight read), not as padded 32-row MFMA tiles.  Built with
// -ffp-contract=off: the bilinear corner indices must stay bit-identical to ATen's (see lifter.hip).
#include "kernels.h"

namespace capf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

typedef BilinearCorner CornerF;     // kernels.h: the one corner rule of both sampling sites
template <bool BORDER>
__device__ __forceinline__ CornerF corner_f(float gx, float gy, int H, int W) { return bilinear_corner<BORDER>(gx, gy, H, W); }
template <bool BF>
__device__ __forceinline__ float ld1(const float* pix, int c) {
    if (!BF) return pix[c];
    return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(pix)[c] << 16);
}
template <bool BF>
__device__ __forceinline__ const float* pixp(const float* base, long pixel_index, int C) {
    if (!BF) return base + pixel_index * C;
    return reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(base) + pixel_index * C);
}

// (frame, joint) pairs per block.  1: a wave's serial chain (sample -> dot products over weights streamed from L2) is what
// bounds these kernels, and B * J * L independent waves only just fill the chip at batch 64 — sharing weight reads
// between pairs (EG = 4 was tried) lengthens every chain 4x and quarters the parallelism: 36 -> 113 us per launch.
static constexpr int EG = 1;

// One block = EG consecutive (b, p) pairs; wave l = level l.  X layout [B, J, L1, C] ("b p l c").
template <bool BF, bool PROJ = true>
__global__ __launch_bounds__(256) void embed_kernel(EmbedArgs a) {
    __shared__ float S[4][EG][512];           // sampled rows, level l: C_l <= 512 channels
    __shared__ float ref_s[EG][2];
    const int l = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bp0 = blockIdx.x * EG;
    // ---- ref = crop / [96, 128] - 1 (IEEE division and subtraction, bit exact), written back in place
    if (threadIdx.x < EG * 2) {
        const int g = threadIdx.x >> 1, xy = threadIdx.x & 1;
        if (bp0 + g < a.BJ) {
            const float v = a.kcrop[(bp0 + g) * 2 + xy];
            const float r = __fsub_rn(__fdiv_rn(v, xy ? 128.0f : 96.0f), 1.0f);
            ref_s[g][xy] = r;
            a.kcrop[(bp0 + g) * 2 + xy] = r;
        } else {
            ref_s[g][xy] = 0.f;
        }
    }
    __syncthreads();
    const int C = a.C;
    // ---- token 0: coord_embed(k2d) + pos[0, p]    (spread over the four waves: wave l takes pair g = l)
    if (l < EG && bp0 + l < a.BJ) {
        const int bp = bp0 + l, p = bp % a.J;
        const float kx = a.k2d[bp * 2 + 0], ky = a.k2d[bp * 2 + 1];
        for (int c = lane; c < C; c += 64)
            a.X[((long)bp * a.L1) * C + c] = (kx * a.cw[c * 2 + 0] + ky * a.cw[c * 2 + 1]) + a.cb[c] + a.pos[(long)p * C + c];
    }
    if (l >= a.L) return;
    // ---- sampling at the reference point, padding zeros (align_corners=True)
    const int H = a.H[l], W = a.W[l], Cl = a.Cl[l];
#pragma unroll
    for (int g = 0; g < EG; ++g) {
        const int bp = bp0 + g;
        if (bp >= a.BJ) {
            for (int c = lane; c < Cl; c += 64) S[l][g][c] = 0.f;
            continue;
        }
        const int b = bp / a.J;
        const CornerF k = corner_f<false>(ref_s[g][0], ref_s[g][1], H, W);
        if (a.idx[l] && lane == 0) {
            a.idx[l][bp * 2 + 0] = k.x0;
            a.idx[l][bp * 2 + 1] = k.y0;
        }
        const bool vx0 = (unsigned)k.x0 < (unsigned)W, vx1 = (unsigned)(k.x0 + 1) < (unsigned)W;
        const bool vy0 = (unsigned)k.y0 < (unsigned)H, vy1 = (unsigned)(k.y0 + 1) < (unsigned)H;
        const float wx0 = 1.0f - k.wx1, wy0 = 1.0f - k.wy1;
        const float w00 = (vx0 && vy0) ? wx0 * wy0 : 0.f, w01 = (vx1 && vy0) ? k.wx1 * wy0 : 0.f;
        const float w10 = (vx0 && vy1) ? wx0 * k.wy1 : 0.f, w11 = (vx1 && vy1) ? k.wx1 * k.wy1 : 0.f;
        const int xa = min(max(k.x0, 0), W - 1), xb = min(max(k.x0 + 1, 0), W - 1);
        const int ya = min(max(k.y0, 0), H - 1), yb = min(max(k.y0 + 1, 0), H - 1);
        const long ib = (long)b * H * W;
        const float* p00 = pixp<BF>(a.feat[l], ib + (long)ya * W + xa, Cl);
        const float* p01 = pixp<BF>(a.feat[l], ib + (long)ya * W + xb, Cl);
        const float* p10 = pixp<BF>(a.feat[l], ib + (long)yb * W + xa, Cl);
        const float* p11 = pixp<BF>(a.feat[l], ib + (long)yb * W + xb, Cl);
        for (int c = lane; c < Cl; c += 64) {
            const float v = ((ld1<BF>(p00, c) * w00 + ld1<BF>(p01, c) * w01) + ld1<BF>(p10, c) * w10) + ld1<BF>(p11, c) * w11;
            S[l][g][c] = v;
            if (a.sampled[l]) a.sampled[l][(long)bp * Cl + c] = v;
        }
    }
    if (!PROJ) return;                        // feat_embed + pos-embed: embed_feat_kernel, behind this launch, off the sampled rows
    __builtin_amdgcn_wave_barrier();          // S[l] is written and read by this wave only
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the LDS writes above have landed
    // ---- feat_embed[l]: X[b, p, 1 + l, j] = S . W_l[j, :] + b_l[j] + pos[1 + l, p, j],  lane -> outputs j, j + 64, ...
    const float* Wl = a.fw[l];
    for (int j = lane; j < C; j += 64) {
        float acc[EG];
#pragma unroll
        for (int g = 0; g < EG; ++g) acc[g] = 0.f;
        const float* wr = Wl + (long)j * 4;                      // quad-interleaved pack: Wq[c / 4][j][4]
#pragma unroll 8
        for (int c = 0; c < Cl; c += 4) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wr + (long)c * C);
#pragma unroll
            for (int g = 0; g < EG; ++g) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(&S[l][g][c]);
                acc[g] += ((s[0] * w[0] + s[1] * w[1]) + s[2] * w[2]) + s[3] * w[3];
            }
        }
        const float bj = a.fb[l][j];
#pragma unroll
        for (int g = 0; g < EG; ++g) {
            const int bp = bp0 + g;
            if (bp < a.BJ) {
                const int p = bp % a.J;
                a.X[((long)bp * a.L1 + 1 + l) * C + j] = (acc[g] + bj) + a.pos[((long)(1 + l) * a.J + p) * C + j];
            }
        }
    }
}

__global__ __launch_bounds__(256) void embed_feat_kernel(EmbedArgs a);       // (below, beside its twin ctx_proj_kernel)

hipError_t launch_embed(const EmbedArgs& a, hipStream_t s) {
    if (a.L > 4 || a.C % 4 != 0) return hipErrorInvalidValue;
    for (int l = 0; l < a.L; ++l)
        if (a.Cl[l] > 512 || a.Cl[l] % 4 != 0) return hipErrorInvalidValue;
    dim3 grid((a.BJ + EG - 1) / EG), block(256);
    bool split = a.C % 32 == 0;
    for (int l = 0; l < a.L; ++l) split = split && a.sampled[l] && a.Cl[l] % 8 == 0 && a.Cl[l] <= 384;
    if (!split) {
        if (a.feat_bf16) hipLaunchKernelGGL(embed_kernel<true>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(embed_kernel<false>, grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.feat_bf16) hipLaunchKernelGGL((embed_kernel<true, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((embed_kernel<false, false>), grid, block, 0, s, a);
    hipLaunchKernelGGL(embed_feat_kernel, dim3((a.BJ + 31) / 32, a.L, a.C / 32), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// DeformableBlock attention half.  One block = CG consecutive (b, p) pairs; wave l = level l (token 1 + l).
//   q      = LayerNorm(X[b,p,1+l] + X[b,p,0])                         (:119-120, eps 1e-5)
//   logit  = q . Wa^T + ba  (NH*NS)      off = tanh(q . Wo^T + bo)  (NH*NS x 2)            (:122-124)
//   pos    = off + ref;  v_s = bilinear_border(feat_l, pos_s);  u_h = sum_s softmax(logit_h)_s v_s      (:126-129, :134)
//   X[b,p,1+l, h*HD + j] += u_h . Wp_l[j, :] + bp_l[j]              (embed_proj is linear and the softmax weights
//                                                                    sum to 1, so it commutes with the sample sum)
// ---------------------------------------------------------------------------------------------------------------
static constexpr int CTX_NH = 4, CTX_NS = 4, CTX_NK = CTX_NH * CTX_NS;      // 4 heads x 4 samples (pose_dformer.py:202)

// four consecutive channels (c % 4 == 0) of the NHWC pixel that starts `elem_off` elements after `base`, as fp32
// (bf16 storage: one 8-byte load)
template <bool BF>
__device__ __forceinline__ f32x4 ld4(const float* base, long elem_off, int c) {
    if (!BF) return *reinterpret_cast<const f32x4*>(base + elem_off + c);
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem_off + c);
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                 __uint_as_float(r.y & 0xffff0000u)};
}

// Work distribution inside a wave (one (b, p, level) per wave).  Everything that is a per-sample scalar in the
// reference — softmax weight, tanh offset, clipped position, corner index, the four bilinear weights — is computed
// ONCE by the lane that owns that sample (16 lanes), not redundantly by all 64 lanes: a wave64 VALU instruction
// costs 4 cycles whatever the number of useful lanes, and 32 tanhf + 16 expf per wave were ~10 us of the serial chain.
#ifndef CTX_EXP
#define CTX_EXP 0                 // (tools/ab_define.sh knock-outs, timing only: 1 no attention / offset dot products, 2 no gather, 4 no embed_proj)
#endif
#ifndef CTX_MIN_BLOCKS
#define CTX_MIN_BLOCKS 1          // (tools/ab_define.sh: A/B builds)
#endif
template <bool BF, bool PROJ = true>
__global__ __launch_bounds__(256, CTX_MIN_BLOCKS) void ctx_attn_kernel(CtxAttnArgs a) {
    extern __shared__ float sm[];
    // per wave (level): Q [C] | AO [3*NK] | SW [NK][4] weights | SO [NK][4] pixel offsets (int) | U [NH * Cl]
    const int C = a.C;
    const int l = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (l >= a.L) return;
    float* base = sm + a.woff[l];
    float* q = base;
    float* ao = q + C;
    float* SW = ao + 3 * CTX_NK;
    int* SO = reinterpret_cast<int*>(SW + 4 * CTX_NK);
    float* U = SW + 8 * CTX_NK;
    const int bp = blockIdx.x;
    const int H = a.H[l], W = a.W[l], Cl = a.Cl[l];
    const int b = bp / a.J;
    // ---- LayerNorm(x_l + x_0), two-pass like the reference kernel; C <= 256 (4 values per lane)
    const float* x0 = a.X + ((long)bp * a.L1) * C;
    const float* xl = x0 + (long)(1 + l) * C;
    float v[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C ? xl[c] + x0[c] : 0.f;
        s += v[i];
    }
    const float mean = wsum(s) / (float)C;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        const float d = c < C ? v[i] - mean : 0.f;
        qq += d * d;
    }
    const float rstd = 1.0f / sqrtf(wsum(qq) / (float)C + a.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < C) q[c] = (v[i] - mean) * rstd * a.ln_g[c] + a.ln_b[c];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- 3*NK dot products of length C: lane k < 48 owns output k of [attention_weights | sampling_offsets]
    if (lane < 3 * CTX_NK) {
        const float* wr = a.Wao + (long)lane * 4;               // quad-interleaved pack: Wq[c / 4][3 * NK outputs][4]
        float acc = 0.f;
#pragma unroll 16
        for (int c = 0; c < ((CTX_EXP & 1) ? 4 : C); c += 4) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wr + (long)c * (3 * CTX_NK));
            const f32x4 t = *reinterpret_cast<const f32x4*>(q + c);
            acc += ((t[0] * w[0] + t[1] * w[1]) + t[2] * w[2]) + t[3] * w[3];
        }
        acc += a.bao[lane];
        ao[lane] = lane < CTX_NK ? acc : tanhf(acc);        // offsets: tanh once, by the owning lane
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- per-sample scalars: lane k < NK owns sample k = h * NS + s
    if (lane < CTX_NK) {
        const int h = lane / CTX_NS;
        float mx = -INFINITY, den = 0.f;
#pragma unroll
        for (int k = 0; k < CTX_NS; ++k) mx = fmaxf(mx, ao[h * CTX_NS + k]);
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < CTX_NS; ++k) {                  // same summation order as the softmax over the 4 samples
            const float e = expf(ao[h * CTX_NS + k] - mx);
            den += e;
            if (k == lane % CTX_NS) mine = e;
        }
        const float ws = mine / den;
        const float px = ao[CTX_NK + 2 * lane + 0] + a.ref[bp * 2 + 0];
        const float py = ao[CTX_NK + 2 * lane + 1] + a.ref[bp * 2 + 1];
        const CornerF cq = corner_f<true>(px, py, H, W);
        if (a.cidx) {                                       // debug taps (capf_set_debug): positions and NW corners
            const long t = ((((long)bp * a.L + l) * CTX_NK) + lane) * 2;
            a.cpos[t] = px; a.cpos[t + 1] = py;
            a.cidx[t] = cq.x0; a.cidx[t + 1] = cq.y0;
        }
        // border mode: coordinates are already clipped; the +1 corner can only fall outside when its weight is
        // exactly 0, so clamping its index is equivalent to ATen's masked load.
        const int xb = min(cq.x0 + 1, W - 1), yb = min(cq.y0 + 1, H - 1);
        const float wx0 = 1.0f - cq.wx1, wy0 = 1.0f - cq.wy1;
        *reinterpret_cast<f32x4*>(SW + lane * 4) = f32x4{ws * (wx0 * wy0), ws * (cq.wx1 * wy0), ws * (wx0 * cq.wy1), ws * (cq.wx1 * cq.wy1)};
        SO[lane * 4 + 0] = (cq.y0 * W + cq.x0) * Cl; SO[lane * 4 + 1] = (cq.y0 * W + xb) * Cl;
        SO[lane * 4 + 2] = (yb * W + cq.x0) * Cl;    SO[lane * 4 + 3] = (yb * W + xb) * Cl;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- gather + weighted sum: a lane owns 4 consecutive channels (one 16-byte load per corner)
    const float* feat = pixp<BF>(a.feat[l], (long)b * H * W, Cl);
    for (int c = lane * 4; c < ((CTX_EXP & 2) ? 0 : Cl); c += 256) {
#pragma unroll
        for (int h = 0; h < CTX_NH; ++h) {
            f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < CTX_NS; ++k) {
                const int kk = h * CTX_NS + k;
                const f32x4 w = *reinterpret_cast<const f32x4*>(SW + kk * 4);
                const f32x4 f0 = ld4<BF>(feat, SO[kk * 4 + 0], c), f1 = ld4<BF>(feat, SO[kk * 4 + 1], c);
                const f32x4 f2 = ld4<BF>(feat, SO[kk * 4 + 2], c), f3 = ld4<BF>(feat, SO[kk * 4 + 3], c);
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] += ((f0[e] * w[0] + f1[e] * w[1]) + f2[e] * w[2]) + f3[e] * w[3];
            }
            *reinterpret_cast<f32x4*>(U + h * Cl + c) = u;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (!PROJ) {                                  // embed_proj runs behind this launch (ctx_proj_kernel): the sums leave as [b p][h][Cl], the wave's last act
        float* ug = a.U[l] + (long)bp * CTX_NH * Cl;   // (stores inside the gather loop sit in front of the next head's loads in the memory counter)
        for (int i = lane * 4; i < CTX_NH * Cl; i += 256) *reinterpret_cast<f32x4*>(ug + i) = *reinterpret_cast<const f32x4*>(U + i);
        return;
    }
    // ---- embed_proj[l] + residual: lane -> (output j, head slot hs); heads hs, hs + nhs, ...
    const int HD = C / CTX_NH;                    // 32 for embed 128
    const float* Wp = a.Wp[l];
    const int nhs = HD <= 64 ? 64 / HD : 1;
    const int hs = HD <= 64 ? lane / HD : 0;
    for (int j = HD <= 64 ? lane % HD : lane; j < HD && hs < nhs; j += 64) {
        float acc[CTX_NH];
#pragma unroll
        for (int t = 0; t < CTX_NH; ++t) acc[t] = 0.f;
        const float* wr = Wp + (long)j * 4;                     // quad-interleaved pack: Wq[c / 4][HD][4]
#pragma unroll 8
        for (int c = 0; c < ((CTX_EXP & 4) ? 4 : Cl); c += 4) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wr + (long)c * HD);
#pragma unroll
            for (int t = 0; t < CTX_NH; ++t) {
                const int h = hs + t * nhs;
                if (h < CTX_NH) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(U + h * Cl + c);
                    acc[t] += ((u[0] * w[0] + u[1] * w[1]) + u[2] * w[2]) + u[3] * w[3];
                }
            }
        }
        const float bj = a.bp[l][j];
        float* xo = a.X + ((long)bp * a.L1 + 1 + l) * C;
#pragma unroll
        for (int t = 0; t < CTX_NH; ++t) {
            const int h = hs + t * nhs;
            if (h < CTX_NH) xo[h * HD + j] += acc[t] + bj;
        }
    }
}

// A 32 x 32 tile of  rows [32 of them, pitch Cl] . W[32 output channels, Cl]^T  with the calling 256-thread block: W in the quad-interleaved pack
// Wq[c / 4][N][4] (N = the pack's output channels, j0 = this tile's first one).  The four waves split K: wave w takes the 8-column steps w, w + 4, ...
// (at most 12 of them: Cl <= 384), ALL of its loads requested before its first MFMA -- one load round trip per wave instead of one per step -- and
// the partial tiles meet in LDS.  True in wave 0 for lanes whose row exists: acc register 4 g + e = output channel j0 + 8 g + 4 (lane >> 5) + e of row
// r0 + (lane & 31).  Lanes 0-31 carry k = c .. c + 3, lanes 32-63 k = c + 4 .. c + 7 of a step: one k of each per v_mfma_f32_32x32x2_f32.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bool rows_tile_f32(const float* __restrict__ rows, int nrows, int r0, int Cl, const float* __restrict__ Wq, int N, int j0,
                                              f32x16_t& acc, float (*part)[16][64]) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int nit = Cl >> 3;
    const int r = r0 + col;
    const bool ok = r < nrows;
    const float* urow = rows + (long)(ok ? r : 0) * Cl + 4 * half;
    const float* wq = Wq + (long)(j0 + col) * 4 + (long)half * ((long)N * 4);
    constexpr int MAXIT = 12;
    f32x4 w[MAXIT], u[MAXIT];
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {
        const int it = wave + 4 * i;                        // (wave-uniform)
        if (it < nit) {
            w[i] = *reinterpret_cast<const f32x4*>(wq + (long)(2 * it) * ((long)N * 4));
            u[i] = *reinterpret_cast<const f32x4*>(urow + 8 * it);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {
        if (wave + 4 * i < nit) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i][e], ok ? u[i][e] : 0.f, acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) part[wave - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (wave > 0 || !ok) return false;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += (part[0][e][lane] + part[1][e][lane]) + part[2][e][lane];
    return true;
#else
    return false;
#endif
}

// embed_proj[l] + residual of ALL levels as one launch (pose_dformer.py:130-135): X[b, p, 1 + l, h HD + j] += U_l[(b, p, h), :] . Wp_l[j, :] + bp_l[j].
// Inside ctx_attn_kernel these were FMA dot products over weights streamed through L2 by every (frame, joint) block -- half of that kernel's
// time at every batch (knock-outs, EXPERIMENTS R6.9) although they are 0.1 GFLOP: a chain of dependent load batches per wave.  Here a block
// takes 32 rows (8 joints x 4 heads) of one level through v_mfma_f32_32x32x2_f32 -- fp32 operands, exact products, fp32 accumulation: the
// arithmetic of the reference's nn.Linear -- with HD = 32 output channels as the MFMA's other dimension.
struct CtxProjArgs {
    const float* U[4]; const float* Wp[4]; const float* bp[4];
    int Cl[4];
    float* X;
    int rows, L1, C;             // rows = BJ * NH
};

__global__ __launch_bounds__(256) void ctx_proj_kernel(CtxProjArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float part[3][16][64];
    const int l = blockIdx.y, lane = threadIdx.x & 63, half = lane >> 5;
    f32x16_t acc;
    if (!rows_tile_f32(a.U[l], a.rows, blockIdx.x * 32, a.Cl[l], a.Wp[l], 32, 0, acc, part)) return;
    const int r = blockIdx.x * 32 + (lane & 31);
    const int bpi = r >> 2, h = r & 3;                     // (CTX_NH == 4)
    float* xo = a.X + ((long)bpi * a.L1 + 1 + l) * a.C + h * 32 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bp[l] + 8 * g + 4 * half);
        f32x4 x = *reinterpret_cast<const f32x4*>(xo + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] += acc[4 * g + e] + b[e];
        *reinterpret_cast<f32x4*>(xo + 8 * g) = x;
    }
#endif
}

// feat_embed[l] + pos-embed of all levels (pose_dformer.py:220-225) the same way, behind embed_kernel<.., false>:
// X[b, p, 1 + l, j] = (S_l[(b, p), :] . W_l[j, :] + b_l[j]) + pos[1 + l, p, j];  grid (row tiles, levels, C / 32)
__global__ __launch_bounds__(256) void embed_feat_kernel(EmbedArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float part[3][16][64];
    const int l = blockIdx.y, j0 = blockIdx.z * 32, lane = threadIdx.x & 63, half = lane >> 5;
    f32x16_t acc;
    if (!rows_tile_f32(a.sampled[l], a.BJ, blockIdx.x * 32, a.Cl[l], a.fw[l], a.C, j0, acc, part)) return;
    const int bp = blockIdx.x * 32 + (lane & 31), p = bp % a.J;
    float* xo = a.X + ((long)bp * a.L1 + 1 + l) * a.C + j0 + 4 * half;
    const float* po = a.pos + ((long)(1 + l) * a.J + p) * a.C + j0 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.fb[l] + j0 + 8 * g + 4 * half);
        const f32x4 pe = *reinterpret_cast<const f32x4*>(po + 8 * g);
        f32x4 x;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (acc[4 * g + e] + b[e]) + pe[e];
        *reinterpret_cast<f32x4*>(xo + 8 * g) = x;
    }
#endif
}

hipError_t launch_ctx_attn(const CtxAttnArgs& a_in, hipStream_t s) {
    CtxAttnArgs a = a_in;
    if (a.NH != CTX_NH || a.NS != CTX_NS || a.L > 4 || a.C > 256 || a.C % 16 != 0) return hipErrorInvalidValue;
    int off = 0;
    for (int l = 0; l < a.L; ++l) {
        if (a.Cl[l] % 4 != 0) return hipErrorInvalidValue;
        a.woff[l] = off;
        off += a.C + 3 * CTX_NK + 8 * CTX_NK + CTX_NH * a.Cl[l];
        off = (off + 3) & ~3;                      // 16-byte aligned sections
    }
    const size_t lds = sizeof(float) * (size_t)off;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    dim3 grid(a.BJ), block(256);
    bool split = a.U[0] != nullptr && a.C == CTX_NH * 32;          // (HD = 32: the MFMA's width)
    for (int l = 0; l < a.L; ++l) split = split && a.U[l] && a.Cl[l] % 8 == 0 && a.Cl[l] <= 384;
    if (!split) {
        if (a.feat_bf16) hipLaunchKernelGGL(ctx_attn_kernel<true>, grid, block, lds, s, a);
        else hipLaunchKernelGGL(ctx_attn_kernel<false>, grid, block, lds, s, a);
        return hipGetLastError();
    }
    if (a.feat_bf16) hipLaunchKernelGGL((ctx_attn_kernel<true, false>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((ctx_attn_kernel<false, false>), grid, block, lds, s, a);
    CtxProjArgs q{};
    for (int l = 0; l < a.L; ++l) { q.U[l] = a.U[l]; q.Wp[l] = a.Wp[l]; q.bp[l] = a.bp[l]; q.Cl[l] = a.Cl[l]; }
    q.X = a.X; q.rows = a.BJ * CTX_NH; q.L1 = a.L1; q.C = a.C;
    hipLaunchKernelGGL(ctx_proj_kernel, dim3((q.rows + 31) / 32, a.L), dim3(256), 0, s, q);
    return hipGetLastError();
}

}  // namespace capf
