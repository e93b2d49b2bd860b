// Kernels of the lifter's TRAINING step: LayerNorm with saved statistics
// and its backward, deterministic column reductions (bias / LayerNorm-affine / pos-embed gradients),
// GELU forward/backward on saved pre-activations, zero-padded transposes that turn the weight- and
// input-gradient products into the K-contiguous GEMMs of igemm_f32.hip, backward of the tiny attention
// and of the deformable sampler (gradient w.r.t. sampling positions and attention logits only — the
// feature maps come from the frozen backbone, conpose.py:22-25), MPJPE loss + gradient (loss.py:16-22)
// and a fused AdamW update (train.py:345).  No atomics: every reduction has a fixed order.
#include <algorithm>
#include "igemm_f32h2_ws_tile.h"
#include "kernels.h"

namespace capf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ long rowmap_t(const RowMap& r, int m) {
    if (r.G == 1) return (long)m * r.S1 + r.off;
    const int q = m / r.G;
    return (long)q * r.S1 + (long)(m - q * r.G) * r.S2 + r.off;
}

// ---- LayerNorm forward that keeps xhat = (x - mean) * rstd and rstd for the backward --------------
template <int MAXV>
__global__ void layernorm_train_kernel(const float* __restrict__ in, RowMap imap, const float* __restrict__ add,
                                       RowMap amap, const float* __restrict__ g, const float* __restrict__ b,
                                       float eps, float* __restrict__ out, float* __restrict__ xhat,
                                       float* __restrict__ rstd_out, int rows, int C) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* x = in + rowmap_t(imap, r);
    const float* a = add ? add + rowmap_t(amap, r) : nullptr;
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
    const float mean = wave_sum_t(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        const float d = (c < C) ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum_t(q) / (float)C + eps);
    if (lane == 0) rstd_out[r] = rstd;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
            const float h = (v[i] - mean) * rstd;
            xhat[(long)r * C + c] = h;
            out[(long)r * C + c] = h * g[c] + b[c];
        }
    }
}

hipError_t launch_layernorm_train(const float* in, RowMap imap, const float* add, RowMap amap, const float* g,
                                  const float* b, float eps, float* out, float* xhat, float* rstd, int rows, int C,
                                  hipStream_t s) {
    dim3 grid((rows + 3) / 4), block(256);
    if (C <= 128)
        hipLaunchKernelGGL(layernorm_train_kernel<2>, grid, block, 0, s, in, imap, add, amap, g, b, eps, out, xhat, rstd, rows, C);
    else if (C <= 640)
        hipLaunchKernelGGL(layernorm_train_kernel<10>, grid, block, 0, s, in, imap, add, amap, g, b, eps, out, xhat, rstd, rows, C);
    else if (C <= 1536)
        hipLaunchKernelGGL(layernorm_train_kernel<24>, grid, block, 0, s, in, imap, add, amap, g, b, eps, out, xhat, rstd, rows, C);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- LayerNorm backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma --------
// One wave handles GRP consecutive rows; dx is ACCUMULATED into dX at omap(row) and, when `second` is
// given, the sum over the group is accumulated at smap(first row of the group) (the query of a
// DeformableBlock is LN(x_l + x_0): token 0 receives the sum over the 4 level rows, pose_dformer.py:120).
template <int MAXV>
__global__ void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                     float* __restrict__ dX, RowMap omap, float* __restrict__ second, RowMap smap,
                                     int groups, int GRP, int C) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= groups) return;
    const int lane = threadIdx.x & 63;
    float acc2[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) acc2[i] = 0.f;
    for (int t = 0; t < GRP; ++t) {
        const int r = w * GRP + t;
        float gv[MAXV], hv[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            gv[i] = hv[i] = 0.f;
            if (c < C) {
                gv[i] = dy[(long)r * C + c] * gamma[c];
                hv[i] = xhat[(long)r * C + c];
            }
            s1 += gv[i];
            s2 += gv[i] * hv[i];
        }
        const float m1 = wave_sum_t(s1) / (float)C, m2 = wave_sum_t(s2) / (float)C;
        const float rs = rstd[r];
        float* dst = dX + rowmap_t(omap, r);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < C) {
                const float dx = rs * (gv[i] - m1 - hv[i] * m2);
                dst[c] += dx;
                acc2[i] += dx;
            }
        }
    }
    if (second) {
        float* dst = second + rowmap_t(smap, w * GRP);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < C) dst[c] += acc2[i];
        }
    }
}

hipError_t launch_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dX,
                                RowMap omap, float* second, RowMap smap, int rows, int GRP, int C, hipStream_t s) {
    const int groups = rows / GRP;
    dim3 grid((groups + 3) / 4), block(256);
    if (C <= 128)
        hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, block, 0, s, dy, xhat, rstd, gamma, dX, omap, second, smap, groups, GRP, C);
    else if (C <= 640)
        hipLaunchKernelGGL(layernorm_bwd_kernel<10>, grid, block, 0, s, dy, xhat, rstd, gamma, dX, omap, second, smap, groups, GRP, C);
    else if (C <= 1536)
        hipLaunchKernelGGL(layernorm_bwd_kernel<24>, grid, block, 0, s, dy, xhat, rstd, gamma, dX, omap, second, smap, groups, GRP, C);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- deterministic column reduction:  dst[c] = sum_r A[amap(r) + c] * B(r, c) -----------------------
// B absent -> 1; bmode 1: B[bmap(r) + c]; bmode 2: B[bmap(r)] (one scalar per row).
// Stage 1: grid (C/64, chunks), 256 threads = 16 row lanes x 16 column quads, fixed row order per lane and a
// fixed 16-way LDS reduce; stage 2 sums the chunk partials in order.  No atomics.
// `partial2` (optional): the plain column sums of A as a second result of the same pass (LayerNorm: d(gamma) =
// sum dY * xhat and d(beta) = sum dY read dY once).
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ A, RowMap amap, const float* __restrict__ Bm, RowMap bmap,
                                 int bmode, float* __restrict__ partial, float* __restrict__ partial2, int rows, int C,
                                 int rows_per_chunk) {
    // 256 threads = 16 row lanes x 16 column quads (64 columns, 16-byte loads); four rows in flight per lane
    __shared__ float red[2][16][64];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 64 + cq * 4;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(rows, r0 + rows_per_chunk);
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, acc2[4] = {0.f, 0.f, 0.f, 0.f};
    auto load4 = [&](const float* p, float (&v)[4], bool vec) {
        if (vec) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = col + e < C ? p[e] : 0.f;
        }
    };
    const bool avec = col + 4 <= C && ((amap.S1 | amap.S2 | amap.off) & 3) == 0 && ((size_t)A & 15) == 0;
    const bool bvec = bmode == 1 && col + 4 <= C && ((bmap.S1 | bmap.S2 | bmap.off) & 3) == 0 && ((size_t)Bm & 15) == 0;
    if (col < C) {
        for (int r = r0 + rl; r < r1; r += 64) {              // same row order per lane whatever the unrolling: r, r+16, r+32, ...
            float a[4][4], b[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + 16 * u;
                if (rr < r1) {
                    load4(A + rowmap_t(amap, rr) + col, a[u], avec);
                    if (bmode == 1) load4(Bm + rowmap_t(bmap, rr) + col, b[u], bvec);
                    else if (bmode == 2) { const float sc = Bm[rowmap_t(bmap, rr)]; b[u][0] = b[u][1] = b[u][2] = b[u][3] = sc; }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + 16 * u < r1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[e] += bmode ? a[u][e] * b[u][e] : a[u][e];
                        acc2[e] += a[u][e];
                    }
                }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][rl][cq * 4 + e] = acc[e];
        red[1][rl][cq * 4 + e] = acc2[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {                                  // fixed 16-way reduce, one output column per thread
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        float* dstp = which ? partial2 : partial;
        if (dstp && blockIdx.x * 64 + c < C) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[which][k][c];
            dstp[(long)blockIdx.y * C + blockIdx.x * 64 + c] = t;
        }
    }
}

// stage 2: 64 columns per block as 16 column quads x 16 chunk lanes; each lane sums every 16th chunk partial in order (four
// 16-byte loads in flight), then a fixed 16-way LDS reduce
__device__ __forceinline__ void colreduce_final_body(float (&red)[2][16][64], const int bx, const float* __restrict__ partial, int chunks, int C,
                                                     float* __restrict__ dst, long dst_stride, int accumulate,
                                                     const float* __restrict__ partial2, float* __restrict__ dst2) {
    const int cq = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int col = bx * 64 + cq * 4;
    const bool vec = col + 4 <= C && (C & 3) == 0;
    auto sum_col = [&](const float* src, float (&acc)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = 0.f;
        if (col >= C) return;
        for (int k = kl; k < chunks; k += 64) {
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + 16 * u;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = 0.f;
                if (kk < chunks) {
                    const float* q = src + (long)kk * C + col;
                    if (vec) { const float4 t = *reinterpret_cast<const float4*>(q); v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
                    else { for (int e = 0; e < 4; ++e) if (col + e < C) v[u][e] = q[e]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[u][e];
        }
    };
    float s1[4], s2[4];
    sum_col(partial, s1);
    if (dst2) sum_col(partial2, s2);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][kl][cq * 4 + e] = s1[e];
        red[1][kl][cq * 4 + e] = dst2 ? s2[e] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = bx * 64 + (threadIdx.x & 63);
        if (c < C && (which == 0 || dst2)) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[which][k][threadIdx.x & 63];
            if (which == 0) {
                float* d = dst + (long)c * dst_stride;
                *d = accumulate ? *d + t : t;
            } else {
                dst2[c] = t;
            }
        }
    }
}

__global__ __launch_bounds__(256) void colreduce_final_kernel(const float* __restrict__ partial, int chunks, int C,
                                                              float* __restrict__ dst, long dst_stride, int accumulate,
                                                              const float* __restrict__ partial2, float* __restrict__ dst2) {
    __shared__ float red[2][16][64];
    colreduce_final_body(red, blockIdx.x, partial, chunks, C, dst, dst_stride, accumulate, partial2, dst2);
}

// the second stages of many column reductions as ONE launch (the backward defers them: Engine::t_col_flush); same body, same bits
__global__ __launch_bounds__(256) void colreduce_final_batch_kernel(ColFinalBatch b) {
    __shared__ float red[2][16][64];
    int j = 0;
    while (j + 1 < b.count && (int)blockIdx.x >= b.blk_end[j]) ++j;
    const int bx = (int)blockIdx.x - (j ? b.blk_end[j - 1] : 0);
    colreduce_final_body(red, bx, b.partial[j], b.chunks[j], b.C[j], b.dst[j], b.dst_stride[j], 0, b.partial2[j], b.dst2[j]);
}

hipError_t launch_colreduce_final_batch(const ColFinalBatch& b, hipStream_t s) {
    if (b.count <= 0) return hipSuccess;
    hipLaunchKernelGGL(colreduce_final_batch_kernel, dim3((unsigned)b.blk_end[b.count - 1]), dim3(256), 0, s, b);
    return hipGetLastError();
}

int colreduce_chunks(int rows, int C, int nout, size_t scratch_elems, int* rows_per_chunk) {
    const int colblocks = (C + 63) / 64;
    int chunks = (2048 + colblocks - 1) / colblocks;
    chunks = std::min(std::min(chunks, 512), std::max(1, rows / 64));
    if (scratch_elems) chunks = std::min<long>(chunks, std::max<long>(1, (long)(scratch_elems / ((size_t)C * nout))));
    else chunks = std::min(chunks, 64);
    const int rpc = (rows + chunks - 1) / chunks;
    if (rows_per_chunk) *rows_per_chunk = rpc;
    return (rows + rpc - 1) / rpc;
}

// scratch must hold (dst2 ? 2 : 1) * chunks * C floats; chunks is chosen so that the first stage has ~2048 blocks
hipError_t launch_colreduce(const float* A, RowMap amap, const float* Bm, RowMap bmap, int bmode, int rows, int C,
                            float* dst, long dst_stride, int accumulate, float* scratch, hipStream_t s, float* dst2,
                            size_t scratch_elems, ColFinalBatch* defer) {
    const int colblocks = (C + 63) / 64, nout = dst2 ? 2 : 1;
    int rpc = 0;
    const int chunks = colreduce_chunks(rows, C, nout, scratch_elems, &rpc);
    float* p2 = dst2 ? scratch + (size_t)chunks * C : nullptr;
    hipLaunchKernelGGL(colreduce_kernel, dim3(colblocks, chunks), dim3(256), 0, s, A, amap, Bm, bmap, bmode, scratch, p2, rows, C,
                       rpc);
    if (defer && !accumulate && defer->count < COL_BATCH_MAX) {      // second stage with the other reductions' (the scratch stays the caller's until then)
        const int j = defer->count++;
        defer->partial[j] = scratch; defer->partial2[j] = p2; defer->dst[j] = dst; defer->dst2[j] = dst2;
        defer->chunks[j] = chunks; defer->C[j] = C; defer->dst_stride[j] = (int)dst_stride;
        defer->blk_end[j] = (j ? defer->blk_end[j - 1] : 0) + colblocks;
        return hipGetLastError();
    }
    hipLaunchKernelGGL(colreduce_final_kernel, dim3(colblocks), dim3(256), 0, s, scratch, chunks, C, dst, dst_stride,
                       accumulate, p2, dst2);
    return hipGetLastError();
}

// dst[i] (+)= sum_k slabs[k][i]  (split-K partial slabs of the weight-gradient GEMMs)
__global__ void slab_sum_kernel(const float* __restrict__ slabs, int nslab, long n, float* __restrict__ dst) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nslab; ++k) s += slabs[(long)k * n + i];
        dst[i] = s;
    }
}

hipError_t launch_slab_sum(const float* slabs, int nslab, long n, float* dst, hipStream_t s) {
    const long want = (n + 255) / 256;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, s, slabs, nslab, n, dst);
    return hipGetLastError();
}

// one launch for many (slabs -> gradient) sums: a block takes 1024 consecutive elements of one job (16 bytes per lane), eight slabs in flight
__global__ __launch_bounds__(256) void slab_sum_batch_kernel(SlabBatch b) {
    int j = 0;
    while (j + 1 < b.count && (int)blockIdx.x >= b.blk_end[j]) ++j;      // (block-uniform: scalar loads from the kernel arguments)
    const int first = j ? b.blk_end[j - 1] : 0;
    const long n = b.n[j];
    const long i = ((long)((int)blockIdx.x - first) * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float* __restrict__ src = b.src[j] + i;
    const int ns = b.nslab[j];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= ns; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (long)(k + u) * n);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; k < ns; ++k) acc += *reinterpret_cast<const f32x4*>(src + (long)k * n);
    float* dst = b.dst[j] + i;
    if (((size_t)dst & 15) == 0) *reinterpret_cast<f32x4*>(dst) = acc;
    else { dst[0] = acc[0]; dst[1] = acc[1]; dst[2] = acc[2]; dst[3] = acc[3]; }
}

hipError_t launch_slab_sum_batch(const SlabBatch& b, hipStream_t s) {
    if (b.count <= 0) return hipSuccess;
    hipLaunchKernelGGL(slab_sum_batch_kernel, dim3((unsigned)b.blk_end[b.count - 1]), dim3(256), 0, s, b);
    return hipGetLastError();
}

// ---- exact (erf) GELU on a saved pre-activation, and its derivative --------------------------------
__global__ void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}

__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float cdf = 0.5f * (1.0f + erff(v[e] * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * v[e] * v[e]);
            o[e] = g[e] * (cdf + v[e] * pdf);
        }
        reinterpret_cast<f32x4*>(dx)[i] = o;
    }
}

hipError_t launch_gelu_fwd(const float* x, float* y, long n, hipStream_t s) {
    const long n4 = n / 4, want = (n4 + 255) / 256;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, s, x, y, n4);
    return hipGetLastError();
}

hipError_t launch_gelu_bwd(const float* x, const float* dy, float* dx, long n, hipStream_t s) {
    const long n4 = n / 4, want = (n4 + 255) / 256;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, s, x, dy, dx, n4);
    return hipGetLastError();
}

// ---- out[c][m] = in[imap(m) + c] for m < M, 0 for M <= m < Mp  (32x32 LDS tiles) -------------------
__global__ void transpose_pad_kernel(const float* __restrict__ in, RowMap imap, int M, int C, float* __restrict__ out,
                                     int Mp) {
    __shared__ float tile[32][33];
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: 8 rows of 32
    for (int k = ty; k < 32; k += 8) {
        const int m = m0 + k, c = c0 + tx;
        tile[k][tx] = (m < M && c < C) ? in[rowmap_t(imap, m) + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, m = m0 + tx;
        if (c < C && m < Mp) out[(long)c * Mp + m] = tile[tx][k];
    }
}

hipError_t launch_transpose_pad(const float* in, RowMap imap, int M, int C, float* out, int Mp, hipStream_t s) {
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((Mp + 31) / 32, (C + 31) / 32), dim3(256), 0, s, in, imap, M, C, out, Mp);
    return hipGetLastError();
}

// ---- weight gradient of an nn.Linear WITHOUT transposes:  dW[n][k] = sum_m dY[m][n] X[m][k],  db[n] = sum_m dY[m][n] ---------------
// Both operands are contiguous along their OUTPUT index, not along the contraction index m, which is why t_linear_bwd used to transpose
// them (two transpose_pad launches + two column-reduction launches per linear: 153 + 156 tiny launches per step).  Here
// 32-row tiles [32 m][64 n] of dY and [32 m][64 k] of X go to LDS as they lie in memory (LDS-DMA, 256 contiguous bytes per row) and the
// MFMA fragments are read across them: lane (i, h) of v_mfma_f32_32x32x2_f32 needs element (m + h, i) = one ds_read2st64_b32 for two
// k-steps (row stride = 64 dwords), 32 consecutive lanes = 32 consecutive dwords: conflict-free.  64 x 64 output tile per block (four
// waves, 32 x 32 each), three-stage ring, split over m in grid.y (each slice writes a raw slab, summed by slab_sum like before); the blocks
// of the first k tile also sum their dY tile's columns: the bias gradient, as N more floats behind the slice's N * K slab.
// Requires N % 4 == 0, K % 4 == 0 (16-byte quads; ragged 64-tiles load zeros) and plain row pitches (the ctx blocks' strided token maps keep the old path).
struct WgradArgs {
    const float* dY; long ldy;      // [M][ldy], columns 0 .. N - 1
    const float* X; long ldx;       // [M][ldx], columns 0 .. K - 1
    float* out;                     // slice s writes N * K (+ N) floats at out + s * slab
    long slab;
    int M, N, K, cps, want_bias;    // cps = 32-row chunks per slice
};

__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int S = 3, TILE = 32 * 64;                   // floats per operand tile
    __shared__ __attribute__((aligned(16))) float lds[S * 2 * TILE + 4 * 64];
    typedef __attribute__((address_space(3))) void* lptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tk = (a.K + 63) >> 6;                         // (ragged N / K: columns beyond them load as zeros and are not stored)
    const int tile_n = blockIdx.x / tk, tile_k = blockIdx.x - tile_n * tk;
    const int n0 = tile_n * 64, k0 = tile_k * 64;
    const int c_begin = blockIdx.y * a.cps;
    const int chunks_total = (a.M + 31) >> 5;
    const int c_end = min(chunks_total, c_begin + a.cps);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dY + n0), 0, 0x7FFFFF00u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + k0), 0, 0x7FFFFF00u, 0x00020000);
    // DMA: instruction i (0..7) of a tile covers rows 4 i .. 4 i + 3 (16 lanes x 16 B per row); a wave issues i = wave and wave + 4
    const int drow = lane >> 4, dq = lane & 15;
    const bool y_col = n0 + dq * 4 < a.N, x_col = k0 + dq * 4 < a.K;
    auto fire = [&](int c, int stage) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = wave + 4 * h;
            const long m = (long)c * 32 + 4 * i + drow;
            const bool ok = c < c_end && m < a.M;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lptr)(lds + (stage * 2) * TILE + i * 256), 16, ok && y_col ? (unsigned)((m * a.ldy + dq * 4) * 4) : 0x80000000u, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr)(lds + (stage * 2 + 1) * TILE + i * 256), 16, ok && x_col ? (unsigned)((m * a.ldx + dq * 4) * 4) : 0x80000000u, 0, 0, 0);
        }
    };
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
    const int fi = lane & 31, fh = lane >> 5;
    float bsum = 0.f;                                      // (tile_k == 0) this thread's column n0 + (tid & 63), rows (tid >> 6) * 8 .. + 7 of every chunk
    fire(c_begin, 0);
    fire(c_begin + 1, 1);
    int st = 0;
    for (int c = c_begin; c < c_end; ++c) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this chunk's four instructions of mine have landed (the next chunk's four may fly)
        __builtin_amdgcn_s_barrier();
        fire(c + 2, st == 0 ? 2 : st - 1);                 // (into the stage everybody finished reading before this barrier)
        const float* ys = lds + (st * 2) * TILE;
        const float* xs = ys + TILE;
#pragma unroll
        for (int mm = 0; mm < 32; mm += 4) {               // lane half fh takes rows mm + fh and mm + 2 + fh: two MFMA k-steps per read pair
            const float a0 = ys[(mm + fh) * 64 + wn + fi], a1 = ys[(mm + 2 + fh) * 64 + wn + fi];
            const float b0 = xs[(mm + fh) * 64 + wk + fi], b1 = xs[(mm + 2 + fh) * 64 + wk + fi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
        }
        if (a.want_bias && tile_k == 0) {
            const int col = tid & 63, r0 = (tid >> 6) * 8;
#pragma unroll
            for (int r = 0; r < 8; ++r) bsum += ys[(r0 + r) * 64 + col];
        }
        st = st == 2 ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // D[i][j]: lane holds column j = k0 + wk + fi, register 4 g + e = row n0 + wn + 8 g + 4 fh + e
    float* o = a.out + (long)blockIdx.y * a.slab;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + wn + 8 * g + 4 * fh + e, kk = k0 + wk + fi;
            if (n < a.N && kk < a.K) o[(long)n * a.K + kk] = acc[4 * g + e];
        }
    if (a.want_bias && tile_k == 0) {
        float* red = lds + S * 2 * TILE;
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < 64 && n0 + tid < a.N) o[(long)a.N * a.K + n0 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    }
#endif
}

// The same product on the 16-bit matrix pipe (the step's linears with N and K multiples of 128): both operands are ACTIVATIONS, so both are
// split on the way from LDS to the MFMA -- two fp16 pieces under a power-of-two scale per wave, operand and 32-row chunk, three piece
// products (igemm_f32h2_ws_tile.h: 2^-21 of a term at worst, below the fp32 accumulation error of a sum over thousands of rows).  A lane's
// fragment for v_mfma_f32_32x32x16_f16 is eight consecutive m of one column: eight ds_read_b32 down the [32 m][128] tile, 32 consecutive
// lanes = 32 consecutive dwords.  128 x 128 outputs per block, 64 x 64 per wave: a split fragment feeds two MFMA blocks, 48 VALU
// instructions per MFMA triple instead of 96 (the split, not the matrix pipe, bounds this kernel).  The scales only ever go DOWN along a
// slice's chunks (an operand's largest value so far decides): a chunk of small values under a scale made for larger ones loses
// nothing the fp32 sum it joins would keep (absolute error 2^-39 of that largest value per term), no accumulator ever scales up, and an
// all-zero chunk (DropPath) costs no rescale.  Two-stage ring of [32][128] fp32 tiles (64 KiB), two blocks per CU.
__global__ __launch_bounds__(256, 2) void wgrad_tn_h2_kernel(WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int S = 2, TW = 128, TILE = 32 * TW;
    __shared__ __attribute__((aligned(16))) float lds[S * 2 * TILE + 256];
    typedef __attribute__((address_space(3))) void* lptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tk = a.K >> 7;
    const int tile_n = blockIdx.x / tk, tile_k = blockIdx.x - tile_n * tk;
    const int n0 = tile_n * TW, k0 = tile_k * TW;
    const int c_begin = blockIdx.y * a.cps;
    const int chunks_total = (a.M + 31) >> 5;
    const int c_end = min(chunks_total, c_begin + a.cps);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dY + n0), 0, 0x7FFFFF00u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + k0), 0, 0x7FFFFF00u, 0x00020000);
    // DMA: instruction i (0..15) of a tile covers rows 2 i, 2 i + 1 (32 lanes x 16 B per row); a wave issues i = wave, wave + 4, wave + 8, wave + 12
    const int drow = lane >> 5, dq = lane & 31;
    auto fire = [&](int c, int stage) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int i = wave + 4 * h;
            const long m = (long)c * 32 + 2 * i + drow;
            const bool ok = c < c_end && m < a.M;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lptr)(lds + (stage * 2) * TILE + i * 256), 16, ok ? (unsigned)((m * a.ldy + dq * 4) * 4) : 0x80000000u, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr)(lds + (stage * 2 + 1) * TILE + i * 256), 16, ok ? (unsigned)((m * a.ldx + dq * 4) * 4) : 0x80000000u, 0, 0, 0);
        }
    };
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
    const int fi = lane & 31, fh = lane >> 5;
    float bsum = 0.f;                                      // (tile_k == 0) column n0 + (tid & 127), rows (tid >> 7) * 16 .. + 15 of every chunk
    int sa = 190, sb = 190;                                // biased exponents of the two operand scales so far (h2_scale_exp's largest: nothing seen yet)
    fire(c_begin, 0);
    for (int c = c_begin; c < c_end; ++c) {
        const int st = (c - c_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // chunk c has landed for everybody; everybody is done reading the other stage
        fire(c + 1, st ^ 1);
        const float* ys = lds + (st * 2) * TILE;
        const float* xs = ys + TILE;
        float av[2][16], bv[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = (j >> 3) * 16 + 8 * fh + (j & 7);
                av[t][j] = ys[row * TW + wn + 32 * t + fi];
                bv[t][j] = xs[row * TW + wk + 32 * t + fi];
            }
        float ma = 0.f, mb = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) { ma = fmaxf(ma, fabsf(av[t][j])); mb = fmaxf(mb, fabsf(bv[t][j])); }
        const int na = min(sa, h2_scale_exp(h2_wave_max(ma))), nb = min(sb, h2_scale_exp(h2_wave_max(mb)));
        if (na + nb != sa + sb) {                          // (wave-uniform) the accumulators move DOWN to the new scales: exact
            const float f = __int_as_float(max(0, 127 + (na + nb) - (sa + sb)) << 23);     // (below 2^-126: gone, as it would be in the sum)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][u][r] *= f;
        }
        sa = na; sb = nb;
        const float fa = __int_as_float(sa << 23), fb = __int_as_float(sb << 23);
        ws_f16x8 a1[2][2], a2[2][2], b1[2][2], b2[2][2];   // [32-wide block][16-deep step]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ws_u32x4 pa1, pa2, pb1, pb2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned x1, x2;
                    h2_split2(av[t][ks * 8 + 2 * q], av[t][ks * 8 + 2 * q + 1], fa, x1, x2);
                    pa1[q] = x1; pa2[q] = x2;
                    h2_split2(bv[t][ks * 8 + 2 * q], bv[t][ks * 8 + 2 * q + 1], fb, x1, x2);
                    pb1[q] = x1; pb2[q] = x2;
                }
                a1[t][ks] = __builtin_bit_cast(ws_f16x8, pa1); a2[t][ks] = __builtin_bit_cast(ws_f16x8, pa2);
                b1[t][ks] = __builtin_bit_cast(ws_f16x8, pb1); b2[t][ks] = __builtin_bit_cast(ws_f16x8, pb2);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[t][ks], b1[u][ks], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[t][ks], b2[u][ks], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[t][ks], b1[u][ks], acc[t][u], 0, 0, 0);
                }
        if (a.want_bias && tile_k == 0) {
            const int col = tid & 127, r0 = (tid >> 7) * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) bsum += ys[(r0 + r) * TW + col];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // D[i][j]: lane holds column j = k0 + wk + 32 u + fi, register 4 g + e = row n0 + wn + 32 t + 8 g + 4 fh + e
    const float inv = __int_as_float((381 - sa - sb) << 23);
    float* o = a.out + (long)blockIdx.y * a.slab;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[(long)(n0 + wn + 32 * t + 8 * g + 4 * fh + e) * a.K + k0 + wk + 32 * u + fi] = acc[t][u][4 * g + e] * inv;
    if (a.want_bias && tile_k == 0) {
        float* red = lds + S * 2 * TILE;
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < 128) o[(long)a.N * a.K + n0 + tid] = red[tid] + red[128 + tid];
    }
#endif
}

// slabs: at least splits * (N * K + N) floats; *splits_out slices were written (1: straight into out = dW, bias at out + N * K)
hipError_t launch_wgrad_tn(const float* dY, long ldy, const float* X, long ldx, int M, int N, int K, float* out, long slab, int splits,
                           int want_bias, hipStream_t s, bool h2) {
    if (h2 && (N % 128 || K % 128)) return hipErrorInvalidValue;
    if (N % 4 || K % 4 || M <= 0 || splits < 1 || (double)M * (double)ldy * 4.0 >= 2.0e9 || (double)M * (double)ldx * 4.0 >= 2.0e9)
        return hipErrorInvalidValue;
    WgradArgs a{dY, ldy, X, ldx, out, slab, M, N, K, 0, want_bias};
    const int chunks = (M + 31) / 32;
    a.cps = (chunks + splits - 1) / splits;
    const int slices = (chunks + a.cps - 1) / a.cps;
    if (h2) hipLaunchKernelGGL(wgrad_tn_h2_kernel, dim3((N / 128) * (K / 128), slices), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(wgrad_tn_kernel, dim3(((N + 63) / 64) * ((K + 63) / 64), slices), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- backward of the tiny attention (Attention.forward pose_dformer.py:46-59) -----------------------
// One block (256 threads for 17 tokens, one wave for 5) per (group, head): q, k, v and dO of the N <= 17 tokens are staged in LDS (coalesced reads of the
// d contiguous floats of each token), the N x N probabilities are recomputed from the saved qkv exactly as
// the forward computes them, and dq / dk / dv go out as d contiguous floats per token.
//   S = scale q k^T, P = softmax(S), dP = dO v^T, dS = P o (dP - rowsum(P o dP)),
//   dq = scale dS k, dk = scale dS^T q, dv = P^T dO
template <int NMAX>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                           float* __restrict__ dqkv, int N, int heads, int d, float scale) {
    extern __shared__ float sm[];
    const int ld = d + 1;                      // padded token stride: rows of different tokens fall in different banks
    float* q = sm;
    float* k = q + NMAX * ld;
    float* v = k + NMAX * ld;
    float* go = v + NMAX * ld;
    float* P = go + NMAX * ld;                 // [NMAX][NMAX + 1]
    float* dS = P + NMAX * (NMAX + 1);
    const int lane = threadIdx.x;
    const long g = blockIdx.x / heads;
    const int h = blockIdx.x - (int)g * heads;
    const int Cq = 3 * heads * d, Co = heads * d;
    const float* qb = qkv + (g * N) * Cq + h * d;
    const float* dob = dO + (g * N) * Co + h * d;
    const int nd = N * d;
    for (int idx = lane; idx < nd; idx += blockDim.x) {
        const int t = idx / d, c = idx - t * d;
        const float* row = qb + (long)t * Cq + c;
        q[t * ld + c] = row[0];
        k[t * ld + c] = row[heads * d];
        v[t * ld + c] = row[2 * heads * d];
        go[t * ld + c] = dob[(long)t * Co + c];
    }
    __syncthreads();
    for (int idx = lane; idx < N * N; idx += blockDim.x) {
        const int i = idx / N, j = idx - i * N;
        float s = 0.f, dp = 0.f;
        for (int c = 0; c < d; ++c) {
            s += q[i * ld + c] * k[j * ld + c];
            dp += go[i * ld + c] * v[j * ld + c];
        }
        P[i * (NMAX + 1) + j] = s * scale;
        dS[i * (NMAX + 1) + j] = dp;
    }
    __syncthreads();
    if (lane < N) {                            // one softmax row per lane
        float* p = P + lane * (NMAX + 1);
        float* ds = dS + lane * (NMAX + 1);
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, p[j]);
        float den = 0.f;
        for (int j = 0; j < N; ++j) { p[j] = expf(p[j] - mx); den += p[j]; }
        float D = 0.f;
        for (int j = 0; j < N; ++j) { p[j] /= den; D += p[j] * ds[j]; }
        for (int j = 0; j < N; ++j) ds[j] = p[j] * (ds[j] - D);
    }
    __syncthreads();
    float* dq = dqkv + (g * N) * Cq + h * d;
    for (int idx = lane; idx < nd; idx += blockDim.x) {
        const int t = idx / d, c = idx - t * d;
        float sq = 0.f, sk = 0.f, sv = 0.f;
        for (int j = 0; j < N; ++j) {
            sq += dS[t * (NMAX + 1) + j] * k[j * ld + c];
            sk += dS[j * (NMAX + 1) + t] * q[j * ld + c];
            sv += P[j * (NMAX + 1) + t] * go[j * ld + c];
        }
        float* o = dq + (long)t * Cq + c;
        o[0] = sq * scale;
        o[heads * d] = sk * scale;
        o[2 * heads * d] = sv;
    }
}

hipError_t launch_attention_bwd(const float* qkv, const float* dO, float* dqkv, int groups, int N, int heads, int d,
                                hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)d);
    const long pairs = (long)groups * heads;
    if (pairs <= 0) return hipSuccess;
    if (pairs > 0x7fffffffL) return hipErrorInvalidValue;
    dim3 grid((unsigned)pairs), block(N <= 5 ? 64 : 256);
    if (N <= 5) {
        const size_t lds = (size_t)(4 * 5 * (d + 1) + 2 * 5 * 6) * sizeof(float);
        hipLaunchKernelGGL(attention_bwd_kernel<5>, grid, block, lds, s, qkv, dO, dqkv, N, heads, d, scale);
    } else if (N <= 17) {
        const size_t lds = (size_t)(4 * 17 * (d + 1) + 2 * 17 * 18) * sizeof(float);
        if (lds > 64 * 1024) return hipErrorInvalidValue;
        hipLaunchKernelGGL(attention_bwd_kernel<17>, grid, block, lds, s, qkv, dO, dqkv, N, heads, d, scale);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- backward of the deformable sampler w.r.t. attention logits and offset pre-activations ----------
// forward (lifter.hip): U[c] = sum_s w_s v_s[c],  w = softmax(logit),  v_s = bilinear_border(feat, tanh(o_s) + ref)
//   dw_s   = sum_c dU[c] v_s[c]
//   dpos_s = w_s * sum_c dU[c] dv_s[c]/dpos  * (size-1)/2 * [coordinate not clipped]   (ATen grid_sampler
//            backward: clip_coordinates_set_grad zeroes the gradient where the unnormalised coordinate is
//            <= 0 or >= size-1)
//   dlogit = softmax backward, do = dpos * (1 - tanh^2)
// One block per (b, p); wave = level; lanes stride over channels; wave-level reductions.
template <bool BF>
__device__ __forceinline__ float ldf_t(const float* pix, int c) {
    if (!BF) return pix[c];
    return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(pix)[c] << 16);
}
template <bool BF>
__device__ __forceinline__ const float* pixptr_t(const float* base, long pixel_index, int C) {
    if (!BF) return base + pixel_index * C;
    return reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(base) + pixel_index * C);
}

template <bool BF>
__device__ __forceinline__ f32x4 ld4_t(const float* pix, int q) {      // channels 4 q .. 4 q + 3 of a pixel
    if (!BF) return *reinterpret_cast<const f32x4*>(pix + 4 * q);
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(pix) + 4 * q);
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
}

// One block per (b, p); wave = level.  A wave works on 64 / G (head, sample) pairs at a time, G lanes per pair with one channel QUAD each
// (16-byte corner loads; G = the largest power of two <= min(64, C / 4)): at 32 channels eight pairs' corners are in flight per wave
// instead of one pair on half the lanes, and a pair's three sums reduce over G lanes instead of 64.  The pairs' results meet in 384 bytes
// of LDS per wave for the softmax backward.
template <int NS, bool BF>
__global__ void deform_bwd_kernel(DeformArgs a, float* __restrict__ dAO, int ldd) {
    __shared__ float part[4][16][6];                  // [level][pair]: dw, gx, gy, softmax weight, tanh(ox), tanh(oy)
    const int bp = blockIdx.x;
    const int l = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    if (l >= a.L) return;
    const int b = bp / a.J;
    const int H = a.H[l], W = a.W[l], C = a.C[l];
    const float* feat = pixptr_t<BF>(a.feat[l], (long)b * H * W, C);
    const int nk = a.NH * NS;
    const long row = (long)bp * a.L + l;
    const float* ao = a.AO + row * ldd;
    float* dao = dAO + row * ldd;
    const float rx = a.ref[bp * 2 + 0], ry = a.ref[bp * 2 + 1];
    const float* dU = a.dU[l] + (long)bp * a.NH * C;
    const int Q = C >> 2;
    int G = 64;
    while (G > Q) G >>= 1;
    const int grp = lane / G, ql = lane - grp * G, PP = 64 / G;
    for (int k0 = 0; k0 < nk; k0 += PP) {
        const bool live = k0 + grp < nk;
        const int k = live ? k0 + grp : nk - 1;
        const int h = k / NS, s = k - h * NS;
        float lg[NS], mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NS; ++t) { lg[t] = ao[h * NS + t]; mx = fmaxf(mx, lg[t]); }
        float den = 0.f, mine = 0.f;
#pragma unroll
        for (int t = 0; t < NS; ++t) { const float e = expf(lg[t] - mx); den += e; if (t == s) mine = e; }
        const float wsk = mine / den;
        const float th0 = tanhf(ao[nk + 2 * k + 0]), th1 = tanhf(ao[nk + 2 * k + 1]);
        const float ux = ((th0 + rx + 1.0f) / 2.0f) * (float)(W - 1);
        const float uy = ((th1 + ry + 1.0f) / 2.0f) * (float)(H - 1);
        const float mxk = (ux <= 0.f || ux >= (float)(W - 1)) ? 0.f : 1.f;
        const float myk = (uy <= 0.f || uy >= (float)(H - 1)) ? 0.f : 1.f;
        const float x = fminf((float)(W - 1), fmaxf(ux, 0.f)), y = fminf((float)(H - 1), fmaxf(uy, 0.f));
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf;
        const float wx1 = x - xf, wy1 = y - yf, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool vx = x0 + 1 <= W - 1, vy = y0 + 1 <= H - 1;       // +1 corner inside the map
        const int xb = vx ? x0 + 1 : x0, yb = vy ? y0 + 1 : y0;
        const float* p00 = pixptr_t<BF>(feat, (long)y0 * W + x0, C);
        const float* p01 = pixptr_t<BF>(feat, (long)y0 * W + xb, C);
        const float* p10 = pixptr_t<BF>(feat, (long)yb * W + x0, C);
        const float* p11 = pixptr_t<BF>(feat, (long)yb * W + xb, C);
        const float c00 = wx0 * wy0, c01 = wx1 * wy0, c10 = wx0 * wy1, c11 = wx1 * wy1;
        float a_dw = 0.f, a_gx = 0.f, a_gy = 0.f;
        for (int q = ql; q < Q; q += G) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(dU + (long)h * C + 4 * q);
            const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 f00 = ld4_t<BF>(p00, q), f01 = vx ? ld4_t<BF>(p01, q) : zero, f10 = vy ? ld4_t<BF>(p10, q) : zero,
                        f11 = (vx && vy) ? ld4_t<BF>(p11, q) : zero;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a_dw += g[e] * (((f00[e] * c00 + f01[e] * c01) + f10[e] * c10) + f11[e] * c11);
                a_gx += g[e] * ((f01[e] - f00[e]) * wy0 + (f11[e] - f10[e]) * wy1);
                a_gy += g[e] * ((f10[e] - f00[e]) * wx0 + (f11[e] - f01[e]) * wx1);
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1) {            // (G is a power of two: the xor partners stay inside the pair's lanes)
            a_dw += __shfl_xor(a_dw, o, 64);
            a_gx += __shfl_xor(a_gx, o, 64);
            a_gy += __shfl_xor(a_gy, o, 64);
        }
        if (live && ql == 0) {
            float* pk = part[l][k];
            pk[0] = a_dw;
            pk[1] = a_gx * wsk * 0.5f * (float)(W - 1) * mxk;
            pk[2] = a_gy * wsk * 0.5f * (float)(H - 1) * myk;
            pk[3] = wsk; pk[4] = th0; pk[5] = th1;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();                      // (a wave reads back only what it wrote itself)
    if (lane < nk) {
        const int k = lane, h = k / NS;
        float dotw = 0.f;
#pragma unroll
        for (int t = 0; t < NS; ++t) dotw += part[l][h * NS + t][3] * part[l][h * NS + t][0];
        const float* pk = part[l][k];
        dao[k] = pk[3] * (pk[0] - dotw);
        dao[nk + 2 * k + 0] = pk[1] * (1.f - pk[4] * pk[4]);
        dao[nk + 2 * k + 1] = pk[2] * (1.f - pk[5] * pk[5]);
    }
    for (int k = 3 * nk + lane; k < ldd; k += 64) dao[k] = 0.f;      // padding columns of the 64-wide row
}

hipError_t launch_deform_bwd(const DeformArgs& a, float* dAO, int ldd, hipStream_t s) {
    if (a.NS != 4 || a.L > 4 || a.NH * a.NS > 16) return hipErrorInvalidValue;
    for (int l = 0; l < a.L; ++l)
        if (a.C[l] < 4 || (a.C[l] & 3)) return hipErrorInvalidValue;
    if (a.feat_bf16) hipLaunchKernelGGL((deform_bwd_kernel<4, true>), dim3(a.B * a.J), dim3(64 * a.L), 0, s, a, dAO, ldd);
    else hipLaunchKernelGGL((deform_bwd_kernel<4, false>), dim3(a.B * a.J), dim3(64 * a.L), 0, s, a, dAO, ldd);
    return hipGetLastError();
}

// ---- MPJPE (loss.py:16-22): loss = mean_r ||pred_r - gt_r||_2 ; dpred = (pred - gt) / (||.|| * rows) ---
__global__ void mpjpe_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int rows,
                             float* __restrict__ loss, float* __restrict__ dpred, float gscale) {
    __shared__ float red[1024];
    float acc = 0.f;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const float dx = pred[r * 3 + 0] - gt[r * 3 + 0], dy = pred[r * 3 + 1] - gt[r * 3 + 1],
                    dz = pred[r * 3 + 2] - gt[r * 3 + 2];
        const float n = sqrtf(dx * dx + dy * dy + dz * dz);
        acc += n;
        if (dpred) {
            const float inv = n > 0.f ? gscale / (n * (float)rows) : 0.f;
            dpred[r * 3 + 0] = dx * inv; dpred[r * 3 + 1] = dy * inv; dpred[r * 3 + 2] = dz * inv;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)rows;
}

hipError_t launch_mpjpe(const float* pred, const float* gt, int rows, float* loss, float* dpred, float gscale,
                        hipStream_t s) {
    hipLaunchKernelGGL(mpjpe_kernel, dim3(1), dim3(1024), 0, s, pred, gt, rows, loss, dpred, gscale);
    return hipGetLastError();
}

// the same loss for any width D of the last axis (the reference's MPJPE takes [..., D]: 2-D keypoints, loss.py:16-22)
__global__ void mpjpe_nd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int rows, int D,
                                float* __restrict__ loss, float* __restrict__ dpred, float gscale) {
    __shared__ float red[1024];
    float acc = 0.f;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        float q = 0.f;
        for (int c = 0; c < D; ++c) {
            const float d = pred[(long)r * D + c] - gt[(long)r * D + c];
            q += d * d;
        }
        const float n = sqrtf(q);
        acc += n;
        if (dpred) {
            const float inv = n > 0.f ? gscale / (n * (float)rows) : 0.f;
            for (int c = 0; c < D; ++c) dpred[(long)r * D + c] = (pred[(long)r * D + c] - gt[(long)r * D + c]) * inv;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)rows;
}

hipError_t launch_mpjpe_nd(const float* pred, const float* gt, int rows, int D, float* loss, float* dpred, float gscale,
                           hipStream_t s) {
    hipLaunchKernelGGL(mpjpe_nd_kernel, dim3(1), dim3(1024), 0, s, pred, gt, rows, D, loss, dpred, gscale);
    return hipGetLastError();
}

// ---- fused AdamW over one flat parameter buffer (torch.optim.AdamW semantics, train.py:345) -----------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, float gscale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;        // 1 / world_size after a SUM all-reduce (exact for 1.0f)
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = pi;
    }
}

hipError_t launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                        float wd, int step, hipStream_t s, float gscale) {
    const float bc1 = 1.0f - powf(b1, (float)step), bc2s = sqrtf(1.0f - powf(b2, (float)step));
    const long want = (n + 255) / 256;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2,
                       eps, wd, bc1, bc2s, gscale);
    return hipGetLastError();
}

// dst[r, :] = src[smap(r), :] * scale[r / div]
__global__ void scale_rows_kernel(const float* __restrict__ src, RowMap smap, const float* __restrict__ scale, int div,
                                  float* __restrict__ dst, int rows, int C) {
    const long total = (long)rows * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i - (long)r * C);
        dst[i] = src[rowmap_t(smap, r) + c] * (scale ? scale[r / div] : 1.0f);
    }
}

hipError_t launch_scale_rows(const float* src, RowMap smap, const float* scale, int div, float* dst, int rows, int C,
                             hipStream_t s) {
    const long want = ((long)rows * C + 255) / 256;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, s, src, smap, scale, div,
                       dst, rows, C);
    return hipGetLastError();
}

__global__ void head_dgrad_kernel(const float* __restrict__ dOut, const float* __restrict__ W, float* __restrict__ dY,
                                  int rows, int C, int NO) {
    const long total = (long)rows * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i - (long)r * C);
        float s = 0.f;
        for (int o = 0; o < NO; ++o) s += dOut[r * NO + o] * W[(long)o * C + c];
        dY[i] = s;
    }
}

hipError_t launch_head_dgrad(const float* dOut, const float* W, float* dY, int rows, int C, int NO, hipStream_t s) {
    const long want = ((long)rows * C + 255) / 256;
    hipLaunchKernelGGL(head_dgrad_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, s, dOut, W, dY, rows, C, NO);
    return hipGetLastError();
}

__global__ void pos_grad_kernel(const float* __restrict__ dX, float* __restrict__ dpos, int B, int J, int L1, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // over (l, p, c)
    if (i >= L1 * J * C) return;
    const int c = i % C, p = (i / C) % J, l = i / (C * J);
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dX[(((long)b * J + p) * L1 + l) * C + c];
    dpos[i] = s;
}

hipError_t launch_pos_grad(const float* dX, float* dpos, int B, int J, int L1, int C, hipStream_t s) {
    const int n = L1 * J * C;
    hipLaunchKernelGGL(pos_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dX, dpos, B, J, L1, C);
    return hipGetLastError();
}

}  // namespace capf
