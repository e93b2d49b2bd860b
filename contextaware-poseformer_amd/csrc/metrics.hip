// Evaluation metrics of the path's callers, one pose per thread:
//   MPJPE    ContextPose/mvn/models/loss.py:16-22
//   P_MPJPE  loss.py:25-68   (similarity Procrustes: scale, rotation, translation)
//   N_MPJPE  loss.py:71-84   (scale only)
//   MPJVE    loss.py:87-101  (first differences along the frame axis of the evaluated subset)
// and their per-action aggregation (datasets/human36m.py:358-417), plus the three validity-masked keypoint
// losses train.py:21 imports (loss.py:104-137).
//
// The reference computes P_MPJPE with a batched 3x3 numpy SVD on the host.  Here the optimal rotation comes from
// Horn's closed form instead: the unit quaternion that maximises sum_i x_i . (R y_i) is the top eigenvector of a
// symmetric 4x4 matrix built from H = sum_i x_i y_i^T, its eigenvalue IS trace(S D) (the reference's `tr` after the
// reflection fix, D = diag(1,1,sign det)), and the solution is a proper rotation by construction.  The 4x4 problem
// is solved by cyclic Jacobi in fp64 per thread: a few hundred flops per pose, no library, no host round trip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace capf {

static constexpr int MAXJ = 32;

// largest eigenpair of the symmetric 4x4 matrix A (destroyed); q = unit eigenvector
__device__ inline double top_eigen4(double A[4][4], double q[4]) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 4; ++p)
            for (int r = p + 1; r < 4; ++r) off += A[p][r] * A[p][r];
        if (off < 1e-30) break;
        for (int p = 0; p < 3; ++p)
            for (int r = p + 1; r < 4; ++r) {
                const double apr = A[p][r];
                if (fabs(apr) < 1e-300) continue;
                const double theta = (A[r][r] - A[p][p]) / (2.0 * apr);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {          // A <- A J (columns p, r)
                    const double akp = A[k][p], akr = A[k][r];
                    A[k][p] = c * akp - s * akr;
                    A[k][r] = s * akp + c * akr;
                }
                for (int k = 0; k < 4; ++k) {          // A <- J^T A (rows p, r)
                    const double apk = A[p][k], ark = A[r][k];
                    A[p][k] = c * apk - s * ark;
                    A[r][k] = s * apk + c * ark;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkr = V[k][r];
                    V[k][p] = c * vkp - s * vkr;
                    V[k][r] = s * vkp + c * vkr;
                }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (A[i][i] > A[best][best]) best = i;
    for (int k = 0; k < 4; ++k) q[k] = V[k][best];
    return A[best][best];
}

// err[i] = {MPJPE, P_MPJPE, N_MPJPE, velocity error vs pose prev[i] (0 when prev[i] < 0)} of pose i, each the mean
// over the J joints.  pred / gt: [n, J, 3] fp32.
__global__ void pose_errors_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int n, int J,
                                   const int* __restrict__ prev, float* __restrict__ err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* Y = pred + (size_t)i * J * 3;      // keypoints_pred
    const float* X = gt + (size_t)i * J * 3;        // keypoints_gt
    double muX[3] = {0, 0, 0}, muY[3] = {0, 0, 0};
    double e1 = 0.0, syy = 0.0, sxy = 0.0;
    for (int j = 0; j < J; ++j) {
        double d2 = 0.0;
        for (int c = 0; c < 3; ++c) {
            const double x = X[j * 3 + c], y = Y[j * 3 + c];
            muX[c] += x; muY[c] += y;
            d2 += (y - x) * (y - x);
            syy += y * y; sxy += x * y;
        }
        e1 += sqrt(d2);
    }
    e1 /= J;
    // N_MPJPE (loss.py:80-83): scale = mean_j(sum_c gt*pred) / mean_j(sum_c pred^2)
    const double sc = (sxy / J) / (syy / J);
    double e3 = 0.0;
    for (int j = 0; j < J; ++j) {
        double d2 = 0.0;
        for (int c = 0; c < 3; ++c) {
            const double d = sc * Y[j * 3 + c] - X[j * 3 + c];
            d2 += d * d;
        }
        e3 += sqrt(d2);
    }
    e3 /= J;
    // P_MPJPE (loss.py:36-68)
    for (int c = 0; c < 3; ++c) { muX[c] /= J; muY[c] /= J; }
    double nX = 0.0, nY = 0.0, S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};     // S[a][b] = sum_j Y0[j][a] * X0[j][b]
    for (int j = 0; j < J; ++j) {
        double x0[3], y0[3];
        for (int c = 0; c < 3; ++c) { x0[c] = X[j * 3 + c] - muX[c]; y0[c] = Y[j * 3 + c] - muY[c]; nX += x0[c] * x0[c]; nY += y0[c] * y0[c]; }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) S[a][b] += y0[a] * x0[b];
    }
    nX = sqrt(nX); nY = sqrt(nY);
    const double inv = 1.0 / (nX * nY);          // X0 /= normX, Y0 /= normY (:45-46)
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) S[a][b] *= inv;
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double q[4];
    const double tr = top_eigen4(N, q);          // = sum of singular values with the reflection fix (:56-60)
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double R[3][3] = {{w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)},
                            {2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)},
                            {2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z}};   // x ~ R y
    const double a = tr * nX / nY;               // scale (:62)
    double e2 = 0.0;
    for (int j = 0; j < J; ++j) {
        double y0[3], d2 = 0.0;
        for (int c = 0; c < 3; ++c) y0[c] = Y[j * 3 + c] - muY[c];
        for (int r = 0; r < 3; ++r) {
            const double al = a * (R[r][0] * y0[0] + R[r][1] * y0[1] + R[r][2] * y0[2]) + muX[r];   // a * pred R + t (:63-66)
            const double d = al - X[j * 3 + r];
            d2 += d * d;
        }
        e2 += sqrt(d2);
    }
    e2 /= J;
    // velocity error against the previous pose of the same evaluated subset (np.diff over the masked rows, loss.py:98-99)
    double e4 = 0.0;
    const int pi = prev ? prev[i] : i - 1;
    if (pi >= 0) {
        const float* Yp = pred + (size_t)pi * J * 3;
        const float* Xp = gt + (size_t)pi * J * 3;
        for (int j = 0; j < J; ++j) {
            double d2 = 0.0;
            for (int c = 0; c < 3; ++c) {
                // fp32 differences first, as np.diff on the float32 arrays does
                const float vy = Y[j * 3 + c] - Yp[j * 3 + c], vx = X[j * 3 + c] - Xp[j * 3 + c];
                const double d = (double)(vy - vx);
                d2 += d * d;
            }
            e4 += sqrt(d2);
        }
        e4 /= J;
    }
    err[(size_t)i * 4 + 0] = (float)e1;
    err[(size_t)i * 4 + 1] = (float)e2;
    err[(size_t)i * 4 + 2] = (float)e3;
    err[(size_t)i * 4 + 3] = (float)e4;
}

hipError_t launch_pose_errors(const float* pred, const float* gt, int n, int J, const int* prev, float* err, hipStream_t s) {
    if (J <= 0 || J > MAXJ) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pose_errors_kernel, dim3((n + 127) / 128), dim3(128), 0, s, pred, gt, n, J, prev, err);
    return hipGetLastError();
}

// One block per segment (action): sums[seg] = {sum MPJPE, sum P_MPJPE, sum N_MPJPE, sum velocity error} over the poses
// with seg[i] == segment, counts[seg] = {poses, poses that have a predecessor}.  fp64, fixed reduction order
// (deterministic; no atomics).  seg == nullptr: everything is segment 0.
__global__ void segment_sums_kernel(const float* __restrict__ err, const int* __restrict__ seg, const int* __restrict__ prev,
                                    int n, double* __restrict__ sums, int* __restrict__ counts) {
    __shared__ double red[256][4];
    __shared__ int cnt[256][2];
    const int sid = blockIdx.x, t = threadIdx.x;
    double a[4] = {0, 0, 0, 0};
    int c0 = 0, c1 = 0;
    for (int i = t; i < n; i += 256) {
        if (seg && seg[i] != sid) continue;
        ++c0;
        for (int k = 0; k < 3; ++k) a[k] += (double)err[(size_t)i * 4 + k];
        if ((prev ? prev[i] : i - 1) >= 0) { ++c1; a[3] += (double)err[(size_t)i * 4 + 3]; }
    }
    for (int k = 0; k < 4; ++k) red[t][k] = a[k];
    cnt[t][0] = c0; cnt[t][1] = c1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            for (int k = 0; k < 4; ++k) red[t][k] += red[t + o][k];
            cnt[t][0] += cnt[t + o][0]; cnt[t][1] += cnt[t + o][1];
        }
        __syncthreads();
    }
    if (t == 0) {
        for (int k = 0; k < 4; ++k) sums[(size_t)sid * 4 + k] = red[0][k];
        counts[sid * 2 + 0] = cnt[0][0]; counts[sid * 2 + 1] = cnt[0][1];
    }
}

hipError_t launch_segment_sums(const float* err, const int* seg, const int* prev, int n, int n_seg, double* sums, int* counts,
                               hipStream_t s) {
    hipLaunchKernelGGL(segment_sums_kernel, dim3(n_seg), dim3(256), 0, s, err, seg, prev, n, sums, counts);
    return hipGetLastError();
}

// KeypointsMSELoss / KeypointsMSESmoothLoss / KeypointsMAELoss (loss.py:104-137):
//   loss = sum(f(gt - pred) * validity) / (D * max(1, sum(validity)));  validity is [rows, 1] broadcast over D
//   mode 0: f = d^2;  1: d^2, and terms above `thr` replaced by term^0.1 * thr^0.9;  2: |d|
// One block; also writes dloss/dpred when dpred != nullptr.
__global__ void keypoints_loss_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ val,
                                      int rows, int D, int mode, float thr, float* __restrict__ loss, float* __restrict__ dpred) {
    __shared__ double red[1024];
    __shared__ double denom_s;
    const int t = threadIdx.x;
    double v = 0.0;
    for (int r = t; r < rows; r += blockDim.x) v += (double)val[r];
    red[t] = v;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) denom_s = (double)D * fmax(1.0, red[0]);
    __syncthreads();
    const double denom = denom_s;
    const float thr09 = powf(thr, 0.9f);
    double acc = 0.0;
    for (int i = t; i < rows * D; i += blockDim.x) {
        const float w = val[i / D];
        const float d = gt[i] - pred[i];
        float term, g;                      // g = d term / d pred
        if (mode == 2) {
            term = fabsf(d) * w;
            g = (d > 0.f ? -1.f : (d < 0.f ? 1.f : 0.f)) * w;
        } else {
            term = d * d * w;
            g = -2.f * d * w;
            if (mode == 1 && term > thr) {
                g *= 0.1f * powf(term, -0.9f) * thr09;
                term = powf(term, 0.1f) * thr09;
            }
        }
        acc += (double)term;
        if (dpred) dpred[i] = (float)((double)g / denom);
    }
    __syncthreads();
    red[t] = acc;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) loss[0] = (float)(red[0] / denom);
}

hipError_t launch_keypoints_loss(const float* pred, const float* gt, const float* validity, int rows, int D, int mode, float thr,
                                 float* loss, float* dpred, hipStream_t s) {
    hipLaunchKernelGGL(keypoints_loss_kernel, dim3(1), dim3(1024), 0, s, pred, gt, validity, rows, D, mode, thr, loss, dpred);
    return hipGetLastError();
}

}  // namespace capf
