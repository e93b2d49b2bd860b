// The step just before the hot path and the step just after it, each as one
// launch instead of the reference's dozen elementwise ops:
//   preprocess_*  — data_prefetcher.preload, ContextPose/mvn/datasets/utils.py:33-82: uint8 BGR crop ->
//                   normalised fp32 RGB NHWC (channel flip :45, /255 - mean (/ std) :47-50), root-relative
//                   ground truth (:52-53), optional train-time horizontal flip with left/right joint swap
//                   (:55-65) or flip-test stacking of the original and the mirrored sample (:67-80);
//   fliptest_fuse — ContextPose/train.py:177-180: un-mirror the second prediction and average.
// Same fp32 operation order as the reference as it runs on a GPU ((u * (1/255) - mean) / std; 192 - x - 1), so
// results are bit-identical to the torch expressions.  Built with -ffp-contract=off.
#include "kernels.h"

namespace capf {

// H36M skeleton: joints_left / joints_right of mvn/datasets/utils.py:12-13 as a swap table
__device__ __constant__ int kSwap[17] = {0, 4, 5, 6, 1, 2, 3, 7, 8, 9, 10, 14, 15, 16, 11, 12, 13};

// out[s, b, h, w, c] for s in {0 (as is), 1 (mirrored along W)}; `mirror_first` mirrors sample 0 (train flip)
__global__ void preprocess_images_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                         float m0, float m1, float m2, float s0, float s1, float s2, int use_std,
                                         int nsets, int mirror_first) {
    const long npix = (long)B * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix * nsets; i += (long)gridDim.x * blockDim.x) {
        const int set = (int)(i / npix);
        const long pix = i - (long)set * npix;
        const int w = (int)(pix % W);
        const long row = pix / W;
        const bool mirror = set == 1 || (set == 0 && mirror_first);
        const long src = (row * W + (mirror ? W - 1 - w : w)) * 3;
        const float b = (float)in[src + 0], g = (float)in[src + 1], r = (float)in[src + 2];   // BGR in memory
        // `images / 255.0` with a Python scalar is a multiplication by the fp32 reciprocal on the GPU the
        // reference's prefetcher runs on (ATen div_true_kernel_cuda, CPU-scalar fast path); mean / std are tensors
        const float inv255 = 1.0f / 255.0f;
        float o0 = __fsub_rn(__fmul_rn(r, inv255), m0), o1 = __fsub_rn(__fmul_rn(g, inv255), m1),
              o2 = __fsub_rn(__fmul_rn(b, inv255), m2);
        if (use_std) { o0 = __fdiv_rn(o0, s0); o1 = __fdiv_rn(o1, s1); o2 = __fdiv_rn(o2, s2); }
        float* o = out + i * 3;
        o[0] = o0; o[1] = o1; o[2] = o2;
    }
}

// keypoints / ground truth: one thread per (set, b, joint)
__global__ void preprocess_points_kernel(const float* __restrict__ gt_in, float* __restrict__ gt_out,
                                         const float* __restrict__ k2d_in, float* __restrict__ k2d_out,
                                         const float* __restrict__ kc_in, float* __restrict__ kc_out, int B, int nsets,
                                         int mirror_first) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsets * B * 17) return;
    const int j = t % 17, b = (t / 17) % B, set = t / (17 * B);
    const bool mirror = set == 1 || (set == 0 && mirror_first);
    const int js = mirror ? kSwap[j] : j;            // value written at joint j comes from joint js
    {
        const float* s = k2d_in + ((long)b * 17 + js) * 2;
        float* d = k2d_out + (((long)set * B + b) * 17 + j) * 2;
        d[0] = mirror ? -s[0] : s[0];
        d[1] = s[1];
    }
    {
        const float* s = kc_in + ((long)b * 17 + js) * 2;
        float* d = kc_out + (((long)set * B + b) * 17 + j) * 2;
        d[0] = mirror ? __fsub_rn(__fsub_rn(192.0f, s[0]), 1.0f) : s[0];
        d[1] = s[1];
    }
    if (set == 0 && gt_in) {     // root-relative ground truth (only the train flip mirrors it)
        const float* root = gt_in + (long)b * 17 * 3;
        const float* s = gt_in + ((long)b * 17 + js) * 3;
        float* d = gt_out + ((long)b * 17 + j) * 3;
        float x = js == 0 ? 0.f : __fsub_rn(s[0], root[0]);
        const float y = js == 0 ? 0.f : __fsub_rn(s[1], root[1]);
        const float z = js == 0 ? 0.f : __fsub_rn(s[2], root[2]);
        if (mirror_first) x = -x;
        d[0] = x; d[1] = y; d[2] = z;
    }
}

hipError_t launch_preprocess(const unsigned char* images_bgr, int B, int H, int W, const float mean[3], const float* stdv,
                             int mode, float* images_out, const float* gt_in, float* gt_out, const float* k2d_in,
                             float* k2d_out, const float* kc_in, float* kc_out, hipStream_t s) {
    const int nsets = mode == 2 ? 2 : 1, mirror_first = mode == 1 ? 1 : 0;
    const long n = (long)B * H * W * nsets;
    const long want = (n + 255) / 256;
    hipLaunchKernelGGL(preprocess_images_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, s, images_bgr,
                       images_out, B, H, W, mean[0], mean[1], mean[2], stdv ? stdv[0] : 1.f, stdv ? stdv[1] : 1.f,
                       stdv ? stdv[2] : 1.f, stdv ? 1 : 0, nsets, mirror_first);
    const int np = nsets * B * 17;
    hipLaunchKernelGGL(preprocess_points_kernel, dim3((np + 127) / 128), dim3(128), 0, s, gt_in, gt_out, k2d_in, k2d_out, kc_in,
                       kc_out, B, nsets, mirror_first);
    return hipGetLastError();
}

// out[b, j, :] = 0.5 * (pred[0, b, j, :] + unmirror(pred[1, b, :, :])[j])     (train.py:177-180)
__global__ void fliptest_fuse_kernel(const float* __restrict__ pred2, float* __restrict__ out, int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 17) return;
    const int j = t % 17, b = t / 17;
    const float* p = pred2 + ((long)b * 17 + j) * 3;
    const float* q = pred2 + (((long)B + b) * 17 + kSwap[j]) * 3;
    float* o = out + (long)t * 3;
    // torch.mean over a dim of size 2: (a + b) / 2
    o[0] = __fdiv_rn(__fadd_rn(p[0], -q[0]), 2.0f);
    o[1] = __fdiv_rn(__fadd_rn(p[1], q[1]), 2.0f);
    o[2] = __fdiv_rn(__fadd_rn(p[2], q[2]), 2.0f);
}

hipError_t launch_fliptest_fuse(const float* pred2, int B, float* out, hipStream_t s) {
    hipLaunchKernelGGL(fliptest_fuse_kernel, dim3((B * 17 + 127) / 128), dim3(128), 0, s, pred2, out, B);
    return hipGetLastError();
}

// ---- N3: the per-frame affine crop (mvn/utils/img.py:16-69, human36m.py:281-302) ------------------------
// get_affine_transform with rot = 0, shift = 0: three float32 point pairs (centre, centre + (0, -(src_w-1)/2),
// and the 90-degree companion of img.py:11-13) -> the 2x3 double matrix cv2.getAffineTransform solves for.
// Host code; compiled without FMA contraction so that it rounds like the numpy expressions it restates.
static bool solve3(double a[3][4]) {      // Gaussian elimination with partial pivoting, in place; solution in a[i][3]
    for (int c = 0; c < 3; ++c) {
        int piv = c;
        for (int r = c + 1; r < 3; ++r)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        for (int k = 0; k < 4; ++k) { const double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
        for (int r = c + 1; r < 3; ++r) {
            const double f = a[r][c] / a[c][c];
            for (int k = c; k < 4; ++k) a[r][k] -= f * a[c][k];
        }
    }
    for (int r = 2; r >= 0; --r) {
        double v = a[r][3];
        for (int k = r + 1; k < 3; ++k) v -= a[r][k] * a[k][3];
        a[r][3] = v / a[r][r];
    }
    return true;
}

bool affine_from_center_scale(const double center[2], const double scale[2], int out_w, int out_h, double M[6]) {
    const double src_w = scale[0] * 200.0;
    const float src_dir1 = (float)((src_w - 1) * -0.5), dst_dir1 = (float)((out_w - 1) * -0.5);
    float src[3][2], dst[3][2];
    src[0][0] = (float)center[0];
    src[0][1] = (float)center[1];
    src[1][0] = (float)(center[0] + 0.0);
    src[1][1] = (float)(center[1] + (double)src_dir1);
    const double dcx = (out_w - 1) * 0.5, dcy = (out_h - 1) * 0.5;
    dst[0][0] = (float)dcx;
    dst[0][1] = (float)dcy;
    dst[1][0] = (float)(dcx + 0.0);
    dst[1][1] = (float)(dcy + (double)dst_dir1);
    for (int w = 0; w < 2; ++w) {         // third point: b + (-(a-b).y, (a-b).x) in float32
        float (*p)[2] = w == 0 ? src : dst;
        const float dx = p[0][0] - p[1][0], dy = p[0][1] - p[1][1];
        p[2][0] = p[1][0] + (-dy);
        p[2][1] = p[1][1] + dx;
    }
    for (int row = 0; row < 2; ++row) {
        double a[3][4];
        for (int i = 0; i < 3; ++i) { a[i][0] = src[i][0]; a[i][1] = src[i][1]; a[i][2] = 1.0; a[i][3] = dst[i][row]; }
        if (!solve3(a)) return false;
        for (int k = 0; k < 3; ++k) M[row * 3 + k] = a[k][3];
    }
    return true;
}

// cv2.warpAffine(frame, M, (out_w, out_h), INTER_LINEAR, BORDER_CONSTANT 0) for 8-bit 3-channel frames: OpenCV's
// fixed-point pipeline (10-bit coordinates, 5-bit fractions, 15-bit weights) restated per output pixel.
// frames[b]: device pointer of frame b; dims[b] = {rows, cols, row pitch in bytes}; M[b]: forward 2x3 matrix.
__global__ void warp_affine_u8_kernel(const unsigned char* const* __restrict__ frames, const int* __restrict__ dims,
                                      const double* __restrict__ M, unsigned char* __restrict__ out, int B, int out_h, int out_w) {
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long total = (long)B * out_h * out_w;
    if (t >= total) return;
    const int x = (int)(t % out_w);
    const int y = (int)((t / out_w) % out_h);
    const int b = (int)(t / ((long)out_w * out_h));
    const double* m = M + (long)b * 6;
    double d = m[0] * m[4] - m[1] * m[3];
    d = d != 0.0 ? 1.0 / d : 0.0;
    const double a00 = m[4] * d, a11 = m[0] * d, a01 = m[1] * -d, a10 = m[3] * -d;
    const double b0 = -a00 * m[2] - a01 * m[5];
    const double b1 = -a10 * m[2] - a11 * m[5];
    const long adelta = __double2ll_rn(a00 * (double)x * 1024.0);
    const long bdelta = __double2ll_rn(a10 * (double)x * 1024.0);
    const long X0 = __double2ll_rn((a01 * (double)y + b0) * 1024.0) + 16;
    const long Y0 = __double2ll_rn((a11 * (double)y + b1) * 1024.0) + 16;
    const long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    long sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const int ax = (int)(X & 31), ay = (int)(Y & 31);
    const int H = dims[b * 3 + 0], W = dims[b * 3 + 1], pitch = dims[b * 3 + 2];
    const unsigned char* src = frames[b];
    const int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
    const bool y0ok = sy >= 0 && sy < H, y1ok = sy + 1 >= 0 && sy + 1 < H;
    const bool x0ok = sx >= 0 && sx < W, x1ok = sx + 1 >= 0 && sx + 1 < W;
    unsigned char* o = out + t * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int acc = 16384;
        if (y0ok && x0ok) acc += w00 * src[(long)sy * pitch + sx * 3 + c];
        if (y0ok && x1ok) acc += w01 * src[(long)sy * pitch + (sx + 1) * 3 + c];
        if (y1ok && x0ok) acc += w10 * src[(long)(sy + 1) * pitch + sx * 3 + c];
        if (y1ok && x1ok) acc += w11 * src[(long)(sy + 1) * pitch + (sx + 1) * 3 + c];
        o[c] = (unsigned char)(acc >> 15);
    }
}

hipError_t launch_warp_affine_u8(const unsigned char* const* frames, const int* dims, const double* M, unsigned char* out,
                                 int B, int out_h, int out_w, hipStream_t s) {
    const long total = (long)B * out_h * out_w;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(warp_affine_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, dims, M, out, B,
                       out_h, out_w);
    return hipGetLastError();
}

}  // namespace capf
