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

}  // namespace capf
