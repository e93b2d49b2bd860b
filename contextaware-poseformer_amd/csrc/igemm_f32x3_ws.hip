// fp32 3x3 / stride-1 / pad-1 convolution on the bf16 matrix pipe, operands split into three bf16 pieces (igemm_f32x3_ws_tile.h):
// launchers, grouped kernel, weight pack.  Takes, for launches that fill the chip, the BasicBlock convs of HRNet under
// compute_dtype = fp32 (pose_hrnet.py:66-95) from the Winograd kernels of igemm_wino.hip -- same fp32 tensors in and out, results
// to fp32 accumulation order (tools/f32x3_ws.hip: 2.6e-7 of the sum of |terms| at worst against an fp64 evaluation, rms 1.4e-8;
// the F(4,3) kernel this replaces: 1e-5).
// Measured alone (tools/f32x3_ws.hip; F(4,3) kernel in brackets), batch 64: 32 ch 64^2 37.6 us (43.6), 64 ch 32^2 31.6 (38.0),
// 128 ch 16^2 30.6 (48.5), 256 ch 8^2 50.8 (81.3); batch 512: 352 (341), 271 (281), 236 (243), 229 (248).
#include "igemm_f32x3_ws_tile.h"
#include "kernels.h"

namespace capf {

static constexpr int X3_NS = 32;               // output channels per tile: two blocks per CU (67.6 KiB of LDS each)

long f32x3_pack_elems(int Cout, int Cin) { return x3_pack_elems(Cout, Cin, X3_NS); }

// GemmArgs -> tile geometry; false = not a problem this tile takes
static bool x3_from_args(const GemmArgs& a, X3Problem* q) {
    if (!a.conv || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W || a.act == ACT_GELU || a.rscale || a.out_bf16 ||
        a.omap.G != 1 || (a.res && a.rmap.G != 1) || a.M <= 0 || a.H <= 0 || a.W <= 0 || a.M % (a.H * a.W) != 0)
        return false;
    if ((a.omap.S1 & 3) || (a.omap.off & 3) || (a.res && ((a.rmap.S1 & 3) || (a.rmap.off & 3)))) return false;     // 16-byte pieces
    if ((double)a.M * (double)a.omap.S1 * 4.0 >= 2.0e9 || (a.res && (double)a.M * (double)a.rmap.S1 * 4.0 >= 2.0e9)) return false;
    if (!x3_plan(a.M / (a.H * a.W), a.H, a.W, a.Cin, a.N, X3_NS, q)) return false;
    q->x = a.A;
    q->g.wp = reinterpret_cast<const unsigned short*>(a.Wp3);
    q->g.bias = a.bias;
    q->res = a.res ? a.res + a.rmap.off : nullptr;
    q->y = a.out + a.omap.off;
    q->g.ldy = (int)a.omap.S1;
    q->g.ldr = a.res ? (int)a.rmap.S1 : (int)a.omap.S1;
    q->g.relu = a.act == ACT_RELU;
    return true;
}

bool gemm_f32x3_ok(const GemmArgs& a) {
    if (a.x3_h2) return gemm_f32h2_ok(a);                  // (the two-piece tile addresses per tile: no 2 GB tensor limit)
    X3Problem q;
    return x3_from_args(a, &q);
}

// As for the bf16 tile (igemm_bf16_ws.hip): which kernel a conv runs on is a function of the conv ALONE, so that every schedule of
// the engine produces the same bits.  From 370 MFLOP and batch 5 up (the HRNet-32 branch convs -- 75.5 MFLOP per frame -- from batch 5:
// round 5's sweep with the two-piece tile, ms per forward at batch 4 / 5 / 6 / 7 / 8: 2.89 (direct kernels) / 2.79 / 2.85 / 2.96 / 3.01; the rule
// of rounds 4-5, 400 MFLOP and batch 6, left batch 5 on the direct kernels at 3.62.  Round 4's numbers for the three-piece tile, ms per forward
// against the direct kernel with split-K / the Winograd kernels: batch 6 3.67 / 3.80 / -, 8 3.76 / 3.90 / 5.47, 16 4.27 / 4.75 / 5.86,
// 24 4.90 / 6.08 / 6.37; below, the direct kernel wins: batch 4 3.49 / 2.92).  (diag builds: CAPF_F32X3_MIN_MFLOP)
static bool x3_big_enough(int B, int H, int W, int Cin, int Cout) {
    static const double min_flop = [] { const char* e = diag_env("CAPF_F32X3_MIN_MFLOP"); return (e ? atof(e) : 370.0) * 1e6; }();
    return B >= 5 && 2.0 * (double)B * H * W * Cout * 9.0 * Cin >= min_flop;
}

bool f32x3_takes(int B, int H, int W, int Cin, int Cout, bool h2) {
    if (!x3_big_enough(B, H, W, Cin, Cout)) return false;
    if (h2) return f32h2_shape_ok(B, H, W, Cin, Cout);
    X3Problem q;
    return x3_plan(B, H, W, Cin, Cout, X3_NS, &q);
}

bool gemm_f32x3_wanted(const GemmArgs& a) {            // (one geometry computation: this runs on the launch path)
    return a.Wp3 && a.H > 0 && a.W > 0 && a.M % (a.H * a.W) == 0 && x3_big_enough(a.M / (a.H * a.W), a.H, a.W, a.Cin, a.N) && gemm_f32x3_ok(a);
}

struct X3GroupArgs {
    X3Problem g[MAXG];
    int start[MAXG + 1];
    int tiles[MAXG];
    int n;
};

__global__ __launch_bounds__(256, 2) void igemm_f32x3_group_ws_kernel(X3GroupArgs ga) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char x3_lds[];
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < ga.n && b >= ga.start[pi + 1]) ++pi;
    const int l = b - ga.start[pi];
    const int per_xcd = (ga.start[pi + 1] - ga.start[pi]) >> 3;
    const int bid = (l & 7) * per_xcd + (l >> 3);          // block b of a problem runs on XCD b % 8: that XCD's contiguous eighth of the tiles
    if (bid >= ga.tiles[pi]) return;
    igemm_f32x3_ws_tile<X3_NS / 32>(ga.g[pi], bid, x3_lds);
#endif
}

hipError_t launch_gemm_f32x3_group(const GemmArgs* list, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > MAXG) return hipErrorInvalidValue;
    if (list[0].x3_h2) return launch_gemm_f32h2_group(list, n, s);      // (the two-fp16-piece tile: one plan, one kind of pack per engine)
    struct Item { X3Problem q; int cost; };
    Item it[MAXG];
    for (int i = 0; i < n; ++i) {
        if (!list[i].Wp3 || !x3_from_args(list[i], &it[i].q)) return hipErrorInvalidValue;
        it[i].cost = it[i].q.g.C;                           // a tile's K loop: longest first, so that the launch does not end on them
    }
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && it[j].cost > it[j - 1].cost; --j) { Item t = it[j]; it[j] = it[j - 1]; it[j - 1] = t; }
    X3GroupArgs ga;
    ga.n = n;
    int start = 0;
    for (int i = 0; i < n; ++i) {
        ga.g[i] = it[i].q;
        ga.tiles[i] = it[i].q.g.tiles_m * it[i].q.g.NSL;
        ga.start[i] = start;
        start += (ga.tiles[i] + 7) & ~7;
    }
    ga.start[n] = start;
    for (int i = n; i < MAXG; ++i) { ga.start[i + 1] = start; ga.tiles[i] = 0; ga.g[i] = ga.g[0]; }
    static DynLdsAttr attr_once;
    const hipError_t attr = attr_once.ensure(reinterpret_cast<const void*>(&igemm_f32x3_group_ws_kernel), x3_lds_bytes(X3_NS));
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(igemm_f32x3_group_ws_kernel, dim3(start), dim3(256), x3_lds_bytes(X3_NS), s, ga);
    return hipGetLastError();
}

hipError_t launch_gemm_f32x3(const GemmArgs& a, hipStream_t s) { return launch_gemm_f32x3_group(&a, 1, s); }

const char* gemm_f32x3_kernel_name(const GemmArgs& a) { return a.x3_h2 ? gemm_f32h2_kernel_name(a) : "igemm_f32x3_group_ws"; }

// BN fold + three-way split + re-layout for the tile: with v = w[slice * NS + n][cc * 16 + 8 h + e][kh][kw] * gamma / sqrt(var + eps) (the
// fp32 value launch_pack_conv folds), piece 0 = bf16(v), piece 1 = bf16(v - piece 0), piece 2 = bf16(v - piece 0 - piece 1) -- exact
// remainders, v = piece 0 + piece 1 + piece 2 -- at Wp[slice][Cin / 16][piece][tap][n][quad position][8], h = quad position ^ ((n >> 3) & 1)
// (the LDS image's bank swizzle, so that the DMA is a linear copy); rows beyond Cout zero; bias as launch_pack_conv
__global__ void pack_conv_f32x3_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                       unsigned short* __restrict__ Wp, float* __restrict__ bias, int Cout, int Cin, int NS, long total) {
    const int ncc = Cin / 16;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {   // i: one weight, its three pieces
        long k = i;
        const int e = (int)(k & 7); k >>= 3;
        const int qp = (int)(k & 1); k >>= 1;
        const int n = (int)(k % NS); k /= NS;
        const int tap = (int)(k % 9); k /= 9;
        const int cc = (int)(k % ncc);
        const int sl = (int)(k / ncc);
        const int ng = sl * NS + n, c = cc * 16 + (qp ^ ((n >> 3) & 1)) * 8 + e;
        float v = 0.f;
        if (ng < Cout) {
            const float sc = gamma ? gamma[ng] / sqrtf(var[ng] + eps) : 1.f;
            v = w[(((long)ng * Cin + c) * 3 + tap / 3) * 3 + tap % 3] * sc;
            if (bias && cc == 0 && tap == 0 && qp == 0 && e == 0) bias[ng] = gamma ? beta[ng] - mean[ng] * sc : 0.f;
        }
        const long piece = 9L * NS * 16;
        const long base = ((long)(sl * ncc + cc) * 3) * piece + ((long)tap * NS + n) * 16 + qp * 8 + e;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            const unsigned short b = to_bf16(v);
            Wp[base + pc * piece] = b;
            v -= __uint_as_float((unsigned)b << 16);
        }
    }
}

hipError_t launch_pack_conv_f32x3(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                  void* Wp_bf16, float* bias, int Cout, int Cin, hipStream_t s) {
    if (Cin % 16 != 0 || Cout <= 0) return hipErrorInvalidValue;
    const long total = f32x3_pack_elems(Cout, Cin) / 3;
    const long want = (total + 255) / 256;
    hipLaunchKernelGGL(pack_conv_f32x3_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, s, w, gamma, beta, mean, var, eps,
                       static_cast<unsigned short*>(Wp_bf16), bias, Cout, Cin, X3_NS, total);
    return hipGetLastError();
}

}  // namespace capf
