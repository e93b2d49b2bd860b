// fp32 3x3 / stride-1 / pad-1 convolution on the 16-bit matrix pipe, operands as two block-scaled fp16 pieces and three piece products
// per fp32 MAC (igemm_f32h2_ws_tile.h): launchers, grouped kernels, weight pack.  The default route of the BasicBlock convs of HRNet
// under compute_dtype = fp32 (pose_hrnet.py:66-95) wherever the three-bf16-piece tile of igemm_f32x3_ws.hip was eligible -- same fp32
// tensors in and out, half the MFMAs.  Which convs: gemm_f32x3_wanted() (one rule for both tiles); which of the two tiles: GemmArgs::x3_h2
// (the engine's plan: CAPF_PLAN_F32X3_EXACT keeps the three-piece tile).
// Measured alone on one box (tools/f32h2_ws.hip; three-piece tile in brackets), batch 64: 32 ch 64^2 30.1 us (35.1), 64 ch 32^2 22.1
// (30.0), 128 ch 16^2 21.9 (30.7), 256 ch 8^2 33.8 (50.7); batch 512: 257 (321), 160 (264), 125 (223), 109 (215) with 64-channel tiles
// on the three wide branches.
#include "igemm_f32h2_ws_tile.h"
#include "kernels.h"

namespace capf {

long f32h2_pack_elems(int Cout, int Cin) { return h2_pack_elems(Cout, Cin); }

// GemmArgs -> tile geometry; false = not a problem this tile takes (the same conditions as the three-piece tile's x3_from_args).
// Tile width: 64 output channels (half the split work and pixel fragment reads per MFMA, two blocks per CU) once a conv still fills
// the chip with them -- at least 512 such tiles; 32 otherwise (three blocks per CU).  Either width computes the same bits: the K order
// and the block scales do not depend on it.
static bool h2_from_args(const GemmArgs& a, H2Problem* q) {
    if (!a.conv || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W || a.act == ACT_GELU || a.rscale || a.out_bf16 ||
        a.omap.G != 1 || (a.res && a.rmap.G != 1) || a.M <= 0 || a.H <= 0 || a.W <= 0 || a.M % (a.H * a.W) != 0)
        return false;
    if ((a.omap.S1 & 3) || (a.omap.off & 3) || (a.res && ((a.rmap.S1 & 3) || (a.rmap.off & 3)))) return false;     // 16-byte pieces
    if ((double)WS_MAX_P * (double)a.omap.S1 * 4.0 >= 2.0e9 || (a.res && (double)WS_MAX_P * (double)a.rmap.S1 * 4.0 >= 2.0e9)) return false;   // (one tile's rows)
    if (!h2_plan(a.M / (a.H * a.W), a.H, a.W, a.Cin, a.N, 32, q)) return false;
    static const long wide_min = [] { const char* e = diag_env("CAPF_H2_WIDE_MIN_TILES"); return e ? atol(e) : 512L; }();      // (diag builds: A/B runs)
    if (a.N % 64 == 0 && (long)q->g.tiles_m * (a.N / 64) >= wide_min) { q->g.NS = 64; q->g.NSL = a.N / 64; }
    q->x = a.A;
    q->g.wp = reinterpret_cast<const unsigned short*>(a.Wp3);
    q->winv = reinterpret_cast<const float*>(q->g.wp + h2_piece_elems(a.N, a.Cin));
    q->g.bias = a.bias;
    q->res = a.res ? a.res + a.rmap.off : nullptr;
    q->y = a.out + a.omap.off;
    q->g.ldy = (int)a.omap.S1;
    q->g.ldr = a.res ? (int)a.rmap.S1 : (int)a.omap.S1;
    q->g.relu = a.act == ACT_RELU;
    // planes (igemm_f32h2_ws_tile.h): the producer's and the consumer's tiles must be the same pixels -- the plan only pairs a BasicBlock's
    // conv1 / conv2, equal shapes by construction -- chunks of 16 channels must be whole, a producer adds no residual
    q->ein = a.h2_ein;
    q->eout = a.h2_eout;
    q->utab = a.h2_utab;
    if (a.h2_eout && (a.res || a.N % 16 != 0 || a.omap.S1 != a.N)) return false;
    if (a.h2_ein && a.Cin % 16 != 0) return false;
    return true;
}

bool gemm_f32h2_ok(const GemmArgs& a) {
    H2Problem q;
    return h2_from_args(a, &q);
}
int f32h2_tiles_m(int B, int H, int W, int* tile_pixels) {
    H2Problem q;
    if (!h2_plan(B, H, W, 16, 16, 32, &q)) return 0;
    if (tile_pixels) *tile_pixels = q.g.P;
    return q.g.tiles_m;
}
bool f32h2_unit_table(int H, int W, int Cin, unsigned* out) { return h2_unit_table(H, W, Cin, out); }
bool f32h2_shape_ok(int B, int H, int W, int Cin, int Cout) {
    H2Problem q;
    return h2_plan(B, H, W, Cin, Cout, 32, &q);
}

struct H2GroupArgs {
    H2Problem g[MAXG];
    int start[MAXG + 1];
    int tiles[MAXG];
    int n;
};

template <int TN, bool PIN = false, bool POUT = false>
__global__ __launch_bounds__(256, TN == 1 ? 3 : 2) void igemm_f32h2_group_ws_kernel(H2GroupArgs ga) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char h2_lds[];
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < ga.n && b >= ga.start[pi + 1]) ++pi;
    const int l = b - ga.start[pi];
    const int per_xcd = (ga.start[pi + 1] - ga.start[pi]) >> 3;
    const int bid = (l & 7) * per_xcd + (l >> 3);          // block b of a problem runs on XCD b % 8: that XCD's contiguous eighth of the tiles
    if (bid >= ga.tiles[pi]) return;
    igemm_f32h2_ws_tile<TN, PIN, POUT>(ga.g[pi], bid, h2_lds);
#endif
}

template <int TN, bool PIN = false, bool POUT = false>
static hipError_t h2_launch(const H2Problem* list, int n, hipStream_t s) {
    struct Item { H2Problem q; int cost; };
    Item it[MAXG];
    for (int i = 0; i < n; ++i) { it[i].q = list[i]; it[i].cost = list[i].g.C; }   // a tile's K loop: longest first, so that the launch does not end on them
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && it[j].cost > it[j - 1].cost; --j) { Item t = it[j]; it[j] = it[j - 1]; it[j - 1] = t; }
    H2GroupArgs ga;
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
    static DynLdsAttr attr;
    const hipError_t e = attr.ensure(reinterpret_cast<const void*>(&igemm_f32h2_group_ws_kernel<TN, PIN, POUT>), h2_lds_bytes(32 * TN));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((igemm_f32h2_group_ws_kernel<TN, PIN, POUT>), dim3(start), dim3(256), h2_lds_bytes(32 * TN), s, ga);
    return hipGetLastError();
}

// the problems of one dependency level: those on 32-channel tiles as one grid, those on 64-channel tiles as another (different
// register / LDS budgets: three resident blocks against two)
hipError_t launch_gemm_f32h2_group(const GemmArgs* list, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > MAXG) return hipErrorInvalidValue;
    // one grid per (tile width, planes in, planes out) class present in the level: a level's convs are the same kind of BasicBlock conv
    // almost always, so this is one or two launches as before
    H2Problem cls[2][3][MAXG];
    int cnt[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < n; ++i) {
        H2Problem q;
        if (!list[i].Wp3 || !list[i].x3_h2 || !h2_from_args(list[i], &q)) return hipErrorInvalidValue;
        if (q.ein && q.eout) return hipErrorInvalidValue;
        const int w = q.g.NS == 64 ? 1 : 0, k = q.ein ? 1 : (q.eout ? 2 : 0);
        cls[w][k][cnt[w][k]++] = q;
    }
    hipError_t e = hipSuccess;
    if (cnt[1][0] && e == hipSuccess) e = h2_launch<2>(cls[1][0], cnt[1][0], s);
    if (cnt[1][1] && e == hipSuccess) e = h2_launch<2, true, false>(cls[1][1], cnt[1][1], s);
    if (cnt[1][2] && e == hipSuccess) e = h2_launch<2, false, true>(cls[1][2], cnt[1][2], s);
    if (cnt[0][0] && e == hipSuccess) e = h2_launch<1>(cls[0][0], cnt[0][0], s);
    if (cnt[0][1] && e == hipSuccess) e = h2_launch<1, true, false>(cls[0][1], cnt[0][1], s);
    if (cnt[0][2] && e == hipSuccess) e = h2_launch<1, false, true>(cls[0][2], cnt[0][2], s);
    return e;
}

const char* gemm_f32h2_kernel_name(const GemmArgs&) { return "igemm_f32h2_group_ws"; }

// ---- weight pack: BN fold (the fp32 value v launch_pack_conv folds), one power-of-two scale t per output channel with max |v| t in
// [2^14, 2^15), two fp16 pieces of v t -- piece 0 = fp16(v t), piece 1 = fp16(v t - piece 0), |v t - piece 0 - piece 1| <= 2^-23 |v t| --
// at Wp[slice][Cin / 16][piece][tap][n][quad position][8], h = quad position ^ ((n >> 3) & 1) (the LDS image's bank swizzle: the DMA is a
// linear copy), then the fp32 inverse scales 1 / t [slices * 32]; rows beyond Cout zero with scale 1; bias as launch_pack_conv
__global__ __launch_bounds__(256) void h2_wscale_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                                        float eps, float* __restrict__ winv, int Cout, int Cin) {
    __shared__ float red[256];
    const int ng = blockIdx.x;
    float m = 0.f;
    if (ng < Cout) {
        const float sc = gamma ? gamma[ng] / sqrtf(var[ng] + eps) : 1.f;
        for (int i = threadIdx.x; i < Cin * 9; i += 256) m = fmaxf(m, fabsf(w[(long)ng * Cin * 9 + i] * sc));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) winv[ng] = ng < Cout ? __int_as_float((254 - h2_scale_exp(__float_as_int(red[0]))) << 23) : 1.f;
}

__global__ void pack_conv_f32h2_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                       unsigned short* __restrict__ Wp, const float* __restrict__ winv, float* __restrict__ bias,
                                       int Cout, int Cin, long total) {
    const int ncc = Cin / 16;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {   // i: one weight, its two pieces
        long k = i;
        const int e = (int)(k & 7); k >>= 3;
        const int qp = (int)(k & 1); k >>= 1;
        const int n = (int)(k & 31); k >>= 5;
        const int tap = (int)(k % 9); k /= 9;
        const int cc = (int)(k % ncc);
        const int sl = (int)(k / ncc);
        const int ng = sl * 32 + n, c = cc * 16 + (qp ^ ((n >> 3) & 1)) * 8 + e;
        float v = 0.f;
        if (ng < Cout) {
            const float sc = gamma ? gamma[ng] / sqrtf(var[ng] + eps) : 1.f;
            v = w[(((long)ng * Cin + c) * 3 + tap / 3) * 3 + tap % 3] * sc;
            if (bias && cc == 0 && tap == 0 && qp == 0 && e == 0) bias[ng] = gamma ? beta[ng] - mean[ng] * sc : 0.f;
            v *= __uint_as_float(0x7F000000u - __float_as_uint(winv[ng]));      // the channel's scale: 1 / (a power of two), exact
        }
        const long piece = 9L * 32 * 16;
        const long base = ((long)(sl * ncc + cc) * 2) * piece + ((long)tap * 32 + n) * 16 + qp * 8 + e;
        const _Float16 p0 = (_Float16)v;
        const _Float16 p1 = (_Float16)(v - (float)p0);
        Wp[base] = __builtin_bit_cast(unsigned short, p0);
        Wp[base + piece] = __builtin_bit_cast(unsigned short, p1);
    }
}

hipError_t launch_pack_conv_f32h2(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                  void* Wp_f16, float* bias, int Cout, int Cin, hipStream_t s) {
    if (Cin % 16 != 0 || Cout <= 0) return hipErrorInvalidValue;
    unsigned short* Wp = static_cast<unsigned short*>(Wp_f16);
    float* winv = reinterpret_cast<float*>(Wp + h2_piece_elems(Cout, Cin));
    const int npad = ((Cout + 31) / 32) * 32;
    hipLaunchKernelGGL(h2_wscale_kernel, dim3(npad), dim3(256), 0, s, w, gamma, var, eps, winv, Cout, Cin);
    const long total = h2_piece_elems(Cout, Cin) / 2;
    const long want = (total + 255) / 256;
    hipLaunchKernelGGL(pack_conv_f32h2_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, s, w, gamma, beta, mean, var, eps, Wp,
                       winv, bias, Cout, Cin, total);
    return hipGetLastError();
}

}  // namespace capf
