// bf16 3x3 / stride-1 / pad-1 convolution on the "2-D halo" tile (igemm_bf16_ws_tile.h): launchers, grouped kernel, weight pack.
// Replaces, for launches that fill the chip, the row-halo tile of igemm_bf16.hip for the BasicBlock convs of HRNet
// (pose_hrnet.py:66-95) and the 3x3 convs of the ResNet-50 / refine bottlenecks (networks/resnet.py:58-93, refineNet.py:3-45).
// Measured alone at batch 256 (tools/bf16_ws.hip; row-halo tile in brackets): 96 ch 32^2 979 TFLOP/s (546), 192 ch 16^2 1107 (671),
// 384 ch 8^2 1108 (776), 48 ch 64^2 605 (411).
#include "igemm_bf16_ws_tile.h"
#include "kernels.h"

namespace capf {

static int ws_ns(int N) { return N % 96 == 0 ? 96 : (N <= 32 ? 32 : 64); }

long bf16_ws_pack_elems(int Cout, int Cin) {
    const int NS = ws_ns(Cout);
    return (long)((Cout + NS - 1) / NS) * (Cin / 16) * 9 * NS * 16;
}

// GemmArgs -> tile geometry; false = not a problem this tile takes
static bool ws_from_args(const GemmArgs& a, WsProblem* p) {
    if (!a.conv || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W || a.act == ACT_GELU || a.rscale ||
        a.omap.G != 1 || (a.res && a.rmap.G != 1) || a.M <= 0 || a.H <= 0 || a.W <= 0 || a.M % (a.H * a.W) != 0)
        return false;
    if ((a.omap.S1 & 7) || (a.omap.off & 7) || (a.res && ((a.rmap.S1 & 7) || (a.rmap.off & 7)))) return false;     // 16-byte pieces
    if ((double)a.M * (double)a.omap.S1 * 2.0 >= 2.0e9 || (a.res && (double)a.M * (double)a.rmap.S1 * 2.0 >= 2.0e9)) return false;
    if (!ws_plan(a.M / (a.H * a.W), a.H, a.W, a.Cin, a.N, p)) return false;
    p->x = reinterpret_cast<const unsigned short*>(a.A);
    p->wp = reinterpret_cast<const unsigned short*>(a.Wp3);
    p->bias = a.bias;
    p->res = a.res ? reinterpret_cast<const unsigned short*>(a.res) + a.rmap.off : nullptr;
    p->y = reinterpret_cast<unsigned short*>(a.out) + a.omap.off;
    p->ldy = (int)a.omap.S1;
    p->ldr = a.res ? (int)a.rmap.S1 : (int)a.omap.S1;
    p->relu = a.act == ACT_RELU;
    return true;
}

bool gemm_bf16_ws_ok(const GemmArgs& a) {
    WsProblem p;
    return ws_from_args(a, &p);
}

int gemm_bf16_ws_tiles(const GemmArgs& a) {
    WsProblem p;
    return ws_from_args(a, &p) ? p.tiles_m * p.NSL : 0;
}

// Which kernel a conv runs on is a function of the conv ALONE (its shape and batch), never of what else shares its launch: the
// engine's schedules (one chain, two chains, program order) group a level's convs differently and promise identical bits
// (capf.h, tests/test_gpu_ops.py::test_two_chain_schedule_is_bit_identical_to_one_chain).  The HRNet branches of a level have equal
// FLOPs and very different tile counts (1024 ... 64 at batch 64), so the rule is on the problem's work: the tile takes a conv from
// 1 GFLOP up -- below that (HRNet-32 under batch ~14, HRNet-48 under ~6) a level is a handful of 256-pixel tiles and the ring
// kernel's 64 x 64 tiles fill the chip better.  (diag builds: CAPF_BF16_WS_MIN_MFLOP)
// Round 6 (tools/sweep_thresholds.py, tools/ws_sweep.py -> profiles/r06_threshold_sweep.txt): the FLOP rule alone let the tile in far too
// early -- HRNet-48 at batch 8 / 16 ran 21 % / 10 % SLOWER with it than on the row-halo / ring kernels, CPN at batch 4 - 16 5 %; the three
// backbones cross over between batch 16 and 32, where a launch's 256-pixel tiles start to fill the chip's 512 slots.  So: 1 GFLOP AND batch 24.
bool gemm_bf16_ws_wanted(const GemmArgs& a) {
    static const double min_flop = [] { const char* e = diag_env("CAPF_BF16_WS_MIN_MFLOP"); return (e ? atof(e) : 1000.0) * 1e6; }();
    static const long min_batch = [] { const char* e = diag_env("CAPF_BF16_WS_MIN_BATCH"); return e ? atol(e) : 24L; }();
    return a.Wp3 && a.Ho > 0 && a.Wo > 0 && (long)a.M >= min_batch * a.Ho * a.Wo && 2.0 * (double)a.M * a.N * 9.0 * a.Cin >= min_flop && gemm_bf16_ws_ok(a);
}

// Problems are laid out one after the other, longest K loop first, each padded to a multiple of 8 blocks so that block b of a
// problem runs on XCD b % 8 and walks that XCD's contiguous eighth of the tiles (neighbouring tiles share halo rows and, for
// several channel slices, the pixel tile in L2).  Measured and rejected: dealing the grid out in rounds, every problem in
// proportion to its size, so that memory-bound (48-channel) and matrix-bound (384-channel) tiles are co-resident throughout:
// HRNet-48 level at batch 256 224 -> 247 us, cfg2 14.39k -> 13.9k frames/s.
// Also measured and rejected: two streams -- the matrix-bound problems longest first, the memory-bound 48-channel one dealt evenly
// between them so that a CU's two slots hold one tile of each kind: 226 -> 228 us.  The tiles are bound by their own serial issue /
// wait chains (SQ_WAIT_INST_ANY 35-40 %, SQ_WAIT_ANY 30-36 % of the wave-cycles with two waves per SIMD), not by a shared roof.
struct WsGroupArgs {
    WsProblem g[MAXG];
    int start[MAXG + 1];
    int tiles[MAXG];
    int n;
};

__global__ __launch_bounds__(256, 2) void igemm_bf16_group_ws_kernel(WsGroupArgs ga) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < ga.n && b >= ga.start[pi + 1]) ++pi;
    const int l = b - ga.start[pi];
    const int per_xcd = (ga.start[pi + 1] - ga.start[pi]) >> 3;
    const int bid = (l & 7) * per_xcd + (l >> 3);
    if (bid >= ga.tiles[pi]) return;
    const WsProblem& p = ga.g[pi];
    switch (p.NS) {
        case 96: igemm_bf16_ws_tile<3>(p, bid, ws_lds); break;
        case 64: igemm_bf16_ws_tile<2>(p, bid, ws_lds); break;
        default: igemm_bf16_ws_tile<1>(p, bid, ws_lds); break;
    }
#endif
}

hipError_t launch_gemm_bf16_ws_group(const GemmArgs* list, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > MAXG) return hipErrorInvalidValue;
    struct Item { WsProblem p; double cost; };
    Item it[MAXG];
    int max_ns = 32;
    for (int i = 0; i < n; ++i) {
        if (!list[i].Wp3 || !ws_from_args(list[i], &it[i].p)) return hipErrorInvalidValue;
        it[i].cost = (double)(it[i].p.C / 16) * it[i].p.NS;      // a tile's K loop: longest first, so that the launch does not end on them
        if (it[i].p.NS > max_ns) max_ns = it[i].p.NS;
    }
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && it[j].cost > it[j - 1].cost; --j) { Item t = it[j]; it[j] = it[j - 1]; it[j - 1] = t; }
    WsGroupArgs ga;
    ga.n = n;
    int start = 0;
    for (int i = 0; i < n; ++i) {
        ga.g[i] = it[i].p;
        ga.tiles[i] = it[i].p.tiles_m * it[i].p.NSL;
        ga.start[i] = start;
        start += (ga.tiles[i] + 7) & ~7;
    }
    ga.start[n] = start;
    for (int i = n; i < MAXG; ++i) { ga.start[i + 1] = start; ga.tiles[i] = 0; ga.g[i] = ga.g[0]; }
    const size_t lds_bytes = 2 * (size_t)ws_stage_bytes(max_ns);
    static DynLdsAttr attr_once;
    const hipError_t attr = attr_once.ensure(reinterpret_cast<const void*>(&igemm_bf16_group_ws_kernel), 2 * ws_stage_bytes(96));
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(igemm_bf16_group_ws_kernel, dim3(start), dim3(256), lds_bytes, s, ga);
    return hipGetLastError();
}

// Measured and not adopted (commit 899a361 "persistent weight-resident form", profiles/r04_pmc_bf16_ws_level.txt): a persistent block per CU
// for the narrow branch (Cin 48: whole filter staged once, the next tile's pixels in flight under the current tile, one barrier per
// tile) -- bit-identical, 92 -> 83 us alone, but nothing end to end (cfg2 14.23k vs 14.19k frames/s): with one wave per SIMD its K
// loop, its VALU-heavy epilogue and its waits are strictly serial.
hipError_t launch_gemm_bf16_ws(const GemmArgs& a, hipStream_t s) { return launch_gemm_bf16_ws_group(&a, 1, s); }

const char* gemm_bf16_ws_kernel_name(const GemmArgs& a) {
    const int ns = ws_ns(a.N);
    return ns == 96 ? "igemm_bf16_ws<w4,256x96,conv>" : (ns == 64 ? "igemm_bf16_ws<w4,256x64,conv>" : "igemm_bf16_ws<w4,256x32,conv>");
}

// BN fold + re-layout for the tile: Wp[slice][Cin / 16][tap][n][quad position][8] = bf16(w[slice * NS + n][cc * 16 + 8 h + e][kh][kw] *
// gamma / sqrt(var + eps)), h = quad position ^ ((n >> 3) & 1) (the LDS image's bank swizzle, so that the DMA is a linear copy),
// rows beyond Cout zero; bias as launch_pack_conv
__global__ void pack_conv_bf16_ws_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                         unsigned short* __restrict__ Wp, float* __restrict__ bias, int Cout, int Cin, int NS, long total) {
    const int ncc = Cin / 16;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
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
        Wp[i] = to_bf16(v);
    }
}

hipError_t launch_pack_conv_bf16_ws(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                                    float eps, void* Wp_bf16, float* bias, int Cout, int Cin, hipStream_t s) {
    if (Cin % 16 != 0 || Cout <= 0) return hipErrorInvalidValue;
    const long total = bf16_ws_pack_elems(Cout, Cin);
    const long want = (total + 255) / 256;
    hipLaunchKernelGGL(pack_conv_bf16_ws_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, s, w, gamma, beta, mean, var, eps,
                       static_cast<unsigned short*>(Wp_bf16), bias, Cout, Cin, ws_ns(Cout), total);
    return hipGetLastError();
}

}  // namespace capf
