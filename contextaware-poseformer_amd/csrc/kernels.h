// Launcher declarations for the gfx950 kernels (internal; the public ABI is include/capf.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

namespace capf {

// A/B and tuning switches read the environment ONLY in the diagnostic build (make DIAG=1 -> tools/ab/libcapf_diag.so,
// -DCAPF_DIAG, loaded through CAPF_LIB by the tools); the product library ignores them, so an exported variable can never
// change a production plan or its numerics.  What tests need to steer is an explicit field: capf_config::plan_flags.
inline const char* diag_env(const char* name) {
#ifdef CAPF_DIAG
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// fp32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per instruction);
// the software form (5 integer ops per value) made the bf16 epilogues VALU-bound.  Host passes never call these.
__host__ __device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {          // lo in bits 0-15
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
#else
    (void)lo; (void)hi;
    return 0u;
#endif
}
__host__ __device__ __forceinline__ unsigned short to_bf16(float f) { return (unsigned short)(pack_bf16x2(f, 0.f) & 0xFFFFu); }

// Bilinear corner of F.grid_sample(mode='bilinear', align_corners=True) at normalised coordinates (gx, gy): the integer NW
// corner and the fractional weights of the +1 corners, exactly as ATen computes them (GridSampler.h:27-36, 58-60, 143-171:
// ((g + 1) / 2) * (size - 1); padding 'border' clips the COORDINATE to [0, size - 1] before floor, 'zeros' keeps it).  The
// result feeds an index gather that must be bit-exact, so every translation unit that instantiates this is built with
// -ffp-contract=off (csrc/Makefile: lifter.hip, lifter_fused.hip) -- no FMA contraction, IEEE division.
struct BilinearCorner {
    int x0, y0;
    float wx1, wy1;
};
#if defined(__HIPCC__)
template <bool BORDER>
__device__ __forceinline__ BilinearCorner bilinear_corner(float gx, float gy, int H, int W) {
    float x = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    float y = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    if (BORDER) {   // clip_coordinates: min(size - 1, max(x, 0)) before floor
        x = fminf((float)(W - 1), fmaxf(x, 0.0f));
        y = fminf((float)(H - 1), fmaxf(y, 0.0f));
    }
    const float xf = floorf(x), yf = floorf(y);
    BilinearCorner c;
    c.x0 = (int)xf;
    c.y0 = (int)yf;
    c.wx1 = x - xf;
    c.wy1 = y - yf;
    return c;
}
#endif

// addr(m) = (m / G) * S1 + (m % G) * S2 + off   (elements).  G == 1 -> plain leading dimension S1.
struct RowMap {
    int G;
    long S1, S2, off;
};
inline RowMap row_ld(long ld, long off = 0) { return RowMap{1, ld, 0, off}; }

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

// n / d for 0 <= n < 2^31:  q = (umulhi(n, mul) + n) >> shift   (host: make_fastdiv)
struct FastDiv {
    unsigned mul, shift;
};
FastDiv make_fastdiv(unsigned d);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: launchers that need more than the 64 KiB default of
// dynamic LDS keep one of these per kernel (a function-local static) and call ensure() before every launch -- one hipGetDevice, and the
// attribute call the first time a device is seen (a handle per device in one process: capf_create takes a device index)
struct DynLdsAttr {
    unsigned long long seen = 0;     // bit d: set on device d (a benign race: setting the attribute twice is harmless)
    hipError_t ensure(const void* kernel, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && ((seen >> dev) & 1ull)) return hipSuccess;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess && dev >= 0 && dev < 64) seen |= 1ull << dev;
        return e;
    }
};

// One implicit-GEMM problem:  out[m, n] = act( sum_k A[m, k] * Wp[n, k] + bias[n] + res[m, n] )
//   conv mode: A[m, k] gathered from an NHWC tensor, m = (b, ho, wo), k = (kh, kw, ci)
//   rows mode: A[m, k] = A[amap(m) + k]
struct GemmArgs {
    const float* A;
    const float* Wp;    // packed weights [N][Kpad], K-contiguous, zero padded to Kpad (multiple of 32)
    const float* Wp2;   // bf16 3x3 stride-1 convs: the same weights in the row-halo layout ([N][9 * Cin], igemm_bf16.hip), or
                        // nullptr; launch_gemm_bf16 / _group switch to that kernel for launches of >= 2048 tiles
    const float* Wp3;   // bf16 3x3 stride-1 convs: the same weights in the 2-D halo tile's layout (igemm_bf16_ws.hip), or nullptr;
                        // launch_gemm_bf16 / _group run the problems gemm_bf16_ws_wanted() accepts on that kernel.
                        // fp32 3x3 stride-1 convs: the weights as three bf16 pieces (igemm_f32x3_ws.hip), or nullptr;
                        // launch_gemm_wino / _group run the problems gemm_f32x3_wanted() accepts on that kernel
    int x3_h2;          // fp32 3x3 stride-1 convs: Wp3 holds two block-scaled fp16 pieces (igemm_f32h2_ws.hip, launch_pack_conv_f32h2) and the
                        // problems gemm_f32x3_wanted() accepts run on THAT tile (three piece products per fp32 MAC instead of six)
    const float* Wh2;   // the same weights as the two-fp16-piece pack of igemm_f32h2.hip (launch_pack_f32h2_gemm: Wp's [N][Kpad] geometry, then [N]
                        // inverse channel scales), or nullptr: launch_gemm_f32 / _group run the problem on that kernel where gemm_f32h2g_ok()
                        // accepts it (the HBM-bound pointwise kernels keep theirs)
    const float* bias;  // [N] or nullptr
    const float* res;   // residual, addressed by rmap, or nullptr
    float* out;         // addressed by omap
    int M, N, K, Kpad;
    int conv;           // 1 = conv mode
    int Cin, H, W, Ho, Wo, ks, stride, pad;
    RowMap amap, omap, rmap;
    int act;
    FastDiv fd_hw, fd_wo;   // filled by launch_gemm_f32 (conv mode): division by Ho*Wo and by Wo
    unsigned long long spread;   // conv mode: bits kh*ks set for kh < ks (tap-mask construction)
    const float* rscale;    // optional per-row scale of (acc + bias): rscale[m / rs_div] (DropPath), or nullptr
    int rs_div;
    int splits, cps;        // split-K (rows mode): grid.y slices of `cps` chunks, slab s at out + s*split_stride
    long split_stride;
    int out_bf16;           // small-Cin stem kernel only: fp32 image in, bf16 activations out
    // conv-mode split-K with an in-kernel, deterministic reduction (small batches: a long-K conv with a handful of tiles):
    // slice ky writes its raw partial tile to split_ws + ky * split_stride ([M][N]); the LAST slice to finish a tile (a
    // device-scope counter per tile, self-resetting) sums the slices in order 0..splits-1 and runs the epilogue
    float* split_ws;
    int* split_cnt;
    long split_ws_elems;    // capacity of the scratch the caller lends (floats / counters); the launchers carve it up
    int split_cnt_elems;
    const float* ln_g;      // rows mode: LayerNorm the A rows over their K columns on the fly (gamma, beta [K]); nullptr = off
    const float* ln_b;
    float ln_eps;
    // bf16 conv mode: out = act(acc + bias (+ res)) + bilinear_upsample(up) -- a map [B][up_H][up_W][N] (bf16, dense) resized to the conv's
    // Ho x Wo with align_corners = True and added AFTER the activation (CPN globalNet.py:66: lateral + upsampled path), in the epilogue of
    // igemm_bf16_kernel<..., UPADD>; nullptr = off.  up_sh / up_sw = (up_H - 1) / (Ho - 1), (up_W - 1) / (Wo - 1) as launch_bilinear_resize has them
    const void* up;
    int up_H, up_W;
    float up_sh, up_sw;
    // the two-fp16-piece conv tile only (igemm_f32h2_ws_tile.h, PLANES): A holds the producer's fp16 planes and h2_ein their [tile][chunk]
    // scale exponents / out is written as planes and h2_eout receives the exponents.  nullptr = plain fp32 tensors
    const int* h2_ein;
    int* h2_eout;
    const unsigned* h2_utab;     // ... and the map geometry's unit table (f32h2_unit_table; nullptr: the prologue computes its addresses)
};

// all res blocks of the lifter as one launch (lifter_chain.hip): per block LayerNorm weights, the two-fp16-piece packs of the four projections
// (launch_pack_f32h2_gemm_rows' [N][K floats] of {piece 0 | piece 1} chunks + [N] inverse scales, through launch_res_chain_repack) and their fp32 biases
struct ResBlockW {
    const float *ln1_g, *ln1_b, *wqkv, *bqkv, *wproj, *bproj, *ln2_g, *ln2_b, *wfc1, *bfc1, *wfc2, *bfc2;
};
bool res_chain_ok(int dim, int tokens, int heads, int nblk);
// a projection's two-piece pack re-laid out in MFMA fragment order (same bits; what ResBlockW's w* point at)
hipError_t launch_res_chain_repack(const float* h2g_pack, float* chain_pack, int N, int K, hipStream_t s);
hipError_t launch_res_chain(float* X, int rows, int tokens, int heads, float eps, const ResBlockW* blk, int nblk, hipStream_t s);
hipError_t launch_mlp_chain(float* X, RowMap rows_map, int rows, float eps, const ResBlockW& blk, hipStream_t s);     // the MLP half alone, on mapped rows

hipError_t launch_gemm_f32(const GemmArgs& a, hipStream_t s);
// several independent fp32 convs in one grid (see igemm_f32.hip "Grouped launch"); n <= MAXG, every problem
// must satisfy gemm_f32_groupable()
static constexpr int MAXG = 8;
bool gemm_f32_groupable(const GemmArgs& a);
bool gemm_f32_rows_splitk(const GemmArgs& a);        // rows-mode GEMM that launch_gemm_f32 splits along K (few tiles, long K, scratch lent)
hipError_t launch_gemm_f32_group(const GemmArgs* list, int n, hipStream_t s);
const char* gemm_f32_kernel_name(const GemmArgs& a);   // which template instantiation launch_gemm_f32 picks
// fp32 pointwise (1x1 / stride 1) conv for the HBM-bound bottleneck convs of layer1 (igemm_f32_pw.hip): ping-pong schedule,
// coalesced epilogue with prefetched residual; launch_gemm_f32 routes eligible problems (>= 2048 tiles) to it
bool gemm_f32_pw_ok(const GemmArgs& a);
hipError_t launch_gemm_f32_pw(const GemmArgs& a, hipStream_t s);
const char* gemm_f32_pw_kernel_name();
// two pointwise convs back to back (64 -> 256 + residual + ReLU, then 256 -> 64 + ReLU on its output) as ONE launch: the first conv's
// accumulators are the second conv's A operand (igemm_f32_pwchain.hip); bit-identical to the two launches
bool gemm_f32_pwchain_ok(const GemmArgs& a, const GemmArgs& b);
hipError_t launch_gemm_f32_pwchain(const GemmArgs& a, const GemmArgs& b, hipStream_t s);
const char* gemm_f32_pwchain_kernel_name();
// the first bottleneck of a layer1 under compute_dtype = bf16 as one persistent kernel (bneck_bf16.hip): conv1 -> conv2 -> conv3 + downsample
// shortcut + ReLU, intermediates on chip; tap = also store conv1's, conv2's and the downsample's outputs where the unfused ops write them
bool bneck0_bf16_ok(const GemmArgs& c1, const GemmArgs& c2, const GemmArgs& ds, const GemmArgs& c3);
hipError_t launch_bneck0_bf16(const GemmArgs& c1, const GemmArgs& c2, const GemmArgs& ds, const GemmArgs& c3, bool tap, hipStream_t s);
const char* bneck0_bf16_kernel_name();
// ... and an identity bottleneck (256 -> 64 -> 64 -> 256, y = relu(conv3 + x)) the same way
bool bneck1_bf16_ok(const GemmArgs& c1, const GemmArgs& c2, const GemmArgs& c3);
hipError_t launch_bneck1_bf16(const GemmArgs& c1, const GemmArgs& c2, const GemmArgs& c3, bool tap, hipStream_t s);
const char* bneck1_bf16_kernel_name();
// the bf16 twin (igemm_bf16_pwchain.hip): CPN's / HRNet's layer1 pairs under compute_dtype = bf16
bool gemm_bf16_pwchain_ok(const GemmArgs& a, const GemmArgs& b);
hipError_t launch_gemm_bf16_pwchain(const GemmArgs& a, const GemmArgs& b, hipStream_t s);
const char* gemm_bf16_pwchain_kernel_name();

// Winograd F(2,3)-along-W variant of the 3x3 / stride-1 / pad-1 fp32 conv (igemm_wino.hip): same GemmArgs as the direct conv,
// Wp = weights packed by launch_pack_conv_wino ([N][12 * Cin]); needs Cin % 32 == 0, even W, N % 4 == 0
bool gemm_wino_ok(const GemmArgs& a);
hipError_t launch_gemm_wino(const GemmArgs& a, hipStream_t s);
hipError_t launch_gemm_wino_group(const GemmArgs* list, int n, hipStream_t s);
const char* gemm_wino_kernel_name(const GemmArgs& a);
hipError_t launch_pack_conv_wino(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                                 float eps, float* Wp, float* bias, int Cout, int Cin, hipStream_t s, int variant = 23);

// conv weight fold + pack:  Wp[n][(kh*ks+kw)*Cin+ci] = w[n][ci][kh][kw] * gamma[n]/sqrt(var[n]+eps)
//                            bias[n] = beta[n] - mean[n]*gamma[n]/sqrt(var[n]+eps)
hipError_t launch_pack_conv(const float* w, const float* gamma, const float* beta, const float* mean,
                            const float* var, float eps, float* Wp, float* bias, int Cout, int Cin,
                            int ks, int Kpad, hipStream_t s);
// linear pack: Wp[n][k] = w[n][k] (zero padded to Kpad); rows [n0, n0+N) of the destination
hipError_t launch_pack_conv_bf16(const float* w, const float* gamma, const float* beta, const float* mean,
                                 const float* var, float eps, void* Wp_bf16, float* bias, int Cout, int Cin, int ks,
                                 int Kpad, hipStream_t s);
// bf16 conv (igemm_bf16.hip): A / res / out bf16 NHWC, Wp bf16 [N][Kpad], Kpad % 64 == 0, bias fp32
hipError_t launch_gemm_bf16(const GemmArgs& a, hipStream_t s);
int f32h2_tiles_m(int B, int H, int W, int* tile_pixels = nullptr);            // pixel tiles of the two-fp16-piece conv tile (rows of a planes tensor's exponent table); 0: not eligible
bool gemm_bf16_groupable(const GemmArgs& a);
bool gemm_bf16_upadd_ok(const GemmArgs& a);      // a.up (post-activation upsampled add) can run in igemm_bf16_kernel<.., UPADD>
// bf16 twin of launch_gemm_f32_group; *variant (optional) = the device kernel it chose: 0 ring (igemm_bf16_group_kernel),
// 1 ping-pong (igemm_bf16_group_pp_kernel), 2 ping-pong with row-halo tiles (igemm_bf16_group_rh_kernel), 3 the 2-D halo tile
// (igemm_bf16_group_ws_kernel; problems of the list it cannot take go out as a second, ring / ping-pong launch), -1 single launch
hipError_t launch_gemm_bf16_group(const GemmArgs* list, int n, hipStream_t s, int* variant = nullptr);
const char* gemm_bf16_kernel_name(const GemmArgs& a);
// row-halo variant of the 3x3 / stride-1 bf16 conv (one staged A tile serves the three kw taps): chunk width 64 / 48 / 32 or
// 0 = not eligible; weights packed by launch_pack_conv_bf16_rh ([N][9 * Cin], K order (kh, Cin / CW, kw, CW))
int bf16_rh_width(int Cin);
int gemm_bf16_rh_cw(const GemmArgs& a);
hipError_t launch_gemm_bf16_rh(const GemmArgs& a, hipStream_t s);
hipError_t launch_pack_conv_bf16_rh(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                                    float eps, void* Wp_bf16, float* bias, int Cout, int Cin, int CW, hipStream_t s);
// "2-D halo" tile of the 3x3 / stride-1 bf16 conv (igemm_bf16_ws.hip, igemm_bf16_ws_tile.h): 256 pixels x 32 / 64 / 96 channels per
// block with the accumulators resident for the whole K, 16-channel chunks staged once for all nine taps; weights packed by
// launch_pack_conv_bf16_ws (bf16_ws_pack_elems(Cout, Cin) bf16 elements), passed as GemmArgs::Wp3
bool gemm_bf16_ws_ok(const GemmArgs& a);
int gemm_bf16_ws_tiles(const GemmArgs& a);              // blocks the problem needs (0 = not eligible)
bool gemm_bf16_ws_wanted(const GemmArgs& a);            // eligible, carries Wp3, and large enough for this tile (a function of the conv alone)
long bf16_ws_pack_elems(int Cout, int Cin);
hipError_t launch_gemm_bf16_ws(const GemmArgs& a, hipStream_t s);
hipError_t launch_gemm_bf16_ws_group(const GemmArgs* list, int n, hipStream_t s);
const char* gemm_bf16_ws_kernel_name(const GemmArgs& a);
hipError_t launch_pack_conv_bf16_ws(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                                    float eps, void* Wp_bf16, float* bias, int Cout, int Cin, hipStream_t s);

// fp32 3x3 / stride-1 conv on the bf16 matrix pipe (igemm_f32x3_ws.hip, igemm_f32x3_ws_tile.h): fp32 tensors in and out, every operand split
// losslessly into three bf16 pieces, the six piece products of weight >= 2^-18 accumulated in fp32 -- results to fp32 accumulation
// order.  Any width up to 256, Cin % 16 == 0, Cout % 4 == 0; weights packed by launch_pack_conv_f32x3 (f32x3_pack_elems(Cout, Cin)
// bf16 elements), passed as GemmArgs::Wp3
bool gemm_f32x3_ok(const GemmArgs& a);
bool gemm_f32x3_wanted(const GemmArgs& a);             // eligible, carries Wp3, and large enough for this tile (a function of the conv alone)
bool f32x3_takes(int B, int H, int W, int Cin, int Cout, bool h2);   // the shape half of that rule (what the engine asks before it picks a weight layout)
long f32x3_pack_elems(int Cout, int Cin);
hipError_t launch_gemm_f32x3(const GemmArgs& a, hipStream_t s);
hipError_t launch_gemm_f32x3_group(const GemmArgs* list, int n, hipStream_t s);
const char* gemm_f32x3_kernel_name(const GemmArgs& a);
hipError_t launch_pack_conv_f32x3(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                  void* Wp_bf16, float* bias, int Cout, int Cin, hipStream_t s);
// the same convs with HALF the MFMAs (igemm_f32h2_ws.hip, igemm_f32h2_ws_tile.h): every operand as two fp16 pieces under an exact power-of-two
// block scale (per output channel for the weights, per block and 16-channel chunk for the pixels, found in the kernel), three piece products
// per fp32 MAC, fp32 accumulation; operands to 2^-23, one product to 2^-21 -- below the fp32 accumulation error of the dot product it
// belongs to.  Same eligibility and size rule (gemm_f32x3_wanted); the problems of a list with GemmArgs::x3_h2 set run on this tile
// (launch_gemm_f32x3_group forwards them).  Weights packed by launch_pack_conv_f32h2 (f32h2_pack_elems(Cout, Cin) 16-bit elements)
long f32h2_pack_elems(int Cout, int Cin);
hipError_t launch_gemm_f32h2_group(const GemmArgs* list, int n, hipStream_t s);
bool gemm_f32h2_ok(const GemmArgs& a);
static constexpr int F32H2_UNIT_TABLE_WORDS = 7 * 256;
bool f32h2_unit_table(int H, int W, int Cin, unsigned* out);     // the tile's unit table of a map geometry (igemm_f32h2_ws_tile.h); false: none, the kernel computes
bool f32h2_shape_ok(int B, int H, int W, int Cin, int Cout);     // (tensors of any size: the tile addresses from per-tile bases)
const char* gemm_f32h2_kernel_name(const GemmArgs& a);
hipError_t launch_pack_conv_f32h2(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                  void* Wp_f16, float* bias, int Cout, int Cin, hipStream_t s);
// ... and everything else that is fp32 and MFMA-shaped -- 1x1 / stride-2 / lone convs, the lifter's linears -- on the same two-piece arithmetic
// (igemm_f32h2.hip): the A tile is staged as fp32 exactly as igemm_f32.hip stages it and split by the wave that consumes it (scale per wave,
// 32 rows and 32-deep chunk); weights packed by launch_pack_f32h2_gemm over the fp32 pack's geometry (GemmArgs::Wh2)
long f32h2_gemm_pack_elems(int N, int Kpad);            // floats
bool gemm_f32h2g_ok(const GemmArgs& a);
bool gemm_f32_on_h2g(const GemmArgs& a);               // ... and launch_gemm_f32 / _group send it there (igemm_f32.hip: not the pointwise kernel's expansions)
hipError_t launch_gemm_f32h2g(const GemmArgs& a, hipStream_t s);
hipError_t launch_gemm_f32h2g_group(const GemmArgs* list, int n, hipStream_t s);     // convs with plain row maps, one grid
const char* gemm_f32h2g_kernel_name(const GemmArgs& a, bool grouped);
hipError_t launch_pack_f32h2_gemm(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                  float* Wp, float* bias, int N, int Cin, int ks, int K, int Kpad, hipStream_t s);   // ks = 0: linear [N][K]
// rows [n0, n0 + n) of an [Ntot][Kpad] pack from one nn.Linear weight [n][K] (several linears concatenated along N share a pack)
hipError_t launch_pack_f32h2_gemm_rows(const float* w, float* Wp, int n0, int n, int Ntot, int K, int Kpad, hipStream_t s);
// One matrix of the training step's weight table: W [N][K] (row pitch ld) -> its two-piece pack at base + fwd_off (Kpad = K rounded up to 32)
// and the pack of W^T [K][N] at base + bwd_off (Kpad = N rounded up to 32); either offset may be -1 (not wanted).  tile_start: first block of
// this matrix in the launch (ceil(N / 32) * ceil(K / 32) blocks each); max_off: its N row + K column maxima in the scratch
struct H2TrainW {
    const float* w;
    long fwd_off, bwd_off;
    int N, K, ld, tile_start, max_off, pad_;
};
hipError_t launch_pack_f32h2_train(const H2TrainW* tab_dev, int n, int tiles, float* base, int* maxima, long maxima_elems, hipStream_t s);
bool gemm_bf16_smallc_ok(const GemmArgs& a);            // the stem conv (Cin = 3) with a bf16 result
hipError_t launch_gemm_bf16_smallc(const GemmArgs& a, hipStream_t s);
const char* gemm_bf16_smallc_kernel_name(const GemmArgs& a);
hipError_t launch_pack_linear(const float* w, float* Wp, int N, int K, int Kpad, hipStream_t s);
// quad-interleaved pack for the fused lifter kernels: Wq[(k / 4) * Ntot + n0 + n][k % 4] = w[n][k]  (K % 4 == 0): lane n of a
// wave reads the 16-byte quad next to lane n - 1's instead of a row K floats away
hipError_t launch_pack_linear_quad(const float* w, float* Wq, int N, int K, int n0, int Ntot, hipStream_t s);
// lifter projections on the bf16 MFMA path (igemm_bf16.hip): A bf16 [M][K], W bf16 [N][Kpad]; gelu_bf16_out = 0: fp32 out
// (+ fp32 residual) through the row maps; 1: GELU then bf16 out [M][N]
hipError_t launch_gemm_bf16_rows(const void* A_bf16, const void* W_bf16, const float* bias, int M, int N, int K, int Kpad,
                                 float* out, RowMap omap, const float* res, RowMap rmap, int gelu_bf16_out, hipStream_t s);
const char* gemm_bf16_rows_kernel_name(int M, int N);

// out = relu( sum_i up_{s_i}(in_i) ), NHWC, s_i = nearest-upsample factor (1 = same resolution)
struct FuseSumArgs {
    const float* in[4];
    int shift[4];  // log2 of the upsample factor
    int n_in;
    float* out;
    int B, H, W, C;
    int relu;
    int bf16;      // tensors are bf16 (arithmetic stays fp32)
};
hipError_t launch_fuse_sum(const FuseSumArgs& a, hipStream_t s);
// up to 4 independent sums (the outputs of one HRNet fuse module) as ONE launch; same arithmetic per element as launch_fuse_sum
hipError_t launch_fuse_sum_group(const FuseSumArgs* a, int n, hipStream_t s);

// 3x3 s2 p1 max-pool NHWC (resnet.py:140), bilinear align_corners=True resize NHWC (+ optional add)
hipError_t launch_maxpool3x3s2(const float* in, float* out, int B, int H, int W, int C, int Ho, int Wo,
                               hipStream_t s, int bf16 = 0);
hipError_t launch_bilinear_resize(const float* in, float* out, int B, int H, int W, int C, int Ho, int Wo,
                                  hipStream_t s, int bf16 = 0, const float* add = nullptr);   // out = resize(in) (+ add, same shape as out)

// n small float copies in one launch; the table ({src, dst, n} per segment) is in device memory
struct CopySegment {
    const float* src;
    float* dst;
    int n, pad;
};
hipError_t launch_copy_segments(const CopySegment* table_dev, int n, hipStream_t s);

// ---- lifter -----------------------------------------------------------------------------------
// kcrop -> ref in place (conpose.py:34-35);  X[b,p,0,:] = coord_embed(k2d[b,p]) + pos[0,p,:]
hipError_t launch_prep_embed(float* kcrop, const float* k2d, const float* w, const float* bias,
                             const float* pos, float* X, int B, int J, int L1, int C, hipStream_t s);
// reference-point sampling, padding zeros (pose_dformer.py:216-218): S[b,p,:] = bilinear(feat, ref[b,p])
hipError_t launch_sample_ref(const float* feat, const float* ref, float* S, int* idx, int B, int J, int H,
                             int W, int C, hipStream_t s, int feat_bf16 = 0);
// LayerNorm over rows: out[r,:] = LN(in[imap(r)] (+ add[amap(r)]))   width C
hipError_t launch_layernorm(const float* in, RowMap imap, const float* add, RowMap amap, const float* g,
                            const float* b, float eps, float* out, int rows, int C, hipStream_t s, int out_bf16 = 0);
// deformable sampling (pose_dformer.py:122-135 minus the embed_proj GEMM):
//   AO [rows=(b,p,l), NH*NS + 2*NH*NS] = [attention logits | offset pre-activations]
//   U_l[(b,p,h), :] = sum_s softmax_s(logit[h,s]) * bilinear_border(feat_l, tanh(off[h,s]) + ref[b,p])
struct DeformArgs {
    const float* feat[4];
    int H[4], W[4], C[4];
    float* U[4];
    const float* AO;
    const float* ref;
    int B, J, L, NH, NS;
    int feat_bf16;           // the context maps are bf16
    int ld_ao;               // row pitch of AO (0 = 3*NH*NS, the inference layout; 64 in training)
    const float* dU[4];      // backward only: gradient w.r.t. U[l]
    float* cpos;             // optional taps (capf_set_debug): sampling positions pos = tanh(off) + ref, [B*J, L, NH*NS, 2] ...
    int* cidx;               // ... and the NW corner (x0, y0) of each, border mode (bit-exact check against the oracle)
};
hipError_t launch_deform_sample(const DeformArgs& a, hipStream_t s);
// the corner arithmetic of both sampling sites on caller-supplied coordinates (op-level parity test): grid [n, 2] normalised
// (x, y) -> idx [n, 2] = (x0, y0), frac [n, 2] = (wx1, wy1); border = 1: padding_mode='border' (pose_dformer.py:128), 0: zeros (:217)
hipError_t launch_bilinear_corners(const float* grid, int n, int H, int W, int border, int* idx, float* frac, hipStream_t s);
// ---- fused front half of the lifter (lifter_fused.hip) -----------------------------------------------------------
// crop-keypoint normalisation (in place) + coord_embed + reference-point sampling (padding zeros) + feat_embed + pos
struct EmbedArgs {
    float* kcrop;                // [BJ, 2] in: crop pixels, out: ref
    const float* k2d;            // [BJ, 2]
    const float* cw; const float* cb;       // coord_embed weight [C, 2], bias [C]
    const float* pos;            // Spatial_pos_embed [1, L1, J, C]
    const float* feat[4]; int H[4], W[4], Cl[4];
    const float* fw[4]; const float* fb[4]; // feat_embed[l] weight [C, Cl], bias [C]
    float* sampled[4];           // optional taps: sampled rows [BJ, Cl]
    int* idx[4];                 // optional taps: NW corner (x0, y0) per (b, p)
    float* X;                    // tokens [B, J, L1, C]
    int BJ, J, L, L1, C;
    int feat_bf16;
};
hipError_t launch_embed(const EmbedArgs& a, hipStream_t s);
// DeformableBlock attention half (LayerNorm + logits / offsets + sampling + embed_proj + residual), see lifter_fused.hip
struct CtxAttnArgs {
    const float* feat[4]; int H[4], W[4], Cl[4];
    const float* Wp[4]; const float* bp[4]; // embed_proj[l] weight [C/NH, Cl], bias [C/NH]
    const float* Wao; const float* bao;     // [attention_weights | sampling_offsets] quad-interleaved: Wq[C / 4][3*NH*NS][4]; bias
    int ldw;
    const float* ln_g; const float* ln_b; float eps;
    const float* ref;            // [BJ, 2]
    float* X;                    // tokens [B, J, L1, C], updated in place (tokens 1..L)
    int BJ, J, L, L1, C, NH, NS;
    int feat_bf16;
    int woff[4];                 // per-wave LDS section offsets (floats), filled by the launcher
    float* cpos;                 // optional taps (capf_set_debug): sampling positions [BJ, L, NH*NS, 2] and their NW corners
    int* cidx;
    float* U[4];                 // optional: per level [BJ * NH][Cl] scratch -- the per-head weighted sample sums leave the kernel and embed_proj +
                                 // residual run as ONE fp32-MFMA launch over all levels behind it (ctx_proj_kernel); null: inside the kernel
};
hipError_t launch_ctx_attn(const CtxAttnArgs& a, hipStream_t s);
// tiny multi-head attention: QKV [G*N, 3*heads*d] -> O [G*N, heads*d]; N tokens per group
hipError_t launch_attention(const float* qkv, float* out, int groups, int N, int heads, int d, hipStream_t s, int out_bf16 = 0);
// head (pose_dformer.py:240): out[r, 0..2] = Linear(LN(X[r,:]))
hipError_t launch_head(const float* X, const float* g, const float* b, float eps, const float* w,
                       const float* wb, float* out, int rows, int C, int NO, hipStream_t s);

// ---- training step (train_kernels.hip) ----------------------------------------------------------------
hipError_t launch_layernorm_train(const float* in, RowMap imap, const float* add, RowMap amap, const float* g,
                                  const float* b, float eps, float* out, float* xhat, float* rstd, int rows, int C,
                                  hipStream_t s);
hipError_t launch_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dX,
                                RowMap omap, float* second, RowMap smap, int rows, int GRP, int C, hipStream_t s);
// dst[c*dst_stride] (+)= sum_r A[amap(r)+c] * B(r,c); bmode 0 none, 1 B[bmap(r)+c], 2 B[bmap(r)]; scratch >= 64*C floats
// `defer` (optional): the second stage is NOT launched but appended to the batch (launch_colreduce_final_batch runs every waiting one in a
// single launch; the scratch must stay untouched until then)
static constexpr int COL_BATCH_MAX = 40;
struct ColFinalBatch {
    const float* partial[COL_BATCH_MAX];
    const float* partial2[COL_BATCH_MAX];
    float* dst[COL_BATCH_MAX];
    float* dst2[COL_BATCH_MAX];
    int chunks[COL_BATCH_MAX], C[COL_BATCH_MAX], dst_stride[COL_BATCH_MAX];
    int blk_end[COL_BATCH_MAX];       // running block count (64 columns per block)
    int count;
};
int colreduce_chunks(int rows, int C, int nout, size_t scratch_elems, int* rows_per_chunk = nullptr);   // row chunks of the first stage: scratch = nout * chunks * C floats
hipError_t launch_colreduce(const float* A, RowMap amap, const float* Bm, RowMap bmap, int bmode, int rows, int C,
                            float* dst, long dst_stride, int accumulate, float* scratch, hipStream_t s,
                            float* dst2 = nullptr, size_t scratch_elems = 0, ColFinalBatch* defer = nullptr);   // dst2: plain column sums of A as well
hipError_t launch_colreduce_final_batch(const ColFinalBatch& b, hipStream_t s);
hipError_t launch_slab_sum(const float* slabs, int nslab, long n, float* dst, hipStream_t s);
// the same sums for up to SLAB_BATCH_MAX weight gradients in ONE launch (the backward defers them: Engine::t_slab_flush).  Every job: n % 4 == 0,
// 16-byte aligned slabs; slabs summed in order k = 0, 1, ... per element exactly as slab_sum_kernel does (same bits)
static constexpr int SLAB_BATCH_MAX = 48;
struct SlabBatch {
    const float* src[SLAB_BATCH_MAX];
    float* dst[SLAB_BATCH_MAX];
    int n[SLAB_BATCH_MAX];            // elements per slab
    int nslab[SLAB_BATCH_MAX];
    int blk_end[SLAB_BATCH_MAX];      // running block count: job j owns blocks [blk_end[j - 1], blk_end[j]), 1024 elements each
    int count;
};
hipError_t launch_slab_sum_batch(const SlabBatch& b, hipStream_t s);
// dW[n][k] = sum_m dY[m][n] X[m][k] (+ db[n] = sum_m dY[m][n] behind it) from row-major operands, no transposes (train_kernels.hip);
// N % 4 == 0, K % 4 == 0; slice s of `splits` row ranges writes N * K (+ N) floats at out + s * slab.  h2: both operands as two fp16
// pieces split in the kernel, three piece products on the 16-bit matrix pipe, 128 x 128 tiles (N % 128 == 0, K % 128 == 0)
hipError_t launch_wgrad_tn(const float* dY, long ldy, const float* X, long ldx, int M, int N, int K, float* out, long slab, int splits,
                           int want_bias, hipStream_t s, bool h2 = false);
hipError_t launch_gelu_fwd(const float* x, float* y, long n, hipStream_t s);
hipError_t launch_gelu_bwd(const float* x, const float* dy, float* dx, long n, hipStream_t s);
hipError_t launch_transpose_pad(const float* in, RowMap imap, int M, int C, float* out, int Mp, hipStream_t s);
hipError_t launch_attention_bwd(const float* qkv, const float* dO, float* dqkv, int groups, int N, int heads, int d,
                                hipStream_t s);
hipError_t launch_deform_bwd(const DeformArgs& a, float* dAO, int ldd, hipStream_t s);
hipError_t launch_mpjpe(const float* pred, const float* gt, int rows, float* loss, float* dpred, float gscale,
                        hipStream_t s);
hipError_t launch_mpjpe_nd(const float* pred, const float* gt, int rows, int D, float* loss, float* dpred, float gscale,
                           hipStream_t s);
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                        float wd, int step, hipStream_t s, float gscale = 1.0f);
// dst[r, :] = src[smap(r), :] * scale[r / div]   (DropPath mask on a gradient), width C
hipError_t launch_scale_rows(const float* src, RowMap smap, const float* scale, int div, float* dst, int rows, int C,
                             hipStream_t s);
// head backward pieces: dY[r, c] = sum_o dOut[r, o] * W[o, c]  (NO <= 4)
hipError_t launch_head_dgrad(const float* dOut, const float* W, float* dY, int rows, int C, int NO, hipStream_t s);
// dpos[l, p, c] = sum_b dX[b, p, l, c]
hipError_t launch_pos_grad(const float* dX, float* dpos, int B, int J, int L1, int C, hipStream_t s);

// ---- neighbours of the path (preprocess.hip): N1 input preprocessing, N2 flip-test fusion ---------------
// mode 0: as is; 1: train-time horizontal flip of the whole batch; 2: flip-test (outputs hold [orig | mirrored])
hipError_t launch_preprocess(const unsigned char* images_bgr, int B, int H, int W, const float mean[3], const float* stdv,
                             int mode, float* images_out, const float* gt_in, float* gt_out, const float* k2d_in,
                             float* k2d_out, const float* kc_in, float* kc_out, hipStream_t s);
hipError_t launch_fliptest_fuse(const float* pred2, int B, float* out, hipStream_t s);
// N3 (preprocess.hip): get_affine_transform (host) and cv2.warpAffine INTER_LINEAR for 8-bit BGR frames
bool affine_from_center_scale(const double center[2], const double scale[2], int out_w, int out_h, double M[6]);
hipError_t launch_warp_affine_u8(const unsigned char* const* frames, const int* dims, const double* M, unsigned char* out,
                                 int B, int out_h, int out_w, hipStream_t s);

// ---- evaluation metrics (metrics.hip): N2 -------------------------------------------------------------------
hipError_t launch_pose_errors(const float* pred, const float* gt, int n, int J, const int* prev, float* err, hipStream_t s);
hipError_t launch_segment_sums(const float* err, const int* seg, const int* prev, int n, int n_seg, double* sums, int* counts,
                               hipStream_t s);
hipError_t launch_keypoints_loss(const float* pred, const float* gt, const float* validity, int rows, int D, int mode, float thr,
                                 float* loss, float* dpred, hipStream_t s);

}  // namespace capf
