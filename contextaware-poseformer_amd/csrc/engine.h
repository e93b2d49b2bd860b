// Internal model of the hot path: parameter schema, static layer plan, workspace layout.
// (Public ABI: include/capf.h.)  Host-side C++; no torch types anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "capf.h"
#include "kernels.h"

namespace capf {

struct Param {
    std::string name;
    int64_t shape[4] = {0, 0, 0, 0};
    int ndim = 0;
    int kind = 0;
    const float* ptr = nullptr;  // borrowed device pointer (capf_set_param)
    int64_t numel() const {
        int64_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= shape[i];
        return n;
    }
};

// Activation buffer inside the caller-owned workspace.  Sizes / offsets are per frame (elements);
// the run-time address is ws + offset * batch, so one plan serves every batch <= max_batch.
struct Buffer {
    size_t elems = 0;       // per frame, rounded up to 64 elements (256 B)
    size_t offset = 0;      // per frame, assigned by assign_offsets()
    int def_op = -1;        // first op that writes it
    int last_op = -1;       // last op that reads it (INT_MAX: persists to the end of forward)
    std::string tag;        // debug name ("feat0", "tok", ...)
};

struct Tensor {  // NHWC view of a buffer (per frame), or [rows, C] for the lifter (H = rows, W = 1)
    int buf = -1;
    int H = 0, W = 0, C = 0;
};

// A private packed copy derived from borrowed parameters (rebuilt by capf_params_changed).
struct Pack {
    int kind = 0;            // 0 conv+BN fold, 1 linear (possibly several linears concatenated along N)
    int w[4] = {-1, -1, -1, -1}, b[4] = {-1, -1, -1, -1};   // param indices (linear: up to 4 concatenated)
    int n_lin = 0;
    int bn_g = -1, bn_b = -1, bn_m = -1, bn_v = -1;
    int N = 0, K = 0, Kpad = 0, Cin = 0, ks = 1;
    size_t w_off = 0, b_off = 0;   // element offsets inside the pack arena
    size_t w2_off = 0;             // wino packs: the direct-kernel layout [N][Kpad2] as well (small batches run the direct kernel)
    int Kpad2 = 0;
    bool quad = false;             // linear for the fused lifter kernels: Wq[k / 4][n][4] (lanes n read consecutive 16-byte quads)
    bool direct = false;           // linear with Kpad == K and no concat: use the parameter in place
    bool bf16 = false;             // conv weights packed as bf16 (Kpad % 64 == 0)
    bool rh = false;               // bf16 3x3 stride-1 conv: a second copy in the row-halo layout ([N][9 * Cin] bf16) at w2_off
    bool ws = false;               // bf16 3x3 stride-1 conv: a copy in the 2-D halo tile's layout (igemm_bf16_ws.hip) at w3_off
    bool x3 = false;               // fp32 3x3 stride-1 conv: a copy for the split-fp32 tiles at w3_off -- two block-scaled fp16 pieces
                                   // (igemm_f32h2_ws.hip; Engine::x3_h2) or three bf16 pieces (igemm_f32x3_ws.hip)
    size_t w3_off = 0;
    bool h2g = false;              // fp32 conv / linear: a copy as two block-scaled fp16 pieces for igemm_f32h2.hip ([N][KpadH] floats + [N] inverse
    size_t wc_off = 0;             // chain: the h2g pack in MFMA fragment order for lifter_chain.hip (launch_res_chain_repack)
    bool chain = false;
    size_t wh_off = 0;             // channel scales) at wh_off; KpadH = the direct fp32 layout's padded K
    int KpadH = 0;
    bool wino = false;             // conv weights in the Winograd F(2,3) layout of igemm_wino.hip (Kpad = 12 * Cin)
    bool wino_skip = false;        // ... which no batch up to cfg.max_batch can reach (a split-fp32 tile takes the conv from its first Winograd batch to
                                   // max_batch): not packed, no arena space; the conv never runs a Winograd kernel (build())
};

enum OpKind {
    OP_GEMM = 0, OP_FUSE, OP_MAXPOOL, OP_RESIZE, OP_PREP_EMBED, OP_SAMPLE_REF, OP_LAYERNORM, OP_DEFORM,
    OP_ATTENTION, OP_HEAD, OP_FORK, OP_JOIN, OP_EMBED, OP_CTX_ATTN, OP_RES_CHAIN, OP_MLP_CHAIN
};

struct Op {
    OpKind kind = OP_GEMM;
    std::string name;
    // buffers (ids; -1 unused; -2 = external "images" input)
    int in[4] = {-1, -1, -1, -1};
    int aux = -1;                 // residual / add input
    int out = -1;
    int aux2 = -1;                // secondary output (idx buffer)
    // gemm
    int pack = -1;
    int conv = 0, Cin = 0, H = 0, W = 0, Ho = 0, Wo = 0, ks = 1, stride = 1, pad = 0;
    long rows_per_frame = 0;      // M = rows_per_frame * batch
    int N = 0, K = 0, act = 0;
    RowMap amap{1, 0, 0, 0}, omap{1, 0, 0, 0}, rmap{1, 0, 0, 0};
    int res_param = -1;           // residual read from a parameter (pos-embed) instead of a buffer
    // fuse / resize / pool
    int n_in = 0, shift[4] = {0, 0, 0, 0}, relu = 0, C = 0;
    // lifter misc
    int p0 = -1, p1 = -1, p2 = -1, p3 = -1;  // parameter indices (meaning depends on kind)
    float eps = 0.f;
    int i0 = 0, i1 = 0, i2 = 0, i3 = 0;      // small ints (meaning depends on kind)
    int lvlH[4] = {0, 0, 0, 0}, lvlW[4] = {0, 0, 0, 0}, lvlC[4] = {0, 0, 0, 0};
    int outs[4] = {-1, -1, -1, -1};
    int idxs[4] = {-1, -1, -1, -1};          // OP_EMBED: corner-index tap buffers
    int pq[4] = {-1, -1, -1, -1};   // ... and their quad-interleaved packs
    int pw[4] = {-1, -1, -1, -1}, pb[4] = {-1, -1, -1, -1};   // per-level linear parameters (OP_EMBED feat_embed, OP_CTX_ATTN embed_proj)
    int ln_w = -1, ln_b = -1;                // OP_GEMM rows mode: LayerNorm the A rows on the fly (parameter indices), eps in `eps`
    double flops_per_frame = 0.0;
    int bf16 = 0;                 // tensors of this op are bf16 (conv: bf16 MFMA kernel)
    int wino = 0;                 // 3x3 stride-1 fp32 conv on the Winograd kernel (igemm_wino.hip)
    int pw_pair = 0;              // one of a 64 -> 256 / 256 -> 64 pointwise pair that igemm_f32_pwchain.hip can run as one launch: stays on the fp32
                                  // kernels in every plan (the chained and the two-launch routes are bit-identical; test_pointwise_chain_*)
    int x3_lo = 0, x3_hi = -1;    // batches [x3_lo, x3_hi] at which a split-fp32 tile takes this conv (f32x3_takes; set by build(), empty = never)
    int out_bf16 = 0;             // fp32 stem conv writing bf16 activations
    int h2_exps = -1, h2_role = 0, h2_peer = -1;   // a BasicBlock's conv1 (role 1: writes planes + exponents to buffer h2_exps) / conv2 (role 2: reads them);
                                  // h2_peer = the other op's index: both must run the two-fp16-piece tile at a batch for the pair to use planes
    std::vector<int> chain;       // OP_RES_CHAIN: per block {pack qkv, proj, fc1, fc2, param norm1.weight, .bias, norm2.weight, .bias};
                                  // OP_MLP_CHAIN: {pack fc1, fc2, param norm2.weight, .bias}, rows through amap
    long h2_utab = -1;            // two-fp16-piece conv tile: word offset of this conv's map geometry in the engine's unit tables (-1: none)
    int bneck_c3 = -1;            // conv1 / conv2 / downsample of a first bottleneck that may run as one kernel with its conv3 (that op's index; plan.cpp bneck0_mark)
    int lane = 0;                 // stream lane inside a fork/join region (0 = the caller's stream)
    int region = -1;              // index of the enclosing fork/join region, -1 outside
};

struct NamedTensor {
    int buf;
    int64_t shape[4];   // shape[0] = -1 means "batch"
    int ndim;
    int is_int;
};

// float offsets inside the training region of the workspace (train.cpp :: Engine::train_layout)
struct TrainLayout {
    struct Ctx { size_t xh1, rs1, y1, ao, U[4], xh2, rs2, y2, hp, hg; };
    struct Att { size_t xh1, rs1, y1, qkv, o, xh2, rs2, y2, hp, hg; };
    size_t X, S[4];
    Ctx ctx[4];
    Att res[4], joint[4];
    size_t xhh, rsh, yh;
    size_t dX, gA, gB, gC, cat, dU[4], tA, tB, wT, slabs, red;
    size_t h2w, h2max;                       // the step's weights as two-fp16-piece packs (W and W^T: Engine::t_h2_specs), and their maxima scratch
    size_t slabs_elems = 0, red_elems = 0;   // capacities of the two scratch areas above (what the weight-gradient slicing may use)
    size_t red_cap = 0;                      // ... and one column reduction's partial sums of red_elems (the same arrangement: t_col_flush)
    size_t slab_cap = 0;                     // what ONE weight gradient's slabs may take of slabs_elems (the slab area holds several layers' slabs
                                             // until Engine::t_slab_flush sums them in one launch)
    size_t total;
};

// Per-launch event log of the product schedule (capf_forward_profile): launch k is bracketed by ev[k] and
// ev[k + 1]; leader[k] is the first op of the launch, op_leader[op] the leader of the launch an op rode in.
struct LaunchLog {
    std::vector<hipEvent_t> ev;
    std::vector<int> leader;
    std::vector<int> op_leader;
    std::vector<int> op_variant;       // per leader op: which device kernel a grouped bf16 launch chose (kernels.h), -1 otherwise
    hipError_t mark(hipStream_t s, const int* members, int n) {
        hipEvent_t e;
        hipError_t r = hipEventCreate(&e);
        if (r != hipSuccess) return r;
        ev.push_back(e);
        if (n > 0) {
            leader.push_back(members[0]);
            for (int i = 0; i < n; ++i) op_leader[members[i]] = members[0];
        }
        return hipEventRecord(e, s);
    }
    ~LaunchLog() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
};

struct Engine {
    capf_config cfg{};
    int device = -1;
    std::string err;

    std::vector<Param> params;
    std::map<std::string, int> param_index;
    std::vector<Buffer> bufs;
    std::vector<Pack> packs;
    std::vector<Op> ops;
    int n_backbone_ops = 0;
    int cur_lane = 0, cur_region = -1, n_regions = 0, n_events = 0;
    std::vector<std::pair<int, int>> regions;   // [fork op, join op]
    std::vector<std::vector<std::vector<int>>> region_levels;   // per region: dependency levels -> op indices
    void schedule_regions();
    std::map<std::string, NamedTensor> named;
    size_t ws_elems_per_frame = 0;
    size_t pack_elems = 0;

    float* pack_arena = nullptr;   // device, owned
    size_t bias_tab_off = 0;       // inside the arena: the CopySegment table of the packed linears' bias vectors (launch_copy_segments)
    std::vector<CopySegment> bias_tab;   // its host image (kept alive for the asynchronous upload); rebuilt by every full repack
    bool bias_tab_on_device = false;
    float* split_ws = nullptr;     // device, owned: split-K slabs + per-tile counters of the small-batch conv launches
    int* split_cnt = nullptr;
    // the two-chain schedule (lanes == 3) runs two grouped chains CONCURRENTLY: the side chain's split-K convs get slabs and
    // counters of their own (a conv splits or not by its shape alone, so both chains may hold splitting convs at once)
    float* split_ws_side = nullptr;
    int* split_cnt_side = nullptr;
    bool on_side_chain = false;    // set around run_region_grouped(side[0], ...)
    static constexpr long SPLIT_WS_ELEMS = 4L << 20;
    static constexpr int SPLIT_CNT_ELEMS = 16384;
    float* ws = nullptr;           // device, borrowed
    size_t ws_bytes = 0;
    bool packed = false;
    bool debug = false;            // run the debug-copy ops (capf_set_debug)
    int wino_min_batch = 24;       // below this batch the Winograd-eligible convs the split-fp32 tile does not take run the direct kernel (with
                                   // split-K; batch 16: 4.75 vs 5.86 ms per forward, batch 24: 6.08 vs 6.37)
    // does the conv leave the direct kernel (for a split-fp32 tile or a Winograd kernel: launch_gemm_wino decides which) at this batch?
    // A pure function of the op and the batch: the tile's batch range is precomputed per op by build() (no cache, no mutable state --
    // const-handle queries on other threads may ask while a forward is being enqueued)
    bool wino_now(const Op& op, int batch) const {
        return op.wino && ((batch >= op.x3_lo && batch <= op.x3_hi) || (batch >= wino_min_batch && !packs[op.pack].wino_skip));
    }
    bool wino_f43_cpn = false;
    int wino_f43_min_hw = 0, wino_f43_max_hw = 1 << 30;   // F(4,3) only for maps with min <= H * W <= max pixels
    bool wino_f43 = true;          // plan: F(4,3) where W % 4 == 0, F(2,3) for the other even widths (CAPF_WINO_F43=0: F(2,3) everywhere, A/B runs)
    std::vector<int> last_variants;   // capf_forward_profile_launches: grouped-bf16 kernel variant per leader op
    bool use_rh = true;            // plan: row-halo layout + kernel for the bf16 3x3 stride-1 convs (CAPF_BF16_RH=0: off, A/B runs)
    bool use_x3 = true;            // plan: split-fp32 tile for the Winograd-eligible fp32 3x3 stride-1 convs (plan_flags & CAPF_PLAN_NO_F32X3 clears it)
    bool use_h2g = true;           // plan: every other fp32 conv / linear of an inference batch >= 5 on the two-fp16-piece GEMM (igemm_f32h2.hip; plan_flags &
                                   // CAPF_PLAN_NO_F32H2_GEMM clears it)
    static constexpr int H2G_MIN_BATCH = 5;
    bool h2g_lifter_dirty = true;  // the linear packs' two-piece copies are stale (parameters changed since they were packed)
    int ensure_h2g_lifter(hipStream_t s);
    bool x3_h2 = true;             // ... the two-fp16-piece tile (three piece products per MAC); plan_flags & CAPF_PLAN_F32X3_EXACT: the three-bf16-piece tile (six)
    bool use_ws = true;            // plan: 2-D halo layout + kernel for the bf16 3x3 stride-1 convs (plan_flags & CAPF_PLAN_NO_WS clears it)
    bool use_wino = true;          // plan: Winograd F(2,3) kernel for the eligible 3x3 stride-1 fp32 convs (CAPF_WINO=0: direct kernel everywhere, A/B runs)
    bool fused_lifter = true;      // plan: fused embed / context-attention kernels + LayerNorm folded into the GEMMs (CAPF_LIFTER_FUSED=0: the one-kernel-per-op plan, for A/B runs)
    int lanes = 3;                 // fork/join regions: 0 in program order, 1 one side stream per lane, 2 grouped launches on one stream,
                                   // 3 grouped launches as two chains on two streams (capf_set_lanes)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> events;
    int last_batch = 0;
    const float* images = nullptr;
    const float* k2d = nullptr;
    float* kcrop = nullptr;
    float* out = nullptr;

    // ---- schema / plan construction (plan.cpp)
    int add_param(const std::string& name, int kind, std::initializer_list<int64_t> shape);
    int new_buffer(size_t elems, const std::string& tag);
    Tensor conv_bn(const std::string& conv, const std::string& bn, const Tensor& x, int Cout, int ks, int stride,
                   int act, const Tensor* residual);
    void build_hrnet(Tensor img, Tensor feats[4]);
    void build_cpn(Tensor img, Tensor feats[4]);
    void build_lifter(const Tensor feats[4]);
    bool build();
    void assign_offsets();
    bool bf16() const { return cfg.compute_dtype == CAPF_BF16; }
    size_t act_elems(size_t n) const { return bf16() ? (n + 1) / 2 : n; }   // backbone activation size in float slots
    void use(int buf);   // mark buffer as read by the op being appended
    void push(Op op);    // append an op, tagging it with the current lane / region
    void fork(int n);    // open a region of n independent lanes (independent branches run on side streams)
    void set_lane(int l) { cur_lane = l; }
    void join();

    // ---- training step (train.cpp)
    int feat_buf[4] = {-1, -1, -1, -1}, feat_H[4] = {0, 0, 0, 0}, feat_W[4] = {0, 0, 0, 0}, feat_C[4] = {0, 0, 0, 0};
    std::vector<int> ctx_ao_pack;        // pack index of [attention_weights | sampling_offsets] per context block
    std::vector<int> ctx_tap_pos, ctx_tap_idx;   // per context block: debug tap buffers (sampling positions, NW corner indices)
    std::vector<long> grad_off;          // per parameter: offset in the flat lifter gradient, -1 for the backbone
    long grad_elems = 0;
    int train_batch = 0;                 // batch of the forward_train whose activations are still in the workspace (0: none)
    int64_t train_generation = 0;        // bumped by every run that (over)writes the workspace
    void invalidate_train() { train_batch = 0; ++train_generation; t_h2_base = nullptr; }
    int batch_limit = 0;                 // largest batch the 32-bit tensor addressing of the kernels allows (build())
    void train_layout(int B, TrainLayout& L) const;
    size_t train_elems(int B) const;
    int forward_train(hipStream_t s, int B, const float* masks);
    int backward(hipStream_t s, int B, const float* dOut, float* flat_grad, const float* masks);
    int t_gemm(hipStream_t s, const float* A, RowMap amap, int M, int N, int K, const float* W, int Kpad, const float* bias,
               float* out, RowMap omap, const float* res, RowMap rmap, int act, const float* rscale, int rs_div, const float* Wh2 = nullptr);
    // The lifter's linears on the two-fp16-piece GEMM during a training step (forward: y = x W^T, backward: dX = dY W): which matrices,
    // where their packs sit in the training region (offsets from TrainLayout::h2w), and the table the pack launch reads (kernels.h)
    struct H2TrainSpec { int param, pack; int N, K, ld; bool fwd, bwd; };   // source: parameter `param`, or (param < 0) the row pack `pack`
    std::vector<H2TrainSpec> t_h2_specs;
    std::vector<H2TrainW> t_h2_tab;      // host image of the device table (offsets filled by t_h2_plan, pointers by t_h2_prepare)
    std::map<const float*, int> t_h2_index;
    size_t t_h2_tab_off = 0;             // inside the pack arena
    size_t t_h2_elems = 0, t_h2_max_elems = 0;
    int t_h2_tiles = 0;
    bool t_h2_on_device = false;
    // weight-gradient slabs waiting for their sum (backward): the slab area is a bump allocator, the sums run as ONE launch when it is
    // full, when SLAB_BATCH_MAX jobs wait, and at the end of backward -- 64 slab_sum launches per step -> a handful
    SlabBatch t_slab_jobs{};
    size_t t_slab_cur = 0;
    float* t_slab_take(hipStream_t s, float* area, size_t cap, size_t elems, int* rc);
    int t_slab_defer(hipStream_t s, const float* slabs, int nslab, long n, float* dst);
    int t_slab_flush(hipStream_t s);
    // the same for the second stages of the backward's column reductions (bias / LayerNorm gradients): 32 launches per step -> one or two
    ColFinalBatch t_col_jobs{};
    size_t t_col_cur = 0;
    int t_colreduce(hipStream_t s, const TrainLayout& L, float* tw, const float* A, RowMap amap, const float* Bm, RowMap bmap, int bmode,
                    int rows, int C, float* dst, long dst_stride, float* dst2, size_t cap_elems);
    int t_col_flush(hipStream_t s);
    float* t_h2_base = nullptr;          // this step's packs (set by t_h2_prepare; nullptr: the step runs on the fp32 matrix pipe)
    void t_h2_plan();
    int t_h2_prepare(hipStream_t s, const TrainLayout& L, float* tw, int B);
    const float* t_h2_pack(const float* W, bool transposed) const;
    int t_linear_bwd(hipStream_t s, const TrainLayout& L, float* tw, const float* dY, RowMap dymap, int rows, int N, int K,
                     const float* Xin, RowMap xmap, const float* W, float* dX, RowMap dxmap, bool acc_dx, float* gW, float* gb);

    // ---- execution (engine.cpp)
    float* bptr(int buf, int batch) const { return ws + bufs[buf].offset * (size_t)batch; }
    int repack(hipStream_t s, bool lifter_only = false);
    int run(hipStream_t s, int batch, int first_op, int last_op, hipEvent_t* ev = nullptr, LaunchLog* log = nullptr);
    int exec_op(const Op& op, hipStream_t s, int batch);
    FuseSumArgs fuse_args(const Op& op, int batch) const;
    // op i and op i + 1 are a 64 -> 256 / 256 -> 64 pointwise conv pair that runs as ONE launch at this batch (igemm_f32_pwchain.hip)
    bool pwchain_head(int i, int batch, int last_op) const;
    bool use_pwchain = true;       // plan_flags & CAPF_PLAN_NO_PWCHAIN clears it
    bool has_res_chain = false;    // the plan holds an OP_RES_CHAIN (lifter_chain.hip): its two-piece packs are needed at every batch
    std::vector<unsigned> utab_host;     // the unit tables of the two-fp16-piece conv tile, one per map geometry (H, W, Cin) of the plan (build())
    size_t utab_off = 0;                 // ... and their place in the pack arena (uploaded once: they depend on the plan alone)
    bool utab_on_device = false;
    bool use_bneck = true;         // plan_flags & CAPF_PLAN_NO_BNECK clears it
    bool batch_reduce = true;      // plan_flags & CAPF_PLAN_NO_BATCHED_REDUCE clears it: the backward's second-stage reductions one launch each
    // op i opens the fork / join region of a first bottleneck (conv1, conv2 | downsample) directly followed by its conv3, and the block runs as
    // ONE launch at this batch (bneck_bf16.hip); m = {conv1, conv2, downsample, conv3}
    bool bneck0_head(int i, int batch, int last_op, int m[4]) const;
    int bneck0_member(int i, int batch) const;       // op i rides in such a launch: index of its fork op, -1 otherwise
    // ops i, i + 1, i + 2 are an identity bottleneck (conv1 256 -> 64, conv2 3x3, conv3 64 -> 256 + conv1's input) that runs as ONE launch at this batch
    bool bneck1_head(int i, int batch, int last_op) const;
    int bneck1_member(int i, int batch) const;       // op i rides in such a launch: index of its conv1, -1 otherwise
    bool use_upadd = true;         // plan_flags & CAPF_PLAN_NO_UPADD clears it (CPN bf16: lateral conv + upsampled add in one launch)
    int run_region_grouped(hipStream_t s, int batch, int region, LaunchLog* log, unsigned lane_mask = ~0u);
    GemmArgs gemm_args(const Op& op, int batch, bool planes = true) const;
    bool use_h2_planes = false;    // plan_flags & CAPF_PLAN_H2_PLANES sets it (opt-in: measured slower, EXPERIMENTS R6.5)
};

}  // namespace capf

struct capf_handle {
    capf::Engine e;
};
