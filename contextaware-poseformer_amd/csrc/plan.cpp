// Static layer plan + parameter schema of the hot path.
//
// Restates, as a flat list of kernel launches over NHWC buffers, the module graphs of
//   PoseHighResolutionNet   ContextPose/mvn/models/pose_hrnet.py:312-501
//   CPN50                   ContextPose/mvn/models/networks/{network,resnet,globalNet,refineNet}.py
//   PoseTransformer         ContextPose/mvn/models/pose_dformer.py:144-241
// and registers every state_dict entry under the reference's name, so that
// released checkpoints load strict=True into the host module built from this schema.
#include <limits.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <tuple>

#include "engine.h"

namespace capf {

static const int EXT_IMAGES = -2;

static size_t round64(size_t n) { return (n + 63) / 64 * 64; }
static int round32(int n) { return (n + 31) / 32 * 32; }
static int round64(int n) { return (n + 63) / 64 * 64; }

int Engine::add_param(const std::string& name, int kind, std::initializer_list<int64_t> shape) {
    auto it = param_index.find(name);
    if (it != param_index.end()) return it->second;
    Param p;
    p.name = name;
    p.kind = kind;
    p.ndim = (int)shape.size();
    int i = 0;
    for (int64_t s : shape) p.shape[i++] = s;
    params.push_back(p);
    param_index[name] = (int)params.size() - 1;
    return (int)params.size() - 1;
}

int Engine::new_buffer(size_t elems, const std::string& tag) {
    Buffer b;
    b.elems = round64(elems);
    b.def_op = (int)ops.size();
    b.last_op = (int)ops.size();
    b.tag = tag;
    bufs.push_back(b);
    return (int)bufs.size() - 1;
}

void Engine::use(int buf) {
    if (buf >= 0) bufs[buf].last_op = std::max(bufs[buf].last_op, (int)ops.size());
}

void Engine::push(Op op) {
    op.lane = cur_lane;
    op.region = cur_region;
    ops.push_back(op);
}

void Engine::fork(int n) {
    Op op;
    op.kind = OP_FORK;
    op.name = "fork";
    op.i0 = n;
    op.i1 = n_events;          // first event id of this region: 1 for the fork + (n-1) for the join
    n_events += n;
    cur_region = n_regions++;
    cur_lane = 0;
    regions.push_back({(int)ops.size(), -1});
    push(op);
}

void Engine::join() {
    Op op;
    op.kind = OP_JOIN;
    op.name = "join";
    const Op& f = ops[regions[cur_region].first];
    op.i0 = f.i0;
    op.i1 = f.i1;
    cur_lane = 0;
    regions[cur_region].second = (int)ops.size();
    push(op);
    cur_region = -1;
}

static void keep(Engine& e, int buf) { e.bufs[buf].last_op = INT_MAX; }

static void name_tensor(Engine& e, const std::string& name, int buf, std::initializer_list<int64_t> shape,
                        int is_int = 0) {
    NamedTensor t{};
    t.buf = buf;
    t.ndim = (int)shape.size();
    int i = 0;
    for (int64_t s : shape) t.shape[i++] = s;
    t.is_int = is_int;
    e.named[name] = t;
    keep(e, buf);
}

// conv (bias=False) + eval BatchNorm (+ residual) (+ ReLU) as ONE implicit-GEMM launch.
Tensor Engine::conv_bn(const std::string& conv, const std::string& bn, const Tensor& x, int Cout, int ks,
                       int stride, int act, const Tensor* residual) {
    const int pad = ks / 2;
    Tensor y;
    y.H = (x.H + 2 * pad - ks) / stride + 1;
    y.W = (x.W + 2 * pad - ks) / stride + 1;
    y.C = Cout;

    Pack pk;
    pk.kind = 0;
    pk.w[0] = add_param(conv + ".weight", CAPF_P_CONV_W, {Cout, x.C, ks, ks});
    pk.bn_g = add_param(bn + ".weight", CAPF_P_BN_W, {Cout});
    pk.bn_b = add_param(bn + ".bias", CAPF_P_BN_B, {Cout});
    pk.bn_m = add_param(bn + ".running_mean", CAPF_P_BN_MEAN, {Cout});
    pk.bn_v = add_param(bn + ".running_var", CAPF_P_BN_VAR, {Cout});
    add_param(bn + ".num_batches_tracked", CAPF_P_BN_NBT, {});
    pk.N = Cout;
    pk.Cin = x.C;
    pk.ks = ks;
    pk.K = ks * ks * x.C;
    const bool use_bf16 = bf16() && x.C % 8 == 0;       // the Cin = 3 stem stays on the fp32 kernel
    pk.bf16 = use_bf16;
    pk.Kpad = use_bf16 ? round64(pk.K) : round32(pk.K);
    // Winograd F(2,3) along W (igemm_wino.hip) for the 3x3 stride-1 fp32 convs: 1.5x fewer MFMAs; block tiles of 64 x 64,
    // 64 x 32 (32-channel outputs) or 32 x 64 (few tiles) are chosen per problem at launch
    const bool use_wino = this->use_wino && !use_bf16 && ks == 3 && stride == 1 && x.C % 32 == 0 && x.W % 2 == 0 && Cout % 32 == 0;
    pk.wino = use_wino;
    // F(4,3) (half the MFMAs of the direct conv) where the row length allows four-pixel tiles, F(2,3) (two thirds) otherwise.
    // HRNet only: F(4,3) pays inside the grouped multi-branch launches (+3.2 % end to end), not for CPN's lone convs (-0.8 %).
    if (use_wino) { pk.Kpad2 = pk.Kpad; pk.Kpad = ((x.W % 4 == 0 && wino_f43 && (cfg.backbone == CAPF_HRNET || wino_f43_cpn) && x.H * x.W >= wino_f43_min_hw && x.H * x.W <= wino_f43_max_hw) ? 18 : 12) * x.C; }
    // bf16: the 3x3 stride-1 convs also keep their weights in the row-halo layout (igemm_bf16.hip: one staged activation tile
    // for the three kw taps); launches of >= 2048 tiles run that kernel, smaller ones the ring kernel on the standard layout
    if (use_bf16 && use_rh && ks == 3 && stride == 1 && bf16_rh_width(x.C) && Cout % 4 == 0) { pk.rh = true; pk.Kpad2 = 9 * x.C; }
    // ... and in the layout of the 2-D halo tile (igemm_bf16_ws.hip), which takes them from 512 tiles per launch
    if (use_bf16 && use_ws && ks == 3 && stride == 1 && x.C % 16 == 0 && Cout % 8 == 0) pk.ws = true;
    // fp32: the Winograd-eligible convs also keep their weights as three bf16 pieces for the split-fp32 tile (igemm_f32x3_ws.hip), which
    // takes them from 370 MFLOP per conv and batch 5 up (f32x3_takes)
    if (use_wino && use_x3 && x.W <= 256) pk.x3 = true;
    // fp32: every conv with 16-byte-aligned channel counts also keeps its weights as two block-scaled fp16 pieces in the direct layout's
    // geometry (igemm_f32h2.hip): what runs on the plain fp32 MFMA kernel at batch < 5 runs there from batch 5 (1x1 / stride-2 fuse and
    // transition convs, lone convs; the HBM-bound pointwise kernels of layer1 keep theirs)
    if (!use_bf16 && use_h2g && x.C % 4 == 0 && Cout % 4 == 0) { pk.h2g = true; pk.KpadH = use_wino ? pk.Kpad2 : pk.Kpad; }
    packs.push_back(pk);

    Op op;
    op.kind = OP_GEMM;
    op.name = conv;
    op.conv = 1;
    op.pack = (int)packs.size() - 1;
    op.in[0] = x.buf;
    op.Cin = x.C; op.H = x.H; op.W = x.W; op.Ho = y.H; op.Wo = y.W;
    op.ks = ks; op.stride = stride; op.pad = pad;
    op.rows_per_frame = (long)y.H * y.W;
    op.N = Cout; op.K = pk.K; op.act = act;
    op.omap = row_ld(Cout);
    op.rmap = row_ld(Cout);
    op.flops_per_frame = 2.0 * y.H * y.W * (double)Cout * pk.K;
    use(x.buf);
    if (residual) {
        op.aux = residual->buf;
        use(residual->buf);
    }
    op.bf16 = use_bf16 ? 1 : 0;
    op.wino = use_wino ? 1 : 0;
    op.out_bf16 = (bf16() && !use_bf16) ? 1 : 0;
    y.buf = new_buffer(act_elems((size_t)y.H * y.W * Cout), conv);
    op.out = y.buf;
    push(op);
    return y;
}

static Tensor fuse_sum(Engine& e, const std::string& name, const Tensor* terms, const int* shifts, int n,
                       const Tensor& like, int relu) {
    Op op;
    op.kind = OP_FUSE;
    op.name = name;
    op.n_in = n;
    for (int i = 0; i < n; ++i) {
        op.in[i] = terms[i].buf;
        op.shift[i] = shifts[i];
        e.use(terms[i].buf);
    }
    op.H = like.H; op.W = like.W; op.C = like.C; op.relu = relu;
    op.bf16 = e.bf16() ? 1 : 0;
    Tensor y = like;
    y.buf = e.new_buffer(e.act_elems((size_t)like.H * like.W * like.C), name);
    op.out = y.buf;
    e.push(op);
    return y;
}

// ---------------------------------------------------------------------------------------------------
// HRNet (pose_hrnet.py)
// ---------------------------------------------------------------------------------------------------
static Tensor hr_basic_block(Engine& e, const std::string& p, const Tensor& x) {   // :66-95
    Tensor y = e.conv_bn(p + ".conv1", p + ".bn1", x, x.C, 3, 1, ACT_RELU, nullptr);
    const int i1 = (int)e.ops.size() - 1;
    Tensor out = e.conv_bn(p + ".conv2", p + ".bn2", y, x.C, 3, 1, ACT_RELU, &x);
    const int i2 = (int)e.ops.size() - 1;
    // conv1's output has ONE reader, conv2, and both convs have the same tiles: where both run the two-fp16-piece tile (a matter of the
    // batch, decided per launch in gemm_args) it travels as split fp16 planes + one scale exponent per (tile, 16-channel chunk) --
    // igemm_f32h2_ws_tile.h PLANES: the consumer's K loop loses its maximum / split / scale-exchange phase.  Same bytes, same buffer.
    if (e.use_h2_planes && e.x3_h2 && !e.bf16() && x.C % 16 == 0 && e.packs[e.ops[i1].pack].x3 && e.packs[e.ops[i2].pack].x3) {
        const int tiles_pf = f32h2_tiles_m(1, x.H, x.W);
        if (tiles_pf > 0) {
            const int b = e.new_buffer((size_t)tiles_pf * (x.C / 16) + 16, p + ".conv1.exps");
            e.bufs[b].def_op = i1;
            e.bufs[b].last_op = i2;
            e.ops[i1].h2_exps = b; e.ops[i1].h2_role = 1; e.ops[i1].h2_peer = i2;
            e.ops[i2].h2_exps = b; e.ops[i2].h2_role = 2; e.ops[i2].h2_peer = i1;
        }
    }
    return out;
}

// A first bottleneck whose five ops may run as ONE kernel (bneck_bf16.hip; Engine::bneck0_head decides per batch): that kernel reads x while it
// writes y and -- in the variant the layer-wise tests run -- conv1's, conv2's and the shortcut's outputs too, so all five tensors stay alive
// (no two of them share workspace) from conv1 to conv3; the ops remember conv3 (capf_op_describe: their checkpoint is the whole block).
// `fork` = index of the region's fork op: fork, conv1, conv2, downsample, join, conv3.
static void bneck0_mark(Engine& e, int fork) {
    if (!e.use_bneck || !e.bf16() || fork + 5 >= (int)e.ops.size()) return;
    const int c1 = fork + 1, c2 = fork + 2, ds = fork + 3, c3 = fork + 5;
    const Op &o1 = e.ops[c1], &o3 = e.ops[c3];
    if (e.ops[fork].kind != OP_FORK || e.ops[fork + 4].kind != OP_JOIN || o1.Cin != 64 || o1.N != 64 || o3.N != 256 || e.ops[c2].stride != 1 || e.ops[c2].ks != 3) return;
    for (int b : {o1.in[0], o1.out, e.ops[c2].out, e.ops[ds].out, o3.out}) {
        if (b < 0) continue;
        e.bufs[b].def_op = std::min(e.bufs[b].def_op, c1);
        e.bufs[b].last_op = std::max(e.bufs[b].last_op, c3);
    }
    for (int k : {c1, c2, ds}) e.ops[k].bneck_c3 = c3;
}

// ... and an identity bottleneck (conv1 at `c1`, conv2, conv3 + x): x, conv1's and conv2's outputs and y alive from conv1 to conv3
static void bneck1_mark(Engine& e, int c1) {
    if (!e.use_bneck || !e.bf16() || c1 + 2 >= (int)e.ops.size()) return;
    const Op &o1 = e.ops[c1], &o2 = e.ops[c1 + 1], &o3 = e.ops[c1 + 2];
    if (o1.kind != OP_GEMM || o2.kind != OP_GEMM || o3.kind != OP_GEMM || o1.Cin != 256 || o1.N != 64 || o2.ks != 3 || o2.stride != 1 || o3.N != 256 || o3.aux != o1.in[0]) return;
    for (int b : {o1.in[0], o1.out, o2.out, o3.out}) {
        if (b < 0) continue;
        e.bufs[b].def_op = std::min(e.bufs[b].def_op, c1);
        e.bufs[b].last_op = std::max(e.bufs[b].last_op, c1 + 2);
    }
    e.ops[c1].bneck_c3 = c1 + 2;
    e.ops[c1 + 1].bneck_c3 = c1 + 2;
}

static Tensor hr_bottleneck(Engine& e, const std::string& p, const Tensor& x, int planes, bool down) {  // :98-136
    const int fork_at = (int)e.ops.size();
    if (down) e.fork(2);             // the projection shortcut only reads x: independent of conv1 / conv2
    Tensor y = e.conv_bn(p + ".conv1", p + ".bn1", x, planes, 1, 1, ACT_RELU, nullptr);
    y = e.conv_bn(p + ".conv2", p + ".bn2", y, planes, 3, 1, ACT_RELU, nullptr);
    Tensor r = x;
    if (down) {
        e.set_lane(1);
        r = e.conv_bn(p + ".downsample.0", p + ".downsample.1", x, planes * 4, 1, 1, ACT_NONE, nullptr);
        e.join();
    }
    Tensor out = e.conv_bn(p + ".conv3", p + ".bn3", y, planes * 4, 1, 1, ACT_RELU, &r);
    if (down) bneck0_mark(e, fork_at);
    else bneck1_mark(e, fork_at);
    return out;
}

// HighResolutionModule (:139-303).  xs: in/out; branch_out (optional) receives the branch outputs.
static void hr_module(Engine& e, const std::string& p, std::vector<Tensor>& xs, int n_out, int blocks,
                      std::vector<Tensor>* branch_out) {
    const int nb = (int)xs.size();
    std::vector<Tensor> br(nb);
    e.fork(nb);                      // the branches are independent until the fuse: one stream lane each
    for (int i = 0; i < nb; ++i) {
        e.set_lane(i);
        Tensor y = xs[i];
        for (int k = 0; k < blocks; ++k)
            y = hr_basic_block(e, p + ".branches." + std::to_string(i) + "." + std::to_string(k), y);
        br[i] = y;
    }
    e.join();
    if (branch_out) *branch_out = br;
    std::vector<Tensor> outs(n_out);
    e.fork(n_out);                   // each fused output only reads the branch outputs: independent lanes again
    for (int i = 0; i < n_out; ++i) {
        e.set_lane(i);
        Tensor terms[4];
        int shifts[4] = {0, 0, 0, 0};
        for (int j = 0; j < nb; ++j) {
            const std::string fp = p + ".fuse_layers." + std::to_string(i) + "." + std::to_string(j);
            if (j == i) {
                terms[j] = br[j];
            } else if (j > i) {   // 1x1 conv + BN at low resolution; nearest x2^(j-i) applied by the fuse kernel
                terms[j] = e.conv_bn(fp + ".0", fp + ".1", br[j], br[i].C, 1, 1, ACT_NONE, nullptr);
                shifts[j] = j - i;
            } else {              // (i-j) stride-2 3x3 convs, ReLU on all but the last
                Tensor t = br[j];
                for (int k = 0; k < i - j; ++k) {
                    const bool last = (k == i - j - 1);
                    const std::string cp = fp + "." + std::to_string(k);
                    t = e.conv_bn(cp + ".0", cp + ".1", t, last ? br[i].C : br[j].C, 3, 2,
                                  last ? ACT_NONE : ACT_RELU, nullptr);
                }
                terms[j] = t;
            }
        }
        outs[i] = fuse_sum(e, p + ".fuse" + std::to_string(i), terms, shifts, nb, br[i], 1);
    }
    e.join();
    xs = outs;
}

void Engine::build_hrnet(Tensor img, Tensor feats[4]) {
    const std::string B = "backbone";
    Tensor x = conv_bn(B + ".conv1", B + ".bn1", img, 64, 3, 2, ACT_RELU, nullptr);     // :465-467
    x = conv_bn(B + ".conv2", B + ".bn2", x, 64, 3, 2, ACT_RELU, nullptr);              // :468-470
    for (int k = 0; k < 4; ++k) x = hr_bottleneck(*this, B + ".layer1." + std::to_string(k), x, 64, k == 0);

    std::vector<Tensor> ys = {x};
    std::vector<int> pre_ch = {256};
    for (int stage = 2; stage <= 4; ++stage) {
        const int nb = stage;
        // transition (:377-411)
        std::vector<Tensor> xs(nb);
        const std::string tp = B + ".transition" + std::to_string(stage - 1) + ".";
        const int npre = (int)pre_ch.size();
        fork(nb);                    // every transition path reads the previous stage's outputs only
        for (int i = 0; i < nb; ++i) {
            set_lane(std::min(i, 3));
            const int ch = cfg.hr_channels[i];
            if (i < npre) {
                if (ch != pre_ch[i])
                    xs[i] = conv_bn(tp + std::to_string(i) + ".0", tp + std::to_string(i) + ".1", ys[i], ch, 3, 1,
                                    ACT_RELU, nullptr);
                else
                    xs[i] = ys[i];
            } else {
                Tensor t = ys[npre - 1];
                for (int j = 0; j < i + 1 - npre; ++j) {
                    const int oc = (j == i - npre) ? ch : pre_ch[npre - 1];
                    const std::string cp = tp + std::to_string(i) + "." + std::to_string(j);
                    t = conv_bn(cp + ".0", cp + ".1", t, oc, 3, 2, ACT_RELU, nullptr);
                }
                xs[i] = t;
            }
        }
        join();
        const int nmod = cfg.hr_modules[stage - 2];
        for (int m = 0; m < nmod; ++m) {
            const bool last = (stage == 4 && m == nmod - 1);          // multi_scale_output=False (:359-360)
            std::vector<Tensor> br;
            hr_module(*this, B + ".stage" + std::to_string(stage) + "." + std::to_string(m), xs, last ? 1 : nb,
                      cfg.hr_blocks, (stage == 4 && m == 0) ? &br : nullptr);
            if (stage == 4 && m == 0) {
                // forward() returns x_list[1..3], which stage4[0] overwrote in place with its BRANCH
                // outputs (:289-290, :501).
                for (int i = 1; i < 4; ++i) feats[i] = br[i];
            }
        }
        ys = xs;
        pre_ch.assign(cfg.hr_channels, cfg.hr_channels + nb);
    }
    feats[0] = ys[0];
}

// ---------------------------------------------------------------------------------------------------
// CPN-50 (networks/)
// ---------------------------------------------------------------------------------------------------
static Tensor pool_or_resize(Engine& e, OpKind kind, const std::string& name, const Tensor& x, int Ho, int Wo,
                             const Tensor* add = nullptr) {
    Op op;
    op.kind = kind;
    op.name = name;
    op.in[0] = x.buf;
    e.use(x.buf);
    if (add) { op.aux = add->buf; e.use(add->buf); }           // resize only: out = resize(x) + add
    op.H = x.H; op.W = x.W; op.C = x.C; op.Ho = Ho; op.Wo = Wo;
    op.bf16 = e.bf16() ? 1 : 0;
    Tensor y{-1, Ho, Wo, x.C};
    y.buf = e.new_buffer(e.act_elems((size_t)Ho * Wo * x.C), name);
    op.out = y.buf;
    e.push(op);
    return y;
}

static void register_unused_conv_bn(Engine& e, const std::string& conv, const std::string& bn, int Cout, int Cin,
                                    int ks) {
    e.add_param(conv + ".weight", CAPF_P_CONV_W, {Cout, Cin, ks, ks});
    e.add_param(bn + ".weight", CAPF_P_BN_W, {Cout});
    e.add_param(bn + ".bias", CAPF_P_BN_B, {Cout});
    e.add_param(bn + ".running_mean", CAPF_P_BN_MEAN, {Cout});
    e.add_param(bn + ".running_var", CAPF_P_BN_VAR, {Cout});
    e.add_param(bn + ".num_batches_tracked", CAPF_P_BN_NBT, {});
}

void Engine::build_cpn(Tensor img, Tensor feats[4]) {
    const std::string R = "backbone.resnet";
    Tensor x = conv_bn(R + ".conv1", R + ".bn1", img, 64, 7, 2, ACT_RELU, nullptr);       // resnet.py:137-139
    x = pool_or_resize(*this, OP_MAXPOOL, R + ".maxpool", x, (x.H + 2 - 3) / 2 + 1, (x.W + 2 - 3) / 2 + 1);
    const int nblk[4] = {3, 4, 6, 3}, planes[4] = {64, 128, 256, 512}, strides[4] = {1, 2, 2, 2};
    Tensor c[4];
    for (int li = 0; li < 4; ++li) {
        for (int k = 0; k < nblk[li]; ++k) {                                               // :58-93, :119-133
            const std::string p = R + ".layer" + std::to_string(li + 1) + "." + std::to_string(k);
            const int st = (k == 0) ? strides[li] : 1;
            const int fork_at = (int)ops.size();
            if (k == 0) fork(2);         // projection shortcut: independent of conv1 / conv2
            Tensor y = conv_bn(p + ".conv1", p + ".bn1", x, planes[li], 1, 1, ACT_RELU, nullptr);
            y = conv_bn(p + ".conv2", p + ".bn2", y, planes[li], 3, st, ACT_RELU, nullptr);
            Tensor r = x;
            if (k == 0) {
                set_lane(1);
                r = conv_bn(p + ".downsample.0", p + ".downsample.1", x, planes[li] * 4, 1, st, ACT_NONE, nullptr);
                join();
            }
            x = conv_bn(p + ".conv3", p + ".bn3", y, planes[li] * 4, 1, 1, ACT_RELU, &r);
            if (k == 0 && st == 1) bneck0_mark(*this, fork_at);
            if (k > 0) bneck1_mark(*this, fork_at);
        }
        c[li] = x;
    }
    // globalNet.forward (globalNet.py:61-83); res_out = [x4,x3,x2,x1]
    const std::string G = "backbone.global_net";
    Tensor fms[4];
    for (int i = 0; i < 4; ++i) {
        const Tensor& src = c[3 - i];
        const std::string lp = G + ".laterals." + std::to_string(i);
        if (i > 0 && bf16() && use_upadd) {
            // bf16: the low-resolution conv of the upsampled path first, then the lateral conv with `+ bilinear_x2(t)` behind its ReLU in the
            // epilogue (igemm_bf16_kernel<.., UPADD>): no resize-add launch, the lateral map never travels to HBM and back on its own
            // (lateral: write + read of 453 MB at the largest level and batch 128), and it meets the upsampled term in fp32
            const std::string upn = G + ".upsamples." + std::to_string(i - 1);
            Tensor t = conv_bn(upn + ".1", upn + ".2", fms[i - 1], 256, 1, 1, ACT_NONE, nullptr);
            ops.back().flops_per_frame *= 4.0;
            fms[i] = conv_bn(lp + ".0", lp + ".1", src, 256, 1, 1, ACT_RELU, nullptr);
            Op& lop = ops.back();
            lop.in[1] = t.buf;                       // (OP_GEMM conv: in[1] = the map added, upsampled, behind the activation; i0 x i1 its size)
            lop.i0 = t.H; lop.i1 = t.W;
            bufs[t.buf].last_op = std::max(bufs[t.buf].last_op, (int)ops.size() - 1);
            const std::string pp = G + ".predict." + std::to_string(i);
            register_unused_conv_bn(*this, pp + ".0", pp + ".1", 256, 256, 1);
            register_unused_conv_bn(*this, pp + ".3", pp + ".5", 17, 256, 3);
            continue;
        }
        Tensor lat = conv_bn(lp + ".0", lp + ".1", src, 256, 1, 1, ACT_RELU, nullptr);
        if (i == 0) {
            fms[i] = lat;
        } else {
            // reference: up = BN(conv1x1(bilinear x2(feature_{i-1}))), feature_i = lateral_i + up (globalNet.py:40-46, :66).  A
            // 1x1 conv without bias + eval-mode BN is a per-pixel affine map and bilinear interpolation is a convex combination
            // of pixels, so the two COMMUTE exactly in real arithmetic: the conv runs on the LOW-resolution map (a quarter of
            // the pixels: 116 -> 29 GFLOP and 300 -> 75 us for the largest level at batch 128), the resize adds the lateral.
            const std::string upn = G + ".upsamples." + std::to_string(i - 1);
            Tensor t = conv_bn(upn + ".1", upn + ".2", fms[i - 1], 256, 1, 1, ACT_NONE, nullptr);
            ops.back().flops_per_frame *= 4.0;       // ALGORITHMIC FLOPs stay the reference's (conv at the upsampled resolution,
                                                     // SURVEY 8d); capf_op_executed_flops reports what runs (rows_per_frame)
            fms[i] = pool_or_resize(*this, OP_RESIZE, upn + ".0", t, t.H * 2, t.W * 2, &lat);
        }
        // predict heads: computed-then-discarded by the reference (:71) -> parameters only
        const std::string pp = G + ".predict." + std::to_string(i);
        register_unused_conv_bn(*this, pp + ".0", pp + ".1", 256, 256, 1);
        register_unused_conv_bn(*this, pp + ".3", pp + ".5", 17, 256, 3);
    }
    // refineNet.forward (refineNet.py:72-88)
    const std::string F = "backbone.refine_net";
    const int oh = 64, ow = 48;   // cpn/test_config.py:24 output_shape
    fork(4);                      // the four cascades are independent
    for (int i = 0; i < 4; ++i) {
        set_lane(i);
        Tensor y = fms[i];
        for (int k = 0; k < 3 - i; ++k) {
            const std::string p = F + ".cascade." + std::to_string(i) + "." + std::to_string(k);
            Tensor t = conv_bn(p + ".conv1", p + ".bn1", y, 128, 1, 1, ACT_RELU, nullptr);
            t = conv_bn(p + ".conv2", p + ".bn2", t, 128, 3, 1, ACT_RELU, nullptr);
            Tensor r = conv_bn(p + ".downsample.0", p + ".downsample.1", y, 256, 1, 1, ACT_NONE, nullptr);
            y = conv_bn(p + ".conv3", p + ".bn3", t, 256, 1, 1, ACT_RELU, &r);
        }
        // cascade 3 is nn.Upsample alone (refineNet.py:58-64 with num = 0) on a map that already has output_shape: under align_corners the
        // source index of output pixel i is i * (H - 1) / (H - 1) = i with weight 1 (ATen: lambda 0 -> 1 * x[i] + 0 * x[i + 1]), the identity for
        // finite values -- the level IS the lateral map; no launch, no second copy of it (201 MB written + read back at batch 128)
        if (y.H == oh && y.W == ow) { feats[i] = y; continue; }
        feats[i] = pool_or_resize(*this, OP_RESIZE, F + ".cascade." + std::to_string(i) + ".resize", y, oh, ow);
    }
    join();
    // final_predict is never called (:76-88): parameters only
    const std::string fp = F + ".final_predict";
    register_unused_conv_bn(*this, fp + ".0.conv1", fp + ".0.bn1", 128, 1024, 1);
    register_unused_conv_bn(*this, fp + ".0.conv2", fp + ".0.bn2", 128, 128, 3);
    register_unused_conv_bn(*this, fp + ".0.conv3", fp + ".0.bn3", 256, 128, 1);
    register_unused_conv_bn(*this, fp + ".0.downsample.0", fp + ".0.downsample.1", 256, 1024, 1);
    register_unused_conv_bn(*this, fp + ".1", fp + ".2", 17, 256, 3);
}

// ---------------------------------------------------------------------------------------------------
// lifter (pose_dformer.py)
// ---------------------------------------------------------------------------------------------------
static int pidx(Engine& e, const std::string& n) { return e.param_index.at(n); }

struct LinearRef {
    int w, b, N, K;
};

static LinearRef reg_linear(Engine& e, const std::string& p, int N, int K) {
    LinearRef r;
    r.w = e.add_param(p + ".weight", CAPF_P_LIN_W, {N, K});
    r.b = e.add_param(p + ".bias", CAPF_P_LIN_B, {N});
    r.N = N;
    r.K = K;
    return r;
}

static void reg_ln(Engine& e, const std::string& p, int C) {
    e.add_param(p + ".weight", CAPF_P_LN_W, {C});
    e.add_param(p + ".bias", CAPF_P_LN_B, {C});
}

static int make_linear_pack(Engine& e, const std::vector<std::string>& names, bool as_bf16 = false, bool quad = false) {
    Pack pk;
    pk.kind = 1;
    pk.quad = quad;
    pk.n_lin = (int)names.size();
    int N = 0, K = 0;
    for (int i = 0; i < pk.n_lin; ++i) {
        pk.w[i] = pidx(e, names[i] + ".weight");
        pk.b[i] = pidx(e, names[i] + ".bias");
        N += (int)e.params[pk.w[i]].shape[0];
        K = (int)e.params[pk.w[i]].shape[1];
    }
    pk.N = N;
    pk.K = K;
    pk.Kpad = as_bf16 ? round64(K) : round32(K);
    if (quad) pk.Kpad = K;                     // (K % 4 == 0: embed dims and level channel counts)
    pk.direct = !as_bf16 && !quad && (pk.n_lin == 1 && pk.Kpad == K);
    pk.bf16 = as_bf16;                         // bf16 copy [N][Kpad] for the bf16 MFMA projections (compute_dtype = bf16)
    // the fp32 projections also as two fp16 pieces for igemm_f32h2.hip, packed LAZILY (Engine::ensure_h2g_lifter): a training loop changes the
    // weights every step and its forward runs train.cpp's GEMMs, so the copy is rebuilt only when an inference forward needs it
    if (!as_bf16 && !quad && e.use_h2g && !e.bf16() && K % 4 == 0 && N % 4 == 0) { pk.h2g = true; pk.KpadH = round32(K); }
    e.packs.push_back(pk);
    return (int)e.packs.size() - 1;
}

// rows-mode GEMM:  out[omap(m) + n] = act(A[amap(m)] . W[n] + b[n] + res[rmap(m) + n])
static void gemm_rows(Engine& e, const std::string& name, int pack, int a_buf, RowMap amap, long rows_pf, int out_buf,
                      RowMap omap, int act, int res_buf, RowMap rmap, int res_param = -1, const std::string& ln = "",
                      float ln_eps = 0.f) {
    const Pack& pk = e.packs[pack];
    Op op;
    op.kind = OP_GEMM;
    op.name = name;
    op.conv = 0;
    op.pack = pack;
    op.in[0] = a_buf;
    op.amap = amap;
    op.rows_per_frame = rows_pf;
    op.N = pk.N; op.K = pk.K; op.act = act;
    op.out = out_buf;
    op.omap = omap;
    op.aux = res_buf;
    op.rmap = rmap;
    op.res_param = res_param;
    if (!ln.empty()) {                       // LayerNorm(A rows) folded into the GEMM's fragment path (igemm_f32.hip, LNA)
        op.ln_w = pidx(e, ln + ".weight");
        op.ln_b = pidx(e, ln + ".bias");
        op.eps = ln_eps;
    }
    op.flops_per_frame = 2.0 * rows_pf * (double)pk.N * pk.K;
    if (pk.kind == 1 && pk.bf16) {           // bf16 operands (A buffer holds bf16 rows), fp32 accumulate
        op.bf16 = 2;
        op.out_bf16 = act == ACT_GELU ? 1 : 0;   // fc1: GELU, bf16 hidden rows; everything else lands in the fp32 stream
    }
    e.use(a_buf);
    e.use(res_buf);
    e.use(out_buf);
    e.push(op);
}

static void layernorm(Engine& e, const std::string& name, const std::string& ln, float eps, int in_buf, RowMap imap,
                      int add_buf, RowMap amap, int out_buf, long rows_pf, int C, bool out_bf16 = false) {
    Op op;
    op.kind = OP_LAYERNORM;
    op.name = name;
    op.in[0] = in_buf;
    op.amap = imap;
    op.aux = add_buf;
    op.rmap = amap;
    op.out = out_buf;
    op.p0 = pidx(e, ln + ".weight");
    op.p1 = pidx(e, ln + ".bias");
    op.eps = eps;
    op.rows_per_frame = rows_pf;
    op.C = C;
    op.out_bf16 = out_bf16 ? 1 : 0;
    e.use(in_buf);
    e.use(add_buf);
    e.use(out_buf);
    e.push(op);
}

static void debug_copy(Engine& e, const std::string& name, int src, size_t elems, std::initializer_list<int64_t> shape) {
    Op op;
    op.kind = OP_FUSE;       // a 1-input fuse without ReLU is a plain copy
    op.name = "copy." + name;
    op.n_in = 1;
    op.in[0] = src;
    op.H = 1; op.W = 1; op.C = (int)elems; op.relu = 0;
    op.i0 = 1;               // debug-only op
    e.use(src);
    op.out = e.new_buffer(elems, name);
    e.push(op);
    name_tensor(e, name, op.out, shape);
}

void Engine::build_lifter(const Tensor feats[4]) {
    const std::string V = "volume_net";
    const int J = cfg.num_joints, L = cfg.levels, L1 = L + 1, C = cfg.embed_dim_ratio, D = C * L1;
    const int NH = cfg.deform_heads, NS = cfg.deform_samples, HD = C / NH;
    int Cl[4];
    for (int l = 0; l < L; ++l) Cl[l] = feats[l].C;

    // ---- schema, in the reference's registration order (pose_dformer.py:174-208)
    const int pos = add_param(V + ".Spatial_pos_embed", CAPF_P_RAW, {1, L1, J, C});
    reg_linear(*this, V + ".coord_embed", C, 2);
    for (int l = 0; l < L; ++l) reg_linear(*this, V + ".feat_embed." + std::to_string(l), C, Cl[l]);
    auto reg_block = [&](const std::string& p, int dim) {
        reg_ln(*this, p + ".norm1", dim);
        reg_linear(*this, p + ".attn.qkv", 3 * dim, dim);
        reg_linear(*this, p + ".attn.proj", dim, dim);
        reg_ln(*this, p + ".norm2", dim);
        reg_linear(*this, p + ".mlp.fc1", 2 * dim, dim);
        reg_linear(*this, p + ".mlp.fc2", dim, 2 * dim);
    };
    const int DEP = cfg.depth > 0 ? cfg.depth : L;         // blocks per group (ContextPose_mpi pose_dformer.py:199)
    for (int i = 0; i < DEP; ++i) reg_block(V + ".joint_blocks." + std::to_string(i), D);
    for (int i = 0; i < DEP; ++i) reg_block(V + ".res_blocks." + std::to_string(i), C);
    if (cfg.context_blocks) {
        for (int i = 0; i < L; ++i) {
            const std::string p = V + ".context_blocks." + std::to_string(i);
            reg_ln(*this, p + ".norm1", C);
            reg_linear(*this, p + ".attention_weights", NH * NS, C);
            reg_linear(*this, p + ".sampling_offsets", 2 * NH * NS, C);
            for (int l = 0; l < L; ++l) reg_linear(*this, p + ".embed_proj." + std::to_string(l), HD, Cl[l]);
            reg_ln(*this, p + ".norm2", C);
            reg_linear(*this, p + ".mlp.fc1", 2 * C, C);
            reg_linear(*this, p + ".mlp.fc2", C, 2 * C);
        }
    }
    reg_ln(*this, V + ".head.0", D);
    reg_linear(*this, V + ".head.1", 3, D);

    // ---- buffers (per frame)
    const int X = new_buffer((size_t)J * D, "tokens");            // [J, L1, C]  ("b p l c")
    const int Q = new_buffer((size_t)J * D, "ln_out");
    const int Hb = new_buffer((size_t)J * 2 * D, "mlp_hidden");
    const int QKV = new_buffer((size_t)J * 3 * D, "qkv");
    const int O = new_buffer((size_t)J * D, "attn_out");
    keep(*this, X); keep(*this, Q); keep(*this, Hb); keep(*this, QKV); keep(*this, O);

    // ---- embedding + reference-point sampling (pose_dformer.py:214-226)
    // LayerNorm folded into the following GEMM (igemm_f32.hip, LNA) where the normalised width is small: every block
    // recomputes the statistics of its own rows, which is free at K = 128 (norm + GEMM 22.7 -> 18.3 us at batch 64) and
    // a loss at K = 640, where 30 column tiles would each re-read 160 KB of rows (29 + 9.5 -> 44 us): the joint
    // blocks keep their LayerNorm launch
    // (compute_dtype = bf16: the projections take bf16 A rows, which the LayerNorm kernel writes directly)
    // lifter projections (qkv / proj / fc1 / fc2) on the bf16 MFMA path; CAPF_PLAN_LIFTER_FP32 keeps them (and their LayerNorm folding) fp32
    const bool lb = bf16() && !(cfg.plan_flags & CAPF_PLAN_LIFTER_FP32);
    auto ln_fold_ok = [&](int dim) { return fused_lifter && dim <= 256 && !lb; };
    const bool ln_fold = ln_fold_ok(C);
    if (fused_lifter) {
        Op op;
        op.kind = OP_EMBED;
        op.name = "embed";
        op.out = X;
        op.p0 = pidx(*this, V + ".coord_embed.weight");
        op.p1 = pidx(*this, V + ".coord_embed.bias");
        op.p2 = pos;
        op.i0 = J; op.i1 = L; op.i2 = L1; op.C = C;
        op.bf16 = bf16() ? 1 : 0;
        use(X);
        for (int l = 0; l < L; ++l) {
            const std::string ls = std::to_string(l);
            op.in[l] = feats[l].buf;
            op.lvlH[l] = feats[l].H; op.lvlW[l] = feats[l].W; op.lvlC[l] = Cl[l];
            use(feats[l].buf);
            op.pw[l] = pidx(*this, V + ".feat_embed." + ls + ".weight");
            op.pb[l] = pidx(*this, V + ".feat_embed." + ls + ".bias");
            op.pq[l] = make_linear_pack(*this, {V + ".feat_embed." + ls}, false, true);
            op.outs[l] = new_buffer((size_t)J * Cl[l], "sampled" + ls);
            op.idxs[l] = new_buffer((size_t)J * 2, "idx" + ls);
            name_tensor(*this, "sampled" + ls, op.outs[l], {-1, J, Cl[l]});
            name_tensor(*this, "idx" + ls, op.idxs[l], {-1, J, 2}, 1);
            op.flops_per_frame += 2.0 * J * (double)C * Cl[l];
        }
        push(op);
    } else {
    {
        Op op;
        op.kind = OP_PREP_EMBED;
        op.name = "prep_embed";
        op.out = X;
        op.p0 = pidx(*this, V + ".coord_embed.weight");
        op.p1 = pidx(*this, V + ".coord_embed.bias");
        op.p2 = pos;
        op.i0 = J; op.i1 = L1; op.C = C;
        use(X);
        push(op);
    }
    for (int l = 0; l < L; ++l) {
        const std::string ls = std::to_string(l);
        Op op;
        op.kind = OP_SAMPLE_REF;
        op.name = "sample_ref." + ls;
        op.in[0] = feats[l].buf;
        op.H = feats[l].H; op.W = feats[l].W; op.C = Cl[l]; op.i0 = J;
        op.bf16 = bf16() ? 1 : 0;
        use(feats[l].buf);
        const int S = new_buffer((size_t)J * Cl[l], "sampled" + ls);
        const int I = new_buffer((size_t)J * 2, "idx" + ls);
        op.out = S;
        op.aux2 = I;
        push(op);
        name_tensor(*this, "sampled" + ls, S, {-1, J, Cl[l]});
        name_tensor(*this, "idx" + ls, I, {-1, J, 2}, 1);
        const int pk = make_linear_pack(*this, {V + ".feat_embed." + ls});
        // out X[b,p,1+l,:] = S W^T + b + pos[0,1+l,p,:]
        gemm_rows(*this, "feat_embed." + ls, pk, S, row_ld(Cl[l]), J, X, row_ld(D, (long)(1 + l) * C), ACT_NONE, -1,
                  RowMap{J, 0, C, (long)(1 + l) * J * C}, pos);
    }
    }

    // ---- deformable context blocks (pose_dformer.py:115-141), tokens 1..L with token 0 as query bias
    if (cfg.context_blocks) {
        const int AO = new_buffer((size_t)J * L * 3 * NH * NS, "attn_off");
        keep(*this, AO);
        int U[4];
        for (int l = 0; l < L; ++l) {
            U[l] = new_buffer((size_t)J * NH * Cl[l], "deform_u" + std::to_string(l));
            keep(*this, U[l]);
        }
        const RowMap tok{L, D, C, C};           // row (b,p,l') -> X[b,p,1+l',:]
        const RowMap tok0{L, D, 0, 0};          // row (b,p,l') -> X[b,p,0,:]
        for (int i = 0; i < L; ++i) {
            const std::string p = V + ".context_blocks." + std::to_string(i);
            const std::string n = "ctx" + std::to_string(i);
            const int pk_ao = make_linear_pack(*this, {p + ".attention_weights", p + ".sampling_offsets"});
            ctx_ao_pack.push_back(pk_ao);                   // (row layout: the training step's GEMMs, train.cpp)
            // debug taps of the border-mode sampling site (pose_dformer.py:126-128): positions and NW corner indices
            const int tap_pos = new_buffer((size_t)J * L * NH * NS * 2, "cpos" + std::to_string(i));
            const int tap_idx = new_buffer((size_t)J * L * NH * NS * 2, "cidx" + std::to_string(i));
            name_tensor(*this, "cpos" + std::to_string(i), tap_pos, {-1, J, L * NH * NS, 2});
            name_tensor(*this, "cidx" + std::to_string(i), tap_idx, {-1, J, L * NH * NS, 2}, 1);
            ctx_tap_pos.push_back(tap_pos);
            ctx_tap_idx.push_back(tap_idx);
            if (fused_lifter) {
                Op op;
                op.kind = OP_CTX_ATTN;
                op.name = n + ".attn";
                op.pack = make_linear_pack(*this, {p + ".attention_weights", p + ".sampling_offsets"}, false, true);
                op.out = X;
                use(X);
                op.p0 = pidx(*this, p + ".norm1.weight");
                op.p1 = pidx(*this, p + ".norm1.bias");
                op.eps = 1e-5f;
                for (int l = 0; l < L; ++l) {
                    op.in[l] = feats[l].buf;
                    op.lvlH[l] = feats[l].H; op.lvlW[l] = feats[l].W; op.lvlC[l] = Cl[l];
                    use(feats[l].buf);
                    op.pw[l] = pidx(*this, p + ".embed_proj." + std::to_string(l) + ".weight");
                    op.pb[l] = pidx(*this, p + ".embed_proj." + std::to_string(l) + ".bias");
                    op.pq[l] = make_linear_pack(*this, {p + ".embed_proj." + std::to_string(l)}, false, true);
                    op.outs[l] = U[l];           // (the sample sums cross HBM to the embed_proj launch: lifter_fused.hip ctx_proj_kernel)
                    use(U[l]);
                    op.flops_per_frame += 2.0 * J * NH * (double)HD * Cl[l];
                }
                op.flops_per_frame += 2.0 * J * L * (double)C * 3 * NH * NS;
                op.i0 = J; op.i1 = L; op.i2 = NH; op.i3 = NS; op.C = C;
                op.bf16 = bf16() ? 1 : 0;
                op.idxs[0] = tap_pos; op.idxs[1] = tap_idx;
                push(op);
            } else {
            layernorm(*this, n + ".norm1", p + ".norm1", 1e-5f, X, tok, X, tok0, Q, (long)J * L, C);
            gemm_rows(*this, n + ".attn_off", pk_ao, Q, row_ld(C), (long)J * L, AO, row_ld(3 * NH * NS), ACT_NONE, -1,
                      row_ld(0));
            {
                Op op;
                op.kind = OP_DEFORM;
                op.name = n + ".deform";
                op.aux = AO;
                use(AO);
                for (int l = 0; l < L; ++l) {
                    op.in[l] = feats[l].buf;
                    op.lvlH[l] = feats[l].H; op.lvlW[l] = feats[l].W; op.lvlC[l] = Cl[l];
                    op.outs[l] = U[l];
                    use(feats[l].buf);
                    use(U[l]);
                }
                op.i0 = J; op.i1 = L; op.i2 = NH; op.i3 = NS;
                op.bf16 = bf16() ? 1 : 0;
                op.idxs[0] = tap_pos; op.idxs[1] = tap_idx;
                push(op);
            }
            for (int l = 0; l < L; ++l) {
                const int pk = make_linear_pack(*this, {p + ".embed_proj." + std::to_string(l)});
                const RowMap dst{NH, D, HD, (long)(1 + l) * C};    // row (b,p,h) -> X[b,p,1+l,h*HD:]
                gemm_rows(*this, n + ".embed_proj." + std::to_string(l), pk, U[l], row_ld(Cl[l]), (long)J * NH, X, dst,
                          ACT_NONE, X, dst);
            }
            }
            if (fused_lifter && use_h2g && !lb && !bf16() && res_chain_ok(C, 5, 8, 1)) {
                // the MLP half as one launch on the context tokens' rows (lifter_chain.hip, ATTN = false): norm2 -> fc1 + GELU -> fc2 + x
                Op op;
                op.kind = OP_MLP_CHAIN;
                op.name = n + ".mlp";
                op.in[0] = X;
                op.out = X;
                op.amap = tok;
                op.C = C;
                op.eps = 1e-5f;
                op.rows_per_frame = (long)J * L;
                for (const char* lin : {".mlp.fc1", ".mlp.fc2"}) {
                    op.chain.push_back(make_linear_pack(*this, {p + lin}));
                    packs.back().chain = true;
                }
                op.chain.push_back(pidx(*this, p + ".norm2.weight"));
                op.chain.push_back(pidx(*this, p + ".norm2.bias"));
                op.flops_per_frame = 2.0 * J * L * (double)C * (2 * C + 2 * C);
                use(X);
                push(op);
                has_res_chain = true;
                continue;
            }
            if (ln_fold) {
                gemm_rows(*this, n + ".fc1", make_linear_pack(*this, {p + ".mlp.fc1"}), X, tok, (long)J * L, Hb,
                          row_ld(2 * C), ACT_GELU, -1, row_ld(0), -1, p + ".norm2", 1e-5f);
            } else {
                layernorm(*this, n + ".norm2", p + ".norm2", 1e-5f, X, tok, -1, row_ld(0), Q, (long)J * L, C, lb);
                gemm_rows(*this, n + ".fc1", make_linear_pack(*this, {p + ".mlp.fc1"}, lb), Q, row_ld(C), (long)J * L, Hb,
                          row_ld(2 * C), ACT_GELU, -1, row_ld(0));
            }
            gemm_rows(*this, n + ".fc2", make_linear_pack(*this, {p + ".mlp.fc2"}, lb), Hb, row_ld(2 * C), (long)J * L, X, tok,
                      ACT_NONE, X, tok);
        }
    }
    debug_copy(*this, "tok_ctx", X, (size_t)J * D, {-1, J, L1, C});

    // ---- Block x L over the L1 level-tokens of each joint, then over the J joint tokens (:231-238)
    auto attn_blocks = [&](const std::string& group, const std::string& tag, int dim, long rows_pf, int tokens,
                           int groups_pf) {
        const bool ln_fold = ln_fold_ok(dim);
        const int nblk = cfg.depth > 0 ? cfg.depth : L;
        // the res blocks (tokens of ONE joint, 128 wide) as one launch: a workgroup takes 6 joints through every block without leaving
        // the CU (lifter_chain.hip); fp32 lifter on the two-piece packs only -- the bf16 plan and CAPF_PLAN_NO_FUSED_LIFTER /
        // CAPF_PLAN_NO_F32H2_GEMM keep one launch per op
        if (fused_lifter && use_h2g && !lb && !bf16() && groups_pf > 1 && res_chain_ok(dim, tokens, cfg.num_heads, nblk)) {
            Op op;
            op.kind = OP_RES_CHAIN;
            op.name = tag + ".chain";
            op.in[0] = X;
            op.out = X;
            op.i0 = tokens; op.i1 = cfg.num_heads; op.i2 = nblk; op.C = dim;
            op.eps = 1e-6f;
            op.rows_per_frame = rows_pf;
            for (int i = 0; i < nblk; ++i) {
                const std::string p = V + "." + group + "." + std::to_string(i);
                for (const char* lin : {".attn.qkv", ".attn.proj", ".mlp.fc1", ".mlp.fc2"}) {
                    op.chain.push_back(make_linear_pack(*this, {p + lin}));
                    packs.back().chain = true;
                }
                for (const char* ln : {".norm1.weight", ".norm1.bias", ".norm2.weight", ".norm2.bias"}) op.chain.push_back(pidx(*this, p + ln));
                op.flops_per_frame += 2.0 * rows_pf * (double)dim * (3 * dim + dim + 2 * dim + 2 * dim) + 4.0 * groups_pf * tokens * tokens * dim;
            }
            use(X);
            push(op);
            has_res_chain = true;
            return;
        }
        for (int i = 0; i < nblk; ++i) {
            const std::string p = V + "." + group + "." + std::to_string(i);
            const std::string n = tag + std::to_string(i);
            if (ln_fold) {
                gemm_rows(*this, n + ".qkv", make_linear_pack(*this, {p + ".attn.qkv"}), X, row_ld(dim), rows_pf, QKV,
                          row_ld(3 * dim), ACT_NONE, -1, row_ld(0), -1, p + ".norm1", 1e-6f);
            } else {
                layernorm(*this, n + ".norm1", p + ".norm1", 1e-6f, X, row_ld(dim), -1, row_ld(0), Q, rows_pf, dim, lb);
                gemm_rows(*this, n + ".qkv", make_linear_pack(*this, {p + ".attn.qkv"}, lb), Q, row_ld(dim), rows_pf, QKV,
                          row_ld(3 * dim), ACT_NONE, -1, row_ld(0));
            }
            {
                Op op;
                op.kind = OP_ATTENTION;
                op.name = n + ".attn";
                op.in[0] = QKV;
                op.out = O;
                op.i0 = groups_pf; op.i1 = tokens; op.i2 = cfg.num_heads; op.i3 = dim / cfg.num_heads;
                op.flops_per_frame = 4.0 * groups_pf * tokens * tokens * dim;
                op.out_bf16 = lb ? 1 : 0;      // the attention output is only ever the A operand of proj
                use(QKV); use(O);
                push(op);
            }
            gemm_rows(*this, n + ".proj", make_linear_pack(*this, {p + ".attn.proj"}, lb), O, row_ld(dim), rows_pf, X,
                      row_ld(dim), ACT_NONE, X, row_ld(dim));
            if (ln_fold) {
                gemm_rows(*this, n + ".fc1", make_linear_pack(*this, {p + ".mlp.fc1"}), X, row_ld(dim), rows_pf, Hb,
                          row_ld(2 * dim), ACT_GELU, -1, row_ld(0), -1, p + ".norm2", 1e-6f);
            } else {
                layernorm(*this, n + ".norm2", p + ".norm2", 1e-6f, X, row_ld(dim), -1, row_ld(0), Q, rows_pf, dim, lb);
                gemm_rows(*this, n + ".fc1", make_linear_pack(*this, {p + ".mlp.fc1"}, lb), Q, row_ld(dim), rows_pf, Hb,
                          row_ld(2 * dim), ACT_GELU, -1, row_ld(0));
            }
            gemm_rows(*this, n + ".fc2", make_linear_pack(*this, {p + ".mlp.fc2"}, lb), Hb, row_ld(2 * dim), rows_pf, X,
                      row_ld(dim), ACT_NONE, X, row_ld(dim));
        }
    };
    attn_blocks("res_blocks", "res", C, (long)J * L1, L1, J);
    debug_copy(*this, "tok_res", X, (size_t)J * D, {-1, J, L1, C});
    attn_blocks("joint_blocks", "joint", D, (long)J, J, 1);
    debug_copy(*this, "tok_joint", X, (size_t)J * D, {-1, J, L1, C});

    {   // head (:240)
        Op op;
        op.kind = OP_HEAD;
        op.name = "head";
        op.in[0] = X;
        op.p0 = pidx(*this, V + ".head.0.weight");
        op.p1 = pidx(*this, V + ".head.0.bias");
        op.p2 = pidx(*this, V + ".head.1.weight");
        op.p3 = pidx(*this, V + ".head.1.bias");
        op.eps = 1e-5f;
        op.rows_per_frame = J;
        op.C = D;
        op.i0 = 3;
        op.flops_per_frame = 2.0 * J * D * 3;
        use(X);
        push(op);
    }
}

// Greedy offset assignment over buffer lifetimes (ops run in order on one stream).
void Engine::assign_offsets() {
    struct Live { size_t off, size; int last; };
    std::vector<int> order(bufs.size());
    for (size_t i = 0; i < bufs.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bufs[a].def_op < bufs[b].def_op; });
    // Ops of different lanes of a fork/join region run concurrently: a buffer whose life touches a region
    // stays allocated for the whole region (no reuse across lanes while they may overlap).
    for (Buffer& b : bufs)
        for (const auto& rg : regions) {
            const bool touches = b.def_op <= rg.second && b.last_op >= rg.first;
            if (touches && b.last_op < rg.second) b.last_op = rg.second;
            if (touches && b.def_op > rg.first) b.def_op = rg.first;
        }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bufs[a].def_op < bufs[b].def_op; });
    std::vector<Live> live;
    size_t top = 0;
    for (int id : order) {
        Buffer& b = bufs[id];
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Live& l) { return l.last < b.def_op; }), live.end());
        std::sort(live.begin(), live.end(), [](const Live& x, const Live& y) { return x.off < y.off; });
        size_t off = 0;
        for (const Live& l : live) {
            if (off + b.elems <= l.off) break;
            off = std::max(off, l.off + l.size);
        }
        b.offset = off;
        live.push_back(Live{off, b.elems, b.last_op});
        top = std::max(top, off + b.elems);
    }
    ws_elems_per_frame = top;
}

// Dependency levels of every fork/join region (run_region_grouped): an op's level is one more than the
// highest level of any earlier op of the region it conflicts with on a buffer (read-after-write,
// write-after-read, write-after-write).
void Engine::schedule_regions() {
    region_levels.assign(regions.size(), {});
    for (size_t r = 0; r < regions.size(); ++r) {
        const int lo = regions[r].first + 1, hi = regions[r].second;
        std::vector<int> level(hi - lo, 0);
        auto reads = [&](const Op& o, int b) {
            if (b < 0) return false;
            for (int i = 0; i < 4; ++i) if (o.in[i] == b) return true;
            return o.aux == b;
        };
        auto writes = [&](const Op& o, int b) {
            if (b < 0) return false;
            if (o.out == b || o.aux2 == b) return true;
            for (int i = 0; i < 4; ++i) if (o.outs[i] == b) return true;
            return false;
        };
        auto touched = [&](const Op& o) {
            std::vector<int> v;
            for (int i = 0; i < 4; ++i) { if (o.in[i] >= 0) v.push_back(o.in[i]); if (o.outs[i] >= 0) v.push_back(o.outs[i]); }
            if (o.aux >= 0) v.push_back(o.aux);
            if (o.out >= 0) v.push_back(o.out);
            if (o.aux2 >= 0) v.push_back(o.aux2);
            return v;
        };
        int max_level = 0;
        for (int i = lo; i < hi; ++i) {
            const Op& oi = ops[i];
            for (int j = lo; j < i; ++j) {
                const Op& oj = ops[j];
                bool dep = false;
                for (int b : touched(oi))
                    if ((writes(oj, b)) || (reads(oj, b) && writes(oi, b))) { dep = true; break; }
                if (dep) level[i - lo] = std::max(level[i - lo], level[j - lo] + 1);
            }
            max_level = std::max(max_level, level[i - lo]);
        }
        // A fuse sum nothing else in the region depends on (the HRNet fuse: its outputs are read after the join) moves to the
        // LAST level, where run_region_grouped issues all of a module's sums as one launch instead of 2-4 small ones.
        for (int i = lo; i < hi; ++i) {
            const Op& oi = ops[i];
            if (oi.kind != OP_FUSE || oi.i0 == 1) continue;
            bool sink = true;
            for (int j = i + 1; j < hi && sink; ++j) {
                const Op& oj = ops[j];
                for (int b : touched(oj))
                    if (writes(oi, b) || (reads(oi, b) && writes(oj, b))) { sink = false; break; }
            }
            if (sink) level[i - lo] = max_level;
        }
        region_levels[r].assign(max_level + 1, {});
        for (int i = lo; i < hi; ++i) region_levels[r][level[i - lo]].push_back(i);
    }
}

bool Engine::build() {
    if (cfg.height % 32 != 0 || cfg.width % 32 != 0) {
        err = "height and width must be multiples of 32";
        return false;
    }
    if (cfg.levels != 4 || cfg.num_joints <= 0 || cfg.embed_dim_ratio % (4 * cfg.num_heads) != 0 ||
        cfg.embed_dim_ratio % (4 * cfg.deform_heads) != 0 || cfg.deform_samples != 4 || cfg.deform_heads != 4) {
        err = "unsupported lifter configuration (levels must be 4, 4x4 deformable sampling, embed_dim_ratio % 32 == 0)";
        return false;
    }
    if (cfg.depth != 0 && cfg.depth != cfg.levels && (cfg.depth < 1 || cfg.depth > 8 || cfg.context_blocks || cfg.training)) {
        err = "depth != levels: 1..8 blocks per group, only for the variant without context blocks (ContextPose_mpi) and only for "
              "inference plans";
        return false;
    }
    if (cfg.plan_flags & CAPF_PLAN_NO_FUSED_LIFTER) fused_lifter = false;
    if (cfg.plan_flags & CAPF_PLAN_NO_WINOGRAD) use_wino = false;
    if (cfg.plan_flags & CAPF_PLAN_NO_ROW_HALO) use_rh = false;
    if (cfg.plan_flags & CAPF_PLAN_WINOGRAD_F23_ONLY) wino_f43 = false;
    if (cfg.plan_flags & CAPF_PLAN_NO_PWCHAIN) use_pwchain = false;
    if (cfg.plan_flags & CAPF_PLAN_NO_UPADD) use_upadd = false;
    if (cfg.plan_flags & CAPF_PLAN_NO_BNECK) use_bneck = false;
    if (cfg.plan_flags & CAPF_PLAN_NO_BATCHED_REDUCE) batch_reduce = false;
    if (cfg.plan_flags & CAPF_PLAN_H2_PLANES) use_h2_planes = true;
    if (cfg.plan_flags & CAPF_PLAN_NO_WS) use_ws = false;
    if (cfg.plan_flags & CAPF_PLAN_NO_F32X3) use_x3 = false;
    if (cfg.plan_flags & CAPF_PLAN_F32X3_EXACT) x3_h2 = false;
    if (cfg.plan_flags & CAPF_PLAN_NO_F32H2_GEMM) use_h2g = false;
    // tuning knobs of the diagnostic build only (diag_env is a constant nullptr in the product library)
    if (const char* fz = diag_env("CAPF_LIFTER_FUSED")) fused_lifter = atoi(fz) != 0;
    if (const char* wz = diag_env("CAPF_WINO")) use_wino = atoi(wz) != 0;
    if (const char* rz = diag_env("CAPF_BF16_RH")) use_rh = atoi(rz) != 0;
    if (const char* wb = diag_env("CAPF_WINO_MIN_BATCH")) wino_min_batch = atoi(wb);
    if (const char* wf = diag_env("CAPF_WINO_F43")) wino_f43 = atoi(wf) != 0;
    if (const char* wf = diag_env("CAPF_WINO_F43_MINHW")) wino_f43_min_hw = atoi(wf);
    if (const char* wf = diag_env("CAPF_WINO_F43_CPN")) wino_f43_cpn = atoi(wf) != 0;
    if (const char* wf = diag_env("CAPF_WINO_F43_MAXHW")) wino_f43_max_hw = atoi(wf);
    // the fused front half (lifter_fused.hip) holds a token row in 4 values per lane and a sampled row in 2 KiB of LDS: wider
    // configurations (embed_dim_ratio 272 / 288 / ..., > 512-channel context maps) take the one-kernel-per-op plan instead
    if (cfg.embed_dim_ratio > 256) fused_lifter = false;
    if (cfg.backbone == CAPF_HRNET && cfg.hr_channels[3] > 512) fused_lifter = false;
    Tensor img{EXT_IMAGES, cfg.height, cfg.width, 3};
    Tensor feats[4];
    if (cfg.backbone == CAPF_HRNET) {
        for (int i = 0; i < 4; ++i)
            if (cfg.hr_channels[i] <= 0 || cfg.hr_channels[i] % 4 != 0) {
                err = "HRNet branch widths must be positive multiples of 4";
                return false;
            }
        build_hrnet(img, feats);
    } else if (cfg.backbone == CAPF_CPN50) {
        build_cpn(img, feats);
    } else {
        err = "unknown backbone";
        return false;
    }
    for (int l = 0; l < 4; ++l) {
        name_tensor(*this, "feat" + std::to_string(l), feats[l].buf, {-1, feats[l].H, feats[l].W, feats[l].C}, bf16() ? 2 : 0);
        const int expect = cfg.backbone == CAPF_CPN50 ? cfg.base_dim : cfg.base_dim << l;
        if (feats[l].C != expect) {
            err = "poseformer.base_dim does not match the backbone's context-map widths";
            return false;
        }
    }
    n_backbone_ops = (int)ops.size();
    for (int l = 0; l < 4; ++l) {
        feat_buf[l] = feats[l].buf; feat_H[l] = feats[l].H; feat_W[l] = feats[l].W; feat_C[l] = feats[l].C;
    }
    build_lifter(feats);
    // flat gradient layout: the volume_net.* parameters in schema (= registration) order
    grad_off.assign(params.size(), -1);
    grad_elems = 0;
    for (size_t i = 0; i < params.size(); ++i)
        if (params[i].name.rfind("volume_net.", 0) == 0) {
            grad_off[i] = grad_elems;
            grad_elems += params[i].numel();
        }
    assign_offsets();
    schedule_regions();
    // the GEMM kernels address a whole activation tensor (and a conv's input window) with 32-bit offsets
    // (launch_gemm_f32: M * ld < 4e9 elements of out / res, < 4e9 bytes of a rows-mode operand)
    {
        double worst = 1.0;          // largest per-frame tensor, in elements
        for (const Op& op : ops)
            if (op.kind == OP_GEMM) {
                worst = std::max(worst, (double)op.rows_per_frame * (double)std::max(op.omap.S1, (long)op.N));
                if (op.conv) worst = std::max(worst, (double)op.H * op.W * op.Cin);
                else worst = std::max(worst, (double)op.rows_per_frame * (double)std::max(op.amap.S1, (long)op.K) * 4.0 / (op.amap.G > 0 ? op.amap.G : 1));
            }
        batch_limit = (int)std::min(2.0e9, 3.9e9 / worst);
    }
    // the layer1 bottleneck pairs (conv3 -> next block's conv1) the pointwise-chain kernel takes: structural half of gemm_f32_pwchain_ok
    for (size_t i = 0; i + 1 < ops.size(); ++i) {
        Op& a = ops[i];
        Op& b = ops[i + 1];
        if (a.kind != OP_GEMM || b.kind != OP_GEMM || !a.conv || !b.conv || a.bf16 || b.bf16) continue;
        if (a.ks != 1 || b.ks != 1 || a.stride != 1 || b.stride != 1 || a.Cin != 64 || a.N != 256 || b.Cin != 256 || b.N != 64) continue;
        if (b.in[0] != a.out || b.region != a.region || b.lane != a.lane || a.aux < 0 || b.aux >= 0) continue;
        a.pw_pair = b.pw_pair = 1;
    }
    // at which batches a split-fp32 tile takes a Winograd-eligible conv (Engine::wino_now): f32x3_takes is monotone in the batch up to the
    // tile's 2 GB tensor limit, so the range is [first batch it accepts, last batch it accepts]
    for (Op& op : ops) {
        if (op.kind != OP_GEMM || !op.wino || !packs[op.pack].x3) continue;
        // (upper end: the three-piece tile addresses whole tensors with 31-bit byte offsets; the two-piece tile only counts pixels)
        const double cap = x3_h2 ? 2.0e9 / ((double)op.H * op.W) : 2.0e9 / ((double)op.H * op.W * (double)std::max(op.Cin, op.N) * 4.0);
        int hi = (int)std::min(1.0e6, cap);
        while (hi >= 1 && hi > (int)cap - 4 && !f32x3_takes(hi, op.H, op.W, op.Cin, op.N, x3_h2)) --hi;   // (the limit itself is exclusive)
        if (hi < 1 || !f32x3_takes(hi, op.H, op.W, op.Cin, op.N, x3_h2)) continue;
        int lo = 1, top = hi;                       // smallest accepted batch by bisection
        while (lo < top) {
            const int mid = lo + (top - lo) / 2;
            if (f32x3_takes(mid, op.H, op.W, op.Cin, op.N, x3_h2)) top = mid; else lo = mid + 1;
        }
        op.x3_lo = lo; op.x3_hi = hi;
        // the Winograd layout of this conv is dead weight when the tile covers every batch the Winograd kernels could be asked for
        if (lo <= wino_min_batch && hi >= cfg.max_batch) packs[op.pack].wino_skip = true;
    }
    // unit tables of the two-fp16-piece conv tile: one per map geometry among the convs it can take (igemm_f32h2_ws_tile.h, UNIT TABLE)
    utab_host.clear();
    if (x3_h2 && use_x3) {
        std::map<std::tuple<int, int, int>, long> seen;
        for (Op& op : ops) {
            if (op.kind != OP_GEMM || !op.wino || !packs[op.pack].x3 || op.x3_hi < op.x3_lo) continue;
            const auto key = std::make_tuple(op.H, op.W, op.Cin);
            auto it = seen.find(key);
            if (it == seen.end()) {
                std::vector<unsigned> t(F32H2_UNIT_TABLE_WORDS);
                long at = -1;
                if (f32h2_unit_table(op.H, op.W, op.Cin, t.data())) {
                    at = (long)utab_host.size();
                    utab_host.insert(utab_host.end(), t.begin(), t.end());
                }
                it = seen.emplace(key, at).first;
            }
            op.h2_utab = it->second;
        }
    }
    // pack arena layout
    size_t off = 0;
    for (Pack& pk : packs) {
        if (pk.direct) continue;
        pk.w_off = off;
        off += pk.wino_skip ? 64 : round64(pk.bf16 ? ((size_t)pk.N * pk.Kpad + 1) / 2 : (size_t)pk.N * pk.Kpad);
        if (pk.wino) {
            pk.w2_off = off;
            off += round64((size_t)pk.N * pk.Kpad2);
        }
        if (pk.rh) {
            pk.w2_off = off;
            off += round64(((size_t)pk.N * pk.Kpad2 + 1) / 2);
        }
        if (pk.ws) {
            pk.w3_off = off;
            off += round64(((size_t)bf16_ws_pack_elems(pk.N, pk.Cin) + 1) / 2);
        }
        if (pk.x3) {
            pk.w3_off = off;
            off += round64(((size_t)(x3_h2 ? f32h2_pack_elems(pk.N, pk.Cin) : f32x3_pack_elems(pk.N, pk.Cin)) + 1) / 2);
        }
        pk.b_off = off;
        off += round64((size_t)pk.N);
    }
    for (Pack& pk : packs)                  // (direct linears included: the h2 copy is a pack of its own)
        if (pk.h2g) {
            pk.wh_off = off;
            off += round64((size_t)f32h2_gemm_pack_elems(pk.N, pk.KpadH));
            if (pk.chain) {
                pk.wc_off = off;
                off += round64((size_t)f32h2_gemm_pack_elems(pk.N, pk.KpadH));
            }
        }
    {   // room for the bias-copy table: one CopySegment per linear of every packed (non-direct) linear pack
        size_t nseg = 0;
        for (const Pack& pk : packs)
            if (pk.kind == 1 && !pk.direct) nseg += (size_t)pk.n_lin;
        bias_tab_off = off;
        off += round64(nseg * (sizeof(CopySegment) / sizeof(float)));
    }
    t_h2_plan();                            // the training step's weight table (train.cpp); its device image lives in the arena
    t_h2_tab_off = off;
    off += round64(t_h2_specs.size() * (sizeof(H2TrainW) / sizeof(float)) + 16);
    utab_off = off;
    off += round64(utab_host.size() + 16);
    pack_elems = off;
    return true;
}

}  // namespace capf
