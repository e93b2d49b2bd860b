// Executor + C ABI (include/capf.h).  Walks the static plan and enqueues the gfx950 kernels on the
// caller's stream; no allocation, no synchronisation, no host round trip inside capf_forward.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "engine.h"

namespace capf {

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return CAPF_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

// Rebuild the private packed copies from the borrowed parameters.
int Engine::repack(hipStream_t s, bool lifter_only) {
    for (const Param& p : params) {
        if (p.kind == CAPF_P_BN_NBT) continue;
        if (!p.ptr) {
            err = "parameter not set: " + p.name;
            return CAPF_ERR_STATE;
        }
    }
    if (!utab_on_device && !utab_host.empty()) {        // the conv tile's unit tables: a function of the plan, uploaded once (synchronous: a pageable source)
        HIP_TRY(hipMemcpy(pack_arena + utab_off, utab_host.data(), utab_host.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        utab_on_device = true;
    }
    for (const Pack& pk : packs) {                      // the convs' two-fp16-piece copies (igemm_f32h2.hip); the linears' are packed lazily
        if (!pk.h2g || pk.kind != 0 || lifter_only) continue;
        HIP_TRY(launch_pack_f32h2_gemm(params[pk.w[0]].ptr, params[pk.bn_g].ptr, params[pk.bn_b].ptr, params[pk.bn_m].ptr, params[pk.bn_v].ptr,
                                       1e-5f, pack_arena + pk.wh_off, nullptr, pk.N, pk.Cin, pk.ks, pk.K, pk.KpadH, s));
    }
    h2g_lifter_dirty = true;
    std::vector<CopySegment> jobs;                      // the packed linears' bias vectors: one launch for all of them (below)
    for (const Pack& pk : packs) {
        if (pk.direct) continue;
        if (lifter_only && pk.kind == 0) continue;      // conv+BN packs belong to the frozen backbone
        float* W = pack_arena + pk.w_off;
        float* B = pack_arena + pk.b_off;
        if (pk.kind == 0 && pk.bf16) {
            HIP_TRY(launch_pack_conv_bf16(params[pk.w[0]].ptr, params[pk.bn_g].ptr, params[pk.bn_b].ptr, params[pk.bn_m].ptr,
                                          params[pk.bn_v].ptr, 1e-5f, W, B, pk.N, pk.Cin, pk.ks, pk.Kpad, s));
            if (pk.rh)
                HIP_TRY(launch_pack_conv_bf16_rh(params[pk.w[0]].ptr, params[pk.bn_g].ptr, params[pk.bn_b].ptr, params[pk.bn_m].ptr,
                                                 params[pk.bn_v].ptr, 1e-5f, pack_arena + pk.w2_off, B, pk.N, pk.Cin,
                                                 bf16_rh_width(pk.Cin), s));
            if (pk.ws)
                HIP_TRY(launch_pack_conv_bf16_ws(params[pk.w[0]].ptr, params[pk.bn_g].ptr, params[pk.bn_b].ptr, params[pk.bn_m].ptr,
                                                 params[pk.bn_v].ptr, 1e-5f, pack_arena + pk.w3_off, B, pk.N, pk.Cin, s));
        } else if (pk.kind == 0 && pk.wino) {
            if (!pk.wino_skip)
            HIP_TRY(launch_pack_conv_wino(params[pk.w[0]].ptr, params[pk.bn_g].ptr, params[pk.bn_b].ptr, params[pk.bn_m].ptr,
                                          params[pk.bn_v].ptr, 1e-5f, W, B, pk.N, pk.Cin, s, pk.Kpad == 18 * pk.Cin ? 43 : 23));
            HIP_TRY(launch_pack_conv(params[pk.w[0]].ptr, params[pk.bn_g].ptr, params[pk.bn_b].ptr, params[pk.bn_m].ptr,
                                     params[pk.bn_v].ptr, 1e-5f, pack_arena + pk.w2_off, B, pk.N, pk.Cin, pk.ks, pk.Kpad2, s));
            if (pk.x3)
                HIP_TRY((x3_h2 ? launch_pack_conv_f32h2 : launch_pack_conv_f32x3)(params[pk.w[0]].ptr, params[pk.bn_g].ptr, params[pk.bn_b].ptr,
                                                                                  params[pk.bn_m].ptr, params[pk.bn_v].ptr, 1e-5f,
                                                                                  pack_arena + pk.w3_off, B, pk.N, pk.Cin, s));
        } else if (pk.kind == 0) {
            HIP_TRY(launch_pack_conv(params[pk.w[0]].ptr, params[pk.bn_g].ptr, params[pk.bn_b].ptr,
                                     params[pk.bn_m].ptr, params[pk.bn_v].ptr, 1e-5f, W, B, pk.N, pk.Cin, pk.ks,
                                     pk.Kpad, s));
        } else if (pk.bf16) {            // linear weights as bf16 [N][Kpad] (a 1x1 "conv" without BatchNorm), bias fp32
            int n0 = 0;
            for (int i = 0; i < pk.n_lin; ++i) {
                const int n = (int)params[pk.w[i]].shape[0];
                unsigned short* Wb = reinterpret_cast<unsigned short*>(W) + (size_t)n0 * pk.Kpad;
                HIP_TRY(launch_pack_conv_bf16(params[pk.w[i]].ptr, nullptr, nullptr, nullptr, nullptr, 0.f, Wb, nullptr, n, pk.K, 1,
                                              pk.Kpad, s));
                jobs.push_back(CopySegment{params[pk.b[i]].ptr, B + n0, n, 0});
                n0 += n;
            }
        } else if (pk.quad) {            // fused lifter kernels: Wq[k / 4][n][4], the linears concatenated along n
            int n0 = 0;
            for (int i = 0; i < pk.n_lin; ++i) {
                const int n = (int)params[pk.w[i]].shape[0];
                HIP_TRY(launch_pack_linear_quad(params[pk.w[i]].ptr, W, n, pk.K, n0, pk.N, s));
                jobs.push_back(CopySegment{params[pk.b[i]].ptr, B + n0, n, 0});
                n0 += n;
            }
        } else {
            int n0 = 0;
            for (int i = 0; i < pk.n_lin; ++i) {
                const int n = (int)params[pk.w[i]].shape[0];
                HIP_TRY(launch_pack_linear(params[pk.w[i]].ptr, W + (size_t)n0 * pk.Kpad, n, pk.K, pk.Kpad, s));
                jobs.push_back(CopySegment{params[pk.b[i]].ptr, B + n0, n, 0});
                n0 += n;
            }
        }
    }
    if (!jobs.empty()) {
        CopySegment* tab_dev = reinterpret_cast<CopySegment*>(pack_arena + bias_tab_off);
        const bool same = bias_tab_on_device && bias_tab.size() == jobs.size() &&
                          memcmp(bias_tab.data(), jobs.data(), jobs.size() * sizeof(CopySegment)) == 0;
        if (!same) {                                    // (parameters moved or first pack: a per-step lifter repack finds the table in place)
            HIP_TRY(hipStreamSynchronize(s));           // an upload still reading the previous host image must not see it change
            bias_tab = jobs;
            HIP_TRY(hipMemcpyAsync(tab_dev, bias_tab.data(), bias_tab.size() * sizeof(CopySegment), hipMemcpyHostToDevice, s));
            bias_tab_on_device = true;
        }
        HIP_TRY(launch_copy_segments(tab_dev, (int)jobs.size(), s));
    }
    packed = true;
    return CAPF_OK;
}

// The lifter's linears as two fp16 pieces (igemm_f32h2.hip), rebuilt on the forward's stream when an inference forward of batch >= 5 finds
// them stale: several linears concatenated along N are packed on their own rows, the inverse scales of all rows follow the whole matrix
int Engine::ensure_h2g_lifter(hipStream_t s) {
    if (!h2g_lifter_dirty) return CAPF_OK;
    for (const Pack& pk : packs) {
        if (!pk.h2g || pk.kind != 1) continue;
        int n0 = 0;
        for (int i = 0; i < pk.n_lin; ++i) {
            const int n = (int)params[pk.w[i]].shape[0];
            HIP_TRY(launch_pack_f32h2_gemm_rows(params[pk.w[i]].ptr, pack_arena + pk.wh_off, n0, n, pk.N, pk.K, pk.KpadH, s));
            n0 += n;
        }
        if (pk.chain) HIP_TRY(launch_res_chain_repack(pack_arena + pk.wh_off, pack_arena + pk.wc_off, pk.N, pk.KpadH, s));
    }
    h2g_lifter_dirty = false;
    return CAPF_OK;
}

GemmArgs Engine::gemm_args(const Op& op, int batch, bool planes) const {
    auto ptr = [&](int buf) -> float* { return (buf >= 0 && ws) ? bptr(buf, batch) : nullptr; };
    const Pack& pk = packs[op.pack];
    GemmArgs a{};
    a.A = op.in[0] == -2 ? images : ptr(op.in[0]);
    if (pk.direct) {
        a.Wp = params[pk.w[0]].ptr;
        a.bias = params[pk.b[0]].ptr;
    } else {
        a.Wp = pack_arena + pk.w_off;
        a.bias = pack_arena + pk.b_off;
    }
    a.res = op.res_param >= 0 ? params[op.res_param].ptr : ptr(op.aux);
    a.out = ptr(op.out);
    a.M = (int)(op.rows_per_frame * batch);
    a.N = op.N; a.K = op.K; a.Kpad = pk.Kpad;
    if (pk.rh) a.Wp2 = pack_arena + pk.w2_off;
    if (pk.ws || pk.x3) a.Wp3 = pack_arena + pk.w3_off;
    a.x3_h2 = pk.x3 && x3_h2;
    if (a.x3_h2 && op.h2_utab >= 0 && utab_on_device) a.h2_utab = reinterpret_cast<const unsigned*>(pack_arena + utab_off) + op.h2_utab;
    // the plain fp32 MFMA kernels' problems on the two-fp16-piece GEMM from batch 5 (launch_gemm_f32 / _group route them; below, the fp32
    // kernels with split-K win); LayerNorm folds of up to 256 columns included (igemm_f32h2.hip, LNA)
    if (pk.h2g && batch >= H2G_MIN_BATCH && !op.bf16 && !op.pw_pair && !(op.wino && wino_now(op, batch))) a.Wh2 = pack_arena + pk.wh_off;
    if (op.wino && !wino_now(op, batch)) {           // small batch: the direct kernel on the direct-layout copy of the weights
        a.Wp = pack_arena + pk.w2_off;
        a.Kpad = pk.Kpad2;
    } else if (op.wino && pk.wino_skip) {
        a.Wp = nullptr;                              // no Winograd layout was packed: only a split-fp32 tile (Wp3) may take this launch
    }
    a.conv = op.conv;
    a.Cin = op.Cin; a.H = op.H; a.W = op.W; a.Ho = op.Ho; a.Wo = op.Wo;
    a.ks = op.ks; a.stride = op.stride; a.pad = op.pad;
    a.amap = op.amap; a.omap = op.omap; a.rmap = op.rmap;
    a.act = op.act;
    a.out_bf16 = op.out_bf16;
    if (planes && op.h2_role && a.x3_h2 && a.Wp3 && wino_now(op, batch)) {
        // planes between a BasicBlock's two convs: only where BOTH launches go to the two-fp16-piece tile at this batch
        const GemmArgs peer = gemm_args(ops[op.h2_peer], batch, false);
        GemmArgs me = a;
        me.conv = op.conv; me.Cin = op.Cin; me.H = op.H; me.W = op.W; me.Ho = op.Ho; me.Wo = op.Wo; me.ks = op.ks; me.stride = op.stride; me.pad = op.pad;
        me.omap = op.omap; me.rmap = op.rmap; me.act = op.act;
        if (peer.x3_h2 && peer.Wp3 && wino_now(ops[op.h2_peer], batch) && gemm_f32x3_wanted(me) && gemm_f32x3_wanted(peer)) {
            int* e = reinterpret_cast<int*>(ptr(op.h2_exps));
            if (op.h2_role == 1) a.h2_eout = e; else a.h2_ein = e;
        }
    }
    if (op.conv && op.in[1] >= 0) {                    // + bilinear_upsample(in[1]) behind the activation (build_cpn: lateral + upsampled path)
        a.up = ptr(op.in[1]);
        a.up_H = op.i0; a.up_W = op.i1;
        a.up_sh = op.Ho > 1 ? (float)(op.i0 - 1) / (float)(op.Ho - 1) : 0.f;
        a.up_sw = op.Wo > 1 ? (float)(op.i1 - 1) / (float)(op.Wo - 1) : 0.f;
    }
    static const bool splitk_on = [] { const char* e = diag_env("CAPF_SPLITK"); return !e || atoi(e) != 0; }();   // A/B runs only
    if ((op.conv || (op.kind == OP_GEMM && op.ln_w < 0 && op.res_param < 0)) && !op.bf16 && split_ws && lanes != 1 && splitk_on) {   // one stream: launches use the scratch one after the other
        a.split_ws = on_side_chain ? split_ws_side : split_ws;
        a.split_cnt = on_side_chain ? split_cnt_side : split_cnt;
        a.split_ws_elems = SPLIT_WS_ELEMS; a.split_cnt_elems = SPLIT_CNT_ELEMS;
    }
    if (op.ln_w >= 0) {
        a.ln_g = params[op.ln_w].ptr;
        a.ln_b = params[op.ln_b].ptr;
        a.ln_eps = op.eps;
    }
    return a;
}

bool Engine::bneck0_head(int i, int batch, int last_op, int m[4]) const {
    if (!use_bneck || !bf16() || i < 0 || i >= (int)ops.size() || ops[i].kind != OP_FORK || ops[i].i0 != 2 || ops[i].region < 0) return false;
    const int j = regions[ops[i].region].second;                 // the join; conv3 follows it
    if (j != i + 4 || j + 1 >= last_op || j + 1 >= (int)ops.size()) return false;
    int c1 = -1, c2 = -1, ds = -1;
    for (int k = i + 1; k < j; ++k) {
        const Op& o = ops[k];
        if (o.kind != OP_GEMM || !o.conv || o.bf16 != 1) return false;
        if (o.lane == 1) ds = k; else if (c1 < 0) c1 = k; else c2 = k;
    }
    const int c3 = j + 1;
    if (c1 < 0 || c2 < 0 || ds < 0 || ops[c3].kind != OP_GEMM || !ops[c3].conv || ops[c3].bf16 != 1) return false;
    if (ops[c2].in[0] != ops[c1].out || ops[ds].in[0] != ops[c1].in[0] || ops[c3].in[0] != ops[c2].out || ops[c3].aux != ops[ds].out) return false;
    // (from 64 Ki pixels: below, the 256 persistent blocks get fewer than four tiles each and the five launches win)
    if (ops[c1].rows_per_frame * batch < 65536) return false;
    m[0] = c1; m[1] = c2; m[2] = ds; m[3] = c3;
    return bneck0_bf16_ok(gemm_args(ops[c1], batch), gemm_args(ops[c2], batch), gemm_args(ops[ds], batch), gemm_args(ops[c3], batch));
}

int Engine::bneck0_member(int i, int batch) const {
    int m[4];
    for (int f = i - 5; f < i; ++f)
        if (f >= 0 && bneck0_head(f, batch, (int)ops.size(), m) && (i == m[0] || i == m[1] || i == m[2] || i == m[3])) return f;
    return -1;
}

bool Engine::bneck1_head(int i, int batch, int last_op) const {
    if (!use_bneck || !bf16() || i < 0 || i + 2 >= last_op || i + 2 >= (int)ops.size()) return false;
    const Op &c1 = ops[i], &c2 = ops[i + 1], &c3 = ops[i + 2];
    for (const Op* o : {&c1, &c2, &c3})
        if (o->kind != OP_GEMM || !o->conv || o->bf16 != 1 || o->region != c1.region || o->lane != c1.lane || o->in[1] >= 0) return false;
    if (c1.bneck_c3 != i + 2 || c2.in[0] != c1.out || c3.in[0] != c2.out || c3.aux != c1.in[0] || c1.aux >= 0 || c2.aux >= 0) return false;
    if (c1.rows_per_frame * batch < 65536) return false;
    return bneck1_bf16_ok(gemm_args(c1, batch), gemm_args(c2, batch), gemm_args(c3, batch));
}

int Engine::bneck1_member(int i, int batch) const {
    for (int f = i - 2; f <= i; ++f)
        if (f >= 0 && bneck1_head(f, batch, (int)ops.size())) return f;
    return -1;
}

bool Engine::pwchain_head(int i, int batch, int last_op) const {
    if (!use_pwchain || i < 0 || i + 1 >= last_op || i + 1 >= (int)ops.size()) return false;
    const Op& a = ops[i];
    const Op& b = ops[i + 1];
    if (a.kind != OP_GEMM || !a.conv || b.kind != OP_GEMM || !b.conv || a.bf16 != b.bf16 || a.bf16 > 1) return false;
    if (b.in[0] != a.out || b.region != a.region || b.lane != a.lane) return false;
    if (a.bf16 && (bneck0_member(i, batch) >= 0 || bneck1_member(i, batch) >= 0 || bneck1_head(i + 1, batch, last_op))) return false;   // (the conv3 of a
                                                      // fused bottleneck does not chain into the next block; a fused next block runs its own conv1)
    return a.bf16 ? gemm_bf16_pwchain_ok(gemm_args(a, batch), gemm_args(b, batch)) : gemm_f32_pwchain_ok(gemm_args(a, batch), gemm_args(b, batch));
}

FuseSumArgs Engine::fuse_args(const Op& op, int batch) const {
    FuseSumArgs a{};
    a.n_in = op.n_in;
    for (int i = 0; i < op.n_in; ++i) {
        a.in[i] = op.in[i] >= 0 ? bptr(op.in[i], batch) : nullptr;
        a.shift[i] = op.shift[i];
    }
    a.out = op.out >= 0 ? bptr(op.out, batch) : nullptr;
    a.B = batch; a.H = op.H; a.W = op.W; a.C = op.C; a.relu = op.relu;
    a.bf16 = op.bf16;
    return a;
}

// one non-control op on stream s
int Engine::exec_op(const Op& op, hipStream_t s, int batch) {
    auto ptr = [&](int buf) -> float* {
        if (buf >= 0) return bptr(buf, batch);
        return nullptr;
    };
    switch (op.kind) {
        case OP_GEMM: {
            if (op.bf16 == 2) {
                const GemmArgs a = gemm_args(op, batch);
                HIP_TRY(launch_gemm_bf16_rows(a.A, a.Wp, a.bias, a.M, a.N, a.K, a.Kpad, a.out, a.omap, a.res, a.rmap, op.out_bf16, s));
            } else if (op.bf16) HIP_TRY(launch_gemm_bf16(gemm_args(op, batch), s));
            else if (wino_now(op, batch)) HIP_TRY(launch_gemm_wino(gemm_args(op, batch), s));
            else HIP_TRY(launch_gemm_f32(gemm_args(op, batch), s));
            break;
        }
        case OP_FUSE: {
            if (op.i0 == 1 && !debug) break;
            HIP_TRY(launch_fuse_sum(fuse_args(op, batch), s));
            break;
        }
        case OP_MAXPOOL:
            HIP_TRY(launch_maxpool3x3s2(ptr(op.in[0]), ptr(op.out), batch, op.H, op.W, op.C, op.Ho, op.Wo, s, op.bf16));
            break;
        case OP_RESIZE:
            HIP_TRY(launch_bilinear_resize(ptr(op.in[0]), ptr(op.out), batch, op.H, op.W, op.C, op.Ho, op.Wo, s, op.bf16, ptr(op.aux)));
            break;
        case OP_PREP_EMBED:
            HIP_TRY(launch_prep_embed(kcrop, k2d, params[op.p0].ptr, params[op.p1].ptr, params[op.p2].ptr,
                                      ptr(op.out), batch, op.i0, op.i1, op.C, s));
            break;
        case OP_SAMPLE_REF:
            HIP_TRY(launch_sample_ref(ptr(op.in[0]), kcrop, ptr(op.out), reinterpret_cast<int*>(ptr(op.aux2)),
                                      batch, op.i0, op.H, op.W, op.C, s, op.bf16));
            break;
        case OP_LAYERNORM:
            HIP_TRY(launch_layernorm(ptr(op.in[0]), op.amap, ptr(op.aux), op.rmap, params[op.p0].ptr,
                                     params[op.p1].ptr, op.eps, ptr(op.out), (int)(op.rows_per_frame * batch),
                                     op.C, s, op.out_bf16));
            break;
        case OP_DEFORM: {
            DeformArgs a{};
            for (int l = 0; l < op.i1; ++l) {
                a.feat[l] = ptr(op.in[l]);
                a.H[l] = op.lvlH[l]; a.W[l] = op.lvlW[l]; a.C[l] = op.lvlC[l];
                a.U[l] = ptr(op.outs[l]);
            }
            a.AO = ptr(op.aux);
            a.ref = kcrop;
            a.B = batch; a.J = op.i0; a.L = op.i1; a.NH = op.i2; a.NS = op.i3;
            a.feat_bf16 = op.bf16;
            if (debug && op.idxs[0] >= 0) { a.cpos = ptr(op.idxs[0]); a.cidx = reinterpret_cast<int*>(ptr(op.idxs[1])); }
            HIP_TRY(launch_deform_sample(a, s));
            break;
        }
        case OP_EMBED: {
            EmbedArgs a{};
            a.kcrop = kcrop; a.k2d = k2d;
            a.cw = params[op.p0].ptr; a.cb = params[op.p1].ptr; a.pos = params[op.p2].ptr;
            for (int l = 0; l < op.i1; ++l) {
                a.feat[l] = ptr(op.in[l]);
                a.H[l] = op.lvlH[l]; a.W[l] = op.lvlW[l]; a.Cl[l] = op.lvlC[l];
                a.fw[l] = pack_arena + packs[op.pq[l]].w_off; a.fb[l] = params[op.pb[l]].ptr;
                a.sampled[l] = ptr(op.outs[l]);
                a.idx[l] = reinterpret_cast<int*>(ptr(op.idxs[l]));
            }
            a.X = ptr(op.out);
            a.BJ = batch * op.i0; a.J = op.i0; a.L = op.i1; a.L1 = op.i2; a.C = op.C;
            a.feat_bf16 = op.bf16;
            HIP_TRY(launch_embed(a, s));
            break;
        }
        case OP_CTX_ATTN: {
            CtxAttnArgs a{};
            const Pack& pk = packs[op.pack];
            for (int l = 0; l < op.i1; ++l) {
                a.feat[l] = ptr(op.in[l]);
                a.H[l] = op.lvlH[l]; a.W[l] = op.lvlW[l]; a.Cl[l] = op.lvlC[l];
                a.Wp[l] = pack_arena + packs[op.pq[l]].w_off; a.bp[l] = params[op.pb[l]].ptr;
                a.U[l] = op.outs[l] >= 0 ? ptr(op.outs[l]) : nullptr;
            }
            a.Wao = pack_arena + pk.w_off; a.bao = pack_arena + pk.b_off; a.ldw = pk.Kpad;
            a.ln_g = params[op.p0].ptr; a.ln_b = params[op.p1].ptr; a.eps = op.eps;
            a.ref = kcrop;
            a.X = ptr(op.out);
            a.BJ = batch * op.i0; a.J = op.i0; a.L = op.i1; a.L1 = op.i1 + 1; a.C = op.C; a.NH = op.i2; a.NS = op.i3;
            a.feat_bf16 = op.bf16;
            if (debug && op.idxs[0] >= 0) { a.cpos = ptr(op.idxs[0]); a.cidx = reinterpret_cast<int*>(ptr(op.idxs[1])); }
            HIP_TRY(launch_ctx_attn(a, s));
            break;
        }
        case OP_ATTENTION:
            HIP_TRY(launch_attention(ptr(op.in[0]), ptr(op.out), op.i0 * batch, op.i1, op.i2, op.i3, s, op.out_bf16));
            break;
        case OP_RES_CHAIN: {
            ResBlockW blk[8];
            for (int i = 0; i < op.i2; ++i) {
                const int* c = &op.chain[8 * i];
                const Pack *q = &packs[c[0]], *pr = &packs[c[1]], *f1 = &packs[c[2]], *f2 = &packs[c[3]];
                if (!q->chain || !pr->chain || !f1->chain || !f2->chain || q->KpadH != q->K || f2->KpadH != f2->K) return CAPF_ERR_STATE;
                blk[i] = ResBlockW{params[c[4]].ptr, params[c[5]].ptr, pack_arena + q->wc_off, params[q->b[0]].ptr, pack_arena + pr->wc_off,
                                   params[pr->b[0]].ptr, params[c[6]].ptr, params[c[7]].ptr, pack_arena + f1->wc_off, params[f1->b[0]].ptr,
                                   pack_arena + f2->wc_off, params[f2->b[0]].ptr};
            }
            HIP_TRY(launch_res_chain(ptr(op.out), (int)(op.rows_per_frame * batch), op.i0, op.i1, op.eps, blk, op.i2, s));
            break;
        }
        case OP_MLP_CHAIN: {
            const Pack *f1 = &packs[op.chain[0]], *f2 = &packs[op.chain[1]];
            if (!f1->chain || !f2->chain || f1->KpadH != f1->K || f2->KpadH != f2->K) return CAPF_ERR_STATE;
            ResBlockW w{};
            w.ln2_g = params[op.chain[2]].ptr; w.ln2_b = params[op.chain[3]].ptr;
            w.wfc1 = pack_arena + f1->wc_off; w.bfc1 = params[f1->b[0]].ptr;
            w.wfc2 = pack_arena + f2->wc_off; w.bfc2 = params[f2->b[0]].ptr;
            HIP_TRY(launch_mlp_chain(ptr(op.out), op.amap, (int)(op.rows_per_frame * batch), op.eps, w, s));
            break;
        }
        case OP_HEAD:
            HIP_TRY(launch_head(ptr(op.in[0]), params[op.p0].ptr, params[op.p1].ptr, op.eps, params[op.p2].ptr,
                                params[op.p3].ptr, out, (int)(op.rows_per_frame * batch), op.C, op.i0, s));
            break;
        default: break;
    }
    return CAPF_OK;
}

// Fork/join region as dependency levels on ONE stream: the fp32 convs of a level (one per HRNet branch, or
// the many small convs of a fuse layer) go out as grouped launches, everything else one by one.  Any
// topological order is valid on a single stream, and no two buffers of a region share memory
// (assign_offsets keeps them alive for the whole region).  With `log` every launch is bracketed by
// events (profiling): log gets (event index, leader op) pairs and member ops point at their leader.
int Engine::run_region_grouped(hipStream_t s, int batch, int region, LaunchLog* log, unsigned lane_mask) {
    auto mine = [&](const Op& op) { return ((lane_mask >> op.lane) & 1u) != 0; };
    auto groupable = [&](const Op& op, const GemmArgs& a) {
        if (op.kind != OP_GEMM) return false;
        if (op.bf16) return gemm_bf16_groupable(a);
        if (wino_now(op, batch)) return gemm_wino_ok(a);
        return gemm_f32_groupable(a);
    };
    for (const std::vector<int>& level : region_levels[region]) {
        for (int pass = 0; pass < 3; ++pass) {           // pass 0: direct fp32 convs, 1: bf16 convs, 2: Winograd fp32 convs
            GemmArgs group[MAXG];
            int members[MAXG];
            int n = 0;
            auto flush = [&]() -> int {
                if (n == 0) return CAPF_OK;
                if (log) HIP_TRY(log->mark(s, members, n));
                if (pass == 0) HIP_TRY(launch_gemm_f32_group(group, n, s));
                else if (pass == 1) {
                    int v = -1;
                    HIP_TRY(launch_gemm_bf16_group(group, n, s, &v));
                    if (log && !log->op_variant.empty()) log->op_variant[members[0]] = v;
                }
                else HIP_TRY(launch_gemm_wino_group(group, n, s));
                n = 0;
                return CAPF_OK;
            };
            for (int oi : level) {
                const Op& op = ops[oi];
                if (op.kind != OP_GEMM || !mine(op)) continue;
                const int kind = op.bf16 ? 1 : (wino_now(op, batch) ? 2 : 0);
                if (kind != pass) continue;
                const GemmArgs a = gemm_args(op, batch);
                if (!groupable(op, a)) continue;
                group[n] = a;
                members[n++] = oi;
                if (n == MAXG) { int rc = flush(); if (rc) return rc; }
            }
            int rc = flush();
            if (rc) return rc;
        }
        {   // the fuse sums of a level (schedule_regions gathers a module's sums in its last level): one launch
            FuseSumArgs fg[4];
            int fm[4], nf = 0;
            for (int oi : level) {
                const Op& op = ops[oi];
                if (op.kind != OP_FUSE || (op.i0 == 1 && !debug) || !mine(op)) continue;
                if (nf > 0 && (op.bf16 != ops[fm[0]].bf16 || (op.C % 8 == 0) != (ops[fm[0]].C % 8 == 0))) continue;
                if (nf == 4) break;
                fg[nf] = fuse_args(op, batch);
                fm[nf++] = oi;
            }
            if (nf < 2) nf = 0;
            if (nf) {
                if (log) HIP_TRY(log->mark(s, fm, nf));
                HIP_TRY(launch_fuse_sum_group(fg, nf, s));
            }
            for (int oi : level) {
                const Op& op = ops[oi];
                if (!mine(op)) continue;
                if (op.kind == OP_GEMM && groupable(op, gemm_args(op, batch))) continue;
                if (op.kind == OP_FUSE && op.i0 == 1 && !debug) continue;
                bool done = false;
                for (int k = 0; k < nf; ++k) done |= fm[k] == oi;
                if (done) continue;
                if (log) HIP_TRY(log->mark(s, &oi, 1));
                int rc = exec_op(op, s, batch);
                if (rc) return rc;
            }
        }
    }
    return CAPF_OK;
}

// ev: per-op events (profiling, everything in program order on one stream).  log: per-launch events of
// the product schedule (grouped launches included).
int Engine::run(hipStream_t s, int batch, int first_op, int last_op, hipEvent_t* ev, LaunchLog* log) {
    hipStream_t main_stream = s;
    if (use_h2g && (batch >= H2G_MIN_BATCH || has_res_chain) && last_op > n_backbone_ops && !bf16()) {      // (the fused res blocks read the packs at every batch)
        const int rc = ensure_h2g_lifter(s);
        if (rc) return rc;
    }
    const bool grouped = (lanes == 2 || lanes == 3) && !ev;
    const bool par = lanes == 1 && !ev && !log && side[0];
    for (int oi = first_op; oi < last_op; ++oi) {
        const Op& op = ops[oi];
        s = (par && op.lane > 0) ? side[op.lane - 1] : main_stream;
        if (ev) HIP_TRY(hipEventRecord(ev[oi], s));
        switch (op.kind) {
            case OP_FORK: {
                int bm[4];
                if (!ev && bneck0_head(oi, batch, last_op, bm)) {
                    // a first bottleneck as one launch; prefix runs (the layer-wise tests' capf_forward_prefix) and debug runs take the variant
                    // that also stores conv1's / conv2's / the shortcut's outputs where the five launches would have
                    if (log) HIP_TRY(log->mark(main_stream, bm, 4));
                    HIP_TRY(launch_bneck0_bf16(gemm_args(ops[bm[0]], batch), gemm_args(ops[bm[1]], batch), gemm_args(ops[bm[2]], batch),
                                               gemm_args(ops[bm[3]], batch), debug || last_op < (int)ops.size(), main_stream));
                    oi = bm[3];
                    break;
                }
                if (grouped && regions[op.region].second <= last_op) {
                    // lanes == 3: the lanes of a region as TWO grouped chains on two streams (lanes 0 + 3 on the caller's, 1 + 2 on a
                    // side stream), so that one chain's launch ramp / tail overlaps the other's body.  Measured at batch 64 (one
                    // box, frames/s): one chain 6418; {0,3}|{1,2} 6497; {0,1}|{2,3} 5964; {0,2}|{1,3} 6101; {0,1,2}|{3} 6149;
                    // {0}|{1,2,3} 6398.  Across the configurations: cfg1 +0.4..1.2 %, cfg2 +0.9 %, cfg4 +1.3 %, but -0.8 % at batch 512 (every launch already fills
                    // the chip many times over), and below batch 16 a region is a handful of tiles (and may use the split-K scratch):
                    // one chain outside 16..256.
                    const bool two = lanes == 3 && !log && side[0] && op.i0 >= 2 && batch >= 16 && batch <= 256;
                    if (two) {
                        HIP_TRY(hipEventRecord(events[op.i1], main_stream));
                        HIP_TRY(hipStreamWaitEvent(side[0], events[op.i1], 0));
                        on_side_chain = true;
                        int rc = run_region_grouped(side[0], batch, op.region, nullptr, 0x6u);
                        on_side_chain = false;
                        if (rc) return rc;
                        rc = run_region_grouped(main_stream, batch, op.region, nullptr, ~0x6u);
                        if (rc) return rc;
                        HIP_TRY(hipEventRecord(events[op.i1 + 1], side[0]));
                        HIP_TRY(hipStreamWaitEvent(main_stream, events[op.i1 + 1], 0));
                        oi = regions[op.region].second;
                        break;
                    }
                    int rc = run_region_grouped(main_stream, batch, op.region, log);
                    if (rc) return rc;
                    oi = regions[op.region].second;      // continue after the join
                } else if (par) {
                    HIP_TRY(hipEventRecord(events[op.i1], main_stream));
                    for (int l = 1; l < op.i0; ++l) HIP_TRY(hipStreamWaitEvent(side[l - 1], events[op.i1], 0));
                }
                break;
            }
            case OP_JOIN:
                if (par) {
                    for (int l = 1; l < op.i0; ++l) {
                        HIP_TRY(hipEventRecord(events[op.i1 + l], side[l - 1]));
                        HIP_TRY(hipStreamWaitEvent(main_stream, events[op.i1 + l], 0));
                    }
                }
                break;
            default: {
                if (op.kind == OP_FUSE && op.i0 == 1 && !debug) break;
                if (!ev && op.kind == OP_GEMM && bneck1_head(oi, batch, last_op)) {      // an identity bottleneck as one launch (bneck_bf16.hip, DS = false)
                    const int tri[3] = {oi, oi + 1, oi + 2};
                    if (log) HIP_TRY(log->mark(s, tri, 3));
                    HIP_TRY(launch_bneck1_bf16(gemm_args(op, batch), gemm_args(ops[oi + 1], batch), gemm_args(ops[oi + 2], batch),
                                               debug || last_op < (int)ops.size(), s));
                    oi += 2;
                    break;
                }
                // a 64 -> 256 pointwise conv directly followed by the 256 -> 64 one that reads it (layer1's conv3 -> next conv1): one launch
                if (!ev && pwchain_head(oi, batch, last_op)) {
                    const int pair[2] = {oi, oi + 1};
                    if (log) HIP_TRY(log->mark(s, pair, 2));
                    if (op.bf16) HIP_TRY(launch_gemm_bf16_pwchain(gemm_args(op, batch), gemm_args(ops[oi + 1], batch), s));
                    else HIP_TRY(launch_gemm_f32_pwchain(gemm_args(op, batch), gemm_args(ops[oi + 1], batch), s));
                    ++oi;
                    break;
                }
                if (log) HIP_TRY(log->mark(s, &oi, 1));
                int rc = exec_op(op, s, batch);
                if (rc) return rc;
            }
        }
    }
    if (ev) HIP_TRY(hipEventRecord(ev[last_op], main_stream));
    if (log) HIP_TRY(log->mark(main_stream, nullptr, 0));
    return CAPF_OK;
}

}  // namespace capf

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
using capf::Engine;

static std::string g_create_error;

extern "C" {

const char* capf_version(void) { return "capf 0.5 (gfx950)"; }
int capf_abi_version(void) { return CAPF_ABI_VERSION; }

const char* capf_last_error(const capf_handle* h) { return h ? h->e.err.c_str() : g_create_error.c_str(); }

int capf_create(const capf_config* cfg, int device, capf_handle** out) {
    if (!cfg || !out) {
        g_create_error = "null argument";
        return CAPF_ERR_INVALID;
    }
    capf_handle* h = new capf_handle();
    Engine& e = h->e;
    e.cfg = *cfg;
    e.device = device;
    e.lanes = 2;
    if (cfg->compute_dtype != CAPF_F32 && cfg->compute_dtype != CAPF_BF16) {
        g_create_error = "compute_dtype must be CAPF_F32 or CAPF_BF16";
        delete h;
        return CAPF_ERR_UNSUPPORTED;
    }
    if (cfg->plan_flags & ~16383) {
        g_create_error = "unknown capf_plan_flag bits";
        delete h;
        return CAPF_ERR_INVALID;
    }
    if (cfg->max_batch <= 0) {
        g_create_error = "max_batch must be positive";
        delete h;
        return CAPF_ERR_INVALID;
    }
    if (!e.build()) {
        g_create_error = e.err;
        delete h;
        return CAPF_ERR_UNSUPPORTED;
    }
    if (device >= 0) {
        hipError_t r = hipSetDevice(device);
        if (r == hipSuccess && e.pack_elems) r = hipMalloc(reinterpret_cast<void**>(&e.pack_arena), e.pack_elems * sizeof(float));
        if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void**>(&e.split_ws), Engine::SPLIT_WS_ELEMS * sizeof(float));
        if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void**>(&e.split_cnt), Engine::SPLIT_CNT_ELEMS * sizeof(int));
        if (r == hipSuccess) r = hipMemset(e.split_cnt, 0, Engine::SPLIT_CNT_ELEMS * sizeof(int));
        if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void**>(&e.split_ws_side), Engine::SPLIT_WS_ELEMS * sizeof(float));
        if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void**>(&e.split_cnt_side), Engine::SPLIT_CNT_ELEMS * sizeof(int));
        if (r == hipSuccess) r = hipMemset(e.split_cnt_side, 0, Engine::SPLIT_CNT_ELEMS * sizeof(int));
        for (int i = 0; i < 3 && r == hipSuccess; ++i) r = hipStreamCreateWithFlags(&e.side[i], hipStreamNonBlocking);
        e.events.resize(e.n_events);
        for (auto& x : e.events)
            if (r == hipSuccess) r = hipEventCreateWithFlags(&x, hipEventDisableTiming);
        if (r != hipSuccess) {
            g_create_error = std::string("hip: ") + hipGetErrorString(r);
            delete h;
            return CAPF_ERR_HIP;
        }
    }
    *out = h;
    return CAPF_OK;
}

void capf_destroy(capf_handle* h) {
    if (!h) return;
    if (h->e.pack_arena) (void)hipFree(h->e.pack_arena);
    if (h->e.split_ws) (void)hipFree(h->e.split_ws);
    if (h->e.split_cnt) (void)hipFree(h->e.split_cnt);
    if (h->e.split_ws_side) (void)hipFree(h->e.split_ws_side);
    if (h->e.split_cnt_side) (void)hipFree(h->e.split_cnt_side);
    for (auto& x : h->e.events)
        if (x) (void)hipEventDestroy(x);
    for (auto& st : h->e.side)
        if (st) (void)hipStreamDestroy(st);
    delete h;
}

int capf_max_batch(const capf_handle* h) { return h ? (h->e.batch_limit < h->e.cfg.max_batch ? h->e.batch_limit : h->e.cfg.max_batch) : CAPF_ERR_INVALID; }

int capf_num_params(const capf_handle* h) { return h ? (int)h->e.params.size() : CAPF_ERR_INVALID; }

int capf_param_info(const capf_handle* h, int index, const char** name, int64_t shape[4], int* ndim, int* kind) {
    if (!h || index < 0 || index >= (int)h->e.params.size()) return CAPF_ERR_INVALID;
    const capf::Param& p = h->e.params[index];
    if (name) *name = p.name.c_str();
    if (shape) memcpy(shape, p.shape, sizeof(p.shape));
    if (ndim) *ndim = p.ndim;
    if (kind) *kind = p.kind;
    return CAPF_OK;
}

int capf_set_param(capf_handle* h, const char* name, const void* dev_ptr, const int64_t* shape, int ndim) {
    if (!h || !name) return CAPF_ERR_INVALID;
    Engine& e = h->e;
    auto it = e.param_index.find(name);
    if (it == e.param_index.end()) {
        e.err = std::string("unknown parameter: ") + name;
        return CAPF_ERR_INVALID;
    }
    capf::Param& p = e.params[it->second];
    bool ok = (ndim == p.ndim);
    for (int i = 0; ok && i < ndim; ++i) ok = (shape[i] == p.shape[i]);
    if (!ok) {
        e.err = std::string("shape mismatch for ") + name;
        return CAPF_ERR_INVALID;
    }
    p.ptr = static_cast<const float*>(dev_ptr);
    e.packed = false;
    return CAPF_OK;
}

int capf_lifter_params_changed(capf_handle* h, void* stream) {
    if (!h) return CAPF_ERR_INVALID;
    if (h->e.device < 0 || !h->e.packed) {
        h->e.err = "capf_params_changed must have run once before capf_lifter_params_changed";
        return CAPF_ERR_STATE;
    }
    return h->e.repack(static_cast<hipStream_t>(stream), true);
}

int capf_params_changed(capf_handle* h, void* stream) {
    if (!h) return CAPF_ERR_INVALID;
    if (h->e.device < 0) {
        h->e.err = "plan-only handle (device < 0)";
        return CAPF_ERR_STATE;
    }
    return h->e.repack(static_cast<hipStream_t>(stream));
}

size_t capf_workspace_bytes(const capf_handle* h, int batch) {
    if (!h || batch <= 0) return 0;
    size_t n = h->e.ws_elems_per_frame * (size_t)batch;
    if (h->e.cfg.training) n += h->e.train_elems(batch);
    return n * sizeof(float);
}

int capf_set_workspace(capf_handle* h, void* dev_ptr, size_t bytes) {
    if (!h) return CAPF_ERR_INVALID;
    h->e.ws = static_cast<float*>(dev_ptr);
    h->e.ws_bytes = bytes;
    h->e.invalidate_train();
    return CAPF_OK;
}

int capf_set_lanes(capf_handle* h, int on) {
    if (!h) return CAPF_ERR_INVALID;
    if (on < 0 || on > 3) return CAPF_ERR_INVALID;
    h->e.lanes = on;
    return CAPF_OK;
}

int capf_set_debug(capf_handle* h, int on) {
    if (!h) return CAPF_ERR_INVALID;
    h->e.debug = on != 0;
    return CAPF_OK;
}

static size_t capf_workspace_bytes_impl(const Engine& e, int batch) {
    size_t n = e.ws_elems_per_frame * (size_t)batch;
    if (e.cfg.training) n += e.train_elems(batch);
    return n * sizeof(float);
}

static int check_run(Engine& e, int batch) {
    if (e.device < 0) {
        e.err = "plan-only handle (device < 0)";
        return CAPF_ERR_STATE;
    }
    if (batch <= 0 || batch > e.cfg.max_batch) {
        e.err = "batch out of range (1..max_batch)";
        return CAPF_ERR_INVALID;
    }
    if (batch > e.batch_limit) {
        e.err = "batch too large for this input size: the kernels address one activation tensor with 32-bit offsets (limit " +
                std::to_string(e.batch_limit) + " frames)";
        return CAPF_ERR_UNSUPPORTED;
    }
    if (!e.packed) {
        e.err = "capf_params_changed has not been called since the last capf_set_param";
        return CAPF_ERR_STATE;
    }
    if (!e.ws || e.ws_bytes < capf_workspace_bytes_impl(e, batch)) {
        e.err = "workspace missing or too small";
        return CAPF_ERR_STATE;
    }
    return CAPF_OK;
}

int capf_forward(capf_handle* h, void* stream, const float* images_nhwc, const float* k2d, float* kcrop_inout,
                 int batch, float* out) {
    if (!h || !images_nhwc || !k2d || !kcrop_inout || !out) return CAPF_ERR_INVALID;
    Engine& e = h->e;
    int rc = check_run(e, batch);
    if (rc) return rc;
    e.images = images_nhwc; e.k2d = k2d; e.kcrop = kcrop_inout; e.out = out;
    e.last_batch = batch;
    e.invalidate_train();
    return e.run(static_cast<hipStream_t>(stream), batch, 0, (int)e.ops.size());
}

int capf_backbone_forward(capf_handle* h, void* stream, const float* images_nhwc, int batch) {
    if (!h || !images_nhwc) return CAPF_ERR_INVALID;
    Engine& e = h->e;
    int rc = check_run(e, batch);
    if (rc) return rc;
    e.images = images_nhwc;
    e.last_batch = batch;
    e.invalidate_train();
    return e.run(static_cast<hipStream_t>(stream), batch, 0, e.n_backbone_ops);
}

int capf_forward_train(capf_handle* h, void* stream, const float* images_nhwc, const float* k2d, float* kcrop_inout,
                       int batch, float* out, const float* drop_masks) {
    if (!h || !images_nhwc || !k2d || !kcrop_inout || !out) return CAPF_ERR_INVALID;
    Engine& e = h->e;
    if (!e.cfg.training) {
        e.err = "handle was created with training = 0";
        return CAPF_ERR_STATE;
    }
    int rc = check_run(e, batch);
    if (rc) return rc;
    e.images = images_nhwc; e.k2d = k2d; e.kcrop = kcrop_inout; e.out = out;
    e.last_batch = batch;
    hipStream_t s = static_cast<hipStream_t>(stream);
    e.invalidate_train();
    rc = e.run(s, batch, 0, e.n_backbone_ops);
    if (rc) return rc;
    return e.forward_train(s, batch, drop_masks);
}

int capf_backward(capf_handle* h, void* stream, const float* grad_out, int batch, float* flat_grad, const float* drop_masks) {
    if (!h || !grad_out || !flat_grad) return CAPF_ERR_INVALID;
    return h->e.backward(static_cast<hipStream_t>(stream), batch, grad_out, flat_grad, drop_masks);
}

int64_t capf_train_generation(const capf_handle* h) { return h ? h->e.train_generation : -1; }

int64_t capf_grad_elems(const capf_handle* h) { return h ? h->e.grad_elems : -1; }

int capf_train_h2_matrices(const capf_handle* h) { return h ? (int)h->e.t_h2_specs.size() : 0; }

int capf_grad_info(const capf_handle* h, int index, int64_t* offset) {
    if (!h || index < 0 || index >= (int)h->e.params.size() || !offset) return CAPF_ERR_INVALID;
    *offset = h->e.grad_off[index];
    return CAPF_OK;
}

int capf_mpjpe(void* stream, const float* pred, const float* gt, int rows, float* loss, float* dpred, float grad_scale) {
    if (!pred || !gt || !loss || rows <= 0) return CAPF_ERR_INVALID;
    return capf::launch_mpjpe(pred, gt, rows, loss, dpred, grad_scale, static_cast<hipStream_t>(stream)) == hipSuccess
               ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_mpjpe_nd(void* stream, const float* pred, const float* gt, int rows, int dim, float* loss, float* dpred, float grad_scale) {
    if (!pred || !gt || !loss || rows <= 0 || dim <= 0) return CAPF_ERR_INVALID;
    return capf::launch_mpjpe_nd(pred, gt, rows, dim, loss, dpred, grad_scale, static_cast<hipStream_t>(stream)) == hipSuccess
               ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_adamw_step(void* stream, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return CAPF_ERR_INVALID;
    return capf::launch_adamw(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step,
                              static_cast<hipStream_t>(stream), grad_scale) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_lifter_forward(capf_handle* h, void* stream, const float* k2d, float* kcrop_inout, int batch, float* out) {
    if (!h || !k2d || !kcrop_inout || !out) return CAPF_ERR_INVALID;
    Engine& e = h->e;
    int rc = check_run(e, batch);
    if (rc) return rc;
    e.k2d = k2d; e.kcrop = kcrop_inout; e.out = out;
    e.last_batch = batch;
    e.invalidate_train();
    return e.run(static_cast<hipStream_t>(stream), batch, e.n_backbone_ops, (int)e.ops.size());
}

int capf_tensor(const capf_handle* h, const char* name, const void** dev_ptr, int64_t shape[4], int* ndim) {
    if (!h || !name) return CAPF_ERR_INVALID;
    const Engine& e = h->e;
    auto it = e.named.find(name);
    if (it == e.named.end()) return CAPF_ERR_INVALID;
    const capf::NamedTensor& t = it->second;
    if (shape) {
        for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
        shape[0] = e.last_batch;
    }
    if (ndim) *ndim = t.ndim;
    if (dev_ptr) *dev_ptr = (e.ws && e.last_batch > 0) ? e.bptr(t.buf, e.last_batch) : nullptr;
    return t.is_int;      // 0 fp32, 1 int32, 2 bf16
}

int capf_op_pack_conv(void* stream, const float* w, const float* gamma, const float* beta, const float* mean,
                      const float* var, float eps, float* wp, float* bias, int Cout, int Cin, int ks) {
    const int Kpad = (ks * ks * Cin + 31) / 32 * 32;
    return capf::launch_pack_conv(w, gamma, beta, mean, var, eps, wp, bias, Cout, Cin, ks, Kpad,
                                  static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_conv(void* stream, const float* x, const float* wp, const float* bias, const float* residual, float* y,
                 int B, int H, int W, int Cin, int Cout, int ks, int stride, int act) {
    capf::GemmArgs a{};
    const int pad = ks / 2;
    a.A = x; a.Wp = wp; a.bias = bias; a.res = residual; a.out = y;
    a.Ho = (H + 2 * pad - ks) / stride + 1;
    a.Wo = (W + 2 * pad - ks) / stride + 1;
    a.M = B * a.Ho * a.Wo; a.N = Cout; a.K = ks * ks * Cin; a.Kpad = (a.K + 31) / 32 * 32;
    a.conv = 1; a.Cin = Cin; a.H = H; a.W = W; a.ks = ks; a.stride = stride; a.pad = pad;
    a.omap = capf::row_ld(Cout); a.rmap = capf::row_ld(Cout); a.amap = capf::row_ld(0);
    a.act = act;
    return capf::launch_gemm_f32(a, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_pack_conv_wino(void* stream, const float* w, const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, float* wp, float* bias, int Cout, int Cin, int variant) {
    return capf::launch_pack_conv_wino(w, gamma, beta, mean, var, eps, wp, bias, Cout, Cin, static_cast<hipStream_t>(stream), variant) == hipSuccess
               ? CAPF_OK : CAPF_ERR_UNSUPPORTED;
}

static bool wino_desc(capf::GemmArgs& a, const float* x, const float* wp, const float* bias, const float* residual, float* y, int B,
                      int H, int W, int Cin, int Cout, int act, int variant) {
    a = capf::GemmArgs{};
    a.A = x; a.Wp = wp; a.bias = bias; a.res = residual; a.out = y;
    a.Ho = H; a.Wo = W; a.M = B * H * W; a.N = Cout; a.K = 9 * Cin; a.Kpad = (variant == 43 ? 18 : 12) * Cin;
    a.conv = 1; a.Cin = Cin; a.H = H; a.W = W; a.ks = 3; a.stride = 1; a.pad = 1;
    a.omap = capf::row_ld(Cout); a.rmap = capf::row_ld(Cout); a.amap = capf::row_ld(0);
    a.act = act;
    return capf::gemm_wino_ok(a);
}

int capf_op_conv_wino(void* stream, const float* x, const float* wp, const float* bias, const float* residual, float* y, int B,
                      int H, int W, int Cin, int Cout, int act, int variant) {
    capf::GemmArgs a;
    if ((variant != 23 && variant != 43) || !wino_desc(a, x, wp, bias, residual, y, B, H, W, Cin, Cout, act, variant)) return CAPF_ERR_UNSUPPORTED;
    return capf::launch_gemm_wino(a, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_conv_wino_group(void* stream, int n, const capf_conv_desc* d, int variant) {
    if (n <= 0 || n > capf::MAXG || !d || (variant != 23 && variant != 43)) return CAPF_ERR_INVALID;
    capf::GemmArgs g[capf::MAXG];
    for (int i = 0; i < n; ++i)
        if (d[i].ks != 3 || d[i].stride != 1 ||
            !wino_desc(g[i], d[i].x, d[i].w_packed, d[i].bias, d[i].residual, d[i].y, d[i].B, d[i].H, d[i].W, d[i].Cin, d[i].Cout, d[i].act, variant))
            return CAPF_ERR_UNSUPPORTED;
    return capf::launch_gemm_wino_group(g, n, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_conv_group(void* stream, int n, const capf_conv_desc* d) {
    if (n <= 0 || n > capf::MAXG || !d) return CAPF_ERR_INVALID;
    capf::GemmArgs g[capf::MAXG];
    for (int i = 0; i < n; ++i) {
        capf::GemmArgs a{};
        const int pad = d[i].ks / 2;
        a.A = d[i].x; a.Wp = d[i].w_packed; a.bias = d[i].bias; a.res = d[i].residual; a.out = d[i].y;
        a.Ho = (d[i].H + 2 * pad - d[i].ks) / d[i].stride + 1;
        a.Wo = (d[i].W + 2 * pad - d[i].ks) / d[i].stride + 1;
        a.M = d[i].B * a.Ho * a.Wo; a.N = d[i].Cout; a.K = d[i].ks * d[i].ks * d[i].Cin; a.Kpad = (a.K + 31) / 32 * 32;
        a.conv = 1; a.Cin = d[i].Cin; a.H = d[i].H; a.W = d[i].W; a.ks = d[i].ks; a.stride = d[i].stride; a.pad = pad;
        a.omap = capf::row_ld(d[i].Cout); a.rmap = capf::row_ld(d[i].Cout); a.amap = capf::row_ld(0);
        a.act = d[i].act;
        if (!capf::gemm_f32_groupable(a)) return CAPF_ERR_UNSUPPORTED;
        g[i] = a;
    }
    return capf::launch_gemm_f32_group(g, n, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int64_t capf_op_f32h2_gemm_pack_elems(int N, int K) { return N > 0 && K > 0 ? capf::f32h2_gemm_pack_elems(N, (K + 31) / 32 * 32) : 0; }

int capf_op_pack_f32h2_gemm(void* stream, const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                            float eps, float* wp, float* bias, int N, int Cin, int ks, int K) {
    if (!w || !wp || N <= 0 || (N & 3) || K <= 0 || (ks > 0 && K != ks * ks * Cin)) return CAPF_ERR_UNSUPPORTED;
    return capf::launch_pack_f32h2_gemm(w, gamma, beta, mean, var, eps, wp, bias, N, Cin, ks, K, (K + 31) / 32 * 32,
                                        static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

static capf::GemmArgs h2g_conv_args(const float* x, const float* wp, const float* bias, const float* residual, float* y, int B, int H, int W,
                                    int Cin, int Cout, int ks, int stride, int act) {
    capf::GemmArgs a{};
    const int pad = ks / 2;
    a.A = x; a.Wh2 = wp; a.bias = bias; a.res = residual; a.out = y;
    a.Ho = (H + 2 * pad - ks) / stride + 1;
    a.Wo = (W + 2 * pad - ks) / stride + 1;
    a.M = B * a.Ho * a.Wo; a.N = Cout; a.K = ks * ks * Cin; a.Kpad = (a.K + 31) / 32 * 32;
    a.conv = 1; a.Cin = Cin; a.H = H; a.W = W; a.ks = ks; a.stride = stride; a.pad = pad;
    a.omap = capf::row_ld(Cout); a.rmap = capf::row_ld(Cout); a.amap = capf::row_ld(0);
    a.act = act;
    return a;
}

int capf_op_conv_f32h2g(void* stream, const float* x, const float* wp, const float* bias, const float* residual, float* y,
                        int B, int H, int W, int Cin, int Cout, int ks, int stride, int act) {
    if (ks < 1 || stride < 1) return CAPF_ERR_INVALID;
    const capf::GemmArgs a = h2g_conv_args(x, wp, bias, residual, y, B, H, W, Cin, Cout, ks, stride, act);
    if (!capf::gemm_f32h2g_ok(a)) return CAPF_ERR_UNSUPPORTED;
    return capf::launch_gemm_f32h2g(a, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_conv_f32h2g_group(void* stream, int n, const capf_conv_desc* d) {
    if (n <= 0 || n > capf::MAXG || !d) return CAPF_ERR_INVALID;
    capf::GemmArgs g[capf::MAXG];
    for (int i = 0; i < n; ++i) {
        if (d[i].ks < 1 || d[i].stride < 1) return CAPF_ERR_INVALID;
        g[i] = h2g_conv_args(static_cast<const float*>(d[i].x), static_cast<const float*>(d[i].w_packed), d[i].bias,
                             static_cast<const float*>(d[i].residual), static_cast<float*>(d[i].y), d[i].B, d[i].H, d[i].W, d[i].Cin, d[i].Cout,
                             d[i].ks, d[i].stride, d[i].act);
        if (!capf::gemm_f32h2g_ok(g[i])) return CAPF_ERR_UNSUPPORTED;
    }
    return capf::launch_gemm_f32h2g_group(g, n, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_linear_f32h2g(void* stream, const float* x, const float* wp, const float* bias, const float* residual, float* y,
                          int M, int N, int K, int act) {
    if (K % 32 != 0) return CAPF_ERR_UNSUPPORTED;
    capf::GemmArgs a{};
    a.A = x; a.Wh2 = wp; a.bias = bias; a.res = residual; a.out = y;
    a.M = M; a.N = N; a.K = K; a.Kpad = K;
    a.amap = capf::row_ld(K); a.omap = capf::row_ld(N); a.rmap = capf::row_ld(N);
    a.act = act;
    if (!capf::gemm_f32h2g_ok(a)) return CAPF_ERR_UNSUPPORTED;
    return capf::launch_gemm_f32h2g(a, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_wgrad(void* stream, const float* dY, const float* X, int M, int N, int K, float* dw_db, int two_piece) {
    if (!dY || !X || !dw_db || M <= 0 || N <= 0 || K <= 0 || N % 4 || K % 4 || (two_piece && (N % 128 || K % 128))) return CAPF_ERR_UNSUPPORTED;
    return capf::launch_wgrad_tn(dY, N, X, K, M, N, K, dw_db, (long)N * K + N, 1, 1, static_cast<hipStream_t>(stream), two_piece != 0) == hipSuccess
               ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_linear_ln_f32h2g(void* stream, const float* x, const float* ln_gamma, const float* ln_beta, float eps, const float* wp,
                             const float* bias, const float* residual, float* y, int M, int N, int K, int act) {
    if (K % 32 != 0 || !ln_gamma || !ln_beta) return CAPF_ERR_UNSUPPORTED;
    capf::GemmArgs a{};
    a.A = x; a.Wh2 = wp; a.bias = bias; a.res = residual; a.out = y;
    a.M = M; a.N = N; a.K = K; a.Kpad = K;
    a.amap = capf::row_ld(K); a.omap = capf::row_ld(N); a.rmap = capf::row_ld(N);
    a.act = act;
    a.ln_g = ln_gamma; a.ln_b = ln_beta; a.ln_eps = eps;
    if (!capf::gemm_f32h2g_ok(a)) return CAPF_ERR_UNSUPPORTED;
    return capf::launch_gemm_f32h2g(a, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_linear(void* stream, const float* x, const float* w, const float* bias, const float* residual, float* y,
                   int M, int N, int K, int act) {
    if (K % 32 != 0) return CAPF_ERR_UNSUPPORTED;
    capf::GemmArgs a{};
    a.A = x; a.Wp = w; a.bias = bias; a.res = residual; a.out = y;
    a.M = M; a.N = N; a.K = K; a.Kpad = K;
    a.amap = capf::row_ld(K); a.omap = capf::row_ld(N); a.rmap = capf::row_ld(N);
    a.act = act;
    return capf::launch_gemm_f32(a, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_bilinear_corners(void* stream, const float* grid, int n, int H, int W, int border, int32_t* idx, float* frac) {
    if (!grid || !idx || !frac || n <= 0 || H <= 0 || W <= 0) return CAPF_ERR_INVALID;
    return capf::launch_bilinear_corners(grid, n, H, W, border, idx, frac, static_cast<hipStream_t>(stream)) == hipSuccess
               ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_pack_conv_bf16(void* stream, const float* w, const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, void* wp, float* bias, int Cout, int Cin, int ks) {
    const int Kpad = (ks * ks * Cin + 63) / 64 * 64;
    return capf::launch_pack_conv_bf16(w, gamma, beta, mean, var, eps, wp, bias, Cout, Cin, ks, Kpad,
                                       static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_conv_bf16(void* stream, const void* x, const void* wp, const float* bias, const void* residual, void* y, int B,
                      int H, int W, int Cin, int Cout, int ks, int stride, int act) {
    capf::GemmArgs a{};
    const int pad = ks / 2;
    a.A = static_cast<const float*>(x); a.Wp = static_cast<const float*>(wp); a.bias = bias;
    a.res = static_cast<const float*>(residual); a.out = static_cast<float*>(y);
    a.Ho = (H + 2 * pad - ks) / stride + 1;
    a.Wo = (W + 2 * pad - ks) / stride + 1;
    a.M = B * a.Ho * a.Wo; a.N = Cout; a.K = ks * ks * Cin; a.Kpad = (a.K + 63) / 64 * 64;
    a.conv = 1; a.Cin = Cin; a.H = H; a.W = W; a.ks = ks; a.stride = stride; a.pad = pad;
    a.omap = capf::row_ld(Cout); a.rmap = capf::row_ld(Cout); a.amap = capf::row_ld(0);
    a.act = act;
    return capf::launch_gemm_bf16(a, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_UNSUPPORTED;
}

static capf::GemmArgs rh_args(const void* x, const void* wp, const float* bias, const void* residual, void* y, int B, int H,
                              int W, int Cin, int Cout, int act) {
    capf::GemmArgs a{};
    a.A = static_cast<const float*>(x); a.Wp = static_cast<const float*>(wp); a.bias = bias;
    a.res = static_cast<const float*>(residual); a.out = static_cast<float*>(y);
    a.Ho = H; a.Wo = W;
    a.M = B * H * W; a.N = Cout; a.K = 9 * Cin; a.Kpad = 9 * Cin;
    a.conv = 1; a.Cin = Cin; a.H = H; a.W = W; a.ks = 3; a.stride = 1; a.pad = 1;
    a.omap = capf::row_ld(Cout); a.rmap = capf::row_ld(Cout); a.amap = capf::row_ld(0);
    a.act = act;
    return a;
}

int capf_op_conv_bf16_rh_width(int Cin) {
    capf::GemmArgs a = rh_args(nullptr, nullptr, nullptr, nullptr, nullptr, 1, 8, 8, Cin, 8, 0);
    return capf::gemm_bf16_rh_cw(a);
}

int capf_op_pack_conv_bf16_rh(void* stream, const float* w, const float* gamma, const float* beta, const float* mean,
                              const float* var, float eps, void* wp, float* bias, int Cout, int Cin) {
    const int cw = capf_op_conv_bf16_rh_width(Cin);
    if (!cw) return CAPF_ERR_UNSUPPORTED;
    return capf::launch_pack_conv_bf16_rh(w, gamma, beta, mean, var, eps, wp, bias, Cout, Cin, cw,
                                          static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_conv_bf16_rh(void* stream, const void* x, const void* wp, const float* bias, const void* residual, void* y, int B,
                         int H, int W, int Cin, int Cout, int act) {
    const capf::GemmArgs a = rh_args(x, wp, bias, residual, y, B, H, W, Cin, Cout, act);
    return capf::launch_gemm_bf16_rh(a, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_UNSUPPORTED;
}

int64_t capf_op_conv_bf16_ws_pack_elems(int Cout, int Cin) { return Cin % 16 == 0 && Cout > 0 ? capf::bf16_ws_pack_elems(Cout, Cin) : 0; }

int capf_op_pack_conv_bf16_ws(void* stream, const float* w, const float* gamma, const float* beta, const float* mean,
                              const float* var, float eps, void* wp, float* bias, int Cout, int Cin) {
    if (!w || !wp || Cin % 16 != 0 || Cout % 8 != 0) return CAPF_ERR_UNSUPPORTED;
    return capf::launch_pack_conv_bf16_ws(w, gamma, beta, mean, var, eps, wp, bias, Cout, Cin, static_cast<hipStream_t>(stream)) ==
                   hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_conv_bf16_ws_group(void* stream, int n, const capf_conv_desc* d) {
    if (n <= 0 || n > capf::MAXG || !d) return CAPF_ERR_INVALID;
    capf::GemmArgs g[capf::MAXG];
    for (int i = 0; i < n; ++i) {
        if (d[i].ks != 3 || d[i].stride != 1) return CAPF_ERR_UNSUPPORTED;
        g[i] = rh_args(d[i].x, nullptr, d[i].bias, d[i].residual, d[i].y, d[i].B, d[i].H, d[i].W, d[i].Cin, d[i].Cout, d[i].act);
        g[i].Wp3 = d[i].w_packed;
        if (!capf::gemm_bf16_ws_ok(g[i])) return CAPF_ERR_UNSUPPORTED;
    }
    return capf::launch_gemm_bf16_ws_group(g, n, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int64_t capf_op_conv_f32x3_pack_elems(int Cout, int Cin) { return Cin % 16 == 0 && Cout > 0 ? capf::f32x3_pack_elems(Cout, Cin) : 0; }

int capf_op_pack_conv_f32x3(void* stream, const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                            float eps, void* wp, float* bias, int Cout, int Cin) {
    if (!w || !wp || Cin % 16 != 0 || Cout % 4 != 0) return CAPF_ERR_UNSUPPORTED;
    return capf::launch_pack_conv_f32x3(w, gamma, beta, mean, var, eps, wp, bias, Cout, Cin, static_cast<hipStream_t>(stream)) ==
                   hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_conv_f32x3_group(void* stream, int n, const capf_conv_desc* d) {
    if (n <= 0 || n > capf::MAXG || !d) return CAPF_ERR_INVALID;
    capf::GemmArgs g[capf::MAXG];
    for (int i = 0; i < n; ++i) {
        if (d[i].ks != 3 || d[i].stride != 1) return CAPF_ERR_UNSUPPORTED;
        g[i] = capf::GemmArgs{};
        g[i].A = static_cast<const float*>(d[i].x);
        g[i].Wp3 = static_cast<const float*>(d[i].w_packed);
        g[i].bias = d[i].bias;
        g[i].res = static_cast<const float*>(d[i].residual);
        g[i].out = static_cast<float*>(d[i].y);
        g[i].M = d[i].B * d[i].H * d[i].W;
        g[i].N = d[i].Cout; g[i].K = 9 * d[i].Cin;
        g[i].conv = 1;
        g[i].Cin = d[i].Cin; g[i].H = d[i].H; g[i].W = d[i].W; g[i].Ho = d[i].H; g[i].Wo = d[i].W;
        g[i].ks = 3; g[i].stride = 1; g[i].pad = 1;
        g[i].omap = capf::row_ld(d[i].Cout);
        g[i].rmap = capf::row_ld(d[i].Cout);
        g[i].act = d[i].act;
        if (!capf::gemm_f32x3_ok(g[i])) return CAPF_ERR_UNSUPPORTED;
    }
    return capf::launch_gemm_f32x3_group(g, n, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int64_t capf_op_conv_f32h2_pack_elems(int Cout, int Cin) { return Cin % 16 == 0 && Cout > 0 ? capf::f32h2_pack_elems(Cout, Cin) : 0; }

int capf_op_pack_conv_f32h2(void* stream, const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                            float eps, void* wp, float* bias, int Cout, int Cin) {
    if (!w || !wp || Cin % 16 != 0 || Cout % 4 != 0) return CAPF_ERR_UNSUPPORTED;
    return capf::launch_pack_conv_f32h2(w, gamma, beta, mean, var, eps, wp, bias, Cout, Cin, static_cast<hipStream_t>(stream)) ==
                   hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_conv_f32h2_group(void* stream, int n, const capf_conv_desc* d) {
    if (n <= 0 || n > capf::MAXG || !d) return CAPF_ERR_INVALID;
    capf::GemmArgs g[capf::MAXG];
    for (int i = 0; i < n; ++i) {
        if (d[i].ks != 3 || d[i].stride != 1) return CAPF_ERR_UNSUPPORTED;
        g[i] = capf::GemmArgs{};
        g[i].A = static_cast<const float*>(d[i].x);
        g[i].Wp3 = static_cast<const float*>(d[i].w_packed);
        g[i].x3_h2 = 1;
        g[i].bias = d[i].bias;
        g[i].res = static_cast<const float*>(d[i].residual);
        g[i].out = static_cast<float*>(d[i].y);
        g[i].M = d[i].B * d[i].H * d[i].W;
        g[i].N = d[i].Cout; g[i].K = 9 * d[i].Cin;
        g[i].conv = 1;
        g[i].Cin = d[i].Cin; g[i].H = d[i].H; g[i].W = d[i].W; g[i].Ho = d[i].H; g[i].Wo = d[i].W;
        g[i].ks = 3; g[i].stride = 1; g[i].pad = 1;
        g[i].omap = capf::row_ld(d[i].Cout);
        g[i].rmap = capf::row_ld(d[i].Cout);
        g[i].act = d[i].act;
        if (!capf::gemm_f32x3_ok(g[i])) return CAPF_ERR_UNSUPPORTED;
    }
    return capf::launch_gemm_f32h2_group(g, n, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_conv_f32h2_tiles(int B, int H, int W, int* tile_pixels) { return capf::f32h2_tiles_m(B, H, W, tile_pixels); }

int capf_op_conv_f32h2_planes(void* stream, const capf_conv_desc* d, const int32_t* exps_in, int32_t* exps_out) {
    if (!d || d->ks != 3 || d->stride != 1 || (exps_in && exps_out)) return CAPF_ERR_UNSUPPORTED;
    capf::GemmArgs g{};
    g.A = static_cast<const float*>(d->x);
    g.Wp3 = static_cast<const float*>(d->w_packed);
    g.x3_h2 = 1;
    g.bias = d->bias;
    g.res = static_cast<const float*>(d->residual);
    g.out = static_cast<float*>(d->y);
    g.M = d->B * d->H * d->W;
    g.N = d->Cout; g.K = 9 * d->Cin;
    g.conv = 1;
    g.Cin = d->Cin; g.H = d->H; g.W = d->W; g.Ho = d->H; g.Wo = d->W;
    g.ks = 3; g.stride = 1; g.pad = 1;
    g.omap = capf::row_ld(d->Cout);
    g.rmap = capf::row_ld(d->Cout);
    g.act = d->act;
    g.h2_ein = exps_in;
    g.h2_eout = exps_out;
    if (!capf::gemm_f32x3_ok(g)) return CAPF_ERR_UNSUPPORTED;
    const hipError_t e = capf::launch_gemm_f32h2_group(&g, 1, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CAPF_OK : (e == hipErrorInvalidValue ? CAPF_ERR_UNSUPPORTED : CAPF_ERR_HIP);
}

int capf_op_conv_bf16_group(void* stream, int n, const capf_conv_desc* d, const void* const* w_rh, int32_t* variant) {
    if (n <= 0 || n > capf::MAXG || !d) return CAPF_ERR_INVALID;
    capf::GemmArgs g[capf::MAXG];
    for (int i = 0; i < n; ++i) {
        capf::GemmArgs a{};
        const int pad = d[i].ks / 2;
        a.A = d[i].x; a.Wp = d[i].w_packed; a.bias = d[i].bias; a.res = d[i].residual; a.out = d[i].y;
        a.Wp2 = w_rh ? static_cast<const float*>(w_rh[i]) : nullptr;
        a.Ho = (d[i].H + 2 * pad - d[i].ks) / d[i].stride + 1;
        a.Wo = (d[i].W + 2 * pad - d[i].ks) / d[i].stride + 1;
        a.M = d[i].B * a.Ho * a.Wo; a.N = d[i].Cout; a.K = d[i].ks * d[i].ks * d[i].Cin; a.Kpad = (a.K + 63) / 64 * 64;
        a.conv = 1; a.Cin = d[i].Cin; a.H = d[i].H; a.W = d[i].W; a.ks = d[i].ks; a.stride = d[i].stride; a.pad = pad;
        a.omap = capf::row_ld(d[i].Cout); a.rmap = capf::row_ld(d[i].Cout); a.amap = capf::row_ld(0);
        a.act = d[i].act;
        if (!capf::gemm_bf16_groupable(a)) return CAPF_ERR_UNSUPPORTED;
        g[i] = a;
    }
    int v = -1;
    const hipError_t e = capf::launch_gemm_bf16_group(g, n, static_cast<hipStream_t>(stream), &v);
    if (variant) *variant = v;
    return e == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_op_linear_bf16(void* stream, const void* x_bf16, const void* w_bf16, const float* bias, const float* residual, void* y,
                        int M, int N, int K, int gelu_bf16_out) {
    if (!x_bf16 || !w_bf16 || !y || K % 64 != 0) return CAPF_ERR_INVALID;
    return capf::launch_gemm_bf16_rows(x_bf16, w_bf16, bias, M, N, K, K, static_cast<float*>(y), capf::row_ld(N), residual,
                                       capf::row_ld(N), gelu_bf16_out, static_cast<hipStream_t>(stream)) == hipSuccess
               ? CAPF_OK : CAPF_ERR_UNSUPPORTED;
}

int capf_preprocess(void* stream, const uint8_t* images_bgr, int batch, int height, int width, const float mean[3],
                    const float* std3, int mode, float* images_out, const float* gt_in, float* gt_out, const float* k2d_in,
                    float* k2d_out, const float* kcrop_in, float* kcrop_out) {
    if (!images_bgr || !images_out || !k2d_in || !k2d_out || !kcrop_in || !kcrop_out || !mean || batch <= 0 || mode < 0 ||
        mode > 2 || (gt_in && !gt_out))
        return CAPF_ERR_INVALID;
    return capf::launch_preprocess(images_bgr, batch, height, width, mean, std3, mode, images_out, gt_in, gt_out, k2d_in,
                                   k2d_out, kcrop_in, kcrop_out, static_cast<hipStream_t>(stream)) == hipSuccess
               ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_fliptest_fuse(void* stream, const float* pred2, int batch, float* out) {
    if (!pred2 || !out || batch <= 0) return CAPF_ERR_INVALID;
    return capf::launch_fliptest_fuse(pred2, batch, out, static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_affine_from_center_scale(const double center[2], const double scale[2], int out_w, int out_h, double m[6]) {
    if (!center || !scale || !m || out_w <= 1 || out_h <= 1) return CAPF_ERR_INVALID;
    return capf::affine_from_center_scale(center, scale, out_w, out_h, m) ? CAPF_OK : CAPF_ERR_INVALID;
}

int capf_warp_affine(void* stream, const uint8_t* const* frames, const int32_t* dims, const double* m, int batch, int out_h,
                     int out_w, uint8_t* out) {
    if (!frames || !dims || !m || !out || batch <= 0 || out_h <= 0 || out_w <= 0) return CAPF_ERR_INVALID;
    return capf::launch_warp_affine_u8(frames, dims, m, out, batch, out_h, out_w, static_cast<hipStream_t>(stream)) == hipSuccess
               ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_pose_errors(void* stream, const float* pred, const float* gt, int n, int joints, const int32_t* prev, float* err) {
    if (!pred || !gt || !err || n <= 0 || joints <= 0) return CAPF_ERR_INVALID;
    hipError_t r = capf::launch_pose_errors(pred, gt, n, joints, prev, err, static_cast<hipStream_t>(stream));
    return r == hipSuccess ? CAPF_OK : (r == hipErrorInvalidValue ? CAPF_ERR_UNSUPPORTED : CAPF_ERR_HIP);
}

int capf_segment_sums(void* stream, const float* err, const int32_t* segment, const int32_t* prev, int n, int n_segments,
                      double* sums, int32_t* counts) {
    if (!err || !sums || !counts || n <= 0 || n_segments <= 0 || (n_segments > 1 && !segment)) return CAPF_ERR_INVALID;
    return capf::launch_segment_sums(err, segment, prev, n, n_segments, sums, counts, static_cast<hipStream_t>(stream)) == hipSuccess
               ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_keypoints_loss(void* stream, int mode, const float* pred, const float* gt, const float* validity, int rows, int dim,
                        float threshold, float* loss, float* dpred) {
    if (!pred || !gt || !validity || !loss || rows <= 0 || dim <= 0 || mode < 0 || mode > 2) return CAPF_ERR_INVALID;
    return capf::launch_keypoints_loss(pred, gt, validity, rows, dim, mode, threshold, loss, dpred,
                                       static_cast<hipStream_t>(stream)) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}

int capf_num_ops(const capf_handle* h) { return h ? (int)h->e.ops.size() : CAPF_ERR_INVALID; }

int capf_op_info(const capf_handle* h, int index, int batch, const char** name, const char** kernel, double* flops) {
    if (!h || index < 0 || index >= (int)h->e.ops.size() || batch <= 0) return CAPF_ERR_INVALID;
    const capf::Op& op = h->e.ops[index];
    static const char* kn[] = {"", "fuse_sum", "maxpool3x3s2", "bilinear_resize", "prep_embed", "sample_ref",
                               "layernorm", "deform_sample", "attention", "head", "", "", "embed", "ctx_attn", "res_chain", "mlp_chain"};
    if (name) *name = op.name.c_str();
    const int n_all = (int)h->e.ops.size();
    if (kernel && op.kind == capf::OP_GEMM && h->e.bneck1_member(index, batch) >= 0) {
        *kernel = capf::bneck1_bf16_kernel_name();                      // (the block's three convs ride in one launch)
        if (flops) *flops = op.flops_per_frame * batch;
        return CAPF_OK;
    }
    if (kernel && op.kind == capf::OP_GEMM && h->e.bneck0_member(index, batch) >= 0) {
        *kernel = capf::bneck0_bf16_kernel_name();                      // (the block's four convs ride in one launch)
        if (flops) *flops = op.flops_per_frame * batch;
        return CAPF_OK;
    }
    if (kernel && (h->e.pwchain_head(index, batch, n_all) || h->e.pwchain_head(index - 1, batch, n_all))) {
        *kernel = op.bf16 ? capf::gemm_bf16_pwchain_kernel_name() : capf::gemm_f32_pwchain_kernel_name();      // (both ops ride in one launch)
        if (name) *name = op.name.c_str();
        if (flops) *flops = op.flops_per_frame * batch;
        return CAPF_OK;
    }
    if (kernel) *kernel = op.kind != capf::OP_GEMM ? kn[op.kind] : (op.bf16 == 2 ? capf::gemm_bf16_rows_kernel_name((int)(op.rows_per_frame * batch), op.N)
                                                                      : op.bf16 ? capf::gemm_bf16_kernel_name(h->e.gemm_args(op, batch))
                                                                      : h->e.wino_now(op, batch) ? capf::gemm_wino_kernel_name(h->e.gemm_args(op, batch))
                                                                              : capf::gemm_f32_kernel_name(h->e.gemm_args(op, batch)));
    if (flops) *flops = op.flops_per_frame * batch;
    return CAPF_OK;
}

// FLOPs the MFMA pipe is asked to execute for one op at `batch` (2 x MACs issued, K padding included, tile-edge padding
// not): the Winograd kernels issue 18 (F(4,3), per four outputs) or 12 (F(2,3), per two) MACs per (cin, cout) where the
// direct conv issues 36 / 18, i.e. 1/2 or 2/3 of the algorithmic count capf_op_info reports; the split-fp32 tile (igemm_f32x3_ws.hip)
// issues six bf16 MACs per fp32 MAC -- on the bf16 pipe, whose peak is 16 x the fp32 pipe's.
int capf_op_executed_flops(const capf_handle* h, int index, int batch, double* flops) {
    if (!h || index < 0 || index >= (int)h->e.ops.size() || batch <= 0 || !flops) return CAPF_ERR_INVALID;
    const capf::Engine& e = h->e;
    const capf::Op& op = e.ops[index];
    *flops = op.flops_per_frame * batch;
    if (op.kind != capf::OP_GEMM) return CAPF_OK;
    const capf::Pack& pk = e.packs[op.pack];
    const double MN = 2.0 * (double)op.rows_per_frame * batch * op.N;
    if (op.conv && e.wino_now(op, batch) && pk.x3 && capf::gemm_f32x3_wanted(e.gemm_args(op, batch)))
        *flops = (e.x3_h2 ? 3.0 : 6.0) * MN * op.K;               // split-fp32 tiles: three fp16 / six bf16 piece products per fp32 product, on the 16-bit pipe
    else if (op.conv && e.wino_now(op, batch)) *flops = MN * op.Cin * (pk.Kpad == 18 * pk.Cin ? 4.5 : 6.0);
    else if (const capf::GemmArgs ga = e.gemm_args(op, batch); !op.bf16 && capf::gemm_f32_on_h2g(ga))
        *flops = 3.0 * MN * pk.KpadH;                              // two-fp16-piece GEMM: three piece products per fp32 product, on the 16-bit pipe
    else if (op.wino) *flops = MN * pk.Kpad2;                      // small batch: the direct kernel on the direct layout
    else if (pk.rh && op.conv) *flops = MN * op.K;                 // row-halo layout has no K padding (decided per launch; lower bound)
    else *flops = MN * (pk.direct ? op.K : pk.Kpad);
    return CAPF_OK;
}

// Algorithmic (compulsory) HBM bytes of one op at `batch`: every operand read once, the result written once.
int capf_op_bytes(const capf_handle* h, int index, int batch, double* bytes) {
    if (!h || index < 0 || index >= (int)h->e.ops.size() || batch <= 0 || !bytes) return CAPF_ERR_INVALID;
    const capf::Engine& e = h->e;
    const capf::Op& op = e.ops[index];
    const double B = batch, act = op.bf16 ? 2.0 : 4.0;
    double b = 0.0;
    switch (op.kind) {
        case capf::OP_GEMM: {
            const capf::Pack& pk = e.packs[op.pack];
            const double M = (double)op.rows_per_frame * B;
            const double in_elems = op.conv ? B * op.H * op.W * op.Cin : M * op.K;
            if (op.bf16 == 2) {          // lifter projection: bf16 operands, fp32 (or, after GELU, bf16) result
                b = in_elems * 2.0 + (double)pk.N * pk.K * 2.0 + (double)pk.N * 4.0 + M * op.N * (op.out_bf16 ? 2.0 : 4.0) +
                    (op.aux >= 0 ? M * op.N * 4.0 : 0.0);
                break;
            }
            b = in_elems * (op.conv && !op.bf16 ? 4.0 : act)                    // fp32 stem reads the fp32 image
                + (double)pk.N * pk.K * (pk.bf16 ? 2.0 : 4.0) + (double)pk.N * 4.0
                + M * op.N * (op.out_bf16 ? 2.0 : act)
                + ((op.aux >= 0 || op.res_param >= 0) ? M * op.N * act : 0.0)
                + ((op.conv && op.in[1] >= 0) ? B * op.i0 * op.i1 * op.N * act : 0.0);      // (the low-resolution map added behind the activation)
            break;
        }
        case capf::OP_FUSE: {
            const double out = B * op.H * op.W * op.C;
            b = out * act;
            for (int i = 0; i < op.n_in; ++i) b += out * act / (double)(1 << (2 * op.shift[i]));
            break;
        }
        case capf::OP_MAXPOOL:
        case capf::OP_RESIZE:
            b = B * op.C * act * ((double)op.H * op.W + (double)op.Ho * op.Wo * (op.aux >= 0 ? 2.0 : 1.0));   // (+ the added map)
            break;
        case capf::OP_LAYERNORM:
            b = (double)op.rows_per_frame * B * op.C * 4.0 * (op.aux >= 0 ? 3.0 : 2.0);
            break;
        case capf::OP_ATTENTION:
            b = (double)op.i0 * B * op.i1 * op.i2 * op.i3 * 4.0 * 4.0;          // q, k, v in; o out
            break;
        case capf::OP_SAMPLE_REF:
            b = B * op.i0 * op.C * (4.0 * act + 4.0);                           // 4 corners per joint + the sampled row
            break;
        case capf::OP_DEFORM:
            for (int l = 0; l < op.i1; ++l) b += B * op.i0 * op.i2 * op.lvlC[l] * (4.0 * op.i3 * act + 4.0);
            b += B * op.i0 * op.i1 * 3.0 * op.i2 * op.i3 * 4.0;
            break;
        case capf::OP_HEAD:
            b = (double)op.rows_per_frame * B * (op.C + 3.0) * 4.0;
            break;
        case capf::OP_RES_CHAIN:                                                 // the token rows in and out, the blocks' weights once
            b = (double)op.rows_per_frame * B * op.C * 4.0 * 2.0 + (double)op.i2 * 8.0 * op.C * op.C * 4.0;
            break;
        case capf::OP_MLP_CHAIN:
            b = (double)op.rows_per_frame * B * op.C * 4.0 * 2.0 + 4.0 * op.C * op.C * 4.0;
            break;
        case capf::OP_PREP_EMBED:
            b = B * op.i0 * (op.C + 4.0) * 4.0;
            break;
        case capf::OP_EMBED:
            b = B * op.i0 * (op.i2 * op.C + 4.0) * 4.0;                          // tokens written
            for (int l = 0; l < op.i1; ++l) b += B * op.i0 * op.lvlC[l] * 4.0 * act + (double)op.C * op.lvlC[l] * 4.0;
            break;
        case capf::OP_CTX_ATTN:
            for (int l = 0; l < op.i1; ++l)
                b += B * op.i0 * op.i2 * op.i3 * op.lvlC[l] * 4.0 * act + (double)(op.C / op.i2) * op.lvlC[l] * 4.0;
            b += B * op.i0 * (op.i1 + 1 + op.i1) * op.C * 4.0;                    // tokens read, tokens 1..L written
            break;
        default: break;
    }
    *bytes = b;
    return CAPF_OK;
}

int capf_op_schedule(const capf_handle* h, int index, int32_t* region, int32_t* level, int32_t* lane, int32_t* reads,
                     int32_t* writes) {
    if (!h || index < 0 || index >= (int)h->e.ops.size()) return CAPF_ERR_INVALID;
    const capf::Engine& e = h->e;
    const capf::Op& op = e.ops[index];
    const bool control = op.kind == capf::OP_FORK || op.kind == capf::OP_JOIN;
    if (region) *region = control ? -1 : op.region;
    if (lane) *lane = op.lane;
    if (level) {
        *level = -1;
        if (!control && op.region >= 0) {
            const auto& lv = e.region_levels[op.region];
            for (size_t l = 0; l < lv.size(); ++l)
                for (int oi : lv[l])
                    if (oi == index) *level = (int32_t)l;
        }
    }
    if (reads) {
        for (int i = 0; i < 4; ++i) reads[i] = op.in[i];
        reads[4] = op.aux;
    }
    if (writes) {
        writes[0] = op.out;
        writes[1] = op.aux2;
        for (int i = 0; i < 4; ++i) writes[2 + i] = op.outs[i];
    }
    return CAPF_OK;
}

int capf_forward_prefix(capf_handle* h, void* stream, const float* images_nhwc, const float* k2d, float* kcrop_inout, int batch,
                        float* out, int n_ops) {
    if (!h || !images_nhwc) return CAPF_ERR_INVALID;
    Engine& e = h->e;
    if (n_ops < 0 || n_ops > (int)e.ops.size()) return CAPF_ERR_INVALID;
    if (n_ops > e.n_backbone_ops && (!k2d || !kcrop_inout || !out)) return CAPF_ERR_INVALID;
    int rc = check_run(e, batch);
    if (rc) return rc;
    e.images = images_nhwc; e.k2d = k2d; e.kcrop = kcrop_inout; e.out = out;
    e.last_batch = batch;
    e.invalidate_train();
    return e.run(static_cast<hipStream_t>(stream), batch, 0, n_ops);
}

int capf_op_describe(const capf_handle* h, int index, capf_op_desc* d) {
    if (!h || !d || index < 0 || index >= (int)h->e.ops.size()) return CAPF_ERR_INVALID;
    const Engine& e = h->e;
    const capf::Op& op = e.ops[index];
    memset(d, 0, sizeof(*d));
    d->kind = op.kind == capf::OP_GEMM ? 0 : op.kind == capf::OP_FUSE ? 1 : op.kind == capf::OP_MAXPOOL ? 2
              : op.kind == capf::OP_RESIZE ? 3 : op.kind == capf::OP_LAYERNORM ? 4 : op.kind == capf::OP_ATTENTION ? 5 : -1;
    if (op.kind == capf::OP_FUSE && op.i0 == 1) d->kind = -1;          // debug copy
    d->backbone = index < e.n_backbone_ops;
    d->p_weight = d->p_bn_weight = d->p_bias = d->p_ln_weight = d->p_ln_bias = -1;
    d->rows_per_frame = (int)op.rows_per_frame;
    d->eps = op.eps;
    {
        const capf::RowMap* m[3] = {&op.amap, &op.omap, &op.rmap};
        for (int k = 0; k < 3; ++k) { d->maps[k][0] = m[k]->G; d->maps[k][1] = m[k]->S1; d->maps[k][2] = m[k]->S2; d->maps[k][3] = m[k]->off; }
    }
    const int act_dt = e.bf16() ? 2 : 0;
    d->in_dtype = d->out_dtype = (d->backbone ? act_dt : 0);
    d->H = op.H; d->W = op.W; d->Ho = op.Ho; d->Wo = op.Wo;
    if (op.kind == capf::OP_GEMM) {
        const capf::Pack& pk = e.packs[op.pack];
        d->conv = op.conv; d->Cin = op.conv ? op.Cin : op.K; d->Cout = op.N;
        d->ks = op.ks; d->stride = op.stride; d->pad = op.pad; d->act = op.act;
        d->has_residual = op.aux >= 0 || op.res_param >= 0;
        if (op.conv && op.in[1] >= 0) { d->up_H = op.i0; d->up_W = op.i1; }
        d->mfma_bf16 = (op.bf16 || op.out_bf16) ? 1 : 0;
        if (op.conv) {
            d->in_dtype = op.in[0] == -2 ? 0 : (op.bf16 ? 2 : 0);
            d->out_dtype = (op.bf16 || op.out_bf16) ? 2 : 0;
            d->p_weight = pk.w[0]; d->p_bn_weight = pk.bn_g;
        } else {
            d->in_dtype = op.bf16 == 2 ? 2 : 0;
            d->out_dtype = op.out_bf16 ? 2 : 0;
            d->p_weight = pk.n_lin == 1 ? pk.w[0] : -1;
            d->p_bias = pk.n_lin == 1 ? pk.b[0] : -1;
            d->p_ln_weight = op.ln_w; d->p_ln_bias = op.ln_b;
        }
    } else if (op.kind == capf::OP_LAYERNORM) {
        d->Cin = d->Cout = op.C;
        d->in_dtype = 0; d->out_dtype = op.out_bf16 ? 2 : 0;
        d->has_residual = op.aux >= 0;
        d->p_ln_weight = op.p0; d->p_ln_bias = op.p1;
        d->maps[1][0] = 1; d->maps[1][1] = op.C; d->maps[1][2] = 0; d->maps[1][3] = 0;        // normalised rows are written densely
    } else if (op.kind == capf::OP_ATTENTION) {
        d->attn[0] = op.i0; d->attn[1] = op.i1; d->attn[2] = op.i2; d->attn[3] = op.i3;
        d->rows_per_frame = op.i0 * op.i1;
        d->Cin = 3 * op.i2 * op.i3; d->Cout = op.i2 * op.i3;
        d->in_dtype = 0; d->out_dtype = op.out_bf16 ? 2 : 0;
        d->maps[0][0] = 1; d->maps[0][1] = d->Cin; d->maps[1][0] = 1; d->maps[1][1] = d->Cout;
    } else {
        d->Cin = d->Cout = op.C;
        d->n_in = op.n_in; d->relu = op.relu;
        d->has_residual = op.kind == capf::OP_RESIZE && op.aux >= 0;      // out = resize(in) + aux
        for (int i = 0; i < 4; ++i) d->shift[i] = op.shift[i];
        if (op.kind == capf::OP_FUSE) { d->Ho = op.H; d->Wo = op.W; }
    }
    d->checkpoint = (op.bneck_c3 >= 0 ? op.bneck_c3 : op.region >= 0 ? e.regions[op.region].second : index) + 1;
    return CAPF_OK;
}

int capf_op_describe_sized(const capf_handle* h, int index, void* desc, size_t desc_bytes) {
    if (!desc || desc_bytes == 0) return CAPF_ERR_INVALID;
    capf_op_desc full;
    const int rc = capf_op_describe(h, index, &full);
    if (rc != CAPF_OK) return rc;
    memset(desc, 0, desc_bytes);
    memcpy(desc, &full, desc_bytes < sizeof(full) ? desc_bytes : sizeof(full));
    return CAPF_OK;
}

int capf_op_tensor(const capf_handle* h, int index, int slot, const void** dev_ptr) {
    if (!h || !dev_ptr || index < 0 || index >= (int)h->e.ops.size() || slot < 0 || slot > 5) return CAPF_ERR_INVALID;
    const Engine& e = h->e;
    const capf::Op& op = e.ops[index];
    const int buf = slot < 4 ? op.in[slot] : slot == 4 ? op.aux : op.out;
    if (buf == -2) { *dev_ptr = e.images; return CAPF_OK; }
    if (buf < 0 || !e.ws || e.last_batch <= 0) return CAPF_ERR_INVALID;
    *dev_ptr = e.bptr(buf, e.last_batch);
    return CAPF_OK;
}

int capf_op_h2_planes(const capf_handle* h, int index, int batch, const void** exps, int32_t* tile_pixels) {
    if (!h || index < 0 || index >= (int)h->e.ops.size() || batch <= 0) return CAPF_ERR_INVALID;
    const Engine& e = h->e;
    const capf::Op& op = e.ops[index];
    if (!op.h2_role || !e.ws) return 0;
    const capf::GemmArgs a = e.gemm_args(op, batch);
    if (!a.h2_ein && !a.h2_eout) return 0;
    if (exps) *exps = a.h2_ein ? static_cast<const void*>(a.h2_ein) : static_cast<const void*>(a.h2_eout);
    int tp = 0;
    capf::f32h2_tiles_m(batch, op.H, op.W, &tp);
    if (tile_pixels) *tile_pixels = tp;
    return op.h2_role;
}

int capf_forward_profile(capf_handle* h, void* stream, const float* images_nhwc, const float* k2d, float* kcrop_inout,
                         int batch, float* out, float* op_ms, int n_ops) {
    if (!h || !images_nhwc || !k2d || !kcrop_inout || !out || !op_ms) return CAPF_ERR_INVALID;
    Engine& e = h->e;
    const int n = (int)e.ops.size();
    if (n_ops < n) return CAPF_ERR_INVALID;
    int rc = check_run(e, batch);
    if (rc) return rc;
    e.images = images_nhwc; e.k2d = k2d; e.kcrop = kcrop_inout; e.out = out;
    e.last_batch = batch;
    e.invalidate_train();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& x : ev)
        if (hipEventCreate(&x) != hipSuccess) return CAPF_ERR_HIP;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = e.run(s, batch, 0, n, ev.data());
    if (rc == CAPF_OK && hipStreamSynchronize(s) != hipSuccess) rc = CAPF_ERR_HIP;
    for (int i = 0; i < n && rc == CAPF_OK; ++i)
        if (hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]) != hipSuccess) rc = CAPF_ERR_HIP;
    for (auto& x : ev) (void)hipEventDestroy(x);
    return rc;
}

int capf_forward_profile_launches(capf_handle* h, void* stream, const float* images_nhwc, const float* k2d,
                                  float* kcrop_inout, int batch, float* out, float* op_ms, int32_t* op_leader, int n_ops) {
    if (!h || !images_nhwc || !k2d || !kcrop_inout || !out || !op_ms || !op_leader) return CAPF_ERR_INVALID;
    Engine& e = h->e;
    const int n = (int)e.ops.size();
    if (n_ops < n) return CAPF_ERR_INVALID;
    int rc = check_run(e, batch);
    if (rc) return rc;
    e.images = images_nhwc; e.k2d = k2d; e.kcrop = kcrop_inout; e.out = out;
    e.last_batch = batch;
    e.invalidate_train();
    capf::LaunchLog log;
    log.op_leader.assign(n, -1);
    log.op_variant.assign(n, -1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = e.run(s, batch, 0, n, nullptr, &log);
    e.last_variants = log.op_variant;
    if (rc == CAPF_OK && hipStreamSynchronize(s) != hipSuccess) rc = CAPF_ERR_HIP;
    for (int i = 0; i < n; ++i) { op_ms[i] = 0.f; op_leader[i] = log.op_leader[i]; }
    for (size_t k = 0; k < log.leader.size() && rc == CAPF_OK; ++k)
        if (hipEventElapsedTime(&op_ms[log.leader[k]], log.ev[k], log.ev[k + 1]) != hipSuccess) rc = CAPF_ERR_HIP;
    return rc;
}

int capf_forward_profile_variants(const capf_handle* h, int32_t* op_variant, int n_ops) {
    if (!h || !op_variant || n_ops < (int)h->e.last_variants.size()) return CAPF_ERR_INVALID;
    for (size_t i = 0; i < h->e.last_variants.size(); ++i) op_variant[i] = h->e.last_variants[i];
    return CAPF_OK;
}

int capf_forward_stats(const capf_handle* h, int batch, int64_t* launches, double* flops) {
    if (!h) return CAPF_ERR_INVALID;
    int64_t n = 0;
    double f = 0.0;
    for (const capf::Op& op : h->e.ops) {
        if (op.kind == capf::OP_FUSE && op.i0 == 1) continue;
        if (op.kind == capf::OP_FORK || op.kind == capf::OP_JOIN) continue;
        ++n;
        f += op.flops_per_frame * batch;
    }
    if (launches) *launches = n;
    if (flops) *flops = f;
    return CAPF_OK;
}

}  // extern "C"
