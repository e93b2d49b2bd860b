// Training step of the lifter: forward that keeps what the backward needs, and the
// backward itself.  The backbone is frozen (conpose.py:22-25) and runs through the inference plan; its four
// context maps are constants here, so gradients flow only into the 191 `volume_net.*` parameters
// (pose_dformer.py:144-208) — exactly the tensors DDP all-reduces in the reference (train.py:361-362).
//
// Every product is an igemm_f32 launch: with dY [M,N], X [M,K], W [N,K] (all row-major),
//   dX = dY . W        -> A = dY (K' = N),            Wp = W^T  [K][Npad]   (transpose_pad of the weight)
//   dW = dY^T . X      -> A = dY^T [N][Mpad] (K' = M), Wp = X^T  [K][Mpad]   (split-K over M, slabs summed in order)
//   db = column sums of dY (two-stage deterministic reduction)
// Gradients are written ONCE each (no accumulation, no atomics) into one flat fp32 buffer laid out in
// state_dict order, so a single RCCL all-reduce covers what DDP sends in three buckets.
#include <string.h>

#include <algorithm>

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

static size_t r64(size_t n) { return (n + 63) / 64 * 64; }
static int r32(int n) { return (n + 31) / 32 * 32; }

// ---------------------------------------------------------------------------------------------------
// layout of the training region (after the inference workspace): offset = cursor; cursor += pf*B + fixed
// ---------------------------------------------------------------------------------------------------
void Engine::train_layout(int B, TrainLayout& L) const {
    const int J = cfg.num_joints, Lv = cfg.levels, L1 = Lv + 1, C = cfg.embed_dim_ratio, D = C * L1;
    const int NH = cfg.deform_heads;
    size_t cur = 0;
    auto take = [&](size_t pf, size_t fixed = 0) {
        const size_t o = cur;
        cur += r64(pf * (size_t)B + fixed);
        return o;
    };
    L.X = take((size_t)J * D);
    for (int l = 0; l < Lv; ++l) L.S[l] = take((size_t)J * feat_C[l]);
    for (int i = 0; i < Lv; ++i) {
        TrainLayout::Ctx& c = L.ctx[i];
        const size_t R = (size_t)J * Lv;
        c.xh1 = take(R * C); c.rs1 = take(R); c.y1 = take(R * C); c.ao = take(R * 64);
        for (int l = 0; l < Lv; ++l) c.U[l] = take((size_t)J * NH * feat_C[l]);
        c.xh2 = take(R * C); c.rs2 = take(R); c.y2 = take(R * C); c.hp = take(R * 2 * C); c.hg = take(R * 2 * C);
    }
    for (int g = 0; g < 2; ++g)
        for (int i = 0; i < Lv; ++i) {
            TrainLayout::Att& a = g == 0 ? L.res[i] : L.joint[i];
            const size_t E = (size_t)J * D;       // rows * dim is J*D elements per frame for both groups
            const size_t R = g == 0 ? (size_t)J * L1 : (size_t)J;
            a.xh1 = take(E); a.rs1 = take(R); a.y1 = take(E); a.qkv = take(3 * E); a.o = take(E);
            a.xh2 = take(E); a.rs2 = take(R); a.y2 = take(E); a.hp = take(2 * E); a.hg = take(2 * E);
        }
    L.xhh = take((size_t)J * D); L.rsh = take(J); L.yh = take((size_t)J * D);
    // backward scratch
    L.dX = take((size_t)J * D);
    L.gA = take((size_t)J * 3 * D);
    L.gB = take((size_t)J * 3 * D);
    L.gC = take((size_t)J * 3 * D);
    L.cat = take(0, (size_t)64 * (C + 1));
    for (int l = 0; l < Lv; ++l) L.dU[l] = take((size_t)J * NH * feat_C[l]);
    const size_t maxNR = (size_t)J * 3 * D;                    // max over linears of (N or K) * rows per frame
    L.tA = take(maxNR, (size_t)3 * D * 32);
    L.tB = take(maxNR, (size_t)3 * D * 32);
    L.wT = take(0, (size_t)3 * D * D + 64 * 2 * D);            // largest transposed weight [K][Npad]
    L.slab_cap = (size_t)16 * (3 * D * D + 3 * D);             // split-K slabs of the largest weight gradient (+ its bias gradient) ...
    L.slabs_elems = 8 * L.slab_cap;                            // ... and room for several layers' slabs: their sums are deferred (t_slab_flush)
    L.slabs = take(0, L.slabs_elems);
    L.red_cap = (size_t)64 * 3 * D;                            // one column reduction's partial sums ...
    L.red_elems = 40 * L.red_cap;                              // ... times the reductions whose second stages wait for one launch (t_col_flush)
    L.red = take(0, L.red_elems);
    L.h2w = take(0, t_h2_elems);
    L.h2max = take(0, t_h2_max_elems);
    L.total = cur;
}

size_t Engine::train_elems(int B) const {
    TrainLayout L;
    train_layout(B, L);
    return L.total;
}

// ---------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------
// ---- the step's weights as two-fp16-piece packs -------------------------------------------------------------------------------------
// Which matrices: every nn.Linear the step multiplies by with at least 64 output columns in that product (the 16-wide per-head
// projections of the deformable blocks are HBM-bound on the fp32 kernel's 256 x 32 tile and stay there).
void Engine::t_h2_plan() {
    t_h2_specs.clear();
    t_h2_tab.clear();
    t_h2_elems = t_h2_max_elems = 0;
    t_h2_tiles = 0;
    if (!use_h2g || bf16() || !cfg.training) return;
    const std::string V = "volume_net";
    const int Lv = cfg.levels, L1 = Lv + 1, C = cfg.embed_dim_ratio, D = C * L1;
    const int NH = cfg.deform_heads, NS = cfg.deform_samples;
    auto add = [&](const std::string& name, int N, int K, bool fwd, bool bwd) {
        const auto it = param_index.find(name + ".weight");
        if (it == param_index.end()) return;
        if (N < 64) fwd = false;                             // (output columns of y = x W^T ...
        if (K < 64) bwd = false;                             //  ... and of dX = dY W)
        if (fwd || bwd) t_h2_specs.push_back(H2TrainSpec{it->second, -1, N, K, K, fwd, bwd});
    };
    for (int l = 0; l < Lv; ++l) add(V + ".feat_embed." + std::to_string(l), C, feat_C[l], true, false);
    for (int i = 0; i < Lv && cfg.context_blocks; ++i) {
        const std::string p = V + ".context_blocks." + std::to_string(i);
        if ((size_t)i < ctx_ao_pack.size() && 3 * NH * NS >= 64)
            t_h2_specs.push_back(H2TrainSpec{-1, ctx_ao_pack[i], 3 * NH * NS, C, packs[ctx_ao_pack[i]].Kpad, true, C >= 64});
        for (int l = 0; l < Lv; ++l) add(p + ".embed_proj." + std::to_string(l), C / NH, feat_C[l], true, true);
        add(p + ".mlp.fc1", 2 * C, C, true, true);
        add(p + ".mlp.fc2", C, 2 * C, true, true);
    }
    for (int g = 0; g < 2; ++g)
        for (int i = 0; i < Lv; ++i) {
            const std::string p = V + (g == 0 ? ".res_blocks." : ".joint_blocks.") + std::to_string(i);
            const int dim = g == 0 ? C : D;
            add(p + ".attn.qkv", 3 * dim, dim, true, true);
            add(p + ".attn.proj", dim, dim, true, true);
            add(p + ".mlp.fc1", 2 * dim, dim, true, true);
            add(p + ".mlp.fc2", dim, 2 * dim, true, true);
        }
    size_t off = 0, moff = 0;
    int tiles = 0;
    for (const H2TrainSpec& sp : t_h2_specs) {
        H2TrainW e{};
        e.N = sp.N; e.K = sp.K; e.ld = sp.ld;
        e.fwd_off = e.bwd_off = -1;
        if (sp.fwd) { e.fwd_off = (long)off; off += r64((size_t)f32h2_gemm_pack_elems(sp.N, r32(sp.K))); }
        if (sp.bwd) { e.bwd_off = (long)off; off += r64((size_t)f32h2_gemm_pack_elems(sp.K, r32(sp.N))); }
        e.tile_start = tiles;
        tiles += ((sp.N + 31) / 32) * ((sp.K + 31) / 32);
        e.max_off = (int)moff;
        moff += (size_t)sp.N + sp.K;
        t_h2_tab.push_back(e);
    }
    t_h2_elems = off;
    t_h2_max_elems = r64(moff);
    t_h2_tiles = tiles;
}

// pack every matrix of the table from the CURRENT parameters (three launches); the packs live until the next forward_train
int Engine::t_h2_prepare(hipStream_t s, const TrainLayout& L, float* tw, int B) {
    t_h2_base = nullptr;
    t_h2_index.clear();
    if (t_h2_specs.empty() || B < H2G_MIN_BATCH) return CAPF_OK;
    bool same = t_h2_on_device;
    std::vector<const float*> now(t_h2_specs.size());
    for (size_t i = 0; i < t_h2_specs.size(); ++i) {
        const H2TrainSpec& sp = t_h2_specs[i];
        now[i] = sp.param >= 0 ? params[sp.param].ptr : pack_arena + packs[sp.pack].w_off;
        if (t_h2_tab[i].w != now[i]) same = false;
        t_h2_index[now[i]] = (int)i;
    }
    H2TrainW* tab_dev = reinterpret_cast<H2TrainW*>(pack_arena + t_h2_tab_off);
    if (!same) {
        // the table is pageable host memory that an earlier step's upload may still be reading, and the device copy may still be in use by
        // that step's pack kernels: drain the stream before either changes (parameters are re-bound once in a blue moon, not per step)
        HIP_TRY(hipStreamSynchronize(s));
        for (size_t i = 0; i < t_h2_specs.size(); ++i) t_h2_tab[i].w = now[i];
        HIP_TRY(hipMemcpyAsync(tab_dev, t_h2_tab.data(), t_h2_tab.size() * sizeof(H2TrainW), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        t_h2_on_device = true;
    }
    HIP_TRY(launch_pack_f32h2_train(tab_dev, (int)t_h2_tab.size(), t_h2_tiles, tw + L.h2w, reinterpret_cast<int*>(tw + L.h2max),
                                    (long)t_h2_max_elems, s));
    t_h2_base = tw + L.h2w;
    return CAPF_OK;
}

const float* Engine::t_h2_pack(const float* W, bool transposed) const {
    if (!t_h2_base) return nullptr;
    const auto it = t_h2_index.find(W);
    if (it == t_h2_index.end()) return nullptr;
    const long off = transposed ? t_h2_tab[it->second].bwd_off : t_h2_tab[it->second].fwd_off;
    return off < 0 ? nullptr : t_h2_base + off;
}

int Engine::t_gemm(hipStream_t s, const float* A, RowMap amap, int M, int N, int K, const float* W, int Kpad,
                   const float* bias, float* out, RowMap omap, const float* res, RowMap rmap, int act,
                   const float* rscale, int rs_div, const float* Wh2) {
    GemmArgs a{};
    a.A = A; a.Wp = W; a.bias = bias; a.res = res; a.out = out;
    if (!Wh2) Wh2 = t_h2_pack(W, false);                    // (this step's two-piece copy of W, where the table holds one)
    if (Wh2 && r32(K) == Kpad) a.Wh2 = Wh2;                 // (the pack's chunks are the fp32 matrix's: same Kpad)
    a.M = M; a.N = N; a.K = K; a.Kpad = Kpad;
    a.amap = amap; a.omap = omap; a.rmap = rmap; a.act = act;
    a.rscale = rscale; a.rs_div = rs_div;
    HIP_TRY(launch_gemm_f32(a, s));
    return CAPF_OK;
}

// ---- deferred slab sums ---------------------------------------------------------------------------------------------------------------
// a weight gradient's slabs: `elems` floats of the slab area (64-float granules); a full area sums what waits first
float* Engine::t_slab_take(hipStream_t s, float* area, size_t cap, size_t elems, int* rc) {
    const size_t need = (elems + 63) & ~(size_t)63;
    *rc = CAPF_OK;
    if (need > cap) { err = "capf_backward: a weight gradient's slabs exceed the slab area"; *rc = CAPF_ERR_STATE; return nullptr; }
    if (t_slab_cur + need > cap) {
        if ((*rc = t_slab_flush(s))) return nullptr;
    }
    float* p = area + t_slab_cur;
    t_slab_cur += need;
    return p;
}

int Engine::t_slab_defer(hipStream_t s, const float* slabs, int nslab, long n, float* dst) {
    if (!batch_reduce) {                                     // CAPF_PLAN_NO_BATCHED_REDUCE: the per-layer kernel, right away; the area is free again
        HIP_TRY(launch_slab_sum(slabs, nslab, n, dst, s));
        if (t_slab_jobs.count == 0) t_slab_cur = 0;
        return CAPF_OK;
    }
    if ((n & 3) || n >= (1L << 31) - 4096) {                 // (no 16-byte form: summed at once, the area stays taken until the next flush)
        HIP_TRY(launch_slab_sum(slabs, nslab, n, dst, s));
        return CAPF_OK;
    }
    SlabBatch& b = t_slab_jobs;
    const int j = b.count;
    b.src[j] = slabs; b.dst[j] = dst; b.n[j] = (int)n; b.nslab[j] = nslab;
    b.blk_end[j] = (j ? b.blk_end[j - 1] : 0) + (int)((n + 1023) / 1024);
    b.count = j + 1;
    if (b.count == SLAB_BATCH_MAX) return t_slab_flush(s);
    return CAPF_OK;
}

// a column reduction of the backward: first stage now (into its own piece of the red area), second stage with the others' in t_col_flush.
// cap_elems: what the first stage may use (0: the kernel's 64-chunk default) -- it decides the chunking, i.e. the summation order
int Engine::t_colreduce(hipStream_t s, const TrainLayout& L, float* tw, const float* A, RowMap amap, const float* Bm, RowMap bmap, int bmode,
                        int rows, int C, float* dst, long dst_stride, float* dst2, size_t cap_elems) {
    const int nout = dst2 ? 2 : 1;
    const size_t need = (((size_t)nout * colreduce_chunks(rows, C, nout, cap_elems) * C) + 63) & ~(size_t)63;
    if (need > L.red_elems) { err = "capf_backward: a column reduction's partial sums exceed the red area"; return CAPF_ERR_STATE; }
    if (t_col_cur + need > L.red_elems || t_col_jobs.count == COL_BATCH_MAX) {
        if (int rc = t_col_flush(s)) return rc;
    }
    float* scratch = tw + L.red + t_col_cur;
    t_col_cur += need;
    HIP_TRY(launch_colreduce(A, amap, Bm, bmap, bmode, rows, C, dst, dst_stride, 0, scratch, s, dst2, cap_elems, batch_reduce ? &t_col_jobs : nullptr));
    if (!batch_reduce) t_col_cur = 0;                        // (its second stage ran: the scratch is free again)
    return CAPF_OK;
}

int Engine::t_col_flush(hipStream_t s) {
    if (t_col_jobs.count > 0) HIP_TRY(launch_colreduce_final_batch(t_col_jobs, s));
    t_col_jobs.count = 0;
    t_col_cur = 0;
    return CAPF_OK;
}

int Engine::t_slab_flush(hipStream_t s) {
    if (t_slab_jobs.count > 0) HIP_TRY(launch_slab_sum_batch(t_slab_jobs, s));
    t_slab_jobs.count = 0;
    t_slab_cur = 0;                                          // (stream order: the next layer's slabs are written behind the sums that read these)
    return CAPF_OK;
}

// gradients of y = x W^T + b :  gW [N][K], gb [N], and (optionally) dX (+)= dY W
int Engine::t_linear_bwd(hipStream_t s, const TrainLayout& L, float* tw, const float* dY, RowMap dymap, int rows, int N,
                         int K, const float* Xin, RowMap xmap, const float* W, float* dX, RowMap dxmap, bool acc_dx,
                         float* gW, float* gb) {
    // dW (and db, when it sits right behind dW in the flat gradient -- every nn.Linear's weight / bias pair does) straight from the
    // row-major dY and X: no transposes, no column-reduction launches (wgrad_tn_kernel; multiples of four and plain row pitches only)
    // (its operand tiles are 16-byte LDS-DMA loads: row pitches and offsets must be multiples of four elements)
    if (gW && N % 4 == 0 && K % 4 == 0 && dymap.G == 1 && xmap.G == 1 && ((dymap.S1 | dymap.off | xmap.S1 | xmap.off) & 3) == 0 &&
        (double)rows * (double)std::max(dymap.S1, xmap.S1) * 4.0 < 2.0e9) {
        const bool bias_here = gb && gb == gW + (long)N * K;
        if (gb && !bias_here) {
            if (int rc = t_colreduce(s, L, tw, dY, dymap, nullptr, row_ld(0), 0, rows, N, gb, 1, nullptr, L.red_cap)) return rc;
        }
        const bool h2 = t_h2_base && N % 128 == 0 && K % 128 == 0;      // (this step runs its products on the 16-bit matrix pipe)
        const int tiles = h2 ? (N / 128) * (K / 128) : ((N + 63) / 64) * ((K + 63) / 64), chunks = (rows + 31) / 32;
        const long slab = (long)N * K + (bias_here ? N : 0);
        // row slices: ~2048 blocks per launch (four rounds of two per CU, so that a tile count that is no multiple of 256 costs a few per
        // cent, not a half-empty round), at least 16 chunks each, as many as the slab buffer holds -- the small layers (128 x 128:
        // four tiles, 43520 rows) ran 64 blocks of 85 chunks
        const long slab_cap = (long)L.slab_cap;                     // (the layout's own number: train_layout)
        int splits = std::max(1, ((h2 ? 512 : 2048) + tiles - 1) / tiles);       // (the two-piece kernel: one round of two blocks per CU -- its slabs are 128 x 128)
        splits = std::min(splits, std::max(1, chunks / 16));
        splits = (int)std::min<long>(splits, std::max<long>(1, slab_cap / slab));
        const int cps = (chunks + splits - 1) / splits, slices = (chunks + cps - 1) / cps;
        float* slabs = nullptr;
        if (slices > 1) {
            int rc = CAPF_OK;
            slabs = t_slab_take(s, tw + L.slabs, L.slabs_elems, (size_t)slices * slab, &rc);
            if (rc) return rc;
        }
        HIP_TRY(launch_wgrad_tn(dY + dymap.off, dymap.S1, Xin + xmap.off, xmap.S1, rows, N, K, slices > 1 ? slabs : gW, slab, splits,
                                bias_here ? 1 : 0, s, h2));
        if (slices > 1) {
            if (int rc = t_slab_defer(s, slabs, slices, slab, gW)) return rc;      // (summed with the other layers' slabs: t_slab_flush)
        }
        gW = nullptr;
        gb = nullptr;
    }
    if (gb) {
        if (int rc = t_colreduce(s, L, tw, dY, dymap, nullptr, row_ld(0), 0, rows, N, gb, 1, nullptr, L.red_cap)) return rc;
    }
    if (gW) {
        const int Mp = r32(rows);
        float* tA = tw + L.tA;
        float* tB = tw + L.tB;
        HIP_TRY(launch_transpose_pad(dY, dymap, rows, N, tA, Mp, s));
        HIP_TRY(launch_transpose_pad(Xin, xmap, rows, K, tB, Mp, s));
        GemmArgs a{};
        a.A = tA; a.Wp = tB; a.out = gW;
        a.M = N; a.N = K; a.K = Mp; a.Kpad = Mp;
        a.amap = row_ld(Mp); a.omap = row_ld(K); a.rmap = row_ld(K);
        const int tiles = ((N + 63) / 64) * ((K + 63) / 64), chunks = Mp / 32;
        int splits = std::min(16, std::max(1, 512 / std::max(1, tiles)));
        splits = std::min(splits, chunks);
        if (splits > 1) {
            int rc = CAPF_OK;
            a.out = t_slab_take(s, tw + L.slabs, L.slabs_elems, (size_t)splits * N * K, &rc);
            if (rc) return rc;
            a.splits = splits;
            a.cps = (chunks + splits - 1) / splits;
            a.splits = (chunks + a.cps - 1) / a.cps;
            a.split_stride = (long)N * K;
        }
        HIP_TRY(launch_gemm_f32(a, s));
        if (a.splits > 1) {
            if (int rc = t_slab_defer(s, a.out, a.splits, (long)N * K, gW)) return rc;
        }
    }
    if (dX) {
        const int Np = r32(N);
        if (const float* h2t = t_h2_pack(W, true)) {         // W^T is in the step's table as a two-piece pack: no transpose, 16-bit matrix pipe
            GemmArgs a{};
            a.A = dY; a.Wp = h2t; a.Wh2 = h2t; a.out = dX; a.res = acc_dx ? dX : nullptr;
            a.M = rows; a.N = K; a.K = N; a.Kpad = Np;
            a.amap = dymap; a.omap = dxmap; a.rmap = dxmap; a.act = ACT_NONE;
            if (gemm_f32h2g_ok(a)) {
                HIP_TRY(launch_gemm_f32h2g(a, s));
                return CAPF_OK;
            }
        }
        float* wT = tw + L.wT;
        HIP_TRY(launch_transpose_pad(W, row_ld(K), N, K, wT, Np, s));
        int rc = t_gemm(s, dY, dymap, rows, K, N, wT, Np, nullptr, dX, dxmap, acc_dx ? dX : nullptr, dxmap, ACT_NONE,
                        nullptr, 1);
        if (rc) return rc;
    }
    return CAPF_OK;
}

static const float* P(const Engine& e, const std::string& n) { return e.params[e.param_index.at(n)].ptr; }

// ---------------------------------------------------------------------------------------------------
// forward (training): same math as the inference plan, every intermediate kept
// masks: DropPath multipliers (0 or 1/keep_prob), or nullptr for "no drop":
//   ctx[i]: m1[B], m2[B]  |  res[i]: m1[B*J], m2[B*J]  |  joint[i]: m1[B], m2[B]      (i = 0..levels-1)
// ---------------------------------------------------------------------------------------------------
int Engine::forward_train(hipStream_t s, int B, const float* masks) {
    const std::string V = "volume_net";
    const int J = cfg.num_joints, Lv = cfg.levels, L1 = Lv + 1, C = cfg.embed_dim_ratio, D = C * L1;
    const int NH = cfg.deform_heads, NS = cfg.deform_samples, HD = C / NH;
    TrainLayout L;
    train_layout(B, L);
    float* tw = ws + ws_elems_per_frame * (size_t)B;
    float* X = tw + L.X;
    const float* m_ctx = masks;
    const float* m_res = masks ? masks + (size_t)2 * Lv * B : nullptr;
    const float* m_joint = masks ? m_res + (size_t)2 * Lv * B * J : nullptr;

    if (int rc = t_h2_prepare(s, L, tw, B)) return rc;
    HIP_TRY(launch_prep_embed(kcrop, k2d, P(*this, V + ".coord_embed.weight"), P(*this, V + ".coord_embed.bias"),
                              P(*this, V + ".Spatial_pos_embed"), X, B, J, L1, C, s));
    const float* pos = P(*this, V + ".Spatial_pos_embed");
    for (int l = 0; l < Lv; ++l) {
        const std::string fe = V + ".feat_embed." + std::to_string(l);
        float* S = tw + L.S[l];
        HIP_TRY(launch_sample_ref(bptr(feat_buf[l], B), kcrop, S, nullptr, B, J, feat_H[l], feat_W[l], feat_C[l], s, bf16() ? 1 : 0));
        int rc = t_gemm(s, S, row_ld(feat_C[l]), B * J, C, feat_C[l], P(*this, fe + ".weight"), feat_C[l],
                        P(*this, fe + ".bias"), X, row_ld(D, (long)(1 + l) * C), pos, RowMap{J, 0, C, (long)(1 + l) * J * C},
                        ACT_NONE, nullptr, 1);
        if (rc) return rc;
    }
    const RowMap tok{Lv, D, C, C}, tok0{Lv, D, 0, 0};
    for (int i = 0; i < Lv && cfg.context_blocks; ++i) {
        const std::string p = V + ".context_blocks." + std::to_string(i);
        const TrainLayout::Ctx& c = L.ctx[i];
        const int R = B * J * Lv;
        const float* m1 = m_ctx ? m_ctx + (size_t)(2 * i) * B : nullptr;
        const float* m2 = m_ctx ? m_ctx + (size_t)(2 * i + 1) * B : nullptr;
        HIP_TRY(launch_layernorm_train(X, tok, X, tok0, P(*this, p + ".norm1.weight"), P(*this, p + ".norm1.bias"), 1e-5f,
                                       tw + c.y1, tw + c.xh1, tw + c.rs1, R, C, s));
        const Pack& pk = packs[ctx_ao_pack[i]];
        int rc = t_gemm(s, tw + c.y1, row_ld(C), R, 3 * NH * NS, C, pack_arena + pk.w_off, pk.Kpad, pack_arena + pk.b_off,
                        tw + c.ao, row_ld(64), nullptr, row_ld(64), ACT_NONE, nullptr, 1);
        if (rc) return rc;
        DeformArgs da{};
        for (int l = 0; l < Lv; ++l) {
            da.feat[l] = bptr(feat_buf[l], B);
            da.H[l] = feat_H[l]; da.W[l] = feat_W[l]; da.C[l] = feat_C[l];
            da.U[l] = tw + c.U[l];
        }
        da.AO = tw + c.ao; da.ref = kcrop; da.B = B; da.J = J; da.L = Lv; da.NH = NH; da.NS = NS; da.ld_ao = 64;
        da.feat_bf16 = bf16() ? 1 : 0;
        if (debug) { da.cpos = bptr(ctx_tap_pos[i], B); da.cidx = reinterpret_cast<int*>(bptr(ctx_tap_idx[i], B)); }
        HIP_TRY(launch_deform_sample(da, s));
        for (int l = 0; l < Lv; ++l) {
            const std::string ep = p + ".embed_proj." + std::to_string(l);
            const RowMap dst{NH, D, HD, (long)(1 + l) * C};
            rc = t_gemm(s, tw + c.U[l], row_ld(feat_C[l]), B * J * NH, HD, feat_C[l], P(*this, ep + ".weight"), feat_C[l],
                        P(*this, ep + ".bias"), X, dst, X, dst, ACT_NONE, m1, J * NH);
            if (rc) return rc;
        }
        HIP_TRY(launch_layernorm_train(X, tok, nullptr, row_ld(0), P(*this, p + ".norm2.weight"), P(*this, p + ".norm2.bias"),
                                       1e-5f, tw + c.y2, tw + c.xh2, tw + c.rs2, R, C, s));
        rc = t_gemm(s, tw + c.y2, row_ld(C), R, 2 * C, C, P(*this, p + ".mlp.fc1.weight"), C, P(*this, p + ".mlp.fc1.bias"),
                    tw + c.hp, row_ld(2 * C), nullptr, row_ld(0), ACT_NONE, nullptr, 1);
        if (rc) return rc;
        HIP_TRY(launch_gelu_fwd(tw + c.hp, tw + c.hg, (long)R * 2 * C, s));
        rc = t_gemm(s, tw + c.hg, row_ld(2 * C), R, C, 2 * C, P(*this, p + ".mlp.fc2.weight"), 2 * C,
                    P(*this, p + ".mlp.fc2.bias"), X, tok, X, tok, ACT_NONE, m2, J * Lv);
        if (rc) return rc;
    }
    for (int g = 0; g < 2; ++g) {
        const int dim = g == 0 ? C : D, R = g == 0 ? B * J * L1 : B * J;
        const int tokens = g == 0 ? L1 : J, groups = g == 0 ? B * J : B, per = g == 0 ? L1 : J;
        for (int i = 0; i < Lv; ++i) {
            const std::string p = V + (g == 0 ? ".res_blocks." : ".joint_blocks.") + std::to_string(i);
            const TrainLayout::Att& a = g == 0 ? L.res[i] : L.joint[i];
            const float* mb = g == 0 ? m_res : m_joint;
            const size_t ms = g == 0 ? (size_t)B * J : (size_t)B;
            const float* m1 = mb ? mb + (size_t)(2 * i) * ms : nullptr;
            const float* m2 = mb ? mb + (size_t)(2 * i + 1) * ms : nullptr;
            HIP_TRY(launch_layernorm_train(X, row_ld(dim), nullptr, row_ld(0), P(*this, p + ".norm1.weight"),
                                           P(*this, p + ".norm1.bias"), 1e-6f, tw + a.y1, tw + a.xh1, tw + a.rs1, R, dim, s));
            int rc = t_gemm(s, tw + a.y1, row_ld(dim), R, 3 * dim, dim, P(*this, p + ".attn.qkv.weight"), dim,
                            P(*this, p + ".attn.qkv.bias"), tw + a.qkv, row_ld(3 * dim), nullptr, row_ld(0), ACT_NONE, nullptr, 1);
            if (rc) return rc;
            HIP_TRY(launch_attention(tw + a.qkv, tw + a.o, groups, tokens, cfg.num_heads, dim / cfg.num_heads, s));
            rc = t_gemm(s, tw + a.o, row_ld(dim), R, dim, dim, P(*this, p + ".attn.proj.weight"), dim,
                        P(*this, p + ".attn.proj.bias"), X, row_ld(dim), X, row_ld(dim), ACT_NONE, m1, per);
            if (rc) return rc;
            HIP_TRY(launch_layernorm_train(X, row_ld(dim), nullptr, row_ld(0), P(*this, p + ".norm2.weight"),
                                           P(*this, p + ".norm2.bias"), 1e-6f, tw + a.y2, tw + a.xh2, tw + a.rs2, R, dim, s));
            rc = t_gemm(s, tw + a.y2, row_ld(dim), R, 2 * dim, dim, P(*this, p + ".mlp.fc1.weight"), dim,
                        P(*this, p + ".mlp.fc1.bias"), tw + a.hp, row_ld(2 * dim), nullptr, row_ld(0), ACT_NONE, nullptr, 1);
            if (rc) return rc;
            HIP_TRY(launch_gelu_fwd(tw + a.hp, tw + a.hg, (long)R * 2 * dim, s));
            rc = t_gemm(s, tw + a.hg, row_ld(2 * dim), R, dim, 2 * dim, P(*this, p + ".mlp.fc2.weight"), 2 * dim,
                        P(*this, p + ".mlp.fc2.bias"), X, row_ld(dim), X, row_ld(dim), ACT_NONE, m2, per);
            if (rc) return rc;
        }
    }
    HIP_TRY(launch_layernorm_train(X, row_ld(D), nullptr, row_ld(0), P(*this, V + ".head.0.weight"), P(*this, V + ".head.0.bias"),
                                   1e-5f, tw + L.yh, tw + L.xhh, tw + L.rsh, B * J, D, s));
    HIP_TRY(launch_head(X, P(*this, V + ".head.0.weight"), P(*this, V + ".head.0.bias"), 1e-5f, P(*this, V + ".head.1.weight"),
                        P(*this, V + ".head.1.bias"), out, B * J, D, 3, s));
    train_batch = B;
    return CAPF_OK;
}

// ---------------------------------------------------------------------------------------------------
// backward: dOut [B,1,17,3] -> flat_grad (state_dict order of volume_net.*)
// ---------------------------------------------------------------------------------------------------
int Engine::backward(hipStream_t s, int B, const float* dOut, float* flat, const float* masks) {
    if (B != train_batch) {
        err = "capf_backward: the activations of the matching capf_forward_train are gone (no such call, another batch size, or "
              "a later forward / workspace change overwrote them)";
        return CAPF_ERR_STATE;
    }
    const std::string V = "volume_net";
    const int J = cfg.num_joints, Lv = cfg.levels, L1 = Lv + 1, C = cfg.embed_dim_ratio, D = C * L1;
    const int NH = cfg.deform_heads, NS = cfg.deform_samples, HD = C / NH;
    TrainLayout L;
    train_layout(B, L);
    float* tw = ws + ws_elems_per_frame * (size_t)B;
    t_slab_jobs.count = 0;                                   // (an earlier backward that failed half-way may have left jobs behind)
    t_slab_cur = 0;
    t_col_jobs.count = 0;
    t_col_cur = 0;
    float* dX = tw + L.dX;
    float* gA = tw + L.gA;
    float* gB = tw + L.gB;
    auto G = [&](const std::string& n) { return flat + grad_off[param_index.at(n)]; };
    const float* m_ctx = masks;
    const float* m_res = masks ? masks + (size_t)2 * Lv * B : nullptr;
    const float* m_joint = masks ? m_res + (size_t)2 * Lv * B * J : nullptr;

    HIP_TRY(hipMemsetAsync(dX, 0, sizeof(float) * (size_t)B * J * D, s));

    // ---- head: out = LN(X) W^T + b
    {
        const int R = B * J;
        if (int rc2 = t_colreduce(s, L, tw, dOut, row_ld(3), nullptr, row_ld(0), 0, R, 3, G(V + ".head.1.bias"), 1, nullptr, 0)) return rc2;
        for (int o = 0; o < 3; ++o)
            if (int rc2 = t_colreduce(s, L, tw, tw + L.yh, row_ld(D), dOut + o, row_ld(3), 2, R, D, G(V + ".head.1.weight") + (size_t)o * D, 1, nullptr, 0)) return rc2;
        HIP_TRY(launch_head_dgrad(dOut, P(*this, V + ".head.1.weight"), gA, R, D, 3, s));
        if (int rc2 = t_colreduce(s, L, tw, gA, row_ld(D), tw + L.xhh, row_ld(D), 1, R, D, G(V + ".head.0.weight"), 1, G(V + ".head.0.bias"), L.red_cap)) return rc2;      // d(gamma) and d(beta) in one pass
        HIP_TRY(launch_layernorm_bwd(gA, tw + L.xhh, tw + L.rsh, P(*this, V + ".head.0.weight"), dX, row_ld(D), nullptr, row_ld(0), R, 1, D, s));
    }

    // ---- the MLP half of any block: x_out = x + m2 * fc2(gelu(fc1(LN2(x))))
    auto mlp_bwd = [&](const std::string& p, RowMap xm, int R, int dim, float eps_unused, size_t xh2, size_t rs2, size_t y2,
                       size_t hp, size_t hg, const float* m2, int div) -> int {
        (void)eps_unused;
        const float* dBr = dX;
        RowMap dm = xm;
        if (m2) {      // gradient of the branch output = dX * mask
            HIP_TRY(launch_scale_rows(dX, xm, m2, div, gB, R, dim, s));
            dBr = gB;
            dm = row_ld(dim);
        }
        int rc = t_linear_bwd(s, L, tw, dBr, dm, R, dim, 2 * dim, tw + hg, row_ld(2 * dim), P(*this, p + ".mlp.fc2.weight"), gA,
                              row_ld(2 * dim), false, G(p + ".mlp.fc2.weight"), G(p + ".mlp.fc2.bias"));
        if (rc) return rc;
        HIP_TRY(launch_gelu_bwd(tw + hp, gA, gA, (long)R * 2 * dim, s));
        rc = t_linear_bwd(s, L, tw, gA, row_ld(2 * dim), R, 2 * dim, dim, tw + y2, row_ld(dim), P(*this, p + ".mlp.fc1.weight"), gB,
                          row_ld(dim), false, G(p + ".mlp.fc1.weight"), G(p + ".mlp.fc1.bias"));
        if (rc) return rc;
        if (int rc2 = t_colreduce(s, L, tw, gB, row_ld(dim), tw + xh2, row_ld(dim), 1, R, dim, G(p + ".norm2.weight"), 1, G(p + ".norm2.bias"), L.red_cap)) return rc2;      // d(gamma) and d(beta) in one pass
        HIP_TRY(launch_layernorm_bwd(gB, tw + xh2, tw + rs2, P(*this, p + ".norm2.weight"), dX, xm, nullptr, row_ld(0), R, 1, dim, s));
        return CAPF_OK;
    };

    // ---- attention blocks, joint group first (reverse of the forward)
    for (int g = 1; g >= 0; --g) {
        const int dim = g == 0 ? C : D, R = g == 0 ? B * J * L1 : B * J;
        const int tokens = g == 0 ? L1 : J, groups = g == 0 ? B * J : B, per = g == 0 ? L1 : J;
        for (int i = Lv - 1; i >= 0; --i) {
            const std::string p = V + (g == 0 ? ".res_blocks." : ".joint_blocks.") + std::to_string(i);
            const TrainLayout::Att& a = g == 0 ? L.res[i] : L.joint[i];
            const float* mb = g == 0 ? m_res : m_joint;
            const size_t ms = g == 0 ? (size_t)B * J : (size_t)B;
            const float* m1 = mb ? mb + (size_t)(2 * i) * ms : nullptr;
            const float* m2 = mb ? mb + (size_t)(2 * i + 1) * ms : nullptr;
            int rc = mlp_bwd(p, row_ld(dim), R, dim, 0.f, a.xh2, a.rs2, a.y2, a.hp, a.hg, m2, per);
            if (rc) return rc;
            const float* dBr = dX;
            if (m1) {
                HIP_TRY(launch_scale_rows(dX, row_ld(dim), m1, per, gB, R, dim, s));
                dBr = gB;
            }
            rc = t_linear_bwd(s, L, tw, dBr, row_ld(dim), R, dim, dim, tw + a.o, row_ld(dim), P(*this, p + ".attn.proj.weight"), gA,
                              row_ld(dim), false, G(p + ".attn.proj.weight"), G(p + ".attn.proj.bias"));
            if (rc) return rc;
            HIP_TRY(launch_attention_bwd(tw + a.qkv, gA, gB, groups, tokens, cfg.num_heads, dim / cfg.num_heads, s));
            rc = t_linear_bwd(s, L, tw, gB, row_ld(3 * dim), R, 3 * dim, dim, tw + a.y1, row_ld(dim), P(*this, p + ".attn.qkv.weight"),
                              gA, row_ld(dim), false, G(p + ".attn.qkv.weight"), G(p + ".attn.qkv.bias"));
            if (rc) return rc;
            if (int rc2 = t_colreduce(s, L, tw, gA, row_ld(dim), tw + a.xh1, row_ld(dim), 1, R, dim, G(p + ".norm1.weight"), 1, G(p + ".norm1.bias"), L.red_cap)) return rc2;      // d(gamma) and d(beta) in one pass
            HIP_TRY(launch_layernorm_bwd(gA, tw + a.xh1, tw + a.rs1, P(*this, p + ".norm1.weight"), dX, row_ld(dim), nullptr, row_ld(0), R, 1, dim, s));
        }
    }

    // ---- deformable context blocks
    const RowMap tok{Lv, D, C, C}, tok0{Lv, D, 0, 0};
    for (int i = Lv - 1; i >= 0 && cfg.context_blocks; --i) {
        const std::string p = V + ".context_blocks." + std::to_string(i);
        const TrainLayout::Ctx& c = L.ctx[i];
        const int R = B * J * Lv;
        const float* m1 = m_ctx ? m_ctx + (size_t)(2 * i) * B : nullptr;
        const float* m2 = m_ctx ? m_ctx + (size_t)(2 * i + 1) * B : nullptr;
        int rc = mlp_bwd(p, tok, R, C, 0.f, c.xh2, c.rs2, c.y2, c.hp, c.hg, m2, J * Lv);
        if (rc) return rc;
        DeformArgs da{};
        for (int l = 0; l < Lv; ++l) {
            const std::string ep = p + ".embed_proj." + std::to_string(l);
            const RowMap src{NH, D, HD, (long)(1 + l) * C};      // row (b,p,h) -> dX[b,p,1+l,h*HD:]
            const float* dBr = dX;
            RowMap dm = src;
            if (m1) {
                HIP_TRY(launch_scale_rows(dX, src, m1, J * NH, gB, B * J * NH, HD, s));
                dBr = gB;
                dm = row_ld(HD);
            }
            rc = t_linear_bwd(s, L, tw, dBr, dm, B * J * NH, HD, feat_C[l], tw + c.U[l], row_ld(feat_C[l]), P(*this, ep + ".weight"),
                              tw + L.dU[l], row_ld(feat_C[l]), false, G(ep + ".weight"), G(ep + ".bias"));
            if (rc) return rc;
            da.feat[l] = bptr(feat_buf[l], B);
            da.H[l] = feat_H[l]; da.W[l] = feat_W[l]; da.C[l] = feat_C[l];
            da.dU[l] = tw + L.dU[l];
        }
        da.AO = tw + c.ao; da.ref = kcrop; da.B = B; da.J = J; da.L = Lv; da.NH = NH; da.NS = NS; da.ld_ao = 64;
        da.feat_bf16 = bf16() ? 1 : 0;
        HIP_TRY(launch_deform_bwd(da, gA, 64, s));                                  // gA = dAO [R, 64]
        // [attention_weights | sampling_offsets] were one GEMM with N = 48: gradients land in a [48, C] temp
        const Pack& pk = packs[ctx_ao_pack[i]];
        const int NA = NH * NS, NO = 2 * NH * NS;
        float* gWcat = tw + L.cat;                      // [48][C]
        float* gbcat = gWcat + (size_t)(NH * NS * 3) * C;   // [48], right behind the weight gradient: one launch leaves both
        float* dq = tw + L.gC;                          // [R, C]
        rc = t_linear_bwd(s, L, tw, gA, row_ld(64), R, NA + NO, C, tw + c.y1, row_ld(C), pack_arena + pk.w_off, dq,
                          row_ld(C), false, gWcat, gbcat);
        if (rc) return rc;
        if ((rc = t_slab_flush(s))) return rc;           // (the copies below read the temp: its slabs are summed now, with whatever else waits)
        HIP_TRY(hipMemcpyAsync(G(p + ".attention_weights.weight"), gWcat, sizeof(float) * NA * C, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(G(p + ".sampling_offsets.weight"), gWcat + (size_t)NA * C, sizeof(float) * NO * C, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(G(p + ".attention_weights.bias"), gbcat, sizeof(float) * NA, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(G(p + ".sampling_offsets.bias"), gbcat + NA, sizeof(float) * NO, hipMemcpyDeviceToDevice, s));
        if (int rc2 = t_colreduce(s, L, tw, dq, row_ld(C), tw + c.xh1, row_ld(C), 1, R, C, G(p + ".norm1.weight"), 1, G(p + ".norm1.bias"), L.red_cap)) return rc2;      // d(gamma) and d(beta) in one pass
        HIP_TRY(launch_layernorm_bwd(dq, tw + c.xh1, tw + c.rs1, P(*this, p + ".norm1.weight"), dX, tok, dX, tok0, R, Lv, C, s));
    }

    // ---- embeddings: X[b,p,0] = coord_embed(k2d) + pos[0,p];  X[b,p,1+l] = feat_embed_l(S_l) + pos[1+l,p]
    {
        const int R = B * J;
        for (int l = 0; l < Lv; ++l) {
            const std::string fe = V + ".feat_embed." + std::to_string(l);
            int rc = t_linear_bwd(s, L, tw, dX, row_ld(D, (long)(1 + l) * C), R, C, feat_C[l], tw + L.S[l], row_ld(feat_C[l]),
                                  P(*this, fe + ".weight"), nullptr, row_ld(0), false, G(fe + ".weight"), G(fe + ".bias"));
            if (rc) return rc;
        }
        if (int rc2 = t_colreduce(s, L, tw, dX, row_ld(D), nullptr, row_ld(0), 0, R, C, G(V + ".coord_embed.bias"), 1, nullptr, 0)) return rc2;
        for (int j = 0; j < 2; ++j)
            if (int rc2 = t_colreduce(s, L, tw, dX, row_ld(D), k2d + j, row_ld(2), 2, R, C, G(V + ".coord_embed.weight") + j, 2, nullptr, 0)) return rc2;
        HIP_TRY(launch_pos_grad(dX, G(V + ".Spatial_pos_embed"), B, J, L1, C, s));
    }
    if (int rc = t_col_flush(s)) return rc;                  // every bias / LayerNorm gradient still in partial sums
    return t_slab_flush(s);                                  // every weight gradient still in slabs
}

}  // namespace capf
