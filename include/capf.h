/*
 * capf.h — C ABI of the MI355X-native Context-Aware PoseFormer hot path.
 *
 * The reference (QitaoZhao/ContextAware-PoseFormer) is pure Python/PyTorch: it has no FFI, plugin or
 * operator registry.  Its hot path sits behind one nn.Module call,
 *
 *     CA_PF.forward(images[B,H,W,3], keypoints_2d[B,17,2], keypoints_2d_crop[B,17,2]) -> [B,1,17,3]
 *                                                   ContextPose/mvn/models/conpose.py:30-42
 *
 * so the "binding a maintainer would add" is a ctypes stub inside that forward.
 * Every entry point below names the reference code it replaces.  Conventions:
 *   - plain C types only; device memory is caller-owned (the PyTorch host module keeps owning
 *     parameters, so optimizers / DDP / checkpoints keep working); the library borrows pointers.
 *   - every call returns 0 on success or a negative capf_status; capf_last_error() has the text.
 *   - stream-ordered, no hidden synchronisation, no allocation inside capf_forward.
 *   - a handle is not thread-safe; one handle per (process, device).
 */
#ifndef CAPF_H
#define CAPF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libcapf.so is built with -fvisibility=hidden: the functions declared in this header are its whole dynamic symbol table */
#pragma GCC visibility push(default)

typedef struct capf_handle capf_handle;

enum capf_status {
    CAPF_OK = 0,
    CAPF_ERR_INVALID = -1,     /* bad argument / unknown name / shape mismatch */
    CAPF_ERR_UNSUPPORTED = -2, /* configuration the kernels do not cover */
    CAPF_ERR_HIP = -3,         /* a HIP runtime call or kernel launch failed */
    CAPF_ERR_STATE = -4        /* call order: params / workspace not set */
};

enum capf_backbone { CAPF_HRNET = 0, CAPF_CPN50 = 1 };
enum capf_dtype { CAPF_F32 = 0, CAPF_BF16 = 1 };

/* parameter kinds reported by capf_param_info (lets the host build matching leaf modules) */
enum capf_param_kind {
    CAPF_P_CONV_W = 0,   /* nn.Conv2d.weight  [Cout,Cin,kh,kw]  (bias=False everywhere) */
    CAPF_P_BN_W = 1, CAPF_P_BN_B = 2, CAPF_P_BN_MEAN = 3, CAPF_P_BN_VAR = 4, CAPF_P_BN_NBT = 5,
    CAPF_P_LIN_W = 6, CAPF_P_LIN_B = 7,     /* nn.Linear [out,in], [out] */
    CAPF_P_LN_W = 8, CAPF_P_LN_B = 9,       /* nn.LayerNorm */
    CAPF_P_RAW = 10                         /* bare nn.Parameter (Spatial_pos_embed) */
};

/*
 * Model description == the subset of the reference's global config that CA_PF.__init__ reads:
 *   config.model.backbone.{type,STAGE2..4.NUM_CHANNELS,NUM_MODULES,NUM_BLOCKS}  (conpose.py:14-20,
 *       pose_hrnet.py:330-370, mvn/utils/cfg.py:24-66)
 *   config.model.poseformer.{base_dim,embed_dim_ratio,levels}                    (pose_dformer.py:167-172)
 */
typedef struct capf_config {
    int32_t backbone;          /* capf_backbone */
    int32_t hr_channels[4];    /* HRNet branch widths: {32,64,128,256} (W32) / {48,96,192,384} (W48) */
    int32_t hr_modules[3];     /* NUM_MODULES of stage2..4: {1,4,3} */
    int32_t hr_blocks;         /* NUM_BLOCKS per branch: 4 */
    int32_t base_dim;          /* poseformer.base_dim: 32 / 48 / 256(cpn) */
    int32_t embed_dim_ratio;   /* 128 */
    int32_t levels;            /* 4 feature levels (H36M model: also the depth of every block group, pose_dformer.py:169) */
    int32_t num_joints;        /* 17 */
    int32_t num_heads;         /* 8  (Block) */
    int32_t deform_heads;      /* 4  (DeformableBlock, pose_dformer.py:202) */
    int32_t deform_samples;    /* 4 */
    int32_t context_blocks;    /* 1 = H36M model; 0 = MPI-INF-3DHP variant without DeformableBlocks */
    int32_t compute_dtype;     /* capf_dtype: MFMA operand type of the backbone convs / lifter GEMMs */
    int32_t max_batch;         /* workspace is sized for this many frames */
    int32_t height, width;     /* input image size (256x256, 256x192, 384x288, ...) */
    int32_t training;          /* 1: size the workspace for capf_forward_train / capf_backward as well */
    int32_t plan_flags;        /* 0 = the product plan.  capf_plan_flag bits take one kernel family out of the plan (parity
                                  tests compare the two routes; nothing else -- no environment variable -- changes a plan) */
    int32_t depth;             /* blocks per group (res_blocks / joint_blocks).  0 = levels.  The MPI-INF-3DHP variant reads it from
                                  config.model.poseformer.depth (ContextPose_mpi/model/pose_dformer.py:199, 217-227; 1..8); the H36M
                                  model has depth == levels by construction, and so does the training path */
} capf_config;

enum capf_plan_flag {
    CAPF_PLAN_NO_FUSED_LIFTER = 1,  /* one kernel per lifter op instead of embed_kernel / ctx_attn_kernel / LayerNorm-in-GEMM */
    CAPF_PLAN_NO_WINOGRAD = 2,      /* fp32 3x3 stride-1 convs on the direct MFMA kernel */
    CAPF_PLAN_NO_ROW_HALO = 4,      /* bf16 3x3 stride-1 convs on the direct bf16 kernel */
    CAPF_PLAN_WINOGRAD_F23_ONLY = 8, /* F(2,3) where F(4,3) would be chosen */
    CAPF_PLAN_NO_PWCHAIN = 16,      /* layer1's conv3 -> next conv1 pairs as two pointwise launches instead of one chained kernel */
    CAPF_PLAN_NO_WS = 32,           /* bf16 3x3 stride-1 convs without the 2-D halo tile (row-halo / direct kernels as in round 3) */
    CAPF_PLAN_LIFTER_FP32 = 64,     /* compute_dtype = CAPF_BF16: keep the lifter's qkv / proj / fc1 / fc2 on the fp32 kernels (bf16 backbone only);
                                     * the accuracy / speed trade bench.py reports as `vs_fp32_oracle`                                    */
    CAPF_PLAN_NO_F32X3 = 128,       /* fp32 3x3 stride-1 convs without either split-fp32 tile (igemm_f32h2_ws.hip / igemm_f32x3_ws.hip):
                                     * the Winograd kernels on the fp32 matrix pipe (from batch 24; the direct kernel below)              */
    CAPF_PLAN_NO_F32H2_GEMM = 512,  /* every OTHER fp32 conv / linear (1x1, stride 2, lone convs, the lifter's projections -- forward, input and weight
                                     * gradients of a training step included) on the fp32 matrix pipe at every batch (igemm_f32.hip) instead of the
                                     * two-fp16-piece GEMM from batch 5 (igemm_f32h2.hip)                                                        */
    CAPF_PLAN_NO_UPADD = 1024,      /* compute_dtype = CAPF_BF16, CPN: globalNet's `lateral + upsampled path` (globalNet.py:66) as a resize-add launch behind the
                                     * lateral conv (round 5's plan) instead of inside the conv's epilogue (igemm_bf16_kernel<.., UPADD>)        */
    CAPF_PLAN_H2_PLANES = 2048,     /* OPT-IN (measured slower, EXPERIMENTS R6.5): a BasicBlock's conv1 output travels to conv2 as split fp16 planes
                                     * (capf_op_conv_f32h2_planes) where both convs run the two-fp16-piece tile; default: plain fp32 tensors          */
    CAPF_PLAN_NO_BNECK = 4096,      /* compute_dtype = CAPF_BF16: the first bottleneck of layer1 (conv1 / conv2 / downsample / conv3; networks/resnet.py:58-93,
                                     * pose_hrnet.py:98-136) as its five launches instead of ONE kernel with t1 / t2 / the shortcut on chip (bneck_bf16.hip) */
    CAPF_PLAN_NO_BATCHED_REDUCE = 8192, /* training: every weight gradient's slab sum and every bias / LayerNorm gradient's second reduction stage as its
                                     * own launch behind its producer (96 launches per step) instead of a handful of batched ones when the scratch
                                     * fills and at the end of capf_backward (csrc/train.cpp t_slab_flush / t_col_flush).  Same summation order per
                                     * element either way: bit-identical gradients (tests/test_gpu_train.py); an A/B aid                        */
    CAPF_PLAN_F32X3_EXACT = 256     /* ... on round 4's tile instead of the default one: every operand split EXACTLY into three bf16 pieces,
                                     * six piece products per fp32 MAC (igemm_f32x3_ws.hip).  The default (ABI 5) carries an operand as two
                                     * block-scaled fp16 pieces (to 2^-23) and issues three products: half the MFMAs, the same measured
                                     * distance to an fp64 evaluation (see capf_op_conv_f32h2_group)                                      */
};

/* ---- lifetime -------------------------------------------------------------------------------
 * Replaces CA_PF.__init__ (conpose.py:10-27): builds the layer plan (pose_hrnet.py:312-462 /
 * networks/network.py:9-28 / pose_dformer.py:144-208) and the parameter schema.
 * device < 0: "plan only" — no HIP call is made; schema / workspace queries work (CPU tests).   */
int capf_create(const capf_config* cfg, int device, capf_handle** out);
void capf_destroy(capf_handle* h);
/* Largest batch one call accepts: min(cfg.max_batch, what the kernels' 32-bit tensor addressing allows at this input
 * size — about 3800 frames for HRNet at 256x256).  Larger batches return CAPF_ERR_UNSUPPORTED with a clear message. */
int capf_max_batch(const capf_handle* h);
const char* capf_last_error(const capf_handle* h);  /* h may be NULL: last create error */
const char* capf_version(void);
/* ABI revision of THIS header.  A caller compiled against capf.h checks capf_abi_version() == CAPF_ABI_VERSION before it passes structs
 * (capf_config, capf_conv_desc, capf_op_desc) across the boundary: revision 5 = round 5 (capf_op_desc as declared below -- 240 bytes since
 * revision 4, 104 before --, plan flags up to CAPF_PLAN_F32X3_EXACT, the capf_op_*_f32h2 entry points, capf_op_describe_sized).  The
 * version string carries the same number ("capf 0.5 (gfx950)").                                                                            */
#define CAPF_ABI_VERSION 6
int capf_abi_version(void);

/* ---- parameter schema == the reference's state_dict ------------- */
int capf_num_params(const capf_handle* h);
int capf_param_info(const capf_handle* h, int index, const char** name, int64_t shape[4],
                    int* ndim, int* kind);

/* Borrow a device pointer for one state_dict entry (fp32; BN num_batches_tracked is ignored).
 * Replaces nn.Module parameter lookup during forward.  The pointer must stay valid until the
 * next capf_set_param for that name or capf_destroy. */
int capf_set_param(capf_handle* h, const char* name, const void* dev_ptr, const int64_t* shape,
                   int ndim);

/* Re-derive the private packed copies (BN folded into conv weights+bias, weights re-laid out
 * K-major for the MFMA kernels, bf16 copies).  Call after load_state_dict / optimizer.step().
 * Stream-ordered on `stream` (hipStream_t passed as void*). */
int capf_params_changed(capf_handle* h, void* stream);

/* Same, restricted to the private copies derived from volume_net.* parameters (the frozen backbone's
 * folded conv weights are left alone): what a training step needs after its optimizer update. */
int capf_lifter_params_changed(capf_handle* h, void* stream);

/* ---- workspace (activations); caller-owned so the host allocator (torch) stays in charge ----- */
size_t capf_workspace_bytes(const capf_handle* h, int batch);
int capf_set_workspace(capf_handle* h, void* dev_ptr, size_t bytes);

/* ---- the hot path ----------------------------------------------------------------------------
 * Replaces CA_PF.forward (conpose.py:30-42) = NHWC->NCHW permute (:32, folded into the stem conv's
 * loads), in-place crop-keypoint normalisation (:34-35, done in place on kcrop_inout exactly like
 * the reference), backbone forward (:38, pose_hrnet.py:464-501 / networks/network.py:16-22) and
 * PoseTransformer.forward (:40, pose_dformer.py:210-241).
 *   images_nhwc [B,H,W,3] fp32, k2d [B,17,2] fp32, kcrop_inout [B,17,2] fp32 (MUTATED),
 *   out [B,1,17,3] fp32.  All device pointers.  Enqueued on `stream`; returns immediately.      */
int capf_forward(capf_handle* h, void* stream, const float* images_nhwc, const float* k2d,
                 float* kcrop_inout, int batch, float* out);

/* Backbone only: writes nothing to `out`; the four context maps stay in the workspace (NHWC) and
 * can be read through capf_tensor("feat0".."feat3").  Replaces self.backbone(images) conpose.py:38. */
int capf_backbone_forward(capf_handle* h, void* stream, const float* images_nhwc, int batch);

/* ---- training step of the lifter (the backbone is frozen, conpose.py:22-25) --
 * capf_forward_train: capf_forward that keeps the lifter's intermediates for capf_backward.
 *     Replaces model(images, k2d, kcrop) under model.train() (train.py:183).  drop_masks: the DropPath
 *     multipliers (0 or 1/keep_prob; timm DropPath, pose_dformer.py:71,101) laid out as
 *     ctx[i]{m1[B],m2[B]} | res[i]{m1[B*17],m2[B*17]} | joint[i]{m1[B],m2[B]}, i = 0..levels-1; NULL = no drop.
 * capf_backward: replaces loss.backward() through the lifter (train.py:195): grad_out [B,1,17,3] ->
 *     flat_grad, one fp32 buffer holding the gradient of every volume_net.* parameter in schema order
 *     (capf_grad_info), each written exactly once (no atomics) — so ONE all-reduce covers DDP's traffic.
 * capf_mpjpe: MPJPE.forward (loss.py:16-22) + its gradient w.r.t. pred (scaled by grad_scale).
 * capf_adamw_step: torch.optim.AdamW update (train.py:345) of one flat parameter buffer.            */
int capf_forward_train(capf_handle* h, void* stream, const float* images_nhwc, const float* k2d,
                       float* kcrop_inout, int batch, float* out, const float* drop_masks);
int capf_backward(capf_handle* h, void* stream, const float* grad_out, int batch, float* flat_grad,
                  const float* drop_masks);
/* Saved activations live in the workspace: ANY later capf_forward* / capf_backbone_forward / capf_lifter_forward /
 * capf_set_workspace invalidates them, and capf_backward then returns CAPF_ERR_STATE instead of differentiating the
 * wrong step.  capf_train_generation changes with every such run: a host autograd node records it after its
 * capf_forward_train and compares before capf_backward (two forwards followed by one combined backward).
 * The PARAMETERS must not change between a capf_forward_train and its capf_backward either (an optimizer step belongs after the
 * backward, as in train.py:195-201): the backward multiplies by the packs of W^T that the forward built from the parameters it saw. */
int64_t capf_train_generation(const capf_handle* h);
int64_t capf_grad_elems(const capf_handle* h);
int capf_grad_info(const capf_handle* h, int param_index, int64_t* offset);   /* -1: not a lifter parameter */
/* How many of the lifter's nn.Linear matrices a training step at batch >= 5 multiplies by on the two-fp16-piece GEMM (forward y = x W^T
 * and / or backward dX = dY W; packs of W and W^T are rebuilt from the current parameters at the start of every capf_forward_train).
 * 0: every product of the step runs on the fp32 matrix pipe (CAPF_PLAN_NO_F32H2_GEMM, compute_dtype = bf16, training = 0). */
int capf_train_h2_matrices(const capf_handle* h);
int capf_mpjpe(void* stream, const float* pred, const float* gt, int rows, float* loss, float* dpred,
               float grad_scale);
/* the same loss for rows of any width `dim` (MPJPE.forward accepts [..., D]: 2-D keypoints as well, loss.py:16-22) */
int capf_mpjpe_nd(void* stream, const float* pred, const float* gt, int rows, int dim, float* loss, float* dpred,
                  float grad_scale);
/* grad_scale multiplies every gradient element on the way in: 1 / world_size turns the SUM all-reduce of the flat
 * gradient into DDP's average (train.py:361-362) without a separate pass over the 56 MB buffer; 1.0f is exact. */
int capf_adamw_step(void* stream, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                    int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    float grad_scale);

/* Lifter only, on the context maps left in the workspace by the last capf_backbone_forward /
 * capf_forward of the same batch.  Replaces self.volume_net(...) conpose.py:40
 * (PoseTransformer.forward pose_dformer.py:210-241).  kcrop_inout is normalised in place. */
int capf_lifter_forward(capf_handle* h, void* stream, const float* k2d, float* kcrop_inout, int batch,
                        float* out);

/* How the independent branches of the backbone (the 2-4 resolution branches of an HRNet module, the
 * fused outputs, the CPN refine cascades) are issued:
 *   0 = everything in program order on the caller's stream;
 *   1 = on library-owned side streams, forked from / joined to the caller's stream with events;
 *   2 = by dependency level on the caller's stream, the convolutions of a level sharing ONE grouped launch;
 *   3 = (default) as 2, but at batch 16..256 the lanes of a region form TWO such chains -- lanes 0 + 3 on the caller's stream,
 *       1 + 2 on a library-owned side stream (fork / join with events) -- so that one chain's launch ramp and tail overlap
 *       the other's body.
 * The results are bit-identical in modes 0, 2 and 3 (mode 1 has no split-K scratch: equal to roundoff at small batches). */
int capf_set_lanes(capf_handle* h, int on);

/* When on, forward also snapshots the token buffer after each block group (tok_ctx/tok_res/tok_joint). */
int capf_set_debug(capf_handle* h, int on);

/* Intermediates of the LAST forward, for stage-level parity tests.  Names:
 *   feat0..feat3   context maps, NHWC [B,h,w,C_l]
 *   sampled0..3    reference-point samples [B,17,C_l]          (pose_dformer.py:216-218)
 *   idx0..idx3     int32 [B,17,2] (ix0, iy0) bilinear NW corner of those samples (bit-exact check)
 *   cpos0..3       with capf_set_debug: sampling positions of DeformableBlock i, fp32 [B,17,L*16,2] (level, head*4+sample)
 *   cidx0..3       ... and the int32 NW corner (ix0, iy0) of each, padding_mode='border' (pose_dformer.py:126-128)
 *   tok_ctx / tok_res / tok_joint   token buffer [B,17,L+1,c] after each block group (layout b p l c)
 * Pointers are into the workspace and valid until the next forward on this handle.
 * Returns 0 for an fp32 tensor, 1 for an int32 tensor, 2 for a bf16 tensor (context maps of a
 * CAPF_BF16 handle), negative on error. */
int capf_tensor(const capf_handle* h, const char* name, const void** dev_ptr, int64_t shape[4],
                int* ndim);

/* Number of kernel launches and algorithmic FLOPs (2*MAC of convs + GEMMs) of one forward at
 * `batch`; used by bench.py for the roofline line. */
/* after capf_forward_profile_launches: for every leader op of a grouped bf16 conv launch, which device kernel the launch ran
 * (0 igemm_bf16_group_kernel: ring schedule, 1 igemm_bf16_group_pp_kernel: ping-pong, 2 igemm_bf16_group_rh_kernel: ping-pong
 * with row-halo tiles, 3 igemm_bf16_group_ws_kernel: the 2-D halo tile); -1 for every other op.  Lets bench.py name launches the way rocprofv3 does.                         */
int capf_forward_profile_variants(const capf_handle* h, int32_t* op_variant, int n_ops);
int capf_forward_stats(const capf_handle* h, int batch, int64_t* launches, double* flops);

/* ---- stateless operator entry points (op-level parity tests and micro-benchmarks) ---------------
 * capf_op_pack_conv : fold eval-mode BatchNorm into a conv weight and re-lay it out K-major:
 *     w_packed[Cout][Kpad] (Kpad = ks*ks*Cin rounded up to 32), bias[Cout]; gamma == NULL -> no BN.
 * capf_op_conv      : y = act(conv2d(x; w_packed) + bias (+ residual)), NHWC, padding = ks/2.
 *     == nn.Conv2d(bias=False) + nn.BatchNorm2d(eval) (+ residual add) (+ ReLU) of
 *     pose_hrnet.py:66-136 / networks/resnet.py:58-93 as ONE implicit-GEMM launch.
 * capf_op_linear    : y[M,N] = act(x[M,K] @ w[N,K]^T + bias (+ residual)); act 0 none, 1 ReLU, 2 GELU(erf)
 *     == nn.Linear (+GELU) of pose_dformer.py:15-31; K % 32 == 0.                                      */
int capf_op_pack_conv(void* stream, const float* w_oihw, const float* gamma, const float* beta,
                      const float* mean, const float* var, float eps, float* w_packed, float* bias,
                      int Cout, int Cin, int ks);
int capf_op_conv(void* stream, const float* x_nhwc, const float* w_packed, const float* bias,
                 const float* residual, float* y_nhwc, int B, int H, int W, int Cin, int Cout, int ks,
                 int stride, int act);
/* capf_op_conv_group : up to 8 INDEPENDENT capf_op_conv problems (Cin % 4 == 0, ks <= 5) as ONE grouped
 *     launch -- what the engine issues for the same-depth convs of the HRNet branches
 *     (pose_hrnet.py:242-255: the branches of a HighResolutionModule do not depend on each other) and of
 *     a fuse layer (:257-303).  Results are bit-identical to n capf_op_conv calls. */
typedef struct capf_conv_desc {
    const float* x;         /* [B,H,W,Cin] NHWC */
    const float* w_packed;  /* capf_op_pack_conv output */
    const float* bias;
    const float* residual;  /* [B,Ho,Wo,Cout] or NULL */
    float* y;               /* [B,Ho,Wo,Cout] */
    int32_t B, H, W, Cin, Cout, ks, stride, act;
} capf_conv_desc;
int capf_op_conv_group(void* stream, int n, const capf_conv_desc* convs);
/* The same 3x3 / stride-1 / pad-1 convolution through the Winograd-along-W kernels (csrc/igemm_wino.hip), results equal to
 * the direct kernel's to fp32 roundoff.  variant 23: F(2,3), 1.5x fewer MFMAs, wp [Cout][12 * Cin], even W;  variant 43:
 * F(4,3), 2x fewer MFMAs, wp [Cout][18 * Cin], W % 4 == 0.  capf_op_pack_conv_wino folds BatchNorm like capf_op_pack_conv
 * and applies the weight transform.  Needs Cin % 32 == 0, Cout % 4 == 0 (CAPF_ERR_UNSUPPORTED otherwise).
 * capf_op_conv_wino_group: up to 8 such convs of one variant in one grid (capf_conv_desc with ks 3, stride 1). */
int capf_op_pack_conv_wino(void* stream, const float* w, const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, float* wp, float* bias, int Cout, int Cin, int variant);
int capf_op_conv_wino(void* stream, const float* x, const float* wp, const float* bias, const float* residual, float* y, int B,
                      int H, int W, int Cin, int Cout, int act, int variant);
int capf_op_conv_wino_group(void* stream, int n, const capf_conv_desc* d, int variant);
int capf_op_linear(void* stream, const float* x, const float* w, const float* bias, const float* residual,
                   float* y, int M, int N, int K, int act);
/* capf_op_bilinear_corners: the corner rule of both sampling sites (F.grid_sample, bilinear, align_corners=True) on
 * caller-supplied normalised coordinates grid [n,2] = (x, y): idx [n,2] = NW corner (x0, y0), frac [n,2] = weights of the
 * +1 corners.  border = 1: padding_mode='border' (DeformableBlock, pose_dformer.py:128), 0: 'zeros' (:217).  The same
 * device function serves capf_forward; inside a forward the corners actually used are exposed through capf_tensor
 * ("idx{l}" / "cidx{i}" under capf_set_debug).  Must equal ATen's index arithmetic bit for bit.                     */
int capf_op_bilinear_corners(void* stream, const float* grid, int n, int H, int W, int border, int32_t* idx, float* frac);
/* bf16 twins (igemm_bf16.hip, v_mfma_f32_32x32x16_bf16): x / residual / y are bf16 NHWC, w_packed is bf16
 * [Cout][Kpad] with Kpad = ks*ks*Cin rounded up to 64, bias stays fp32.  Cin % 8 == 0, Cout % 4 == 0. */
int capf_op_pack_conv_bf16(void* stream, const float* w_oihw, const float* gamma, const float* beta,
                           const float* mean, const float* var, float eps, void* w_packed_bf16, float* bias,
                           int Cout, int Cin, int ks);
int capf_op_conv_bf16(void* stream, const void* x_nhwc_bf16, const void* w_packed_bf16, const float* bias,
                      const void* residual_bf16, void* y_nhwc_bf16, int B, int H, int W, int Cin, int Cout,
                      int ks, int stride, int act);

/* "Row-halo" variant of the 3x3 / stride-1 / pad-1 bf16 conv (what capf_forward runs for the BasicBlock convs of a bf16 model
 * from 2048 tiles per launch): K order (kh, Cin / cw, kw, cw) so that one staged activation tile serves the three kw taps.
 * cw = capf_op_conv_bf16_rh_width(Cin) (64, 48 or 32; 0 = Cin not supported); w_packed is bf16 [Cout][9 * Cin].            */
int capf_op_conv_bf16_rh_width(int Cin);
int capf_op_pack_conv_bf16_rh(void* stream, const float* w_oihw, const float* gamma, const float* beta, const float* mean,
                              const float* var, float eps, void* w_packed_bf16, float* bias, int Cout, int Cin);
int capf_op_conv_bf16_rh(void* stream, const void* x_nhwc_bf16, const void* w_packed_bf16, const float* bias,
                         const void* residual_bf16, void* y_nhwc_bf16, int B, int H, int W, int Cin, int Cout, int act);

/* "2-D halo" tile of the 3x3 / stride-1 / pad-1 bf16 conv (csrc/igemm_bf16_ws.hip; what capf_forward runs for the BasicBlock convs
 * of a bf16 model, pose_hrnet.py:66-95, and the 3x3 of the ResNet / refine bottlenecks, networks/resnet.py:58-93, from 512 tiles
 * per launch): a block keeps 256 output pixels x 32 / 64 / 96 channels in its accumulators for the whole K and stages every
 * 16-channel chunk of its pixels (with halo) once for all nine taps.  Cin % 16 == 0, Cout % 8 == 0, (rows + 2) x (W + 2) <= 416
 * for some row count dividing H.  w_packed holds capf_op_conv_bf16_ws_pack_elems(Cout, Cin) bf16 elements written by
 * capf_op_pack_conv_bf16_ws (BatchNorm folded as in capf_op_pack_conv_bf16; bias fp32 [Cout], may be NULL).
 * capf_op_conv_bf16_ws_group: up to 8 such convs in ONE grid, the way capf_forward issues a level; capf_conv_desc with bf16 x /
 * residual / y and w_packed in this layout (ks = 3, stride = 1).                                                                                              */
int64_t capf_op_conv_bf16_ws_pack_elems(int Cout, int Cin);
int capf_op_pack_conv_bf16_ws(void* stream, const float* w_oihw, const float* gamma, const float* beta, const float* mean,
                              const float* var, float eps, void* w_packed_bf16, float* bias, int Cout, int Cin);
int capf_op_conv_bf16_ws_group(void* stream, int n, const capf_conv_desc* convs);

/* Split-fp32 tile of the 3x3 / stride-1 / pad-1 fp32 conv (csrc/igemm_f32x3_ws.hip; what capf_forward runs for the BasicBlock convs of
 * an fp32 model, pose_hrnet.py:66-95, from 370 MFLOP per conv and batch 5): fp32 tensors in and out; every operand is split, exactly, into three
 * bf16 numbers and the six piece products of weight >= 2^-18 run on the bf16 matrix pipe with fp32 accumulation -- the dropped
 * products are below the rounding of one fp32 multiply, so results agree with the direct fp32 kernel to accumulation order.
 * (Non-finite / denormal operands: an infinite input gives NaN -- Inf - Inf in the split -- where the fp32 pipe gives +-Inf; fp32 denormals are
 * flushed.)  Cin % 16 == 0, Cout % 4 == 0, W <= 256.  w_packed holds capf_op_conv_f32x3_pack_elems(Cout, Cin) bf16 elements written by
 * capf_op_pack_conv_f32x3 (BatchNorm folded as in capf_op_pack_conv, then split; bias fp32 [Cout], may be NULL).
 * capf_op_conv_f32x3_group: up to 8 such convs in ONE grid (capf_conv_desc with w_packed in this layout, ks = 3, stride = 1).       */
int64_t capf_op_conv_f32x3_pack_elems(int Cout, int Cin);
int capf_op_pack_conv_f32x3(void* stream, const float* w_oihw, const float* gamma, const float* beta, const float* mean,
                            const float* var, float eps, void* w_packed_bf16, float* bias, int Cout, int Cin);
int capf_op_conv_f32x3_group(void* stream, int n, const capf_conv_desc* convs);

/* The default split-fp32 tile (csrc/igemm_f32h2_ws.hip; what capf_forward runs for those convs unless CAPF_PLAN_F32X3_EXACT / _NO_F32X3):
 * fp32 tensors in and out; an operand a travels as two fp16 numbers a1 = fp16(s a), a2 = fp16(s a - a1) under an exact power-of-two scale s
 * (weights: one per output channel, fixed at pack time; pixels: one per block of 256 output pixels and 16-channel chunk, found inside the
 * kernel from the values it staged -- nothing outside the kernel sees a scale), |s a - a1 - a2| <= 2^-23 |s a|; the three products
 * a1 w1 + a1 w2 + a2 w1 run on the 16-bit matrix pipe with fp32 accumulation (dropped: a2 w2 <= 2^-22 |a w|).  One term is therefore within
 * 2^-21 of the fp32 product, which over a dot product of 9 Cin terms is below the fp32 accumulation error of ANY fp32 evaluation: the tests
 * hold this tile to the same bound as the three-piece tile (<= 1e-6 of the sum of |terms| against fp64, <= 2 x this library's direct fp32
 * MFMA kernel on the same problem).  Ranges: values whose magnitude is below 2^-18 of their block's largest lose relative precision
 * (absolute error <= 2^-39 of that largest value); scales are kept within 2^+-63: block maxima in [2^-49, 2^77) = 1.8e-15 .. 1.5e23 get
 * their exact scale, smaller ones lose precision gradually (all-zero blocks are exact), a block maximum of 2^77 or more overflows fp16
 * (Inf, then NaN); Inf / NaN inputs give NaN for their whole block and chunk (CAPF_PLAN_NO_F32X3 keeps the fp32 pipe's IEEE behaviour).
 * THE BOUND, per output y = sum_k a_k w_k, with M_c the largest |a| among the values its block staged for 16-channel chunk c (256 output
 * pixels + their 1-pixel halo; for capf_op_*_f32h2_gemm: the 32 rows x 32-deep chunk of a wave) and W_c = sum over that chunk of |w_k|:
 *     |y - exact| <= 2.5e-6 sum_k |a_k w_k|  +  2^-38 sum_c M_c W_c
 * -- the first term is an fp32 accumulation's own (observed <= 1e-6 on every fuzzed problem, and at most 2 x this library's fp32-pipe kernel
 * on the same operands everywhere; a sum dominated by ONE term costs any fp32 evaluation 1 - 2e-6); the second only shows when a block holds
 * values more than 2^18 apart (one outlier pixel 2^20 above a flat tile costs the flat tile's outputs ~2e-6 of THEIR sum of |terms|).
 * tests/test_gpu_ops.py test_f32h2_dynamic_range_inside_a_block asserts exactly this with outliers of 2^8 .. 2^20 for the conv tile, the GEMM
 * (conv and rows mode; 3e-6 there) and the weight gradient (5e-6: 2048-term sums), and the plain 1e-6 for blocks that hold no outlier.
 * A scale follows the data downwards without limit and upwards by at most 2^80 above the smallest scale its tile has used (the accumulators
 * must not overflow behind a chunk of zeros).
 * BATCH COMPOSITION: a block's scale depends on everything the block stages.  Conv tiles never span frames unless H W < 256 (8x8 maps: four
 * frames per tile); the GEMM's 32-row blocks of the lifter's [B 17, K] operands do.  A frame's output BITS may therefore depend on its
 * neighbours in the batch (within the bound above); CAPF_PLAN_NO_F32X3 | CAPF_PLAN_NO_F32H2_GEMM gives batch-independent bits.  Same shapes
 * as above; w_packed holds capf_op_conv_f32h2_pack_elems(Cout, Cin) 16-bit elements (pieces, then the fp32 inverse channel scales).
 * Tensor sizes: a tile addresses its pixels and its output rows from per-tile bases, so x / y / residual may exceed 2 GiB (only B * H * W
 * has to stay below 2^31): conv2 and both transition1 convs at 512 frames (2.1 GB of fp32 each) run here.                                  */
int64_t capf_op_conv_f32h2_pack_elems(int Cout, int Cin);
int capf_op_pack_conv_f32h2(void* stream, const float* w_oihw, const float* gamma, const float* beta, const float* mean,
                            const float* var, float eps, void* w_packed_f16, float* bias, int Cout, int Cin);
int capf_op_conv_f32h2_group(void* stream, int n, const capf_conv_desc* convs);

/* PLANES (round 6): a tensor that only travels from one two-fp16-piece tile conv to the next -- a BasicBlock's conv1 -> conv2, pose_hrnet.py:78-88 --
 * can stay SPLIT in memory: per pixel and 16-channel chunk [piece 0: 16 fp16 | piece 1: 16 fp16], the same 64 bytes at the same addresses as the fp32
 * values, under one power-of-two scale per (256-pixel tile, chunk) whose biased exponent is kept in a [tiles][channels / 16] int32 table.  The consumer
 * stages the pieces as they are (no maximum, no split, no scale exchange in its K loop) and only moves the two halo rows that came from the neighbouring
 * tiles onto the smallest of the three scales (a power of two: exact).  Stored values are fp32 numbers of 22 significant bits (|v - stored| <= 2^-23 |v|,
 * what the consumer's own split would have made of them); THE BOUND above holds with M_c = the largest value of the three tiles a tile's rows come from.
 * capf_op_conv_f32h2_planes: one such conv.  exps_in != NULL: conv->x holds planes (Cin % 16 == 0) and exps_in their table; exps_out != NULL: conv->y
 * is written as planes (Cout % 16 == 0, no residual) and exps_out receives capf_op_conv_f32h2_tiles(B, H, W, ..) x Cout / 16 exponents.  Producer and
 * consumer must have the same B, H, W (same tiles).  capf_forward uses the pair for every BasicBlock whose two convs run this tile
 * ONLY under CAPF_PLAN_H2_PLANES: the split moves, it does not disappear, and the forward measured 3 % slower with it (EXPERIMENTS R6.5).                                                                               */
int capf_op_conv_f32h2_tiles(int B, int H, int W, int* tile_pixels);     /* tiles; *tile_pixels (may be NULL): output pixels per tile, flat pixel p -> tile p / that */
int capf_op_conv_f32h2_planes(void* stream, const capf_conv_desc* conv, const int32_t* exps_in, int32_t* exps_out);

/* The two-fp16-piece arithmetic for every OTHER fp32 conv / linear (csrc/igemm_f32h2.hip; what capf_forward runs from batch 5 for the 1x1 and
 * stride-2 convs of the fuse / transition layers, pose_hrnet.py:225-303, lone convs, and -- in inference plans -- the lifter's nn.Linear layers,
 * pose_dformer.py:15-59): the activation tile is staged as fp32 and split by the wave that consumes it (one power-of-two scale per wave,
 * 32 rows and 32-deep K chunk), the weights are split at pack time (one scale per output channel); same ranges and the same accuracy
 * statement as capf_op_conv_f32h2_group.  capf_op_pack_f32h2_gemm writes capf_op_f32h2_gemm_pack_elems(N, K) floats: the fp32 pack's
 * [N][Kpad] geometry (Kpad = K rounded up to 32; conv: ks >= 1, w OIHW, K = ks * ks * Cin, BatchNorm folded, bias written if not NULL;
 * linear: ks = 0, Cin ignored, w [N][K], bias untouched) holding [piece 0: 32 fp16 | piece 1: 32 fp16] per 32-deep chunk, then the N
 * fp32 inverse channel scales.  capf_op_conv_f32h2g / capf_op_linear_f32h2g: capf_op_conv / capf_op_linear on that pack (Cin % 4 == 0,
 * Cout % 4 == 0, ks <= 5; K % 32 == 0, N % 4 == 0); capf_op_conv_f32h2g_group: up to 8 such convs in one grid.  A conv's input may exceed
 * 2 GiB (offsets count from the tile's first input pixel); outputs and rows-mode operands below 4e9 elements / bytes as for capf_op_conv.
 * The training step uses the same kernel for y = x W^T and dX = dY W (packs of W and W^T rebuilt from the parameters at the start of every
 * capf_forward_train, DropPath's per-row branch scale in the epilogue) and its sibling wgrad_tn_h2_kernel for dW = dY^T x (both operands
 * split in the kernel): capf_train_h2_matrices().                                                                                       */
int64_t capf_op_f32h2_gemm_pack_elems(int N, int K);
int capf_op_pack_f32h2_gemm(void* stream, const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                            float eps, float* w_packed, float* bias, int N, int Cin, int ks, int K);
int capf_op_conv_f32h2g(void* stream, const float* x_nhwc, const float* w_packed, const float* bias, const float* residual, float* y,
                        int B, int H, int W, int Cin, int Cout, int ks, int stride, int act);
int capf_op_conv_f32h2g_group(void* stream, int n, const capf_conv_desc* convs);
int capf_op_linear_f32h2g(void* stream, const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                          int M, int N, int K, int act);
/* ... with LayerNorm(x rows over their K <= 256 columns; gamma, beta [K], eps) folded in front of the product -- Block.norm1 -> attn.qkv and
 * norm2 -> mlp.fc1 of the res blocks, norm2 -> mlp.fc1 of the DeformableBlocks (pose_dformer.py:62-79, 137-138): statistics in fp32 per row
 * (two passes), x_hat = ((x - mean) * rstd) * gamma + beta in fp32, and THAT value is split into the two fp16 pieces.                  */
/* The training step's weight gradient as an operator (train.py:195, loss.backward() through an nn.Linear): dW[N, K] = dY[M, N]^T X[M, K] and
 * db[N] = column sums of dY, written to dw_db[N K + N], straight from the row-major operands (csrc/train_kernels.hip).  two_piece = 1: BOTH
 * operands as two block-scaled fp16 pieces (wgrad_tn_h2_kernel: N, K multiples of 128; one scale per wave, operand and 32-row chunk that only
 * ever goes down along M -- the bound above with M_c = the largest value of the operand block so far); 0: the fp32 matrix pipe
 * (wgrad_tn_kernel: N, K multiples of 4).  One row slice, i.e. the summation order of a training step whose slab buffer holds one slab. */
int capf_op_wgrad(void* stream, const float* dY, const float* X, int M, int N, int K, float* dw_db, int two_piece);
int capf_op_linear_ln_f32h2g(void* stream, const float* x, const float* ln_gamma, const float* ln_beta, float eps, const float* w_packed,
                             const float* bias, const float* residual, float* y, int M, int N, int K, int act);

/* Up to 8 independent bf16 convs in one grid (what capf_forward issues per dependency level of the HRNet branches of a bf16 model):
 * capf_conv_desc with bf16 x / w_packed / residual / y.  w_row_halo[i] (optional array, entries may be NULL): the same weights
 * from capf_op_pack_conv_bf16_rh; launches of >= 2048 tiles then run the row-halo tiles for those problems, exactly as the
 * engine does.  *variant (optional) = the device kernel chosen (see capf_forward_profile_variants).                           */
int capf_op_conv_bf16_group(void* stream, int n, const capf_conv_desc* convs, const void* const* w_row_halo, int32_t* variant);

/* nn.Linear on the bf16 MFMA path (the lifter's qkv / proj / fc1 / fc2 under compute_dtype = CAPF_BF16, pose_dformer.py:15-59):
 * x bf16 [M,K], w bf16 [N,K] (K % 64 == 0), bias fp32; gelu_bf16_out = 0: y fp32 [M,N] = x w^T + bias (+ fp32 residual);
 * gelu_bf16_out = 1: y bf16 [M,N] = GELU(x w^T + bias) (exact erf).  fp32 accumulation in both.                        */
int capf_op_linear_bf16(void* stream, const void* x_bf16, const void* w_bf16, const float* bias, const float* residual, void* y,
                        int M, int N, int K, int gelu_bf16_out);

/* ---- the steps on either side of the path ---------------------------------
 * capf_preprocess: data_prefetcher.preload (ContextPose/mvn/datasets/utils.py:33-82) as one launch pair:
 *   images_bgr uint8 [B,H,W,3] -> images_out fp32 RGB NHWC ((u/255 - mean) / std; std == NULL: CPN, mean only);
 *   gt_in [B,1,17,3] -> gt_out root-relative (:52-53); k2d / kcrop [B,17,2] copied or mirrored.
 *   mode 0: as is.  mode 1: train-time horizontal flip of the whole batch incl. left/right joint swap and
 *   `192 - x - 1` (:55-65).  mode 2: flip-test stacking (:67-80): images_out [2,B,H,W,3], k2d_out / kcrop_out
 *   [2,B,17,2] hold the original followed by the mirrored sample, so ONE capf_forward of batch 2B serves both.
 * capf_fliptest_fuse: train.py:177-180 — pred2 [2,B,1,17,3] -> out [B,1,17,3] = mean(pred, un-mirrored pred). */
int capf_preprocess(void* stream, const uint8_t* images_bgr, int batch, int height, int width, const float mean[3],
                    const float* std3, int mode, float* images_out, const float* gt_in, float* gt_out,
                    const float* k2d_in, float* k2d_out, const float* kcrop_in, float* kcrop_out);
int capf_fliptest_fuse(void* stream, const float* pred2, int batch, float* out);

/* ---- N2, second half: evaluation metrics over gathered predictions (train.py:381-436) -----------------
 * capf_pose_errors: per pose i of pred / gt [n, joints, 3] (joints <= 32), err[i] = {e_MPJPE, e_P_MPJPE, e_N_MPJPE,
 *   e_velocity}, each the mean over joints:  MPJPE loss.py:16-22;  P_MPJPE loss.py:25-68 (similarity Procrustes; the
 *   host numpy SVD of :48-60 is replaced by Horn's quaternion closed form, fp64 Jacobi per thread);  N_MPJPE
 *   loss.py:71-84;  velocity = |(pred_i - pred_prev) - (gt_i - gt_prev)| with prev = prev[i] (the previous pose of the
 *   same evaluated subset, i.e. np.diff over the masked rows, loss.py:87-101); prev == NULL means i-1; prev[i] < 0: 0.
 * capf_segment_sums: per-action aggregation of evaluate_using_pred (datasets/human36m.py:358-417): sums[a] =
 *   {sum e_MPJPE, sum e_P_MPJPE, sum e_N_MPJPE, sum e_velocity} over poses with segment[i] == a (fp64, fixed order:
 *   deterministic), counts[a] = {poses, poses with a predecessor}.  segment == NULL with n_segments == 1: all poses.
 *   The reference's numbers are then MPJPE_a = sums[a][0]/counts[a][0], MPJVE_a = sums[a][3]/counts[a][1], ...
 * capf_keypoints_loss: KeypointsMSELoss (mode 0) / KeypointsMSESmoothLoss (1, `threshold`) / KeypointsMAELoss (2),
 *   loss.py:104-137: pred / gt [rows, dim], validity [rows] (the reference's [...,1] mask); writes the scalar loss and,
 *   if dpred != NULL, dloss/dpred.                                                                              */
int capf_pose_errors(void* stream, const float* pred, const float* gt, int n, int joints, const int32_t* prev, float* err);
int capf_segment_sums(void* stream, const float* err, const int32_t* segment, const int32_t* prev, int n, int n_segments,
                      double* sums, int32_t* counts);
int capf_keypoints_loss(void* stream, int mode, const float* pred, const float* gt, const float* validity, int rows, int dim,
                        float threshold, float* loss, float* dpred);

/* ---- N3: the per-frame affine crop in front of the prefetcher ------------------------
 * capf_affine_from_center_scale: get_affine_transform(center, scale, 0, (out_w, out_h)) of
 *   mvn/utils/img.py:16-48 (rot 0, shift 0): the 2x3 row-major double matrix cv2.getAffineTransform returns
 *   for the three float32 point pairs.  Host code, no GPU needed.
 * capf_warp_affine: crop_image (img.py:51-69, called from Human36M.__getitem__ human36m.py:298-300) for a whole
 *   batch: out[b] = cv2.warpAffine(frame_b, m_b, (out_w, out_h), flags=INTER_LINEAR), constant border 0, 8-bit
 *   3-channel.  frames: DEVICE array of `batch` device pointers; dims: device int32 [batch][3] = rows, cols,
 *   row pitch in bytes; m: device double [batch][6] FORWARD matrices; out: uint8 [batch, out_h, out_w, 3].
 *   OpenCV's fixed-point arithmetic (10-bit coordinates, 5-bit fractions, 15-bit weights) is restated from its
 *   published algorithm -- OpenCV is absent from the build container, so this row's parity is unpinned. */
/* capf_jpeg_info / capf_jpeg_decode: the image decode in FRONT of that crop -- cv2.imread(path, IMREAD_COLOR | IMREAD_IGNORE_ORIENTATION) of
 *   Human36M.__getitem__ (human36m.py:292-295) for a baseline JPEG held in host memory: uint8 BGR [height][width][3] on the device, ready
 *   for capf_warp_affine.  The host walks the entropy-coded segment (Huffman decode, DC prediction, restart markers); dequantisation +
 *   inverse DCT, chroma upsampling and YCbCr -> BGR run on the GPU.  The arithmetic is libjpeg's default decode path restated -- "islow"
 *   integer IDCT (jidctint.c), "fancy" triangle upsampling (jdsample.c), 16-bit fixed point colour conversion (jdcolor.c) -- which is what
 *   cv2.imread and Pillow run: the result is bit-exact against Pillow's libjpeg-turbo decode (tests/test_jpeg.py; OpenCV itself is not
 *   in the image).  Baseline / extended sequential Huffman, 8 bit, grey or YCbCr in one interleaved scan, 4:4:4 / 4:2:2 / 4:2:0, restart
 *   intervals; anything else (progressive, arithmetic, CMYK, other sampling) returns CAPF_ERR_UNSUPPORTED and the caller keeps its host
 *   decoder.  capf_jpeg_info: geometry + the scratch bytes capf_jpeg_decode needs (device memory, 16-byte aligned).  capf_jpeg_decode
 *   enqueues on `stream` and waits for the coefficient upload (it reuses a per-thread host staging buffer): a loader-side call, never
 *   part of capf_forward.  capf_jpeg_coefficients: the host half alone -- quantised coefficients in natural order, component after
 *   component, blocks [rows][cols][64] over the MCU-padded image (no GPU needed).                                                      */
int capf_jpeg_info(const uint8_t* data, size_t n_bytes, int32_t* width, int32_t* height, int32_t* components, int32_t* h_samp,
                   int32_t* v_samp, size_t* scratch_bytes);
int capf_jpeg_coefficients(const uint8_t* data, size_t n_bytes, int16_t* coef, size_t coef_elems);
int capf_jpeg_decode(void* stream, const uint8_t* data, size_t n_bytes, uint8_t* out_bgr, size_t out_pitch_bytes, void* scratch,
                     size_t scratch_bytes);
int capf_affine_from_center_scale(const double center[2], const double scale[2], int out_w, int out_h, double m[6]);
int capf_warp_affine(void* stream, const uint8_t* const* frames, const int32_t* dims, const double* m, int batch,
                     int out_h, int out_w, uint8_t* out);

/* ---- measurement aids (bench.py roofline line; no reference counterpart) -------------------------
 * capf_op_info: op `index` in launch order: its plan name, the kernel (template instantiation) it
 *   launches at `batch`, and its algorithmic FLOPs at `batch`.
 * capf_forward_profile: capf_forward with a hipEvent pair recorded on `stream` around every launch;
 *   synchronises the stream and writes the elapsed milliseconds per op into op_ms[0..n_ops).  Every op is
 *   launched on its own, in program order (capf_set_lanes is ignored).
 * capf_forward_profile_launches: the same for the PRODUCT schedule (capf_set_lanes 0 or 2 / 3; side streams are
 *   not used: mode 3 is timed as its one-chain form): one event pair per launch.  op_leader[i] = first op of the launch op i rode in (-1: not
 *   launched); op_ms[i] = elapsed ms of that launch if i is a leader, else 0.  A grouped launch therefore
 *   shows up as one time for several ops.                                                              */
int capf_num_ops(const capf_handle* h);
int capf_op_info(const capf_handle* h, int index, int batch, const char** name, const char** kernel,
                 double* flops);
/* FLOPs the matrix pipe is asked to EXECUTE for op `index` at `batch` (2 x issued MACs; K padding included, tile-edge padding
 * not).  Differs from capf_op_info's ALGORITHMIC count for the Winograd kernels (F(4,3): 1/2, F(2,3): 2/3 of the direct
 * convolution's multiplies): bench.py reports both, so that `roofline.frac` (algorithmic / peak) is never read
 * as matrix-pipe utilisation. */
int capf_op_executed_flops(const capf_handle* h, int index, int batch, double* flops);
/* Algorithmic (compulsory) HBM bytes of op `index` at `batch`: each operand read once, the result written once
 * (bf16 tensors 2 B/element).  bench.py's HBM-side roofline divides these by the measured launch durations. */
int capf_op_bytes(const capf_handle* h, int index, int batch, double* bytes);
/* capf_op_schedule: where op `index` sits in the launch schedule: its fork/join region (-1 outside / control op),
 *   its dependency level inside the region (capf_set_lanes mode 2 issues a region level by level), its lane
 *   (mode 1: side stream), and the workspace buffer ids it reads (5 slots) and writes (6 slots), -1 = unused,
 *   -2 = the external image.  Lets host-side tests check that the schedule is a valid topological order. */
int capf_op_schedule(const capf_handle* h, int index, int32_t* region, int32_t* level, int32_t* lane,
                     int32_t reads[5], int32_t writes[6]);
/* ---- layer-wise ("teacher-forced") parity aids ---------------------------------------------------------------
 * A deep bf16 network is chaotic at the rounding level: two correct implementations that differ only in fp32 summation
 * order drift apart to the full bf16 noise floor after ~50 layers, so an end-to-end comparison cannot be tighter than that
 * floor.  What CAN be tight is one op at a time on the engine's OWN inputs:
 *   capf_forward_prefix : the product schedule (grouped launches and all) for the first n_ops ops only; k2d / kcrop_inout /
 *                         out may be NULL while n_ops stays inside the backbone.
 *   capf_op_describe    : geometry, operand types and parameter indices of op `index`, and `checkpoint` = the prefix
 *                         length after which every tensor the op touched is still intact in the workspace (buffers are
 *                         reused along the plan; a fork/join region keeps all of its buffers until its join).
 *   capf_op_tensor      : device pointer of one operand of op `index` after such a prefix (slot 0..3 inputs, 4 residual,
 *                         5 output; the external image for the stem's input).
 * tests/test_gpu_layerwise.py recomputes every backbone op on the CPU from the engine's inputs and compares outputs.   */
typedef struct capf_op_desc {
    int32_t kind;              /* 0 conv / linear on the MFMA kernels, 1 fuse-sum, 2 max-pool 3x3 s2, 3 bilinear resize, 4 LayerNorm over rows,
                                * 5 multi-head self-attention over the tokens of a group, -1 other */
    int32_t backbone;          /* 1: the op belongs to the backbone plan */
    int32_t conv;              /* kind 0: 1 = convolution (NHWC), 0 = rows-mode linear */
    int32_t Cin, H, W, Cout, Ho, Wo, ks, stride, pad, act;   /* act: 0 none, 1 ReLU, 2 GELU */
    int32_t in_dtype, out_dtype;                             /* capf_tensor convention: 0 fp32, 2 bf16 */
    int32_t mfma_bf16;         /* operands are rounded to bf16 for the matrix pipe */
    int32_t n_in, shift[4], relu;                            /* fuse-sum: inputs, log2 nearest-upsample factors, ReLU */
    int32_t p_weight, p_bn_weight;                           /* capf_param_info indices of <conv>.weight and <bn>.weight; -1 */
    int32_t has_residual;
    int32_t checkpoint;
    /* rows-mode ops of the lifter (kind 0 with conv == 0, kinds 4 and 5; pose_dformer.py:15-79): row m of an operand lives at element
     * (m / G) * S1 + (m % G) * S2 + off of its buffer, maps[k] = {G, S1, S2, off} for k = 0 input rows, 1 output rows, 2 residual rows
     * (kind 4: the rows ADDED to the input before normalising).  kind 0: Cin = K, Cout = N, p_weight / p_bias [N,K] / [N]; an
     * in-place update (proj, fc2: output rows == residual rows) reads its residual from the state BEFORE the op, i.e. after
     * capf_forward_prefix(index), its result after capf_forward_prefix(index + 1).  kind 4: Cin = width, p_ln_weight / p_ln_bias, eps.
     * kind 5: input rows [groups * tokens, 3 * heads * head_dim] ordered (q | k | v) x heads x head_dim (pose_dformer.py:49), output
     * rows [groups * tokens, heads * head_dim]; groups per frame / tokens / heads / head_dim in attn[4].                            */
    int32_t rows_per_frame;
    int32_t p_bias, p_ln_weight, p_ln_bias;
    float eps;
    int32_t attn[4];
    int64_t maps[3][4];
    /* kind 0, conv: up_H > 0 = the tensor in slot 1, [B, up_H, up_W, Cout] in the output's dtype, is resized to Ho x Wo (bilinear, align_corners =
     * True) and ADDED BEHIND the activation, in fp32, before the one storage rounding: CPN's `lateral + upsampled path` (globalNet.py:66) inside the
     * lateral conv's launch (compute_dtype = bf16; CAPF_PLAN_NO_UPADD: a kind-3 op behind the conv instead).  0 otherwise.                          */
    int32_t up_H, up_W;
} capf_op_desc;
int capf_forward_prefix(capf_handle* h, void* stream, const float* images_nhwc, const float* k2d, float* kcrop_inout,
                        int batch, float* out, int n_ops);
int capf_op_describe(const capf_handle* h, int index, capf_op_desc* desc);          /* writes sizeof(capf_op_desc) of THIS header's revision */
/* ... for callers that cannot rule out a header / library mismatch: never writes more than desc_bytes (what the caller's struct holds);
 * fields beyond that are dropped, a shorter library struct leaves the caller's tail zeroed.                                            */
int capf_op_describe_sized(const capf_handle* h, int index, void* desc, size_t desc_bytes);
int capf_op_tensor(const capf_handle* h, int index, int slot, const void** dev_ptr);
/* Does op `index` exchange a PLANES tensor (capf_op_conv_f32h2_planes) with its BasicBlock partner at this batch?  Returns 0: no (plain fp32 on both
 * sides), 1: its OUTPUT (slot 5) holds planes, 2: its INPUT (slot 0) does; *exps: the device pointer of the int32 [tiles][channels / 16] exponent table
 * (valid after a forward of that batch), *tile_pixels: output pixels per tile.  The layer-wise tests decode the planes with these.                 */
int capf_op_h2_planes(const capf_handle* h, int index, int batch, const void** exps, int32_t* tile_pixels);

int capf_forward_profile(capf_handle* h, void* stream, const float* images_nhwc, const float* k2d,
                         float* kcrop_inout, int batch, float* out, float* op_ms, int n_ops);
int capf_forward_profile_launches(capf_handle* h, void* stream, const float* images_nhwc, const float* k2d,
                                  float* kcrop_inout, int batch, float* out, float* op_ms, int32_t* op_leader,
                                  int n_ops);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* CAPF_H */
