"""CPU oracle for the Context-Aware PoseFormer hot path (image + 17 2D keypoints -> 17x3 joints).

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file, and only as the checker / the timed CPU baseline.  The
product path (contextaware-poseformer_amd/) never imports it and has no CPU fallback.

What it is: a from-the-math restatement, in plain functional PyTorch-CPU fp32 (floating-point
path -> a torch fp32 reference is the appropriate oracle; the integer part, the bilinear corner
indices, is additionally restated in numpy float32 in `bilinear_corners`), of

    CA_PF.forward                      ContextPose/mvn/models/conpose.py:30-42
    PoseHighResolutionNet.forward      ContextPose/mvn/models/pose_hrnet.py:464-501
    CPN.forward (+ResNet/global/refine) ContextPose/mvn/models/networks/network.py:16-22
    PoseTransformer.forward            ContextPose/mvn/models/pose_dformer.py:210-241
    MPJPE.forward                      ContextPose/mvn/models/loss.py:16-22

It consumes a flat {state_dict_name: tensor} mapping with exactly the reference's names (SURVEY.md
§8b / Appendix B), so the same synthetic checkpoint feeds the reference (in the build container),
this oracle and the HIP path.

Parity pin: the reference has no tests / golden vectors of its own, so this oracle is
pinned against outputs of the reference itself, imported on CPU in the build container by
oracle/make_goldens.py, which writes tests/golden/*.npz; tests/test_oracle_golden.py replays them
(runs on CPU, no reference needed).  Third-party arithmetic (ATen conv / grid_sampler / layer_norm,
timm DropPath, einops.rearrange) is restated from its documented semantics; the bilinear rule follows
ATen/native/GridSampler.h:27-36,58-60,143-171 of the installed torch (the reference pins 1.11.0).

`emulate_bf16=True` (ca_pf_forward / hrnet_forward / cpn_forward / lifter_forward): the SAME algorithm with the storage
roundings of the engine's compute_dtype = bf16 mode applied at the points where the engine stores bf16:
conv weights after the BatchNorm fold, the image on its way into the stem, every conv / fuse-sum / max-pool / resize
output, the LayerNorm rows / attention outputs / GELU hidden rows that feed the lifter's qkv / proj / fc1 / fc2
projections and those projections' weights; accumulation, bias, residual adds, LayerNorm statistics, softmax, both
samplers and the token stream stay fp32, as in the engine.  It is the checker for BASELINE configs[2] / [4]: against IT
the HIP path may differ only by fp32 summation order (and the rare bf16 rounding flip that causes), so the bound is tight,
while the distance of either from the fp32 reference is the (reported) rounding budget of the bf16 mode.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, used everywhere in the reference backbones


# ----------------------------------------------------------------------------------------------
# small building blocks
# ----------------------------------------------------------------------------------------------
def _conv(P, name, x, stride=1, pad=0):
    return F.conv2d(x, P[name + ".weight"], None, stride, pad)


def _bn(P, name, x):
    # eval-mode BatchNorm2d: y = (x - mean) / sqrt(var + eps) * gamma + beta
    return F.batch_norm(x, P[name + ".running_mean"], P[name + ".running_var"],
                        P[name + ".weight"], P[name + ".bias"], False, 0.0, BN_EPS)


def bf16_round(x):
    """fp32 -> bf16 (round to nearest even, what v_cvt_pk_bf16_f32 does) -> fp32."""
    return x.to(torch.bfloat16).to(torch.float32)


class Numerics:
    """Where values are rounded to bf16 storage.  FP32: nowhere (the reference's arithmetic).  BF16: the engine's
    compute_dtype = bf16 mode (see the module docstring)."""

    def __init__(self, bf16=False):
        self.bf16 = bf16

    def r(self, x):
        return bf16_round(x) if self.bf16 else x


FP32, BF16 = Numerics(False), Numerics(True)


class _NoRound(Numerics):
    """bf16 operands (folded weights), result left in fp32: a conv whose output meets another term before it is stored."""

    def __init__(self):
        super().__init__(True)

    def r(self, x):
        return x


_NOROUND = _NoRound()


class _StreamFP32(Numerics):
    """The A/B VERDICT r4 / r5 asked about ("bf16 only as MFMA operands, residual stream fp32"): every backbone
    activation stays fp32 in memory and is rounded to bf16 where a conv reads it as an MFMA operand; weights as in BF16.  Backbone only
    (ca_pf_forward(emulate_bf16="stream_fp32") keeps the lifter on the engine's bf16 placement)."""

    def __init__(self):
        super().__init__(True)
        self.stream = True

    def r(self, x):
        return x


BF16_STREAM_FP32 = _StreamFP32()


def _cbr(P, conv, bn, x, stride=1, pad=0, relu=True, res=None, nm=FP32):
    """conv (bias=False) + eval BatchNorm (+ residual) (+ ReLU): ONE launch of the engine (csrc/plan.cpp conv_bn), so in
    bf16 mode ONE rounding at the end.  fp32 mode is literally relu(bn(conv(x)) + res)."""
    if not nm.bf16:
        y = _bn(P, bn, _conv(P, conv, x, stride, pad))
    else:
        # the engine folds BatchNorm into the weights in fp32 and THEN rounds them (elementwise.hip pack_conv_kernel):
        #   sc = gamma / sqrt(var + eps);  w' = bf16(w * sc);  bias = beta - mean * sc  (fp32)
        sc = P[bn + ".weight"] / torch.sqrt(P[bn + ".running_var"] + BN_EPS)
        w = bf16_round(P[conv + ".weight"] * sc.view(-1, 1, 1, 1))
        if getattr(nm, "stream", False):
            x = bf16_round(x)                                 # (fp32 in memory, bf16 as the MFMA operand)
        y = F.conv2d(x, w, P[bn + ".bias"] - P[bn + ".running_mean"] * sc, stride, pad)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    return nm.r(y)


def _linear(P, name, x):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _linear_mm(P, name, x, nm):
    """nn.Linear on the MFMA path: in bf16 mode the weight is a bf16 copy and x must already hold bf16 values; fp32 bias,
    fp32 accumulation, fp32 result."""
    if not nm.bf16:
        return _linear(P, name, x)
    return F.linear(x, bf16_round(P[name + ".weight"]), P.get(name + ".bias"))


def _ln(P, name, x, eps):
    w = P[name + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, P[name + ".bias"], eps)


# ----------------------------------------------------------------------------------------------
# HRNet (pose_hrnet.py)
# ----------------------------------------------------------------------------------------------
def _basic_block(P, pre, x, nm=FP32):
    """pose_hrnet.py:66-95 (never has a downsample inside HRNet stages: in==out, stride 1)."""
    y = _cbr(P, pre + ".conv1", pre + ".bn1", x, 1, 1, nm=nm)
    return _cbr(P, pre + ".conv2", pre + ".bn2", y, 1, 1, res=x, nm=nm)


def _bottleneck(P, pre, x, nm=FP32):
    """pose_hrnet.py:98-136; downsample (1x1 conv + BN) exists iff its keys are present."""
    y = _cbr(P, pre + ".conv1", pre + ".bn1", x, nm=nm)
    y = _cbr(P, pre + ".conv2", pre + ".bn2", y, 1, 1, nm=nm)
    if (pre + ".downsample.0.weight") in P:
        x = _cbr(P, pre + ".downsample.0", pre + ".downsample.1", x, relu=False, nm=nm)
    return _cbr(P, pre + ".conv3", pre + ".bn3", y, res=x, nm=nm)


def _hr_module(P, pre, xs, n_out, nm=FP32):
    """HighResolutionModule.forward, pose_hrnet.py:285-303.

    Returns (fused outputs, branch outputs).  The reference mutates its input list in place
    (:289-290); the caller reproduces the aliasing consequence explicitly (see hrnet_forward).
    """
    nb = len(xs)
    br = []
    for i in range(nb):
        y = xs[i]
        for k in range(4):                                   # NUM_BLOCKS = 4 everywhere (cfg.py:44,53,62)
            y = _basic_block(P, f"{pre}.branches.{i}.{k}", y, nm)
        br.append(y)
    outs = []
    for i in range(n_out):
        acc = None
        for j in range(nb):
            if j == i:
                t = br[j]
            elif j > i:                                      # 1x1 conv + BN + nearest upsample (:238-245)
                fp = f"{pre}.fuse_layers.{i}.{j}"
                t = _cbr(P, fp + ".0", fp + ".1", br[j], relu=False, nm=nm)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:                                            # chain of 3x3 s2 convs (:249-275)
                t = br[j]
                for k in range(i - j):
                    fp = f"{pre}.fuse_layers.{i}.{j}.{k}"
                    t = _cbr(P, fp + ".0", fp + ".1", t, 2, 1, relu=(k != i - j - 1), nm=nm)
            acc = t if acc is None else acc + t              # summation order j = 0,1,2,.. (:294-300)
        outs.append(nm.r(F.relu(acc)))                       # (bf16 mode: fuse_sum_kernel adds in fp32, stores bf16)
    return outs, br


def hrnet_forward(P, x, pre="backbone", nm=FP32):
    """PoseHighResolutionNet.forward, pose_hrnet.py:464-501.  x: [B,3,H,W] -> 4 maps (NCHW)."""
    x = nm.r(x)                                              # (bf16 mode: the stem kernel rounds the image into LDS)
    x = _cbr(P, pre + ".conv1", pre + ".bn1", x, 2, 1, nm=nm)
    x = _cbr(P, pre + ".conv2", pre + ".bn2", x, 2, 1, nm=nm)
    for k in range(4):
        x = _bottleneck(P, f"{pre}.layer1.{k}", x, nm)

    def trans(name, src):
        # _make_transition_layer pose_hrnet.py:377-411: either one 3x3 s1 conv (same branch,
        # channel change) or a 3x3 s2 conv from the last branch (new branch); +BN+ReLU.
        if (name + ".0.weight") in P:
            return _cbr(P, name + ".0", name + ".1", src, 1, 1, nm=nm)
        return _cbr(P, name + ".0.0", name + ".0.1", src, 2, 1, nm=nm)

    xs = [trans(pre + ".transition1.0", x), trans(pre + ".transition1.1", x)]
    ys, _ = _hr_module(P, pre + ".stage2.0", xs, 2, nm)

    xs = [ys[0], ys[1], trans(pre + ".transition2.2", ys[-1])]
    for m in range(4):
        ys, _ = _hr_module(P, f"{pre}.stage3.{m}", xs if m == 0 else ys, 3, nm)

    xs = [ys[0], ys[1], ys[2], trans(pre + ".transition3.3", ys[-1])]
    ys, br0 = _hr_module(P, pre + ".stage4.0", xs, 4, nm)
    ys, _ = _hr_module(P, pre + ".stage4.1", ys, 4, nm)
    ys, _ = _hr_module(P, pre + ".stage4.2", ys, 1, nm)
    # :501 returns [y_list[0], x_list[1], x_list[2], x_list[3]]; x_list was mutated in place by
    # stage4[0] (:289-290), so entries 1..3 are stage4[0]'s *branch* outputs.
    return [ys[0], br0[1], br0[2], br0[3]]


# ----------------------------------------------------------------------------------------------
# CPN-50 (networks/resnet.py, globalNet.py, refineNet.py)
# ----------------------------------------------------------------------------------------------
def _res_bottleneck(P, pre, x, stride, nm=FP32):
    """networks/resnet.py:58-93 (expansion 4, stride on the 3x3)."""
    y = _cbr(P, pre + ".conv1", pre + ".bn1", x, nm=nm)
    y = _cbr(P, pre + ".conv2", pre + ".bn2", y, stride, 1, nm=nm)
    if (pre + ".downsample.0.weight") in P:
        x = _cbr(P, pre + ".downsample.0", pre + ".downsample.1", x, stride, relu=False, nm=nm)
    return _cbr(P, pre + ".conv3", pre + ".bn3", y, res=x, nm=nm)


def cpn_forward(P, x, pre="backbone", out_hw=(64, 48), nm=FP32, taps=None):
    """CPN.forward networks/network.py:16-22 -> 4 maps [B,256,64,48].

    The `predict` heads of globalNet (globalNet.py:71) and refineNet.final_predict are computed and
    discarded / never called by the reference; they have no effect on the outputs and are skipped.
    """
    r = pre + ".resnet"
    x = nm.r(x)
    x = _cbr(P, r + ".conv1", r + ".bn1", x, 2, 3, nm=nm)                    # resnet.py:137-139
    x = F.max_pool2d(x, 3, 2, 1)                                             # :140 (exact on bf16 values)
    feats = []
    for li, (n, s) in enumerate(zip([3, 4, 6, 3], [1, 2, 2, 2])):            # :141-144, resnet50
        for k in range(n):
            x = _res_bottleneck(P, f"{r}.layer{li + 1}.{k}", x, s if k == 0 else 1, nm)
        feats.append(x)
    res_out = feats[::-1]                                                    # [x4,x3,x2,x1] :147

    g = pre + ".global_net"
    fms, u = [], None
    for i in range(4):                                                       # globalNet.py:61-83
        f = _cbr(P, f"{g}.laterals.{i}.0", f"{g}.laterals.{i}.1", res_out[i], nm=nm)
        if i > 0 and not nm.bf16:
            # feature_i = lateral_i + BN(conv1x1(bilinear x2(feature_{i-1})))   (:66-70), in the reference's order
            u = F.interpolate(fms[i - 1], scale_factor=2, mode="bilinear", align_corners=True)
            f = f + _cbr(P, f"{g}.upsamples.{i - 1}.1", f"{g}.upsamples.{i - 1}.2", u, relu=False)
        elif i > 0:
            # bf16 emulation follows the ENGINE's order (csrc/plan.cpp build_cpn): a bias-free 1x1 conv + eval BatchNorm is a per-pixel
            # affine map and commutes with the interpolation, so the engine convolves the LOW-resolution map, stores THAT in bf16,
            # and (round 6) adds its interpolation to the lateral conv's fp32 result inside that conv's epilogue: the lateral is never
            # rounded on its own, the sum is rounded once
            low = _cbr(P, f"{g}.upsamples.{i - 1}.1", f"{g}.upsamples.{i - 1}.2", fms[i - 1], relu=False, nm=nm)
            f = _cbr(P, f"{g}.laterals.{i}.0", f"{g}.laterals.{i}.1", res_out[i], nm=_NOROUND)
            if taps is not None:
                taps.setdefault("cpn_lateral", []).append(f)
                taps.setdefault("cpn_up_low", []).append(low)
            f = nm.r(F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=True) + f)
        fms.append(f)
    if taps is not None:
        taps["cpn_fms"] = fms

    rn = pre + ".refine_net"
    outs = []
    for i in range(4):                                                       # refineNet.py:72-88
        y = fms[i]
        for k in range(3 - i):
            bp = f"{rn}.cascade.{i}.{k}"
            t = _cbr(P, bp + ".conv1", bp + ".bn1", y, nm=nm)
            t = _cbr(P, bp + ".conv2", bp + ".bn2", t, 1, 1, nm=nm)
            d = _cbr(P, bp + ".downsample.0", bp + ".downsample.1", y, relu=False, nm=nm)
            y = _cbr(P, bp + ".conv3", bp + ".bn3", t, res=d, nm=nm)
        outs.append(nm.r(F.interpolate(y, size=out_hw, mode="bilinear", align_corners=True)))
    return outs


# ----------------------------------------------------------------------------------------------
# bilinear sampling (F.grid_sample, bilinear, align_corners=True) — explicit restatement
# ----------------------------------------------------------------------------------------------
def bilinear_corners(grid, H, W, padding):
    """Integer corner indices + fp32 weights of grid_sample(bilinear, align_corners=True).

    grid: float32 array [..., 2] holding (x, y) in normalised [-1, 1] coordinates.
    padding: 'zeros' (pose_dformer.py:217 default) or 'border' (:128).
    Returns dict(ix0, iy0 int32; wx1, wy1 float32 = fractional parts; valid masks for the four
    corners in order nw, ne, sw, se).  Follows ATen GridSampler.h: unnormalise ((g+1)/2*(size-1)),
    border mode clips the *coordinate* to [0, size-1] before floor; zeros mode keeps the coordinate
    and drops out-of-range corners.
    """
    g = np.asarray(grid, dtype=np.float32)
    x = (g[..., 0] + np.float32(1)) / np.float32(2) * np.float32(W - 1)
    y = (g[..., 1] + np.float32(1)) / np.float32(2) * np.float32(H - 1)
    if padding == "border":
        x = np.minimum(np.maximum(x, np.float32(0)), np.float32(W - 1))
        y = np.minimum(np.maximum(y, np.float32(0)), np.float32(H - 1))
    x0f, y0f = np.floor(x), np.floor(y)
    ix0, iy0 = x0f.astype(np.int64), y0f.astype(np.int64)
    wx1 = (x - x0f).astype(np.float32)
    wy1 = (y - y0f).astype(np.float32)
    vx0, vx1 = (ix0 >= 0) & (ix0 < W), (ix0 + 1 >= 0) & (ix0 + 1 < W)
    vy0, vy1 = (iy0 >= 0) & (iy0 < H), (iy0 + 1 >= 0) & (iy0 + 1 < H)
    return dict(ix0=ix0.astype(np.int32), iy0=iy0.astype(np.int32), wx1=wx1, wy1=wy1,
                valid=np.stack([vy0 & vx0, vy0 & vx1, vy1 & vx0, vy1 & vx1], -1))


def grid_sample_explicit(feat, grid, padding):
    """Same result as F.grid_sample(feat, grid, 'bilinear', padding, align_corners=True) built
    from `bilinear_corners` (gather + lerp); feat [B,C,H,W], grid [B,h,w,2] -> [B,C,h,w]."""
    B, C, H, W = feat.shape
    c = bilinear_corners(grid.numpy(), H, W, padding)
    ix0 = torch.from_numpy(c["ix0"].astype(np.int64))
    iy0 = torch.from_numpy(c["iy0"].astype(np.int64))
    wx1, wy1 = torch.from_numpy(c["wx1"]), torch.from_numpy(c["wy1"])
    valid = torch.from_numpy(c["valid"])
    wts = [(1 - wx1) * (1 - wy1), wx1 * (1 - wy1), (1 - wx1) * wy1, wx1 * wy1]   # nw ne sw se
    offs = [(0, 0), (1, 0), (0, 1), (1, 1)]
    flat = feat.reshape(B, C, H * W)
    out = torch.zeros(B, C, *grid.shape[1:3])
    for k in range(4):
        xx = (ix0 + offs[k][0]).clamp(0, W - 1)
        yy = (iy0 + offs[k][1]).clamp(0, H - 1)
        idx = (yy * W + xx).reshape(B, 1, -1).expand(B, C, -1)
        v = torch.gather(flat, 2, idx).reshape(B, C, *grid.shape[1:3])
        out = out + v * (wts[k] * valid[..., k]).unsqueeze(1)
    return out


# ----------------------------------------------------------------------------------------------
# lifting transformer (pose_dformer.py)
# ----------------------------------------------------------------------------------------------
def _mlp(P, pre, x, nm=FP32):
    """Mlp.forward pose_dformer.py:24-31 (exact erf GELU, dropout p=0).  bf16 mode: x holds bf16 rows (the LayerNorm kernel
    wrote them), fc1's epilogue stores GELU(.) as bf16, fc2 accumulates in fp32."""
    return _linear_mm(P, pre + ".fc2", nm.r(F.gelu(_linear_mm(P, pre + ".fc1", x, nm))), nm)


def _attn_block(P, pre, x, heads, keep=None, nm=FP32):
    """Block.forward pose_dformer.py:76-79 with Attention.forward :46-59 inlined; LN eps 1e-6 (:166).
    keep: optional (mask1, mask2) per-sample DropPath multipliers (training parity only).
    bf16 mode: LayerNorm rows and the attention output are stored bf16 (operands of qkv / proj / fc1), qkv itself, the
    softmax and the residual stream are fp32."""
    B, N, C = x.shape
    d = C // heads
    h = nm.r(_ln(P, pre + ".norm1", x, 1e-6))
    qkv = _linear_mm(P, pre + ".attn.qkv", h, nm).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(dim=-1)
    h = _linear_mm(P, pre + ".attn.proj", nm.r((a @ v).transpose(1, 2).reshape(B, N, C)), nm)
    x = x + (h if keep is None else h * keep[0])
    h = _mlp(P, pre + ".mlp", nm.r(_ln(P, pre + ".norm2", x, 1e-6)), nm)
    return x + (h if keep is None else h * keep[1])


def grid_sample_in_cells(feat, grid, ix0, iy0):
    """F.grid_sample(feat, grid, 'bilinear', 'border', align_corners=True) with the bilinear CELL of every sample given
    (ix0, iy0 = its NW corner, integer tensors shaped like grid[..., 0]) instead of derived by floor(): differentiable w.r.t.
    `grid` and `feat`, and continuous across cell boundaries for a fixed choice of cells.  grid_sample's derivative w.r.t. the
    position is one-sided at a boundary; two evaluations whose positions differ by roundoff may take different sides there.
    A gradient test hands the cells the implementation under test used (its `cidx` taps) to this function, so that both sides
    differentiate the same branch of the same piecewise-bilinear function (ATen: grid_sampler_compute_source_index with
    clip_coordinates_set_grad -- no gradient through a clipped coordinate -- and within-bounds corner masks)."""
    B, C, H, W = feat.shape
    def unnorm(g, size):                                     # align_corners=True; border: clip to [0, size-1], zero grad outside
        return (((g + 1) / 2) * (size - 1)).clamp(0, size - 1)
    ix, iy = unnorm(grid[..., 0], W), unnorm(grid[..., 1], H)
    fx, fy = ix - ix0.to(ix.dtype), iy - iy0.to(iy.dtype)
    flat = feat.reshape(B, C, H * W)
    out = 0
    for dx, dy, w in ((0, 0, (1 - fx) * (1 - fy)), (1, 0, fx * (1 - fy)), (0, 1, (1 - fx) * fy), (1, 1, fx * fy)):
        xx, yy = ix0 + dx, iy0 + dy
        inside = ((xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)).to(feat.dtype)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(B, 1, -1).expand(B, C, -1)
        v = torch.gather(flat, 2, idx).reshape(B, C, *grid.shape[1:3])
        out = out + v * (w * inside).unsqueeze(1)
    return out


def _deformable_block(P, pre, x, ref, feats, heads=4, samples=4, explicit=False, keep=None, nm=FP32, cells=None):
    """DeformableBlock.forward pose_dformer.py:115-141; LN eps 1e-5 (default nn.LayerNorm, :84).
    bf16 mode: the attention half (query LayerNorm, logits / offsets, sampling of the bf16-valued maps, embed_proj) is fp32
    in the engine (ctx_attn_kernel); only the MLP half runs bf16 operands."""
    x0, xr = x[:, :1], x[:, 1:]
    b, l, p, c = xr.shape
    q = _ln(P, pre + ".norm1", xr + x0, 1e-5)
    w = _linear(P, pre + ".attention_weights", q).view(b, l, p, heads, samples)
    w = F.softmax(w, dim=-1).unsqueeze(-1)
    off = _linear(P, pre + ".sampling_offsets", q).reshape(b, l, p, heads * samples, 2).tanh()
    pos = off + ref.view(b, 1, p, 1, -1)
    sampled = []
    for idx, f in enumerate(feats):
        if cells is not None:                                               # [b, p, l, hs, 2] NW corners (the engine's cidx tap)
            s = grid_sample_in_cells(f, pos[:, idx], cells[:, :, idx, :, 0], cells[:, :, idx, :, 1])
        elif explicit:
            s = grid_sample_explicit(f, pos[:, idx], "border")
        else:
            s = F.grid_sample(f, pos[:, idx], mode="bilinear", padding_mode="border", align_corners=True)
        s = s.permute(0, 2, 3, 1)                                           # b, p, heads*samples, C_l
        sampled.append(_linear(P, f"{pre}.embed_proj.{idx}", s))
    s = torch.stack(sampled, dim=1)                                         # b, l, p, hs, c/heads
    s = (w * s.view(b, l, p, heads, samples, -1)).sum(dim=-2).view(b, l, p, -1)
    xr = xr + (s if keep is None else s * keep[0])
    h = _mlp(P, pre + ".mlp", nm.r(_ln(P, pre + ".norm2", xr, 1e-5)), nm)
    xr = xr + (h if keep is None else h * keep[1])
    return torch.cat([x0, xr], dim=1), pos


def split_drop_masks(masks, B, J=17, levels=4):
    """The flat DropPath multiplier buffer of the C ABI (include/capf.h, capf_forward_train:
    ctx[i]{m1[B],m2[B]} | res[i]{m1[B*J],m2[B*J]} | joint[i]{m1[B],m2[B]}) -> per-block (mask1, mask2) pairs shaped to
    broadcast over the tensors timm's DropPath sees (pose_dformer.py:71,101: one draw per element of dim 0, which is
    the frame for context / joint blocks and the (frame, joint) pair for the level blocks, :231-234)."""
    out = {"ctx": [], "res": [], "joint": []}
    off = 0
    for group, per, shape in (("ctx", B, (B, 1, 1, 1)), ("res", B * J, (B * J, 1, 1)), ("joint", B, (B, 1, 1))):
        for _ in range(levels):
            m1 = masks[off:off + per].view(shape); off += per
            m2 = masks[off:off + per].view(shape); off += per
            out[group].append((m1, m2))
    assert off == masks.numel()
    return out


def lifter_forward(P, k2d, ref, feats, pre="volume_net", levels=4, explicit=False, taps=None,
                   context_blocks=True, drop_masks=None, emulate_bf16=False, depth=None, cells=None):
    """PoseTransformer.forward pose_dformer.py:210-241.

    k2d [B,17,2], ref [B,17,2] (already normalised), feats: 4 NCHW maps -> [B,1,17,3].
    taps: optional dict that receives intermediates for stage-level parity tests.
    cells: optional list (one per context block) of int64 [B,17,levels,16,2] NW corners: the deformable samplers evaluate
    grid_sample_in_cells with them (gradient tests: same bilinear cells as the implementation under test).
    """
    b, p, _ = k2d.shape
    nm = BF16 if emulate_bf16 else FP32
    keep = split_drop_masks(drop_masks, b, p, levels) if drop_masks is not None else None
    x = _linear(P, pre + ".coord_embed", k2d)                               # :214
    toks = []
    for f in feats:                                                         # :216-218 (padding zeros)
        g = ref.unsqueeze(-2)
        s = (grid_sample_explicit(f, g, "zeros") if explicit else
             F.grid_sample(f, g, mode="bilinear", padding_mode="zeros", align_corners=True))
        toks.append(s.squeeze(-1).permute(0, 2, 1).contiguous())            # [B,17,C_l]
    if taps is not None:
        taps["sampled"] = toks
    emb = [_linear(P, f"{pre}.feat_embed.{i}", t) for i, t in enumerate(toks)]   # :220-221
    x = torch.stack([x, *emb], dim=1) + P[pre + ".Spatial_pos_embed"]       # :223-225
    if taps is not None:
        taps["tokens0"] = x
    if context_blocks:
        for i in range(levels):                                             # :228-229 (depth = levels, :169)
            x, pos = _deformable_block(P, f"{pre}.context_blocks.{i}", x, ref, feats, explicit=explicit,
                                       keep=keep["ctx"][i] if keep else None, nm=nm, cells=cells[i] if cells is not None else None)
            if taps is not None:
                taps.setdefault("ctx_pos", []).append(pos)
        if taps is not None:
            taps["tokens_ctx"] = x
    L = x.shape[1]
    x = x.permute(0, 2, 1, 3).reshape(b * p, L, -1)                          # 'b l p c -> (b p) l c' :231
    depth = levels if depth is None else depth        # ContextPose_mpi/model/pose_dformer.py:199: its own config key there
    for i in range(depth):
        x = _attn_block(P, f"{pre}.res_blocks.{i}", x, 8, keep=keep["res"][i] if keep else None, nm=nm)   # :233-234
    x = x.reshape(b, p, -1)                                                 # '(b p) l c -> b p (l c)' :235
    if taps is not None:
        taps["tokens_res"] = x
    for i in range(depth):
        x = _attn_block(P, f"{pre}.joint_blocks.{i}", x, 8, keep=keep["joint"][i] if keep else None, nm=nm)   # :237-238
    if taps is not None:
        taps["tokens_joint"] = x
    x = _linear(P, pre + ".head.1", _ln(P, pre + ".head.0", x, 1e-5))       # :240
    return x.view(b, 1, p, -1)


# ----------------------------------------------------------------------------------------------
# whole path + loss
# ----------------------------------------------------------------------------------------------
def normalise_crop_keypoints_(kcrop):
    """conpose.py:34-35 — IN PLACE, hard-coded 192x256 crop constants."""
    kcrop[..., :2] /= torch.tensor([192 // 2, 256 // 2], dtype=kcrop.dtype)
    kcrop[..., :2] -= torch.tensor([1, 1], dtype=kcrop.dtype)
    return kcrop


def ca_pf_forward(P, images, k2d, kcrop, backbone="hrnet_32", levels=4, explicit=False, taps=None, drop_masks=None,
                  emulate_bf16=False, cells=None):
    """CA_PF.forward conpose.py:30-42.  images [B,H,W,3] NHWC fp32; mutates kcrop in place.
    drop_masks: training-mode DropPath multipliers (see split_drop_masks), None = eval / no drop.
    emulate_bf16: the engine's compute_dtype = bf16 storage roundings (module docstring); False = the reference's fp32.
    cells: see lifter_forward (gradient tests)."""
    nm = BF16 if emulate_bf16 else FP32
    x = images.permute(0, 3, 1, 2).contiguous()
    ref = normalise_crop_keypoints_(kcrop)
    if emulate_bf16 == "stream_fp32":
        assert backbone != "cpn"
        feats = hrnet_forward(P, x, nm=BF16_STREAM_FP32)
    else:
        feats = cpn_forward(P, x, nm=nm, taps=taps) if backbone == "cpn" else hrnet_forward(P, x, nm=nm)
    if taps is not None:
        taps["ref"] = ref.clone()
        taps["features"] = feats
    return lifter_forward(P, k2d, ref, feats, levels=levels, explicit=explicit, taps=taps, drop_masks=drop_masks,
                          emulate_bf16=bool(emulate_bf16), cells=cells)


def mpjpe(pred, gt):
    """MPJPE.forward loss.py:16-22."""
    assert pred.shape == gt.shape
    return torch.mean(torch.norm(pred - gt, dim=len(gt.shape) - 1))


def gelu_exact(x):
    return 0.5 * x * (1.0 + math.erf(x / math.sqrt(2.0)))


# ----------------------------------------------------------------------------------------------
# neighbours of the path: restated from ContextPose/mvn/datasets/utils.py:33-82 and
# train.py:170-181.  Pinned twice: by tests/golden/prefetch.npz — outputs of the reference's OWN data_prefetcher class,
# executed on CPU with its CUDA-stream plumbing stubbed (oracle/_refshim.run_reference_prefetcher) — and by the
# hand-computable vectors in tests/test_oracle_golden.py.
# ----------------------------------------------------------------------------------------------
JOINTS_LEFT = [4, 5, 6, 11, 12, 13]      # datasets/utils.py:12
JOINTS_RIGHT = [1, 2, 3, 14, 15, 16]     # :13


def prefetch_preprocess(images_u8, gt, k2d, kcrop, backbone="hrnet_32", is_train=False, flip=False, flip_test=False,
                        scalar_div="cuda"):
    """data_prefetcher.preload.  `flip` replaces the reference's `random.random() <= 0.5` draw (:55).
    scalar_div: how `images / 255.0` (:47) is evaluated.  "cuda" (default): multiplication by the fp32 reciprocal — what
    ATen's CUDA kernel does for a Python-scalar divisor, and the reference's prefetcher only ever runs on a GPU; this is the
    mode capf_preprocess is held to bit for bit.  "cpu": a true division, what the same line does when the reference class is
    run on CPU — the mode that tests/golden/prefetch.npz (captured from the reference's own class, oracle/make_goldens.py)
    pins bit for bit.  The two differ by at most one ulp of the quotient and in nothing else."""
    images = torch.flip(images_u8, [-1]).float()                                                          # :45
    if scalar_div == "cuda":
        images = images * torch.tensor(1.0, dtype=torch.float32).div(255.0)                              # :47
    else:
        images = images / 255.0
    if backbone in ("hrnet_32", "hrnet_48"):
        mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
        images = (images - mean) / std                                                                  # :47-48
    else:
        images = images - (torch.tensor([122.7717, 115.9465, 102.9801]) / 255.0).view(1, 1, 1, 3)        # :27-29,:50
    gt, k2d, kcrop = gt.clone(), k2d.clone(), kcrop.clone()
    gt[:, :, 1:] -= gt[:, :, :1]                                                                        # :52
    gt[:, :, 0] = 0                                                                                     # :53
    L, R = JOINTS_LEFT, JOINTS_RIGHT
    if flip and is_train:                                                                               # :55-65
        images = torch.flip(images, [-2])
        k2d[..., 0] *= -1
        k2d[..., L + R, :] = k2d[..., R + L, :]
        kcrop[:, :, 0] = 192 - kcrop[:, :, 0] - 1
        kcrop[:, L + R] = kcrop[:, R + L]
        gt[:, :, :, 0] *= -1
        gt[:, :, L + R] = gt[:, :, R + L]
    if (not is_train) and flip_test:                                                                    # :67-80
        images = torch.stack([images, torch.flip(images, [2])], dim=1)
        k2f = k2d.clone()
        k2f[..., 0] *= -1
        k2f[..., L + R, :] = k2f[..., R + L, :]
        k2d = torch.stack([k2d, k2f], dim=1)
        kcf = kcrop.clone()
        kcf[:, :, 0] = 192 - kcf[:, :, 0] - 1
        kcf[:, L + R] = kcf[:, R + L]
        kcrop = torch.stack([kcrop, kcf], dim=1)
    return images.float(), gt.float(), k2d.float(), kcrop.float()


def fliptest_fuse(pred, pred_flip):
    """train.py:177-180."""
    pred_flip = pred_flip.clone()
    pred_flip[:, :, :, 0] *= -1
    pred_flip[:, :, JOINTS_LEFT + JOINTS_RIGHT] = pred_flip[:, :, JOINTS_RIGHT + JOINTS_LEFT]
    return torch.mean(torch.cat((pred, pred_flip), dim=1), dim=1, keepdim=True)
