"""Single-op CPU restatements for the layer-wise ("teacher-forced") parity tests.

TEST INFRASTRUCTURE — NOT PRODUCT CODE (same rules as capf_oracle.py: only tests/ may import it).

Why it exists: a deep bf16 network is chaotic at the rounding level — two correct implementations that differ only in fp32
summation order decorrelate to the full bf16 noise floor within a few dozen layers — so the end-to-end distance between the
HIP path and ANY CPU evaluation (fp32 or bf16-emulating) cannot be bounded tighter than that floor.  One op at a time it can:
each function here recomputes ONE op of the engine's plan (csrc/plan.cpp) from the operands the ENGINE itself produced
(capf_op_tensor), with the same storage roundings (capf_oracle.Numerics), so the outputs agree to fp32 summation order — in
bf16 storage: bit for bit except for the rare value whose fp32 pre-image straddles a rounding boundary (one bf16 ulp).
The ops follow the reference's modules exactly as capf_oracle does:
    conv + eval BatchNorm (+ residual) (+ ReLU)   pose_hrnet.py:66-136, 238-275, 321-327, 383-407; networks/resnet.py:58-93, 137-139;
                                                  globalNet.py:29-45; refineNet.py:3-45
    fuse sum (nearest upsample + add + ReLU)      pose_hrnet.py:294-301
    max-pool 3x3 s2 p1                            networks/resnet.py:140
    bilinear resize, align_corners=True           globalNet.py:40, refineNet.py:61
    nn.Linear (+ residual) (+ GELU), LayerNorm,
    multi-head self-attention of the lifter       pose_dformer.py:15-79 (Mlp :15-31, Attention :34-59, Block :62-79)
"""
import torch
import torch.nn.functional as F

import capf_oracle as oracle


def _nchw(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def conv_bn_act(P, conv, bn, x_nhwc, residual_nhwc, ks, stride, pad, act, bf16, up_nhwc=None):
    """x / residual: the engine's own NHWC operands (fp32 or bf16 tensors) -> (NHWC fp32 holding what the engine must store,
    NHWC fp32 `mass` = sum_k |x_k| |w_k| + |bias| + |residual| per output: the scale fp32 summation-order noise is relative to —
    an output that is the small remainder of large cancelling terms cannot be reproduced to a fraction of ITS magnitude)."""
    nm = oracle.BF16 if bf16 else oracle.FP32
    x = nm.r(_nchw(x_nhwc))                                  # (the stem rounds the fp32 image on its way into LDS)
    res = _nchw(residual_nhwc) if residual_nhwc is not None else None
    assert act in (0, 1)
    if up_nhwc is None:
        y = oracle._cbr(P, conv, bn, x, stride, pad, relu=(act == 1), res=res, nm=nm)
    else:
        # + bilinear_upsample(up) BEHIND the activation, in fp32, one storage rounding at the end (capf_op_desc.up_H: CPN's lateral conv with
        # the upsampled path added in its epilogue, globalNet.py:66; csrc/plan.cpp build_cpn)
        y = oracle._cbr(P, conv, bn, x, stride, pad, relu=(act == 1), res=res, nm=(oracle._NOROUND if bf16 else nm))
        y = nm.r(y + F.interpolate(_nchw(up_nhwc), size=y.shape[-2:], mode="bilinear", align_corners=True))
    sc = P[bn + ".weight"] / torch.sqrt(P[bn + ".running_var"] + oracle.BN_EPS)
    w = (P[conv + ".weight"] * sc.view(-1, 1, 1, 1)).abs()
    mass = F.conv2d(x.abs(), w, (P[bn + ".bias"] - P[bn + ".running_mean"] * sc).abs(), stride, pad)
    if res is not None:
        mass = mass + res.abs()
    if up_nhwc is not None:
        mass = mass + F.interpolate(_nchw(up_nhwc).abs(), size=mass.shape[-2:], mode="bilinear", align_corners=True)
    # largest single term |x_k w_k| an output can contain (bound: largest |x| in its window times the channel's largest |w|):
    # the scale of ONE folded weight landing on the other side of a bf16 rounding boundary (see compare)
    xmax = F.max_pool2d(x.abs().amax(dim=1, keepdim=True), ks, stride, pad)
    term = xmax * w.amax(dim=(1, 2, 3)).view(1, -1, 1, 1)
    return _nhwc(y), _nhwc(mass), _nhwc(term)


def fuse_sum(inputs_nhwc, shifts, relu, bf16):
    nm = oracle.BF16 if bf16 else oracle.FP32
    acc = None
    for t, s in zip(inputs_nhwc, shifts):
        t = _nchw(t)
        if s:
            t = F.interpolate(t, scale_factor=2 ** s, mode="nearest")
        acc = t if acc is None else acc + t
    if relu:
        acc = F.relu(acc)
    return _nhwc(nm.r(acc))


def maxpool(x_nhwc):
    return _nhwc(F.max_pool2d(_nchw(x_nhwc), 3, 2, 1))


def resize(x_nhwc, Ho, Wo, bf16, add_nhwc=None):
    """bilinear, align_corners=True (+ a map of the output's shape added in fp32 before the one storage rounding: the engine's
    globalNet runs the 1x1 conv of an `upsamples` branch BEFORE the interpolation and adds the lateral here)"""
    nm = oracle.BF16 if bf16 else oracle.FP32
    y = F.interpolate(_nchw(x_nhwc), size=(Ho, Wo), mode="bilinear", align_corners=True)
    if add_nhwc is not None:
        y = y + _nchw(add_nhwc)
    return _nhwc(nm.r(y))


def linear_rows(a_rows, w, b, res_rows, gelu, bf16_operands, out_bf16):
    """One nn.Linear of the lifter on the engine's own rows: a_rows [M, K] (bf16 rows if the op runs on the bf16 MFMA path), w [N, K],
    b [N], optional fp32 residual rows -> (what the engine must store, mass = sum |a| |w| + |b| + |residual|)."""
    a = a_rows.float()
    w = oracle.bf16_round(w) if bf16_operands else w
    y = a @ w.t() + b
    mass = a.abs() @ w.abs().t() + b.abs()
    if res_rows is not None:
        y = y + res_rows.float()
        mass = mass + res_rows.float().abs()
    if gelu:
        y = F.gelu(y)                                        # exact erf form (pose_dformer.py:17 nn.GELU)
        mass = mass * 1.2                                    # |gelu'| <= 1.13
    return (oracle.bf16_round(y) if out_bf16 else y), mass


def layernorm_rows(x_rows, add_rows, g, b, eps, out_bf16):
    x = x_rows.float() if add_rows is None else x_rows.float() + add_rows.float()
    y = F.layer_norm(x, (x.shape[-1],), g, b, eps)
    return oracle.bf16_round(y) if out_bf16 else y


def attention_rows(qkv_rows, groups, tokens, heads, hd, out_bf16):
    """qkv rows [groups * tokens, 3 * heads * hd] ordered (q | k | v) x heads x hd (pose_dformer.py:49) -> [groups * tokens, heads * hd]"""
    qkv = qkv_rows.float().view(groups, tokens, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    y = (att @ v).transpose(1, 2).reshape(groups * tokens, heads * hd)
    return oracle.bf16_round(y) if out_bf16 else y


def compare(got, want, bf16, mass=None, term=None):
    """-> dict(max_err, frac_inexact, ok, weight_flips).  `mass` (conv ops): per-output sum of |terms|; fp32 summation in another
    order moves an output by ~1e-6 of it (K <= 3456 terms, eps 6e-8, random-walk growth), bounded here by 2e-5 * mass (Winograd
    F(4,3) amplifies roundoff by its 1/24 .. 8 transform constants: measured 1e-5 of the range per conv).
    fp32 storage: |got - want| <= 2e-5 * mass (no mass: 1e-5 of the tensor's range).
    bf16 storage: got and want must be the SAME or ADJACENT bf16 numbers (one rounding flip), after allowing the fp32 pre-images
    the same 2e-5 * mass; at most 3 % of a tensor may be inexact at all.  One more thing can legitimately differ: the BatchNorm
    fold w * gamma / sqrt(var + eps) is evaluated once by the GPU and once by the CPU, and a folded weight whose fp32 value sits
    on a bf16 rounding boundary may round the other way (probability ~2^-16 per weight).  Such a weight moves every output of ITS
    channel by up to 2^-8 |x_k w_k|: outputs outside the allowance are accepted iff they are within that much (`term`) AND all lie
    in at most two output channels (a kernel bug does not confine itself to one channel's worth of one tap)."""
    got, want = got.float(), want.float()
    d = (got - want).abs()
    slack = 2e-5 * mass if mass is not None else 1e-5 * want.abs().max()
    if not bf16:
        worst = (d / (slack + 1e-30)).max().item()
        return {"max_err": (d.max() / want.abs().max().clamp_min(1e-30)).item(), "frac_inexact": (d > 0).float().mean().item(),
                "ok": worst <= 1.0, "weight_flips": 0}
    mag = torch.maximum(got.abs(), want.abs())
    ulp = torch.pow(2.0, torch.floor(torch.log2(mag.clamp_min(1e-30))) - 7)        # spacing of bf16 numbers at that magnitude
    allowed = ulp + slack
    frac = (d > 0).float().mean().item()
    bad = d > allowed
    flips = 0
    if bool(bad.any()) and term is not None:
        chans = torch.nonzero(bad.reshape(-1, bad.shape[-1]).any(dim=0)).flatten().tolist()
        if len(chans) <= 2 and bool((d[bad] <= (allowed + term / 256.0)[bad]).all()):
            flips = len(chans)
            bad = torch.zeros_like(bad)
    return {"max_err": (d / allowed).max().item(), "frac_inexact": frac, "ok": (not bool(bad.any())) and frac <= 0.03,
            "weight_flips": flips}
