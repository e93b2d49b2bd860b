"""GPU: layer-wise ("teacher-forced") parity of the backbone plan AT THE BASELINE BATCH SIZES.  Every conv / fuse / pool /
resize launch of the product schedule (grouped launches, Winograd, row-halo, ping-pong: whatever the engine picks at that
batch) is recomputed on the CPU from the operands the ENGINE itself produced (oracle/op_oracle.py) and compared output for
output: fp32 to 2e-5 of each output's sum of |terms| (fp32 summation order), bf16 to "the same or the adjacent bf16 number
once the fp32 pre-images are allowed that much; at most 3 % of a tensor inexact at all" (op_oracle.compare).

This is the tight check the end-to-end comparisons cannot be: a deep bf16 network is chaotic at the rounding level (two
correct evaluations that differ in fp32 summation order drift to the full bf16 noise floor; test_gpu_sampling.py prints the
three mutually equidistant points HIP / bf16-emulating oracle / fp32 oracle), so a kernel bug worth 1e-2 would hide inside
that floor end to end — but not here, where it shows up as a whole tensor off in the very op that has it."""
import pytest
import torch

import capf_oracle as oracle
import op_oracle
from capf import synth
from test_gpu_fullsize import _model

pytestmark = pytest.mark.gpu


def layerwise(backbone, dtype, B, H, W, rows, wseed=71, iseed=72, plan_flags=0):
    model, sd = _model(backbone, dtype, wseed, plan_flags)
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=iseed, crop_range=(W, H))
    img_d = img.cuda()
    eng = model.engine_for(img_d)
    names = [n for n, _, _ in eng.schema()]
    n_ops = eng.lib.capf_num_ops(eng.h)
    descs = [eng.op_describe(i) for i in range(n_ops)]
    table = eng.op_table(B)
    todo = [i for i, d in enumerate(descs) if d.backbone and d.kind in (0, 1, 2, 3)]
    stream = torch.cuda.current_stream().cuda_stream
    bf = dtype == "bf16"
    worst, kernels, n_checked, flips, n_up, n_planes = {}, set(), 0, 0, 0, 0
    for cp in sorted(set(descs[i].checkpoint for i in todo)):
        eng.forward_prefix(img_d, cp, stream)
        torch.cuda.synchronize()
        for i in [i for i in todo if descs[i].checkpoint == cp]:
            d = descs[i]
            take = lambda slot, h, w, c, dt: eng.op_tensor_fp32(i, slot, (B, h, w, c), dt)[rows].cpu()      # (planes between a BasicBlock's convs: decoded)
            n_planes += eng.op_h2_planes(i, B)[0] == 2
            got = take(5, d.Ho, d.Wo, d.Cout, d.out_dtype)
            out_bf = d.out_dtype == 2
            mass = term = None
            with torch.no_grad():
                if d.kind == 0:
                    assert d.conv
                    x = take(0, d.H, d.W, d.Cin, d.in_dtype)
                    res = take(4, d.Ho, d.Wo, d.Cout, d.out_dtype) if d.has_residual else None
                    conv = names[d.p_weight][:-len(".weight")]
                    bn = names[d.p_bn_weight][:-len(".weight")]
                    up = take(1, d.up_H, d.up_W, d.Cout, d.out_dtype) if d.up_H > 0 else None      # (+ its bilinear upsample behind the activation)
                    n_up += up is not None
                    want, mass, term = op_oracle.conv_bn_act(sd, conv, bn, x, res, d.ks, d.stride, d.pad, d.act, bool(d.mfma_bf16), up)
                elif d.kind == 1:
                    ins = [take(k, d.H >> d.shift[k], d.W >> d.shift[k], d.Cin, d.in_dtype) for k in range(d.n_in)]
                    want = op_oracle.fuse_sum(ins, [d.shift[k] for k in range(d.n_in)], d.relu, out_bf)
                elif d.kind == 2:
                    want = op_oracle.maxpool(take(0, d.H, d.W, d.Cin, d.in_dtype))
                else:
                    add = take(4, d.Ho, d.Wo, d.Cout, d.out_dtype) if d.has_residual else None
                    want = op_oracle.resize(take(0, d.H, d.W, d.Cin, d.in_dtype), d.Ho, d.Wo, out_bf, add)
            r = op_oracle.compare(got, want, out_bf, mass, term)
            flips += r["weight_flips"]
            kern = table[i][1] or ("fuse_sum", "maxpool", "resize")[d.kind - 1]
            kernels.add(kern)
            w = worst.setdefault(kern, (0.0, 0.0, ""))
            if r["max_err"] >= w[0]:
                worst[kern] = (r["max_err"], max(w[1], r["frac_inexact"]), table[i][0])
            else:
                worst[kern] = (w[0], max(w[1], r["frac_inexact"]), w[2])
            assert r["ok"], (table[i][0], kern, r)
            n_checked += 1
    print(f"{backbone} {dtype} B={B}: {n_checked} backbone ops recomputed from the engine's own operands on {len(rows)} frames"
          f" ({flips} output channels explained by a folded weight on a bf16 rounding boundary)")
    assert flips <= 4
    for k, (e, f, name) in sorted(worst.items()):
        print(f"    {k:38s} worst error {e:9.2e} ({'of the allowance' if bf else 'of the range'})   largest inexact fraction {f:8.2e}   ({name})")
    assert n_checked == len(todo) and n_checked > 85
    layerwise.fused_upsample_adds = n_up
    layerwise.planes_consumers = n_planes
    return kernels


def test_cfg1_layerwise_hrnet32_fp32_batch64():
    k = layerwise("hrnet_32", "fp32", 64, 256, 256, [0, 21, 42, 63])
    assert any(x.startswith("igemm_f32h2_") for x in k)          # (the branch convs: split-fp32 tile from 370 MFLOP per conv, batch >= 5)
    assert layerwise.planes_consumers == 0                       # (planes between a BasicBlock's convs are opt-in: CAPF_PLAN_H2_PLANES)


def test_layerwise_with_planes_between_the_basic_blocks_convs():
    """CAPF_PLAN_H2_PLANES (opt-in; EXPERIMENTS R6.5): every BasicBlock conv1 of stages 2-4 writes its output as split fp16 planes + one scale
    exponent per (tile, 16-channel chunk), every conv2 stages them as they are; the engine's operands are decoded with capf_op_h2_planes and held
    to the same per-op bound as the fp32-tensor plan."""
    from capf.lib import PLAN_H2_PLANES
    k = layerwise("hrnet_32", "fp32", 16, 256, 256, [0, 7, 15], plan_flags=PLAN_H2_PLANES)
    assert any(x.startswith("igemm_f32h2_") for x in k) and layerwise.planes_consumers == 104


def test_cfg3_layerwise_hrnet32_fp32_batch512():
    """configs[3]'s per-GPU batch: the size at which the Winograd kernel's algorithmic rate reads 1.03 of the nominal peak."""
    k = layerwise("hrnet_32", "fp32", 512, 256, 256, [0, 170, 341, 511])
    assert any(x.startswith("igemm_f32h2_") for x in k)


def test_cfg2_layerwise_hrnet48_bf16_batch256():
    k = layerwise("hrnet_48", "bf16", 256, 256, 256, [0, 85, 170, 255])
    assert "bneck0_bf16<8x8>" in k and "bneck1_bf16<8x8>" in k          # layer1.0 / layer1.1-3 as one kernel each: its four convs checked from the operands the kernel itself stored (TAP variant)


def test_cfg4_layerwise_cpn_bf16_batch128():
    k = layerwise("cpn", "bf16", 128, 384, 288, [0, 42, 85, 127])
    assert "bneck0_bf16<8x8>" in k and "bneck1_bf16<8x8>" in k          # (round 6: every layer1 bottleneck fused, networks/resnet.py:58-93)
    assert layerwise.fused_upsample_adds == 3          # globalNet's three lateral convs carry the upsampled path in their epilogue (round 6)


def test_layerwise_small_batches_take_the_other_kernels():
    """B=1 (split-K, direct kernels below the Winograd threshold) and B=2 bf16 (ring schedule instead of ping-pong / row-halo)."""
    layerwise("hrnet_32", "fp32", 1, 256, 256, [0])
    layerwise("hrnet_48", "bf16", 2, 256, 256, [0, 1])
    layerwise("cpn", "fp32", 2, 384, 288, [0, 1])


def lifter_layerwise(backbone, B, H, W, frames, wseed=73, iseed=74):
    """The lifter half of the tight check under compute_dtype = bf16: every LayerNorm (bf16 writer), attention (bf16 writer) and qkv / proj /
    fc1 + GELU / fc2 launch of the three block groups (pose_dformer.py:15-79) recomputed on the CPU from the rows the ENGINE produced.
    proj and fc2 update the token stream in place, so an op's operands are read after capf_forward_prefix(i) and its result after
    capf_forward_prefix(i + 1) (the schedule is deterministic).  fp32 results: 2e-5 of each output's sum of |terms|; bf16 results: the
    same or the adjacent bf16 number."""
    model, sd = _model(backbone, "bf16", wseed)
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=iseed, crop_range=(W, H))
    img_d, k2d_d, kc_d = img.cuda(), k2d.cuda(), kc.cuda()
    out = torch.empty(B, 1, 17, 3, device="cuda")
    eng = model.engine_for(img_d)
    names = [n for n, _, _ in eng.schema()]
    n_ops = eng.lib.capf_num_ops(eng.h)
    descs = [eng.op_describe(i) for i in range(n_ops)]
    table = eng.op_table(B)
    stream = torch.cuda.current_stream().cuda_stream
    todo = [i for i, d in enumerate(descs) if not d.backbone and ((d.kind == 0 and not d.conv) or d.kind in (4, 5))]
    fr = torch.tensor(frames)

    def rows_of(i, slot, k, width, dt):
        d = descs[i]
        G, S1, S2, off = (int(v) for v in d.maps[k])
        m = (fr[:, None] * d.rows_per_frame + torch.arange(d.rows_per_frame)[None, :]).reshape(-1)
        addr = (m // G) * S1 + (m % G) * S2 + off
        mb = torch.arange(B * d.rows_per_frame)
        n = int(((mb // G) * S1 + (mb % G) * S2 + off).max()) + width
        flat = eng.op_tensor(i, slot, (n,), dt)
        return flat[(addr[:, None] + torch.arange(width)[None, :]).cuda()].cpu()

    counts, worst = {}, {}
    for i in todo:
        d = descs[i]
        eng.forward_prefix(img_d, i, stream, k2d_d, kc_d.clone(), out)
        torch.cuda.synchronize()
        a = rows_of(i, 0, 0, d.Cin, d.in_dtype)
        res = rows_of(i, 4, 2, d.Cout if d.kind == 0 else d.Cin, 0) if d.has_residual else None
        eng.forward_prefix(img_d, i + 1, stream, k2d_d, kc_d.clone(), out)
        torch.cuda.synchronize()
        got = rows_of(i, 5, 1, d.Cout, d.out_dtype)
        out_bf = d.out_dtype == 2
        mass = None
        with torch.no_grad():
            if d.kind == 0:
                assert d.p_weight >= 0 and d.p_ln_weight < 0
                want, mass = op_oracle.linear_rows(a, sd[names[d.p_weight]], sd[names[d.p_bias]], res, d.act == 2, d.in_dtype == 2, out_bf)
                kind = ("fc1+gelu" if d.act == 2 else table[i][0].split(".")[-1]) + (" bf16" if d.in_dtype == 2 else " fp32")
            elif d.kind == 4:
                want = op_oracle.layernorm_rows(a, res, sd[names[d.p_ln_weight]], sd[names[d.p_ln_bias]], d.eps, out_bf)
                kind = "layernorm"
            else:
                g, t, h, hd = (int(v) for v in d.attn)
                want = op_oracle.attention_rows(a, g * len(frames), t, h, hd, out_bf)
                kind = "attention"
        r = op_oracle.compare(got, want, out_bf, mass)
        assert r["ok"], (table[i][0], table[i][1], r)
        counts[kind] = counts.get(kind, 0) + 1
        worst[kind] = max(worst.get(kind, 0.0), r["max_err"])
    print(f"{backbone} bf16 B={B}: {len(todo)} lifter ops recomputed from the engine's own rows on {len(frames)} frames")
    for k in sorted(counts):
        print(f"    {k:18s} x {counts[k]:2d}   worst error {worst[k]:9.2e} of the allowance")
    return counts


def test_cfg2_lifter_layerwise_hrnet48_bf16_batch256():
    c = lifter_layerwise("hrnet_48", 256, 256, 256, [0, 85, 170, 255])
    assert c.get("qkv bf16") == 8 and c.get("proj bf16") == 8 and c.get("fc1+gelu bf16") == 12 and c.get("fc2 bf16") == 12
    assert c.get("layernorm") == 20 and c.get("attention") == 8


def test_cfg4_lifter_layerwise_cpn_bf16_batch128():
    c = lifter_layerwise("cpn", 128, 384, 288, [0, 42, 85, 127])
    assert c.get("qkv bf16") == 8 and c.get("fc2 bf16") == 12 and c.get("layernorm") == 20 and c.get("attention") == 8
