#!/usr/bin/env python
"""Re-derive the batch thresholds of DESIGN 4.4 from a sweep on THIS box ("so they are not single-box lore").  (GPU box)

Every rule of the plan that a capf_plan_flag can switch is timed on both sides of its threshold: ms per forward of the default plan and of
the plan with the kernel family taken out, at batches around the rule's switch-over.  The rule is right where the default column wins (or ties)
at and above the threshold and the alternative wins (or ties) below it.  Rules whose threshold is a compile-time constant without a flag (64-channel
tiles from 512 tiles, ping-pong bf16 kernel from 2048 tiles) are covered by profiles/r05_batch_sweep.txt's end-to-end monotonicity only.

    python tools/sweep_thresholds.py > gpurun_out/threshold_sweep.txt       # ~6 minutes
"""
import contextlib
import copy
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
from capf import synth
from capf.lib import PLAN_NO_F32H2_GEMM, PLAN_NO_F32X3, PLAN_NO_PWCHAIN, PLAN_NO_UPADD, PLAN_NO_WINOGRAD, PLAN_NO_WS
from mvn.models.conpose import CA_PF
from mvn.utils.cfg import backbone_preset, config


def forward_ms(backbone, dtype, batch, flags, H=256, W=256, reps=20):
    cfg = backbone_preset(copy.deepcopy(config), backbone)
    cfg.model.backbone.fix_weights = True
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg, compute_dtype=dtype, plan_flags=flags).eval()
    synth.load_synthetic(model, seed=1, bn_mode="random")
    model = model.cuda()
    img, k2d, kc = synth.synth_inputs(batch, H, W, seed=3, crop_range=(192, 256))
    img, k2d, kc0 = img.cuda(), k2d.cuda(), kc.cuda()
    kc = kc0.clone()
    with torch.no_grad():
        for _ in range(5):
            kc.copy_(kc0); model(img, k2d, kc)
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            kc.copy_(kc0)
            e0.record(); model(img, k2d, kc); e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
    del model
    torch.cuda.empty_cache()
    ts.sort()
    return ts[len(ts) // 2]


RULES = [
    ("fp32: two-fp16-piece tile + GEMM from batch 5 (below: direct fp32 kernels with split-K)", "hrnet_32", "fp32", 256, 256,
     PLAN_NO_F32X3 | PLAN_NO_F32H2_GEMM, "fp32 pipe", 5, [2, 3, 4, 5, 6, 8, 12]),
    ("fp32: layer1 conv3 -> conv1 pairs as ONE chained launch from 131072 rows = batch 32 at 64x64", "hrnet_32", "fp32", 256, 256,
     PLAN_NO_PWCHAIN, "two launches", 32, [8, 16, 24, 32, 48, 64]),
    ("fp32 without the split tiles (CAPF_PLAN_NO_F32X3): Winograd from batch 24 (below: two-piece GEMM / direct)", "hrnet_32", "fp32", 256, 256,
     PLAN_NO_F32X3 | PLAN_NO_WINOGRAD, "no Winograd", 24, [8, 16, 24, 32, 48]),
    ("bf16: 2-D halo tile for 3x3 stride-1 convs from 1 GFLOP per conv (HRNet-48: batch 3 for the 48-channel branch)", "hrnet_48", "bf16", 256, 256,
     PLAN_NO_WS, "row-halo / ring kernels", 3, [1, 2, 3, 4, 8, 16]),
    ("bf16 CPN: lateral + upsampled path inside the lateral conv's epilogue (every batch)", "cpn", "bf16", 384, 288,
     PLAN_NO_UPADD, "resize-add launch", 1, [1, 4, 16, 64, 128]),
]


def main():
    print(f"threshold sweep on {torch.cuda.get_device_name(0)}; ms per forward, median of 20 (default plan | alternative)")
    for name, bb, dt, H, W, flags, alt_name, thr, batches in RULES:
        base = PLAN_NO_F32X3 if "without the split tiles" in name else 0
        print(f"\n== {name}\n   alternative = {alt_name} (plan_flags {flags}); rule switches at batch {thr}")
        for b in batches:
            d, a = forward_ms(bb, dt, b, base, H, W), forward_ms(bb, dt, b, flags, H, W)
            side = "rule: default" if b >= thr else "rule: same kernels either way" if base == 0 and b < thr and "from batch" in name else "rule: default"
            win = "default" if d <= a else "alternative"
            print(f"   batch {b:4d}: {d:8.3f} | {a:8.3f}   {a / d:5.2f}x   faster: {win:11s}" + ("   <- threshold" if b == thr else ""))


if __name__ == "__main__":
    main()
