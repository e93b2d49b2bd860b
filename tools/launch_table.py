#!/usr/bin/env python
"""Per-launch table of one forward of the product schedule (capf_forward_profile_launches).  (GPU box)
Usage: python tools/launch_table.py [--batch 64] [--backbone hrnet_32] [--dtype fp32] [--filter volume_net]"""
import argparse
import contextlib
import copy
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
from capf import synth
from mvn.models.conpose import CA_PF
from mvn.utils.cfg import backbone_preset, config


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--backbone", default="hrnet_32")
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--filter", default="")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--plan-flags", type=int, default=0)
    a = ap.parse_args()
    cfg = backbone_preset(copy.deepcopy(config), a.backbone)
    cfg.model.backbone.fix_weights = True
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg, compute_dtype=a.dtype, plan_flags=a.plan_flags).eval()
    synth.load_synthetic(model, seed=1, bn_mode="random")
    model = model.cuda()
    img, k2d, kc = synth.synth_inputs(a.batch, a.height, a.width, seed=1000, crop_range=(192, 256))
    img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
    with torch.no_grad():
        out = model(img, k2d, kc.clone())
    eng = model.engine_for(img)
    table = eng.op_table(a.batch)
    stream = torch.cuda.current_stream().cuda_stream
    acc = None
    for _ in range(a.reps):
        ms, leader = eng.forward_profile_launches(img, k2d, kc.clone(), out, stream)
        acc = ms if acc is None else [x + y for x, y in zip(acc, ms)]
    ms = [x / a.reps for x in acc]
    members = {}
    for i, l in enumerate(leader):
        if l >= 0:
            members.setdefault(l, []).append(i)
    total = 0.0
    for l in sorted(members):
        ops = members[l]
        name = table[l][0] + (f" (+{len(ops) - 1})" if len(ops) > 1 else "")
        flops = sum(table[i][2] for i in ops)
        total += ms[l]
        if a.filter and a.filter not in name:
            continue
        tf = flops / (ms[l] * 1e-3) / 1e12 if ms[l] > 0 else 0.0
        print(f"{l:4d} {name:64s} {table[l][1]:30s} {ms[l] * 1e3:8.1f} us {flops / 1e9:8.3f} GF {tf:7.2f} TF")
    print(f"total {total:.3f} ms over {len(members)} launches")


if __name__ == "__main__":
    main()
