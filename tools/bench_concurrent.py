#!/usr/bin/env python
"""Experiment: how well do independent conv layers of the four HRNet branches overlap when launched on
separate streams (what the engine's lanes do), against running them back to back?  (GPU box)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
from capf import lib as capf

BRANCHES = [(32, 64), (64, 32), (128, 16), (256, 8)]     # (channels, resolution) of HRNet-W32 stage 4 @256x256


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--branches", type=str, default="0,1,2,3")
    ap.add_argument("--only-grouped", action="store_true", help="run only the grouped launch (PMC collection)")
    a = ap.parse_args()
    sel = [int(x) for x in a.branches.split(",")]
    probs = []
    for b in sel:
        c, r = BRANCHES[b]
        x = torch.randn(a.batch, r, r, c, device="cuda")
        w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
        wp, bias = capf.pack_conv(w)
        probs.append((x, wp, bias, torch.cuda.Stream()))

    group = [(x, wp, bias, 3, 1, 1, None) for x, wp, bias, _ in probs]

    def run(concurrent):
        main_s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if concurrent == "group":
            for _ in range(a.iters):
                capf.conv_nhwc_group(group)
        elif concurrent == "two chains":      # {first, last} and {middle ...} as two grouped chains on two streams
            sets = [[group[0], group[-1]], group[1:-1]] if len(group) >= 4 else [group[:1], group[1:]]
            for gs, (_, _, _, s) in zip(sets, probs):
                s.wait_stream(main_s)
                with torch.cuda.stream(s):
                    for _ in range(a.iters):
                        if len(gs) > 1:
                            capf.conv_nhwc_group(gs)
                        else:
                            x, wp, bias = gs[0][:3]
                            capf.conv_nhwc(x, wp, bias, 3, 1, act=1)
            for _, _, _, s in probs[:2]:
                main_s.wait_stream(s)
        elif concurrent:
            evs = []
            for x, wp, bias, s in probs:
                s.wait_stream(main_s)
                with torch.cuda.stream(s):
                    for _ in range(a.iters):
                        capf.conv_nhwc(x, wp, bias, 3, 1, act=1)
            for _, _, _, s in probs:
                main_s.wait_stream(s)
        else:
            for _ in range(a.iters):
                for x, wp, bias, _ in probs:
                    capf.conv_nhwc(x, wp, bias, 3, 1, act=1)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.iters

    if a.only_grouped:
        us = run("group")
        print(f"branches {sel}: one grouped launch {us:8.1f} us per level")
        return
    for _ in range(2):
        run(False), run(True), run("group"), run("two chains")
    flops = sum(2.0 * a.batch * r * r * c * c * 9 for c, r in (BRANCHES[b] for b in sel))
    for name, conc in (("back to back", False), ("one stream per branch", True), ("one grouped launch", "group"),
                       ("two grouped chains", "two chains")):
        us = run(conc)
        print(f"branches {sel}: {name:22s} {us:8.1f} us per level   {flops / us / 1e6:7.2f} TFLOP/s")


if __name__ == "__main__":
    main()
