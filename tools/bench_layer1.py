#!/usr/bin/env python
"""Experiment: HRNet layer1 (4 bottlenecks, 256-channel 64x64 activations: 268 MB per tensor at batch 64, more than
the 256 MB Infinity Cache) run for the whole batch at once vs in sub-batches whose working set stays cache
resident.  (GPU box)   python tools/bench_layer1.py --batch 64 --parts 1,2,4"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
from capf import lib as capf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--parts", type=str, default="1,2,4")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    torch.manual_seed(0)
    B = a.batch
    x0 = torch.randn(B, 64, 64, 256, device="cuda")

    def mk(ci, co, ks):
        w = torch.randn(co, ci, ks, ks, device="cuda") * (1.0 / (ci * ks * ks) ** 0.5)
        return capf.pack_conv(w)

    blocks = [(mk(256, 64, 1), mk(64, 64, 3), mk(64, 256, 1)) for _ in range(3)]     # layer1.1 .. layer1.3 (no projection)

    def run(parts):
        outs = []
        b = B // parts
        for p in range(parts):
            x = x0[p * b:(p + 1) * b]
            for (w1, b1), (w2, b2), (w3, b3) in blocks:
                y = capf.conv_nhwc(x, w1, b1, 1, 1, act=1)
                y = capf.conv_nhwc(y, w2, b2, 3, 1, act=1)
                x = capf.conv_nhwc(y, w3, b3, 1, 1, act=1, residual=x)
            outs.append(x)
        return outs

    for parts in [int(v) for v in a.parts.split(",")]:
        for _ in range(2):
            run(parts)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run(parts)
        e1.record()
        torch.cuda.synchronize()
        print(f"batch {B} in {parts} part(s): {e0.elapsed_time(e1) / a.iters * 1e3:8.1f} us for three bottlenecks")


if __name__ == "__main__":
    main()
