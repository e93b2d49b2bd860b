#!/usr/bin/env python
"""Micro-benchmark of single conv layers through the stateless C-ABI operator (capf_op_conv).
Usage: python tools/bench_conv.py [--batch 64] [--iters 20] [--only IDX]   (GPU box)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
from capf import lib as capf

# (Cin, Cout, ks, stride, H, W)  — the HRNet-W32 @256x256 shape classes
SHAPES = [(32, 32, 1, 1, 64, 64), (32, 32, 3, 1, 64, 64), (64, 64, 3, 1, 32, 32), (128, 128, 3, 1, 16, 16), (256, 256, 3, 1, 8, 8),
          (64, 64, 3, 1, 64, 64), (64, 256, 1, 1, 64, 64), (256, 64, 1, 1, 64, 64), (256, 32, 3, 1, 64, 64),
          (64, 64, 3, 2, 128, 128), (3, 64, 3, 2, 256, 256), (32, 64, 3, 2, 64, 64), (64, 32, 1, 1, 32, 32)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=int, default=-1)
    a = ap.parse_args()
    torch.manual_seed(0)
    for i, (ci, co, ks, st, H, W) in enumerate(SHAPES):
        if a.only >= 0 and i != a.only:
            continue
        x = torch.randn(a.batch, H, W, ci, device="cuda")
        w = torch.randn(co, ci, ks, ks, device="cuda") * 0.05
        wp, bias = capf.pack_conv(w)
        y = capf.conv_nhwc(x, wp, bias, ks, st, act=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            y = capf.conv_nhwc(x, wp, bias, ks, st, act=1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * co * ci * ks * ks
        byts = 4.0 * (x.numel() + y.numel() + wp.numel())
        print(f"[{i:2d}] {ci:3d}->{co:3d} k{ks} s{st} {H}x{W} B{a.batch}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.2f} TFLOP/s"
              f"  {byts / ms / 1e6:7.1f} GB/s(alg)")


if __name__ == "__main__":
    main()
