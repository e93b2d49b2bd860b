#!/usr/bin/env python
"""One lifter projection on the two-fp16-piece GEMM (csrc/igemm_f32h2.hip) for a counter pass: joint-block fc1 at batch 512 by default
(8704 x 640 -> 1280, the 128 x 64 tile with four waves along M).  (GPU box)   python tools/pmc_h2g.py [--m 8704 --k 640 --n 1280 --iters 5]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
from capf import lib as capf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=8704)
    ap.add_argument("--k", type=int, default=640)
    ap.add_argument("--n", type=int, default=1280)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    torch.manual_seed(0)
    x = torch.randn(a.m, a.k, device="cuda")
    w = torch.randn(a.n, a.k, device="cuda") / a.k ** 0.5
    b = torch.randn(a.n, device="cuda") * 0.1
    wp, _ = capf.pack_f32h2_gemm(w)
    for _ in range(2):
        capf.linear_f32h2g(x, wp, b, a.n, 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        capf.linear_f32h2g(x, wp, b, a.n, 2)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    fl = 2.0 * a.m * a.n * a.k
    tiles = ((a.m + 127) // 128) * ((a.n + 63) // 64)
    print(f"h2g linear {a.m} x {a.k} -> {a.n} (GELU): {us:8.1f} us per launch, {fl / us / 1e6:7.1f} TFLOP/s algorithmic, {3 * fl / us / 1e6:7.1f} executed; "
          f"{tiles} tiles of 128 x 64; MFMAs per launch = {3 * fl / 32768 / 1e6:.3f} M = {3 * fl / 32768 * 32 / 1e6:.1f} M SQ_VALU_MFMA_BUSY_CYCLES expected")


if __name__ == "__main__":
    main()
