#!/usr/bin/env python
"""Experiment: the layer1 pointwise convs (pose_hrnet.py:98-136) on the fp32 pointwise kernel (igemm_f32_pw.hip, what the plan runs)
vs the two-fp16-piece GEMM (igemm_f32h2.hip).  (GPU box)   python tools/bench_pw_h2.py --batch 64"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
from capf import lib as capf


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    torch.manual_seed(0)
    B = a.batch
    for ci, co, res in [(64, 256, True), (256, 64, False), (64, 64, False), (64, 256, False)]:
        x = torch.randn(B, 64, 64, ci, device="cuda")
        r = torch.randn(B, 64, 64, co, device="cuda") if res else None
        w = torch.randn(co, ci, 1, 1, device="cuda") / ci ** 0.5
        wd, bd = capf.pack_conv(w)
        wp, bias = capf.pack_f32h2_gemm(w)
        t32 = timeit(lambda: capf.conv_nhwc(x, wd, bd, 1, 1, 1, r))
        th2 = timeit(lambda: capf.conv_nhwc_f32h2g(x, wp, bias, 1, 1, 1, r, co))
        gb = (x.numel() + (r.numel() if res else 0) + B * 4096 * co) * 4 / 1e9
        print(f"batch {B} {ci:3d} -> {co:3d} res {int(res)}: fp32 pw {t32:8.1f} us ({gb / t32 * 1e3:5.2f} TB/s)   two-piece GEMM {th2:8.1f} us ({gb / th2 * 1e3:5.2f} TB/s)")


if __name__ == "__main__":
    main()
