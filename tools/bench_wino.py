#!/usr/bin/env python
"""Direct implicit-GEMM vs Winograd F(2,3)-along-W on the 3x3 stride-1 shapes of the path (GPU box).
Usage: python tools/bench_wino.py [--batch 64]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
import torch.nn.functional as F
from capf import lib as capf


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    B = a.batch
    shapes = [(64, 64, 64, 64, "layer1 3x3"), (32, 32, 64, 64, "branch1"), (16, 16, 128, 128, "branch2"), (8, 8, 256, 256, "branch3"),
              (64, 64, 32, 32, "branch0")]
    g = torch.Generator().manual_seed(0)
    probs = []
    for H, W, ci, co, name in shapes:
        x = torch.randn(B, H, W, ci, generator=g).cuda()
        w = (torch.randn(co, ci, 3, 3, generator=g) / (9 * ci) ** 0.5).cuda()
        r = torch.randn(B, H, W, co, generator=g).cuda()
        wp, b = capf.pack_conv(w)
        ww, bw = capf.pack_conv_wino(w)
        w4, b4 = capf.pack_conv_wino(w, variant=43)
        want = F.conv2d(x.permute(0, 3, 1, 2), w, None, 1, 1).permute(0, 2, 3, 1) + r
        want = F.relu(want)
        yd = capf.conv_nhwc(x, wp, b, 3, 1, 1, r)
        yw = capf.conv_nhwc_wino(x, ww, bw, 1, r)
        y4 = capf.conv_nhwc_wino(x, w4, b4, 1, r)
        e4 = (y4 - want).abs().max().item()
        t4 = timeit(lambda: capf.conv_nhwc_wino(x, w4, b4, 1, r))
        ed, ew = (yd - want).abs().max().item(), (yw - want).abs().max().item()
        td = timeit(lambda: capf.conv_nhwc(x, wp, b, 3, 1, 1, r))
        tw = timeit(lambda: capf.conv_nhwc_wino(x, ww, bw, 1, r))
        gf = 2.0 * B * H * W * co * 9 * ci / 1e9
        print(f"{name:12s} {H}x{W} {ci}->{co}: direct {td:7.1f} us ({gf / td * 1e-3:6.1f} TF, err {ed:.1e})   wino {tw:7.1f} us "
              f"({gf / tw * 1e-3:6.1f} TF-equiv, err {ew:.1e})   x{td / tw:.2f}   F43 {t4:7.1f} us err {e4:.1e} x{td / t4:.2f}")
        probs.append((x, wp, b, ww, bw, r, w4, b4))
    # a 4-branch level, grouped: direct (all four) vs wino (branches 1-3) + direct (branch 0)
    lvl = probs[1:5]
    gd = lambda: capf.conv_nhwc_group([(q[0], q[1], q[2], 3, 1, 1, q[5]) for q in lvl])
    gw4 = lambda: capf.conv_nhwc_wino_group([(q[0], q[3], q[4], 1, q[5]) for q in lvl])
    g43 = lambda: capf.conv_nhwc_wino_group([(q[0], q[6], q[7], 1, q[5]) for q in lvl])
    print(f"4-branch level: direct group {timeit(gd):7.1f} us   F(2,3) group {timeit(gw4):7.1f} us   F(4,3) group {timeit(g43):7.1f} us")


if __name__ == "__main__":
    main()
