#!/usr/bin/env python
"""Per-block timeline of one fp32 conv launch (diagnosis; needs a `make DIAG=1` build and CAPF_ABLATE=7, GPU box).
Usage: CAPF_ABLATE=7 python tools/timeline.py --shape 5 [--batch 64]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from capf import lib as capf
from bench_conv import SHAPES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--warm", type=int, default=300)
    ap.add_argument("--group", type=str, default="", help="comma list of HRNet branches (0-3): time ONE grouped launch of their 3x3 convs")
    a = ap.parse_args()
    assert os.environ.get("CAPF_ABLATE") == "7", "run with CAPF_ABLATE=7"
    if a.group:
        probs = []
        for b in [int(v) for v in a.group.split(",")]:
            c, r = [(32, 64), (64, 32), (128, 16), (256, 8)][b]
            x = torch.randn(a.batch, r, r, c, device="cuda")
            w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
            wp, bias = capf.pack_conv(w)
            probs.append((x, wp, bias, 3, 1, 1, None))
        launch = lambda: capf.conv_nhwc_group(probs)
        label = f"group of branches {a.group}"
    else:
        ci, co, ks, st, H, W = SHAPES[a.shape]
        x = torch.randn(a.batch, H, W, ci, device="cuda")
        w = torch.randn(co, ci, ks, ks, device="cuda") * 0.05
        wp, bias = capf.pack_conv(w)
        launch = lambda: capf.conv_nhwc(x, wp, bias, ks, st, act=1)
        label = f"shape {SHAPES[a.shape]}"
    for _ in range(a.warm):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    lib = capf.load_library()
    nb = 8192
    buf = np.zeros((nb, 8), dtype=np.uint64)
    rc = lib.capf_debug_timeline(buf.ctypes.data_as(ctypes.c_void_p), nb)
    assert rc == 0
    used = buf[:, 0] != 0
    t = buf[used].astype(np.int64)
    n = t.shape[0]
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    tot = t[:, 3] - t[:, 0]
    r0 = (t[:, 4] - t[:, 4].min()) * 10e-3          # us since first block start (100 MHz counter)
    dr = (t[:, 7] - t[:, 4]).astype(np.float64) * 10.0      # ns per block (100 MHz counter)
    ghz = float((tot / dr).mean())                          # measured shader clock while the kernel ran
    print(f"  measured shader clock {ghz * 1e3:.0f} MHz")
    print(f"{label} batch {a.batch}: event time {e0.elapsed_time(e1) * 1e3:.1f} us, {n} blocks recorded")
    for name, v in (("prologue", pro), ("k-loop", loop), ("epilogue", epi), ("block", tot)):
        us = v / ghz / 1e3
        print(f"  {name:9s} cycles: mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}"
              f"  p90 {np.percentile(v, 90):9.0f}   (~{us.mean():6.2f} us mean)")
    end = r0 + tot / ghz / 1e3
    print(f"  first block start 0.00 us, last block start {r0.max():.2f} us, last block end {end.max():.2f} us")
    hist, edges = np.histogram(r0, bins=12)
    print("  block starts per time bin:", " ".join(f"{edges[i]:.0f}us:{hist[i]}" for i in range(len(hist))))
    # occupancy curve: resident blocks (and blocks inside their K loop) averaged per time bin, chip-wide
    nb_bins = 20
    T = end.max()
    t1 = r0 + pro / ghz / 1e3
    t2 = t1 + loop / ghz / 1e3
    edges2 = np.linspace(0.0, T, nb_bins + 1)
    res, inloop = [], []
    for i in range(nb_bins):
        lo, hi = edges2[i], edges2[i + 1]
        res.append(np.clip(np.minimum(end, hi) - np.maximum(r0, lo), 0, None).sum() / (hi - lo))
        inloop.append(np.clip(np.minimum(t2, hi) - np.maximum(t1, lo), 0, None).sum() / (hi - lo))
    print("  resident blocks per bin :", " ".join(f"{v:5.0f}" for v in res))
    print("  blocks in their K loop  :", " ".join(f"{v:5.0f}" for v in inloop))
    # concurrency: how many blocks are alive per CU (hw id = se/sh/cu + xcc)
    cu = (t[:, 5] >> 8) & 0xFF
    key = (t[:, 6] & 0xF) * 256 + cu
    ids, counts = np.unique(key, return_counts=True)
    print(f"  distinct (xcc, cu) = {len(ids)}, blocks per CU min {counts.min()} max {counts.max()}")
    k0 = ids[0]
    sel = np.where(key == k0)[0]
    order = sel[np.argsort(r0[sel])]
    print(f"  timeline of CU key {k0}:")
    for i in order[:40]:
        print(f"    block start {r0[i]:7.2f} us  prologue {pro[i] / ghz / 1e3:6.2f}  loop {loop[i] / ghz / 1e3:6.2f}  epilogue {epi[i] / ghz / 1e3:6.2f}"
              f"  end {end[i]:7.2f}")


if __name__ == "__main__":
    main()
