#!/usr/bin/env python
"""Per-block timeline of one bf16 conv launch (diagnosis build: CAPF_LIB=tools/ab/libcapf_diag.so CAPF_BF16_PP=1, GPU box)."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import numpy as np
import torch
from capf import lib as capf

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=64)
ap.add_argument("--ch", type=int, default=48)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--ks", type=int, default=3)
ap.add_argument("--cout", type=int, default=0, help="output channels (default: --ch)")
ap.add_argument("--width", type=int, default=0, help="image width (default: --res)")
ap.add_argument("--rh", action="store_true", help="time the row-halo kernel instead (no timeline)")
a = ap.parse_args()
co, wd = a.cout or a.ch, a.width or a.res
x = torch.randn(a.batch, a.res, wd, a.ch, device="cuda").bfloat16()
w = torch.randn(co, a.ch, a.ks, a.ks, device="cuda") * 0.05
r = torch.randn(a.batch, a.res, wd, co, device="cuda").bfloat16()
if a.rh:
    ww, bw, cw = capf.pack_conv_bf16_rh(w)
    for _ in range(10):
        capf.conv_nhwc_bf16_rh(x, ww, bw, 1, r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        capf.conv_nhwc_bf16_rh(x, ww, bw, 1, r)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    flops = 2.0 * a.batch * a.res * wd * a.ch * co * 9
    print(f"row-halo conv {a.batch}x{a.res}x{a.res}x{a.ch} cw{cw}: {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s")
    if not hasattr(capf.load_library(), "capf_debug_bf16_timeline"):
        sys.exit(0)
ww, bw = capf.pack_conv_bf16(w)
for _ in range(0 if a.rh else 10):
    capf.conv_nhwc_bf16(x, ww, bw, a.ks, 1, 1, r)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(0 if a.rh else 10):
    capf.conv_nhwc_bf16(x, ww, bw, a.ks, 1, 1, r)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100.0
flops = 2.0 * a.batch * a.res * wd * a.ch * co * a.ks * a.ks
hbm = 2.0 * a.batch * a.res * wd * (a.ch + 2 * co)
if not a.rh:
    print(f"conv {a.batch}x{a.res}x{wd}x{a.ch}->{co} ks{a.ks}: {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s  {hbm / us / 1e6:.2f} TB/s (in + residual + out)")
lib = capf.load_library()
nb = 8192
buf = np.zeros((nb, 8), dtype=np.uint64)
assert lib.capf_debug_bf16_timeline(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
t = buf[buf[:, 0] != 0].astype(np.int64)
pro, loop, epi, drain, tot = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 4] - t[:, 0]
rt = (t[:, 7].max() - t[:, 5].min()) / 100.0
tick = tot.sum() / max(1, (t[:, 7] - t[:, 5]).sum()) * 100.0
print(f"{t.shape[0]} blocks, span of the first 8192 blocks {rt:.1f} us, memtime ~{tick:.0f} ticks/us")
for name, v in (("prologue", pro), ("K loop", loop), ("  of it load wait", t[:, 6]), ("epilogue", epi), ("store drain", drain), ("total", tot)):
    print(f"  {name:18s} mean {v.mean() / tick:7.2f} us   p10 {np.percentile(v, 10) / tick:7.2f}   p90 {np.percentile(v, 90) / tick:7.2f}")
print(f"  concurrency: {tot.sum() / tick / (rt * 256):.2f} blocks per CU on average")
