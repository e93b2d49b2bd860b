#!/usr/bin/env python
"""Per-block timeline of one Winograd conv launch (diagnosis build: CAPF_LIB=tools/ab/libcapf_diag.so, GPU box)."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import numpy as np
import torch
from capf import lib as capf

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=64)
ap.add_argument("--ch", type=int, default=64)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--variant", type=int, default=23)
a = ap.parse_args()
x = torch.randn(a.batch, a.res, a.res, a.ch, device="cuda")
w = torch.randn(a.ch, a.ch, 3, 3, device="cuda") * 0.05
r = torch.randn(a.batch, a.res, a.res, a.ch, device="cuda")
ww, bw = capf.pack_conv_wino(w, variant=a.variant)
for _ in range(20):
    capf.conv_nhwc_wino(x, ww, bw, 1, r)
torch.cuda.synchronize()
lib = capf.load_library()
nb = 8192
buf = np.zeros((nb, 8), dtype=np.uint64)
assert lib.capf_debug_wino_timeline(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
t = buf[buf[:, 0] != 0].astype(np.int64)
pro, loop, epi, tot = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
rt = (t[:, 7].max() - t[:, 4].min()) / 100.0    # 100 MHz realtime counter -> us
tick = (t[:, 3] - t[:, 0]).sum() / max(1, (t[:, 7] - t[:, 4]).sum()) * 100.0   # memtime ticks per us
print(f"{t.shape[0]} blocks, launch span {rt:.1f} us, memtime ~{tick:.0f} ticks/us")
for name, v in (("prologue", pro), ("K loop", loop), ("epilogue", epi), ("total", tot)):
    print(f"  {name:9s} mean {v.mean() / tick:7.2f} us   p10 {np.percentile(v, 10) / tick:7.2f}   p90 {np.percentile(v, 90) / tick:7.2f}")
nsc = 3 * a.ch // 32
mf = 4096 if a.variant == 23 else 3072
print(f"  K loop per superchunk: {loop.mean() / tick / nsc:.2f} us (MFMA bound 1.71 us alone, 3.41 us when two blocks share the CU)")
cu = t[:, 5] & 0xFFFFF0F0   # crude CU id (drop wave / simd bits)
print(f"  concurrency: sum of block times / (launch span * 256 CUs) = {tot.sum() / tick / (rt * 256):.2f} blocks per CU on average")

import collections
print("  LDS_ALLOC values (hex -> blocks):", {hex(int(k)): v for k, v in collections.Counter(t[:, 5].tolist()).most_common(6)})
