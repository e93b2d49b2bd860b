#!/usr/bin/env python
"""Per-block timeline of ONE grouped F(4,3) level (the four HRNet-32 branches at batch 64) — diagnosis build only:
    CAPF_LIB=tools/ab/libcapf_diag.so python tools/wino_level_timeline.py [--batch 64]
Per problem: when its blocks start and end inside the launch, their prologue / K loop / epilogue, the K loop per superchunk
against its MFMA time, and the residency over time (blocks alive per CU in 10 us buckets)."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import numpy as np
import torch
from capf import lib as capf

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--base", type=int, default=32)
a = ap.parse_args()
g = torch.Generator().manual_seed(0)
probs = []
for i in range(4):
    c, r = a.base << i, 64 >> i
    x = torch.randn(a.batch, r, r, c, generator=g).cuda()
    w = (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5).cuda()
    res = torch.randn(a.batch, r, r, c, generator=g).cuda()
    wp, b = capf.pack_conv_wino(w, variant=43)
    probs.append((x, wp, b, 1, res))
for _ in range(10):
    capf.conv_nhwc_wino_group(probs)
torch.cuda.synchronize()
lib = capf.load_library()
nb = 8192
buf = np.zeros((nb, 8), dtype=np.uint64)
assert lib.capf_debug_wino_timeline(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
t = buf[buf[:, 0] != 0].astype(np.int64)
r0 = t[:, 4].min()
span = (t[:, 7].max() - r0) / 100.0
tick = (t[:, 3] - t[:, 0]).sum() / max(1, (t[:, 7] - t[:, 4]).sum()) * 100.0          # memtime ticks per us
print(f"{t.shape[0]} blocks, launch span {span:.1f} us, memtime ~{tick:.0f} ticks/us (= shader MHz)")
for cin in sorted(set(t[:, 5].tolist()), reverse=True):
    q = t[t[:, 5] == cin]
    nsc = int(q[0, 6])
    st, en = (q[:, 4] - r0) / 100.0, (q[:, 7] - r0) / 100.0
    pro, loop, epi = (q[:, 1] - q[:, 0]) / tick, (q[:, 2] - q[:, 1]) / tick, (q[:, 3] - q[:, 2]) / tick
    print(f"  Cin {cin:4d}: {q.shape[0]:5d} blocks x {nsc:2d} superchunks | start {st.min():6.1f}..{st.max():6.1f} us, end {en.min():6.1f}..{en.max():6.1f} | "
          f"prologue {pro.mean():5.2f}  K loop {loop.mean():6.2f} ({loop.mean() / nsc:5.2f} per superchunk; 24 MFMAs = {24 * 64 / tick:4.2f} us)  epilogue {epi.mean():5.2f} us")
ph = np.zeros((nb, 8), dtype=np.uint64)
if hasattr(lib, "capf_debug_wino_phases") and lib.capf_debug_wino_phases(ph.ctypes.data_as(ctypes.c_void_p), nb) == 0:
    ph = ph[buf[:, 0] != 0].astype(np.float64)
    names = ["compute", "drain+barrier", "DMA issue", "data wait", "barrier", "first frags"]
    for cin in sorted(set(t[:, 5].tolist()), reverse=True):
        m = t[:, 5] == cin
        nsc = float(t[m][0, 6])
        per = ph[m].mean(axis=0) / nsc
        print(f"  Cin {cin:4d} cycles per superchunk (wave 0): " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, per[:6])) + f"   sum {per[:6].sum():6.0f}")
edges = np.arange(0, span + 10, 10)
alive = [(((t[:, 4] - r0) / 100.0 < hi) & ((t[:, 7] - r0) / 100.0 > lo)).sum() / 256.0 for lo, hi in zip(edges[:-1], edges[1:])]
print("  blocks alive per CU in 10-us buckets: " + " ".join(f"{v:.2f}" for v in alive))
busy = float((t[:, 6] * 24 * 64 * 4).sum())                                               # MFMA cycles x SIMDs
print(f"  MFMA cycles needed: {busy / 1e6:.1f} M SIMD-cycles = {busy / 1024 / tick:.1f} us of the launch at this clock ({busy / 1024 / tick / span:.2f} of the span)")
