#!/usr/bin/env python
"""ONE grouped conv level of the product schedule, launched on its own (PMC collection / timing on the GPU box):
    --kind wino43   the four HRNet-32 branches (32@64^2, 64@32^2, 128@16^2, 256@8^2) at batch 64, fp32 Winograd F(4,3): what
                    cfg1's dominant kernel igemm_wino_group_kernel runs 64 times per forward
    --kind x3       the same four branches on the split-fp32 tile (igemm_f32x3_ws.hip: fp32 in / out, six bf16 MFMAs per fp32 one's worth of k)
    --kind bf16rh   the four HRNet-48 branches (48@64^2, 96@32^2, 192@16^2, 384@8^2) at batch 256, bf16 row-halo tiles: cfg2's
                    dominant kernel igemm_bf16_group_rh_kernel
Prints the launch time by HIP events, the executed / algorithmic FLOPs and the MFMA cycles the launch needs."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
from capf import lib as capf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="wino43", choices=["wino43", "wino23", "bf16rh", "bf16ws", "x3", "h2"])
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--branches", default="0,1,2,3")
    a = ap.parse_args()
    bf = a.kind in ("bf16rh", "bf16ws")
    ws = a.kind == "bf16ws"
    B = a.batch or (256 if bf else 64)
    base = 48 if bf else 32
    sel = [int(v) for v in a.branches.split(",")]
    g = torch.Generator().manual_seed(0)
    probs, alg = [], 0.0
    for i in sel:
        c, r = base << i, 64 >> i
        x = torch.randn(B, r, r, c, generator=g).cuda()
        w = (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5).cuda()
        res = torch.randn(B, r, r, c, generator=g).cuda()
        alg += 2.0 * B * r * r * c * c * 9
        if a.kind == "x3":
            wp, b = capf.pack_conv_f32x3(w)
            probs.append((x, wp, b, 1, res, c))
        elif a.kind == "h2":
            wp, b = capf.pack_conv_f32h2(w)
            probs.append((x, wp, b, 1, res, c))
        elif ws:
            wp, b = capf.pack_conv_bf16_ws(w)
            probs.append((x.bfloat16(), wp, b, 1, res.bfloat16(), c))
        elif bf:
            wp, b = capf.pack_conv_bf16(w)
            wrh, _, _ = capf.pack_conv_bf16_rh(w)
            probs.append((x.bfloat16(), wp, b, 3, 1, 1, res.bfloat16(), wrh))
        else:
            wp, b = capf.pack_conv_wino(w, variant=43 if a.kind == "wino43" else 23)
            probs.append((x, wp, b, 1, res))

    def launch():
        if a.kind == "x3":
            capf.conv_nhwc_f32x3_group(probs)
            return 4
        if a.kind == "h2":
            capf.conv_nhwc_f32h2_group(probs)
            return 5
        if ws:
            capf.conv_nhwc_bf16_ws_group(probs)
            return 3
        if bf:
            return capf.conv_nhwc_bf16_group(probs)[1]
        capf.conv_nhwc_wino_group(probs)
        return -1

    variant = launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    frac = {"wino43": 0.5, "wino23": 2.0 / 3.0, "bf16rh": 1.0, "bf16ws": 1.0, "x3": 6.0, "h2": 3.0}[a.kind]     # x3 / h2: six bf16 / three fp16 products per fp32 product
    ex = alg * frac
    x3 = a.kind in ("x3", "h2")
    # v_mfma_f32_32x32x2_f32: 4096 FLOP, 64 cycles / SIMD;  v_mfma_f32_32x32x16_bf16: 32768 FLOP, 32 cycles / SIMD
    n_mfma = ex / (32768.0 if (bf or x3) else 4096.0)
    busy = n_mfma * (32 if (bf or x3) else 64)
    peak = 2500.0 if (bf or x3) else 157.3
    print(f"{a.kind} batch {B} branches {sel}: {us:8.1f} us per grouped launch (variant {variant});  algorithmic {alg / 1e9:.2f} GFLOP = "
          f"{alg / us / 1e6:7.1f} TFLOP/s ({alg / us / 1e6 / (peak / frac if x3 else peak):.3f} of peak),  executed {ex / 1e9:.2f} GFLOP = {ex / us / 1e6:7.1f} TFLOP/s "
          f"({ex / us / 1e6 / peak:.3f});  {n_mfma / 1e6:.3f} M MFMAs = {busy / 1e6:.1f} M SQ_VALU_MFMA_BUSY_CYCLES expected")


if __name__ == "__main__":
    main()
