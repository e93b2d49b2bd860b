#!/usr/bin/env python
"""Experiment: does running the batch as P independent sub-batches on P streams (each with its own engine and
workspace) fill the per-kernel ramp/tail bubbles better than one big launch sequence?  (GPU box)
Usage: python tools/bench_split.py --batch 64 --parts 2 [--lanes 1]"""
import argparse
import contextlib
import copy
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
from capf import synth
from mvn.models.conpose import CA_PF
from mvn.utils.cfg import backbone_preset, config


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="fp32")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    cfg.model.backbone.fix_weights = True
    models, inputs, streams = [], [], []
    b = a.batch // a.parts
    for p in range(a.parts):
        with contextlib.redirect_stdout(io.StringIO()):
            m = CA_PF(cfg, compute_dtype=a.dtype).eval()
        synth.load_synthetic(m, seed=1, bn_mode="random")
        m = m.to(dev)
        img, k2d, kc = synth.synth_inputs(b, 256, 256, seed=1000 + p, crop_range=(192, 256))
        img, k2d, kc = img.to(dev), k2d.to(dev), kc.to(dev)
        m.engine_for(img).set_lanes(a.lanes)
        models.append(m)
        inputs.append((img, k2d, kc, kc.clone()))
        streams.append(torch.cuda.Stream(dev))

    def step():
        outs = []
        for m, (img, k2d, kc0, kcw), s in zip(models, inputs, streams):
            with torch.cuda.stream(s):
                kcw.copy_(kc0)
                outs.append(m(img, k2d, kcw))
        return outs

    with torch.no_grad():
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(f"batch {a.batch} as {a.parts} x {b} on {a.parts} streams, lanes {a.lanes}: {el / a.steps * 1e3:.3f} ms/step  "
          f"{a.batch * a.steps / el:.1f} frames/s")


if __name__ == "__main__":
    main()
