#!/usr/bin/env python
"""EXPERIMENT: does replaying the batch-1 forward as a HIP graph beat enqueueing its 169 launches?  (gpurun -- python tools/graph_b1.py [batch])"""
import copy, contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))
import torch
from capf import synth
from mvn.models.conpose import CA_PF
from mvn.utils.cfg import backbone_preset, config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
cfg.model.backbone.fix_weights = True
with contextlib.redirect_stdout(io.StringIO()):
    model = CA_PF(cfg).eval()
synth.load_synthetic(model, seed=1, bn_mode="random")
model = model.cuda()
img, k2d, kc = synth.synth_inputs(B, 256, 256, seed=2)
img, k2d, kc0 = img.cuda(), k2d.cuda(), kc.cuda()
kc = kc0.clone()

def step():
    kc.copy_(kc0)
    return model(img, k2d, kc)

with torch.no_grad():
    for _ in range(5):
        ref = step().clone()
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / N * 1e3
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / N * 1e3
    print(f"batch {B}: enqueued {eager:.3f} ms per forward, graph replay {graph:.3f} ms; outputs equal: {torch.equal(out, ref)}")
