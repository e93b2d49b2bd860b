"""EXPERIMENT (see exp_split.py): where does the split step's time go?  Host issue time per forward, a half batch alone, two half
batches on one stream, on two streams with and without the per-step fork / join."""
import argparse, contextlib, copy, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd")); sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--steps", type=int, default=30)
x = ap.parse_args()
a = bench.parse(["--config", str(x.config)])
from capf import synth
from mvn.models.conpose import CA_PF
from mvn.utils.cfg import backbone_preset, config
dev = torch.device("cuda", 0)
cfg = backbone_preset(copy.deepcopy(config), a.backbone)
cfg.model.backbone.fix_weights = True
cfg.model.poseformer.embed_dim_ratio = a.embed
dt = "bf16" if a.dtype == "bf16" else "fp32"
def make():
    with contextlib.redirect_stdout(io.StringIO()):
        return CA_PF(cfg, compute_dtype=dt).eval()
m0 = make(); sd = synth.load_synthetic(m0, seed=1, bn_mode="random"); m0 = m0.to(dev)
B, H, W = a.batch, a.height, a.width
img, k2d, kc = synth.synth_inputs(B, H, W, seed=1000, crop_range=(192, 256))
img, k2d, kc = img.to(dev), k2d.to(dev), kc.to(dev)
n = B // 2
subs = []
for s in range(2):
    m = make(); m.load_state_dict(sd); subs.append((m.to(dev), torch.cuda.Stream(dev)))
halves = [(img[s * n:(s + 1) * n].contiguous(), k2d[s * n:(s + 1) * n].contiguous(), kc[s * n:(s + 1) * n].contiguous()) for s in range(2)]
main = torch.cuda.current_stream(dev)

def whole(): m0(img, k2d, kc.clone())
def half_alone(): subs[0][0](halves[0][0], halves[0][1], halves[0][2].clone())
def two_one_stream():
    for s in range(2): subs[s][0](halves[s][0], halves[s][1], halves[s][2].clone())
def two_streams_free():
    for s in range(2):
        with torch.cuda.stream(subs[s][1]): subs[s][0](halves[s][0], halves[s][1], halves[s][2].clone())
def two_streams_joined():
    ev = torch.cuda.Event(); ev.record(main)
    for s in range(2):
        st = subs[s][1]; st.wait_event(ev)
        with torch.cuda.stream(st): subs[s][0](halves[s][0], halves[s][1], halves[s][2].clone())
        e = torch.cuda.Event(); e.record(st); main.wait_event(e)

with torch.no_grad():
    for name, f, frames in (("whole", whole, B), ("half alone", half_alone, n), ("two halves, one stream", two_one_stream, B),
                            ("two halves, two streams, free-running", two_streams_free, B),
                            ("two halves, two streams, fork/join per step", two_streams_joined, B), ("whole", whole, B)):
        for _ in range(3): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(x.steps): f()
        host = time.perf_counter() - t
        torch.cuda.synchronize(); el = time.perf_counter() - t
        print(f"cfg{x.config} {name:46s}: {frames * x.steps / el:9.1f} frames/s  {el / x.steps * 1e3:7.3f} ms/step   host issue {host / x.steps * 1e3:7.3f} ms/step", flush=True)
