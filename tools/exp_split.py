"""EXPERIMENT: one batch-B step issued as S sub-batches of B / S frames on S HIP streams (fork / join inside the step), against
the single-stream step.  python tools/exp_split.py --config 1 --split 2"""
import argparse, contextlib, copy, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd")); sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--split", type=int, default=2)
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
S = x.split
subs = []
for s in range(S):
    m = make(); m.load_state_dict(sd); subs.append((m.to(dev), torch.cuda.Stream(dev)))
n = B // S
main = torch.cuda.current_stream(dev)

def whole():
    return m0(img, k2d, kc.clone())

def split():
    outs = []
    ev = torch.cuda.Event(); ev.record(main)
    for s, (m, st) in enumerate(subs):
        st.wait_event(ev)
        with torch.cuda.stream(st):
            outs.append(m(img[s * n:(s + 1) * n], k2d[s * n:(s + 1) * n], kc[s * n:(s + 1) * n].clone()))
        e = torch.cuda.Event(); e.record(st); main.wait_event(e)
    return torch.cat(outs)

with torch.no_grad():
    for name, f in (("whole", whole), ("split", split), ("whole", whole), ("split", split)):
        for _ in range(3): o = f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(x.steps): o = f()
        torch.cuda.synchronize(); el = time.perf_counter() - t
        print(f"cfg{x.config} {name} S={S}: {B * x.steps / el:9.1f} frames/s  {el / x.steps * 1e3:.3f} ms/step", flush=True)
    ref = whole(); got = split(); torch.cuda.synchronize()
    print("max |split - whole| =", (ref - got).abs().max().item())
