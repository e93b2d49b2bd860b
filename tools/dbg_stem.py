"""debug: bf16 stem conv vs op oracle, error map"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("contextaware-poseformer_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import op_oracle
from capf import synth
from test_gpu_fullsize import _model
backbone, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
model, sd = _model(backbone, "bf16", 71)
img, k2d, kc = synth.synth_inputs(B, H, W, seed=72, crop_range=(W, H))
img_d = img.cuda()
eng = model.engine_for(img_d)
names = [n for n, _, _ in eng.schema()]
d = eng.op_describe(0)
eng.forward_prefix(img_d, d.checkpoint, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
got = eng.op_tensor(0, 5, (B, d.Ho, d.Wo, d.Cout), d.out_dtype).cpu().float()
x = eng.op_tensor(0, 0, (B, d.H, d.W, d.Cin), d.in_dtype).cpu()
conv = names[d.p_weight][:-len(".weight")]; bn = names[d.p_bn_weight][:-len(".weight")]
with torch.no_grad():
    want, mass, term = op_oracle.conv_bn_act(sd, conv, bn, x, None, d.ks, d.stride, d.pad, d.act, bool(d.mfma_bf16))
want = want.float()
err = (got - want).abs()
print(eng.op_table(B)[0], "max err", err.max().item(), "frac bad", (err > 0.05 * want.abs().clamp(min=1)).float().mean().item())
bad = (err > 0.05 * want.abs().clamp(min=1))
print("bad per batch:", bad.flatten(1).float().mean(1)[:8].tolist())
print("bad per channel (first 16):", bad.permute(3, 0, 1, 2).flatten(1).float().mean(1)[:16].tolist())
b0 = bad[0].any(-1)
print("bad rows of frame 0 (ho):", b0.any(1).nonzero().flatten()[:40].tolist())
print("bad cols of frame 0 (wo):", b0.any(0).nonzero().flatten()[:40].tolist())
m = bad.flatten(0, 2).any(-1).nonzero().flatten()
print("first bad flat pixels:", m[:32].tolist(), "count", m.numel(), "of", bad.flatten(0, 2).shape[0])
