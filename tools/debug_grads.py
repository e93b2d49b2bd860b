import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("contextaware-poseformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from conftest import load_golden
import test_gpu_train as T
g = load_golden("w32_256x256_b2")
model, pred, loss = T._train_step("w32_256x256_b2")
named = dict(model.named_parameters())
names = [str(n) for n in g["gradnorm_names"]]
for n, want in zip(names, g["gradnorms"]):
    got = named[n].grad.double().norm().item()
    rel = abs(got - want) / max(1e-9, want)
    flag = "  <<<<" if rel > 2e-3 else ""
    print(f"{n:62s} ref {want:.4e} got {got:.4e} rel {rel:.2e}{flag}")
