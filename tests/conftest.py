import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "contextaware-poseformer_amd"), os.path.join(ROOT, "oracle"),
          os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Build libcapf.so in-tree if it is not there (hipcc cross-compiles without a GPU)."""
    so = os.path.join(ROOT, "contextaware-poseformer_amd", "capf", "libcapf.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    return so


def make_model(backbone, device=None, wseed=0, bn="random", compute_dtype=None, plan_flags=0):
    """Host CA_PF with a synthetic checkpoint; returns (model, state_dict on CPU)."""
    import copy
    from capf import synth
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    import contextlib, io
    cfg = backbone_preset(copy.deepcopy(config), backbone)
    cfg.model.backbone.fix_weights = True
    with contextlib.redirect_stdout(io.StringIO()):
        model = (CA_PF(cfg) if compute_dtype is None and not plan_flags else CA_PF(cfg, compute_dtype=compute_dtype or "fp32", plan_flags=plan_flags)).eval()
    sd = synth.load_synthetic(model, seed=wseed, bn_mode=bn)
    if device is not None:
        model = model.to(device)
    return model, sd


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
