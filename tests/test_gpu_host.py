"""GPU: the host mirror's contract around the C ABI — parameter-change tracking (storage moved by
capf.optim.flatten_, child-level load_state_dict, several engines), the one-set-of-activations rule of the native
training step, and argument validation (reference: ContextPose/mvn/models/conpose.py:30-42, train.py:147-148,
294-312)."""
import numpy as np
import pytest
import torch

from conftest import make_model
from capf import synth
from capf.lib import CapfError

pytestmark = pytest.mark.gpu


def _inputs(B=2, H=256, W=192, seed=3):
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=seed)
    return img.cuda(), k2d.cuda(), kc.cuda()


def test_flatten_after_first_forward_rebinds_and_optimizer_updates_reach_the_kernels():
    from capf.optim import FusedAdamW, flatten_
    model, _ = make_model("hrnet_32", device="cuda", wseed=3)
    img, k2d, kc = _inputs()
    with torch.no_grad():
        before = model(img, k2d, kc.clone())
    eng = model.engine_for(img)
    n0 = eng.rebinds
    flat = flatten_(model.volume_net)              # moves every lifter parameter's storage AFTER the first forward
    with torch.no_grad():
        same = model(img, k2d, kc.clone())
    assert eng.rebinds == n0 + 1                   # the engine noticed the new pointers and re-borrowed them
    assert torch.equal(before, same)
    # a fused optimizer step writes the flat buffer; the next forward must equal a FRESH model holding those values
    opt = FusedAdamW(flat, lr=1e-2, weight_decay=0.1)
    opt.step(torch.randn_like(flat) * 1e-2)
    model.lifter_params_changed()
    with torch.no_grad():
        after = model(img, k2d, kc.clone())
    fresh, _ = make_model("hrnet_32", device="cuda", wseed=3)
    fresh.load_state_dict(model.state_dict())
    with torch.no_grad():
        want = fresh(img, k2d, kc.clone())
    assert torch.equal(after, want)
    assert (after - before).abs().max().item() > 1e-4


def test_backbone_load_state_dict_reaches_every_engine():
    """train.py:294-297 loads the backbone through `model.backbone.load_state_dict` — possibly after engines for two
    input sizes exist.  Both must refold their BatchNorms."""
    model, _ = make_model("hrnet_32", device="cuda", wseed=4)
    a, b = _inputs(1, 256, 192), _inputs(1, 128, 96)
    with torch.no_grad():
        model(a[0], a[1], a[2].clone()); model(b[0], b[1], b[2].clone())
    other, sd_other = make_model("hrnet_32", device="cuda", wseed=99)
    model.backbone.load_state_dict(other.backbone.state_dict())
    model.volume_net.load_state_dict(other.volume_net.state_dict())
    with torch.no_grad():
        for x in (a, b):
            got = model(x[0], x[1], x[2].clone())
            want = other(x[0], x[1], x[2].clone())
            assert torch.equal(got, want)


def test_backward_after_a_later_forward_fails_loudly():
    from mvn.models.loss import MPJPE
    model, _ = make_model("hrnet_32", device="cuda", wseed=5)
    model.train(); model.backbone.eval(); model.drop_path_rate = 0.0
    img, k2d, kc = _inputs()
    _, _, _, gt = synth.synth_inputs(2, 256, 192, seed=3, with_gt=True)
    l1 = MPJPE()(model(img, k2d, kc.clone()), gt.cuda())
    l2 = MPJPE()(model(img, k2d, kc.clone()), gt.cuda())
    with pytest.raises(CapfError, match="overwritten"):
        (l1 + l2).backward()
    # the ABI itself refuses as well: an eval forward in between invalidates the saved activations
    pred = model(img, k2d, kc.clone())
    with torch.no_grad():
        model(img, k2d, kc.clone())
    eng = model.engine_for(img)
    flat = torch.empty(eng.grad_layout_cached()[1], device="cuda")
    with pytest.raises(CapfError, match="activations"):
        eng.backward(torch.ones_like(pred), flat, torch.cuda.current_stream().cuda_stream)


def test_argument_validation_and_noncontiguous_crop_keypoints():
    model, sd = make_model("hrnet_32", device="cuda", wseed=6)
    img, k2d, kc = _inputs()
    with pytest.raises(TypeError):
        model(img, k2d.double(), kc.clone())
    with pytest.raises(ValueError):
        model(img, k2d, kc.cpu())
    with pytest.raises(ValueError):
        model(img, k2d, torch.zeros(2, 17, 3, device="cuda"))
    # a non-contiguous third argument is accepted and still normalised IN PLACE, like the reference
    wide = torch.zeros(2, 17, 4, device="cuda")
    wide[..., :2] = kc
    view = wide[..., :2]
    assert not view.is_contiguous()
    with torch.no_grad():
        got = model(img, k2d, view)
        want_kc = kc.clone()
        want = model(img, k2d, want_kc)
    assert torch.equal(got, want) and torch.equal(wide[..., :2], want_kc) and (wide[..., 2:] == 0).all()
    eng = model.engine_for(img)
    assert 1000 < eng.max_batch() <= 8192
