"""GPU: parity at the FULL sizes of BASELINE.json's configurations.  Tile shapes, grouped-launch configurations and
split-K factors depend on the batch (csrc/igemm_f32.hip pick_tile / launch_gemm_f32_group, csrc/train.cpp), so the
B<=5 cases of test_gpu_parity.py do not exercise what the benchmark runs.  The CPU oracle does ~25 frames/s on the
GPU box's host, so fp32 comparisons use the whole batch where that costs seconds and a slice otherwise; every case
also checks batch independence: frame b of the big batch equals the same frame run on its own.
Each test prints max-abs and mean per-joint distance (the 'MPJPE vs ref' of BASELINE.json's metric)."""
import copy
import contextlib
import io

import numpy as np
import pytest
import torch

import capf_oracle as oracle
from capf import synth
from conftest import load_golden, make_model
from golden_cases import CASES, case_inputs

pytestmark = pytest.mark.gpu


def _report(tag, got, want):
    err = (got - want).abs().max().item()
    mpj = (got - want).norm(dim=-1).mean().item()
    print(f"{tag}: max|hip - oracle| {err:.3e}   mean per-joint distance {mpj:.3e}")
    return err, mpj


def _model(backbone, dtype, wseed):
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), backbone)
    cfg.model.backbone.fix_weights = True
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg, compute_dtype=dtype).eval()
    sd = synth.load_synthetic(model, seed=wseed, bn_mode="random")
    return model.cuda(), sd


def test_cfg1_batch64_hrnet32_fp32_whole_batch_vs_oracle():
    """configs[1]: B=64 HRNet-32 256x256 fp32 — all 64 frames against the CPU oracle, tolerance 1e-3 (north_star)."""
    B = 64
    model, sd = _model("hrnet_32", "fp32", 41)
    img, k2d, kc = synth.synth_inputs(B, 256, 256, seed=42, crop_range=(256, 256))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        want = oracle.ca_pf_forward(sd, img, k2d, kc.clone(), backbone="hrnet_32")
        kc_dev = kc.cuda()
        got = model(img.cuda(), k2d.cuda(), kc_dev).cpu()
        one = model(img[37:38].cuda(), k2d[37:38].cuda(), kc[37:38].clone().cuda()).cpu()
    err, mpj = _report("cfg1 B=64 W32 256x256 fp32", got, want)
    assert err <= 1e-3
    assert (got[37:38] - one).abs().max().item() <= 2e-6        # different tiles at B=1 -> accumulation order only
    ref = kc.clone(); oracle.normalise_crop_keypoints_(ref)
    assert torch.equal(kc_dev.cpu(), ref)                        # in-place ref, bit exact at full size too


def test_cfg2_batch256_hrnet48_bf16_slice_vs_fp32_oracle():
    """configs[2]: B=256 HRNet-48 256x256 bf16 — 16 frames spread over the batch against the fp32 oracle.  Bound = 2x the
    error measured on this build (bf16 operands, fp32 accumulation, ~300 layers); batch independence is bitwise in the
    backbone (same tiles at every batch for a frame? no: tile choice depends on batch) so it is bounded, not bitwise."""
    B = 256
    model, sd = _model("hrnet_48", "bf16", 43)
    img, k2d, kc = synth.synth_inputs(B, 256, 256, seed=44, crop_range=(256, 256))
    pick = list(range(0, B, 16))
    with torch.no_grad():
        want = oracle.ca_pf_forward(sd, img[pick], k2d[pick], kc[pick].clone(), backbone="hrnet_48")
        got = model(img.cuda(), k2d.cuda(), kc.clone().cuda()).cpu()
        sub = model(img[pick].cuda(), k2d[pick].cuda(), kc[pick].clone().cuda()).cpu()
    err, mpj = _report("cfg2 B=256 W48 256x256 bf16 (16-frame slice)", got[pick], want)
    assert err <= BF16_W48_MAX and mpj <= BF16_W48_MEAN
    # bf16 rounding of identical fp32 sums is identical: a frame's result may only move by accumulation-order
    # effects that cross a bf16 rounding boundary somewhere in ~300 layers
    d = (got[pick] - sub).abs().max().item()
    print(f"  batch independence (B=256 vs B=16): max delta {d:.3e}")
    assert d <= BF16_W48_MAX


def test_cfg4_batch128_cpn_384x288_bf16_slice_vs_fp32_oracle():
    """configs[4]: B=128 CPN-50 384x288 bf16 — 8 frames spread over the batch against the fp32 oracle."""
    B = 128
    model, sd = _model("cpn", "bf16", 45)
    img, k2d, kc = synth.synth_inputs(B, 384, 288, seed=46, crop_range=(288, 384))
    pick = list(range(0, B, 16))
    with torch.no_grad():
        want = oracle.ca_pf_forward(sd, img[pick], k2d[pick], kc[pick].clone(), backbone="cpn")
        got = model(img.cuda(), k2d.cuda(), kc.clone().cuda()).cpu()
    err, mpj = _report("cfg4 B=128 CPN 384x288 bf16 (8-frame slice)", got[pick], want)
    assert err <= BF16_CPN_MAX and mpj <= BF16_CPN_MEAN


# bf16 bounds: at most 2x the errors measured on the MI355X for these seeds (tests print the measured values)
# measured (stem, convs incl. the row-halo kernel, lifter projections on bf16): cfg2 1.63e-2 / 7.0e-3, cfg4 8.5e-3 / 5.2e-3
# (max / mean per-joint distance, metres)
BF16_W48_MAX, BF16_W48_MEAN = 2.7e-2, 1.4e-2
BF16_CPN_MAX, BF16_CPN_MEAN = 1.5e-2, 1.0e-2


def _train_model(B, drop):
    from mvn.models.loss import MPJPE
    model, sd = _model("hrnet_32", "fp32", 47)
    model.train(); model.backbone.eval(); model.volume_net.train()
    model.drop_path_rate = 0.2 if drop else 0.0
    img, k2d, kc, gt = synth.synth_inputs(B, 256, 256, seed=48, crop_range=(256, 256), with_gt=True)
    return model, sd, MPJPE(), (img, k2d, kc, gt)


@pytest.mark.parametrize("drop", [False, True], ids=["droppath_off", "droppath_on"])
def test_cfg3_training_step_batch64_all_191_gradients_vs_oracle_autograd(drop):
    """configs[3] (one rank's share of a 512-frame global batch): forward + MPJPE + backward at B=64 against the
    oracle's autograd for ALL 191 lifter gradients.  With DropPath on, the multipliers the host drew for the native
    step are injected into the oracle's keep= path (capf_oracle.split_drop_masks)."""
    B = 64
    model, sd, crit, (img, k2d, kc, gt) = _train_model(B, drop)
    masks = None
    if drop:
        torch.manual_seed(7)
        masks = model._drop_masks(B, torch.device("cuda"))
        assert (masks == 0).any() and masks.numel() == 2 * 4 * (B + 17 * B + B)
        model._drop_masks = lambda b, dev: masks                 # the step below uses exactly these multipliers
    pred = model(img.cuda(), k2d.cuda(), kc.clone().cuda())
    loss = crit(pred, gt.cuda())
    loss.backward()
    torch.cuda.synchronize()
    P = {k: (v.clone().requires_grad_(True) if k.startswith("volume_net.") else v) for k, v in sd.items()}
    want = oracle.ca_pf_forward(P, img, k2d, kc.clone(), backbone="hrnet_32", drop_masks=masks.cpu() if drop else None)
    ol = oracle.mpjpe(want, gt)
    ol.backward()
    err, mpj = _report(f"cfg3 B=64 train (DropPath {'on' if drop else 'off'}) prediction", pred.detach().cpu(), want.detach())
    assert err <= 1e-3 and abs(loss.item() - ol.item()) < 1e-5
    named = dict(model.named_parameters())
    worst, n = 0.0, 0
    for k, p in P.items():
        if not k.startswith("volume_net."):
            continue
        g_hip, g_ref = named[k].grad.cpu(), p.grad
        rel = ((g_hip - g_ref).abs().max() / g_ref.abs().max().clamp_min(1e-12)).item()
        worst = max(worst, rel); n += 1
        assert rel < 2e-3, (k, rel)
    print(f"  {n} gradients, worst max-abs error relative to the gradient's own max: {worst:.2e}")
    assert n == 191


def test_droppath_step_matches_reference_golden():
    """B=2 training step with the multipliers the REFERENCE drew (tests/golden/w32_256x256_b2.npz dp_*)."""
    from mvn.models.loss import MPJPE
    name = "w32_256x256_b2"
    case, g = CASES[name], load_golden(name)
    model, sd = make_model(case["backbone"], device="cuda", wseed=case["wseed"], bn=case["bn"])
    model.train(); model.backbone.eval(); model.volume_net.train()
    masks = torch.from_numpy(g["dp_masks"]).cuda()
    model._drop_masks = lambda b, dev: masks
    img, k2d, kc = case_inputs(case)
    _, _, _, gt = synth.synth_inputs(case["B"], case["H"], case["W"], seed=case["iseed"], crop_range=case["crop"], with_gt=True)
    pred = model(img.cuda(), k2d.cuda(), kc.cuda())
    loss = MPJPE()(pred, gt.cuda())
    loss.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["dp_out"], atol=1e-4)
    assert abs(loss.item() - float(g["dp_train_loss"])) < 1e-5
    named = dict(model.named_parameters())
    for key in [k for k in g.files if k.startswith("dp_grad:")]:
        want = g[key]
        got = named[key[len("dp_grad:"):]].grad.cpu().numpy()
        assert np.abs(got - want).max() <= 2e-3 * max(1e-6, np.abs(want).max()), key
    names = [str(n) for n in g["dp_gradnorm_names"]]
    for n, want in zip(names, g["dp_gradnorms"]):
        got = named[n].grad.double().norm().item()
        assert abs(got - want) <= 2e-3 * max(1e-7, want), n
