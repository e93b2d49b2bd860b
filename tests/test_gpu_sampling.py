"""GPU: the integer part of the path — bilinear corner indices of BOTH sampling sites — bit for bit against the oracle's
numpy restatement of ATen's rule (oracle.bilinear_corners; GridSampler.h:27-36, 58-60, 143-171), and the bf16 mode against
the bf16-emulating oracle, stage by stage.  (The zeros-mode site's idx{l} taps are in test_gpu_parity.py.)"""
import numpy as np
import pytest
import torch

import capf_oracle as oracle
from bf16_report import bf16_stage_report, check_bf16_report
from conftest import make_model
from golden_cases import CASES, case_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["w32_256x256_adv", "w32_256x256_b2", "cpn_384x288_b1"])
@pytest.mark.parametrize("train", [False, True], ids=["fused_inference_kernel", "training_kernel"])
def test_deformable_corner_indices_bit_exact(name, train):
    """The border-mode sampling site (DeformableBlock, pose_dformer.py:126-128): cidx{i} = the NW corners the kernel gathered
    from, cpos{i} = the positions it computed them from.  (a) corners == oracle.bilinear_corners(the kernel's own positions,
    'border') bit for bit, for all 4 blocks x 4 levels x 17 joints x 16 samples; (b) those positions == the oracle's
    tanh(linear(LayerNorm)) + ref to fp32 roundoff.  Both kernels that implement the site are covered: ctx_attn_kernel
    (inference plan) and deform_sample_kernel (capf_forward_train)."""
    case = CASES[name]
    model, sd = make_model(case["backbone"], device="cuda", wseed=case["wseed"], bn=case["bn"])
    img, k2d, kc = case_inputs(case)
    B = case["B"]
    eng = model.engine_for(img.cuda())
    eng.set_debug(True)
    taps = {}
    with torch.no_grad():
        oracle.ca_pf_forward(sd, img, k2d, kc.clone(), backbone=case["backbone"], taps=taps)
    if train:
        model.train(); model.backbone.eval(); model.drop_path_rate = 0.0
        model(img.cuda(), k2d.cuda(), kc.clone().cuda())
    else:
        with torch.no_grad():
            model(img.cuda(), k2d.cuda(), kc.clone().cuda())
    torch.cuda.synchronize()
    n_checked = 0
    for i in range(4):
        pos = eng.tensor(f"cpos{i}").cpu().view(B, 17, 4, 16, 2)
        idx = eng.tensor(f"cidx{i}").cpu().view(B, 17, 4, 16, 2).numpy()
        want_pos = taps["ctx_pos"][i].permute(0, 2, 1, 3, 4)                  # oracle: [b, l, p, 16, 2]
        assert (pos - want_pos).abs().max().item() <= 2e-5
        for l in range(4):
            f = eng.tensor(f"feat{l}")
            H, W = f.shape[1], f.shape[2]
            want = oracle.bilinear_corners(pos[:, :, l].numpy(), H, W, "border")
            np.testing.assert_array_equal(idx[:, :, l, :, 0], want["ix0"])
            np.testing.assert_array_equal(idx[:, :, l, :, 1], want["iy0"])
            assert want["ix0"].min() >= 0 and want["ix0"].max() <= W - 1 and want["iy0"].min() >= 0 and want["iy0"].max() <= H - 1
            n_checked += want["ix0"].size
    assert n_checked == 4 * 4 * B * 17 * 16


def adversarial_grid(H, W):
    g = [-1.0, 1.0, -0.0, 0.0, 1e-42, -1e-42, -1.5, 1.5, -3.0, 3.0, 1e6, -1e6, 0.999999, -0.999999, 1.0000001, -1.0000001]
    for size in (W, H):
        for k in range(size):
            c = np.float32(k) / np.float32(size - 1) * np.float32(2) - np.float32(1)      # pixel centre k, normalised
            g += [float(c), float(np.nextafter(c, np.float32(2))), float(np.nextafter(c, np.float32(-2)))]
    rng = np.random.Generator(np.random.Philox(key=[H * 1000 + W, 7]))
    g += rng.uniform(-1.3, 1.3, size=4096).astype(np.float32).tolist()
    g = np.asarray(g, dtype=np.float32)
    grid = np.stack([g, g[::-1]], -1)                                                     # (x, y) pairs mixing all of the above
    return np.ascontiguousarray(np.concatenate([grid, np.stack([g, np.roll(g, 17)], -1)], 0))


@pytest.mark.parametrize("H,W", [(64, 64), (8, 8), (64, 48), (12, 9)])
def test_corner_rule_on_adversarial_coordinates(H, W):
    """capf_op_bilinear_corners (the device function both sampling sites call) against oracle.bilinear_corners on
    coordinates chosen to break index arithmetic: exact pixel centres and the floats either side of them, the -1 / +1
    borders and beyond (clipping in border mode, out-of-range corners in zeros mode), -0.0, denormals, huge values.
    Indices AND fractional weights must be bit-identical in both padding modes."""
    from capf.lib import bilinear_corners
    grid = adversarial_grid(H, W)
    dev = torch.from_numpy(grid).cuda()
    for mode in ("border", "zeros"):
        idx, frac = bilinear_corners(dev, H, W, border=(mode == "border"))
        want = oracle.bilinear_corners(grid, H, W, mode)
        np.testing.assert_array_equal(idx.cpu().numpy()[:, 0], want["ix0"])
        np.testing.assert_array_equal(idx.cpu().numpy()[:, 1], want["iy0"])
        np.testing.assert_array_equal(frac.cpu().numpy()[:, 0], want["wx1"])
        np.testing.assert_array_equal(frac.cpu().numpy()[:, 1], want["wy1"])


@pytest.mark.parametrize("backbone,B,H,W", [("hrnet_32", 2, 256, 256), ("hrnet_48", 2, 256, 256), ("cpn", 1, 384, 288)])
def test_bf16_path_matches_the_bf16_emulating_oracle(backbone, B, H, W):
    """compute_dtype='bf16' (BASELINE configs[2]/[4]): bf16 MFMA convolutions with bf16 activations, and the lifter's
    qkv / proj / fc1 / fc2 projections on bf16 operands (fp32 accumulation; LayerNorm, softmax, samplers and the residual
    stream stay fp32) against the oracle run with the SAME storage roundings and against the fp32 oracle, stage by stage
    (bounds and their derivation: bf16_report.py; the tight layer-wise check: test_gpu_layerwise.py)."""
    import copy, contextlib, io
    from capf import synth
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), backbone)
    cfg.model.backbone.fix_weights = True
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg, compute_dtype="bf16").eval()
    sd = synth.load_synthetic(model, seed=31, bn_mode="random")
    model = model.cuda()
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=32)
    taps_e, taps_f = {}, {}
    with torch.no_grad():
        want_e = oracle.ca_pf_forward(sd, img, k2d, kc.clone(), backbone=backbone, taps=taps_e, emulate_bf16=True)
        want_f = oracle.ca_pf_forward(sd, img, k2d, kc.clone(), backbone=backbone, taps=taps_f)
        eng = model.engine_for(img.cuda())
        eng.set_debug(True)
        got = model(img.cuda(), k2d.cuda(), kc.cuda()).cpu()
    check_bf16_report(bf16_stage_report(f"{backbone} bf16 B={B}", eng, got, None, taps_e, want_e, taps_f, want_f))
