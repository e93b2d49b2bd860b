"""GPU: the HIP path (through the C ABI, via the host CA_PF) against (a) golden vectors captured from
the real reference and (b) the CPU oracle on the same seeded inputs.  fp32 tolerance 1e-3 absolute on
the 17x3 joints (BASELINE.json north_star); bilinear corner indices bit-exact."""
import numpy as np
import pytest
import torch

import capf_oracle as oracle
from conftest import load_golden, make_model
from golden_cases import CASES, case_inputs

pytestmark = pytest.mark.gpu
TOL_OUT = 1e-3


def _run(case, debug=True):
    model, sd = make_model(case["backbone"], device="cuda", wseed=case["wseed"], bn=case["bn"])
    img, k2d, kc = case_inputs(case)
    kc_dev = kc.cuda()
    img_dev = img.cuda()
    eng = model.engine_for(img_dev)
    eng.set_debug(debug)
    with torch.no_grad():
        out = model(img_dev, k2d.cuda(), kc_dev)
    torch.cuda.synchronize()
    return model, eng, sd, out.cpu(), kc_dev.cpu(), (img, k2d, kc)


@pytest.mark.parametrize("name", [n for n in CASES if not CASES[n].get("mpi")])
def test_forward_matches_reference_golden(name):
    case = CASES[name]
    g = load_golden(name)
    model, eng, sd, out, ref, _ = _run(case)
    np.testing.assert_array_equal(ref.numpy(), g["ref"])                       # in-place normalisation bit exact
    for l in range(4):
        f = eng.tensor(f"feat{l}").cpu()                                       # NHWC
        B, C, H, W = g[f"feat{l}_shape"]
        assert tuple(f.shape) == (B, H, W, C)
        h0, w0 = H // 3, W // 3
        np.testing.assert_allclose(f[:, h0:h0 + 4, w0:w0 + 4, :].numpy(), g[f"feat{l}_slice"], atol=2e-4, rtol=1e-4)
        np.testing.assert_allclose(f.double().abs().sum().item(), g[f"feat{l}_abs"], rtol=1e-4)
        np.testing.assert_allclose(eng.tensor(f"sampled{l}").cpu().numpy(), g[f"sampled{l}"], atol=2e-4, rtol=1e-4)
    B = case["B"]
    tok = eng.tensor("tok_ctx").cpu()                                          # [B,17,5,c] -> reference [B,5,17,c]
    np.testing.assert_allclose(tok.permute(0, 2, 1, 3).numpy(), g["tok_ctx"], atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(eng.tensor("tok_res").cpu().reshape(B * 17, 5, -1).numpy(), g["tok_res"], atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(eng.tensor("tok_joint").cpu().reshape(B, 17, -1).numpy(), g["tok_joint"], atol=1e-3, rtol=1e-4)
    err = np.abs(out.numpy() - g["out"]).max()
    mpj = np.linalg.norm(out.numpy() - g["out"], axis=-1).mean()
    print(f"{name}: max|hip-ref| {err:.2e}  mean joint distance {mpj:.2e}")
    assert err <= TOL_OUT


def test_reference_golden_frames_at_batch_8_run_the_large_batch_kernels():
    """The reference goldens are batch 2; the kernels the benchmark configurations run -- the split-fp32 conv tile (fp32 3x3 convs on the
    bf16 matrix pipe, from batch 6), grouped launches without split-K -- start above that.  The two golden frames repeated four times make a
    batch of 8 whose every replica must reproduce the REFERENCE's joints, context-map slices and token buffers (frames are independent)."""
    name = "w32_256x256_b2"
    case = CASES[name]
    g = load_golden(name)
    model, sd = make_model(case["backbone"], device="cuda", wseed=case["wseed"], bn=case["bn"])
    img, k2d, kc = case_inputs(case)
    R = 4
    img8, k2d8, kc8 = img.repeat(R, 1, 1, 1).cuda(), k2d.repeat(R, 1, 1).cuda(), kc.repeat(R, 1, 1).cuda()
    eng = model.engine_for(img8)
    eng.set_debug(True)
    kernels = set(k for _, k, _ in eng.op_table(2 * R))
    assert any(k.startswith("igemm_f32h2") for k in kernels), kernels
    with torch.no_grad():
        out = model(img8, k2d8, kc8).cpu()
    B = case["B"]
    want = np.tile(g["out"], (R, 1, 1, 1))
    err = np.abs(out.numpy() - want).max()
    for l in range(4):
        f = eng.tensor(f"feat{l}").cpu()
        _, C, H, W = g[f"feat{l}_shape"]
        h0, w0 = H // 3, W // 3
        for r in range(R):
            np.testing.assert_allclose(f[r * B:(r + 1) * B, h0:h0 + 4, w0:w0 + 4, :].numpy(), g[f"feat{l}_slice"], atol=2e-4, rtol=1e-4)
    tok = eng.tensor("tok_joint").cpu().reshape(R, B, 17, -1)
    for r in range(R):
        np.testing.assert_allclose(tok[r].numpy(), g["tok_joint"], atol=1e-3, rtol=1e-4)
    print(f"{name} x {R}: max|hip - reference| {err:.2e} on the split-fp32 plan")
    assert err <= TOL_OUT


@pytest.mark.parametrize("name", ["w32_256x256_adv", "w32_256x256_b2", "cpn_384x288_b1"])
def test_corner_indices_bit_exact(name):
    """idx0..3 written by the sampler == the oracle's integer corner arithmetic on the same ref."""
    case = CASES[name]
    model, eng, sd, out, ref, _ = _run(case)
    for l in range(4):
        f = eng.tensor(f"feat{l}")
        H, W = f.shape[1], f.shape[2]
        want = oracle.bilinear_corners(ref.numpy(), H, W, "zeros")
        got = eng.tensor(f"idx{l}").cpu().numpy()
        np.testing.assert_array_equal(got[..., 0], want["ix0"])
        np.testing.assert_array_equal(got[..., 1], want["iy0"])


@pytest.mark.parametrize("backbone,B,H,W", [("hrnet_32", 3, 256, 192), ("hrnet_32", 5, 128, 96), ("hrnet_48", 2, 256, 192)])
def test_forward_matches_oracle_on_fresh_inputs(backbone, B, H, W):
    """Seeds / sizes not in the golden set (odd batch, small image: ragged tiles everywhere)."""
    from capf import synth
    model, sd = make_model(backbone, device="cuda", wseed=21, bn="random")
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=22, crop_range=(W, H))
    with torch.no_grad():
        want = oracle.ca_pf_forward(sd, img, k2d, kc.clone(), backbone=backbone)
        got = model(img.cuda(), k2d.cuda(), kc.cuda()).cpu()
    err = (got - want).abs().max().item()
    print(f"{backbone} B{B} {H}x{W}: max|hip-oracle| {err:.2e}")
    assert err <= TOL_OUT


def test_batch_independence_and_determinism():
    """Frames are independent: out[b] of a batch == out of that frame alone; and two
    runs are bitwise identical (no atomics / order-dependent reductions on the path)."""
    from capf import synth
    model, _ = make_model("hrnet_32", device="cuda", wseed=5)
    img, k2d, kc = synth.synth_inputs(4, 256, 192, seed=6)
    with torch.no_grad():
        a = model(img.cuda(), k2d.cuda(), kc.clone().cuda())
        b = model(img.cuda(), k2d.cuda(), kc.clone().cuda())
        one = model(img[2:3].cuda(), k2d[2:3].cuda(), kc[2:3].clone().cuda())
    assert torch.equal(a, b)
    assert (a[2:3] - one).abs().max().item() <= 1e-5


@pytest.mark.parametrize("name", [n for n in CASES if CASES[n].get("mpi")])
def test_mpi_variant_matches_reference_golden(name):
    """N4: model.conpose.VolumetricTriangulationNet (context_blocks = 0) vs the sibling reference app, both shipped widths."""
    from test_oracle_golden import _mpi_model
    case = CASES[name]
    g = load_golden(name)
    m, _ = _mpi_model(case, device="cuda")
    img, k2d, kc = case_inputs(case)
    kc_dev = kc.cuda()
    with torch.no_grad():
        out, aux = m(img.cuda(), k2d.cuda(), kc_dev)
    assert aux is None and tuple(out.shape) == (case["B"], 3, 1, 17, 1)
    np.testing.assert_array_equal(kc_dev.cpu().numpy(), g["ref"])
    err = np.abs(out.cpu().numpy() - g["out"]).max()
    print(f"mpi variant {name}: max|hip-ref| {err:.2e}")
    assert err <= TOL_OUT
