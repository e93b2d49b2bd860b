"""GPU: the native training step of the lifter (SURVEY.md §8a rows A12 + T) against gradients captured
from the REAL reference (oracle/make_goldens.py: model.train(), backbone.eval(), DropPath forced off,
MPJPE loss, loss.backward()) on the same synthetic checkpoint and inputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, make_model
from golden_cases import CASES, case_inputs

pytestmark = pytest.mark.gpu


def _train_step(name, drop_path=0.0):
    from capf import synth
    from mvn.models.loss import MPJPE
    case = CASES[name]
    model, sd = make_model(case["backbone"], device="cuda", wseed=case["wseed"], bn=case["bn"])
    model.train(); model.backbone.eval(); model.volume_net.train()
    model.drop_path_rate = drop_path
    img, k2d, kc = case_inputs(case)
    _, _, _, gt = synth.synth_inputs(case["B"], case["H"], case["W"], seed=case["iseed"], crop_range=case["crop"], with_gt=True)
    pred = model(img.cuda(), k2d.cuda(), kc.cuda())
    loss = MPJPE()(pred, gt.cuda())
    loss.backward()
    torch.cuda.synchronize()
    return model, pred, loss


def test_loss_and_gradients_match_reference():
    g = load_golden("w32_256x256_b2")
    model, pred, loss = _train_step("w32_256x256_b2")
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["out"], atol=1e-3)
    assert abs(loss.item() - float(g["train_loss"])) < 1e-5
    named = dict(model.named_parameters())
    assert not any(p.grad is not None for p in model.backbone.parameters())
    # a few full gradients
    for key in [k for k in g.files if k.startswith("grad:")]:
        want = g[key]
        got = named[key[5:]].grad.cpu().numpy()
        scale = max(1e-6, np.abs(want).max())
        err = np.abs(got - want).max() / scale
        print(f"{key[5:]:60s} max|grad| {scale:.3e} rel err {err:.2e}")
        assert err < 2e-3, key
    # and the norm of EVERY lifter gradient
    names = [str(n) for n in g["gradnorm_names"]]
    norms = g["gradnorms"]
    worst = 0.0
    for n, want in zip(names, norms):
        got = named[n].grad.double().norm().item()
        rel = abs(got - want) / max(1e-7, want)
        worst = max(worst, rel)
        assert rel < 2e-3, (n, got, want)
    print("worst relative gradient-norm error", worst)
    assert len(names) == 191


def test_flat_gradient_and_fused_adamw_match_torch():
    """The flat buffer capf_backward writes is what .grad views alias; one fused AdamW step on the flattened
    parameters equals torch.optim.AdamW on a copy."""
    from capf.optim import FusedAdamW, flatten_
    model, pred, loss = _train_step("w32_256x256_b2")
    vn = model.volume_net
    ref_params = [p.detach().clone().requires_grad_(True) for p in vn.parameters()]
    for rp, p in zip(ref_params, vn.parameters()):
        rp.grad = p.grad.detach().clone()
    opt = torch.optim.AdamW(ref_params, lr=6.4e-4, weight_decay=0.1)
    opt.step()
    flat_g = torch.cat([p.grad.reshape(-1) for p in vn.parameters()])
    flat_p = flatten_(vn)
    FusedAdamW(flat_p, lr=6.4e-4, weight_decay=0.1).step(flat_g)
    torch.cuda.synchronize()
    for rp, p in zip(ref_params, vn.parameters()):
        assert (rp.detach() - p.detach()).abs().max().item() < 2e-6
    # parameters changed in place -> the next forward must repack and give a different prediction
    img, k2d, kc = case_inputs(CASES["w32_256x256_b2"])
    with torch.no_grad():
        pred2 = model(img.cuda(), k2d.cuda(), kc.cuda())
    assert (pred2 - pred.detach()).abs().max().item() > 1e-5


def test_droppath_masks_scale_branches():
    """With DropPath on, the step still runs, stays finite, and differs from the no-drop prediction."""
    torch.manual_seed(0)
    _, pred_drop, loss = _train_step("w32_256x256_b2", drop_path=0.5)
    _, pred_nodrop, _ = _train_step("w32_256x256_b2", drop_path=0.0)
    assert torch.isfinite(pred_drop).all() and torch.isfinite(loss)
    assert (pred_drop - pred_nodrop).abs().max().item() > 1e-4


def test_flat_grad_only_leaves_the_same_gradient_in_the_flat_buffer():
    """CA_PF.flat_grad_only: backward writes last_flat_grad and sets no .grad (what bench.py's fused-AdamW step uses); the flat
    buffer holds exactly the slices the default mode hands to autograd."""
    from capf import synth
    from mvn.models.loss import MPJPE
    case = CASES["w32_256x256_b2"]
    img, k2d, kc = case_inputs(case)
    _, _, _, gt = synth.synth_inputs(case["B"], case["H"], case["W"], seed=case["iseed"], crop_range=case["crop"], with_gt=True)
    flats = []
    for only in (False, True):
        model, _ = make_model(case["backbone"], device="cuda", wseed=case["wseed"], bn=case["bn"])
        model.train(); model.backbone.eval(); model.volume_net.train()
        model.drop_path_rate = 0.0
        model.flat_grad_only = only
        MPJPE()(model(img.cuda(), k2d.cuda(), kc.clone().cuda()), gt.cuda()).backward()
        torch.cuda.synchronize()
        assert all((p.grad is None) == only for p in model.volume_net.parameters())
        flats.append(model.last_flat_grad.clone())
        if not only:
            layout, _ = model.engine_for(img.cuda()).grad_layout_cached()
            for n, p in model.volume_net.named_parameters():
                off, cnt = layout["volume_net." + n] if ("volume_net." + n) in layout else layout[n]
                assert torch.equal(p.grad.reshape(-1), flats[0][off: off + cnt])
    assert torch.equal(flats[0], flats[1])


@pytest.mark.parametrize("drop_path", [0.0, 0.3])
def test_training_step_on_the_two_piece_gemm_agrees_with_the_fp32_pipe(drop_path):
    """From batch 5 the step's nn.Linear products -- forward y = x W^T and backward dX = dY W -- run on the two-fp16-piece GEMM off packs
    made from the current parameters at the start of every step (csrc/train.cpp t_h2_prepare; W and W^T of 56 matrices in three
    launches).  Against the same step on the fp32 matrix pipe (CAPF_PLAN_NO_F32H2_GEMM): prediction, loss and every one of the 191
    gradients, with and without DropPath (the per-row branch scale rides in the GEMM epilogue)."""
    from capf import synth
    from capf.lib import PLAN_NO_F32H2_GEMM
    from mvn.models.loss import MPJPE
    case = CASES["w32_256x256_b2"]
    B = 12
    img, k2d, kc, gt = synth.synth_inputs(B, case["H"], case["W"], seed=4242, crop_range=case["crop"], with_gt=True)
    got = []
    for flags in (0, PLAN_NO_F32H2_GEMM):
        model, _ = make_model(case["backbone"], device="cuda", wseed=case["wseed"], bn=case["bn"], plan_flags=flags)
        model.train(); model.backbone.eval(); model.volume_net.train()
        model.drop_path_rate = drop_path
        torch.manual_seed(7)
        pred = model(img.cuda(), k2d.cuda(), kc.clone().cuda())
        loss = MPJPE()(pred, gt.cuda())
        loss.backward()
        torch.cuda.synchronize()
        eng = model.engine_for(img.cuda())
        n_h2 = eng.lib.capf_train_h2_matrices(eng.h)
        assert (n_h2 >= 40) == (flags == 0), n_h2
        got.append((pred.detach().clone(), loss.item(), {n: p.grad.detach().clone() for n, p in model.volume_net.named_parameters()}))
    (p1, l1, g1), (p0, l0, g0) = got
    assert (p1 - p0).abs().max().item() < 2e-6 * max(1.0, p0.abs().max().item())
    assert abs(l1 - l0) < 1e-6 * max(1.0, abs(l0))
    worst = 0.0
    for n in g0:
        scale = max(1e-9, g0[n].abs().max().item())
        worst = max(worst, (g1[n] - g0[n]).abs().max().item() / scale)
        assert (g1[n] - g0[n]).abs().max().item() <= 2e-5 * scale, n
    print(f"two-piece GEMM vs fp32 pipe, batch {B}, drop_path {drop_path}: worst gradient difference {worst:.2e} of the gradient's largest entry")
    assert len(g0) == 191


@pytest.mark.parametrize("B", [12, 96])
def test_batched_second_stage_reductions_leave_the_same_bits(B):
    """capf_backward defers every weight gradient's slab sum and every bias / LayerNorm gradient's second reduction stage into a handful of
    batched launches (csrc/train.cpp t_slab_flush / t_col_flush, train_kernels.hip slab_sum_batch_kernel / colreduce_final_batch_kernel:
    64 + 32 launches per step -> ~8).  Per element the batched kernels add the same partial sums in the same order as the per-layer
    launches (CAPF_PLAN_NO_BATCHED_REDUCE): loss and all 191 gradients bit for bit, with DropPath on.  Batch 12: most weight gradients
    are one slice (no slabs at all) and the column reductions carry the test; batch 96: split-K slabs for every matrix and the flushes in
    front of the context blocks' temp copies (the flush-when-full path needs ~0.6 GB of slabs: it runs in test_gpu_fullsize.py's
    batch-512 steps, which hold the gradients to the oracle / fp64)."""
    from capf import synth
    from capf.lib import PLAN_NO_BATCHED_REDUCE
    from mvn.models.loss import MPJPE
    case = CASES["w32_256x256_b2"]
    img, k2d, kc, gt = synth.synth_inputs(B, case["H"], case["W"], seed=977, crop_range=case["crop"], with_gt=True)
    got = []
    for flags in (0, PLAN_NO_BATCHED_REDUCE):
        model, _ = make_model(case["backbone"], device="cuda", wseed=case["wseed"], bn=case["bn"], plan_flags=flags)
        model.train(); model.backbone.eval(); model.volume_net.train()
        model.drop_path_rate = 0.3
        torch.manual_seed(11)
        pred = model(img.cuda(), k2d.cuda(), kc.clone().cuda())
        loss = MPJPE()(pred, gt.cuda())
        loss.backward()
        torch.cuda.synchronize()
        got.append((loss.item(), {n: p.grad.detach().clone() for n, p in model.volume_net.named_parameters()}))
        del model
    (l1, g1), (l0, g0) = got
    assert l1 == l0
    assert len(g0) == 191
    for n in g0:
        assert torch.isfinite(g0[n]).all(), n
        assert torch.equal(g1[n], g0[n]), n
