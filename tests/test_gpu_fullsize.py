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
from bf16_report import BUDGET_CAP, bf16_stage_report, check_bf16_report
from conftest import load_golden, make_model
from golden_cases import CASES, case_inputs

pytestmark = pytest.mark.gpu


def _report(tag, got, want):
    err = (got - want).abs().max().item()
    mpj = (got - want).norm(dim=-1).mean().item()
    print(f"{tag}: max|hip - oracle| {err:.3e}   mean per-joint distance {mpj:.3e}")
    return err, mpj


def _model(backbone, dtype, wseed, plan_flags=0):
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), backbone)
    cfg.model.backbone.fix_weights = True
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg, compute_dtype=dtype, plan_flags=plan_flags).eval()
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


def _bf16_fullsize(tag, backbone, B, H, W, wseed, iseed, pick):
    """One bf16 run at a BASELINE batch: a slice of frames spread over the batch against the bf16-emulating oracle and the fp32
    oracle, stage by stage (bounds and their derivation: bf16_report.py; the tight layer-wise check: test_gpu_layerwise.py);
    plus batch independence against the slice run on its own."""
    model, sd = _model(backbone, "bf16", wseed)
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=iseed, crop_range=(W, H))
    taps_e, taps_f = {}, {}
    with torch.no_grad():
        want_e = oracle.ca_pf_forward(sd, img[pick], k2d[pick], kc[pick].clone(), backbone=backbone, taps=taps_e, emulate_bf16=True)
        want_f = oracle.ca_pf_forward(sd, img[pick], k2d[pick], kc[pick].clone(), backbone=backbone, taps=taps_f)
        eng = model.engine_for(img.cuda())
        eng.set_debug(True)
        got = model(img.cuda(), k2d.cuda(), kc.clone().cuda()).cpu()
        rep = bf16_stage_report(tag, eng, got, pick, taps_e, want_e, taps_f, want_f)
        sub = model(img[pick].cuda(), k2d[pick].cuda(), kc[pick].clone().cuda()).cpu()
    check_bf16_report(rep)
    # tile shapes and kernels depend on the batch -> another fp32 summation order -> (chaos at the rounding level, bf16_report.py)
    # another point at the noise floor: bounded like the distance to the emulation
    d = (got[pick] - sub).abs().max().item()
    print(f"  batch independence (B={B} vs B={len(pick)}): max delta {d:.3e}")
    assert d <= 1.5 * (rep["joints"][1] + rep["joints"][2]) + 1e-3 and d <= BUDGET_CAP


def test_cfg2_batch256_hrnet48_bf16_slice_vs_bf16_emulating_oracle():
    """configs[2]: B=256 HRNet-48 256x256 bf16 — 16 frames spread over the batch."""
    _bf16_fullsize("cfg2 B=256 W48 256x256 bf16 (16-frame slice)", "hrnet_48", 256, 256, 256, 43, 44, list(range(0, 256, 16)))


def test_cfg4_batch128_cpn_384x288_bf16_slice_vs_bf16_emulating_oracle():
    """configs[4]: B=128 CPN-50 384x288 bf16 — 8 frames spread over the batch."""
    _bf16_fullsize("cfg4 B=128 CPN 384x288 bf16 (8-frame slice)", "cpn", 128, 384, 288, 45, 46, list(range(0, 128, 16)))


@pytest.mark.parametrize("backbone,H,W", [("hrnet_48", 256, 256), ("cpn", 384, 288)])
def test_batch64_fp32_slice_of_the_other_backbones_vs_oracle(backbone, H, W):
    """The advertised fp32 combinations that are not BASELINE rows (HRNet-48 fp32, CPN fp32) at a bench-sized batch: tile
    shapes / Winograd variants / grouped launches of B=64, 8 frames spread over the batch vs the fp32 oracle at 1e-3."""
    B = 64
    model, sd = _model(backbone, "fp32", 51)
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=52, crop_range=(W, H))
    pick = list(range(3, B, 8))
    with torch.no_grad():
        want = oracle.ca_pf_forward(sd, img[pick], k2d[pick], kc[pick].clone(), backbone=backbone)
        got = model(img.cuda(), k2d.cuda(), kc.clone().cuda()).cpu()
    err, mpj = _report(f"{backbone} B=64 {H}x{W} fp32 (8-frame slice)", got[pick], want)
    assert err <= 1e-3


def check_cells_against_index_rule(eng, B, tag):
    """At a PRODUCTION batch: the NW corners the deformable sampler gathered from (cidx{i}) == oracle.bilinear_corners(the positions the
    kernel itself computed (cpos{i}), 'border') bit for bit, for all 4 blocks x 4 levels x B x 17 x 16 samples; prints how many samples
    sit within 4 ulp of a cell boundary (where grid_sample's one-sided derivative makes the choice of cell matter)."""
    n, near = 0, 0
    for i in range(4):
        pos = eng.tensor(f"cpos{i}")[:B].cpu().view(B, 17, 4, 16, 2).numpy()
        idx = eng.tensor(f"cidx{i}")[:B].cpu().view(B, 17, 4, 16, 2).numpy()
        for l in range(4):
            f = eng.tensor(f"feat{l}")
            H, W = f.shape[1], f.shape[2]
            want = oracle.bilinear_corners(pos[:, :, l], H, W, "border")
            np.testing.assert_array_equal(idx[:, :, l, :, 0], want["ix0"])
            np.testing.assert_array_equal(idx[:, :, l, :, 1], want["iy0"])
            for g, size in ((pos[:, :, l, :, 0], W), (pos[:, :, l, :, 1], H)):
                x = np.clip(((g + np.float32(1)) / np.float32(2)) * np.float32(size - 1), 0, size - 1).astype(np.float32)
                inside = (x > 0) & (x < size - 1)                 # (a coordinate the border clip pinned to 0 / size - 1 has no choice of cell)
                near += int((inside & (np.abs(x - np.rint(x)) <= 4 * np.spacing(np.maximum(np.abs(x), np.float32(1))))).sum())
            n += want["ix0"].size
    print(f"  {tag}: {n} deformable samples, corner indices == ATen's rule on the kernel's own positions bit for bit; "
          f"{near} unclipped coordinates within 4 ulp of a cell boundary")
    assert n == 4 * 4 * B * 17 * 16
    return near


# parameters whose gradient contains NO derivative of a sample w.r.t. its position (everything behind the last deformable sampler, and
# the last context block's value / weight path): grid_sample's one-sided position derivative cannot touch them, so they are compared
# against the oracle with its OWN floor() cells
def _independent_of_cells(k):
    return (k.startswith(("volume_net.res_blocks.", "volume_net.joint_blocks.", "volume_net.head.")) or
            k.startswith(("volume_net.context_blocks.3.embed_proj.", "volume_net.context_blocks.3.attention_weights.",
                          "volume_net.context_blocks.3.mlp.", "volume_net.context_blocks.3.norm2.")))


def test_cfg1_batch64_inference_corner_indices_bit_exact():
    """The fused inference kernel (ctx_attn_kernel) at configs[1]'s batch."""
    B = 64
    model, sd = _model("hrnet_32", "fp32", 41)
    img, k2d, kc = synth.synth_inputs(B, 256, 256, seed=42, crop_range=(256, 256))
    eng = model.engine_for(img.cuda())
    eng.set_debug(True)
    with torch.no_grad():
        model(img.cuda(), k2d.cuda(), kc.cuda())
    torch.cuda.synchronize()
    check_cells_against_index_rule(eng, B, "cfg1 B=64 inference (ctx_attn_kernel)")


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
    eng = model.engine_for(img.cuda())
    eng.set_debug(True)                                        # cidx{i} taps: the bilinear cells of the deformable samplers
    pred = model(img.cuda(), k2d.cuda(), kc.clone().cuda())
    loss = crit(pred, gt.cuda())
    loss.backward()
    torch.cuda.synchronize()
    check_cells_against_index_rule(eng, B, f"cfg3 B=64 train (deform_sample_kernel, DropPath {'on' if drop else 'off'})")
    # the oracle differentiates the deformable samplers in the cells the engine used (see the batch-512 test below)
    cells = [eng.tensor(f"cidx{i}")[:B].cpu().view(B, 17, 4, 16, 2).long() for i in range(4)]
    P = {k: (v.clone().requires_grad_(True) if k.startswith("volume_net.") else v) for k, v in sd.items()}
    want = oracle.ca_pf_forward(P, img, k2d, kc.clone(), backbone="hrnet_32", drop_masks=masks.cpu() if drop else None, cells=cells)
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
        assert rel < 1e-4, (k, rel)                            # (fp32 against fp32: measured 4e-6)
    print(f"  {n} gradients, worst max-abs error relative to the gradient's own max: {worst:.2e}")
    assert n == 191
    # ... and an INDEPENDENT yardstick: the oracle differentiated in its own floor() cells, for every gradient that contains no
    # derivative of a sample w.r.t. its position
    P2 = {k: (v.clone().requires_grad_(True) if k.startswith("volume_net.") else v) for k, v in sd.items()}
    w2 = oracle.ca_pf_forward(P2, img, k2d, kc.clone(), backbone="hrnet_32", drop_masks=masks.cpu() if drop else None)
    oracle.mpjpe(w2, gt).backward()
    worst2, n2 = 0.0, 0
    for k, p in P2.items():
        if k.startswith("volume_net.") and _independent_of_cells(k):
            rel = ((named[k].grad.cpu() - p.grad).abs().max() / p.grad.abs().max().clamp_min(1e-12)).item()
            worst2 = max(worst2, rel); n2 += 1
            assert rel < 1e-4, (k, rel)
    print(f"  {n2} gradients without a position derivative vs the oracle in its OWN floor() cells: worst {worst2:.2e}")
    assert n2 == 116


def test_cfg3_training_step_batch512_vs_oracle_autograd():
    """configs[3] AT ITS OWN SIZE: 512 frames per GPU (the reference's batch_size is per process, train.py:72-80).  Forward +
    MPJPE + backward of the whole 512-frame batch against the oracle's autograd (DropPath off): prediction for all 512 frames,
    loss, all 191 gradients; and batch independence against the same frames run as a 64-frame batch (other tile shapes,
    grouped-launch configurations and split-K factors: accumulation order only).  This is the size at which the Winograd
    kernel's bookkeeping reads 1.03 of the nominal peak — the oracle says whether it does all the work."""
    B = 512
    model, sd, crit, (img, k2d, kc, gt) = _train_model(B, False)
    torch.set_num_threads(min(64, torch.get_num_threads()))
    eng = model.engine_for(img.cuda())
    eng.set_debug(True)                                        # cidx{i} taps: the bilinear cells of the deformable samplers
    pred = model(img.cuda(), k2d.cuda(), kc.clone().cuda())
    loss = crit(pred, gt.cuda())
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
    P = {k: (v.clone().requires_grad_(True) if k.startswith("volume_net.") else v) for k, v in sd.items()}
    chunks = []
    with torch.no_grad():                                      # frozen backbone: no graph, 64 frames at a time bounds host memory
        for b0 in range(0, B, 64):
            x = img[b0:b0 + 64].permute(0, 3, 1, 2).contiguous()
            chunks.append(oracle.hrnet_forward(P, x))
    feats = [torch.cat([c[l] for c in chunks], 0) for l in range(4)]
    ref = oracle.normalise_crop_keypoints_(kc.clone())
    want = oracle.lifter_forward(P, k2d, ref, feats)
    ol = oracle.mpjpe(want, gt)
    ol.backward()
    err, mpj = _report("cfg3 B=512 train (DropPath off) prediction", pred.detach().cpu(), want.detach())
    assert err <= 1e-3 and abs(loss.item() - ol.item()) < 1e-5
    # The yardstick for the gradients is the SAME lifter evaluated in fp64 (on the fp32 oracle's context maps).  Two things make
    # a naive bound on the offset-path gradients meaningless and are dealt with instead of being tuned away:
    #  (a) conditioning: a sampling-offset gradient is the small remainder of 34816 cancelling rows (differences of bilinear
    #      corners) -- the fp32 ORACLE itself sits up to 1e-3 of the gradient's max from fp64, so the bound is relative to it;
    #  (b) grid_sample's derivative w.r.t. the position is ONE-SIDED at cell boundaries: of the 2.2 M border-mode samples of this
    #      step a few dozen lie within fp32 roundoff of a boundary, and two correct evaluations whose positions differ in the last
    #      bits differentiate different cells there (a kernel change that only reordered the stem conv's summation moved
    #      context_blocks.0.sampling_offsets.weight by 1.9e-3 relative L2 that way).  The yardstick therefore evaluates its
    #      samplers IN THE CELLS THE ENGINE USED (its cidx taps -- themselves checked bit for bit against ATen's index rule
    #      in test_gpu_sampling.py): capf_oracle.grid_sample_in_cells, the same piecewise-bilinear function without the floor().
    # With the cells shared, (a) disappears from the comparison as well (both sides sum the same rows; measured: every one of the
    # 191 gradients within 1.4e-6 relative L2 and 3.1e-6 of its max per entry, where the fp32 oracle with its own floor() cells
    # sits up to 2.3e-4 from the fp64 yardstick).  Bound: 2e-5 for both.
    check_cells_against_index_rule(eng, B, "cfg3 B=512 train (deform_sample_kernel)")
    cells = [eng.tensor(f"cidx{i}")[:B].cpu().view(B, 17, 4, 16, 2).long() for i in range(4)]
    # independent yardstick (the fp32 oracle above differentiated in its OWN floor() cells): every gradient that contains no derivative
    # of a sample w.r.t. its position
    worst2, n2 = 0.0, 0
    for k, p in P.items():
        if k.startswith("volume_net.") and _independent_of_cells(k):
            rel = ((grads[k] - p.grad).abs().max() / p.grad.abs().max().clamp_min(1e-12)).item()
            worst2 = max(worst2, rel); n2 += 1
            assert rel < 1e-4, (k, rel)
    print(f"  {n2} gradients without a position derivative vs the fp32 oracle in its OWN floor() cells: worst {worst2:.2e}")
    assert n2 == 116

    def lifter64(fs, cells=None):
        Q = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("volume_net.")}
        w = oracle.lifter_forward(Q, k2d.double(), ref.double(), fs, cells=cells)
        oracle.mpjpe(w, gt.double()).backward()
        return {k: q.grad for k, q in Q.items()}, w.detach()

    f64 = [f.double() for f in feats]
    g64, w64 = lifter64(f64)                                  # floor() cells: the yardstick of the fp32 oracle
    g64c, w64c = lifter64(f64, cells)                         # the engine's cells: the yardstick of the engine
    dcell = (w64c - w64).abs().max().item()
    print(f"  fp64 lifter, engine's cells vs floor() cells: prediction differs by {dcell:.2e} (same function, other branch at a few boundaries)")
    assert dcell <= 1e-6
    rows = []
    for k, p in P.items():
        if not k.startswith("volume_net."):
            continue
        t, tc = g64[k], g64c[k]
        nrm, scale = t.norm().clamp_min(1e-30), t.abs().max().clamp_min(1e-30)
        dh, df = grads[k].double() - tc, p.grad.double() - t
        rows.append(((dh.norm() / nrm).item(), (df.norm() / nrm).item(), (dh.abs().max() / scale).item(), (df.abs().max() / scale).item(),
                     ((tc - t).norm() / nrm).item(), k))
    rows.sort(reverse=True)
    print(f"  {len(rows)} gradients at B=512 vs the fp64 lifter; worst five (relative L2: HIP, fp32 oracle | max entry / max: HIP, fp32 "
          f"oracle | what the choice of cells alone moves):")
    for l2h, l2f, mh, mf, l2c, k in rows[:5]:
        print(f"    {k:58s} {l2h:9.2e} {l2f:9.2e} | {mh:9.2e} {mf:9.2e} | {l2c:9.2e}")
    assert len(rows) == 191
    for l2h, l2f, mh, mf, l2c, k in rows:
        assert l2h <= 2e-5 and mh <= 2e-5, (k, l2h, mh, l2f, mf)
    with torch.no_grad():
        model.eval()
        sub = model(img[128:192].cuda(), k2d[128:192].cuda(), kc[128:192].clone().cuda()).cpu()
    d = (pred.detach().cpu()[128:192] - sub).abs().max().item()
    print(f"  frames 128..191 inside B=512 (training plan) vs as a B=64 inference batch: max delta {d:.3e}")
    assert d <= 5e-5


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
