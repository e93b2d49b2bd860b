"""N2 metrics: the numpy oracle against goldens captured from the reference's own loss.py (CPU), and
the HIP kernels behind `mvn.models.loss` / `mvn.datasets.human36m.evaluate_using_pred` against both (GPU)."""
import numpy as np
import pytest

import metrics_oracle as mo
from conftest import load_golden
from golden_cases import metric_inputs


def _sq():
    pred, gt, action_idx, validity = metric_inputs()
    return pred, gt, pred[:, 0], gt[:, 0], action_idx, validity


def test_oracle_matches_reference_loss_goldens():
    g = load_golden("losses")
    pred, gt, p3, g3, action_idx, validity = _sq()
    np.testing.assert_allclose(mo.mpjpe_per_pose(p3, g3).mean(), g["mpjpe"], rtol=2e-6)
    np.testing.assert_allclose(mo.p_mpjpe_per_pose(p3, g3), g["p_mpjpe_per_pose"], rtol=2e-5)     # reference SVD runs in fp32
    np.testing.assert_allclose(mo.p_mpjpe_per_pose(p3, g3).mean(), g["p_mpjpe"], rtol=5e-6)
    np.testing.assert_allclose(mo.n_mpjpe_per_pose(p3, g3).mean(), g["n_mpjpe"], rtol=2e-6)
    np.testing.assert_allclose(mo.velocity_errors(p3, g3).mean(), g["mpjve"], rtol=2e-6)
    np.testing.assert_allclose(mo.per_action(p3, g3, action_idx, 6), g["per_action"], rtol=5e-6)
    np.testing.assert_allclose(mo.keypoints_loss(0, p3, g3, validity), g["kp_mse"], rtol=2e-6)
    np.testing.assert_allclose(mo.keypoints_loss(1, p3, g3, validity, 0.05), g["kp_mse_smooth"], rtol=2e-6)
    np.testing.assert_allclose(mo.keypoints_loss(2, p3, g3, validity), g["kp_mae"], rtol=2e-6)


def test_horn_closed_form_equals_the_svd_form():
    """The kernel's route to the Procrustes rotation (quaternion eigenvector) against the reference's SVD route,
    including poses whose best orthogonal map is a reflection (the det(R) = -1 branch of loss.py:54-58)."""
    rng = np.random.default_rng(5)
    gt = rng.standard_normal((64, 17, 3)) * 0.3
    pred = gt + rng.standard_normal((64, 17, 3)) * 0.05
    pred[:16] = gt[:16] * np.array([1.0, 1.0, -1.0]) + rng.standard_normal((16, 17, 3)) * 0.01     # mirrored poses
    pred[16:24] = rng.standard_normal((8, 17, 3))                                                    # unrelated poses
    np.testing.assert_allclose(mo.p_mpjpe_horn_per_pose(pred, gt), mo.p_mpjpe_per_pose(pred, gt), rtol=1e-9, atol=1e-12)


def test_previous_in_segment():
    from mvn.datasets.human36m import previous_in_segment
    assert previous_in_segment([0, 0, 1, 0, 1, 2]).tolist() == [-1, 0, -1, 1, 2, -1]


@pytest.mark.gpu
def test_metric_classes_match_reference_goldens():
    import torch
    from mvn.models.loss import MPJPE, MPJVE, N_MPJPE, P_MPJPE, KeypointsMAELoss, KeypointsMSELoss, KeypointsMSESmoothLoss
    g = load_golden("losses")
    pred, gt, p3, g3, action_idx, validity = _sq()
    tp, tg = torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda()
    np.testing.assert_allclose(MPJPE()(tp, tg).item(), g["mpjpe"], rtol=2e-6)
    np.testing.assert_allclose(N_MPJPE()(tp, tg).item(), g["n_mpjpe"], rtol=2e-6)
    np.testing.assert_allclose(P_MPJPE()(p3, g3), g["p_mpjpe"], rtol=5e-6)          # numpy in, numpy scalar out (human36m.py:374)
    np.testing.assert_allclose(MPJVE()(p3, g3), g["mpjve"], rtol=2e-6)
    from capf.lib import pose_errors
    err = pose_errors(tp[:, 0].contiguous(), tg[:, 0].contiguous()).cpu().numpy()
    np.testing.assert_allclose(err[:, 1], g["p_mpjpe_per_pose"], rtol=2e-5)
    v = torch.from_numpy(validity).cuda()
    np.testing.assert_allclose(KeypointsMSELoss()(tp[:, 0], tg[:, 0], v).item(), g["kp_mse"], rtol=2e-6)
    np.testing.assert_allclose(KeypointsMSESmoothLoss(0.05)(tp[:, 0], tg[:, 0], v).item(), g["kp_mse_smooth"], rtol=2e-6)
    np.testing.assert_allclose(KeypointsMAELoss()(tp[:, 0], tg[:, 0], v).item(), g["kp_mae"], rtol=2e-6)


@pytest.mark.gpu
def test_evaluate_using_pred_matches_reference_per_action_sums():
    import torch
    from mvn.datasets.human36m import evaluate_using_pred
    g = load_golden("losses")
    pred, gt, p3, g3, action_idx, validity = _sq()
    names = ["Walking-1", "Walking-2", "Eating-1", "Eating-2", "Posing-1", "Posing-2"]
    res = evaluate_using_pred(torch.from_numpy(gt).cuda(), torch.from_numpy(pred).cuda(), action_idx, names)
    per = g["per_action"]
    assert sorted(res) == ["Eating", "Posing", "Walking"]
    for base, (a, b) in {"Walking": (0, 1), "Eating": (2, 3), "Posing": (4, 5)}.items():
        n = per[a, 3] + per[b, 3]
        np.testing.assert_allclose(res[base]["MPJPE"], (per[a, 0] + per[b, 0]) / n, rtol=5e-6)
        np.testing.assert_allclose(res[base]["P_MPJPE"], (per[a, 1] + per[b, 1]) / n, rtol=5e-6)
        np.testing.assert_allclose(res[base]["MPJVE"], (per[a, 2] + per[b, 2]) / n, rtol=5e-6)
    with pytest.raises(ValueError):
        evaluate_using_pred(torch.from_numpy(gt).cuda(), torch.from_numpy(pred[:-1]).cuda(), action_idx, names)


@pytest.mark.gpu
def test_metrics_large_random_set_and_gradients():
    """50k poses against the numpy oracle, and the keypoint-loss gradients against torch autograd of the reference
    expressions (loss.py:104-137)."""
    import torch
    from capf.lib import pose_errors, segment_sums
    rng = np.random.default_rng(9)
    n = 50000
    gt = (rng.standard_normal((n, 17, 3)) * 0.3).astype(np.float32)
    pred = (gt * rng.uniform(0.7, 1.3, (n, 1, 1)) + rng.standard_normal((n, 17, 3)) * 0.03 + rng.standard_normal((n, 1, 3)) * 0.1).astype(np.float32)
    err = pose_errors(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda())
    e = err.cpu().numpy()
    np.testing.assert_allclose(e[:, 0], mo.mpjpe_per_pose(pred, gt), rtol=3e-6)
    np.testing.assert_allclose(e[:, 1], mo.p_mpjpe_per_pose(pred, gt), rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(e[:, 2], mo.n_mpjpe_per_pose(pred, gt), rtol=3e-6)
    np.testing.assert_allclose(e[1:, 3], mo.velocity_errors(pred, gt), rtol=3e-6)
    sums, counts = segment_sums(err)
    assert counts.cpu().tolist() == [[n, n - 1]]
    np.testing.assert_allclose(sums.cpu().numpy()[0, :3], e[:, :3].astype(np.float64).sum(0), rtol=1e-12)
    from mvn.models.loss import KeypointsMAELoss, KeypointsMSELoss, KeypointsMSESmoothLoss
    p = torch.from_numpy(pred[:64]).cuda().requires_grad_(True)
    t = torch.from_numpy(gt[:64]).cuda()
    v = (torch.rand(64, 17, 1, device="cuda") > 0.3).float()
    for cls, ref in ((KeypointsMSELoss(), lambda d: d), (KeypointsMAELoss(), None), (KeypointsMSESmoothLoss(0.02), "smooth")):
        p.grad = None
        cls(p, t, v).backward()
        got = p.grad.clone()
        q = p.detach().clone().requires_grad_(True)
        if ref is None:
            want_loss = (torch.abs(t - q) * v).sum() / (3 * max(1, v.sum().item()))
        else:
            diff = (t - q) ** 2 * v
            if ref == "smooth":
                diff = torch.where(diff > 0.02, torch.pow(diff.clamp_min(1e-30), 0.1) * (0.02 ** 0.9), diff)
            want_loss = diff.sum() / (3 * max(1, v.sum().item()))
        want_loss.backward()
        assert (got - q.grad).abs().max().item() <= 2e-6 * q.grad.abs().max().item() + 1e-9
