"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, float64 inside) of the evaluation metrics next to the hot
path.  Each function cites the reference lines it follows; tests/test_metrics.py pins it against
tests/golden/losses.npz, which oracle/make_goldens.py captured from the REAL ContextPose/mvn/models/loss.py.
Only tests/ may import this module."""
import numpy as np


def mpjpe_per_pose(pred, gt):
    """loss.py:16-22 before the outer mean: [N,J,3] -> [N] mean over joints of the L2 distance."""
    return np.linalg.norm(pred.astype(np.float64) - gt.astype(np.float64), axis=-1).mean(-1)


def p_mpjpe_per_pose(pred, gt):
    """loss.py:25-68 (SVD form, as the reference writes it), per pose."""
    X, Y = gt.astype(np.float64), pred.astype(np.float64)
    muX, muY = X.mean(1, keepdims=True), Y.mean(1, keepdims=True)                    # :36-37
    X0, Y0 = X - muX, Y - muY
    normX = np.sqrt((X0 ** 2).sum((1, 2), keepdims=True))                              # :42-43
    normY = np.sqrt((Y0 ** 2).sum((1, 2), keepdims=True))
    X0, Y0 = X0 / normX, Y0 / normY
    H = np.matmul(X0.transpose(0, 2, 1), Y0)                                           # :48
    U, s, Vt = np.linalg.svd(H)
    V = Vt.transpose(0, 2, 1)
    R = np.matmul(V, U.transpose(0, 2, 1))
    sign = np.sign(np.expand_dims(np.linalg.det(R), 1))                                # :54-58
    V[:, :, -1] *= sign
    s[:, -1] *= sign.flatten()
    R = np.matmul(V, U.transpose(0, 2, 1))
    tr = np.expand_dims(s.sum(1, keepdims=True), 2)
    a = tr * normX / normY                                                             # :62
    t = muX - a * np.matmul(muY, R)
    aligned = a * np.matmul(Y, R) + t                                                  # :66
    return np.linalg.norm(aligned - X, axis=-1).mean(-1)


def p_mpjpe_horn_per_pose(pred, gt):
    """The SAME quantity by the route the HIP kernel takes (csrc/metrics.hip): Horn's quaternion closed form — top
    eigenpair of the symmetric 4x4 matrix built from S = sum_j Y0_j X0_j^T — so that the algebra of the kernel is
    checked against the SVD form on the CPU, not only on the GPU."""
    X, Y = gt.astype(np.float64), pred.astype(np.float64)
    out = np.empty(X.shape[0])
    for i in range(X.shape[0]):
        muX, muY = X[i].mean(0), Y[i].mean(0)
        x0, y0 = X[i] - muX, Y[i] - muY
        nX, nY = np.sqrt((x0 ** 2).sum()), np.sqrt((y0 ** 2).sum())
        S = (y0.T @ x0) / (nX * nY)
        N = np.array([
            [S[0, 0] + S[1, 1] + S[2, 2], S[1, 2] - S[2, 1], S[2, 0] - S[0, 2], S[0, 1] - S[1, 0]],
            [S[1, 2] - S[2, 1], S[0, 0] - S[1, 1] - S[2, 2], S[0, 1] + S[1, 0], S[2, 0] + S[0, 2]],
            [S[2, 0] - S[0, 2], S[0, 1] + S[1, 0], -S[0, 0] + S[1, 1] - S[2, 2], S[1, 2] + S[2, 1]],
            [S[0, 1] - S[1, 0], S[2, 0] + S[0, 2], S[1, 2] + S[2, 1], -S[0, 0] - S[1, 1] + S[2, 2]]])
        lam, vec = np.linalg.eigh(N)
        w, x, y, z = vec[:, -1]
        R = np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])
        a = lam[-1] * nX / nY
        aligned = a * (y0 @ R.T) + muX
        out[i] = np.linalg.norm(aligned - X[i], axis=-1).mean()
    return out


def n_mpjpe_per_pose(pred, gt):
    """loss.py:71-84 per pose: scale = mean_j(sum_c gt*pred) / mean_j(sum_c pred^2)."""
    X, Y = gt.astype(np.float64), pred.astype(np.float64)
    scale = (X * Y).sum(-1).mean(-1) / (Y ** 2).sum(-1).mean(-1)
    return np.linalg.norm(scale[:, None, None] * Y - X, axis=-1).mean(-1)


def velocity_errors(pred, gt):
    """loss.py:87-101 before the mean: [N,J,3] -> [N-1] (float32 first differences, like np.diff on fp32 arrays)."""
    vp, vg = np.diff(pred, axis=0), np.diff(gt, axis=0)
    return np.linalg.norm((vp - vg).astype(np.float64), axis=-1).mean(-1)


def per_action(pred, gt, action_idx, n_actions):
    """human36m.py:370-383: [n_actions, 4] = frame_count * {MPJPE, P_MPJPE, MPJVE}, frame_count."""
    out = np.zeros((n_actions, 4))
    for a in range(n_actions):
        m = action_idx == a
        n = np.count_nonzero(m)
        out[a] = [mpjpe_per_pose(pred[m], gt[m]).sum(), p_mpjpe_per_pose(pred[m], gt[m]).sum(),
                  n * velocity_errors(pred[m], gt[m]).mean(), n]
    return out


def keypoints_loss(mode, pred, gt, validity, threshold=0.0):
    """loss.py:104-137: mode 0 MSE, 1 MSESmooth, 2 MAE.  pred/gt [...,D], validity [...,1]."""
    p, g, v = pred.astype(np.float64), gt.astype(np.float64), validity.astype(np.float64)
    D = p.shape[-1]
    if mode == 2:
        diff = np.abs(g - p) * v
    else:
        diff = (g - p) ** 2 * v
        if mode == 1:
            big = diff > threshold
            diff[big] = diff[big] ** 0.1 * threshold ** 0.9
    return diff.sum() / (D * max(1.0, v.sum()))
