"""Per-action evaluation of gathered predictions — the part of `Human36MMultiViewDataset` the training script
calls after validation (ContextPose/mvn/datasets/human36m.py:358-435 `evaluate_using_pred` / `evaluate`;
consumed at ContextPose/train.py:381-436).  The dataset itself (labels, images, OpenCV crops) is out of scope
(DESIGN.md §7); what is here is the arithmetic: per-pose MPJPE / P-MPJPE / MPJVE on the GPU (capf_pose_errors),
per-action fp64 sums (capf_segment_sums), and the reference's merging of the '-1' / '-2' trials of an action."""
import numpy as np
import torch

from capf import lib as _capf


def previous_in_segment(segment):
    """prev[i] = largest j < i with segment[j] == segment[i], else -1: the row np.diff pairs row i with once the
    rows of one action have been selected by a boolean mask (human36m.py:371-376, loss.py:98-99)."""
    segment = np.asarray(segment)
    prev = np.full(segment.shape[0], -1, np.int32)
    last = {}
    for i, a in enumerate(segment.tolist()):
        prev[i] = last.get(a, -1)
        last[a] = i
    return prev


def evaluate_using_pred(keypoints_gt, keypoints_3d_predicted, labels_action_idx, action_names):
    """human36m.py:358-417.  keypoints_* [N,1,17,3] (CUDA tensors or numpy), labels_action_idx [N] ints,
    action_names: list of 'Name-1' / 'Name-2' strings indexed by action idx.
    Returns {action (trials merged): {'MPJPE','P_MPJPE','MPJVE'}} exactly like the reference."""
    if tuple(keypoints_3d_predicted.shape) != tuple(keypoints_gt.shape):
        raise ValueError('`keypoints_3d_predicted` shape should be %s, got %s' %
                         (tuple(keypoints_gt.shape), tuple(keypoints_3d_predicted.shape)))       # human36m.py:422-425
    from mvn.models.loss import _poses
    pred, gt = _poses(keypoints_3d_predicted), _poses(keypoints_gt)
    seg_host = np.asarray(labels_action_idx).astype(np.int32)
    dev = pred.device
    seg = torch.from_numpy(seg_host).to(dev)
    prev = torch.from_numpy(previous_in_segment(seg_host)).to(dev)
    err = _capf.pose_errors(pred, gt, prev)
    sums, counts = _capf.segment_sums(err, seg, prev, n_segments=len(action_names))
    sums, counts = sums.cpu().numpy(), counts.cpu().numpy()

    # An action with no frames in the evaluated subset: the reference's e1 / e2 / ev on empty arrays are NaN (mean of nothing) and
    # the NaN propagates into the merged action; values stay numpy scalars so n == 0 divides to NaN instead of raising.
    nan = np.float64('nan')
    action_scores = {}
    for a, name in enumerate(action_names):
        n, pairs = int(counts[a, 0]), int(counts[a, 1])
        # the reference stores frame_count * mean(...) per action (:373-376); MPJVE's mean runs over n-1 pairs
        action_scores[name] = {'MPJPE': np.float64(sums[a, 0]) if n > 0 else nan, 'P_MPJPE': np.float64(sums[a, 1]) if n > 0 else nan,
                               'MPJVE': np.float64(n * (sums[a, 3] / pairs)) if pairs > 0 else nan, 'frame_count': n}
    for base in [name[:-2] for name in action_names if name.endswith('-1')]:                   # :386-406
        combined = {'MPJPE': np.float64(0.0), 'P_MPJPE': np.float64(0.0), 'MPJVE': np.float64(0.0), 'frame_count': 0}
        for trial in 1, 2:
            key = '%s-%d' % (base, trial)
            for k in combined:
                combined[k] = combined[k] + action_scores[key][k]
            del action_scores[key]
        action_scores[base] = combined
    with np.errstate(divide='ignore', invalid='ignore'):
        for k in action_scores:                                                               # :408-415
            n = np.float64(action_scores[k]['frame_count'])
            action_scores[k] = {'MPJPE': action_scores[k]['MPJPE'] / n, 'P_MPJPE': action_scores[k]['P_MPJPE'] / n,
                                'MPJVE': action_scores[k]['MPJVE'] / n}
    return action_scores
