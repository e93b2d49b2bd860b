"""Losses and metrics the path's callers import (`from mvn.models.loss import MPJPE, KeypointsMSELoss,
KeypointsMSESmoothLoss, KeypointsMAELoss`, ContextPose/train.py:21; P_MPJPE / N_MPJPE / MPJVE,
ContextPose/mvn/datasets/human36m.py:365-368).  Same class names, argument meaning and return kinds as
ContextPose/mvn/models/loss.py; the arithmetic runs in libcapf.so (capf_mpjpe, capf_pose_errors, capf_segment_sums,
capf_keypoints_loss).  No torch / numpy compute and no CPU fallback: numpy arguments (the reference hands
`.cpu().numpy()` arrays to P_MPJPE and MPJVE) are uploaded, reduced on the GPU and returned as numpy scalars."""
import ctypes

import numpy as np
import torch
from torch import nn


def _cuda(x):
    """numpy array or tensor -> contiguous fp32 CUDA tensor (device of the tensor, else the current device)."""
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    if not x.is_cuda:
        from capf.lib import CapfError
        raise CapfError("metrics run on the MI355X only: got a CPU tensor (pass a CUDA tensor or a numpy array)")
    return x.detach().to(torch.float32).contiguous()


def _poses(x):
    t = _cuda(x)
    return t.reshape(-1, t.shape[-2], 3)


class _MPJPEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        from capf.lib import load_library
        lib = load_library()
        pred_c, gt_c = pred.contiguous(), gt.contiguous()
        dim = pred_c.shape[-1]
        rows = pred_c.numel() // dim
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred_c) if ctx.needs_input_grad[0] else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
        if dim == 3:
            rc = lib.capf_mpjpe(stream, P(pred_c), P(gt_c), rows, P(loss), P(dpred), 1.0)
        else:
            rc = lib.capf_mpjpe_nd(stream, P(pred_c), P(gt_c), rows, dim, P(loss), P(dpred), 1.0)
        if rc:
            raise RuntimeError(f"capf_mpjpe failed ({rc})")
        ctx.dpred = dpred
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.dpred * g if ctx.dpred is not None else None), None


class MPJPE(nn.Module):
    """loss.py:16-22: mean over every leading axis of ||pred - gt||_2 along the last one (any width: 3-D joints, 2-D keypoints).
    CUDA tensors only — there is no CPU fallback; other float dtypes are computed in fp32 and the loss is cast back to the inputs' dtype."""

    def forward(self, keypoints_pred, keypoints_gt):
        assert keypoints_pred.shape == keypoints_gt.shape
        if not (keypoints_pred.is_cuda and keypoints_gt.is_cuda):
            from capf.lib import CapfError
            raise CapfError("MPJPE runs on the MI355X only: got a CPU tensor (no CPU fallback)")
        dt = torch.promote_types(keypoints_pred.dtype, keypoints_gt.dtype)
        if keypoints_pred.dtype != torch.float32 or keypoints_gt.dtype != torch.float32:
            keypoints_pred, keypoints_gt = keypoints_pred.float(), keypoints_gt.float()
        loss = _MPJPEFn.apply(keypoints_pred, keypoints_gt)
        return loss if dt == torch.float32 else loss.to(dt)        # the reference returns the inputs' dtype (torch.norm / mean)


def _set_mean(pred, gt, column, per_pair=False):
    """mean of one per-pose error column over all poses (fp64 on the device, fixed order)."""
    from capf.lib import pose_errors, segment_sums
    p, g = _poses(pred), _poses(gt)
    sums, counts = segment_sums(pose_errors(p, g))
    sums, counts = sums.cpu().numpy(), counts.cpu().numpy()
    n = counts[0, 1] if per_pair else counts[0, 0]
    return sums[0, column] / n if n > 0 else float("nan")


class P_MPJPE(nn.Module):
    """loss.py:25-68 — MPJPE after similarity alignment ("Protocol #2").  Arguments [N,17,3] (numpy, as
    human36m.py:374 passes them, or CUDA tensors); returns a numpy scalar like the reference."""

    def forward(self, keypoints_pred, keypoints_gt):
        assert keypoints_pred.shape == keypoints_gt.shape
        return np.float32(_set_mean(keypoints_pred, keypoints_gt, 1))


class N_MPJPE(nn.Module):
    """loss.py:71-84 — scale-normalised MPJPE on [B,1,17,3] tensors; returns a 0-d tensor like the reference."""

    def forward(self, keypoints_pred, keypoints_gt):
        assert keypoints_pred.shape == keypoints_gt.shape
        v = _set_mean(keypoints_pred, keypoints_gt, 2)
        dev = keypoints_pred.device if isinstance(keypoints_pred, torch.Tensor) else "cpu"
        return torch.tensor(v, dtype=torch.float32, device=dev)


class MPJVE(nn.Module):
    """loss.py:87-101 — mean per-joint velocity error over consecutive rows of [N,17,3]."""

    def forward(self, keypoints_pred, keypoints_gt):
        assert keypoints_pred.shape == keypoints_gt.shape
        return np.float32(_set_mean(keypoints_pred, keypoints_gt, 3, per_pair=True))


class _KeypointsLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, validity, mode, threshold):
        from capf.lib import keypoints_loss
        loss, dpred = keypoints_loss(mode, pred, gt, validity, threshold, want_grad=ctx.needs_input_grad[0])
        ctx.dpred = dpred
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.dpred * g if ctx.dpred is not None else None), None, None, None, None


class _KeypointsLoss(nn.Module):
    mode = 0

    def forward(self, keypoints_pred, keypoints_gt, keypoints_binary_validity):
        if not (keypoints_pred.is_cuda and keypoints_pred.dtype == torch.float32):
            from capf.lib import CapfError
            raise CapfError("keypoint losses run on the MI355X only (no CPU fallback)")
        return _KeypointsLossFn.apply(keypoints_pred, keypoints_gt, keypoints_binary_validity, self.mode,
                                      float(getattr(self, "threshold", 0.0)))


class KeypointsMSELoss(_KeypointsLoss):
    """loss.py:104-112"""
    mode = 0


class KeypointsMSESmoothLoss(_KeypointsLoss):
    """loss.py:115-126"""
    mode = 1

    def __init__(self, threshold=400):
        super().__init__()
        self.threshold = threshold


class KeypointsMAELoss(_KeypointsLoss):
    """loss.py:129-137"""
    mode = 2
