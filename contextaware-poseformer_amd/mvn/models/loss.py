"""Training loss / metric of the hot path.  MPJPE restates ContextPose/mvn/models/loss.py:16-22:
mean over (B, 1, 17) of the L2 norm of (pred - gt) along the last axis.  On the GPU both the loss and its
gradient come from one native kernel (capf_mpjpe); CPU tensors use the plain torch expression."""
import ctypes

import torch
from torch import nn


class _MPJPEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        from capf.lib import load_library
        lib = load_library()
        pred_c, gt_c = pred.contiguous(), gt.contiguous()
        rows = pred_c.numel() // 3
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred_c)
        stream = ctypes.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream)
        rc = lib.capf_mpjpe(stream, ctypes.c_void_p(pred_c.data_ptr()), ctypes.c_void_p(gt_c.data_ptr()), rows,
                            ctypes.c_void_p(loss.data_ptr()), ctypes.c_void_p(dpred.data_ptr()), 1.0)
        if rc:
            raise RuntimeError(f"capf_mpjpe failed ({rc})")
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None


class MPJPE(nn.Module):
    def forward(self, keypoints_pred, keypoints_gt):
        assert keypoints_pred.shape == keypoints_gt.shape
        if keypoints_pred.is_cuda and keypoints_pred.dtype == torch.float32 and keypoints_pred.shape[-1] == 3:
            return _MPJPEFn.apply(keypoints_pred, keypoints_gt)
        return torch.mean(torch.norm(keypoints_pred - keypoints_gt, dim=len(keypoints_gt.shape) - 1))
