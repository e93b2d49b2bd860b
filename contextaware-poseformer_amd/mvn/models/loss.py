"""Training loss / metric of the hot path.  MPJPE restates ContextPose/mvn/models/loss.py:16-22."""
import torch
from torch import nn


class MPJPE(nn.Module):
    """mean over (B, 1, 17) of the L2 norm of (pred - gt) along the last axis."""

    def forward(self, keypoints_pred, keypoints_gt):
        assert keypoints_pred.shape == keypoints_gt.shape
        return torch.mean(torch.norm(keypoints_pred - keypoints_gt, dim=len(keypoints_gt.shape) - 1))
