"""GPU input preprocessing in front of the hot path: a mirror of `data_prefetcher`
(ContextPose/mvn/datasets/utils.py:15-89) — same constructor, same `.next()` contract, same side-stream
overlap — whose dozen elementwise torch ops are ONE native launch pair (capf_preprocess): uint8 BGR crop ->
normalised fp32 RGB NHWC, root-relative ground truth, train-time horizontal flip with left/right joint swap,
flip-test stacking.  The dataset / OpenCV side of the reference (human36m.py) is out of scope; any iterable
yielding (uint8 images [B,256,192,3], gt [B,1,17,3], k2d [B,17,2], kcrop [B,17,2]) CPU batches works."""
import random

import torch

from capf import lib as _capf

joints_left = [4, 5, 6, 11, 12, 13]
joints_right = [1, 2, 3, 14, 15, 16]


class data_prefetcher():
    def __init__(self, loader, device, is_train, flip_test, backbone):
        self.loader = iter(loader)
        self.stream = torch.cuda.Stream(device=device)
        self.device = device
        self.is_train = is_train
        self.flip_test = flip_test
        self.backbone = backbone
        self.preload()

    def preload(self):
        try:
            batch = next(self.loader)
        except StopIteration:
            self.next_batch = None
            return
        with torch.cuda.stream(self.stream):
            images, gt, k2d, kcrop = [t.to(self.device, non_blocking=True) for t in batch]
            if self.is_train and random.random() <= 0.5:              # utils.py:55
                mode = 1
            elif (not self.is_train) and self.flip_test:               # :67
                mode = 2
            else:
                mode = 0
            img, gt_o, k2d_o, kc_o = _capf.preprocess(images, gt.float(), k2d.float(), kcrop.float(), self.backbone, mode)
            if mode == 2:     # the reference stacks on dim 1; [2,B,...] transposed is the same view without a copy
                img, k2d_o, kc_o = img.transpose(0, 1), k2d_o.transpose(0, 1), kc_o.transpose(0, 1)
            self.next_batch = [img, gt_o, k2d_o, kc_o]

    def next(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        batch = self.next_batch
        if batch is not None:
            for t in batch:
                t.record_stream(torch.cuda.current_stream())
        self.preload()
        return batch
