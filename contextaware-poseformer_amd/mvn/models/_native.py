"""Glue between the PyTorch host modules and libcapf.so.

* `make_capf_config` translates the reference's config tree (mvn/utils/cfg.py) into the C struct.
* `ParamTree` materialises the library's parameter schema (== the reference's state_dict names and
  shapes) as real nn.Conv2d / nn.BatchNorm2d / nn.Linear / nn.LayerNorm leaf modules, so
  load_state_dict(strict=True) of reference checkpoints, .parameters(), .eval(), DDP wrapping and
  SyncBatchNorm.convert_sync_batchnorm behave as they do on the reference model.  The leaf modules
  only HOLD parameters; their torch forward is never called — compute happens in the library.
"""
import math

import torch
from torch import nn

from capf.lib import CPN50, HRNET, CapfConfig, CapfError, Engine

MAX_BATCH = 8192


def make_capf_config(config, height=256, width=192, context_blocks=True, compute_dtype="fp32", plan_flags=0):
    bb = config.model.backbone
    pf = config.model.poseformer
    c = CapfConfig()
    if bb.type in ("hrnet_32", "hrnet_48"):
        c.backbone = HRNET
        for stage, want in (("STAGE2", 2), ("STAGE3", 3), ("STAGE4", 4)):
            st = bb[stage]
            # same consistency checks as HighResolutionModule._check_branches (pose_hrnet.py:159-175)
            if st.NUM_BRANCHES != len(st.NUM_BLOCKS):
                raise ValueError("NUM_BRANCHES({}) <> NUM_BLOCKS({})".format(st.NUM_BRANCHES, len(st.NUM_BLOCKS)))
            if st.NUM_BRANCHES != len(st.NUM_CHANNELS):
                raise ValueError("NUM_BRANCHES({}) <> NUM_CHANNELS({})".format(st.NUM_BRANCHES, len(st.NUM_CHANNELS)))
            if st.NUM_BRANCHES != want or st.BLOCK != "BASIC" or st.FUSE_METHOD != "SUM":
                raise ValueError("{}: only the HRNet pose topology (2/3/4 BASIC branches, SUM fuse) is supported".format(stage))
        chans = list(bb.STAGE4.NUM_CHANNELS)
        if list(bb.STAGE2.NUM_CHANNELS) != chans[:2] or list(bb.STAGE3.NUM_CHANNELS) != chans[:3]:
            raise ValueError("stage channel lists must be prefixes of STAGE4.NUM_CHANNELS")
        blocks = set(bb.STAGE2.NUM_BLOCKS) | set(bb.STAGE3.NUM_BLOCKS) | set(bb.STAGE4.NUM_BLOCKS)
        if len(blocks) != 1:
            raise ValueError("NUM_BLOCKS must be uniform")
        for i in range(4):
            c.hr_channels[i] = chans[i]
        for i, stage in enumerate(("STAGE2", "STAGE3", "STAGE4")):
            c.hr_modules[i] = bb[stage].NUM_MODULES
        c.hr_blocks = blocks.pop()
    elif bb.type == "cpn":
        c.backbone = CPN50
    else:
        raise ValueError("unknown backbone type {!r}".format(bb.type))
    c.base_dim = pf.base_dim
    c.embed_dim_ratio = pf.embed_dim_ratio
    c.levels = pf.levels            # depth = config.levels (pose_dformer.py:169)
    # the sibling app's PoseTransformer builds config.depth blocks per group instead (ContextPose_mpi/model/pose_dformer.py:199)
    c.depth = 0 if context_blocks else int(getattr(pf, "depth", pf.levels))
    c.num_joints = 17
    c.num_heads = 8
    c.deform_heads = 4
    c.deform_samples = 4
    c.context_blocks = 1 if context_blocks else 0
    if compute_dtype not in ("fp32", "bf16"):
        raise ValueError("compute_dtype must be 'fp32' or 'bf16'")
    c.compute_dtype = 1 if compute_dtype == "bf16" else 0
    c.max_batch = MAX_BATCH
    c.height, c.width = height, width
    # workspace also holds what capf_backward needs (6.4 MB/frame); the training path is built for depth == levels only
    c.training = 0 if c.depth not in (0, c.levels) else 1
    c.plan_flags = int(plan_flags)  # 0 = the product plan (capf.lib.PLAN_*: take a kernel family out, parity tests only)
    return c


class Container(nn.Module):
    """Plain named container.  Numeric children behave like an nn.ModuleList (the reference keeps its block lists
    in nn.ModuleList, pose_dformer.py:189-203): integer and negative indices, slices, iteration, len."""

    def _numeric(self):
        return all(k.isdigit() for k in self._modules)

    def __getitem__(self, i):
        if isinstance(i, slice):
            out = Container()
            for k, m in list(self._modules.items())[i]:
                out.add_module(k, m)
            return out
        if isinstance(i, int):
            n = len(self._modules)
            if not -n <= i < n:
                raise IndexError("index {} is out of range".format(i))
            if i < 0:
                i += n
            if str(i) in self._modules:
                return self._modules[str(i)]
            return list(self._modules.values())[i]
        return self._modules[str(i)]

    def __iter__(self):
        return iter(self._modules.values())

    def __len__(self):
        return len(self._modules)


def _leaf_for(kinds_shapes):
    """Pick the nn leaf module that owns this group of schema entries."""
    kinds = {k: s for k, s in kinds_shapes}
    if "conv_w" in kinds:
        co, ci, kh, kw = kinds["conv_w"]
        return nn.Conv2d(ci, co, (kh, kw), padding=(kh // 2, kw // 2), bias=False)
    if "bn_w" in kinds:
        return nn.BatchNorm2d(kinds["bn_w"][0], momentum=0.1)
    if "lin_w" in kinds:
        n, k = kinds["lin_w"]
        return nn.Linear(k, n, bias="lin_b" in kinds)
    if "ln_w" in kinds:
        return nn.LayerNorm(kinds["ln_w"][0])
    raise CapfError("schema group without a known leaf: {}".format(kinds))


def build_param_tree(root, schema, prefix):
    """Attach to `root` every schema entry whose name starts with `prefix.`."""
    groups = {}
    for name, shape, kind in schema:
        if not name.startswith(prefix + "."):
            continue
        rel = name[len(prefix) + 1:]
        if kind == "raw":
            parent = _descend(root, rel.split(".")[:-1])
            parent.register_parameter(rel.split(".")[-1], nn.Parameter(torch.zeros(*shape)))
            continue
        path = rel.rsplit(".", 1)[0]
        groups.setdefault(path, []).append((kind, shape))
    for path, ks in groups.items():
        parts = path.split(".")
        parent = _descend(root, parts[:-1])
        parent.add_module(parts[-1], _leaf_for(ks))
    return root


def _descend(root, parts):
    node = root
    for p in parts:
        if p not in node._modules:
            node.add_module(p, Container())
        node = node._modules[p]
    return node


def init_deformable_blocks(volume_net, heads=4, samples=4):
    """DeformableBlock._reset_parameters semantics (pose_dformer.py:103-113): zero offset / attention
    matrices, offsets biased to `samples` steps of 0.01 along `heads` compass directions."""
    if "context_blocks" not in volume_net._modules:
        return
    theta = torch.arange(heads, dtype=torch.float32) * (2.0 * math.pi / heads)
    d = torch.stack([theta.cos(), theta.sin()], -1)
    d = 0.01 * d / d.abs().max(-1, keepdim=True)[0]
    grid = d.view(heads, 1, 2) * torch.arange(1, samples + 1, dtype=torch.float32).view(1, samples, 1)
    with torch.no_grad():
        for blk in volume_net.context_blocks._modules.values():
            blk.sampling_offsets.weight.zero_()
            blk.sampling_offsets.bias.copy_(grid.reshape(-1))
            blk.attention_weights.weight.zero_()
            blk.attention_weights.bias.zero_()


def init_cpn_convs(backbone):
    """networks/resnet.py:111-117, globalNet.py:19-27: conv ~ N(0, sqrt(2/(k*k*Cout))), BN gamma 1 beta 0."""
    for m in backbone.modules():
        if isinstance(m, nn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / n))
