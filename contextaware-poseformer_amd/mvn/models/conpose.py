"""CA_PF — drop-in for ContextPose/mvn/models/conpose.py:10-42 running on libcapf.so.

Same constructor (`CA_PF(config, device)`), same attributes (`.backbone`, `.volume_net`), same
state_dict names/shapes, same call

    model(images[B,H,W,3] fp32 NHWC, keypoints_2d_cpn[B,17,2], keypoints_2d_cpn_crop[B,17,2]) -> [B,1,17,3]

including the reference's in-place normalisation of the third argument (conpose.py:34-35).  The
compute is one capf_forward call: hand-written gfx950 kernels, enqueued on torch's current stream.
There is no eager / CPU fallback: CPU tensors or a missing libcapf.so raise.
"""
import torch
from torch import nn

from capf.lib import CapfError, Engine

from . import _native


class CA_PF(nn.Module):
    def __init__(self, config, device="cuda:0"):
        super().__init__()
        self.num_joints = config.model.backbone.num_joints
        self._config = config
        self._backbone_type = config.model.backbone.type
        # schema from a plan-only handle (no GPU needed): names == reference state_dict
        plan = Engine(_native.make_capf_config(config, 256, 192), device=None)
        schema = plan.schema()
        plan.close()
        self.backbone = _native.build_param_tree(_native.Container(), schema, "backbone")
        self.volume_net = _native.build_param_tree(_native.Container(), schema, "volume_net")
        if self._backbone_type == "cpn":
            _native.init_cpn_convs(self.backbone)
        _native.init_deformable_blocks(self.volume_net)

        if config.model.backbone.fix_weights:
            print("model backbone weights are fixed")
            for p in self.backbone.parameters():
                p.requires_grad = False

        self._engines = {}          # (device index, H, W) -> Engine
        self._dirty = True          # parameters (re)loaded / moved since the last pack
        self._lifter_versions = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._mark_dirty())

    # ---- parameter-change tracking -----------------------------------------------------------
    def _mark_dirty(self):
        self._dirty = True

    def _apply(self, fn, *args, **kwargs):
        self._dirty = True
        return super()._apply(fn, *args, **kwargs)

    def params_changed(self):
        """Tell the engine that parameter VALUES changed in place (optimizer.step on backbone
        weights, manual .data edits).  volume_net parameters are tracked automatically."""
        self._dirty = True

    def _engine(self, images):
        dev = images.device
        if dev.type != "cuda":
            raise CapfError("CA_PF runs on an MI355X only: inputs are on {} (no CPU fallback)".format(dev))
        B, H, W, C = images.shape
        if C != 3:
            raise ValueError("images must be [B,H,W,3] NHWC")
        key = (dev.index, H, W)
        eng = self._engines.get(key)
        if eng is None:
            eng = Engine(_native.make_capf_config(self._config, H, W), device=dev.index)
            eng._packed_versions = None
            self._engines[key] = eng
        lifter_versions = tuple(p._version for p in self.volume_net.parameters())
        if self._dirty or eng._packed_versions != lifter_versions or not eng._bound:
            if self.backbone.training and any(p.requires_grad for p in self.backbone.parameters()):
                raise NotImplementedError("training-mode BatchNorm / backbone gradients are outside the hot "
                                          "path: freeze the backbone (fix_weights) and call backbone.eval()")
            state = self.state_dict(keep_vars=True)
            stream = torch.cuda.current_stream(dev).cuda_stream
            if self._dirty or not eng._bound:
                eng._bound = {}
                eng.bind_state({k: v.data for k, v in state.items()}, stream)
            else:
                eng.params_changed(stream)
            eng._packed_versions = lifter_versions
            if all(e._packed_versions == lifter_versions and e._bound for e in self._engines.values()):
                self._dirty = False
        return eng

    # ---- conpose.py:30-42 --------------------------------------------------------------------
    def forward(self, images, keypoints_2d_cpn, keypoints_2d_cpn_crop):
        if images.dtype != torch.float32:
            raise TypeError("images must be float32")
        images = images.contiguous()
        eng = self._engine(images)
        k2d = keypoints_2d_cpn.contiguous()
        if not keypoints_2d_cpn_crop.is_contiguous():
            raise ValueError("keypoints_2d_cpn_crop is normalised in place and must be contiguous")
        B = images.shape[0]
        out = torch.empty(B, 1, self.num_joints, 3, dtype=torch.float32, device=images.device)
        stream = torch.cuda.current_stream(images.device).cuda_stream
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.volume_net.parameters()):
            raise NotImplementedError("lifter backward is not built yet: wrap inference in torch.no_grad()")
        eng.forward(images, k2d, keypoints_2d_cpn_crop, out, stream)
        return out

    def engine_for(self, images):
        """The native engine serving inputs of this shape/device (tests, bench)."""
        return self._engine(images.contiguous())
