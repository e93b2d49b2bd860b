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


class _LifterStep(torch.autograd.Function):
    """Autograd boundary of the native training step: forward = capf_forward_train, backward =
    capf_backward into one flat gradient buffer whose slices are returned as the parameter gradients
    (so torch optimizers, DDP hooks and `capf.dist.allreduce_mean_` all see ordinary .grad tensors)."""

    @staticmethod
    def forward(ctx, owner, eng, images, k2d, kcrop, masks, names, *params):
        out = torch.empty(images.shape[0], 1, owner.num_joints, 3, dtype=torch.float32, device=images.device)
        stream = torch.cuda.current_stream(images.device).cuda_stream
        eng.forward_train(images, k2d, kcrop, out, stream, masks)
        ctx.token = eng.train_generation()
        ctx.eng, ctx.masks, ctx.names, ctx.owner = eng, masks, names, owner
        ctx.keep = (k2d, kcrop)     # capf_backward re-reads the keypoints / normalised ref: keep them alive
        ctx.shapes = [p.shape for p in params]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.eng
        if eng.train_generation() != ctx.token:
            raise CapfError("backward of a CA_PF forward whose saved activations were overwritten by a later forward on the "
                            "same engine (the native step keeps ONE set of activations in its workspace): call backward "
                            "before the next forward of this model")
        layout, total = eng.grad_layout_cached()
        flat = torch.empty(total, dtype=torch.float32, device=grad_out.device)
        stream = torch.cuda.current_stream(grad_out.device).cuda_stream
        eng.backward(grad_out.contiguous(), flat, stream, ctx.masks)
        ctx.owner.last_flat_grad = flat
        if ctx.owner.flat_grad_only:      # the caller consumes last_flat_grad itself (capf.optim.FusedAdamW + capf.dist)
            return (None,) * (7 + len(ctx.names))
        grads = tuple(flat[layout[n][0]: layout[n][0] + layout[n][1]].view(shp) for n, shp in zip(ctx.names, ctx.shapes))
        return (None,) * 7 + grads


class CA_PF(nn.Module):
    def __init__(self, config, device="cuda:0", compute_dtype="fp32", context_blocks=True, plan_flags=0):
        """compute_dtype: 'fp32' (exact fp32 MFMA, the reference's precision) or 'bf16' (backbone convolutions on
        bf16 MFMA with bf16 activations and fp32 accumulation; the lifter stays fp32) — an extension of the
        reference signature for BASELINE.json's bf16 configurations."""
        super().__init__()
        self.compute_dtype = compute_dtype
        self.plan_flags = plan_flags               # capf.lib.PLAN_* bits: parity tests compare kernel families; 0 = product plan
        self.context_blocks = context_blocks       # False: the MPI-INF-3DHP variant (model/conpose.py)
        self.num_joints = config.model.backbone.num_joints
        self._config = config
        self._backbone_type = config.model.backbone.type
        # schema from a plan-only handle (no GPU needed): names == reference state_dict
        plan = Engine(_native.make_capf_config(config, 256, 192, context_blocks=context_blocks), device=None)
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

        self.drop_path_rate = 0.2   # PoseTransformer(drop_path_rate=0.2), dpr = linspace(0, rate, levels) (pose_dformer.py:147,187)
        self.last_flat_grad = None  # flat fp32 gradient of volume_net.* written by the last backward
        # True: backward leaves the gradient in last_flat_grad ONLY and sets no .grad (autograd would clone each of the 191 slices
        # into its parameter's .grad: 191 small device copies per step that a flat optimizer never reads; they overlap the rest of
        # the step -- no measurable change of a 512-frame step, the point is not to materialise a second copy of the gradient)
        self.flat_grad_only = False
        self._engines = {}          # (device index, H, W) -> Engine
        # Parameter-change tracking.  `_generation` counts events that may have replaced parameter STORAGE or frozen
        # (backbone) VALUES: load_state_dict on the model or either child, .to()/._apply, params_changed().  Every engine
        # remembers the generation it last bound at (one shared flag would be cleared by the first engine that rebinds
        # and leave the engines of other input resolutions stale).  volume_net parameters are additionally fingerprinted
        # per call by (data_ptr, _version): optimizer steps bump versions, capf.optim.flatten_ / `p.data = ...` move
        # storage without bumping anything else.
        self._generation = 0
        hook = lambda module, incompatible: self._mark_dirty()
        self.register_load_state_dict_post_hook(hook)
        self.backbone.register_load_state_dict_post_hook(hook)
        self.volume_net.register_load_state_dict_post_hook(hook)

    # ---- parameter-change tracking -----------------------------------------------------------
    def _mark_dirty(self):
        self._generation += 1

    def _apply(self, fn, *args, **kwargs):
        self._generation += 1
        return super()._apply(fn, *args, **kwargs)

    def params_changed(self):
        """Tell the engine that parameter VALUES changed in place where torch cannot see it (manual .data edits
        of backbone weights, an optimizer stepping backbone parameters).  volume_net parameters, load_state_dict and
        .to() are tracked automatically."""
        self._generation += 1

    def lifter_params_changed(self):
        """Call after an in-place update of volume_net VALUES that torch cannot see (capf.optim.FusedAdamW writes
        them from its own kernel); torch optimizers bump tensor versions and are picked up automatically."""
        stream = torch.cuda.current_stream().cuda_stream
        for eng in self._engines.values():
            if eng._bound:
                eng.lifter_params_changed(stream)

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
            eng = Engine(_native.make_capf_config(self._config, H, W, context_blocks=self.context_blocks,
                                                  compute_dtype=self.compute_dtype, plan_flags=self.plan_flags), device=dev.index)
            eng._bound_generation = None
            eng._lifter_print = None
            self._engines[key] = eng
        lifter = list(self.volume_net.parameters())
        ptrs = tuple(p.data_ptr() for p in lifter)
        versions = tuple(p._version for p in lifter)
        moved = eng._lifter_print is None or eng._lifter_print[0] != ptrs
        if eng._bound_generation != self._generation or moved or not eng._bound:
            # storage may have moved (or frozen values changed): borrow every pointer again, fold + pack everything
            if self.backbone.training and any(p.requires_grad for p in self.backbone.parameters()):
                raise NotImplementedError("training-mode BatchNorm / backbone gradients are outside the hot "
                                          "path: freeze the backbone (fix_weights) and call backbone.eval()")
            state = self.state_dict(keep_vars=True)
            eng._bound = {}
            eng.bind_state({k: v.data for k, v in state.items()}, torch.cuda.current_stream(dev).cuda_stream)
            eng._bound_generation = self._generation
            eng._lifter_print = (ptrs, versions)
            eng.rebinds = getattr(eng, "rebinds", 0) + 1
        elif eng._lifter_print[1] != versions:
            eng.lifter_params_changed(torch.cuda.current_stream(dev).cuda_stream)      # only volume_net values moved (optimizer step)
            eng._lifter_print = (ptrs, versions)
        return eng

    # ---- conpose.py:30-42 --------------------------------------------------------------------
    def forward(self, images, keypoints_2d_cpn, keypoints_2d_cpn_crop):
        if images.dtype != torch.float32:
            raise TypeError("images must be float32")
        if images.device.type != "cuda":
            raise CapfError("CA_PF runs on an MI355X only: inputs are on {} (no CPU fallback)".format(images.device))
        images = images.contiguous()
        B = images.shape[0]
        for name, t in (("keypoints_2d_cpn", keypoints_2d_cpn), ("keypoints_2d_cpn_crop", keypoints_2d_cpn_crop)):
            if t.dtype != torch.float32:
                raise TypeError("{} must be float32, got {}".format(name, t.dtype))
            if t.device != images.device:
                raise ValueError("{} is on {} but images are on {}".format(name, t.device, images.device))
            if tuple(t.shape) != (B, self.num_joints, 2):
                raise ValueError("{} must have shape ({}, {}, 2), got {}".format(name, B, self.num_joints, tuple(t.shape)))
        with torch.cuda.device(images.device):
            eng = self._engine(images)
            k2d = keypoints_2d_cpn.contiguous()
            # the third argument is normalised IN PLACE (conpose.py:34-35); a non-contiguous view (which the reference
            # accepts) goes through a contiguous staging copy that is written back
            kcrop = keypoints_2d_cpn_crop if keypoints_2d_cpn_crop.is_contiguous() else keypoints_2d_cpn_crop.contiguous()
            stream = torch.cuda.current_stream(images.device).cuda_stream
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.volume_net.parameters()):
                named = [(n, p) for n, p in self.volume_net.named_parameters()]
                if not all(p.requires_grad for _, p in named):
                    raise NotImplementedError("partially frozen volume_net is not supported by the native backward")
                names = tuple("volume_net." + n for n, _ in named)
                out = _LifterStep.apply(self, eng, images, k2d, kcrop, self._drop_masks(B, images.device),
                                        names, *[p for _, p in named])
            else:
                out = torch.empty(B, 1, self.num_joints, 3, dtype=torch.float32, device=images.device)
                eng.forward(images, k2d, kcrop, out, stream)
            if kcrop is not keypoints_2d_cpn_crop:
                with torch.no_grad():
                    keypoints_2d_cpn_crop.copy_(kcrop)
        return out

    def _drop_masks(self, B, device):
        """DropPath multipliers for one training step (timm semantics: keep-mask / keep_prob per sample;
        a 'sample' is one batch element for context / joint blocks and one (frame, joint) row group for the
        level blocks, whose batch axis is (b p) — pose_dformer.py:231-234).  None when nothing is dropped."""
        levels = self._config.model.poseformer.levels
        if not self.training or self.drop_path_rate <= 0.0:
            return None
        rates = torch.linspace(0, self.drop_path_rate, levels).tolist()
        parts = []
        for per in (B, B * self.num_joints, B):
            for r in rates:
                for _ in range(2):
                    if r == 0.0:
                        parts.append(torch.ones(per, device=device))
                    else:
                        keep = 1.0 - r
                        parts.append(torch.empty(per, device=device).bernoulli_(keep).div_(keep))
        return torch.cat(parts).contiguous()

    def forward_flip_test(self, images2, keypoints_2d_cpn2, keypoints_2d_cpn_crop2):
        """Flip-test evaluation (train.py:170-181) as ONE forward: inputs are the [2,B,...] stacks that
        capf_preprocess(mode=2) emits (original, mirrored); the two predictions are un-mirrored and averaged
        by capf_fliptest_fuse.  Returns [B,1,17,3].  The crop keypoints are normalised in place."""
        from capf.lib import fliptest_fuse
        two, B = images2.shape[0], images2.shape[1]
        assert two == 2 and images2.is_contiguous() and keypoints_2d_cpn_crop2.is_contiguous()
        pred = self.forward(images2.view(2 * B, *images2.shape[2:]), keypoints_2d_cpn2.reshape(2 * B, 17, 2),
                            keypoints_2d_cpn_crop2.view(2 * B, 17, 2))
        return fliptest_fuse(pred.view(2, B, 1, 17, 3))

    def engine_for(self, images):
        """The native engine serving inputs of this shape/device (tests, bench)."""
        return self._engine(images.contiguous())
