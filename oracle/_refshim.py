"""Import shims for running the *reference* (read-only, /root/reference) in the build container.

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_goldens.py (and nothing else) to import
`/root/reference/ContextPose/mvn` on CPU so that golden vectors can be generated from the real
reference.  The reference never travels to the GPU box; only the vectors do (tests/golden/).

The reference needs two packages that are not installed in this image:
  * timm.models.layers.DropPath   (ContextPose/mvn/models/pose_dformer.py:12)
  * easydict.EasyDict             (ContextPose/mvn/utils/cfg.py:2)
Both are restated here from their documented behaviour (they are third-party, not reference code).
"""
import sys
import types

REFERENCE_ROOT = "/root/reference/ContextPose"


def install():
    """Install the shims and put the reference on sys.path.  Never writes into /root/reference."""
    sys.dont_write_bytecode = True
    import torch
    from torch import nn

    if "timm" not in sys.modules:
        class DropPath(nn.Module):
            """Stochastic depth per sample (timm 0.6.7 semantics): identity in eval or p == 0."""

            def __init__(self, drop_prob=0.0, scale_by_keep=True):
                super().__init__()
                self.drop_prob = float(drop_prob)
                self.scale_by_keep = scale_by_keep

            def forward(self, x):
                if self.drop_prob == 0.0 or not self.training:
                    return x
                keep = 1.0 - self.drop_prob
                mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
                if keep > 0.0 and self.scale_by_keep:
                    mask.div_(keep)
                return x * mask

        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")
        layers.DropPath = DropPath
        models.layers = layers
        timm.models = models
        sys.modules["timm"] = timm
        sys.modules["timm.models"] = models
        sys.modules["timm.models.layers"] = layers

    if "easydict" not in sys.modules:
        class EasyDict(dict):
            """Attribute-access dict, nested dicts converted recursively."""

            def __init__(self, d=None, **kw):
                super().__init__()
                d = dict(d or {})
                d.update(kw)
                for k, v in d.items():
                    self[k] = v

            def __setitem__(self, k, v):
                if isinstance(v, dict) and not isinstance(v, EasyDict):
                    v = EasyDict(v)
                elif isinstance(v, (list, tuple)):
                    v = type(v)(EasyDict(x) if isinstance(x, dict) else x for x in v)
                super().__setitem__(k, v)

            __setattr__ = __setitem__

            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError as e:
                    raise AttributeError(k) from e

        mod = types.ModuleType("easydict")
        mod.EasyDict = EasyDict
        sys.modules["easydict"] = mod

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_config(backbone="hrnet_32", embed_dim_ratio=128):
    """Fresh copy of the reference's default config patched exactly as train.py:266-277 does."""
    install()
    import copy
    from mvn.utils import cfg as refcfg
    c = copy.deepcopy(refcfg.config)
    c.model.backbone.type = backbone
    c.model.backbone.fix_weights = True           # human36m.yaml:21
    c.model.poseformer.embed_dim_ratio = embed_dim_ratio
    if backbone == "hrnet_48":
        c.model.backbone.STAGE2.NUM_CHANNELS = [48, 96]
        c.model.backbone.STAGE3.NUM_CHANNELS = [48, 96, 192]
        c.model.backbone.STAGE4.NUM_CHANNELS = [48, 96, 192, 384]
        c.model.poseformer.base_dim = 48
    elif backbone == "cpn":
        c.model.poseformer.base_dim = 256
    return c


def build_reference(backbone="hrnet_32", embed_dim_ratio=128):
    install()
    import contextlib, io
    from mvn.models.conpose import CA_PF
    c = reference_config(backbone, embed_dim_ratio)
    with contextlib.redirect_stdout(io.StringIO()):
        m = CA_PF(c, device="cpu")
    return m.eval(), c
