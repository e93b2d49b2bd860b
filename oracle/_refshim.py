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
    """Install the timm / easydict shims.  Never writes into /root/reference."""
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
                    if DropPath.record is not None and self.training:
                        DropPath.record.append(torch.ones(x.shape[0]))
                    return x
                keep = 1.0 - self.drop_prob
                mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
                if keep > 0.0 and self.scale_by_keep:
                    mask.div_(keep)
                if DropPath.record is not None:          # golden generation: keep the multipliers that were drawn
                    DropPath.record.append(mask.detach().clone().reshape(-1))
                return x * mask

        DropPath.record = None

        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")
        layers.DropPath = DropPath
        # imported (not used on the path) by ContextPose_mpi/model/pose_dformer.py
        layers.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
        layers.trunc_normal_ = nn.init.trunc_normal_
        models.layers = layers
        timm.models = models
        sys.modules["timm"] = timm
        sys.modules["timm.models"] = models
        sys.modules["timm.models.layers"] = layers
        registry = types.ModuleType("timm.models.registry")          # imported, unused (ContextPose_mpi)
        registry.register_model = lambda fn: fn
        models.registry = registry
        models.__path__ = []
        sys.modules["timm.models.registry"] = registry

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



BUILD_PKG_MARK = "contextaware-poseformer_amd"


class reference_imports:
    """Context manager: inside it, `import mvn...` resolves to /root/reference/ContextPose/mvn and to nothing else.

    The reference's `mvn/` has no __init__.py (a namespace package) while the build's host mirror
    `contextaware-poseformer_amd/mvn/` is a regular package, and a regular package wins over a namespace package
    whatever the order of sys.path.  So for the duration of the block the build's package root is taken OFF
    sys.path and every `mvn*` module is purged from sys.modules; on exit the reference's `mvn*` modules are purged
    again and the previous ones restored (objects already built from the reference's classes keep working).
    `check()` asserts the origin of what was imported: a golden can never be produced from the mirror by accident."""

    def __init__(self, root=REFERENCE_ROOT, packages=("mvn",)):
        self.root, self.packages = root, tuple(packages)

    def _mine(self, k):
        return any(k == p or k.startswith(p + ".") for p in self.packages)

    def __enter__(self):
        install()
        self.saved_path = list(sys.path)
        self.saved_mods = {k: v for k, v in sys.modules.items() if self._mine(k)}
        for k in self.saved_mods:
            del sys.modules[k]
        sys.path[:] = [self.root] + [p for p in sys.path if BUILD_PKG_MARK not in p and p != self.root]
        return self

    def check(self, *module_names):
        for n in module_names:
            f = getattr(sys.modules[n], "__file__", None) or ""
            assert f.startswith(self.root + "/"), f"{n} was imported from {f!r}, not from the reference ({self.root})"

    def __exit__(self, *exc):
        for k in [k for k in sys.modules if self._mine(k)]:
            del sys.modules[k]
        sys.modules.update(self.saved_mods)
        sys.path[:] = self.saved_path
        return False


def _patch_config(c, backbone, embed_dim_ratio):
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


def reference_config(backbone="hrnet_32", embed_dim_ratio=128):
    """Fresh copy of the reference's default config patched exactly as train.py:266-277 does."""
    import copy
    with reference_imports() as ri:
        from mvn.utils import cfg as refcfg
        ri.check("mvn.utils.cfg")
        return _patch_config(copy.deepcopy(refcfg.config), backbone, embed_dim_ratio)


def build_reference(backbone="hrnet_32", embed_dim_ratio=128):
    """The REAL reference CA_PF (ContextPose/mvn/models/conpose.py:10-42) on CPU, eval mode."""
    import contextlib, copy, io
    with reference_imports() as ri:
        from mvn.utils import cfg as refcfg
        from mvn.models.conpose import CA_PF
        ri.check("mvn.utils.cfg", "mvn.models.conpose", "mvn.models.pose_dformer", "mvn.models.pose_hrnet")
        c = _patch_config(copy.deepcopy(refcfg.config), backbone, embed_dim_ratio)
        with contextlib.redirect_stdout(io.StringIO()):
            m = CA_PF(c, device="cpu")
    assert type(m).__module__ == "mvn.models.conpose" and type(m.volume_net).__name__ == "PoseTransformer"
    return m.eval(), c


def reference_losses():
    """The reference's mvn/models/loss.py module object (MPJPE, P_MPJPE, N_MPJPE, MPJVE, Keypoints*Loss)."""
    with reference_imports() as ri:
        import importlib
        mod = importlib.import_module("mvn.models.loss")
        ri.check("mvn.models.loss")
    return mod


MPI_ROOT = "/root/reference/ContextPose_mpi"


def build_reference_mpi(backbone="hrnet_32", depth=None):
    """The sibling app's model (ContextPose_mpi/model/conpose.py) with run_3dhp.py:219-235's config patch.
    Its packages are called `model` / `common` (the build's own mirror package is also called `model`)."""
    import contextlib, copy, importlib, io
    with reference_imports(MPI_ROOT, ("model", "common")) as ri:
        cfgmod = importlib.import_module("common.cfg")
        c = copy.deepcopy(cfgmod.config)
        if backbone == "hrnet_32":
            c.model.backbone.STAGE2.NUM_CHANNELS = [32, 64]
            c.model.backbone.STAGE3.NUM_CHANNELS = [32, 64, 128]
            c.model.backbone.STAGE4.NUM_CHANNELS = [32, 64, 128, 256]
            c.model.poseformer.base_dim = 32
            c.model.poseformer.embed_dim_ratio = 64
        if depth is not None:
            c.model.poseformer.depth = depth          # blocks per group (model/pose_dformer.py:199); common/cfg.py:83 ships 4
        net = importlib.import_module("model.conpose").VolumetricTriangulationNet
        ri.check("common.cfg", "model.conpose")
        with contextlib.redirect_stdout(io.StringIO()):
            m = net(c)
    return m.eval(), c


def run_reference_prefetcher(batch, backbone, is_train, flip_test, flip):
    """ONE batch through the reference's own `data_prefetcher.preload` (ContextPose/mvn/datasets/utils.py:15-89) on CPU.
    The class only runs on a GPU upstream (torch.cuda.Stream, .cuda()), and its package __init__ pulls the whole dataset
    stack (cv2, h5py): the FILE is executed as a module of its own (importlib, no package __init__) with `cv2` stubbed (it is
    imported, never called on this path) and the CUDA plumbing it touches replaced by CPU no-ops for the duration of the call:
    torch.cuda.Stream / torch.cuda.stream / Tensor.cuda / torch.cuda.current_stream.  `flip` replaces the reference's
    `random.random() <= 0.5` draw (:55).  Returns the four tensors `.next()` hands to one_epoch_full."""
    import contextlib
    import importlib.util
    import random
    import torch

    class _Stream:
        def wait_stream(self, other):
            pass

    class _Cv2Stub(types.ModuleType):          # OpenCV is absent here; the file only names cv2 constants as default arguments
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return 0

    stubbed = "cv2" not in sys.modules
    if stubbed:
        sys.modules["cv2"] = _Cv2Stub("cv2")
    saved = (torch.cuda.Stream, torch.cuda.stream, torch.Tensor.cuda, torch.cuda.current_stream, random.random)
    try:
        with reference_imports() as ri:
            torch.cuda.Stream = lambda *a, **k: _Stream()
            torch.cuda.stream = lambda s: contextlib.nullcontext()
            torch.Tensor.cuda = lambda self, *a, **k: self
            torch.cuda.current_stream = lambda *a, **k: _Stream()
            random.random = lambda: 0.0 if flip else 1.0
            spec = importlib.util.spec_from_file_location("reference_datasets_utils", REFERENCE_ROOT + "/mvn/datasets/utils.py")
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            ri.check("mvn.utils.img")
            assert mod.__file__.startswith(REFERENCE_ROOT + "/")
            pf = mod.data_prefetcher([[t.clone() for t in batch]], torch.device("cpu"), is_train, flip_test, backbone)
            out = pf.next()
    finally:
        torch.cuda.Stream, torch.cuda.stream, torch.Tensor.cuda, torch.cuda.current_stream, random.random = saved
        if stubbed:
            del sys.modules["cv2"]
    return out
