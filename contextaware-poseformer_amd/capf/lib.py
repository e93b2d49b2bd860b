"""ctypes binding of the C ABI in include/capf.h.  PyTorch is used for device memory and streams
only (tensor.data_ptr(), torch.cuda.current_stream()); no torch type crosses the ABI."""
import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

# CAPF_LIB: an alternative build of the same ABI (A/B timing of kernel variants on one GPU box; tools only)
LIB_PATH = os.environ.get("CAPF_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcapf.so")
HRNET, CPN50 = 0, 1
F32, BF16 = 0, 1
PLAN_NO_FUSED_LIFTER, PLAN_NO_WINOGRAD, PLAN_NO_ROW_HALO, PLAN_WINOGRAD_F23_ONLY, PLAN_NO_PWCHAIN, PLAN_NO_WS, PLAN_LIFTER_FP32, PLAN_NO_F32X3, PLAN_F32X3_EXACT, PLAN_NO_F32H2_GEMM, PLAN_NO_UPADD, PLAN_H2_PLANES, PLAN_NO_BNECK, PLAN_NO_BATCHED_REDUCE = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192     # capf_plan_flag
ABI_VERSION = 6        # include/capf.h :: CAPF_ABI_VERSION (checked against capf_abi_version() at load)

EXPORTS = [  # every symbol include/capf.h declares (checked by tests/test_abi.py)
    "capf_create", "capf_destroy", "capf_last_error", "capf_version", "capf_num_params", "capf_param_info",
    "capf_set_param", "capf_params_changed", "capf_lifter_params_changed", "capf_workspace_bytes", "capf_set_workspace", "capf_forward",
    "capf_backbone_forward", "capf_lifter_forward", "capf_set_debug", "capf_set_lanes", "capf_op_conv_group", "capf_affine_from_center_scale", "capf_warp_affine", "capf_op_schedule", "capf_forward_profile_launches", "capf_forward_profile_variants", "capf_tensor", "capf_forward_stats",
    "capf_num_ops", "capf_op_info", "capf_forward_profile", "capf_op_pack_conv", "capf_op_conv", "capf_op_linear",
    "capf_preprocess", "capf_fliptest_fuse", "capf_op_pack_conv_bf16", "capf_op_conv_bf16", "capf_op_conv_bf16_rh_width", "capf_op_pack_conv_bf16_rh", "capf_op_conv_bf16_rh", "capf_op_conv_bf16_group", "capf_forward_train", "capf_backward", "capf_grad_elems", "capf_grad_info", "capf_train_h2_matrices", "capf_mpjpe", "capf_adamw_step",
    "capf_pose_errors", "capf_segment_sums", "capf_keypoints_loss", "capf_train_generation", "capf_max_batch", "capf_op_bytes", "capf_op_linear_bf16", "capf_op_pack_conv_wino", "capf_op_conv_wino", "capf_op_conv_wino_group",
    "capf_op_bilinear_corners", "capf_mpjpe_nd", "capf_op_executed_flops",
    "capf_forward_prefix", "capf_op_describe", "capf_op_tensor",
    "capf_op_conv_bf16_ws_pack_elems", "capf_op_pack_conv_bf16_ws", "capf_op_conv_bf16_ws_group",
    "capf_op_conv_f32x3_pack_elems", "capf_op_pack_conv_f32x3", "capf_op_conv_f32x3_group",
    "capf_op_conv_f32h2_pack_elems", "capf_op_pack_conv_f32h2", "capf_op_conv_f32h2_group",
    "capf_abi_version", "capf_op_describe_sized",
    "capf_jpeg_info", "capf_jpeg_coefficients", "capf_jpeg_decode",
    "capf_op_f32h2_gemm_pack_elems", "capf_op_pack_f32h2_gemm", "capf_op_conv_f32h2g", "capf_op_conv_f32h2g_group", "capf_op_linear_f32h2g", "capf_op_linear_ln_f32h2g", "capf_op_wgrad", "capf_op_conv_f32h2_tiles", "capf_op_conv_f32h2_planes", "capf_op_h2_planes",
]


class CapfError(RuntimeError):
    pass


class CapfConfig(ctypes.Structure):
    _fields_ = [
        ("backbone", c_int32), ("hr_channels", c_int32 * 4), ("hr_modules", c_int32 * 3), ("hr_blocks", c_int32),
        ("base_dim", c_int32), ("embed_dim_ratio", c_int32), ("levels", c_int32), ("num_joints", c_int32),
        ("num_heads", c_int32), ("deform_heads", c_int32), ("deform_samples", c_int32), ("context_blocks", c_int32),
        ("compute_dtype", c_int32), ("max_batch", c_int32), ("height", c_int32), ("width", c_int32),
        ("training", c_int32), ("plan_flags", c_int32), ("depth", c_int32),
    ]


class OpDesc(ctypes.Structure):
    """mirrors include/capf.h :: capf_op_desc"""
    _fields_ = [(k, c_int32) for k in ("kind", "backbone", "conv", "Cin", "H", "W", "Cout", "Ho", "Wo", "ks", "stride", "pad", "act",
                                       "in_dtype", "out_dtype", "mfma_bf16", "n_in")] + [("shift", c_int32 * 4)] + \
               [(k, c_int32) for k in ("relu", "p_weight", "p_bn_weight", "has_residual", "checkpoint", "rows_per_frame", "p_bias",
                                       "p_ln_weight", "p_ln_bias")] + [("eps", c_float), ("attn", c_int32 * 4), ("maps", (c_int64 * 4) * 3),
                                                                         ("up_H", c_int32), ("up_W", c_int32)]


_lib = None


def load_library():
    """Load libcapf.so; fail loudly (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64/libhsa-runtime64: it must be loaded FIRST so that libcapf.so
    # binds to the same HIP runtime (two runtimes in one process do not share devices or streams).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise CapfError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                        "(or make -C contextaware-poseformer_amd/csrc); there is no fallback path")
    lib = ctypes.CDLL(LIB_PATH)
    # structs cross this boundary (CapfConfig, ConvDesc, OpDesc): a library built from another revision of capf.h must not be driven
    # with this file's layouts
    abi = lib.capf_abi_version() if hasattr(lib, "capf_abi_version") else None
    if abi != ABI_VERSION:
        raise CapfError(f"{LIB_PATH} implements ABI revision {abi}, this binding expects {ABI_VERSION}: rebuild the library")
    H = c_void_p
    lib.capf_create.argtypes = [POINTER(CapfConfig), c_int, POINTER(H)]
    lib.capf_create.restype = c_int
    lib.capf_destroy.argtypes = [H]
    lib.capf_destroy.restype = None
    lib.capf_last_error.argtypes = [H]
    lib.capf_last_error.restype = c_char_p
    lib.capf_version.restype = c_char_p
    lib.capf_num_params.argtypes = [H]
    lib.capf_param_info.argtypes = [H, c_int, POINTER(c_char_p), POINTER(c_int64), POINTER(c_int), POINTER(c_int)]
    lib.capf_set_param.argtypes = [H, c_char_p, c_void_p, POINTER(c_int64), c_int]
    lib.capf_params_changed.argtypes = [H, c_void_p]
    lib.capf_lifter_params_changed.argtypes = [H, c_void_p]
    lib.capf_workspace_bytes.argtypes = [H, c_int]
    lib.capf_workspace_bytes.restype = c_size_t
    lib.capf_set_workspace.argtypes = [H, c_void_p, c_size_t]
    lib.capf_forward.argtypes = [H, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
    lib.capf_backbone_forward.argtypes = [H, c_void_p, c_void_p, c_int]
    lib.capf_lifter_forward.argtypes = [H, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
    lib.capf_set_debug.argtypes = [H, c_int]
    lib.capf_set_lanes.argtypes = [H, c_int]
    lib.capf_tensor.argtypes = [H, c_char_p, POINTER(c_void_p), POINTER(c_int64), POINTER(c_int)]
    lib.capf_forward_stats.argtypes = [H, c_int, POINTER(c_int64), POINTER(c_double)]
    lib.capf_forward_profile_variants.argtypes = [H, POINTER(c_int32), c_int]
    lib.capf_num_ops.argtypes = [H]
    lib.capf_op_info.argtypes = [H, c_int, c_int, POINTER(c_char_p), POINTER(c_char_p), POINTER(c_double)]
    lib.capf_op_bytes.argtypes = [H, c_int, c_int, POINTER(c_double)]
    lib.capf_op_executed_flops.argtypes = [H, c_int, c_int, POINTER(c_double)]
    lib.capf_forward_prefix.argtypes = [H, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int]
    lib.capf_op_describe.argtypes = [H, c_int, POINTER(OpDesc)]
    lib.capf_op_describe_sized.argtypes = [H, c_int, c_void_p, c_size_t]
    lib.capf_op_tensor.argtypes = [H, c_int, c_int, POINTER(c_void_p)]
    lib.capf_forward_profile.argtypes = [H, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                         POINTER(c_float), c_int]
    lib.capf_forward_profile_launches.argtypes = [H, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                                  POINTER(c_float), POINTER(c_int32), c_int]
    P = c_void_p
    lib.capf_forward_train.argtypes = [H, P, P, P, P, c_int, P, P]
    lib.capf_backward.argtypes = [H, P, P, c_int, P, P]
    lib.capf_train_generation.argtypes = [H]
    lib.capf_train_generation.restype = c_int64
    lib.capf_max_batch.argtypes = [H]
    lib.capf_grad_elems.argtypes = [H]
    lib.capf_grad_elems.restype = c_int64
    lib.capf_grad_info.argtypes = [H, c_int, POINTER(c_int64)]
    lib.capf_train_h2_matrices.argtypes = [H]
    lib.capf_train_h2_matrices.restype = c_int
    lib.capf_mpjpe.argtypes = [P, P, P, c_int, P, P, c_float]
    lib.capf_mpjpe_nd.argtypes = [P, P, P, c_int, c_int, P, P, c_float]
    lib.capf_adamw_step.argtypes = [P, P, P, P, P, c_int64] + [c_float] * 5 + [c_int, c_float]
    lib.capf_op_bilinear_corners.argtypes = [P, P, c_int, c_int, c_int, c_int, P, P]
    lib.capf_op_pack_conv.argtypes = [P, P, P, P, P, P, c_float, P, P, c_int, c_int, c_int]
    lib.capf_op_conv.argtypes = [P, P, P, P, P, P] + [c_int] * 8
    lib.capf_op_linear.argtypes = [P, P, P, P, P, P] + [c_int] * 4
    lib.capf_preprocess.argtypes = [P, P, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_int, P, P, P, P, P, P, P]
    lib.capf_fliptest_fuse.argtypes = [P, P, c_int, P]
    lib.capf_op_pack_conv_bf16.argtypes = [P, P, P, P, P, P, c_float, P, P, c_int, c_int, c_int]
    lib.capf_op_conv_bf16.argtypes = [P, P, P, P, P, P] + [c_int] * 8
    lib.capf_op_conv_bf16_rh_width.argtypes = [c_int]
    lib.capf_op_pack_conv_bf16_rh.argtypes = [P, P, P, P, P, P, c_float, P, P, c_int, c_int]
    lib.capf_op_conv_bf16_rh.argtypes = [P, P, P, P, P, P] + [c_int] * 6
    lib.capf_op_conv_bf16_ws_pack_elems.argtypes = [c_int, c_int]
    lib.capf_op_conv_bf16_ws_pack_elems.restype = c_int64
    lib.capf_op_pack_conv_bf16_ws.argtypes = [P, P, P, P, P, P, c_float, P, P, c_int, c_int]
    lib.capf_op_conv_f32x3_pack_elems.argtypes = [c_int, c_int]
    lib.capf_op_conv_f32x3_pack_elems.restype = c_int64
    lib.capf_op_pack_conv_f32x3.argtypes = [P, P, P, P, P, P, c_float, P, P, c_int, c_int]
    lib.capf_op_conv_f32h2_pack_elems.argtypes = [c_int, c_int]
    lib.capf_op_conv_f32h2_pack_elems.restype = c_int64
    lib.capf_op_pack_conv_f32h2.argtypes = [P, P, P, P, P, P, c_float, P, P, c_int, c_int]
    lib.capf_op_f32h2_gemm_pack_elems.argtypes = [c_int, c_int]
    lib.capf_op_f32h2_gemm_pack_elems.restype = c_int64
    lib.capf_op_pack_f32h2_gemm.argtypes = [P, P, P, P, P, P, c_float, P, P, c_int, c_int, c_int, c_int]
    lib.capf_op_conv_f32h2g.argtypes = [P, P, P, P, P, P] + [c_int] * 8
    lib.capf_op_linear_f32h2g.argtypes = [P, P, P, P, P, P] + [c_int] * 4
    lib.capf_op_linear_ln_f32h2g.argtypes = [P, P, P, P, c_float, P, P, P, P] + [c_int] * 4
    lib.capf_op_wgrad.argtypes = [P, P, P, c_int, c_int, c_int, P, c_int]
    lib.capf_op_pack_conv_wino.argtypes = [P, P, P, P, P, P, c_float, P, P, c_int, c_int, c_int]
    lib.capf_op_conv_wino.argtypes = [P, P, P, P, P, P] + [c_int] * 7
    lib.capf_op_linear_bf16.argtypes = [P, P, P, P, P, P] + [c_int] * 4
    lib.capf_pose_errors.argtypes = [P, P, P, c_int, c_int, P, P]
    lib.capf_segment_sums.argtypes = [P, P, P, P, c_int, c_int, P, P]
    lib.capf_keypoints_loss.argtypes = [P, c_int, P, P, P, c_int, c_int, c_float, P, P]
    _lib = lib
    return lib


PARAM_KINDS = {0: "conv_w", 1: "bn_w", 2: "bn_b", 3: "bn_mean", 4: "bn_var", 5: "bn_nbt", 6: "lin_w", 7: "lin_b",
               8: "ln_w", 9: "ln_b", 10: "raw"}


class Engine:
    """One native handle.  device=None -> plan-only (schema / workspace queries, no GPU)."""

    def __init__(self, cfg: CapfConfig, device=None):
        self.lib = load_library()
        self.cfg = cfg
        self.h = c_void_p()
        rc = self.lib.capf_create(byref(cfg), -1 if device is None else int(device), byref(self.h))
        if rc != 0:
            raise CapfError(f"capf_create failed ({rc}): {self.lib.capf_last_error(None).decode()}")
        self.device = device
        self._ws = None
        self._ws_batch = 0
        self._bound = {}

    def close(self):
        if getattr(self, "h", None):
            self.lib.capf_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc, what):
        if rc < 0:
            raise CapfError(f"{what} failed ({rc}): {self.lib.capf_last_error(self.h).decode()}")
        return rc

    # ---- schema
    def schema(self):
        """[(name, shape tuple, kind str)] == the reference's state_dict (incl. BN buffers)."""
        out = []
        name, shape, nd, kind = c_char_p(), (c_int64 * 4)(), c_int(), c_int()
        for i in range(self.lib.capf_num_params(self.h)):
            self._check(self.lib.capf_param_info(self.h, i, byref(name), shape, byref(nd), byref(kind)), "param_info")
            out.append((name.value.decode(), tuple(shape[j] for j in range(nd.value)), PARAM_KINDS[kind.value]))
        return out

    # ---- parameters
    def bind_state(self, named_tensors, stream=0):
        """Borrow device pointers for every schema entry from {name: cuda fp32 tensor}; fold/pack."""
        import torch
        for name, shape, kind in self.schema():
            if kind == "bn_nbt":
                continue
            t = named_tensors[name]
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise CapfError(f"{name}: need a contiguous fp32 CUDA tensor, got {t.dtype} {t.device}")
            if self._bound.get(name) == t.data_ptr():
                continue
            shp = (c_int64 * 4)(*(list(t.shape) + [0] * (4 - t.dim())))
            self._check(self.lib.capf_set_param(self.h, name.encode(), c_void_p(t.data_ptr()), shp, t.dim()),
                        f"set_param({name})")
            self._bound[name] = t.data_ptr()
        self.params_changed(stream)

    def params_changed(self, stream=0):
        self._check(self.lib.capf_params_changed(self.h, c_void_p(stream)), "params_changed")

    def lifter_params_changed(self, stream=0):
        self._check(self.lib.capf_lifter_params_changed(self.h, c_void_p(stream)), "lifter_params_changed")

    # ---- workspace
    def workspace_bytes(self, batch):
        return self.lib.capf_workspace_bytes(self.h, batch)

    def ensure_workspace(self, batch):
        import torch
        if self._ws is None or batch > self._ws_batch:
            nbytes = self.workspace_bytes(batch)
            self._ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=f"cuda:{self.device}")
            self._ws_batch = batch
            self._check(self.lib.capf_set_workspace(self.h, c_void_p(self._ws.data_ptr()), nbytes), "set_workspace")

    # ---- hot path
    def forward(self, images, k2d, kcrop, out, stream):
        B = images.shape[0]
        self.ensure_workspace(B)
        self._check(self.lib.capf_forward(self.h, c_void_p(stream), c_void_p(images.data_ptr()),
                                          c_void_p(k2d.data_ptr()), c_void_p(kcrop.data_ptr()), B,
                                          c_void_p(out.data_ptr())), "forward")

    # ---- training step
    def grad_layout(self):
        """{parameter name: (offset, numel)} inside the flat lifter gradient; total element count."""
        out, off = {}, c_int64()
        for i, (name, shape, kind) in enumerate(self.schema()):
            self._check(self.lib.capf_grad_info(self.h, i, byref(off)), "grad_info")
            if off.value >= 0:
                n = 1
                for d in shape:
                    n *= d
                out[name] = (off.value, n)
        return out, self.lib.capf_grad_elems(self.h)

    def grad_layout_cached(self):
        if not hasattr(self, "_grad_layout"):
            self._grad_layout = self.grad_layout()
        return self._grad_layout

    def forward_train(self, images, k2d, kcrop, out, stream, masks=None):
        B = images.shape[0]
        self.ensure_workspace(B)
        self._check(self.lib.capf_forward_train(self.h, c_void_p(stream), c_void_p(images.data_ptr()),
                                                c_void_p(k2d.data_ptr()), c_void_p(kcrop.data_ptr()), B,
                                                c_void_p(out.data_ptr()),
                                                c_void_p(masks.data_ptr()) if masks is not None else c_void_p(0)),
                    "forward_train")

    def train_generation(self):
        return self.lib.capf_train_generation(self.h)

    def max_batch(self):
        return self.lib.capf_max_batch(self.h)

    def backward(self, grad_out, flat_grad, stream, masks=None):
        B = grad_out.shape[0]
        self._check(self.lib.capf_backward(self.h, c_void_p(stream), c_void_p(grad_out.data_ptr()), B,
                                           c_void_p(flat_grad.data_ptr()),
                                           c_void_p(masks.data_ptr()) if masks is not None else c_void_p(0)), "backward")

    def backbone_forward(self, images, stream):
        B = images.shape[0]
        self.ensure_workspace(B)
        self._check(self.lib.capf_backbone_forward(self.h, c_void_p(stream), c_void_p(images.data_ptr()), B),
                    "backbone_forward")

    def lifter_forward(self, k2d, kcrop, out, stream):
        B = k2d.shape[0]
        self.ensure_workspace(B)
        self._check(self.lib.capf_lifter_forward(self.h, c_void_p(stream), c_void_p(k2d.data_ptr()),
                                                 c_void_p(kcrop.data_ptr()), B, c_void_p(out.data_ptr())),
                    "lifter_forward")

    def set_lanes(self, on):
        self._check(self.lib.capf_set_lanes(self.h, int(on)), "set_lanes")

    def set_debug(self, on):
        self._check(self.lib.capf_set_debug(self.h, int(on)), "set_debug")

    def tensor(self, name):
        """Copy of a named intermediate of the last forward (torch tensor on the device)."""
        import torch
        ptr, shape, nd = c_void_p(), (c_int64 * 4)(), c_int()
        rc = self._check(self.lib.capf_tensor(self.h, name.encode(), byref(ptr), shape, byref(nd)), f"tensor({name})")
        shp = [shape[i] for i in range(nd.value)]
        n = 1
        for s in shp:
            n *= s
        off = (ptr.value - self._ws.data_ptr()) // 4
        if rc == 2:       # bf16 tensor: two elements per float slot
            return self._ws[off:off + (n + 1) // 2].view(torch.bfloat16)[:n].view(*shp).clone()
        flat = self._ws[off:off + n]
        if rc == 1:
            flat = flat.view(torch.int32)
        return flat.view(*shp).clone()

    def op_table(self, batch):
        """[(op name, kernel name, algorithmic flops at `batch`)] in launch order."""
        name, kern, fl = c_char_p(), c_char_p(), c_double()
        out = []
        for i in range(self.lib.capf_num_ops(self.h)):
            self._check(self.lib.capf_op_info(self.h, i, batch, byref(name), byref(kern), byref(fl)), "op_info")
            out.append((name.value.decode(), kern.value.decode(), fl.value))
        return out

    def op_bytes(self, batch):
        """algorithmic HBM bytes per op at `batch` (capf_op_bytes), in launch order"""
        b, out = c_double(), []
        for i in range(self.lib.capf_num_ops(self.h)):
            self._check(self.lib.capf_op_bytes(self.h, i, batch, byref(b)), "op_bytes")
            out.append(b.value)
        return out

    def op_executed_flops(self, batch):
        """FLOPs the matrix pipe executes per op at `batch` (capf_op_executed_flops: Winograd ops count 1/2 or 2/3), launch order"""
        f, out = c_double(), []
        for i in range(self.lib.capf_num_ops(self.h)):
            self._check(self.lib.capf_op_executed_flops(self.h, i, batch, byref(f)), "op_executed_flops")
            out.append(f.value)
        return out

    # ---- layer-wise parity aids (capf_forward_prefix / capf_op_describe / capf_op_tensor)
    def forward_prefix(self, images, n_ops, stream, k2d=None, kcrop=None, out=None):
        B = images.shape[0]
        self.ensure_workspace(B)
        self._prefix_images = images
        P = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
        self._check(self.lib.capf_forward_prefix(self.h, c_void_p(stream), P(images), P(k2d), P(kcrop), B, P(out), int(n_ops)),
                    "forward_prefix")

    def op_describe(self, index):
        d = OpDesc()
        self._check(self.lib.capf_op_describe_sized(self.h, index, byref(d), ctypes.sizeof(d)), "op_describe")
        return d

    def op_tensor(self, index, slot, shape, dtype_code):
        """View (no copy) of one operand of op `index` after a forward_prefix: slot 0..3 inputs, 4 residual, 5 output;
        shape = full [B, ...] shape, dtype_code 0 fp32 / 2 bf16."""
        import torch
        ptr = c_void_p()
        self._check(self.lib.capf_op_tensor(self.h, index, slot, byref(ptr)), f"op_tensor({index}, {slot})")
        n = 1
        for v in shape:
            n *= v
        img = getattr(self, "_prefix_images", None)
        if img is not None and ptr.value == img.data_ptr():
            return img.view(*shape)
        off = (ptr.value - self._ws.data_ptr()) // 4
        if dtype_code == 2:
            return self._ws[off:off + (n + 1) // 2].view(torch.bfloat16)[:n].view(*shape)
        return self._ws[off:off + n].view(*shape)

    def op_h2_planes(self, index, batch):
        """(role, exps [tiles, C / 16] int32 view, tile_pixels) of op `index` at this batch: role 0 = plain fp32 tensors, 1 = its output holds split
        fp16 planes, 2 = its input does (capf_op_h2_planes); C = the planes tensor's channels."""
        import torch
        ptr, tp = c_void_p(), c_int(0)
        self.lib.capf_op_h2_planes.argtypes = [c_void_p, c_int, c_int, POINTER(c_void_p), POINTER(c_int)]
        role = self.lib.capf_op_h2_planes(self.h, index, batch, byref(ptr), byref(tp))
        if role <= 0:
            return 0, None, 0
        d = self.op_describe(index)
        C = d.Cout if role == 1 else d.Cin
        tiles = self.lib.capf_op_conv_f32h2_tiles(batch, d.H, d.W, None)
        off = (ptr.value - self._ws.data_ptr()) // 4
        return role, self._ws[off:off + tiles * (C // 16)].view(torch.int32).view(tiles, C // 16), tp.value

    def op_tensor_fp32(self, index, slot, shape, dtype_code):
        """op_tensor, with a planes tensor (a BasicBlock's conv1 -> conv2 operand where both run the two-fp16-piece tile) decoded to the fp32 values
        it stands for."""
        t = self.op_tensor(index, slot, shape, dtype_code)
        role, exps, tp = self.op_h2_planes(index, shape[0])
        if (role == 1 and slot == 5) or (role == 2 and slot == 0):
            return planes_to_fp32(t, exps, tp)
        return t

    def forward_profile(self, images, k2d, kcrop, out, stream):
        """One forward with a HIP event pair around every launch (on `stream`); returns ms per op.
        Synchronises the stream — measurement aid for bench.py, not the product path."""
        B = images.shape[0]
        self.ensure_workspace(B)
        n = self.lib.capf_num_ops(self.h)
        ms = (c_float * n)()
        self._check(self.lib.capf_forward_profile(self.h, c_void_p(stream), c_void_p(images.data_ptr()),
                                                  c_void_p(k2d.data_ptr()), c_void_p(kcrop.data_ptr()), B,
                                                  c_void_p(out.data_ptr()), ms, n), "forward_profile")
        return list(ms)

    def op_schedule(self):
        """[(region, level, lane, reads, writes)] per op: see capf_op_schedule."""
        self.lib.capf_op_schedule.argtypes = [c_void_p, c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                              POINTER(c_int32), POINTER(c_int32)]
        out = []
        for i in range(self.lib.capf_num_ops(self.h)):
            rg, lv, ln = c_int32(), c_int32(), c_int32()
            rd, wr = (c_int32 * 5)(), (c_int32 * 6)()
            self._check(self.lib.capf_op_schedule(self.h, i, byref(rg), byref(lv), byref(ln), rd, wr), "op_schedule")
            out.append((rg.value, lv.value, ln.value, [v for v in rd if v != -1], [v for v in wr if v != -1]))
        return out

    def forward_profile_launches(self, images, k2d, kcrop, out, stream):
        """One forward of the product schedule with an event pair around every LAUNCH (grouped launches
        included); returns (ms per op, leader per op): see capf_forward_profile_launches."""
        B = images.shape[0]
        self.ensure_workspace(B)
        n = self.lib.capf_num_ops(self.h)
        ms = (c_float * n)()
        leader = (c_int32 * n)()
        self._check(self.lib.capf_forward_profile_launches(self.h, c_void_p(stream), c_void_p(images.data_ptr()),
                                                           c_void_p(k2d.data_ptr()), c_void_p(kcrop.data_ptr()), B,
                                                           c_void_p(out.data_ptr()), ms, leader, n), "forward_profile_launches")
        return list(ms), list(leader)

    def profile_variants(self):
        """After forward_profile_launches: the device kernel of every grouped bf16 launch, by leader op (-1 elsewhere)."""
        n = self.lib.capf_num_ops(self.h)
        v = (c_int32 * n)()
        self._check(self.lib.capf_forward_profile_variants(self.h, v, n), "forward_profile_variants")
        return list(v)

    def stats(self, batch):
        n, f = c_int64(), c_double()
        self._check(self.lib.capf_forward_stats(self.h, batch, byref(n), byref(f)), "forward_stats")
        return n.value, f.value


# ---- stateless operators (op-level tests / micro-benchmarks) ---------------------------------------
def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _stream(t):
    import torch
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def pack_conv(w, bn=None, eps=1e-5):
    """w [Cout,Cin,k,k] cuda fp32; bn = (gamma, beta, mean, var) or None -> (w_packed [Cout,Kpad], bias [Cout])."""
    import torch
    lib = load_library()
    co, ci, ks, _ = w.shape
    kpad = (ks * ks * ci + 31) // 32 * 32
    wp = torch.empty(co, kpad, device=w.device)
    bias = torch.empty(co, device=w.device)
    g, b, m, v = bn if bn is not None else (None, None, None, None)
    rc = lib.capf_op_pack_conv(_stream(w), _p(w.contiguous()), _p(g), _p(b), _p(m), _p(v), eps, _p(wp), _p(bias), co, ci, ks)
    if rc:
        raise CapfError(f"capf_op_pack_conv failed ({rc})")
    return wp, bias


def conv_nhwc(x, wp, bias, ks, stride=1, act=0, residual=None):
    """x [B,H,W,Cin] cuda fp32 NHWC -> [B,Ho,Wo,Cout]."""
    import torch
    lib = load_library()
    B, H, W, ci = x.shape
    co = wp.shape[0]
    pad = ks // 2
    ho, wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    y = torch.empty(B, ho, wo, co, device=x.device)
    rc = lib.capf_op_conv(_stream(x), _p(x), _p(wp), _p(bias), _p(residual), _p(y), B, H, W, ci, co, ks, stride, act)
    if rc:
        raise CapfError(f"capf_op_conv failed ({rc})")
    return y


def pack_conv_wino(w, bn=None, eps=1e-5, variant=23):
    """w [Cout,Cin,3,3] cuda fp32 -> (Winograd weights [Cout, 12*Cin] for F(2,3) / [Cout, 18*Cin] for F(4,3), bias [Cout])."""
    import torch
    lib = load_library()
    co, ci, ks, _ = w.shape
    assert ks == 3
    wp = torch.empty(co, (18 if variant == 43 else 12) * ci, device=w.device)
    bias = torch.empty(co, device=w.device)
    g, b, m, v = bn if bn is not None else (None, None, None, None)
    rc = lib.capf_op_pack_conv_wino(_stream(w), _p(w.contiguous()), _p(g), _p(b), _p(m), _p(v), eps, _p(wp), _p(bias), co, ci, variant)
    if rc:
        raise CapfError(f"capf_op_pack_conv_wino failed ({rc})")
    return wp, bias


def _wino_variant(wp, ci):
    return 43 if wp.shape[1] == 18 * ci else 23


def conv_nhwc_wino(x, wp, bias, act=0, residual=None):
    """3x3 stride-1 conv through the Winograd kernel (variant from the packed pitch): x [B,H,W,Cin] cuda fp32 NHWC -> [B,H,W,Cout]."""
    import torch
    lib = load_library()
    B, H, W, ci = x.shape
    co = wp.shape[0]
    y = torch.empty(B, H, W, co, device=x.device)
    rc = lib.capf_op_conv_wino(_stream(x), _p(x), _p(wp), _p(bias), _p(residual), _p(y), B, H, W, ci, co, act, _wino_variant(wp, ci))
    if rc:
        raise CapfError(f"capf_op_conv_wino failed ({rc})")
    return y


def conv_nhwc_wino_group(problems):
    """problems: list of (x, wp_wino, bias, act, residual) -> outputs; ONE grouped Winograd launch."""
    import torch
    lib = load_library()
    lib.capf_op_conv_wino_group.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ConvDesc), ctypes.c_int]
    descs = (ConvDesc * len(problems))()
    outs = []
    for d, (x, wp, bias, act, residual) in zip(descs, problems):
        B, H, W, ci = x.shape
        co = wp.shape[0]
        y = torch.empty(B, H, W, co, device=x.device)
        outs.append(y)
        d.x, d.w_packed, d.bias, d.y = x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr()
        d.residual = residual.data_ptr() if residual is not None else None
        d.B, d.H, d.W, d.Cin, d.Cout, d.ks, d.stride, d.act = B, H, W, ci, co, 3, 1, act
    rc = lib.capf_op_conv_wino_group(_stream(problems[0][0]), len(problems), descs, _wino_variant(problems[0][1], problems[0][0].shape[3]))
    if rc:
        raise CapfError(f"capf_op_conv_wino_group failed ({rc})")
    return outs


class ConvDesc(ctypes.Structure):
    """mirrors include/capf.h :: capf_conv_desc"""
    _fields_ = [("x", ctypes.c_void_p), ("w_packed", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("y", ctypes.c_void_p)] + \
               [(k, ctypes.c_int32) for k in ("B", "H", "W", "Cin", "Cout", "ks", "stride", "act")]


def conv_nhwc_group(problems):
    """problems: list of (x, wp, bias, ks, stride, act, residual) -> list of outputs; ONE grouped launch."""
    import torch
    lib = load_library()
    lib.capf_op_conv_group.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ConvDesc)]
    lib.capf_op_conv_group.restype = ctypes.c_int
    descs = (ConvDesc * len(problems))()
    outs = []
    for d, (x, wp, bias, ks, stride, act, residual) in zip(descs, problems):
        B, H, W, ci = x.shape
        co = wp.shape[0]
        pad = ks // 2
        ho, wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        y = torch.empty(B, ho, wo, co, device=x.device)
        outs.append(y)
        d.x, d.w_packed, d.bias, d.y = x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr()
        d.residual = residual.data_ptr() if residual is not None else None
        d.B, d.H, d.W, d.Cin, d.Cout, d.ks, d.stride, d.act = B, H, W, ci, co, ks, stride, act
    rc = lib.capf_op_conv_group(_stream(problems[0][0]), len(problems), descs)
    if rc:
        raise CapfError(f"capf_op_conv_group failed ({rc})")
    return outs


def pack_conv_bf16(w, bn=None, eps=1e-5):
    """-> (bf16 packed weights [Cout, Kpad64], fp32 bias [Cout])."""
    import torch
    lib = load_library()
    co, ci, ks, _ = w.shape
    kpad = (ks * ks * ci + 63) // 64 * 64
    wp = torch.empty(co, kpad, device=w.device, dtype=torch.bfloat16)
    bias = torch.empty(co, device=w.device)
    g, b, m, v = bn if bn is not None else (None, None, None, None)
    rc = lib.capf_op_pack_conv_bf16(_stream(w), _p(w.contiguous()), _p(g), _p(b), _p(m), _p(v), eps, _p(wp), _p(bias), co, ci, ks)
    if rc:
        raise CapfError(f"capf_op_pack_conv_bf16 failed ({rc})")
    return wp, bias


def conv_nhwc_bf16(x, wp, bias, ks, stride=1, act=0, residual=None):
    """x [B,H,W,Cin] cuda bf16 NHWC -> [B,Ho,Wo,Cout] bf16."""
    import torch
    lib = load_library()
    B, H, W, ci = x.shape
    co = wp.shape[0]
    pad = ks // 2
    ho, wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    y = torch.empty(B, ho, wo, co, device=x.device, dtype=torch.bfloat16)
    rc = lib.capf_op_conv_bf16(_stream(x), _p(x), _p(wp), _p(bias), _p(residual), _p(y), B, H, W, ci, co, ks, stride, act)
    if rc:
        raise CapfError(f"capf_op_conv_bf16 failed ({rc})")
    return y


def pack_conv_bf16_rh(w, bn=None, eps=1e-5):
    """3x3 weights for the row-halo bf16 conv -> (bf16 [Cout, 9 * Cin] in (kh, Cin / cw, kw, cw) order, fp32 bias, cw)."""
    import torch
    lib = load_library()
    co, ci, ks, _ = w.shape
    cw = lib.capf_op_conv_bf16_rh_width(ci)
    if ks != 3 or not cw:
        raise CapfError(f"row-halo conv needs a 3x3 kernel and Cin % 32 == 0 (got ks={ks}, Cin={ci})")
    wp = torch.empty(co, 9 * ci, device=w.device, dtype=torch.bfloat16)
    bias = torch.empty(co, device=w.device)
    g, b, m, v = bn if bn is not None else (None, None, None, None)
    rc = lib.capf_op_pack_conv_bf16_rh(_stream(w), _p(w.contiguous()), _p(g), _p(b), _p(m), _p(v), eps, _p(wp), _p(bias), co, ci)
    if rc:
        raise CapfError(f"capf_op_pack_conv_bf16_rh failed ({rc})")
    return wp, bias, cw


def conv_nhwc_bf16_rh(x, wp, bias, act=0, residual=None):
    """x [B,H,W,Cin] cuda bf16 NHWC -> [B,H,W,Cout] bf16 (3x3, stride 1, pad 1)."""
    import torch
    lib = load_library()
    B, H, W, ci = x.shape
    co = wp.shape[0]
    y = torch.empty(B, H, W, co, device=x.device, dtype=torch.bfloat16)
    rc = lib.capf_op_conv_bf16_rh(_stream(x), _p(x), _p(wp), _p(bias), _p(residual), _p(y), B, H, W, ci, co, act)
    if rc:
        raise CapfError(f"capf_op_conv_bf16_rh failed ({rc})")
    return y


def pack_conv_bf16_ws(w, bn=None, eps=1e-5):
    """3x3 weights for the 2-D halo bf16 conv tile (csrc/igemm_bf16_ws.hip) -> (packed bf16 [elems], fp32 bias [Cout])."""
    import torch
    lib = load_library()
    co, ci, ks, _ = w.shape
    n = lib.capf_op_conv_bf16_ws_pack_elems(co, ci)
    if ks != 3 or n <= 0 or co % 8:
        raise CapfError(f"2-D halo conv needs a 3x3 kernel, Cin % 16 == 0 and Cout % 8 == 0 (got ks={ks}, Cin={ci}, Cout={co})")
    wp = torch.empty(n, device=w.device, dtype=torch.bfloat16)
    bias = torch.empty(co, device=w.device)
    g, b, m, v = bn if bn is not None else (None, None, None, None)
    rc = lib.capf_op_pack_conv_bf16_ws(_stream(w), _p(w.contiguous()), _p(g), _p(b), _p(m), _p(v), eps, _p(wp), _p(bias), co, ci)
    if rc:
        raise CapfError(f"capf_op_pack_conv_bf16_ws failed ({rc})")
    return wp, bias


def conv_nhwc_bf16_ws_group(problems):
    """problems: list of (x, wp_ws, bias, act, residual, Cout), x / residual bf16 NHWC -> list of outputs (one grouped launch of the
    2-D halo tile, 3x3 / stride 1 / pad 1)."""
    import torch
    lib = load_library()
    n = len(problems)
    descs = (ConvDesc * n)()
    outs = []
    for i, (x, wp, bias, act, res, co) in enumerate(problems):
        B, H, W, ci = x.shape
        y = torch.empty(B, H, W, co, device=x.device, dtype=torch.bfloat16)
        outs.append(y)
        d = descs[i]
        d.x, d.w_packed, d.bias, d.y = x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr()
        d.residual = res.data_ptr() if res is not None else None
        d.B, d.H, d.W, d.Cin, d.Cout, d.ks, d.stride, d.act = B, H, W, ci, co, 3, 1, act
    lib.capf_op_conv_bf16_ws_group.argtypes = [c_void_p, c_int, POINTER(ConvDesc)]
    rc = lib.capf_op_conv_bf16_ws_group(_stream(problems[0][0]), n, descs)
    if rc:
        raise CapfError(f"capf_op_conv_bf16_ws_group failed ({rc})")
    return outs


def pack_conv_f32x3(w, bn=None, eps=1e-5):
    """3x3 weights for the split-fp32 conv tile (csrc/igemm_f32x3_ws.hip) -> (three bf16 pieces per weight, packed [elems]; fp32 bias [Cout])."""
    import torch
    lib = load_library()
    co, ci, ks, _ = w.shape
    n = lib.capf_op_conv_f32x3_pack_elems(co, ci)
    if ks != 3 or n <= 0 or co % 4:
        raise CapfError(f"split-fp32 conv needs a 3x3 kernel, Cin % 16 == 0 and Cout % 4 == 0 (got ks={ks}, Cin={ci}, Cout={co})")
    wp = torch.empty(n, device=w.device, dtype=torch.bfloat16)
    bias = torch.empty(co, device=w.device)
    g, b, m, v = bn if bn is not None else (None, None, None, None)
    rc = lib.capf_op_pack_conv_f32x3(_stream(w), _p(w.contiguous()), _p(g), _p(b), _p(m), _p(v), eps, _p(wp), _p(bias), co, ci)
    if rc:
        raise CapfError(f"capf_op_pack_conv_f32x3 failed ({rc})")
    return wp, bias


def conv_nhwc_f32x3_group(problems):
    """problems: list of (x, wp_x3, bias, act, residual, Cout), x / residual fp32 NHWC -> list of fp32 outputs (one grouped launch of the
    split-fp32 tile, 3x3 / stride 1 / pad 1)."""
    import torch
    lib = load_library()
    n = len(problems)
    descs = (ConvDesc * n)()
    outs = []
    for i, (x, wp, bias, act, res, co) in enumerate(problems):
        B, H, W, ci = x.shape
        y = torch.empty(B, H, W, co, device=x.device, dtype=torch.float32)
        outs.append(y)
        d = descs[i]
        d.x, d.w_packed, d.bias, d.y = x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr()
        d.residual = res.data_ptr() if res is not None else None
        d.B, d.H, d.W, d.Cin, d.Cout, d.ks, d.stride, d.act = B, H, W, ci, co, 3, 1, act
    lib.capf_op_conv_f32x3_group.argtypes = [c_void_p, c_int, POINTER(ConvDesc)]
    rc = lib.capf_op_conv_f32x3_group(_stream(problems[0][0]), n, descs)
    if rc:
        raise CapfError(f"capf_op_conv_f32x3_group failed ({rc})")
    return outs


def pack_conv_f32h2(w, bn=None, eps=1e-5):
    """3x3 weights for the default split-fp32 conv tile (csrc/igemm_f32h2_ws.hip) -> (packed [elems] int16: two fp16 pieces per weight under
    one power-of-two scale per output channel, then the fp32 inverse scales; fp32 bias [Cout])."""
    import torch
    lib = load_library()
    co, ci, ks, _ = w.shape
    n = lib.capf_op_conv_f32h2_pack_elems(co, ci)
    if ks != 3 or n <= 0 or co % 4:
        raise CapfError(f"split-fp32 conv needs a 3x3 kernel, Cin % 16 == 0 and Cout % 4 == 0 (got ks={ks}, Cin={ci}, Cout={co})")
    wp = torch.empty(n, device=w.device, dtype=torch.int16)
    bias = torch.empty(co, device=w.device)
    g, b, m, v = bn if bn is not None else (None, None, None, None)
    rc = lib.capf_op_pack_conv_f32h2(_stream(w), _p(w.contiguous()), _p(g), _p(b), _p(m), _p(v), eps, _p(wp), _p(bias), co, ci)
    if rc:
        raise CapfError(f"capf_op_pack_conv_f32h2 failed ({rc})")
    return wp, bias


def conv_nhwc_f32h2_group(problems):
    """problems: list of (x, wp_h2, bias, act, residual, Cout), x / residual fp32 NHWC -> list of fp32 outputs (the level's launches of the
    two-fp16-piece tile, 3x3 / stride 1 / pad 1)."""
    import torch
    lib = load_library()
    n = len(problems)
    descs = (ConvDesc * n)()
    outs = []
    for i, (x, wp, bias, act, res, co) in enumerate(problems):
        B, H, W, ci = x.shape
        y = torch.empty(B, H, W, co, device=x.device, dtype=torch.float32)
        outs.append(y)
        d = descs[i]
        d.x, d.w_packed, d.bias, d.y = x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr()
        d.residual = res.data_ptr() if res is not None else None
        d.B, d.H, d.W, d.Cin, d.Cout, d.ks, d.stride, d.act = B, H, W, ci, co, 3, 1, act
    lib.capf_op_conv_f32h2_group.argtypes = [c_void_p, c_int, POINTER(ConvDesc)]
    rc = lib.capf_op_conv_f32h2_group(_stream(problems[0][0]), n, descs)
    if rc:
        raise CapfError(f"capf_op_conv_f32h2_group failed ({rc})")
    return outs


def conv_nhwc_f32h2_planes(x, wp, bias, act, residual, cout, exps_in=None, planes_out=False):
    """One 3x3 / stride-1 conv on the two-fp16-piece tile with a PLANES tensor on one side (capf_op_conv_f32h2_planes): exps_in = the int32
    [tiles, Cin / 16] table of x's planes (x: the float32-typed tensor a planes_out conv returned); planes_out: y comes back as
    (float32-typed planes tensor, exps).  Use planes_to_fp32 to look at one."""
    import torch
    lib = load_library()
    B, H, W, ci = x.shape
    y = torch.empty(B, H, W, cout, device=x.device, dtype=torch.float32)
    d = ConvDesc()
    d.x, d.w_packed, d.bias, d.y = x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr()
    d.residual = residual.data_ptr() if residual is not None else None
    d.B, d.H, d.W, d.Cin, d.Cout, d.ks, d.stride, d.act = B, H, W, ci, cout, 3, 1, act
    tiles = lib.capf_op_conv_f32h2_tiles(B, H, W, None)
    eo = torch.zeros(tiles, cout // 16, device=x.device, dtype=torch.int32) if planes_out else None
    lib.capf_op_conv_f32h2_planes.argtypes = [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p]
    rc = lib.capf_op_conv_f32h2_planes(_stream(x), byref(d), _p(exps_in), _p(eo))
    if rc:
        raise CapfError(f"capf_op_conv_f32h2_planes failed ({rc})")
    return (y, eo) if planes_out else y


def f32h2_tile_pixels(B, H, W):
    """output pixels per tile of the two-fp16-piece conv tile at this geometry (flat pixel p belongs to tile p // that)"""
    lib = load_library()
    px = c_int(0)
    lib.capf_op_conv_f32h2_tiles.argtypes = [c_int, c_int, c_int, POINTER(c_int)]
    if lib.capf_op_conv_f32h2_tiles(B, H, W, byref(px)) <= 0:
        raise CapfError("geometry not eligible for the two-fp16-piece tile")
    return px.value


def planes_to_fp32(planes, exps, tile_pixels):
    """Decode a planes tensor [B, H, W, C] (float32-typed storage of [piece 0: 16 fp16 | piece 1: 16 fp16] per 16-channel chunk) with its
    [tiles, C / 16] exponent table into the fp32 values it stands for; tile_pixels: output pixels per tile (flat pixel p belongs to tile p // tile_pixels)."""
    import torch
    B, H, W, C = planes.shape
    h = planes.contiguous().view(torch.float16).view(B * H * W, C // 16, 2, 16).float()
    tile = torch.arange(B * H * W, device=planes.device) // tile_pixels
    scale = torch.exp2((127 - exps[tile].to(torch.float64))).to(torch.float32)                 # [pixels, C / 16]: 1 / 2^(se - 127)
    v = (h[:, :, 0, :] + h[:, :, 1, :]) * scale[:, :, None]
    return v.reshape(B, H, W, C)


def pack_f32h2_gemm(w, bn=None, eps=1e-5):
    """Weights for the two-fp16-piece GEMM (csrc/igemm_f32h2.hip): a conv filter [Cout, Cin, ks, ks] (BatchNorm folded if given) or an
    nn.Linear weight [N, K] -> (packed fp32-typed buffer [elems]: [N][Kpad] of {piece 0 | piece 1} chunks + [N] inverse scales; fp32 bias
    [N] for convs, None for linears)."""
    import torch
    lib = load_library()
    if w.dim() == 4:
        n, ci, ks, _ = w.shape
        k = ks * ks * ci
    else:
        (n, k), ci, ks = w.shape, 0, 0
    elems = lib.capf_op_f32h2_gemm_pack_elems(n, k)
    wp = torch.empty(elems, device=w.device, dtype=torch.float32)
    bias = torch.zeros(n, device=w.device) if w.dim() == 4 else None
    g, b, m, v = bn if bn is not None else (None, None, None, None)
    rc = lib.capf_op_pack_f32h2_gemm(_stream(w), _p(w.contiguous()), _p(g), _p(b), _p(m), _p(v), eps, _p(wp), _p(bias), n, ci, ks, k)
    if rc:
        raise CapfError(f"capf_op_pack_f32h2_gemm failed ({rc})")
    return wp, bias


def conv_nhwc_f32h2g(x, wp, bias, ks, stride=1, act=0, residual=None, cout=None):
    """y = act(conv2d(x; two-fp16-piece pack) + bias (+ residual)), NHWC fp32, padding ks // 2, one launch of igemm_f32h2g."""
    import torch
    lib = load_library()
    B, H, W, ci = x.shape
    co = cout if cout is not None else bias.numel()
    pad = ks // 2
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    y = torch.empty(B, Ho, Wo, co, device=x.device, dtype=torch.float32)
    rc = lib.capf_op_conv_f32h2g(_stream(x), _p(x), _p(wp), _p(bias), _p(residual), _p(y), B, H, W, ci, co, ks, stride, act)
    if rc:
        raise CapfError(f"capf_op_conv_f32h2g failed ({rc})")
    return y


def conv_nhwc_f32h2g_group(problems):
    """problems: list of (x, wp, bias, ks, stride, act, residual, Cout) -> list of outputs, ONE grid of igemm_f32h2g_group_kernel."""
    import torch
    lib = load_library()
    n = len(problems)
    descs = (ConvDesc * n)()
    outs = []
    for i, (x, wp, bias, ks, stride, act, res, co) in enumerate(problems):
        B, H, W, ci = x.shape
        pad = ks // 2
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        y = torch.empty(B, Ho, Wo, co, device=x.device, dtype=torch.float32)
        outs.append(y)
        d = descs[i]
        d.x, d.w_packed, d.bias, d.y = x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr()
        d.residual = res.data_ptr() if res is not None else None
        d.B, d.H, d.W, d.Cin, d.Cout, d.ks, d.stride, d.act = B, H, W, ci, co, ks, stride, act
    lib.capf_op_conv_f32h2g_group.argtypes = [c_void_p, c_int, POINTER(ConvDesc)]
    rc = lib.capf_op_conv_f32h2g_group(_stream(problems[0][0]), n, descs)
    if rc:
        raise CapfError(f"capf_op_conv_f32h2g_group failed ({rc})")
    return outs


def linear_ln_f32h2g(x, gamma, beta, eps, wp, bias, n, act=0, residual=None):
    """y[M, N] = act(LayerNorm(x[M, K]; gamma, beta, eps) @ W^T + bias (+ residual)) on the two-fp16-piece GEMM (K % 32 == 0, K <= 256, N % 4 == 0)."""
    import torch
    lib = load_library()
    M, K = x.shape
    y = torch.empty(M, n, device=x.device, dtype=torch.float32)
    rc = lib.capf_op_linear_ln_f32h2g(_stream(x), _p(x), _p(gamma), _p(beta), float(eps), _p(wp), _p(bias), _p(residual), _p(y), M, n, K, act)
    if rc:
        raise CapfError(f"capf_op_linear_ln_f32h2g failed ({rc})")
    return y


def linear_f32h2g(x, wp, bias, n, act=0, residual=None):
    """y[M, N] = act(x[M, K] @ W^T + bias (+ residual)) on the two-fp16-piece GEMM (K % 32 == 0, N % 4 == 0)."""
    import torch
    lib = load_library()
    M, K = x.shape
    y = torch.empty(M, n, device=x.device, dtype=torch.float32)
    rc = lib.capf_op_linear_f32h2g(_stream(x), _p(x), _p(wp), _p(bias), _p(residual), _p(y), M, n, K, act)
    if rc:
        raise CapfError(f"capf_op_linear_f32h2g failed ({rc})")
    return y


def wgrad(dy, x, two_piece=True):
    """(dW[N, K], db[N]) = (dy[M, N]^T @ x[M, K], column sums of dy) through the training step's weight-gradient kernels."""
    import torch
    lib = load_library()
    M, N = dy.shape
    K = x.shape[1]
    out = torch.empty(N * K + N, device=x.device, dtype=torch.float32)
    rc = lib.capf_op_wgrad(_stream(x), _p(dy), _p(x), M, N, K, _p(out), 1 if two_piece else 0)
    if rc:
        raise CapfError(f"capf_op_wgrad failed ({rc})")
    return out[:N * K].view(N, K), out[N * K:]


def conv_nhwc_bf16_group(problems):
    """problems: list of (x, wp, bias, ks, stride, act, residual, wp_row_halo or None), all bf16 NHWC -> (outputs, variant)."""
    import torch
    lib = load_library()
    n = len(problems)
    descs = (ConvDesc * n)()
    rh = (c_void_p * n)()
    outs = []
    for i, (x, wp, bias, ks, stride, act, res, wrh) in enumerate(problems):
        B, H, W, ci = x.shape
        co = wp.shape[0]
        pad = ks // 2
        ho, wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        y = torch.empty(B, ho, wo, co, device=x.device, dtype=torch.bfloat16)
        outs.append(y)
        d = descs[i]
        d.x, d.w_packed, d.bias, d.y = x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr()
        d.residual = res.data_ptr() if res is not None else None
        d.B, d.H, d.W, d.Cin, d.Cout, d.ks, d.stride, d.act = B, H, W, ci, co, ks, stride, act
        rh[i] = wrh.data_ptr() if wrh is not None else None
    variant = c_int32(-1)
    lib.capf_op_conv_bf16_group.argtypes = [c_void_p, c_int, POINTER(ConvDesc), POINTER(c_void_p), POINTER(c_int32)]
    rc = lib.capf_op_conv_bf16_group(_stream(problems[0][0]), n, descs, rh, byref(variant))
    if rc:
        raise CapfError(f"capf_op_conv_bf16_group failed ({rc})")
    return outs, variant.value


def linear(x, w, bias=None, act=0, residual=None):
    import torch
    lib = load_library()
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device)
    rc = lib.capf_op_linear(_stream(x), _p(x), _p(w), _p(bias), _p(residual), _p(y), M, N, K, act)
    if rc:
        raise CapfError(f"capf_op_linear failed ({rc})")
    return y


def bilinear_corners(grid, H, W, border):
    """grid: CUDA fp32 [..., 2] normalised (x, y) -> (idx int32 [..., 2] = NW corner (x0, y0), frac fp32 [..., 2]) by the
    device function both sampling sites of capf_forward use."""
    import torch
    lib = load_library()
    g = grid.contiguous()
    n = g.numel() // 2
    idx = torch.empty(g.shape, dtype=torch.int32, device=g.device)
    frac = torch.empty(g.shape, dtype=torch.float32, device=g.device)
    rc = lib.capf_op_bilinear_corners(_stream(g), _p(g), n, int(H), int(W), 1 if border else 0, _p(idx), _p(frac))
    if rc:
        raise CapfError(f"capf_op_bilinear_corners failed ({rc})")
    return idx, frac


def linear_bf16(x, w, bias=None, residual=None, gelu=False):
    """x bf16 [M,K], w bf16 [N,K] -> fp32 [M,N] (+ fp32 residual), or with gelu=True -> bf16 GELU(x w^T + b)."""
    import torch
    lib = load_library()
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.bfloat16 if gelu else torch.float32)
    rc = lib.capf_op_linear_bf16(_stream(x), _p(x.contiguous()), _p(w.contiguous()), _p(bias), _p(residual), _p(y), M, N, K, 1 if gelu else 0)
    if rc:
        raise CapfError(f"capf_op_linear_bf16 failed ({rc})")
    return y


# ---- neighbours of the path: N1 preprocessing, N2 flip-test fusion -------------------------------------
HRNET_MEAN, HRNET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # datasets/utils.py:24-26
CPN_MEAN = tuple(v / 255.0 for v in (122.7717, 115.9465, 102.9801))             # :27-29


def preprocess(images_u8, gt, k2d, kcrop, backbone="hrnet_32", mode=0):
    """uint8 BGR [B,H,W,3] + labels (all CUDA) -> (images fp32 RGB NHWC, gt root-relative, k2d, kcrop).
    mode 0 plain, 1 train-time horizontal flip, 2 flip-test ([2,B,...] outputs: original then mirrored)."""
    import torch
    lib = load_library()
    B, H, W, _ = images_u8.shape
    nsets = 2 if mode == 2 else 1
    dev = images_u8.device
    img_out = torch.empty((nsets, B, H, W, 3) if nsets == 2 else (B, H, W, 3), dtype=torch.float32, device=dev)
    k2d_out = torch.empty((nsets, B, 17, 2) if nsets == 2 else (B, 17, 2), dtype=torch.float32, device=dev)
    kc_out = torch.empty_like(k2d_out)
    gt_out = torch.empty_like(gt) if gt is not None else None
    if backbone == "cpn":
        mean = torch.tensor([122.7717, 115.9465, 102.9801]) / 255.0              # fp32 division, like the reference
        std = None
    else:
        mean, std = torch.tensor(HRNET_MEAN), torch.tensor(HRNET_STD)
    m3 = (c_float * 3)(*mean.tolist())
    s3 = (c_float * 3)(*std.tolist()) if std is not None else None
    rc = lib.capf_preprocess(_stream(images_u8), _p(images_u8.contiguous()), B, H, W, m3, s3, mode, _p(img_out),
                             _p(gt.contiguous()) if gt is not None else c_void_p(0), _p(gt_out), _p(k2d.contiguous()), _p(k2d_out),
                             _p(kcrop.contiguous()), _p(kc_out))
    if rc:
        raise CapfError(f"capf_preprocess failed ({rc})")
    return img_out, gt_out, k2d_out, kc_out


def fliptest_fuse(pred2):
    """pred2 [2,B,1,17,3] (original, mirrored) -> [B,1,17,3]  (train.py:177-180)."""
    import torch
    lib = load_library()
    B = pred2.shape[1]
    out = torch.empty(B, 1, 17, 3, dtype=torch.float32, device=pred2.device)
    rc = lib.capf_fliptest_fuse(_stream(pred2), _p(pred2.contiguous()), B, _p(out))
    if rc:
        raise CapfError(f"capf_fliptest_fuse failed ({rc})")
    return out


# ---- N3: affine crop (mvn/utils/img.py) ---------------------------------------------------------------
def affine_from_center_scale(center, scale, output_size):
    """get_affine_transform(center, scale, 0, output_size) -> 2x3 float64 numpy matrix (host code, no GPU)."""
    import numpy as np
    lib = load_library()
    lib.capf_affine_from_center_scale.argtypes = [POINTER(c_double), POINTER(c_double), c_int, c_int, POINTER(c_double)]
    c = (c_double * 2)(float(center[0]), float(center[1]))
    sc = (c_double * 2)(float(scale[0]), float(scale[1]))
    m = (c_double * 6)()
    rc = lib.capf_affine_from_center_scale(c, sc, int(output_size[0]), int(output_size[1]), m)
    if rc:
        raise CapfError(f"capf_affine_from_center_scale failed ({rc})")
    return np.array(list(m), dtype=np.float64).reshape(2, 3)


def warp_affine(frames, mats, output_size):
    """frames: list of uint8 CUDA tensors [H_i, W_i, 3] (BGR as cv2.imread gives them); mats: [B, 2, 3] float64
    forward matrices (numpy or tensor); output_size = (out_w, out_h) -> uint8 CUDA tensor [B, out_h, out_w, 3]."""
    import numpy as np
    import torch
    lib = load_library()
    lib.capf_warp_affine.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    dev = frames[0].device
    B = len(frames)
    for f in frames:
        if f.dtype != torch.uint8 or f.dim() != 3 or f.shape[2] != 3 or f.stride(2) != 1 or f.stride(1) != 3 or not f.is_cuda:
            raise ValueError("frames must be uint8 CUDA tensors [H, W, 3] with packed pixels")
    ptrs = torch.tensor([f.data_ptr() for f in frames], dtype=torch.int64).to(dev)
    dims = torch.tensor([[f.shape[0], f.shape[1], f.stride(0)] for f in frames], dtype=torch.int32).to(dev)
    m = torch.as_tensor(np.asarray(mats, dtype=np.float64).reshape(B, 6)).to(dev)
    out_w, out_h = int(output_size[0]), int(output_size[1])
    out = torch.empty(B, out_h, out_w, 3, dtype=torch.uint8, device=dev)
    rc = lib.capf_warp_affine(_stream(out), _p(ptrs), _p(dims), _p(m), B, out_h, out_w, _p(out))
    if rc:
        raise CapfError(f"capf_warp_affine failed ({rc})")
    return out


def jpeg_info(data):
    """data: bytes of a JPEG file -> dict(width, height, components, h_samp, v_samp, scratch_bytes); CapfError for what capf_jpeg_decode does
    not take (progressive, arithmetic, CMYK, ...).  Host code, no GPU."""
    lib = load_library()
    lib.capf_jpeg_info.argtypes = [c_char_p, c_size_t] + [POINTER(c_int32)] * 5 + [POINTER(c_size_t)]
    w, h, nc, hs, vs, sb = c_int32(), c_int32(), c_int32(), c_int32(), c_int32(), c_size_t()
    rc = lib.capf_jpeg_info(data, len(data), byref(w), byref(h), byref(nc), byref(hs), byref(vs), byref(sb))
    if rc:
        raise CapfError(f"capf_jpeg_info: not a JPEG this path decodes ({rc})")
    return dict(width=w.value, height=h.value, components=nc.value, h_samp=hs.value, v_samp=vs.value, scratch_bytes=sb.value)


def jpeg_coefficients(data):
    """The host half of the decoder: quantised DCT coefficients (natural order) per component, list of int16 numpy arrays
    [block rows, block cols, 64] over the MCU-padded image.  No GPU."""
    import numpy as np
    lib = load_library()
    info = jpeg_info(data)
    lib.capf_jpeg_coefficients.argtypes = [c_char_p, c_size_t, c_void_p, c_size_t]
    hs, vs, nc = info["h_samp"], info["v_samp"], info["components"]
    mx, my = -(-info["width"] // (8 * hs)), -(-info["height"] // (8 * vs))
    shapes = [(my * vs, mx * hs)] + [(my, mx)] * (nc - 1)
    total = sum(a * b * 64 for a, b in shapes)
    buf = np.zeros(total, np.int16)
    rc = lib.capf_jpeg_coefficients(data, len(data), buf.ctypes.data_as(c_void_p), total)
    if rc:
        raise CapfError(f"capf_jpeg_coefficients failed ({rc})")
    out, o = [], 0
    for a, b in shapes:
        out.append(buf[o:o + a * b * 64].reshape(a, b, 64))
        o += a * b * 64
    return out


def jpeg_decode(data, device="cuda"):
    """cv2.imread(..., IMREAD_COLOR) of a baseline JPEG held in memory: bytes -> uint8 CUDA tensor [H, W, 3], BGR (host Huffman decode, GPU
    IDCT / upsampling / colour conversion; bit-exact against libjpeg-turbo's default decode)."""
    import torch
    lib = load_library()
    info = jpeg_info(data)
    lib.capf_jpeg_decode.argtypes = [c_void_p, c_char_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]
    out = torch.empty(info["height"], info["width"], 3, dtype=torch.uint8, device=device)
    scratch = torch.empty(info["scratch_bytes"], dtype=torch.uint8, device=device)
    rc = lib.capf_jpeg_decode(_stream(out), data, len(data), _p(out), out.stride(0), _p(scratch), scratch.numel())
    if rc:
        raise CapfError(f"capf_jpeg_decode failed ({rc})")
    return out


# ---- N2: evaluation metrics (mvn/models/loss.py:25-101, datasets/human36m.py:358-417) ---------------------------
def pose_errors(pred, gt, prev=None):
    """pred / gt: CUDA fp32 [n, J, 3]; prev: CUDA int32 [n] or None (= i-1).  -> CUDA fp32 [n, 4] =
    per-pose {MPJPE, P_MPJPE, N_MPJPE, velocity error}."""
    import torch
    lib = load_library()
    n, J, _ = pred.shape
    err = torch.empty(n, 4, dtype=torch.float32, device=pred.device)
    rc = lib.capf_pose_errors(_stream(pred), _p(pred.contiguous()), _p(gt.contiguous()), n, J, _p(prev), _p(err))
    if rc:
        raise CapfError(f"capf_pose_errors failed ({rc})")
    return err


def segment_sums(err, segment=None, prev=None, n_segments=1):
    """-> (sums float64 [n_segments, 4], counts int32 [n_segments, 2]) on the device."""
    import torch
    lib = load_library()
    n = err.shape[0]
    sums = torch.empty(n_segments, 4, dtype=torch.float64, device=err.device)
    counts = torch.empty(n_segments, 2, dtype=torch.int32, device=err.device)
    rc = lib.capf_segment_sums(_stream(err), _p(err), _p(segment), _p(prev), n, n_segments, _p(sums), _p(counts))
    if rc:
        raise CapfError(f"capf_segment_sums failed ({rc})")
    return sums, counts


def keypoints_loss(mode, pred, gt, validity, threshold=0.0, want_grad=False):
    """mode 0 MSE, 1 MSESmooth, 2 MAE (loss.py:104-137).  pred / gt [..., D], validity [..., 1] -> (loss[1], dpred | None)."""
    import torch
    lib = load_library()
    D = pred.shape[-1]
    rows = pred.numel() // D
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred, memory_format=torch.contiguous_format) if want_grad else None
    v = validity.to(torch.float32).expand(*pred.shape[:-1], 1).contiguous()
    rc = lib.capf_keypoints_loss(_stream(pred), int(mode), _p(pred.contiguous()), _p(gt.contiguous()), _p(v), rows, D,
                                 float(threshold), _p(loss), _p(dpred))
    if rc:
        raise CapfError(f"capf_keypoints_loss failed ({rc})")
    return loss, dpred
