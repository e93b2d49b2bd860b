"""Fused AdamW over the lifter's parameters (train.py:337-345: optim.AdamW(volume_net params, lr, wd=0.1)).

`flatten_(module)` re-homes every parameter of `module` as a view into ONE flat fp32 buffer (state_dict,
checkpoints and torch optimizers keep working: only .data storage changes), so the update is a single
kernel over 14 M elements (capf_adamw_step) fed by the flat gradient capf_backward writes, and the
data-parallel exchange is a single all-reduce of that same buffer."""
import ctypes

import torch


def flatten_(module):
    """Moves parameter STORAGE (`p.data = view of the flat buffer`).  CA_PF notices on its next forward — it
    fingerprints the data pointers of volume_net's parameters per call (mvn/models/conpose.py::_engine) — and
    re-borrows every pointer, so this may be called before or after the first forward."""
    params = list(module.parameters())
    flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view(p.shape)
        off += n
    return flat


class FusedAdamW:
    """torch.optim.AdamW semantics (decoupled weight decay on every parameter, bias-corrected moments)."""

    def __init__(self, flat_params, lr=6.4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1):
        from .lib import load_library
        self.lib = load_library()
        self.p = flat_params
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat_params)
        self.v = torch.zeros_like(flat_params)
        self.t = 0

    def step(self, flat_grad, grad_scale=1.0):
        """grad_scale: multiplied into every gradient element inside the kernel (1 / world_size after a SUM all-reduce)."""
        self.t += 1
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.p.device).cuda_stream)
        rc = self.lib.capf_adamw_step(stream, P(self.p), P(flat_grad), P(self.m), P(self.v), self.p.numel(), self.lr,
                                      self.betas[0], self.betas[1], self.eps, self.wd, self.t, float(grad_scale))
        if rc:
            raise RuntimeError(f"capf_adamw_step failed ({rc})")
