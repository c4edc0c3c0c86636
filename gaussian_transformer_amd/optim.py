"""Adam over the Gaussian parameter groups through ONE HIP launch (include/gsr_optim.h, csrc/adam.hip).

A torch.optim.Optimizer with the state layout of torch.optim.Adam ("step", "exp_avg", "exp_avg_sq" per tensor), so the
optimiser-state surgery of densify.py -- and anything else written against the reference's optimiser
(scene/gaussian_model.py:155-164: six one-tensor groups, eps 1e-15) -- works on it unchanged.  No weight decay, no
amsgrad, like the reference's.  There is no CPU fallback: parameters must be float32 tensors on a HIP device.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        by_hyper = {}
        keep = []                                   # tensors that must outlive the launch call
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.device.type != "cuda" or not p.is_contiguous():
                    raise _lib.GsrError("HipAdam needs contiguous float32 parameters on a HIP device (no CPU fallback)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] = int(st["step"]) + 1
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(grad)
                key = (p.device, tuple(g["betas"]), float(g["eps"]))
                by_hyper.setdefault(key, []).append(
                    _lib.AdamGroup(p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                   p.numel(), float(g["lr"]), st["step"]))
        for (dev, betas, eps), groups in by_hyper.items():
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                for i in range(0, len(groups), _lib.ADAM_MAX_GROUPS):
                    chunk = groups[i:i + _lib.ADAM_MAX_GROUPS]
                    arr = (_lib.AdamGroup * len(chunk))(*chunk)
                    _lib.check(lib.gsr_adam_step(stream, len(chunk), arr, betas[0], betas[1], eps), "gsr_adam_step")
        return loss
