"""Fused Adam for the Gaussian parameters -- drop-in for the torch.optim.Adam instance the reference
builds in scene/gaussian_model.py:178-211 (per-group learning rates, eps=1e-15, updated every iteration by
update_learning_rate, train.py:91) and steps at train.py:177-186.  One HIP kernel for all parameter tensors
(gsr_adam_step_multi in libgsr_hip.so) instead of torch's multi-kernel foreach implementation.  State-dict layout
(`step`, `exp_avg`, `exp_avg_sq`) matches torch.optim.Adam so checkpoints (train.py:188-190) interchange."""
from __future__ import annotations

import ctypes as C

import torch

from diff_gaussian_rasterization import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        # all tensors of a device in ONE launch (gsr_adam_step_multi): a 3DGS model is six small-to-large tensors
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.GsrError("FusedAdam needs contiguous fp32 parameters on a HIP device (no CPU path)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                t = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                                    float(group["lr"]), float(b1), float(b2), float(group["eps"]), int(st["step"]), 0)
                batches.setdefault(p.device, []).append((t, g))      # (g kept alive until the launch)
        for dev, items in batches.items():
            arr = (_lib.AdamTensor * len(items))(*[t for t, _ in items])
            with torch.cuda.device(dev):
                _lib.check(lib.gsr_adam_step_multi(arr, len(items), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                           "gsr_adam_step_multi")
        return loss
