#!/usr/bin/env python
"""Is the fused Adam's fast / slow mode (257 or 298 us for the same bytes, per process) a property of WHERE its arrays were allocated?  In ONE
process: the 59 M floats of a 1 M-Gaussian model as five tensors with torch-allocated p / g / m / v; the multi-tensor kernel timed (30 launches, median);
then m and v -- the arrays the optimizer owns -- are freed and allocated again behind a dummy of another size, timed again, eight times; then the same for
g, then for p.    python tools/gpu_adam_realloc_probe.py -> one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch
from gsr_optim import FusedAdam

dev = torch.device("cuda:0")
P = 1_000_000
shapes = [(P, 3), (P, 16, 3), (P, 1), (P, 3), (P, 4)]


def timed(opt, reps=30):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for _ in range(5):
        opt.step()
    ev[0].record()
    for i in range(reps):
        opt.step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return round(t[len(t) // 2] * 1e3, 1)


params = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
for p in params:
    p.grad = torch.randn_like(p)
opt = FusedAdam(params, lr=1e-5, eps=1e-15)
out = {"first_us": timed(opt), "after_reallocating_m_v": [], "after_reallocating_g": [], "after_reallocating_p_and_everything": []}
hold = []
for k in range(8):
    for p in params:
        st = opt.state[p]
        st["exp_avg"] = None
        st["exp_avg_sq"] = None
    torch.cuda.empty_cache()
    hold.append(torch.empty((k + 1) * 3_333_333, dtype=torch.uint8, device=dev))
    for p in params:
        st = opt.state[p]
        st["exp_avg"] = torch.zeros_like(p)
        st["exp_avg_sq"] = torch.zeros_like(p)
    out["after_reallocating_m_v"].append(timed(opt))
for k in range(6):
    for p in params:
        p.grad = None
    torch.cuda.empty_cache()
    hold.append(torch.empty((k + 1) * 5_555_555, dtype=torch.uint8, device=dev))
    for p in params:
        p.grad = torch.randn_like(p)
    out["after_reallocating_g"].append(timed(opt))
for k in range(6):
    del opt, params
    torch.cuda.empty_cache()
    hold.append(torch.empty((k + 1) * 7_777_777, dtype=torch.uint8, device=dev))
    params = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    for p in params:
        p.grad = torch.randn_like(p)
    opt = FusedAdam(params, lr=1e-5, eps=1e-15)
    out["after_reallocating_p_and_everything"].append(timed(opt))
print(json.dumps(out))
