#!/usr/bin/env python
"""Does the fused Adam kernel's bandwidth depend on WHERE its four arrays sit relative to one another?  (Round 6: the same kernel ran 257 us with one
library build and 298 us with the next on the same box -- the only difference was the size of a scratch buffer, i.e. the addresses torch's allocator
handed to the gradients.)  One 48 M-float parameter (the SH tensor of 1 M Gaussians), p / g / m / v carved from ONE buffer at controlled relative offsets:
every array starts at k * STRIDE + k * delta for a sweep of delta; 30 launches each, median of HIP-event times.
    python tools/gpu_adam_alignment_probe.py            -> one JSON line"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch
from diff_gaussian_rasterization import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
N = 48_000_000
BYTES = N * 4
STRIDE = ((BYTES + (2 << 20) - 1) // (2 << 20)) * (2 << 20)      # what a 2 MiB-granular allocator would give: bases congruent mod 2 MiB
SLACK = 64 << 20
buf = torch.zeros(4 * STRIDE + 4 * SLACK, dtype=torch.uint8, device=dev)
base = buf.data_ptr()
base_al = (base + (2 << 20) - 1) // (2 << 20) * (2 << 20) - base      # start the carving at a 2 MiB boundary


def view(off):
    return buf[base_al + off: base_al + off + BYTES].view(torch.float32)


def time_at(offsets, reps=30):
    p, g, m, v = (view(o) for o in offsets)
    g.normal_()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(5):
        _lib.check(lib.gsr_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), N, 1e-4, 0.9, 0.999, 1e-15, i + 1, st), "adam")
    ev[0].record()
    for i in range(reps):
        _lib.check(lib.gsr_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), N, 1e-4, 0.9, 0.999, 1e-15, i + 6, st), "adam")
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return round(t[len(t) // 2] * 1e3, 1)


out = {"N_floats": N, "bytes_moved_per_launch": 7 * BYTES, "stride_between_arrays": STRIDE, "us_by_delta": {}}
for delta in (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 19, 5 << 18, 7 << 17, 12345 * 256):
    out["us_by_delta"][str(delta)] = time_at([k * STRIDE + k * delta for k in range(4)])
# gradient alone moved (what the backward's allocation decides), the other three as a 2 MiB-granular allocator leaves them
out["us_by_gradient_offset_only"] = {str(d): time_at([0, STRIDE + d, 2 * STRIDE, 3 * STRIDE]) for d in (0, 4096, 65536, 1 << 20, 5 << 18)}
best = min(out["us_by_delta"].values())
out["GBs_best"] = round(7 * BYTES / best / 1e3, 1)
out["GBs_worst"] = round(7 * BYTES / max(out["us_by_delta"].values()) / 1e3, 1)
print(json.dumps(out))
