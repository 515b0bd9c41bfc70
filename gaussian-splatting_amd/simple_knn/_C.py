"""`from simple_knn._C import distCUDA2` (scene/gaussian_model.py:21) -- SURVEY.md 8(f) N3.

distCUDA2(points[N,3] float32 on the device) -> float32[N]: for every point the mean of the squared distances to its
three nearest other points; the reference turns it into the initial Gaussian scale,
`scales = log(sqrt(clamp_min(dist2, 1e-7)))` (scene/gaussian_model.py:159-160).  Computed by hand-written HIP kernels
behind the C ABI (gsr_knn_mean_dist2, csrc/knn.hip); there is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch

from diff_gaussian_rasterization import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    if not points.is_cuda:
        raise _lib.GsrError("distCUDA2 needs a tensor on a HIP device ('cuda'); there is no CPU path")
    if points.dim() != 2 or points.shape[1] != 3:
        raise _lib.GsrError(f"distCUDA2 expects points of shape [N, 3], got {tuple(points.shape)}")
    pts = points.detach()
    if pts.dtype != torch.float32:
        pts = pts.float()
    pts = pts.contiguous()
    N = int(pts.shape[0])
    out = torch.empty(N, dtype=torch.float32, device=pts.device)
    if N == 0:
        return out
    with torch.cuda.device(pts.device):
        scratch = torch.empty(int(lib.gsr_knn_scratch_bytes(N)), dtype=torch.uint8, device=pts.device)
        _lib.check(lib.gsr_knn_mean_dist2(N, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()),
                                          C.c_void_p(scratch.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)),
                   "gsr_knn_mean_dist2")
    return out
