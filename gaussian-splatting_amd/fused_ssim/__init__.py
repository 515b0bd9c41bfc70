"""MI355X drop-in for the reference's optional `fused_ssim` package (train.py:31-35: `from fused_ssim import fused_ssim`,
called as `fused_ssim(image.unsqueeze(0), gt_image.unsqueeze(0))` at train.py:122).  SURVEY.md 8(f) N1.

    fused_ssim(img1[B,C,H,W], img2[B,C,H,W], padding="same", train=True) -> mean SSIM (scalar tensor)

Same numerics as utils/loss_utils.py:56-87 (the reference's un-fused fallback): 11x11 Gaussian window (sigma 1.5), zero
padding 5, C1 = 0.01^2, C2 = 0.03^2.  Gradient flows to img1 only (the rendered image), as in the reference package.
All compute is in libgsr_hip.so (gsr_ssim_forward / gsr_ssim_backward); there is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch

from diff_gaussian_rasterization import _lib

__all__ = ["fused_ssim", "FusedSSIMMap", "FusedSSIMMean"]


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class FusedSSIMMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, train=True):
        lib = _lib.load()
        if not img1.is_cuda or not img2.is_cuda:
            raise _lib.GsrError("fused_ssim needs HIP tensors ('cuda'); there is no CPU path")
        a = img1.contiguous().float()
        b = img2.contiguous().float()
        Bn, Cn, H, W = a.shape
        planes = Bn * Cn
        ssim_map = torch.empty_like(a)
        need = train and img1.requires_grad
        d1 = torch.empty_like(a) if need else None
        d2 = torch.empty_like(a) if need else None
        d3 = torch.empty_like(a) if need else None
        with torch.cuda.device(a.device):
            st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
            _lib.check(lib.gsr_ssim_forward(planes, H, W, _p(a), _p(b), _p(ssim_map), _p(d1), _p(d2), _p(d3), st),
                       "gsr_ssim_forward")
        if need:
            ctx.save_for_backward(a, b, d1, d2, d3)
        ctx.need = need
        return ssim_map

    @staticmethod
    def backward(ctx, dL_dmap):
        if not ctx.need:
            return None, None, None
        lib = _lib.load()
        a, b, d1, d2, d3 = ctx.saved_tensors
        Bn, Cn, H, W = a.shape
        g = dL_dmap.contiguous().float()
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
            _lib.check(lib.gsr_ssim_backward(Bn * Cn, H, W, _p(a), _p(b), _p(g), _p(d1), _p(d2), _p(d3), _p(out), st),
                       "gsr_ssim_backward")
        return out, None, None


class FusedSSIMMean(torch.autograd.Function):
    """mean(SSIM map) without materialising the map: per-tile partial sums in the forward kernel, one device scalar
    dL/dmean into the backward kernel (gsr_ssim_mean_forward / gsr_ssim_mean_backward)."""

    @staticmethod
    def forward(ctx, img1, img2, train=True):
        lib = _lib.load()
        if not img1.is_cuda or not img2.is_cuda:
            raise _lib.GsrError("fused_ssim needs HIP tensors ('cuda'); there is no CPU path")
        a = img1.contiguous().float()
        b = img2.contiguous().float()
        Bn, Cn, H, W = a.shape
        planes = Bn * Cn
        need = train and img1.requires_grad
        d1 = torch.empty_like(a) if need else None
        d2 = torch.empty_like(a) if need else None
        d3 = torch.empty_like(a) if need else None
        partials = torch.empty(int(lib.gsr_ssim_partial_count(planes, H, W)), dtype=torch.float32, device=a.device)
        mean = torch.empty((), dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
            _lib.check(lib.gsr_ssim_mean_forward(planes, H, W, _p(a), _p(b), _p(partials), _p(mean), _p(d1), _p(d2), _p(d3), st),
                       "gsr_ssim_mean_forward")
        if need:
            ctx.save_for_backward(a, b, d1, d2, d3)
        ctx.need = need
        return mean

    @staticmethod
    def backward(ctx, dL_dmean):
        if not ctx.need:
            return None, None, None
        lib = _lib.load()
        a, b, d1, d2, d3 = ctx.saved_tensors
        Bn, Cn, H, W = a.shape
        g = dL_dmean.contiguous().float().reshape(1)
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
            _lib.check(lib.gsr_ssim_mean_backward(Bn * Cn, H, W, _p(a), _p(b), _p(g), _p(d1), _p(d2), _p(d3), _p(out), st),
                       "gsr_ssim_mean_backward")
        return out, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    if padding != "same":
        raise NotImplementedError("only padding='same' (the reference's call form) is implemented")
    if img1.dim() == 3:
        img1, img2 = img1.unsqueeze(0), img2.unsqueeze(0)
    if img1.numel() == 0:
        return FusedSSIMMap.apply(img1, img2, train).mean()
    return FusedSSIMMean.apply(img1, img2, train)
