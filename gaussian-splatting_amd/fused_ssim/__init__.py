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

__all__ = ["fused_ssim", "fused_train_loss", "FusedSSIMMap", "FusedSSIMMean", "FusedTrainLoss"]


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


class FusedTrainLoss(torch.autograd.Function):
    """loss = (1 - lambda) L1(img1, img2) + lambda (1 - SSIM(img1, img2)) (train.py:119-126) in the two SSIM kernels: one read
    of both images forward, dL/dimg1 written once backward (gsr_train_loss_forward / _backward).  Returns (loss, parts):
    a 0-dim loss and the non-differentiable read-outs parts = [L1, SSIM] (two views of one 3-float buffer the kernel fills,
    so that loss.backward() hands the backward a 0-dim gradient and no slicing kernel runs)."""

    @staticmethod
    def forward(ctx, img1, img2, lambda_dssim):
        lib = _lib.load()
        if not img1.is_cuda or not img2.is_cuda:
            raise _lib.GsrError("fused_train_loss needs HIP tensors ('cuda'); there is no CPU path")
        a = img1.contiguous().float()
        b = img2.contiguous().float()
        Bn, Cn, H, W = a.shape
        planes = Bn * Cn
        need = img1.requires_grad
        d1 = torch.empty_like(a) if need else None
        d2 = torch.empty_like(a) if need else None
        d3 = torch.empty_like(a) if need else None
        partials = torch.empty(2 * int(lib.gsr_ssim_partial_count(planes, H, W)), dtype=torch.float32, device=a.device)
        out = torch.empty(3, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
            _lib.check(lib.gsr_train_loss_forward(planes, H, W, _p(a), _p(b), float(lambda_dssim), _p(partials), _p(out), _p(d1),
                                                  _p(d2), _p(d3), st), "gsr_train_loss_forward")
        if need:
            ctx.save_for_backward(a, b, d1, d2, d3)
        ctx.need, ctx.lam = need, float(lambda_dssim)
        loss, parts = out[0], out[1:3]
        ctx.mark_non_differentiable(parts)
        return loss, parts

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        if not ctx.need:
            return None, None, None
        lib = _lib.load()
        a, b, d1, d2, d3 = ctx.saved_tensors
        Bn, Cn, H, W = a.shape
        g = g_loss.contiguous().float().reshape(1)          # dL/dloss as one device scalar
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
            _lib.check(lib.gsr_train_loss_backward(Bn * Cn, H, W, _p(a), _p(b), _p(g), ctx.lam, _p(d1), _p(d2), _p(d3), _p(out), st),
                       "gsr_train_loss_backward")
        return out, None, None


def fused_train_loss(image, gt_image, lambda_dssim: float = 0.2, return_parts: bool = False):
    """The reference's training loss (train.py:119-126: Ll1 = l1_loss(image, gt); ssim_value = fused_ssim(image, gt);
    loss = (1 - lambda_dssim) * Ll1 + lambda_dssim * (1 - ssim_value)) as ONE forward and ONE backward kernel pair.
    image / gt_image: [3,H,W] or [B,3,H,W].  return_parts: also the (detached) L1 and SSIM values for the progress bar."""
    if image.dim() == 3:
        image, gt_image = image.unsqueeze(0), gt_image.unsqueeze(0)
    if image.numel() == 0:
        raise _lib.GsrError("fused_train_loss: empty image")
    loss, parts = FusedTrainLoss.apply(image, gt_image, lambda_dssim)
    if return_parts:
        return loss, parts[0], parts[1]
    return loss


def fused_ssim(img1, img2, padding="same", train=True):
    if padding != "same":
        raise NotImplementedError("only padding='same' (the reference's call form) is implemented")
    if img1.dim() == 3:
        img1, img2 = img1.unsqueeze(0), img2.unsqueeze(0)
    if img1.numel() == 0:
        return FusedSSIMMap.apply(img1, img2, train).mean()
    return FusedSSIMMean.apply(img1, img2, train)
