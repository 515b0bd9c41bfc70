"""`diff_gaussian_rasterization._C` -- the import surface of the reference's native extension module, answered by libgsr_hip.so.

In-tree callers of the reference:
    utils/loss_utils.py:16-19,24-38,89-91     from diff_gaussian_rasterization._C import fusedssim, fusedssim_backward
                                              fusedssim(C1, C2, img1, img2) -> ssim_map ; fusedssim_backward(C1, C2, img1, img2, dL_dmap) -> dL_dimg1
The other entry points are what the upstream Python package binds ([RECALLED]: the rasterizer is an un-vendored submodule,
.gitmodules:4-7; argument orders as in INTEGRATION.md section 2) -- `rasterize_gaussians`, `rasterize_gaussians_backward`,
`mark_visible`, `adamUpdate` -- so that code written against the extension module itself (not only against
`GaussianRasterizer`) finds the same names with the same positional arguments and return tuples.

Everything here is a thin argument adapter over the C ABI (include/gsr.h); the tensors returned as "geomBuffer / binningBuffer /
imgBuffer" are the caller-owned scratch buffers of the resize callbacks, exactly the role they have upstream.  No CPU path."""
from __future__ import annotations

import ctypes as C
import sys

import torch

from . import _lib
from . import _Buffer, _f32c, _make_settings, _ptr, _sized, _trim_cache_if_pending, GaussianRasterizationSettings

_pkg = sys.modules[__package__]      # (`_require_cuda` / `_stream_ptr` are looked up on the package at call time, like its own code does)


def _require_cuda(t, name):
    _pkg._require_cuda(t, name)


def _stream_ptr(device):
    return _pkg._stream_ptr(device)


__all__ = ["rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible", "fusedssim", "fusedssim_backward", "adamUpdate"]

_C1, _C2 = 0.01 ** 2, 0.03 ** 2      # utils/loss_utils.py:21-22 -- the constants the SSIM kernels are compiled with (csrc/ssim.hip)


def _opt(t):
    """Upstream passes an EMPTY tensor for an argument that is not given (torch.Tensor([])); None is accepted too."""
    return None if t is None or t.numel() == 0 else t


def _settings(bg, scale_modifier, viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width, degree, campos, prefiltered,
              antialiasing, debug):
    return GaussianRasterizationSettings(int(image_height), int(image_width), float(tanfovx), float(tanfovy), bg, float(scale_modifier),
                                         viewmatrix, projmatrix, int(degree), campos, bool(prefiltered), bool(debug), bool(antialiasing))


def rasterize_gaussians(background, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tanfovx, tanfovy, image_height, image_width, sh, degree, campos, prefiltered, antialiasing, debug):
    """-> (num_rendered, color[3,H,W], radii[P], geomBuffer, binningBuffer, imgBuffer, invdepth[1,H,W])      [RECALLED order]"""
    lib = _lib.load()
    _require_cuda(means3D, "means3D")
    _trim_cache_if_pending()
    device = means3D.device
    P = int(means3D.shape[0])
    rs = _settings(background, scale_modifier, viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width, degree, campos,
                   prefiltered, antialiasing, debug)
    H, W = rs.image_height, rs.image_width
    m3, sh_c, col_c = _f32c(means3D), _f32c(_opt(sh)), _f32c(_opt(colors_precomp))
    op_c, sc_c, rot_c, cov_c = _f32c(opacities), _f32c(_opt(scales)), _f32c(_opt(rotations)), _f32c(_opt(cov3D_precomp))
    M = int(sh_c.shape[1]) if sh_c is not None and sh_c.dim() == 3 else 0
    keep: list = []
    with torch.cuda.device(device):
        s = _make_settings(rs, keep, None, no_backward=False)
        color = torch.empty(3, H, W, dtype=torch.float32, device=device)
        invdepth = torch.empty(1, H, W, dtype=torch.float32, device=device)
        radii = torch.empty(P, dtype=torch.int32, device=device)
        geom, binning, img = _Buffer(device, "geom"), _Buffer(device, "binning"), _Buffer(device, "image")
        nr = C.c_int32(0)
        _lib.check(lib.gsr_rasterize_forward(C.byref(s), P, M, _ptr(m3), _ptr(sh_c), _ptr(col_c), _ptr(op_c), _ptr(sc_c), _ptr(rot_c),
                                             _ptr(cov_c), geom.cb, None, binning.cb, None, img.cb, None, _ptr(color), _ptr(invdepth),
                                             _ptr(radii), C.byref(nr), _stream_ptr(device)), "gsr_rasterize_forward")
    return int(nr.value), color, radii, geom.t, binning.t, img.t, invdepth


def rasterize_gaussians_backward(background, means3D, radii, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tanfovx, tanfovy, dL_dout_color, dL_dout_invdepth, sh, degree, campos,
                                 geomBuffer, num_rendered, binningBuffer, imageBuffer, antialiasing, debug):
    """-> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)      [RECALLED order]
    Gradients of inputs that were not given come back as empty tensors (upstream: zero-sized / unused)."""
    lib = _lib.load()
    _require_cuda(means3D, "means3D")
    device = means3D.device
    P = int(means3D.shape[0])
    H, W = int(dL_dout_color.shape[-2]), int(dL_dout_color.shape[-1])
    rs = _settings(background, scale_modifier, viewmatrix, projmatrix, tanfovx, tanfovy, H, W, degree, campos, False, antialiasing, debug)
    m3, sh_c, col_c = _f32c(means3D), _f32c(_opt(sh)), _f32c(_opt(colors_precomp))
    op_c, sc_c, rot_c, cov_c = _f32c(opacities), _f32c(_opt(scales)), _f32c(_opt(rotations)), _f32c(_opt(cov3D_precomp))
    M = int(sh_c.shape[1]) if sh_c is not None and sh_c.dim() == 3 else 0
    f = dict(dtype=torch.float32, device=device)
    empty = torch.empty(0, **f)
    dm2, dop, dm3 = torch.empty(P, 3, **f), torch.empty(P, 1, **f), torch.empty(P, 3, **f)
    dcol = torch.empty(P, 3, **f) if col_c is not None else None
    dcov = torch.empty(P, 6, **f) if cov_c is not None else None
    dsh = torch.empty(P, M, 3, **f) if sh_c is not None else None
    dsc = torch.empty(P, 3, **f) if sc_c is not None else None
    drot = torch.empty(P, 4, **f) if rot_c is not None else None
    if P > 0:
        g_color = _f32c(dL_dout_color)
        g_depth = _f32c(_opt(dL_dout_invdepth))
        scratch = torch.empty(_sized("bwd", device, lib.gsr_backward_scratch_bytes(P, int(num_rendered))), dtype=torch.uint8, device=device)
        keep: list = []
        with torch.cuda.device(device):
            s = _make_settings(rs, keep, None)
            _lib.check(lib.gsr_rasterize_backward(C.byref(s), P, M, int(num_rendered), _ptr(m3), _ptr(sh_c), _ptr(col_c), _ptr(op_c), _ptr(sc_c),
                                                  _ptr(rot_c), _ptr(cov_c), _ptr(radii), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer),
                                                  _ptr(g_color), _ptr(g_depth), _ptr(dm2), _ptr(dcol), _ptr(dop), _ptr(dm3), _ptr(dcov), _ptr(dsh),
                                                  _ptr(dsc), _ptr(drot), _ptr(scratch), None, _stream_ptr(device)), "gsr_rasterize_backward")
    pick = lambda t: empty if t is None else t      # noqa: E731
    return dm2, pick(dcol), dop, dm3, pick(dcov), pick(dsh), pick(dsc), pick(drot)


def mark_visible(means3D, viewmatrix, projmatrix):
    """bool[P]: in front of the near plane (the reference's checkFrustum with prefiltered = False)."""
    lib = _lib.load()
    _require_cuda(means3D, "means3D")
    with torch.no_grad(), torch.cuda.device(means3D.device):
        pos, vm, pm = _f32c(means3D), _f32c(viewmatrix), _f32c(projmatrix)
        present = torch.empty(pos.shape[0], dtype=torch.uint8, device=pos.device)
        _lib.check(lib.gsr_mark_visible(int(pos.shape[0]), _ptr(pos), _ptr(vm), _ptr(pm), _ptr(present), _stream_ptr(pos.device)),
                   "gsr_mark_visible")
    return present.bool()


def _check_ssim_constants(C1, C2):
    if abs(float(C1) - _C1) > 1e-9 or abs(float(C2) - _C2) > 1e-9:
        raise _lib.GsrError(f"fusedssim: C1 = {C1}, C2 = {C2} -- the SSIM kernels are compiled for the reference's constants "
                            f"C1 = 0.01^2, C2 = 0.03^2 (utils/loss_utils.py:21-22)")


def _planes(img):
    if img.dim() == 3:
        return img.unsqueeze(0)
    if img.dim() != 4:
        raise _lib.GsrError("fusedssim: images are [C,H,W] or [B,C,H,W]")
    return img


def fusedssim(C1, C2, img1, img2):
    """utils/loss_utils.py:26: the SSIM MAP of img1 against img2 (11x11 Gaussian window, sigma 1.5, zero padding), same shape as img1."""
    lib = _lib.load()
    _check_ssim_constants(C1, C2)
    _require_cuda(img1, "img1")
    a, b = _f32c(_planes(img1)), _f32c(_planes(img2))
    Bn, Cn, H, W = a.shape
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _lib.check(lib.gsr_ssim_forward(Bn * Cn, H, W, _ptr(a), _ptr(b), _ptr(out), None, None, None, _stream_ptr(a.device)), "gsr_ssim_forward")
    return out.view(img1.shape)


def fusedssim_backward(C1, C2, img1, img2, dL_dmap):
    """utils/loss_utils.py:36: dL/dimg1 from dL/d(ssim_map).  The extension's signature carries no state from the forward, so the three
    partial-derivative planes the backward kernel consumes are produced again here (one more forward pass, as upstream's own kernel
    recomputes its window sums)."""
    lib = _lib.load()
    _check_ssim_constants(C1, C2)
    _require_cuda(img1, "img1")
    a, b, g = _f32c(_planes(img1)), _f32c(_planes(img2)), _f32c(_planes(dL_dmap))
    Bn, Cn, H, W = a.shape
    ssim_map, d1, d2, d3, out = (torch.empty_like(a) for _ in range(5))
    with torch.cuda.device(a.device):
        st = _stream_ptr(a.device)
        _lib.check(lib.gsr_ssim_forward(Bn * Cn, H, W, _ptr(a), _ptr(b), _ptr(ssim_map), _ptr(d1), _ptr(d2), _ptr(d3), st), "gsr_ssim_forward")
        _lib.check(lib.gsr_ssim_backward(Bn * Cn, H, W, _ptr(a), _ptr(b), _ptr(g), _ptr(d1), _ptr(d2), _ptr(d3), _ptr(out), st), "gsr_ssim_backward")
    return out.view(img1.shape)


def adamUpdate(param, param_grad, exp_avg, exp_avg_sq, visible, lr, b1, b2, eps, N, M):
    """The sparse Adam step behind SparseGaussianAdam.step (train.py:180-183; [RECALLED] signature): rows of `param` [N, M] whose
    `visible` flag is set are updated in place together with their two moments; no bias correction."""
    lib = _lib.load()
    _require_cuda(param, "param")
    vis = visible.reshape(-1)
    vis = vis.contiguous().view(torch.uint8) if vis.dtype == torch.bool else vis.to(torch.uint8).contiguous()
    for t, name in ((param, "param"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.GsrError(f"adamUpdate: {name} must be a contiguous fp32 tensor (updated in place)")
    g = _f32c(param_grad)
    with torch.cuda.device(param.device):
        _lib.check(lib.gsr_sparse_adam_step(_ptr(param), _ptr(g), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(vis), int(N), int(M), float(lr), float(b1),
                                            float(b2), float(eps), _stream_ptr(param.device)), "gsr_sparse_adam_step")
