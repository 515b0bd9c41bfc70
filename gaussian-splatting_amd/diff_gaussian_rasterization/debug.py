"""Introspection of the forward pass for tests and bench.py: runs gsr_rasterize_forward through the C ABI and
returns the intermediate state (splat records, tiles_touched, depth order, sorted point list, tile ranges,
final_T, n_contrib) as torch tensors sliced out of the three scratch buffers.  No reference counterpart."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from . import _Buffer, _f32c, _make_settings, _ptr, _stream_ptr, GaussianRasterizationSettings


def _view(buf: torch.Tensor, ptr: int, nbytes: int, dtype) -> torch.Tensor:
    off = ptr - buf.data_ptr()
    assert 0 <= off and off + nbytes <= buf.numel(), "view outside its buffer"
    return buf[off:off + nbytes].view(dtype)


def forward_with_views(rs: GaussianRasterizationSettings, means3D, opacities, shs=None, colors_precomp=None,
                       scales=None, rotations=None, cov3D_precomp=None, tile_rows=None, want_invdepth=True,
                       no_backward=False):
    """no_backward=True runs the INFERENCE instantiation (what a torch.no_grad() render and bench.py's forward metric
    use): final_T / n_contrib / first-emission indices are not written, so those views are omitted."""
    lib = _lib.load()
    device = means3D.device
    P = int(means3D.shape[0])
    H, W = int(rs.image_height), int(rs.image_width)
    t = [_f32c(x) for x in (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)]
    M = int(t[1].shape[1]) if t[1] is not None else 0
    keep: list = []
    with torch.cuda.device(device):
        s = _make_settings(rs, keep, tile_rows, no_backward)
        color = torch.zeros(3, H, W, dtype=torch.float32, device=device)
        invdepth = torch.zeros(1, H, W, dtype=torch.float32, device=device) if want_invdepth else None
        radii = torch.empty(P, dtype=torch.int32, device=device)
        geom, binning, img = _Buffer(device), _Buffer(device), _Buffer(device)
        nr = C.c_int32(0)
        _lib.check(lib.gsr_rasterize_forward(C.byref(s), P, M, _ptr(t[0]), _ptr(t[1]), _ptr(t[2]), _ptr(t[3]), _ptr(t[4]),
                                             _ptr(t[5]), _ptr(t[6]), geom.cb, None, binning.cb, None, img.cb, None,
                                             _ptr(color), _ptr(invdepth), _ptr(radii), C.byref(nr), _stream_ptr(device)),
                   "gsr_rasterize_forward")
        R = int(nr.value)
        out = {"color": color, "invdepth": invdepth, "radii": radii, "R": R,
               "buffers": (geom.t, binning.t, img.t)}
        if P == 0:
            return out
        v = _lib.GsrForwardViews()
        _lib.check(lib.gsr_forward_views(P, R, W, H, _ptr(geom.t), _ptr(binning.t), _ptr(img.t), C.byref(v)),
                   "gsr_forward_views")
        gx, gy = (W + 15) // 16, (H + 15) // 16
        out["splats"] = _view(geom.t, v.splats, P * 64, torch.float32).view(P, 16)
        out["tiles_touched"] = _view(geom.t, v.tiles_touched, P * 4, torch.int32)
        out["depth_order"] = _view(geom.t, v.depth_order, P * 4, torch.int32)
        if v.tile_scan:      # (ABI 4; absent when an older library is bound for an A/B run)
            out["offsets"] = _view(geom.t, v.tile_scan, P * 4, torch.int32)
        out["point_list"] = (_view(binning.t, v.point_list, R * 4, torch.int32) if R > 0
                             else torch.empty(0, dtype=torch.int32, device=device))
        out["ranges"] = _view(img.t, v.ranges, gx * gy * 8, torch.int32).view(gx * gy, 2)
        if not no_backward:
            out["final_T"] = _view(img.t, v.final_T, H * W * 4, torch.float32).view(H, W)
            out["n_contrib"] = _view(img.t, v.n_contrib, H * W * 4, torch.int32).view(H, W)
    return out
