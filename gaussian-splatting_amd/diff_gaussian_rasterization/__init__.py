"""MI355X-native drop-in for the reference's `diff_gaussian_rasterization` package.

Exports exactly what gaussian_renderer/__init__.py:14 imports -- `GaussianRasterizationSettings` and
`GaussianRasterizer` -- with the same constructor fields (gaussian_renderer/__init__.py:36-50), the same
keyword call (gaussian_renderer/__init__.py:102-110), the same 3-tuple return `(color[3,H,W], radii[P],
invdepth[1,H,W])` and the same autograd contract (gradients to means3D, means2D (NDC-scaled dummy, read at
scene/gaussian_model.py:472), shs / colors_precomp, opacities, scales, rotations / cov3D_precomp).

It also exports `SparseGaussianAdam` (train.py:37-41, scene/gaussian_model.py:24-27), which makes the reference
use the "separate_sh" call form -- `rasterizer(dc=features_dc, shs=features_rest, ...)`,
gaussian_renderer/__init__.py:82-100 -- supported here without concatenating the two tensors.
All compute goes through the C ABI of libgsr_hip.so (include/gsr.h, hand-written gfx950 HIP kernels); there is no
CPU or eager-PyTorch fallback -- a missing library raises.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import GsrError, GsrRasterSettings, RESIZE_FN  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "SparseGaussianAdam", "rasterize_gaussians", "GsrError"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool = False


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _bucket(nbytes: int) -> int:
    """Round a scratch size up to one of 8 sizes per octave (<= 12.5 % slack)."""
    nbytes = int(nbytes)
    if nbytes <= (1 << 20):
        return nbytes
    step = 1 << (nbytes.bit_length() - 4)
    return (nbytes + step - 1) // step * step


_last_size: dict = {}
_last_R = 0          # num_rendered of the most recent forward (diagnostics: tools/train_run.py)
_max_R = 0


def _sized(kind: str, device, nbytes: int) -> int:
    """Allocation size for an R-sized scratch buffer (binning state, backward scratch).  R changes by a fraction of a
    percent from frame to frame as the Gaussians move; handing the caching allocator a different size every iteration
    makes it hipMalloc a fresh block now and then, which costs tens of ms on ROCm (seen as one-off 75 ms stalls in
    bench.py's train legs whenever the size crossed a rounding boundary).  So sizes are sticky: the last size of this
    kind is reused while it fits (and is not more than twice too big); when it does not, the new size jumps 12.5 %
    ahead of the need and is rounded to 8 steps per octave -- a training run re-allocates once per ~12 % growth."""
    nbytes = int(nbytes)
    key = (kind, getattr(device, "index", None))
    last = _last_size.get(key, 0)
    if nbytes <= last <= 2 * max(nbytes, 1 << 20):
        return last
    new = _bucket(nbytes + nbytes // 8)
    _last_size[key] = new
    if last > (64 << 20):
        # the size class of a large scratch buffer changed (densification grew the set): blocks of the old class would stay in
        # torch's caching allocator for ever -- a 30 000-iteration run that grew 100 k -> 1.4 M Gaussians ended with 12 GB
        # reserved against 1 GB allocated.  They are handed back at the start of the next forward.
        global _trim_pending
        _trim_pending = True
    return new


_trim_pending = False


def _trim_cache_if_pending():
    global _trim_pending
    if _trim_pending:
        _trim_pending = False
        torch.cuda.empty_cache()


class _Buffer:
    """A growable uint8 device tensor handed to the library through a resize callback (the reference's
    resizeFunctional lambda).  The callback closes over a one-element list, NOT over the _Buffer: a closure that referenced
    `self` would form a reference cycle (self -> ctypes callback -> closure -> self) that only the cyclic garbage collector
    breaks -- and it does not see device memory pressure: a 30 000-iteration training run of round 1's version accumulated
    > 100 GB of dead scratch buffers between collections."""

    def __init__(self, device, kind: str = "state"):
        holder = [torch.empty(0, dtype=torch.uint8, device=device)]
        self._holder = holder

        def _resize(_user, nbytes):
            if holder[0].numel() < nbytes:
                holder[0] = torch.empty(_sized(kind, device, nbytes), dtype=torch.uint8, device=device)
            return holder[0].data_ptr()

        self.cb = RESIZE_FN(_resize)

    @property
    def t(self) -> torch.Tensor:
        return self._holder[0]


def _make_settings(rs: GaussianRasterizationSettings, keep: list, tile_rows, no_backward: bool = False) -> GsrRasterSettings:
    dev_t = [_f32c(rs.bg), _f32c(rs.viewmatrix), _f32c(rs.projmatrix), _f32c(rs.campos)]
    keep.extend(dev_t)
    s = GsrRasterSettings()
    s.image_height = int(rs.image_height)
    s.image_width = int(rs.image_width)
    s.tanfovx = float(rs.tanfovx)
    s.tanfovy = float(rs.tanfovy)
    s.bg = dev_t[0].data_ptr()
    s.scale_modifier = float(rs.scale_modifier)
    s.viewmatrix = dev_t[1].data_ptr()
    s.projmatrix = dev_t[2].data_ptr()
    s.sh_degree = int(rs.sh_degree)
    s.campos = dev_t[3].data_ptr()
    s.prefiltered = int(bool(rs.prefiltered))
    s.debug = int(bool(rs.debug))
    s.antialiasing = int(bool(rs.antialiasing))
    s.no_backward = int(bool(no_backward))
    if tile_rows is None:
        s.tile_y0, s.tile_y1 = 0, 0
    elif int(tile_rows[1]) <= int(tile_rows[0]):
        # empty band (a rank that owns no tile rows): (0, 0) would mean "all rows" in the C ABI, so hand over a
        # band past the last row, which the library clamps to an empty one
        s.tile_y0 = s.tile_y1 = 1 << 20
    else:
        s.tile_y0, s.tile_y1 = int(tile_rows[0]), int(tile_rows[1])
    return s


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise GsrError(f"{name} must live on a HIP device ('cuda'); the MI355X rasterizer has no CPU path")


# ---- opt-in: the Adam step of the two SH tensors inside the backward (gsr_backward_preprocess_sh_adam) --------------------------
_SH_ADAM = {}      # id(rest parameter) -> _ShAdamFusion


class _ShAdamFusion:
    """Registered by `fuse_sh_adam_into_backward`.  While it is active, a rasterizer call in the reference's separate_sh form whose
    `dc=` / `shs=` arguments ARE the two registered leaf parameters does not return their gradients from backward: the per-Gaussian
    backward kernel applies the optimizer's update to them in place (and to the optimizer's moments), bit for bit what
    `optimizer.step()` would have done with the gradient, which therefore never travels to HBM and back.  `optimizer.step()`
    afterwards skips the two tensors (their .grad is None) and steps the rest as usual."""

    def __init__(self, optimizer, dc, rest):
        import weakref
        self.optimizer, self.dc, self.rest = optimizer, weakref.ref(dc), weakref.ref(rest)
        self.sparse = isinstance(optimizer, SparseGaussianAdam)
        self.key = id(rest)

    def matches(self, dc, rest):
        return self.dc() is dc and self.rest() is rest

    def remove(self):
        _SH_ADAM.pop(self.key, None)

    def _group(self, param):
        for g in self.optimizer.param_groups:
            if any(q is param for q in g["params"]):
                return g
        raise GsrError("fused SH Adam: the parameter is not in the optimizer any more (re-register after densification)")

    def arm(self):
        """Moments (created on first use), hyper-parameters and -- dense Adam -- the incremented step counts as a _lib.ShAdam."""
        dc, rest = self.dc(), self.rest()      # (the Parameter objects: the optimizer's state is keyed by them)
        if dc is None or rest is None:
            raise GsrError("fused SH Adam: a registered parameter no longer exists")
        st = []
        for q in (dc, rest):
            s_ = self.optimizer.state[q]
            if len(s_) == 0:
                s_["step"] = torch.tensor(0.0, dtype=torch.float32) if self.sparse else 0
                s_["exp_avg"] = torch.zeros_like(q, memory_format=torch.contiguous_format)
                s_["exp_avg_sq"] = torch.zeros_like(q, memory_format=torch.contiguous_format)
            if not self.sparse:
                s_["step"] = int(s_["step"]) + 1
            st.append(s_)
        g_dc, g_rest = self._group(dc), self._group(rest)
        b1, b2 = (0.9, 0.999) if self.sparse else g_rest["betas"]
        if not self.sparse and tuple(g_dc["betas"]) != tuple(g_rest["betas"]) or g_dc["eps"] != g_rest["eps"]:
            raise GsrError("fused SH Adam: the two SH parameter groups must share betas and eps")
        return _lib.ShAdam(st[0]["exp_avg"].data_ptr(), st[0]["exp_avg_sq"].data_ptr(), st[1]["exp_avg"].data_ptr(),
                           st[1]["exp_avg_sq"].data_ptr(), float(g_dc["lr"]), float(g_rest["lr"]), float(b1), float(b2),
                           float(g_rest["eps"]), 0 if self.sparse else int(st[0]["step"]), 0 if self.sparse else int(st[1]["step"]),
                           1 if self.sparse else 0, 0)


def fuse_sh_adam_into_backward(optimizer, dc_param, rest_param):
    """OPT-IN (no reference counterpart): from now on `loss.backward()` applies `optimizer`'s step to `dc_param` ([P,1,3]) and
    `rest_param` ([P,15,3]) itself when they reach the rasterizer as `dc=` / `shs=` (gaussian_renderer/__init__.py:82-100,
    separate_sh).  `optimizer` is a `SparseGaussianAdam` (rows with radii > 0, as `step(radii > 0, N)` would) or a
    `gsr_optim.FusedAdam` / `torch.optim.Adam`-configured optimizer (every row).  One backward per step; call `.remove()` on the
    returned handle -- or register the new parameters -- when densification replaces the tensors."""
    ent = _ShAdamFusion(optimizer, dc_param, rest_param)
    _SH_ADAM[ent.key] = ent
    return ent


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, tile_rows, grad_sync, dc):
        lib = _lib.load()
        _require_cuda(means3D, "means3D")
        _trim_cache_if_pending()
        device = means3D.device
        P = int(means3D.shape[0])
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        means3D_c, sh_c, col_c = _f32c(means3D), _f32c(sh), _f32c(colors_precomp)
        op_c, sc_c, rot_c, cov_c = _f32c(opacities), _f32c(scales), _f32c(rotations), _f32c(cov3Ds_precomp)
        dc_c = _f32c(dc)
        # split form: dc[P,1,3] + sh[P,M-1,3].  The kernels take it natively for degree-3 storage (M == 16).  A model
        # with max_sh_degree 0 has an empty `sh`: the DC tensor then IS the fused [P,1,3] record ("as_sh").  Other small
        # records (M = 4, 9) or unaligned views are concatenated like the reference's own get_features
        # (scene/gaussian_model.py:121-125) and the gradient is split again in backward ("cat").
        ctx.dc_mode = "none"
        if dc_c is not None:
            if sh_c is None or sh_c.numel() == 0 or sh_c.shape[1] == 0:
                ctx.dc_mode = "as_sh"
                sh_c, dc_c = dc_c.view(P, 1, 3), None
            elif sh_c.shape[1] + 1 != 16 or dc_c.data_ptr() % 16 or sh_c.data_ptr() % 16:
                ctx.dc_mode = "cat"
                sh_c, dc_c = torch.cat([dc_c.view(P, 1, 3), sh_c], dim=1), None
            else:
                ctx.dc_mode = "split"
        M = int(sh_c.shape[1]) if sh_c is not None and sh_c.dim() == 3 else 0
        if dc_c is not None:
            M += 1
        keep: list = []
        with torch.cuda.device(device):
            # inside autograd.Function.forward grad mode is off; needs_input_grad tells whether a backward can follow
            no_backward = not (any(ctx.needs_input_grad[:8]) or ctx.needs_input_grad[11])
            s = _make_settings(raster_settings, keep, tile_rows, no_backward)
            if dc_c is not None:
                s.sh_dc = dc_c.data_ptr()
            color = torch.empty(3, H, W, dtype=torch.float32, device=device)
            invdepth = torch.empty(1, H, W, dtype=torch.float32, device=device)
            if tile_rows is not None:   # rows outside the band are not written by the kernels
                color.zero_()
                invdepth.zero_()
            radii = torch.empty(P, dtype=torch.int32, device=device)
            geom, binning, img = _Buffer(device, "geom"), _Buffer(device, "binning"), _Buffer(device, "image")
            nr = C.c_int32(0)
            args = (C.byref(s), P, M, _ptr(means3D_c), _ptr(sh_c), _ptr(col_c), _ptr(op_c), _ptr(sc_c), _ptr(rot_c),
                    _ptr(cov_c), geom.cb, None, binning.cb, None, img.cb, None, _ptr(color), _ptr(invdepth),
                    _ptr(radii), C.byref(nr), _stream_ptr(device))
            if raster_settings.debug:
                cpu_args = _cpu_copy((means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                      raster_settings))
                try:
                    _lib.check(lib.gsr_rasterize_forward(*args), "gsr_rasterize_forward")
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                    raise ex
            else:
                _lib.check(lib.gsr_rasterize_forward(*args), "gsr_rasterize_forward")
        ctx.raster_settings = raster_settings
        ctx.tile_rows = tile_rows
        ctx.grad_sync = grad_sync
        ctx.num_rendered = int(nr.value)
        global _last_R, _max_R
        _last_R = ctx.num_rendered
        _max_R = max(_max_R, _last_R)
        ctx.M = M
        ctx.op_shape = tuple(opacities.shape)
        ctx.has_means2D = means2D is not None
        ctx.flags = (sh_c is not None, colors_precomp is not None, scales is not None, rotations is not None,
                     cov3Ds_precomp is not None)
        ctx.has_dc = dc_c is not None
        ctx.sh_adam = None
        if ctx.dc_mode == "split" and grad_sync is None and colors_precomp is None and _SH_ADAM:
            ent = _SH_ADAM.get(id(sh))
            if ent is not None and ent.matches(dc, sh) and sh_c is sh and dc_c is dc and sh.requires_grad and dc.requires_grad:
                ctx.sh_adam = ent
        ctx.sh_given = sh is not None
        ctx.dc_shape = tuple(dc.shape) if dc is not None else None
        ctx.save_for_backward(means3D_c, sh_c if sh_c is not None else means3D_c.new_empty(0),
                              col_c if col_c is not None else means3D_c.new_empty(0), op_c,
                              sc_c if sc_c is not None else means3D_c.new_empty(0),
                              rot_c if rot_c is not None else means3D_c.new_empty(0),
                              cov_c if cov_c is not None else means3D_c.new_empty(0),
                              radii, geom.t, binning.t, img.t,
                              dc_c if dc_c is not None else means3D_c.new_empty(0))
        ctx.mark_non_differentiable(radii)
        # an output nobody differentiates through (the inverse-depth image unless depth supervision is on, train.py:130-137)
        # arrives in backward as None instead of a materialised zero image: the blend backward then runs its build without
        # the 1/depth terms ("skip the invdepth terms when it is all-zero / None")
        ctx.set_materialize_grads(False)
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_out_depth):
        lib = _lib.load()
        (means3D, sh, col, op, sc, rot, cov, radii, geom, binning, img, dc) = ctx.saved_tensors
        if grad_out_color is None:          # only the inverse-depth image was used
            rs_ = ctx.raster_settings
            grad_out_color = torch.zeros(3, int(rs_.image_height), int(rs_.image_width), dtype=torch.float32, device=means3D.device)
        has_sh, has_col, has_sc, has_rot, has_cov = ctx.flags
        rs = ctx.raster_settings
        device = means3D.device
        P = int(means3D.shape[0])
        M = ctx.M
        f = dict(dtype=torch.float32, device=device)
        dL_dmeans2D = torch.empty(P, 3, **f)
        dL_dcolors = torch.empty(P, 3, **f) if has_col else None      # intermediates of the kernel unless they are inputs' grads
        dL_dopacity = torch.empty(P, 1, **f)
        dL_dmeans3D = torch.empty(P, 3, **f)
        dL_dcov3D = torch.empty(P, 6, **f) if has_cov else None
        has_dc = ctx.has_dc
        fused_adam = ctx.sh_adam if (ctx.sh_adam is not None and _SH_ADAM.get(ctx.sh_adam.key) is ctx.sh_adam) else None
        dL_dsh = torch.empty(P, M - 1 if has_dc else M, 3, **f) if (has_sh and fused_adam is None) else None
        dL_ddc = torch.empty(P, 1, 3, **f) if (has_dc and fused_adam is None) else None
        dL_dscales = torch.empty(P, 3, **f) if has_sc else None
        dL_drot = torch.empty(P, 4, **f) if has_rot else None
        if P > 0:
            g_color = _f32c(grad_out_color)
            g_depth = _f32c(grad_out_depth) if grad_out_depth is not None else None
            scratch = torch.empty(_sized("bwd", device, lib.gsr_backward_scratch_bytes(P, ctx.num_rendered)), dtype=torch.uint8,
                                  device=device)
            keep: list = []
            with torch.cuda.device(device):
                s = _make_settings(rs, keep, ctx.tile_rows)
                if has_dc:
                    s.sh_dc = dc.data_ptr()
                    s.dL_dsh_dc = dL_ddc.data_ptr() if dL_ddc is not None else None
                st = _stream_ptr(device)
                inputs = (_ptr(means3D), _ptr(sh) if has_sh else None, _ptr(col) if has_col else None, _ptr(op),
                          _ptr(sc) if has_sc else None, _ptr(rot) if has_rot else None, _ptr(cov) if has_cov else None,
                          _ptr(radii))
                outs = (_ptr(dL_dmeans2D), _ptr(dL_dcolors), _ptr(dL_dopacity), _ptr(dL_dmeans3D), _ptr(dL_dcov3D),
                        _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drot))

                def _run():
                    if fused_adam is not None:
                        # blend backward, then the per-Gaussian backward that steps the two SH tensors in place
                        rec_ptr = C.c_void_p(0)
                        _lib.check(lib.gsr_backward_blend(C.byref(s), P, ctx.num_rendered, _ptr(geom), _ptr(binning), _ptr(img),
                                                          _ptr(g_color), _ptr(g_depth), _ptr(scratch), C.byref(rec_ptr), st),
                                   "gsr_backward_blend")
                        adam = fused_adam.arm()
                        _lib.check(lib.gsr_backward_preprocess_sh_adam(
                            C.byref(s), P, M, _ptr(means3D), _ptr(sh), _ptr(op), _ptr(sc) if has_sc else None,
                            _ptr(rot) if has_rot else None, _ptr(cov) if has_cov else None, _ptr(radii), _ptr(geom), rec_ptr,
                            _ptr(dL_dmeans2D), _ptr(dL_dopacity), _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dscales),
                            _ptr(dL_drot), C.byref(adam), st), "gsr_backward_preprocess_sh_adam")
                        return
                    if ctx.grad_sync is None:
                        _lib.check(lib.gsr_rasterize_backward(C.byref(s), P, M, ctx.num_rendered, *inputs, _ptr(geom),
                                                              _ptr(binning), _ptr(img), _ptr(g_color), _ptr(g_depth), *outs,
                                                              _ptr(scratch), None, st), "gsr_rasterize_backward")
                        return
                    # screen-sharded training: blend backward on the own band -> sum the 48-byte per-Gaussian records
                    # across ranks -> per-Gaussian backward (parallel.py)
                    rec_ptr = C.c_void_p(0)
                    _lib.check(lib.gsr_backward_blend(C.byref(s), P, ctx.num_rendered, _ptr(geom), _ptr(binning), _ptr(img),
                                                      _ptr(g_color), _ptr(g_depth), _ptr(scratch), C.byref(rec_ptr), st),
                               "gsr_backward_blend")
                    off = int(rec_ptr.value) - scratch.data_ptr()
                    records = scratch[off:off + P * 48].view(torch.float32).view(P, 12)
                    ctx.grad_sync(records)
                    _lib.check(lib.gsr_backward_preprocess(C.byref(s), P, M, *inputs, _ptr(geom), _ptr(records), *outs, st),
                               "gsr_backward_preprocess")

                if rs.debug:
                    cpu_args = _cpu_copy((means3D, radii, col, sc, rot, cov, sh, grad_out_color, rs))
                    try:
                        _run()
                    except Exception as ex:
                        torch.save(cpu_args, "snapshot_bw.dump")
                        print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                        raise ex
                else:
                    _run()
        dL_dopacity = dL_dopacity.view(ctx.op_shape)
        if ctx.dc_mode == "as_sh":      # the DC tensor travelled as the fused [P,1,3] form
            dL_ddc = dL_dsh.view(ctx.dc_shape)
            dL_dsh = dL_dsh.new_zeros(P, 0, 3) if ctx.sh_given else None
        elif ctx.dc_mode == "cat":
            dL_ddc = dL_dsh[:, :1].reshape(ctx.dc_shape)
            dL_dsh = dL_dsh[:, 1:]
        elif has_dc and dL_ddc is not None:
            dL_ddc = dL_ddc.view(ctx.dc_shape)
        return (dL_dmeans3D, dL_dmeans2D if ctx.has_means2D else None, dL_dsh, dL_dcolors if has_col else None, dL_dopacity, dL_dscales, dL_drot,
                dL_dcov3D if has_cov else None, None, None, None, dL_ddc)


def _cpu_copy(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, tile_rows: Optional[Tuple[int, int]] = None, grad_sync=None, dc=None):
    """Functional form.  `grad_sync(records[P,12])`, if given, is called between the blend backward and the
    per-Gaussian backward (multi-GPU: all-reduce of the 48-byte gradient records, parallel.py).
    `tile_rows=(y0, y1)` (extension) restricts binning + blending to that
    band of 16-pixel tile rows; pixels outside the band come back as zeros.  `dc` (the reference's separate_sh
    form): SH coefficient 0 as [P,1,3]; `sh` then holds coefficients 1.. as [P,M-1,3]."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, tile_rows, grad_sync, dc)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """bool[P]: in front of the near plane (the reference's mark_visible / checkFrustum)."""
        lib = _lib.load()
        _require_cuda(positions, "positions")
        rs = self.raster_settings
        with torch.no_grad(), torch.cuda.device(positions.device):
            pos = _f32c(positions)
            vm, pm = _f32c(rs.viewmatrix), _f32c(rs.projmatrix)
            present = torch.empty(pos.shape[0], dtype=torch.uint8, device=pos.device)
            _lib.check(lib.gsr_mark_visible(int(pos.shape[0]), _ptr(pos), _ptr(vm), _ptr(pm), _ptr(present),
                                            _stream_ptr(pos.device)), "gsr_mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, dc=None, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if dc is not None and shs is None:
            raise Exception('dc (SH coefficient 0) was given without shs (the remaining coefficients)!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings, getattr(self, "tile_rows", None), None, dc)


class SparseGaussianAdam(torch.optim.Adam):
    """`SparseGaussianAdam(params, lr, eps).step(visibility, N)` (scene/gaussian_model.py:194-196, train.py:180-183):
    Adam that updates only the rows of Gaussians that were visible in this iteration; parameter and both moments of the
    others stay untouched.  One fused HIP launch for all parameter groups (gsr_sparse_adam_step_multi; round 5 -- one launch per group before).  State layout
    (`step`, `exp_avg`, `exp_avg_sq`) is torch.optim.Adam's, so the optimizer-state surgery of
    scene/gaussian_model.py:316-405 (prune / cat / replace) works unchanged.
    [RECALLED -- the accelerated rasterizer's source is not vendored] betas are fixed at (0.9, 0.999) and there is no
    bias correction; parity for this class is unpinned."""

    def __init__(self, params, lr, eps):
        super().__init__(params=params, lr=lr, eps=eps)

    @torch.no_grad()
    def step(self, visibility, N):
        lib = _lib.load()
        N = int(N)
        vis = visibility.reshape(-1)
        if vis.dtype == torch.bool:
            vis = vis.contiguous().view(torch.uint8)      # a bool tensor IS one byte per element: no conversion kernel
        elif vis.dtype != torch.uint8:
            vis = vis.to(torch.uint8)
        vis = vis.contiguous()
        by_device: dict = {}      # device -> [(param, grad, state, M, lr, eps)]: ONE launch per device (gsr_sparse_adam_step_multi)
        for group in self.param_groups:
            lr, eps = group["lr"], group["eps"]
            assert len(group["params"]) == 1, "more than one tensor in group"
            param = group["params"][0]
            if param.grad is None:
                continue
            _require_cuda(param, "parameter")
            if param.dtype != torch.float32 or not param.is_contiguous():
                raise GsrError("SparseGaussianAdam needs contiguous fp32 parameters")
            state = self.state[param]
            if len(state) == 0:
                state["step"] = torch.tensor(0.0, dtype=torch.float32)
                state["exp_avg"] = torch.zeros_like(param, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(param, memory_format=torch.preserve_format)
            M = param.numel() // N if N > 0 else 0
            if N > 0 and (param.numel() != N * M or vis.numel() != N):
                raise GsrError(f"parameter of {param.numel()} elements / visibility of {vis.numel()} do not match N = {N}")
            g = param.grad if param.grad.is_contiguous() else param.grad.contiguous()
            by_device.setdefault(param.device, []).append((param, g, state, M, float(lr), float(eps)))
        for dev, items in by_device.items():
            arr = (_lib.SparseAdamTensor * len(items))()
            for k, (param, g, state, M, lr, eps) in enumerate(items):
                arr[k] = _lib.SparseAdamTensor(param.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(), M, lr, eps)
            with torch.cuda.device(dev):
                _lib.check(lib.gsr_sparse_adam_step_multi(arr, len(items), _ptr(vis), N, 0.9, 0.999, _stream_ptr(dev)), "gsr_sparse_adam_step_multi")
