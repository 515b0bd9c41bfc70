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


def wave_timeline(tr, kernel: int) -> dict:
    """Summary of a per-wave trace (`_lib.profile_trace()`; kernel 1 = forward blend, 2 = blend backward): span, drain, SIMD idle time,
    where a wave's lifetime goes, and `tail_loss` = span / (total steps / mid-launch step rate) - 1 -- what a perfectly balanced launch
    could gain at most.  Measurement only (tools/gpu_wave_trace.py, bench.py's blend_timeline)."""
    import numpy as np
    TICK_US = 0.01
    tr = tr[(tr[:, 1] != 0) & (((tr[:, 2] >> np.uint64(40)) & np.uint64(3)) == np.uint64(kernel))]
    if len(tr) == 0:
        return {"waves": 0}
    t0, t1 = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64)
    base = t0.min()
    t0, t1 = t0 - base, t1 - base
    w3 = tr[:, 3]
    steps = (w3 & np.uint64(0xFFFF)).astype(np.float64)
    walk, prep, store = [((w3 >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.float64) for sh in (16, 32, 48)]
    hw = tr[:, 2]
    simd = ((hw >> np.uint64(32)) & np.uint64(15)).astype(np.int64) * 4096 + ((hw >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64) * 4 + \
           ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)      # (XCC, SE|SH|CU, SIMD)
    span = int(t1.max())
    keys, inv = np.unique(simd, return_inverse=True)
    nsimd = len(keys)
    last_end = np.zeros(nsimd, np.int64)
    np.maximum.at(last_end, inv, t1)
    dur = np.maximum(t1 - t0, 1)
    # step rate over time: every wave's steps spread evenly over its lifetime, on a 1-tick grid
    rate = np.zeros(span + 2)
    np.add.at(rate, t0, steps / dur)
    np.add.at(rate, t1, -steps / dur)
    rate = np.cumsum(rate)[:span]
    res_w = np.zeros(span + 2)
    np.add.at(res_w, t0, 1.0)
    np.add.at(res_w, t1, -1.0)
    res_w = np.cumsum(res_w)[:span]
    mid = rate[span // 4: 3 * span // 4]
    rate_mid = float(np.median(mid))
    balanced = steps.sum() / max(rate_mid, 1e-9)
    per_simd_waves = np.bincount(inv, minlength=nsimd)
    return {"waves": int(len(tr)), "simds_seen": int(nsimd), "span_us": round(span * TICK_US, 2),
            "drain_frac": round(float(span - t0.max()) / span, 4),
            "simd_idle_frac": round(float((span - last_end).mean()) / span, 4),
            "mean_resident_waves_per_simd": round(float(dur.sum()) / (span * nsimd), 3),
            "resident_mid_launch_per_simd": round(float(np.median(res_w[span // 4: 3 * span // 4])) / nsimd, 3),
            "waves_per_simd_min_mean_max": [int(per_simd_waves.min()), round(float(per_simd_waves.mean()), 2), int(per_simd_waves.max())],
            "steps": int(steps.sum()), "heaviest_over_mean_steps": round(float(steps.max() / steps.mean()), 3),
            "wave_us_mean_max": [round(float(dur.mean()) * TICK_US, 2), round(float(dur.max()) * TICK_US, 2)],
            # where a wave's lifetime goes (means over the waves, us): walking survivors / getting batches ready (waiting for the gathered
            # records, box tests, staging) / storing the per-instance records / the rest (prologue loads, epilogue)
            "wave_phase_us_mean": {"walk": round(float(walk.mean()) * TICK_US, 2), "batch_prep": round(float(prep.mean()) * TICK_US, 2),
                                   "store": round(float(store.mean()) * TICK_US, 2),
                                   "rest": round(float((dur - walk - prep - store).mean()) * TICK_US, 2)},
            "walk_us_per_step": round(float(walk.sum() / max(steps.sum(), 1.0)) * TICK_US, 4),
            "rate_mid_steps_per_us": round(rate_mid / TICK_US, 1), "balanced_span_us": round(float(balanced) * TICK_US, 2),
            "tail_loss": round(float(span / balanced) - 1.0, 4),
            "rate_by_decile_of_span": [round(float(rate[int(span * k / 10): int(span * (k + 1) / 10)].mean()) / max(rate_mid, 1e-9), 3) for k in range(10)]}
