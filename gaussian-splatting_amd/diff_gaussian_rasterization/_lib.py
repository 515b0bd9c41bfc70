"""ctypes binding of libgsr_hip.so (include/gsr.h).

This file is the "reference-side binding" INTEGRATION.md describes: it replaces the pybind11 module
`diff_gaussian_rasterization._C` of the reference's un-vendored submodule (call sites
gaussian_renderer/__init__.py:14,91-110).  There is NO fallback: if the shared library is missing or was
not built for this GPU, loading raises -- the product path never routes through a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libgsr_hip.so")

GSR_OK = 0
ABI_VERSION = 4
STAGES = ["preprocess", "depth_sort", "scan", "emit", "tile_sort", "ranges", "render", "render_bwd", "preprocess_bwd",
          "gather_bwd", "color", "r_wait"]


class GsrRasterSettings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("bg", C.c_void_p), ("scale_modifier", C.c_float),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("sh_degree", C.c_int32), ("campos", C.c_void_p),
        ("prefiltered", C.c_int32), ("debug", C.c_int32), ("antialiasing", C.c_int32),
        ("tile_y0", C.c_int32), ("tile_y1", C.c_int32), ("no_backward", C.c_int32),
        ("sh_dc", C.c_void_p), ("dL_dsh_dc", C.c_void_p),
    ]


class GsrForwardViews(C.Structure):
    _fields_ = [("splats", C.c_void_p), ("tiles_touched", C.c_void_p), ("depth_order", C.c_void_p),
                ("point_list", C.c_void_p), ("ranges", C.c_void_p), ("final_T", C.c_void_p),
                ("n_contrib", C.c_void_p), ("tile_scan", C.c_void_p)]


RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

EXPORTS = [
    "gsr_abi_version", "gsr_last_error", "gsr_geometry_bytes", "gsr_binning_bytes", "gsr_image_bytes",
    "gsr_backward_scratch_bytes",
    "gsr_rasterize_forward", "gsr_rasterize_backward", "gsr_backward_blend", "gsr_backward_preprocess",
    "gsr_preprocess_forward", "gsr_rasterize_from_splats",
    "gsr_route_scratch_bytes", "gsr_route_count", "gsr_route_pack", "gsr_rasterize_from_packed", "gsr_route_return",
    "gsr_route_pack_fixed", "gsr_rasterize_from_segments",
    "gsr_mark_visible", "gsr_forward_views", "gsr_adam_step", "gsr_adam_step_multi", "gsr_backward_preprocess_sh_adam", "gsr_sparse_adam_step", "gsr_sparse_adam_step_multi", "gsr_ssim_forward", "gsr_ssim_backward",
    "gsr_knn_scratch_bytes", "gsr_knn_mean_dist2", "gsr_ssim_partial_count", "gsr_ssim_mean_forward", "gsr_ssim_mean_backward",
    "gsr_train_loss_forward", "gsr_train_loss_backward", "gsr_density_stats",
    "gsr_profile_enable", "gsr_profile_reset", "gsr_profile_read", "gsr_profile_counters", "gsr_profile_trace", "gsr_set_option",
]

_lib: Optional[C.CDLL] = None


class GsrError(RuntimeError):
    pass


def lib_path() -> str:
    return os.environ.get("GSR_LIB", _LIB_PATH)



class ShAdam(C.Structure):
    """GsrShAdam of include/gsr.h (gsr_backward_preprocess_sh_adam)."""
    _fields_ = [("dc_exp_avg", C.c_void_p), ("dc_exp_avg_sq", C.c_void_p), ("rest_exp_avg", C.c_void_p), ("rest_exp_avg_sq", C.c_void_p),
                ("lr_dc", C.c_double), ("lr_rest", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("step_dc", C.c_int32), ("step_rest", C.c_int32), ("sparse", C.c_int32), ("reserved", C.c_int32)]


class AdamTensor(C.Structure):
    """GsrAdamTensor of include/gsr.h (gsr_adam_step_multi)."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("step", C.c_int32),
                ("reserved", C.c_int32)]

class SparseAdamTensor(C.Structure):
    """GsrSparseAdamTensor of include/gsr.h (gsr_sparse_adam_step_multi)."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("M", C.c_int64),
                ("lr", C.c_double), ("eps", C.c_double)]


def load() -> C.CDLL:
    """Load libgsr_hip.so.  torch must own the HIP runtime: torch bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7) and our library NEEDs the same SONAME, so importing torch first makes both resolve to
    one runtime -- stream handles and device pointers are then interchangeable."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise GsrError(f"{path} not found: build it with `python gaussian-splatting_amd/build.py` "
                       f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first)
    except Exception:  # pragma: no cover - torch-less use (pure C hosts) links /opt/rocm's runtime
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    lib.gsr_abi_version.restype = C.c_int
    # the version is checked BEFORE any other symbol is bound: an older library (tools/build_prev_lib.sh, GSR_LIB=...) fails
    # with this message instead of an AttributeError on a symbol it does not have yet
    if lib.gsr_abi_version() != ABI_VERSION and os.environ.get("GSR_ALLOW_ABI_MISMATCH") != "1":
        raise GsrError(f"{path}: ABI version {lib.gsr_abi_version()} != {ABI_VERSION} (set GSR_ALLOW_ABI_MISMATCH=1 to bind the "
                       f"symbols both versions share, for A/B runs of an older revision)")
    missing = [n for n in EXPORTS if not hasattr(lib, n)]
    if missing and os.environ.get("GSR_ALLOW_ABI_MISMATCH") != "1":
        raise GsrError(f"{path} does not export {missing}")
    lib.gsr_last_error.restype = C.c_char_p
    lib.gsr_geometry_bytes.restype = C.c_size_t
    lib.gsr_geometry_bytes.argtypes = [C.c_int]
    lib.gsr_binning_bytes.restype = C.c_size_t
    lib.gsr_binning_bytes.argtypes = [C.c_int64, C.c_int]
    lib.gsr_image_bytes.restype = C.c_size_t
    lib.gsr_image_bytes.argtypes = [C.c_int, C.c_int]
    lib.gsr_backward_scratch_bytes.restype = C.c_size_t
    lib.gsr_backward_scratch_bytes.argtypes = [C.c_int, C.c_int64]
    vp = C.c_void_p
    lib.gsr_rasterize_forward.restype = C.c_int
    lib.gsr_rasterize_forward.argtypes = [C.POINTER(GsrRasterSettings), C.c_int, C.c_int,
                                          vp, vp, vp, vp, vp, vp, vp,
                                          RESIZE_FN, vp, RESIZE_FN, vp, RESIZE_FN, vp,
                                          vp, vp, vp, C.POINTER(C.c_int32), vp]
    lib.gsr_rasterize_backward.restype = C.c_int
    lib.gsr_rasterize_backward.argtypes = [C.POINTER(GsrRasterSettings), C.c_int, C.c_int, C.c_int32,
                                           vp, vp, vp, vp, vp, vp, vp, vp,
                                           vp, vp, vp, vp, vp,
                                           vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_void_p), vp]
    lib.gsr_backward_blend.restype = C.c_int
    lib.gsr_backward_blend.argtypes = [C.POINTER(GsrRasterSettings), C.c_int, C.c_int32, vp, vp, vp, vp, vp, vp,
                                       C.POINTER(C.c_void_p), vp]
    lib.gsr_backward_preprocess.restype = C.c_int
    lib.gsr_backward_preprocess.argtypes = [C.POINTER(GsrRasterSettings), C.c_int, C.c_int,
                                            vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                            vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_backward_preprocess_sh_adam.restype = C.c_int
    lib.gsr_backward_preprocess_sh_adam.argtypes = [C.POINTER(GsrRasterSettings), C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                                    vp, vp, vp, vp, vp, vp, C.POINTER(ShAdam), vp]
    lib.gsr_preprocess_forward.restype = C.c_int
    lib.gsr_preprocess_forward.argtypes = [C.POINTER(GsrRasterSettings), C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_rasterize_from_splats.restype = C.c_int
    lib.gsr_rasterize_from_splats.argtypes = [C.POINTER(GsrRasterSettings), C.c_int, vp, RESIZE_FN, vp, RESIZE_FN, vp,
                                              RESIZE_FN, vp, vp, vp, C.POINTER(C.c_int32), vp]
    if not missing or "gsr_route_count" not in missing:
        i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        lib.gsr_route_scratch_bytes.restype = C.c_size_t
        lib.gsr_route_scratch_bytes.argtypes = [C.c_int, C.c_int]
        lib.gsr_route_count.restype = C.c_int
        lib.gsr_route_count.argtypes = [C.c_int, vp, C.c_int, i32p, vp, vp, vp]
        lib.gsr_route_pack.restype = C.c_int
        lib.gsr_route_pack.argtypes = [C.c_int, vp, C.c_int, i32p, i64p, vp, vp, vp, vp]
        lib.gsr_rasterize_from_packed.restype = C.c_int
        lib.gsr_rasterize_from_packed.argtypes = [C.POINTER(GsrRasterSettings), C.c_int, vp, RESIZE_FN, vp, RESIZE_FN, vp,
                                                  RESIZE_FN, vp, vp, vp, C.POINTER(C.c_int32), vp]
        lib.gsr_route_return.restype = C.c_int
        lib.gsr_route_return.argtypes = [C.c_int, C.c_int, i64p, vp, vp, vp, vp]
    if not missing or "gsr_route_pack_fixed" not in missing:
        i32p = C.POINTER(C.c_int32)
        lib.gsr_route_pack_fixed.restype = C.c_int
        lib.gsr_route_pack_fixed.argtypes = [C.c_int, vp, C.c_int, i32p, C.c_int, vp, vp, vp, vp, vp]
        lib.gsr_rasterize_from_segments.restype = C.c_int
        lib.gsr_rasterize_from_segments.argtypes = [C.POINTER(GsrRasterSettings), C.c_int, C.c_int, vp, RESIZE_FN, vp, RESIZE_FN, vp,
                                                    RESIZE_FN, vp, vp, vp, C.POINTER(C.c_int32), vp]
    lib.gsr_adam_step.restype = C.c_int
    lib.gsr_adam_step.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, vp]
    lib.gsr_adam_step_multi.restype = C.c_int
    lib.gsr_adam_step_multi.argtypes = [C.POINTER(AdamTensor), C.c_int32, vp]
    lib.gsr_sparse_adam_step.restype = C.c_int
    lib.gsr_sparse_adam_step.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double,
                                         C.c_double, vp]
    lib.gsr_sparse_adam_step_multi.restype = C.c_int
    lib.gsr_sparse_adam_step_multi.argtypes = [C.POINTER(SparseAdamTensor), C.c_int32, vp, C.c_int64, C.c_double, C.c_double, vp]
    lib.gsr_knn_scratch_bytes.restype = C.c_size_t
    lib.gsr_knn_scratch_bytes.argtypes = [C.c_int]
    lib.gsr_knn_mean_dist2.restype = C.c_int
    lib.gsr_knn_mean_dist2.argtypes = [C.c_int, vp, vp, vp, vp]
    lib.gsr_ssim_forward.restype = C.c_int
    lib.gsr_ssim_forward.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_ssim_partial_count.restype = C.c_int64
    lib.gsr_ssim_partial_count.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.gsr_ssim_mean_forward.restype = C.c_int
    lib.gsr_ssim_mean_forward.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_ssim_mean_backward.restype = C.c_int
    lib.gsr_ssim_mean_backward.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    if hasattr(lib, "gsr_train_loss_forward"):
        lib.gsr_train_loss_forward.restype = C.c_int
        lib.gsr_train_loss_forward.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp]
        lib.gsr_train_loss_backward.restype = C.c_int
        lib.gsr_train_loss_backward.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp]
    if hasattr(lib, "gsr_density_stats"):
        lib.gsr_density_stats.restype = C.c_int
        lib.gsr_density_stats.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_ssim_backward.restype = C.c_int
    lib.gsr_ssim_backward.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_mark_visible.restype = C.c_int
    lib.gsr_mark_visible.argtypes = [C.c_int, vp, vp, vp, vp, vp]
    lib.gsr_forward_views.restype = C.c_int
    lib.gsr_forward_views.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, vp, vp, vp, C.POINTER(GsrForwardViews)]
    lib.gsr_profile_enable.restype = C.c_int
    lib.gsr_profile_enable.argtypes = [C.c_int]
    lib.gsr_profile_reset.restype = C.c_int
    lib.gsr_profile_read.restype = C.c_int
    lib.gsr_profile_read.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int]
    if hasattr(lib, "gsr_profile_counters"):
        lib.gsr_profile_counters.restype = C.c_int
        lib.gsr_profile_counters.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int]
    lib.gsr_set_option.restype = C.c_int
    lib.gsr_set_option.argtypes = [C.c_char_p, C.c_int]
    _lib = lib
    # measurement hook: GSR_OPTIONS="name=value,name=value" applies gsr_set_option switches at load time (A/B runs of the
    # test-suite and of bench.py without editing them); unknown names raise
    for kv in filter(None, os.environ.get("GSR_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        check(lib.gsr_set_option(k.strip().encode(), int(v)), f"GSR_OPTIONS {kv}")
    return lib


def check(rc: int, what: str) -> None:
    if rc != GSR_OK:
        msg = load().gsr_last_error()
        raise GsrError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def set_option(name: str, value: int) -> None:
    check(load().gsr_set_option(name.encode(), int(value)), "gsr_set_option")


def profile_enable(on, counters: bool = False, trace: bool = False) -> None:
    """on: per-stage HIP events; counters: the blend kernels' work counters (they slow the kernels: use a separate pass);
    trace: per-wave start / end / placement records of the blend kernels instead of the counters (profile_trace)."""
    load().gsr_profile_enable((1 if on else 0) | (2 if counters else 0) | (4 if trace else 0))


def profile_trace(max_waves: int = 65536):
    """[n, 4] uint64: start, end (100 MHz ticks), placement (HW_ID | XCC << 32 | kernel << 40), steps of the most recent blend launches' waves."""
    import numpy as np
    lib = load()
    lib.gsr_profile_trace.restype = C.c_int
    lib.gsr_profile_trace.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    buf = np.zeros((max_waves, 4), dtype=np.uint64)
    n = lib.gsr_profile_trace(buf.ctypes.data_as(C.POINTER(C.c_uint64)), int(max_waves))
    if n < 0:
        raise GsrError("gsr_profile_trace failed")
    return buf[:n]


def profile_reset() -> None:
    load().gsr_profile_reset()


def profile_counters(reset: bool = True) -> dict:
    out = (C.c_uint64 * 6)()
    check(load().gsr_profile_counters(out, 6, 1 if reset else 0), "gsr_profile_counters")
    return {"fwd_steps": int(out[0]), "fwd_batches": int(out[1]), "bwd_steps": int(out[2]), "bwd_batches": int(out[3]),
            "fwd_max_wave_steps": int(out[4]), "bwd_max_wave_steps": int(out[5])}


def profile_read() -> dict:
    n = len(STAGES)
    ms = (C.c_float * n)()
    cnt = (C.c_int32 * n)()
    load().gsr_profile_read(ms, cnt, n)
    return {STAGES[i]: {"ms": float(ms[i]), "launches": int(cnt[i])} for i in range(n)}
