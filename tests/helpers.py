"""Shared helpers for the test-suite (scene construction, oracle calls, host-math harness)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-splatting_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

from gsr_synth import make_camera, look_at_camera, make_scene, make_edge_scene, make_clustered_scene  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

# The product bins every Gaussian into its SNUG tile rectangle (csrc/gsr_math.h); the oracle's default is the reference's square.
# Every test that compares the product's integers (tiles_touched, lists, ranges, contributor positions) with the oracle's goes
# through this module and gets the snug restatement; tests/test_oracle.py checks that the switch changes no output.
O.SNUG_TILES = True


class reference_tiles:
    """`with reference_tiles():` -- the oracle bins the reference's tile square (frozen reference-side vectors)."""

    def __enter__(self):
        self.prev, O.SNUG_TILES = O.SNUG_TILES, False

    def __exit__(self, *exc):
        O.SNUG_TILES = self.prev


def oracle_settings(cam, bg=None, sh_degree=3, scale_modifier=1.0, antialiasing=False):
    bg = torch.zeros(3) if bg is None else bg
    return O.settings_from_camera(cam, bg, sh_degree, scale_modifier, antialiasing)


class HostCam(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("sh_degree", C.c_int), ("M", C.c_int), ("antialiasing", C.c_int),
                ("tile_y0", C.c_int), ("tile_y1", C.c_int), ("view", C.c_float * 16), ("proj", C.c_float * 16),
                ("campos", C.c_float * 3)]


_host_lib = None


def host_math_lib():
    """g++ build of the product's per-Gaussian math header (tests/host_math_harness.cpp)."""
    global _host_lib
    if _host_lib is None:
        out = os.path.join(ROOT, "tests", "_build")
        os.makedirs(out, exist_ok=True)
        so = os.path.join(out, "libhostmath.so")
        src = os.path.join(ROOT, "tests", "host_math_harness.cpp")
        hdr = os.path.join(PKG, "csrc", "gsr_math.h")
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(PKG, "csrc"),
                                   "-I", os.path.join(ROOT, "include"), src, "-o", so])
        _host_lib = C.CDLL(so)
    return _host_lib


def host_cam(s, M, tile_y0=0, tile_y1=0):
    hc = HostCam()
    hc.W, hc.H = int(s.image_width), int(s.image_height)
    hc.tanfovx, hc.tanfovy = float(s.tanfovx), float(s.tanfovy)
    hc.scale_modifier = float(s.scale_modifier)
    hc.sh_degree, hc.M, hc.antialiasing = int(s.sh_degree), int(M), int(bool(s.antialiasing))
    hc.tile_y0, hc.tile_y1 = tile_y0, tile_y1
    for i, v in enumerate(s.viewmatrix.reshape(-1).tolist()):
        hc.view[i] = v
    for i, v in enumerate(s.projmatrix.reshape(-1).tolist()):
        hc.proj[i] = v
    for i, v in enumerate(s.campos.reshape(-1).tolist()):
        hc.campos[i] = v
    return hc


def fptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def np32(t):
    return None if t is None else np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


def parity_report(key, **kw):
    """Measured parity numbers (max errors, fragile fractions) are printed and collected into gpurun_out/parity_report.json
    (copied to profiles/ by the builder), so that the tolerances are visible, not hidden in asserts."""
    import json
    print(f"[parity] {key}: " + ", ".join(f"{k}={v}" for k, v in kw.items()), flush=True)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "parity_report.json")
        old = {}
        if os.path.exists(path):
            try:
                old = json.load(open(path))
            except Exception:
                old = {}
        old[key] = kw
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    except Exception:
        pass
