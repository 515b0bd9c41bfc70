#!/usr/bin/env python
"""a scene whose depths have a few far outliers -- any trained scene has floaters -- stretches the TRUE key range of the depth
bucket sort until the bulk of the Gaussians shares a few buckets; their segments overflow the LDS capacity, go through global memory, and the host falls
back to the LSD passes.  Round 5 spans the buckets over a ROBUST range (csrc/depthsort.hip ds_hist).  This probe times the bench frame with N Gaussians
moved 40-250x farther away (same screen position, scaled up so they keep their tiles), with the library under test (GSR_LIB) -- run it once per library:
    GSR_LIB=.../lib/libgsr_hip.so python tools/gpu_outlier_probe.py ; GSR_LIB=.../lib_truerange/libgsr_hip.so python tools/gpu_outlier_probe.py
One JSON line: ms per frame and the depth-sort stage with 0 / 24 outliers."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
from gsr_synth import make_camera, make_scene      # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib, rasterize_gaussians      # noqa: E402


def main():
    dev = torch.device("cuda:0")
    W, H, P = 1920, 1080, 1_000_000
    cam = make_camera(W, H)
    out = {"lib": os.environ.get("GSR_LIB", "product")}
    for n_out in (0, 24):
        sc = make_scene(P, cam, seed=0, s_med=0.012)
        if n_out:
            g = torch.Generator().manual_seed(5)
            idx = torch.randperm(P, generator=g)[:n_out]
            f = 40.0 + 210.0 * torch.rand(n_out, generator=g)
            sc.means3D[idx] = sc.means3D[idx] * f[:, None]       # camera at the origin looking +z: same pixel, f times the depth
            sc.scales[idx] = sc.scales[idx] * f[:, None]
            sc.opacities[idx] = 0.9
        d = sc.to(dev)
        camd = cam.to(dev)
        rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                           camd.full_proj_transform, 3, camd.camera_center, False, False, False)

        def step():
            with torch.no_grad():
                return rasterize_gaussians(d.means3D, None, d.shs, None, d.opacities, d.scales, d.rotations, None, rs, None)
        for _ in range(80):      # (covers the library's first LSD stay of 64 frames if the slow path triggers)
            step()
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        _lib.profile_reset()
        _lib.profile_enable(True)
        for _ in range(40):
            step()
        torch.cuda.synchronize()
        st = _lib.profile_read()
        _lib.profile_enable(False)
        out[f"outliers_{n_out}"] = {"ms_per_frame": round(ms, 4), "depth_sort_ms": round(st["depth_sort"]["ms"] / max(1, st["depth_sort"]["launches"]), 4),
                                    "scan_ms": round(st["scan"]["ms"] / max(1, st["scan"]["launches"]), 4) if st.get("scan", {}).get("launches") else 0.0}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
