#!/usr/bin/env python
"""Depth distributions that crowd uniform depth buckets, on the bench frame (1 M Gaussians @1080p), timed with the library under test (GSR_LIB):
  outliers_24    24 Gaussians moved 40-250x farther away (any trained scene has floaters)
  heavy_tails    3 % of the Gaussians moved 0.1-300x nearer / farther (every workgroup of the projection kernel holds some: round 5's "robust" key
                 range, an estimate from per-workgroup extremes, spans them all)
  wall           half of the Gaussians moved onto a slab 0.3 % thick at depth 6
  crowd          3/4 of the Gaussians inside 48 consecutive depth keys (tests/test_gpu_bins_sweep.py depth_crowd)
Round 6 equalises the buckets with a coarse histogram of a key sample (csrc/depthsort.hip ds_hist).  Per case: ms per frame, the depth-sort stage, and the
scan stage -- which only runs on the LSD fallback path, i.e. when a frame met an oversized segment.
    python tools/gpu_depth_distribution_probe.py            one JSON line (also gpurun_out/depth_distribution_probe.json)"""
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
    cases = tuple(os.environ.get("PROBE_CASES", "uniform,outliers_24,heavy_tails,wall,crowd").split(","))      # (PROBE_CASES=crowd under rocprofv3: that frame's kernels)
    for case in cases:
        sc = make_scene(P, cam, seed=0, s_med=0.012)
        g = torch.Generator().manual_seed(5)

        def rescale(idx, f):      # camera at the origin looking +z: same pixel, f times the depth, same footprint
            sc.means3D[idx] = sc.means3D[idx] * f[:, None]
            sc.scales[idx] = sc.scales[idx] * f[:, None]
        if case == "outliers_24":
            idx = torch.randperm(P, generator=g)[:24]
            rescale(idx, 40.0 + 210.0 * torch.rand(24, generator=g))
            sc.opacities[idx] = 0.9
        elif case == "heavy_tails":
            idx = torch.nonzero(torch.rand(P, generator=g) < 0.03).reshape(-1)
            rescale(idx, torch.exp(torch.empty(idx.numel()).uniform_(-2.3, 5.7, generator=g)))
        elif case == "wall":
            idx = torch.nonzero(torch.rand(P, generator=g) < 0.5).reshape(-1)
            rescale(idx, (6.0 + 0.009 * torch.randn(idx.numel(), generator=g)) / sc.means3D[idx, 2])
        elif case == "crowd":
            idx = torch.nonzero(torch.rand(P, generator=g) < 0.75).reshape(-1)
            z = (torch.tensor(5.0).view(torch.int32) + torch.randint(0, int(os.environ.get("PROBE_CROWD_VALUES", "48")), (idx.numel(),), generator=g, dtype=torch.int32)).view(torch.float32)
            rescale(idx, z / sc.means3D[idx, 2])
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
        out[case] = {"ms_per_frame": round(ms, 4), "depth_sort_ms": round(st["depth_sort"]["ms"] / max(1, st["depth_sort"]["launches"]), 4),
                                    "scan_ms": round(st["scan"]["ms"] / max(1, st["scan"]["launches"]), 4) if st.get("scan", {}).get("launches") else 0.0}
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "depth_distribution_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
