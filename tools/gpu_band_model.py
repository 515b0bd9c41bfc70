"""Per-rank latency of the screen-sharded forward, measured on ONE GPU (no multi-GPU box is available to the builder):
for G = 1, 2, 4, 8 every band of the instance-balanced plan is rendered on its own (exactly what rank g of a G-GPU run
executes: replicated preprocess with colours for the band only, depth sort, scan, binning and blend of the band) and timed;
the slowest band sets the frame time of the G-GPU run.  Together with the xGMI model of the strip all-gather
 this gives the predicted strong-scaling curve that the driver's SCALE run can be checked against.

    python tools/gpu_band_model.py            # configs[1], [3], [4] stand-ins
Output: one JSON document on stdout (also gpurun_out/band_model.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, rasterize_gaussians, _lib
from diff_gaussian_rasterization.debug import forward_with_views
from diff_gaussian_rasterization.parallel import BandPlan, row_costs_from_ranges

dev = torch.device("cuda:0")
LINKS, LINK_GBS, HOP_US = 7, 153.0, 15.0      # MI355X node: 7 xGMI links x ~153 GB/s per GPU; collective launch + first-byte latency


def time_band(rs, sc, band, steps=20):
    def step():
        with torch.no_grad():
            rasterize_gaussians(sc.means3D, None, sc.shs, None, sc.opacities, sc.scales, sc.rotations, None, rs, band)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    # median of per-frame event pairs: the allocator occasionally maps a fresh block when the band (and R) changes
    ts = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    ms = ts[len(ts) // 2]
    _lib.profile_reset(); _lib.profile_enable(True)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    st = _lib.profile_read(); _lib.profile_enable(False)
    return ms, {k: round(v["ms"] / max(1, v["launches"]), 4) for k, v in st.items() if v["launches"]}


out = {}
for name, P, W, H in (("configs[1] 1M@1080p", 1_000_000, 1920, 1080), ("configs[3] 1M@4K", 1_000_000, 3840, 2160),
                      ("configs[4] 6M@1080p", 6_000_000, 1920, 1080)):
    cam = make_camera(W, H)
    sc = make_scene(P, cam, seed=0, s_med=0.012).to(dev)
    camd = cam.to(dev)
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                       camd.full_proj_transform, 3, camd.camera_center, False, False, False)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        v0 = forward_with_views(rs, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    row_cost = row_costs_from_ranges(v0["ranges"].long(), gx, gy, banded=False)
    R = int(v0["R"])
    del v0
    t1, st1 = time_band(rs, sc, None)
    res = {"P": P, "W": W, "H": H, "R": R, "1": {"frame_ms": round(t1, 4), "stage_ms": st1}}
    strip_bytes = 3 * W * H * 4
    for G in (2, 4, 8):
        plan = BandPlan.balanced(row_cost, G)
        per = []
        for g in range(G):
            ms, st = time_band(rs, sc, plan.band(g), steps=10)
            per.append({"band": plan.band(g), "ms": round(ms, 4), "stage_ms": st})
        slow = max(p["ms"] for p in per)
        slow_stages = max(sum(p["stage_ms"].values()) for p in per)
        # strip all-gather, direct (all links at once): every rank receives (G-1)/G of the frame over min(G-1, 7) links
        comm_us = HOP_US + strip_bytes * (G - 1) / G / (min(G - 1, LINKS) * LINK_GBS * 1e3)
        res[str(G)] = {"slowest_band_ms": slow, "slowest_band_sum_of_stages_ms": round(slow_stages, 4), "bands": per, "allgather_model_us": round(comm_us, 1),
                       "speedup_if_gather_overlapped": round(t1 / max(slow, comm_us * 1e-3), 2),
                       "speedup_if_gather_serial": round(t1 / (slow + comm_us * 1e-3), 2)}
    out[name] = res
    del sc
    torch.cuda.empty_cache()
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "band_model.json"), "w"), indent=1)
