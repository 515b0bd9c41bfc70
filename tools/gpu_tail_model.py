"""Does the tail of the blend launches cost anything?  ("heaviest wave = 1.8-2.0 x the mean ... nothing splits a long list".)
The tracking forward leaves the number of entries every 8x8 block blended (`block_steps`, image buffer, right behind the tile ranges: DESIGN 2);
a wave's duration is proportional to its steps.  This tool reads them for the bench frame and list-schedules the waves of the forward (one per 8x8
block, launch order) and of the backward (one per 16x8 half tile; steps = the union of its two blocks' survivors, bracketed by max(b0, b1) and
b0 + b1) onto S wave slots, in index order and heaviest tile first (what bwd_plan_kernel does): makespan / (total work / S) is what a perfectly
balanced launch could gain.  One JSON line (also gpurun_out/tail_model.json).  Analysis tool, not product code."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import torch
from gsr_synth import make_camera, make_scene, make_clustered_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings
from diff_gaussian_rasterization.debug import forward_with_views

dev = torch.device("cuda:0")


from sched_model import F, makespan, ps_makespan      # noqa: E402


def analyse(name, sc, W, H):
    cam = make_camera(W, H)
    camd = cam.to(dev)
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                       camd.full_proj_transform, 3, camd.camera_center, False, False, False)
    d = sc.to(dev)
    out = forward_with_views(rs, d.means3D, d.opacities, shs=d.shs, scales=d.scales, rotations=d.rotations)
    torch.cuda.synchronize()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    nt = gx * gy
    img = out["buffers"][2]
    off = out["ranges"].data_ptr() - img.data_ptr()
    off_bs = (off + nt * 8 + 127) // 128 * 128
    bs = img[off_bs:off_bs + nt * 16].view(torch.int32).view(nt, 4).cpu().numpy().astype(np.int64)
    res = {"R": int(out["R"]), "tiles": nt, "fwd_steps": int(bs.sum())}
    np.save(os.path.join(ROOT, "gpurun_out", "block_steps_" + name.split()[-1] + ".npy"), bs.astype(np.int32))
    # forward: workgroup b -> tile (b >> 5) * 8 + (b & 7), quad (b & 31) >> 3  (render_fwd.hip)
    groups = (nt + 7) // 8
    b = np.arange(groups * 32)
    tl, quad = (b >> 5) * 8 + (b & 7), (b & 31) >> 3
    ok = tl < nt
    fwd_jobs = bs[tl[ok], quad[ok]]
    res["fwd"] = {"waves": int(len(fwd_jobs)), "heaviest_over_mean": round(float(fwd_jobs.max() / max(1e-9, fwd_jobs.mean())), 3)}
    for S in (4096, 8192):
        res["fwd"][f"makespan_over_ideal_S{S}"] = round(makespan(fwd_jobs, S) / (fwd_jobs.sum() / S), 4)
    for K in (4, 8):
        res["fwd"][f"ps_launch_order_K{K}"] = round(ps_makespan(fwd_jobs, 1024, K) / (fwd_jobs.sum() / (1024 * F[K])), 4)
        res["fwd"][f"ps_heaviest_first_K{K}"] = round(ps_makespan(np.sort(fwd_jobs)[::-1], 1024, K) / (fwd_jobs.sum() / (1024 * F[K])), 4)
    # backward: top half = quads 0, 1; bottom half = quads 2, 3
    for est, f in (("max", lambda a, c: np.maximum(a, c)), ("sum", lambda a, c: a + c)):
        half = np.stack([f(bs[:, 0], bs[:, 1]), f(bs[:, 2], bs[:, 3])], axis=1)      # [tile, half]
        order_heavy = np.argsort(-bs.sum(axis=1), kind="stable")
        r = {"waves": int(half.size), "heaviest_over_mean": round(float(half.max() / max(1e-9, half.mean())), 3)}
        for S in (3072, 4096, 6144):
            ideal = half.sum() / S
            r[f"index_order_S{S}"] = round(makespan(half.reshape(-1), S) / ideal, 4)
            r[f"heaviest_first_S{S}"] = round(makespan(half[order_heavy].reshape(-1), S) / ideal, 4)
        # processor sharing inside a SIMD (a lone wave issues one VALU per ~5 cycles, four or more share the SIMD's full rate):
        # 1024 SIMDs x K resident waves, the next wave of the launch goes to the SIMD whose wave finished
        order_lpt = np.argsort(-half.reshape(-1), kind="stable")
        for K in (3, 4, 6):
            ideal = half.sum() / (1024 * F[K])
            r[f"ps_index_order_K{K}"] = round(ps_makespan(half.reshape(-1), 1024, K) / ideal, 4)
            r[f"ps_heaviest_tile_first_K{K}"] = round(ps_makespan(half[order_heavy].reshape(-1), 1024, K) / ideal, 4)
            r[f"ps_heaviest_wave_first_K{K}"] = round(ps_makespan(half.reshape(-1)[order_lpt], 1024, K) / ideal, 4)
        res[f"bwd_steps_{est}"] = r
    return name, res


W, H = 1920, 1080
results = {}
for name, sc in (("configs[1] uniform", make_scene(1_000_000, make_camera(W, H), seed=0, s_med=0.012)),
                 ("configs[1] clustered", make_clustered_scene(1_000_000, make_camera(W, H), seed=0))):
    k, v = analyse(name, sc, W, H)
    results[k] = v
line = json.dumps({"what": "list-scheduling model of the blend launches from the measured per-block steps (tools/gpu_tail_model.py): makespan / (total / slots)",
                   "results": results})
print(line)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "tail_model.json"), "w").write(line + "\n")
