"""A/B of library options inside ONE process on ONE box (a bench.py run per configuration costs a minute of box time each; this costs
seconds): the bench frame (configs[1] stand-in) is rendered `frames` times per configuration, configurations interleaved over `reps`
rounds, one line per configuration with the median frame time (wall clock and HIP events) and the library's stage table.

    python tools/gpu_opt_sweep.py [--train] [--frames 100] [--reps 3] "name=value[,name=value]" ...      ("" = defaults)

Measurement tool, not product code."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting_amd")):
    sys.path.insert(0, p)
import torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib, rasterize_gaussians

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", default=[""])
ap.add_argument("--frames", type=int, default=100)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--P", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--train", action="store_true", help="time forward (tracking build) + backward instead of the inference forward")
a = ap.parse_args()
DEFAULTS = {"preprocess_grid_cap": 1024, "snug_tiles": 1, "bwd_heavy_first": 2, "depth_sort_mode": 0, "level2_scan_mode": 0, "tile_sort_mode": 0}

dev = torch.device("cuda:0")
_lib.load()
cam = make_camera(a.width, a.height)
sc = make_scene(a.P, cam, seed=0, s_med=0.012).to(dev)
camd = cam.to(dev)
rs = GaussianRasterizationSettings(a.height, a.width, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                   camd.full_proj_transform, 3, camd.camera_center, False, False, False)
leaves = [t.detach().clone().requires_grad_(a.train) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
wc = torch.randn(3, a.height, a.width, device=dev)


def step():
    if a.train:
        col, _, _ = rasterize_gaussians(leaves[0], None, leaves[1], None, leaves[2], leaves[3], leaves[4], None, rs)
        col.backward(wc)
        for t in leaves:
            t.grad = None
    else:
        with torch.no_grad():
            rasterize_gaussians(leaves[0], None, leaves[1], None, leaves[2], leaves[3], leaves[4], None, rs)


def apply(cfg):
    for k, v in DEFAULTS.items():
        _lib.set_option(k, v)
    for kv in [x for x in cfg.split(",") if x]:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))


res = {c: {"wall": [], "ev": [], "stages": []} for c in a.configs}
for _ in range(30):
    step()
torch.cuda.synchronize()
for rep in range(a.reps):
    for cfg in a.configs:
        apply(cfg)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.frames):
            step()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        # stage table in its own short pass (the event pairs around every stage perturb the pipeline)
        _lib.profile_reset()
        _lib.profile_enable(True)
        for _ in range(max(10, a.frames // 4)):
            step()
        torch.cuda.synchronize()
        st = {k: v["ms"] / v["launches"] for k, v in _lib.profile_read().items() if v["launches"]}
        _lib.profile_enable(False)
        res[cfg]["wall"].append((t1 - t0) * 1e3 / a.frames)
        res[cfg]["ev"].append(e0.elapsed_time(e1) / a.frames)
        res[cfg]["stages"].append(st)
apply("")
out = []
for cfg, r in res.items():
    stages = {}
    for st in r["stages"]:
        for k, v in st.items():
            stages.setdefault(k, []).append(v)
    row = {"config": cfg or "(defaults)", "ms_wall_median": round(statistics.median(r["wall"]), 4), "ms_wall_all": [round(x, 4) for x in r["wall"]],
           "ms_events_median": round(statistics.median(r["ev"]), 4),
           "stage_ms_median": {k: round(statistics.median(v), 4) for k, v in stages.items() if all(isinstance(x, (int, float)) for x in v)}}
    out.append(row)
    print(json.dumps(row))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": "tools/gpu_opt_sweep.py", "train": a.train, "frames": a.frames, "reps": a.reps, "rows": out},
          open(os.path.join(ROOT, "gpurun_out", "opt_sweep_train.json" if a.train else "opt_sweep.json"), "w"), indent=1)
