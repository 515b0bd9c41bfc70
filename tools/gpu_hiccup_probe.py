"""Where do the ~1 ms host-side hiccups of the forward loop come from?  (round 4: same box, same library, `python bench.py --steps 30`:
per-frame GPU-event MEDIAN 0.349 ms in every run, wall-clock MEAN 0.353 in two runs and 0.385 in four -- about one 1 ms pause per timed
loop; the host is in the loop of every frame through the R read-back, so a host pause is a GPU bubble.)  Runs the bench frame's forward
400 frames at a time under a few host-side settings and reports, per setting: wall mean, event median, the frames whose host-side gap
exceeds 1.5 x the median (index, size), and every garbage-collector run that fell into the loop (generation, duration, frame index).

    python tools/gpu_hiccup_probe.py [frames]      -> one JSON line, gpurun_out/hiccup_probe.json
Measurement tool, not product code."""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gaussian-splatting_amd")):
    sys.path.insert(0, _p)
import torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib, rasterize_gaussians

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
_lib.load()
W, H, P = 1920, 1080, 1_000_000
cam = make_camera(W, H)
sc = make_scene(P, cam, seed=0, s_med=0.012).to(dev)
camd = cam.to(dev)
rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                   camd.full_proj_transform, 3, camd.camera_center, False, False, False)


def step():
    with torch.no_grad():
        return rasterize_gaussians(sc.means3D, None, sc.shs, None, sc.opacities, sc.scales, sc.rotations, None, rs, None)[0]


gc_log = []
frame_no = [0]
_t_gc = [0.0]


def _gc_cb(phase, info):
    if phase == "start":
        _t_gc[0] = time.perf_counter()
    else:
        gc_log.append({"frame": frame_no[0], "generation": info["generation"], "ms": round((time.perf_counter() - _t_gc[0]) * 1e3, 3),
                       "collected": info["collected"]})


gc.callbacks.append(_gc_cb)


def run(tag, n, events=True):
    gc_log.clear()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)] if events else None
    stamps = [time.perf_counter()]
    if events:
        evs[0].record()
    for i in range(n):
        frame_no[0] = i
        step()
        if events:
            evs[i + 1].record()
        stamps.append(time.perf_counter())
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    gaps = [(b - a) * 1e3 for a, b in zip(stamps[:-1], stamps[1:])]
    srt = sorted(gaps)
    med = srt[len(srt) // 2]
    big = [(i, round(g, 3)) for i, g in enumerate(gaps) if g > 1.5 * med]
    out = {"setting": tag, "frames": n, "wall_mean_ms": round((t_end - stamps[0]) * 1e3 / n, 4), "host_gap_median_ms": round(med, 4),
           "host_gap_p99_ms": round(srt[min(n - 1, (n * 99) // 100)], 4), "host_gap_max_ms": round(srt[-1], 3),
           "frames_over_1.5x_median": big[:24], "n_frames_over_1.5x_median": len(big),
           "excess_ms_per_frame": round(sum(g - med for _, g in big) / n, 4), "final_sync_ms": round((t_end - stamps[-1]) * 1e3, 3),
           "gc_runs": list(gc_log)[:24], "n_gc_runs": len(gc_log), "gc_counts_after": gc.get_count()}
    if events:
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
        out["event_median_ms"] = round(per[n // 2], 4)
        out["event_mean_ms"] = round(sum(per) / n, 4)
        out["event_max_ms"] = round(per[-1], 3)
    return out


for _ in range(30):
    step()
t0 = time.time()
while time.time() - t0 < 1.0:
    step()
res = []
res.append(run("default (gc on)", N))
res.append(run("default (gc on), second pass", N))
gc.collect()
gc.disable()
res.append(run("gc.disable()", N))
res.append(run("gc.disable(), no events recorded in the loop", N, events=False))
gc.enable()
gc.collect()
gc.freeze()
res.append(run("gc.freeze() after a full collection, gc on", N))
gc.disable()
try:
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[len(os.sched_getaffinity(0)) // 2]})
    res.append(run("gc.disable() + pinned to one core", N))
except Exception as ex:      # noqa: BLE001
    res.append({"setting": "pinning failed", "error": repr(ex)})
line = json.dumps({"what": "host-side hiccups of the forward loop (tools/gpu_hiccup_probe.py), 1 M Gaussians @1080p", "host_cores": os.cpu_count(),
                   "gc_threshold": gc.get_threshold(), "runs": res})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "hiccup_probe.json"), "w").write(line)
print(line)
