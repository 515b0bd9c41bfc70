"""Measured timeline of the two blend launches (is the tail of the launch -- heaviest wave 1.8-2.0 x the mean --
what the blend backward loses?).  With gsr_profile_enable(4) every wave of render_fwd_wave_bf / render_bwd_half records its start and end time
(s_memrealtime, 10 ns), the SIMD it ran on and its blend steps.  From that:
  span_us            first start .. last end
  drain_frac         (last end - last START) / span: the part of the launch during which the dispatcher had nothing left to hand out
  simd_idle_frac     mean over the SIMDs of (last end of the launch - last end on this SIMD) / span: SIMD-time lost behind the slowest SIMD
  mean_resident      average resident waves per SIMD over the span
  rate_mid           blend steps retired per us during the central half of the launch (every wave's steps spread evenly over its lifetime)
  balanced_span_us   total steps / rate_mid: how long the launch would take if it ran at its mid-launch rate from start to end
  tail_loss          span / balanced_span - 1: what a perfectly balanced launch (split long lists, finer work units) could gain AT MOST
One JSON line (also gpurun_out/wave_trace.json; raw traces in gpurun_out/wave_trace_*.npy).  Measurement tool, not product code."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from gsr_synth import make_camera, make_scene, make_clustered_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib, rasterize_gaussians

dev = torch.device("cuda:0")
MODES = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3, 2, 1, 0]
W, H = 1920, 1080
TICK_US = 0.01


def analyse(tr, kernel):
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


def run(name, sc):
    cam = make_camera(W, H)
    camd = cam.to(dev)
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                       camd.full_proj_transform, 3, camd.camera_center, False, False, False)
    d = sc.to(dev)
    L = [t.detach().clone().requires_grad_(True) for t in (d.means3D, d.shs, d.opacities, d.scales, d.rotations)]
    wc = torch.randn(3, H, W, device=dev)
    out = {}
    ref_grads = None
    names = {0: "backward, tiles in index order", 1: "backward, heaviest tiles first (sum of the four blocks)",
             2: "backward, tiles by their heaviest half", 3: "backward, half tiles (waves) heaviest first"}
    for heavy in MODES:
        _lib.set_option("bwd_heavy_first", heavy)
        for _ in range(3):      # warm
            col, _, _ = rasterize_gaussians(L[0], None, L[1], None, L[2], L[3], L[4], None, rs)
            col.backward(wc)
        torch.cuda.synchronize()
        _lib.profile_enable(False, trace=True)
        _lib.profile_trace()      # clear
        if heavy == MODES[0]:
            with torch.no_grad():
                rasterize_gaussians(L[0], None, L[1], None, L[2], L[3], L[4], None, rs)
            tr = _lib.profile_trace()
            np.save(os.path.join(ROOT, "gpurun_out", f"wave_trace_{name}_fwd.npy"), tr)
            out["forward (inference build)"] = analyse(tr, 1)
        col, _, _ = rasterize_gaussians(L[0], None, L[1], None, L[2], L[3], L[4], None, rs)
        _lib.profile_trace()      # drop the tracking forward's records
        col.backward(wc)
        tr = _lib.profile_trace()
        _lib.profile_enable(False)
        np.save(os.path.join(ROOT, "gpurun_out", f"wave_trace_{name}_bwd_heavy{heavy}.npy"), tr)
        out[names[heavy]] = analyse(tr, 2)
        grads = [t.grad.clone() for t in L]
        if ref_grads is None:
            ref_grads = grads
        else:      # the launch order is not allowed to change one bit of any gradient (every wave writes its own record slots)
            out[names[heavy]]["gradients_bit_identical_to_first_mode"] = bool(all(torch.equal(a_, b_) for a_, b_ in zip(grads, ref_grads)))
        for t in L:
            t.grad = None
    _lib.set_option("bwd_heavy_first", 2)
    return out


os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
res = {}
for name, sc in (("uniform", make_scene(1_000_000, make_camera(W, H), seed=0, s_med=0.012)),
                 ("clustered", make_clustered_scene(1_000_000, make_camera(W, H), seed=0))):
    res["configs[1] " + name] = run(name, sc)
line = json.dumps({"what": "per-wave trace of the blend launches (tools/gpu_wave_trace.py; 1 M Gaussians @1080p)", "results": res})
print(line)
open(os.path.join(ROOT, "gpurun_out", "wave_trace.json"), "w").write(line + "\n")
