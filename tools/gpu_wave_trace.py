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


from diff_gaussian_rasterization.debug import wave_timeline as analyse      # noqa: E402


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
