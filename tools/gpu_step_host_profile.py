#!/usr/bin/env python
"""Where does the headline train step's WALL time go on the host side?  The bench's train step (P = 1 M @1080p, 32 views cycled, fused loss, fused Adam), 300
iterations three ways: free-running wall clock per step; host seconds each phase takes to ENQUEUE (forward -- which contains the R read-back wait --, loss,
backward, optimizer); and GPU time per phase from torch events.  Run once per library (GSR_LIB) on ONE box to compare libraries:
    python tools/gpu_step_host_profile.py            -> one JSON line
Measurement tool."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting_amd")):
    sys.path.insert(0, p)
import torch
from gsr_synth import look_at_camera, make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, rasterize_gaussians
from fused_ssim import fused_train_loss
from gsr_optim import FusedAdam
sys.path.insert(0, ROOT)
from bench import make_views

dev = torch.device("cuda:0")
W, H, P = 1920, 1080, 1_000_000
cam = make_camera(W, H)
sc = make_scene(P, cam, seed=0, s_med=0.012).to(dev)
bg = torch.zeros(3, device=dev)
views = make_views(make_camera, look_at_camera, W, H, 32)
rs_views, gts = [], []
for i, vc in enumerate(views):
    vd = vc.to(dev)
    rs_views.append(GaussianRasterizationSettings(H, W, vc.tanfovx, vc.tanfovy, bg, 1.0, vd.world_view_transform, vd.full_proj_transform, 3, vd.camera_center, False, False, False))
    gts.append(torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1 + i)))
params = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
opt = FusedAdam(params, lr=1e-5, eps=1e-15)
it = [0]


def step(host=None, marks=None):
    t = [time.perf_counter()]

    def mark(name):
        now = time.perf_counter()
        if host is not None:
            host[name] = host.get(name, 0.0) + now - t[0]
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))
        t[0] = time.perf_counter()
    vi = it[0] % len(rs_views)
    it[0] += 1
    if marks is not None:
        mark("start")
    opt.zero_grad(set_to_none=True)
    color, radii, invd = rasterize_gaussians(params[0], None, params[1], None, params[2], params[3], params[4], None, rs_views[vi], None)
    mark("forward")
    loss = fused_train_loss(color, gts[vi])
    mark("loss")
    loss.backward()
    mark("backward")
    opt.step()
    mark("optimizer")


for _ in range(64):
    step()
torch.cuda.synchronize()
N = 300
out = {"lib": os.environ.get("GSR_LIB", "product")}
walls = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) / N * 1e3)
out["wall_ms_per_step"] = [round(x, 4) for x in walls]
host = {}
t0 = time.perf_counter()
for _ in range(N):
    step(host)
torch.cuda.synchronize()
out["wall_ms_in_host_pass"] = round((time.perf_counter() - t0) / N * 1e3, 4)
out["host_enqueue_ms"] = {k: round(v / N * 1e3, 4) for k, v in host.items()}
marks = []
for _ in range(N):
    step(None, marks)
torch.cuda.synchronize()
ph = {}
for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
    key = n1 if n1 != "start" else "between"
    ph[key] = ph.get(key, 0.0) + e0.elapsed_time(e1)
out["gpu_ms"] = {k: round(v / N, 4) for k, v in ph.items()}
out["gpu_ms_total"] = round(sum(ph.values()) / N, 4)
# synchronous steps: the latency of one step with nothing overlapped
t0 = time.perf_counter()
for _ in range(100):
    step()
    torch.cuda.synchronize()
out["sync_ms_per_step"] = round((time.perf_counter() - t0) / 100 * 1e3, 4)
# where the largest arrays of the optimizer sit (the SH tensor: parameter, gradient, both moments)
try:
    st_sh = opt.state[params[1]]
    out["sh_ptrs"] = {"p": hex(params[1].data_ptr()), "g": hex(params[1].grad.data_ptr()), "m": hex(st_sh["exp_avg"].data_ptr()), "v": hex(st_sh["exp_avg_sq"].data_ptr())}
except Exception as ex:      # noqa: BLE001
    out["sh_ptrs"] = repr(ex)
# Does the optimizer phase depend on where the allocator puts things?  (round 6: the same Adam kernel at 257 and at 298 us)  A dummy allocation of `pad` bytes
# is made BEFORE the allocator's cache is dropped and the step re-allocates its scratch and gradients; 100 steps each, GPU time of the phases from events.
if os.environ.get("PERTURB", "1") == "1":
    out["gpu_ms_by_allocation_perturbation"] = {}
    for pad in (0, 1 << 20, 7 << 20, 33 << 20, 129 << 20, 0):
        torch.cuda.synchronize()
        dummy = None
        torch.cuda.empty_cache()
        dummy = torch.empty(max(pad, 1), dtype=torch.uint8, device=dev)
        for _ in range(20):
            step()
        marks = []
        for _ in range(100):
            step(None, marks)
        torch.cuda.synchronize()
        ph = {}
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            key = n1 if n1 != "start" else "between"
            ph[key] = ph.get(key, 0.0) + e0.elapsed_time(e1)
        g = [t.grad.data_ptr() % (1 << 21) for t in params]
        out["gpu_ms_by_allocation_perturbation"][f"{pad >> 20} MiB #{len(out['gpu_ms_by_allocation_perturbation'])}"] = {
            "optimizer": round(ph["optimizer"] / 100, 4), "backward": round(ph["backward"] / 100, 4), "forward": round(ph["forward"] / 100, 4),
            "total": round(sum(ph.values()) / 100, 4), "grad_ptr_mod_2MiB": g, "sh_grad_ptr": hex(params[1].grad.data_ptr())}
        del dummy
print(json.dumps(out))
