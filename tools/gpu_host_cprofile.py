#!/usr/bin/env python
"""Host-side cost of one training iteration through the packages (measurement tool): a SMALL scene (P = 20 000 @ 640x360: the GPU side is ~0.15 ms, so the
loop is host-bound and its wall clock IS the host cost), the bench's fused step and the reference's call sequence (render glue with torch activations,
separate_sh, fused_ssim + torch l1, loss.item(), SparseGaussianAdam), each 400 iterations free-running and then under cProfile.
    python tools/gpu_host_cprofile.py            -> wall ms per iteration + the top functions by own time"""
import cProfile
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting_amd")):
    sys.path.insert(0, p)
import torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, SparseGaussianAdam, rasterize_gaussians
from fused_ssim import fused_ssim, fused_train_loss
from gsr_optim import FusedAdam

dev = torch.device("cuda:0")
W, H, P = 640, 360, 20_000
cam = make_camera(W, H)
sc = make_scene(P, cam, seed=0, s_med=0.03).to(dev)
bg = torch.zeros(3, device=dev)
cd = cam.to(dev)
rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg, 1.0, cd.world_view_transform, cd.full_proj_transform, 3, cd.camera_center, False, False, False)
gt = torch.rand(3, H, W, device=dev)

# ---- A: the bench's fused step
pa = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
opt_a = FusedAdam(pa, lr=1e-5, eps=1e-15)


def step_fused():
    opt_a.zero_grad(set_to_none=True)
    color, radii, invd = rasterize_gaussians(pa[0], None, pa[1], None, pa[2], pa[3], pa[4], None, rs, None)
    fused_train_loss(color, gt).backward()
    opt_a.step()


# ---- B: the reference's sequence (gaussian_renderer/__init__.py:18-128 separate_sh form, train.py:104-186 without densification)
xyz = torch.nn.Parameter(sc.means3D.clone())
f_dc = torch.nn.Parameter(sc.shs[:, :1].contiguous().clone())
f_rest = torch.nn.Parameter(sc.shs[:, 1:].contiguous().clone())
raw_op = torch.nn.Parameter(torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)))
raw_sc = torch.nn.Parameter(torch.log(sc.scales))
raw_rot = torch.nn.Parameter(sc.rotations.clone())
groups = [{"params": [xyz], "lr": 1e-5, "name": "xyz"}, {"params": [f_dc], "lr": 2.5e-3, "name": "f_dc"}, {"params": [f_rest], "lr": 1.25e-4, "name": "f_rest"},
          {"params": [raw_op], "lr": 0.025, "name": "opacity"}, {"params": [raw_sc], "lr": 5e-3, "name": "scaling"}, {"params": [raw_rot], "lr": 1e-3, "name": "rotation"}]
opt_b = SparseGaussianAdam(groups, lr=0.0, eps=1e-15)


def step_reference_sequence():
    screenspace = torch.zeros_like(xyz, requires_grad=True) + 0
    screenspace.retain_grad()
    out = GaussianRasterizer(rs)(means3D=xyz, means2D=screenspace, dc=f_dc, shs=f_rest, colors_precomp=None, opacities=torch.sigmoid(raw_op),
                                 scales=torch.exp(raw_sc), rotations=torch.nn.functional.normalize(raw_rot), cov3D_precomp=None)
    image, radii = out[0].clamp(0, 1), out[1]
    Ll1 = torch.abs(image - gt).mean()
    loss = 0.8 * Ll1 + 0.2 * (1.0 - fused_ssim(image.unsqueeze(0), gt.unsqueeze(0)))
    loss.backward()
    loss.item()
    opt_b.step(radii > 0, radii.shape[0])
    opt_b.zero_grad(set_to_none=True)


out = {}
for name, fn in (("fused_step", step_fused), ("reference_sequence", step_reference_sequence)):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    N = 400
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / N * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    rows = []
    for (fname, line, func), (cc, nc, tt, ct, _callers) in sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:30]:
        rows.append({"calls_per_iter": round(nc / N, 2), "own_us_per_iter": round(tt / N * 1e6, 1), "cum_us_per_iter": round(ct / N * 1e6, 1),
                     "where": f"{os.path.basename(os.path.dirname(fname))}/{os.path.basename(fname)}:{line} {func}"[-100:]})
    out[name] = {"wall_ms_per_iteration": round(wall, 4), "top_by_own_time_under_cprofile": rows}
print(json.dumps(out, indent=0))
