"""Where does a density-control step (gsr_scene.densify.densify_and_prune, every 100 iterations) spend its time?  Wall-clock per
phase with synchronisation, allocator statistics before / after; 1 M Gaussians, 3 % cloned / split per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch, torch.nn as nn
from gsr_synth import make_camera, make_scene
from gsr_scene.densify import DensifyStats, densify_and_prune
from diff_gaussian_rasterization import SparseGaussianAdam
dev = torch.device("cuda:0")
cam = make_camera(1920, 1080)
sc = make_scene(1_000_000, cam, seed=0).to(dev)
par = lambda t: nn.Parameter(t.detach().clone().contiguous().requires_grad_(True))
params = {"xyz": par(sc.means3D), "f_dc": par(sc.shs[:, :1]), "f_rest": par(sc.shs[:, 1:]), "opacity": par(torch.logit(sc.opacities.clamp(1e-6, 1 - 1e-6))),
          "scaling": par(torch.log(sc.scales)), "rotation": par(sc.rotations)}
opt = SparseGaussianAdam([{"params": [params[k]], "lr": 1e-5, "name": k} for k in params], lr=1e-5, eps=1e-15)
for p in params.values():
    p.grad = torch.zeros_like(p)
P = params["xyz"].shape[0]
opt.step(torch.ones(P, dtype=torch.bool, device=dev), P)      # creates the moments
opt.zero_grad(set_to_none=True)
stats = DensifyStats.zeros(P, dev)
for it in range(6):
    n = params["xyz"].shape[0]
    stats.xyz_gradient_accum.copy_(torch.rand(n, 1, device=dev)); stats.denom.fill_(1.0)
    radii = torch.randint(1, 30, (n,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    g = (stats.xyz_gradient_accum / stats.denom.clamp_min(1))[stats.denom > 0]
    thr = float(g.kthvalue(int(0.97 * g.numel())).values)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    params, stats, _ = densify_and_prune(opt, stats, max_grad=thr, min_opacity=0.005, extent=4.0, max_screen_size=None, radii=radii)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ms1 = torch.cuda.memory_stats()
    print(f"step {it}: P {n} -> {params['xyz'].shape[0]}  threshold {1e3*(t1-t0):.2f} ms  densify_and_prune {1e3*(t2-t1):.2f} ms  "
          f"device mallocs {ms1['num_device_alloc']-ms0['num_device_alloc']}  frees {ms1['num_device_free']-ms0['num_device_free']}  "
          f"reserved {ms1['reserved_bytes.all.current']/1e9:.2f} GB allocated {ms1['allocated_bytes.all.current']/1e9:.2f} GB", flush=True)
