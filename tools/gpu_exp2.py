"""GPU experiment driver for the TRAIN step: stage times under different options."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib, rasterize_gaussians
dev = torch.device("cuda:0")
W, H, P = 1920, 1080, 1_000_000
s_med = 0.012
cam = make_camera(W, H); sc = make_scene(P, cam, seed=0, s_med=s_med).to(dev); camd = cam.to(dev)
rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                   camd.full_proj_transform, 3, camd.camera_center, False, False, False)
_lib.load()
params = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
gt = torch.rand(3, H, W, device=dev)
def step():
    for p in params: p.grad = None
    m, sh, o, s_, r_ = params
    color, radii, invd = rasterize_gaussians(m, None, sh, None, o, s_, r_, None, rs, None)
    (color - gt).abs().mean().backward()
def measure(tag, steps=15):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    _lib.profile_reset(); _lib.profile_enable(True)
    for _ in range(steps): step()
    torch.cuda.synchronize(); st = _lib.profile_read(); _lib.profile_enable(False)
    print(tag, f"fwd+L1+bwd {ms:.4f} ms", {k: round(v["ms"] / max(1, v["launches"]), 4) for k, v in st.items() if v["launches"] and "bwd" in k}, flush=True)
measure("default")
for a in sys.argv[1:]:
    name, val = a.split("="); _lib.set_option(name, int(val)); measure(a)
