"""GPU experiment driver: forward stage times under different library options (prints one line per config)."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib, rasterize_gaussians

dev = torch.device("cuda:0")
W, H, P = 1920, 1080, 1_000_000
cam = make_camera(W, H); sc = make_scene(P, cam, seed=0).to(dev); camd = cam.to(dev)
rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                   camd.full_proj_transform, 3, camd.camera_center, False, False, False)
_lib.load()

def fwd():
    with torch.no_grad():
        return rasterize_gaussians(sc.means3D, None, sc.shs, None, sc.opacities, sc.scales, sc.rotations, None, rs, None)

def measure(tag, steps=30):
    for _ in range(5): fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fwd()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    _lib.profile_reset(); _lib.profile_enable(True)
    for _ in range(steps): fwd()
    torch.cuda.synchronize(); st = _lib.profile_read(); _lib.profile_enable(False)
    print(tag, f"{ms:.4f} ms", {k: round(v["ms"] / max(1, v["launches"]), 4) for k, v in st.items() if v["launches"]}, flush=True)

configs = [a.split("=") for a in sys.argv[1:]] or [["base", ""]]
measure("default")
for name, val in configs:
    if name == "base": continue
    _lib.set_option(name, int(val)); measure(f"{name}={val}")
