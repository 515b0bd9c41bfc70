"""Preprocess stage time with and without the SH phase (stage timers), 1 M Gaussians @1080p."""
import os, sys
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
cols = torch.rand(P, 3, device=dev)
_lib.load()
def run(tag, **kw):
    for _ in range(5):
        with torch.no_grad():
            rasterize_gaussians(sc.means3D, None, kw.get("sh"), kw.get("col"), sc.opacities, sc.scales, sc.rotations, None, rs, None)
    _lib.profile_reset(); _lib.profile_enable(True)
    for _ in range(20):
        with torch.no_grad():
            rasterize_gaussians(sc.means3D, None, kw.get("sh"), kw.get("col"), sc.opacities, sc.scales, sc.rotations, None, rs, None)
    torch.cuda.synchronize()
    st = _lib.profile_read(); _lib.profile_enable(False)
    print(tag, "preprocess %.4f ms" % (st["preprocess"]["ms"] / st["preprocess"]["launches"]))
run("sh degree 3 ", sh=sc.shs)
run("colors_precomp", col=cols)
