"""Generates tests/golden/*.npz.  Run in the BUILD container (needs /root/reference, read-only):

    python tests/golden/make_golden.py

Two kinds of fixture:
 (1) reference_fragments.npz -- outputs of the REFERENCE's own in-tree Python restatements of hot-path
     sub-steps, imported from /root/reference and run on seeded inputs:
       utils/sh_utils.py:57-112            eval_sh           (SH -> RGB, with gaussian_renderer/__init__.py:77-80 glue)
       utils/general_utils.py:64-110       build_rotation / build_scaling_rotation / strip_symmetric
       scene/gaussian_model.py:33-37       build_covariance_from_scaling_rotation (restated call sequence)
       utils/graphics_utils.py:38-71       getWorld2View2 / getProjectionMatrix
       scene/cameras.py:86-89              world_view_transform / full_proj_transform / camera_center
     These are the only pins the reference offers for this path; the oracle must match them.
     general_utils hard-codes device="cuda"; for generation torch.zeros is wrapped to drop that kwarg.
 (2) oracle_c1.npz -- the frozen oracle's own outputs on BASELINE configs[0] (1k Gaussians, 256x256) so that
     any later change to oracle/ is caught.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
REF = "/root/reference"


def reference_fragments():
    sys.path.insert(0, REF)
    from utils.sh_utils import eval_sh
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix
    import utils.general_utils as gu
    real_zeros = torch.zeros

    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    g = torch.Generator().manual_seed(1234)
    P, Mmax = 257, 16
    out = {}
    # ---- SH ----
    shs = torch.randn(P, Mmax, 3, generator=g) * 0.3
    xyz = torch.randn(P, 3, generator=g) * 3
    campos = torch.tensor([0.3, -0.2, 0.5])
    out["sh_shs"], out["sh_xyz"], out["sh_campos"] = shs.numpy(), xyz.numpy(), campos.numpy()
    for deg in range(4):
        shs_view = shs.transpose(1, 2).view(-1, 3, Mmax)                       # gaussian_renderer/__init__.py:76
        dir_pp = xyz - campos.repeat(P, 1)                                     # :77
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)          # :78
        sh2rgb = eval_sh(deg, shs_view, dir_pp_normalized)                     # :79
        out[f"sh_rgb_deg{deg}"] = torch.clamp_min(sh2rgb + 0.5, 0.0).numpy()   # :80
    # ---- covariance ----
    scales = torch.exp(torch.randn(P, 3, generator=g) * 0.7 - 2.0)
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    torch.zeros = zeros_cpu
    try:
        for mod in (1.0, 1.7):
            L = gu.build_scaling_rotation(mod * scales, rots)                  # scene/gaussian_model.py:34
            actual = L @ L.transpose(1, 2)                                     # :35
            out[f"cov_mod{mod}"] = gu.strip_symmetric(actual).numpy()          # :36
        out["rot_R"] = gu.build_rotation(rots).numpy()
    finally:
        torch.zeros = real_zeros
    out["cov_scales"], out["cov_rots"] = scales.numpy(), rots.numpy()
    # ---- camera ----
    Rm = torch.linalg.qr(torch.randn(3, 3, generator=g))[0].numpy()
    if np.linalg.det(Rm) < 0:
        Rm[:, 0] = -Rm[:, 0]
    T = np.array([0.4, -0.1, 2.5])
    fovx, fovy = math.radians(62.0), math.radians(38.0)
    w2c = getWorld2View2(Rm, T, np.array([0.0, 0.0, 0.0]), 1.0)
    wvt = torch.tensor(w2c).transpose(0, 1)                                    # scene/cameras.py:86
    proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)   # :87
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)                 # :88
    center = wvt.inverse()[3, :3]                                              # :89
    out.update(cam_R=Rm, cam_T=T, cam_fov=np.array([fovx, fovy]), cam_wvt=wvt.numpy(), cam_proj=proj.numpy(),
               cam_full=full.numpy(), cam_center=center.numpy())
    # ---- training loss (the step on the far side of the operator: train.py:119-126, utils/loss_utils.py:40-87) ----
    from utils.loss_utils import ssim as ref_ssim, l1_loss as ref_l1
    img = torch.rand(3, 61, 83, generator=g)
    gt = (img + 0.15 * torch.randn(3, 61, 83, generator=g)).clamp(0, 1)
    out.update(loss_img=img.numpy(), loss_gt=gt.numpy(), loss_ssim=np.float64(ref_ssim(img, gt).item()),
               loss_l1=np.float64(ref_l1(img, gt).item()))
    np.savez_compressed(os.path.join(HERE, "reference_fragments.npz"), **out)
    print("wrote reference_fragments.npz", {k: v.shape for k, v in out.items()})


def oracle_c1():
    from gsr_synth import make_camera, make_scene
    from oracle import torch_oracle as O
    cam = make_camera(256, 256)
    sc = make_scene(1000, cam, seed=0)
    s = O.settings_from_camera(cam, torch.zeros(3))
    with torch.no_grad():
        col, radii, invd, aux = O.rasterize(sc.means3D, None, sc.opacities, s, shs=sc.shs, scales=sc.scales,
                                            rotations=sc.rotations, return_aux=True)
    np.savez_compressed(os.path.join(HERE, "oracle_c1.npz"), color=col.numpy().astype(np.float16),
                        color_sum=np.float64(col.double().sum().item()),
                        color_rowsum=col.double().sum(dim=(0, 2)).numpy(), radii=radii.numpy(),
                        tiles_touched=aux["tiles_touched"].numpy().astype(np.int32), R=np.int64(aux["R"]),
                        point_list=aux["point_list"].numpy().astype(np.int32),
                        ranges=aux["ranges"].numpy().astype(np.int32), n_contrib=aux["n_contrib"].numpy().astype(np.int16),
                        invdepth_rowsum=invd.double().sum(dim=(0, 2)).numpy())
    print("wrote oracle_c1.npz: visible", int((radii > 0).sum()), "R", aux["R"])


if __name__ == "__main__":
    reference_fragments()
    oracle_c1()
