"""Workload statistics of the blend on the bench frame (GPU, torch ops on the library's own binning output):
survivor fraction of the 8x8 box test, batches, early-termination depth."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib
from diff_gaussian_rasterization.debug import forward_with_views
dev = torch.device("cuda:0")
W, H, P = 1920, 1080, 1_000_000
s_med = float(sys.argv[1]) if len(sys.argv) > 1 else 0.012
cam = make_camera(W, H); sc = make_scene(P, cam, seed=0, s_med=s_med).to(dev); camd = cam.to(dev)
rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                   camd.full_proj_transform, 3, camd.camera_center, False, False, False)
o = forward_with_views(rs, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
R = o["R"]; gx = (W + 15) // 16
rng = o["ranges"].long(); cnt = rng[:, 1] - rng[:, 0]
tile_of = torch.repeat_interleave(torch.arange(rng.shape[0], device=dev), cnt)
pos_in_tile = torch.arange(R, device=dev) - rng[tile_of, 0]
pl = o["point_list"].long(); sp = o["splats"][pl]
mx, my, A, B, C, tau = sp[:, 0], sp[:, 1], sp[:, 2], sp[:, 3], sp[:, 4], sp[:, 10]
tx, ty = (tile_of % gx).float() * 16, (tile_of // gx).float() * 16
nc = o["n_contrib"].long()
surv_tot = 0; surv_live = 0; live_tot = 0
for q in range(4):
    x0 = tx + (q & 1) * 8; y0 = ty + (q >> 1) * 8
    x1 = torch.minimum(x0 + 7, torch.tensor(W - 1.0, device=dev)); y1 = torch.minimum(y0 + 7, torch.tensor(H - 1.0, device=dev))
    lx, hx, ly, hy = x0 - mx, x1 - mx, y0 - my, y1 - my
    in_x = (lx <= 0) & (hx >= 0); in_y = (ly <= 0) & (hy >= 0)
    dxe = torch.where(lx > 0, lx, hx); dye = torch.clamp(-B * dxe / C, ly, hy)
    q1 = A * dxe * dxe + 2 * B * dxe * dye + C * dye * dye
    dye2 = torch.where(ly > 0, ly, hy); dxe2 = torch.clamp(-B * dye2 / A, lx, hx)
    q2 = A * dxe2 * dxe2 + 2 * B * dxe2 * dye2 + C * dye2 * dye2
    big = torch.full_like(q1, 3e38)
    qmin = torch.minimum(torch.where(in_x, big, q1), torch.where(in_y, big, q2))
    qmin = torch.where(in_x & in_y, torch.zeros_like(qmin), qmin)
    keep = ~(qmin > tau)
    # how deep does this quadrant's wave walk its list?  until all 64 pixels terminated ~ max n_contrib over the block (+ a batch)
    ncq = nc.view(H // 1 if False else H, W)
    blk_x = (tx.long() + (q & 1) * 8); blk_y = (ty.long() + (q >> 1) * 8)
    # per-tile-quadrant max n_contrib
    ncpad = torch.zeros(((H + 15) // 16) * 16, gx * 16, device=dev, dtype=torch.long); ncpad[:H, :W] = nc
    qmax = ncpad.view((H + 15) // 16, 2, 8, gx, 2, 8)[:, q >> 1, :, :, q & 1, :].amax(dim=(1, 3)).reshape(-1)   # [tiles]
    live = pos_in_tile < qmax[tile_of]
    surv_tot += int(keep.sum()); surv_live += int((keep & live).sum()); live_tot += int(live.sum())
print(f"s_med {s_med}: V {int((o['radii']>0).sum())} R {R}  instance-quadrant pairs {4*R}  box-test survivors {surv_tot} ({surv_tot/(4*R):.3f})"
      f"  within max n_contrib: entries {live_tot} ({live_tot/(4*R):.3f}) survivors {surv_live} ({surv_live/(4*R):.3f})"
      f"  mean n_contrib {nc.float().mean().item():.1f}  mean list {cnt.float().mean().item():.1f}")
