"""Invariant checks at the large BASELINE configs (stand-ins): 6 M Gaussians @1080p (configs[4]) and 1 M @4K (configs[3]),
plus an empty tile band.  Prints timings; asserts structural invariants (no oracle at these sizes)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib, rasterize_gaussians
from diff_gaussian_rasterization.debug import forward_with_views
dev = torch.device("cuda:0")

def run(P, W, H, s_med=0.012):
    cam = make_camera(W, H); sc = make_scene(P, cam, seed=0, s_med=s_med).to(dev); camd = cam.to(dev)
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                       camd.full_proj_transform, 3, camd.camera_center, False, False, False)
    o = forward_with_views(rs, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    R = o["R"]; tt = o["tiles_touched"].long(); rng = o["ranges"].long(); cnt = rng[:, 1] - rng[:, 0]
    assert int(tt.sum()) == R and int(cnt.sum()) == R
    pl = o["point_list"].long(); assert int(pl.max()) < P
    assert torch.equal(torch.bincount(pl, minlength=P), tt)
    d = o["splats"][:, 9][pl]; tile_of = torch.repeat_interleave(torch.arange(rng.shape[0], device=dev), cnt)
    same = tile_of[1:] == tile_of[:-1]
    assert bool((d[1:][same] >= d[:-1][same]).all())
    assert torch.isfinite(o["color"]).all() and float(o["final_T"].min()) >= 0
    del o, pl, d, tile_of, same
    def fwd():
        with torch.no_grad():
            return rasterize_gaussians(sc.means3D, None, sc.shs, None, sc.opacities, sc.scales, sc.rotations, None, rs, None)
    for _ in range(3): fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fwd()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    # training-shaped step at scale: forward + backward (memory + indices), timed warm
    params = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]

    def fb():
        for p in params:
            p.grad = None
        col, radii, invd = rasterize_gaussians(params[0], None, params[1], None, params[2], params[3], params[4], None, rs, None)
        col.mean().backward()
        return radii
    for _ in range(2):
        radii = fb()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        radii = fb()
    torch.cuda.synchronize(); bms = (time.perf_counter() - t0) / 5 * 1e3
    assert all(torch.isfinite(p.grad).all() for p in params)
    # empty band renders nothing but keeps radii
    c2, r2, _ = rasterize_gaussians(sc.means3D, None, sc.shs, None, sc.opacities, sc.scales, sc.rotations, None, rs, (0, 0))
    assert float(c2.abs().max()) == 0.0 and torch.equal(r2, radii)
    print(f"P={P} {W}x{H}: V={int((radii>0).sum())} R={R}  forward {ms:.3f} ms = {W*H/ms/1e3:.0f} Mpix/s   forward+backward {bms:.2f} ms  "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)

run(6_000_000, 1920, 1080)
run(1_000_000, 3840, 2160)
run(2_500_000, 3840, 2160)
