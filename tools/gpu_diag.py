"""GPU diagnostic: localise forward mismatches against the oracle on small scenes (prints, no asserts)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.set_num_threads(8)
from helpers import O, make_camera, make_scene, make_edge_scene, oracle_settings
from test_gpu_parity import run_gpu, run_oracle, mk

for name in ["c1", "edge_lookat"]:
    cam, sc, opts = mk(name)
    s, col, radii, invd, aux = run_oracle(cam, sc, opts)
    vis = radii > 0
    for variant in (0, 1, 2, 3):
        out = run_gpu(s, sc, variant=variant)
        sp = out["splats"].cpu()
        print(f"== {name} variant {variant}: R {out['R']} vs {aux['R']}  radii_eq {torch.equal(out['radii'].cpu(), radii)}")
        if variant == 0:
            for nm, a, b in (("means2D", sp[:, 0:2], aux["means2D"]), ("conic", torch.stack([sp[:, 2], sp[:, 3], sp[:, 4]], 1), aux["conic"]),
                             ("opacity", sp[:, 5], aux["opacity"]), ("rgb", torch.stack([sp[:, 6], sp[:, 7], sp[:, 8]], 1), aux["rgb"]),
                             ("depth", sp[:, 9], aux["depths"])):
                d = (a[vis] - b[vis]).abs().max().item()
                print(f"   splat {nm:8s} max|diff| {d:.3e}  exact {torch.equal(a[vis], b[vis])}")
            print("   point_list eq", torch.equal(out["point_list"].cpu().long(), aux["point_list"]),
                  " ranges eq", torch.equal(out["ranges"].cpu().long(), aux["ranges"]))
        err = (out["color"].cpu() - col).abs().amax(0)
        nc_g, nc_o = out["n_contrib"].cpu().long(), aux["n_contrib"]
        print(f"   image max err {err.max().item():.3e}  frac>1e-5 {(err > 1e-5).float().mean().item():.4f}  "
              f"n_contrib mismatch {(nc_g != nc_o).float().mean().item():.4f}  final_T err {(out['final_T'].cpu() - aux['final_T']).abs().max().item():.3e}"
              f"  invd err {(out['invdepth'].cpu() - invd).abs().max().item():.3e} fragile {aux['fragile'].float().mean().item():.4f}")
        if err.max() > 1e-4:
            y, x = divmod(int(err.argmax()), err.shape[1])
            print(f"   worst px ({x},{y}): gpu {out['color'][:, y, x].cpu().tolist()} oracle {col[:, y, x].tolist()} "
                  f"n_contrib gpu {int(nc_g[y, x])} oracle {int(nc_o[y, x])} T gpu {out['final_T'][y, x].item():.5f} oracle {aux['final_T'][y, x].item():.5f}")
            bad = (err > 1e-4)
            ys, xs = torch.nonzero(bad, as_tuple=True)
            print("   bad px bbox x", int(xs.min()), int(xs.max()), "y", int(ys.min()), int(ys.max()), " count", int(bad.sum()),
                  " bad per 8x8-quadrant-pos:", [(int((bad[(ys0)::16][:, (xs0)::16]).sum())) for ys0 in (0, 8) for xs0 in (0, 8)])
