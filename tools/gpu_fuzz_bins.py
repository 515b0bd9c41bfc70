"""Randomised cross-check of the binning chain on the GPU (round 4: the bucket depth sort, the frame statistics and the fused
level-2 scan are new code): N random frames -- size, splat scale, frame shape, depth distribution (plain / quantised ties / a
crowd inside a few ulps / two clusters decades apart / everything behind one key), rectangle mode -- each rendered with the bucket
depth sort (`depth_sort_mode=2`) and with the LSD radix sort + scan kernels (`=1`); depth order, tile scan, R, sorted point
list, tile ranges and the image must be bit-identical.  One frame in four is also compared with the CPU oracle's bins.
    python tools/gpu_fuzz_bins.py [frames] [seed]         -> one JSON summary line (also gpurun_out/fuzz_bins.json)
    FUZZ_BIG=1: P between 100 K and 2.8 M (the bucket tables' second sweep, the 2048-key LSD tier)
Test infrastructure (imports oracle/ through tests/helpers.py), not product code."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import O, make_camera, make_scene, oracle_settings, reference_tiles
from test_gpu_parity import gpu_settings
from diff_gaussian_rasterization import _lib
from diff_gaussian_rasterization.debug import forward_with_views

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(SEED)
dev = torch.device("cuda:0")
stats = {"frames": 0, "oracle_checked": 0, "kinds": {}, "max_P": 0, "max_R": 0, "failures": []}
t0 = time.time()
for it in range(N):
    P = int(10 ** (rng.uniform(5.0, 6.45) if os.environ.get("FUZZ_BIG") else rng.uniform(0.5, 5.6)))
    W, H = rng.choice([(64, 48), (200, 120), (320, 240), (641, 359), (1280, 720), (1920, 1080)])
    s_med = 10 ** rng.uniform(-2.6, -0.7) * (0.3 if P > 100_000 else 1.0)
    kind = rng.choice(["plain", "plain", "ties", "crowd", "gap", "one_key", "outliers", "outliers"])
    snug = rng.choice([1, 1, 0])
    cam = make_camera(W, H)
    sc = make_scene(P, cam, seed=1000 + it, s_med=s_med)
    g = torch.Generator().manual_seed(it)
    z = sc.means3D[:, 2].clone()
    znew = None
    if kind == "ties":
        znew = 1.5 + torch.randint(0, rng.choice([3, 64, 1000]), (P,), generator=g).float() * 0.0625
    elif kind == "crowd":
        znew = z.clone()
        crowd = torch.rand(P, generator=g) < rng.uniform(0.3, 0.95)
        znew[crowd] = 5.0 + torch.randint(0, rng.choice([2, 48, 4000]), (int(crowd.sum()),), generator=g).float() * 4.76837158203125e-07
    elif kind == "gap":
        near = torch.rand(P, generator=g) < 0.5
        znew = torch.where(near, 0.25 + 0.05 * torch.rand(P, generator=g), 2000.0 + 6000.0 * torch.rand(P, generator=g))
    elif kind == "one_key":
        znew = torch.full_like(z, rng.choice([0.3, 4.0, 900.0]))
    elif kind == "outliers":      # round 5: a narrow bulk + a few floaters far behind / in front of it (the robust key range of ds_hist)
        lo = rng.choice([0.5, 4.0, 60.0])
        znew = lo * (1.0 + rng.choice([0.01, 0.05, 0.5]) * torch.rand(P, generator=g))
        n_far, n_near = rng.choice([1, 7, 40, 400]), rng.choice([0, 3, 30])
        znew[torch.randperm(P, generator=g)[:min(P, n_far)]] = lo * (20.0 + 400.0 * torch.rand(min(P, n_far), generator=g))
        if n_near:
            znew[torch.randperm(P, generator=g)[:min(P, n_near)]] = torch.clamp(lo * (0.02 + 0.3 * torch.rand(min(P, n_near), generator=g)), min=0.21)
    if znew is not None:
        f = (znew / z).unsqueeze(1)
        sc.means3D.mul_(f)
        if kind in ("gap", "outliers"):
            sc.scales.mul_(f)
    s = oracle_settings(cam)
    rs = gpu_settings(s, dev)
    scd = sc.to(dev)
    outs = {}
    _lib.set_option("snug_tiles", snug)
    try:
        for mode in (2, 1):
            _lib.set_option("depth_sort_mode", mode)
            o = forward_with_views(rs, scd.means3D, scd.opacities, shs=scd.shs, scales=scd.scales, rotations=scd.rotations,
                                   no_backward=bool(it & 1))
            outs[mode] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items() if k != "buffers"}
        a, b = outs[2], outs[1]
        bad = [k for k in ("R", "radii", "ranges", "point_list", "depth_order", "offsets", "color") if k in a and k in b and
               not (a[k] == b[k] if not torch.is_tensor(a[k]) else torch.equal(a[k], b[k]))]
        R = int(a["R"])
        if not bad and it % 4 == 0 and P <= 120_000 and R <= 6_000_000:
            if snug:
                pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
                bins = O.bin_and_sort(pre)
            else:
                with reference_tiles():
                    pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
                    bins = O.bin_and_sort(pre)
            if R != bins["R"] or not torch.equal(a["point_list"].cpu().long(), bins["point_list"]) or \
               not torch.equal(a["ranges"].cpu().long(), bins["ranges"]):
                bad.append("oracle_bins")
            stats["oracle_checked"] += 1
    finally:
        _lib.set_option("depth_sort_mode", 0)
        _lib.set_option("snug_tiles", 1)
    stats["frames"] += 1
    stats["kinds"][kind] = stats["kinds"].get(kind, 0) + 1
    stats["max_P"], stats["max_R"] = max(stats["max_P"], P), max(stats["max_R"], R)
    if bad:
        stats["failures"].append({"frame": it, "P": P, "W": W, "H": H, "s_med": s_med, "kind": kind, "snug": snug, "differs": bad})
    del scd, outs
stats["seconds"] = round(time.time() - t0, 1)
stats["seed"] = SEED
print(json.dumps(stats))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(stats, open(os.path.join(ROOT, "gpurun_out", "fuzz_bins_big.json" if os.environ.get("FUZZ_BIG") else "fuzz_bins.json"), "w"), indent=1)
sys.exit(1 if stats["failures"] else 0)
