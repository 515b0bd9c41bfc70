"""CPU model of the forward blend's work under a finer cull: how many (8x8 block, list entry) steps does the wave kernel of
csrc/render_fwd.hip take when the box test of every batch of 64 entries uses the bounding box of the block's LIVE pixels
(transmittance not yet below 1e-4) instead of the whole 8x8 block?  The cull stays exact: an entry dropped this way has
alpha < 1/255 at every live pixel, and terminated pixels take nothing.

    python tools/cpu_live_box_model.py [P] [W] [H] [tiles sampled] [kind: uniform|clustered]

Test infrastructure (imports oracle/): analysis only, never on the product path."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import O, make_camera, make_scene, make_clustered_scene, oracle_settings

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
NT = int(sys.argv[4]) if len(sys.argv) > 4 else 120
KIND = sys.argv[5] if len(sys.argv) > 5 else "uniform"
CHECK = 8

cam = make_camera(W, H)
sc = make_clustered_scene(P, cam, seed=0) if KIND == "clustered" else make_scene(P, cam, seed=0, s_med=0.012)
s = oracle_settings(cam)
with torch.no_grad():
    pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    bins = O.bin_and_sort(pre)
gx, gy = pre["grid"]
xy = pre["means2D"].numpy().astype(np.float32)
con = pre["conic"].numpy().astype(np.float32)
op = pre["opacity"].numpy().astype(np.float32)
tau = O.tau_of_opacity(pre["opacity"]).numpy()
ranges = bins["ranges"].numpy()
plist = bins["point_list"].numpy()


def min_q(mx, my, A, B, C, x0, x1, y0, y1):
    """csrc/render_fwd.hip min_q_over_box, vectorised over entries."""
    lx, hx, ly, hy = x0 - mx, x1 - mx, y0 - my, y1 - my
    in_x = (lx <= 0) & (hx >= 0)
    in_y = (ly <= 0) & (hy >= 0)
    q = np.full(mx.shape, 3.0e38, np.float32)
    dx = np.where(lx > 0, lx, hx)
    dy = np.minimum(hy, np.maximum(ly, -B * dx / C))
    qa = A * dx * dx + 2 * B * dx * dy + C * dy * dy
    q = np.where(~in_x, np.minimum(q, qa), q)
    dy2 = np.where(ly > 0, ly, hy)
    dx2 = np.minimum(hx, np.maximum(lx, -B * dy2 / A))
    qb = A * dx2 * dx2 + 2 * B * dx2 * dy2 + C * dy2 * dy2
    q = np.where(~in_y, np.minimum(q, qb), q)
    return np.where(in_x & in_y, 0.0, q)


def walk(bx0, by0, ids, mode):
    """mode 0: whole block (the product); 1: live box at every batch start; 2: live box re-tested at every termination check."""
    px = (bx0 + np.arange(64) % 8).astype(np.float32)
    py = (by0 + np.arange(64) // 8).astype(np.float32)
    inside = (px < W) & (py < H)
    T = np.where(inside, 1.0, 0.0).astype(np.float32)
    steps = tests = retests = 0
    x0, x1 = float(bx0), float(min(bx0 + 7, W - 1))
    y0, y1 = float(by0), float(min(by0 + 7, H - 1))
    for b in range(0, len(ids), 64):
        g = ids[b:b + 64]
        live = T != 0
        if not live.any():
            break
        if mode:
            x0, x1, y0, y1 = px[live].min(), px[live].max(), py[live].min(), py[live].max()
        mx, my, A, B, C = xy[g, 0], xy[g, 1], con[g, 0], con[g, 1], con[g, 2]
        keep = ~(min_q(mx, my, A, B, C, x0, x1, y0, y1) > tau[g])
        tests += 1
        surv = list(np.nonzero(keep)[0])
        k = 0
        while surv:
            j = surv.pop(0)
            steps += 1
            k += 1
            dx, dy = mx[j] - px, my[j] - py
            power = -0.5 * (A[j] * dx * dx + C[j] * dy * dy) - B[j] * dx * dy
            alpha = np.minimum(0.99, op[g[j]] * np.exp(power))
            valid = (power <= 0) & (alpha >= 1.0 / 255.0)
            testT = T * (1 - alpha)
            term = valid & (testT < 1e-4)
            T = np.where(valid & ~term, testT, np.where(term, 0.0, T)).astype(np.float32)
            if k % CHECK == 0 or not surv:
                live = T != 0
                if not live.any():
                    return steps, tests, retests
                if mode == 2 and surv:
                    nx0, nx1, ny0, ny1 = px[live].min(), px[live].max(), py[live].min(), py[live].max()
                    if (nx0, nx1, ny0, ny1) != (x0, x1, y0, y1):
                        x0, x1, y0, y1 = nx0, nx1, ny0, ny1
                        sj = np.array(surv)
                        kk = ~(min_q(mx[sj], my[sj], A[sj], B[sj], C[sj], x0, x1, y0, y1) > tau[g[sj]])
                        surv = list(sj[kk])
                        retests += 1
    return steps, tests, retests


rng = np.random.default_rng(0)
tiles = rng.choice(gx * gy, size=min(NT, gx * gy), replace=False)
tot = {m: np.zeros(3, np.int64) for m in (0, 1, 2)}
for t in tiles:
    ty, tx = divmod(int(t), gx)
    ids = plist[ranges[t, 0]:ranges[t, 1]]
    for quad in range(4):
        bx0, by0 = tx * 16 + (quad & 1) * 8, ty * 16 + (quad >> 1) * 8
        if bx0 >= W or by0 >= H:
            continue
        for m in (0, 1, 2):
            tot[m] += np.array(walk(bx0, by0, ids, m))
out = {"P": P, "W": W, "H": H, "kind": KIND, "tiles_sampled": int(len(tiles)), "R": int(bins["R"]),
       "whole_block": {"steps": int(tot[0][0]), "batches": int(tot[0][1])},
       "live_box_per_batch": {"steps": int(tot[1][0]), "batches": int(tot[1][1]), "steps_ratio": round(tot[1][0] / tot[0][0], 4)},
       "live_box_per_check": {"steps": int(tot[2][0]), "batches": int(tot[2][1]), "retests": int(tot[2][2]),
                              "steps_ratio": round(tot[2][0] / tot[0][0], 4),
                              "steps_plus_retests_ratio": round((tot[2][0] + tot[2][2]) / tot[0][0], 4)}}
print(json.dumps(out))
