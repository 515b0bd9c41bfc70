"""Lane-efficiency statistics of the blend kernels on the bench frame (GPU, torch ops on the library's own binning output).

For a sample of tiles it counts, for the FORWARD walk (entries up to the point where every pixel of the unit has
terminated) and the BACKWARD walk (entries before the unit's last contributor):
  * survivors of the exact box test at 8x8 (what the kernels use) and at 4x4 granularity,
  * the lane efficiency of the 8x8 design  = contributing (pixel, entry) pairs / (64 x surviving (8x8, entry) pairs),
  * the step count of a design in which the four 16-lane rows of a wave walk their own 4x4 sub-box survivor lists
    independently (steps = max over the four rows), against the steps of the current design (= 8x8 survivors).
Usage: python tools/gpu_lane_stats.py [s_med] [n_tiles]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings
from diff_gaussian_rasterization.debug import forward_with_views

dev = torch.device("cuda:0")
W, H, P = 1920, 1080, 1_000_000
s_med = float(sys.argv[1]) if len(sys.argv) > 1 else 0.012
n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cam = make_camera(W, H)
sc = make_scene(P, cam, seed=0, s_med=s_med).to(dev)
camd = cam.to(dev)
rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                   camd.full_proj_transform, 3, camd.camera_center, False, False, False)
o = forward_with_views(rs, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
gx, gy = (W + 15) // 16, (H + 15) // 16
rng = o["ranges"].long()
pl = o["point_list"].long()
splats = o["splats"]
ncon = o["n_contrib"].long()
fT = o["final_T"]


def min_q_box(mx, my, A, B, C, x0, x1, y0, y1):
    lx, hx, ly, hy = x0 - mx, x1 - mx, y0 - my, y1 - my
    in_x = (lx <= 0) & (hx >= 0)
    in_y = (ly <= 0) & (hy >= 0)
    dxe = torch.where(lx > 0, lx, hx)
    dye = torch.minimum(hy, torch.maximum(ly, -B * dxe / C))
    q1 = A * dxe * dxe + 2 * B * dxe * dye + C * dye * dye
    dye2 = torch.where(ly > 0, ly, hy)
    dxe2 = torch.minimum(hx, torch.maximum(lx, -B * dye2 / A))
    q2 = A * dxe2 * dxe2 + 2 * B * dxe2 * dye2 + C * dye2 * dye2
    big = torch.full_like(q1, 3e38)
    q = torch.minimum(torch.where(in_x, big, q1), torch.where(in_y, big, q2))
    return torch.where(in_x & in_y, torch.zeros_like(q), q)


tiles = torch.linspace(0, gx * gy - 1, n_sample).long().tolist()
acc = {k: 0 for k in ("fwd_s8", "fwd_rowmax", "fwd_s4sum", "fwd_pairs", "bwd_s8", "bwd_rowmax", "bwd_s4sum", "bwd_pairs",
                      "fwd_live8", "bwd_live8", "bwd_touched_inst", "bwd_quad_records")}
for t in tiles:
    ty, tx = divmod(t, gx)
    a, b = int(rng[t, 0]), int(rng[t, 1])
    if b <= a:
        continue
    sp = splats[pl[a:b]]
    N = sp.shape[0]
    mx, my, A, B, C, op, tau = sp[:, 0], sp[:, 1], sp[:, 2], sp[:, 3], sp[:, 4], sp[:, 5], sp[:, 10]
    ys = torch.arange(ty * 16, ty * 16 + 16, device=dev)
    xs = torch.arange(tx * 16, tx * 16 + 16, device=dev)
    py, px = torch.meshgrid(ys, xs, indexing="ij")                 # [16,16]
    inside = (py < H) & (px < W)
    dx = mx[None, None, :] - px[:, :, None].float()
    dy = my[None, None, :] - py[:, :, None].float()
    power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
    alpha = torch.clamp(op * torch.exp(power), max=0.99)
    valid = (power <= 0) & (alpha >= 1.0 / 255.0) & inside[:, :, None]
    nc = torch.zeros(16, 16, dtype=torch.long, device=dev)
    nc[inside] = ncon[py[inside], px[inside]]
    pos = torch.arange(N, device=dev)
    contrib = valid & (pos[None, None, :] < nc[:, :, None])        # backward's active (pixel, entry) pairs
    # forward: a pixel evaluates entries until it terminates (T (1 - alpha) < 1e-4 on a valid entry); p_term = position of
    # the terminating entry, N if the pixel never terminates (then its wave walks the whole list)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha)).double()
    Tincl = torch.cumprod(1.0 - a_eff, dim=2)
    term = valid & (Tincl < 1e-4)
    has_term = term.any(dim=2)
    p_term = torch.where(has_term, torch.argmax(term.to(torch.int8), dim=2), torch.full_like(nc, N))
    p_term = torch.where(inside, p_term, torch.zeros_like(p_term) - 1)      # pixels outside the image never hold the wave
    fwd_pair = valid & (pos[None, None, :] < p_term[:, :, None])
    touched_any = torch.zeros(N, dtype=torch.bool, device=dev)
    surv_b = [None] * 4
    surv_f = [None] * 4
    for q in range(4):
        qy, qx = (q >> 1) * 8, (q & 1) * 8
        bx0, by0 = tx * 16 + qx, ty * 16 + qy
        if bx0 >= W or by0 >= H:
            continue
        x1 = float(min(bx0 + 7, W - 1)); y1 = float(min(by0 + 7, H - 1))
        keep8 = ~(min_q_box(mx, my, A, B, C, float(bx0), x1, float(by0), y1) > tau)
        ncq = nc[qy:qy + 8, qx:qx + 8]
        mxq = int(ncq.max())
        # forward walk depth of the quadrant: until every pixel has met its terminating entry (batches of 64 are ignored)
        fdepth = min(N, int(p_term[qy:qy + 8, qx:qx + 8].max()) + 1)
        live_f = pos < fdepth
        live_b = pos < mxq
        acc["fwd_live8"] += int(live_f.sum()); acc["bwd_live8"] += int(live_b.sum())
        s8f = keep8 & live_f
        s8b = keep8 & live_b
        acc["fwd_s8"] += int(s8f.sum()); acc["bwd_s8"] += int(s8b.sum())
        surv_b[q], surv_f[q] = s8b, s8f
        cq = contrib[qy:qy + 8, qx:qx + 8]
        acc["bwd_pairs"] += int(cq.sum())
        acc["fwd_pairs"] += int(fwd_pair[qy:qy + 8, qx:qx + 8].sum())
        touched_q = cq.any(dim=0).any(dim=0) & s8b
        acc["bwd_quad_records"] += int(touched_q.sum())
        touched_any |= touched_q
        rowmax_f = 0; rowmax_b = 0
        for r in range(4):
            ry, rx = qy + (r >> 1) * 4, qx + (r & 1) * 4
            sx0, sy0 = tx * 16 + rx, ty * 16 + ry
            if sx0 >= W or sy0 >= H:
                continue
            keep4 = ~(min_q_box(mx, my, A, B, C, float(sx0), float(min(sx0 + 3, W - 1)), float(sy0), float(min(sy0 + 3, H - 1))) > tau)
            ncr = nc[ry:ry + 4, rx:rx + 4]
            mxr = int(ncr.max())
            fd = min(N, int(p_term[ry:ry + 4, rx:rx + 4].max()) + 1)
            n4f = int((keep4 & (pos < fd)).sum()); n4b = int((keep4 & (pos < mxr)).sum())
            acc["fwd_s4sum"] += n4f; acc["bwd_s4sum"] += n4b
            rowmax_f = max(rowmax_f, n4f); rowmax_b = max(rowmax_b, n4b)
        acc["fwd_rowmax"] += rowmax_f; acc["bwd_rowmax"] += rowmax_b
    acc["bwd_touched_inst"] += int(touched_any.sum())
    # two quadrants per wave (2 pixels per lane): steps = union of the two survivor sets
    for (qa, qb), key in (((0, 1), "h"), ((2, 3), "h"), ((0, 2), "v"), ((1, 3), "v")):
        if surv_b[qa] is None or surv_b[qb] is None:
            continue
        acc["bwd_pair_" + key] = acc.get("bwd_pair_" + key, 0) + int((surv_b[qa] | surv_b[qb]).sum())
        acc["bwd_pairsum_" + key] = acc.get("bwd_pairsum_" + key, 0) + int(surv_b[qa].sum()) + int(surv_b[qb].sum())
        acc["fwd_pair_" + key] = acc.get("fwd_pair_" + key, 0) + int((surv_f[qa] | surv_f[qb]).sum())
        acc["fwd_pairsum_" + key] = acc.get("fwd_pairsum_" + key, 0) + int(surv_f[qa].sum()) + int(surv_f[qb].sum())

f = lambda a, b: f"{a / max(1, b):.3f}"
print(f"s_med {s_med}, {len(tiles)} tiles sampled")
print(f"FWD: live (8x8,entry) {acc['fwd_live8']}  survivors8 {acc['fwd_s8']} ({f(acc['fwd_s8'], acc['fwd_live8'])})  lane efficiency "
      f"{f(acc['fwd_pairs'], 64 * acc['fwd_s8'])}  row-independent steps (max of 4 rows) {acc['fwd_rowmax']} = {f(acc['fwd_rowmax'], acc['fwd_s8'])} x current; "
      f"sum of 4x4 survivors {acc['fwd_s4sum']} ({f(acc['fwd_s4sum'], 4 * acc['fwd_s8'])} of 4 x s8)")
for d in ("fwd", "bwd"):
    for key in ("h", "v"):
        print(f"{d.upper()} two quadrants per wave ({'16x8' if key == 'h' else '8x16'}): union / sum of survivor steps = "
              f"{f(acc.get(d + '_pair_' + key, 0), acc.get(d + '_pairsum_' + key, 0))}")
print(f"BWD: live (8x8,entry) {acc['bwd_live8']}  survivors8 {acc['bwd_s8']} ({f(acc['bwd_s8'], acc['bwd_live8'])})  lane efficiency "
      f"{f(acc['bwd_pairs'], 64 * acc['bwd_s8'])}  row-independent steps {acc['bwd_rowmax']} = {f(acc['bwd_rowmax'], acc['bwd_s8'])} x current; "
      f"sum of 4x4 survivors {acc['bwd_s4sum']} ({f(acc['bwd_s4sum'], 4 * acc['bwd_s8'])} of 4 x s8); quadrant records {acc['bwd_quad_records']} "
      f"touched instances {acc['bwd_touched_inst']} ({f(acc['bwd_quad_records'], acc['bwd_touched_inst'])} records per touched instance)")
