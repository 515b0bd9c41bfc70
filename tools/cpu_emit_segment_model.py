"""CPU count behind DESIGN 9, candidate (1): emit_scatter (csrc/tilesort.hip) ranks every (Gaussian, tile) INSTANCE against the other instances
of its 4096-instance emission block with ballot matching (~127 VALU per 64 instances).  The instances of one (Gaussian, tile row) are
consecutive in emission order AND land in one level-1 bucket next to each other, so a kernel could rank the row SEGMENTS and expand them
afterwards.  How many segments are there, how long are they, and how many fall into one emission block?

    python tools/cpu_emit_segment_model.py [P] [W] [H] [kind: uniform|clustered]      -> one JSON line

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
KIND = sys.argv[4] if len(sys.argv) > 4 else "uniform"
cam = make_camera(W, H)
sc = make_clustered_scene(P, cam, seed=0) if KIND == "clustered" else make_scene(P, cam, seed=0, s_med=0.012)
s = oracle_settings(cam)
with torch.no_grad():
    pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
rect = pre["rect"].numpy().astype(np.int64) if "rect" in pre else None
if rect is None:
    rmin, rmax = pre["rect_min"].numpy().astype(np.int64), pre["rect_max"].numpy().astype(np.int64)
else:
    rmin, rmax = rect[:, :2], rect[:, 2:]
w = np.maximum(rmax[:, 0] - rmin[:, 0], 0)
h = np.maximum(rmax[:, 1] - rmin[:, 1], 0)
tt = pre["tiles_touched"].numpy().astype(np.int64)
vis = tt > 0                                  # (the rectangle of a culled Gaussian is not meaningful)
tiles = np.where(vis, w * h, 0)
R = int(tiles.sum())
assert R == int(tt.sum()) and np.array_equal(tiles, tt), (R, int(tt.sum()))
segs = int(h[vis].sum())
# emission order = depth order; a block of 4096 instances holds the segments of ~4096 / mean(tiles) Gaussians
depth = pre["depths"].numpy()
order = np.argsort(np.where(vis, depth, np.inf), kind="stable")[: int(vis.sum())]
t_sorted = tiles[order]
h_sorted = h[order]
ends = np.cumsum(t_sorted)
blk_of_last = (ends - 1) // 4096
blk_of_first = (ends - t_sorted) // 4096
n_blocks = int((R + 4095) // 4096)
# segments per block (a Gaussian that straddles a block boundary is counted in both: upper bound)
seg_per_blk = np.bincount(blk_of_first, weights=h_sorted, minlength=n_blocks) + np.bincount(blk_of_last, weights=(blk_of_last != blk_of_first) * h_sorted, minlength=n_blocks)
out = {"what": "row segments against instances in the emission (tools/cpu_emit_segment_model.py)", "P": P, "W": W, "H": H, "kind": KIND,
       "visible": int(vis.sum()), "R_instances": R, "row_segments": segs, "instances_per_segment_mean": round(R / segs, 3),
       "segment_length_hist": {str(k): int((w[vis] == k).sum() * 1) for k in range(1, 9)},
       "segment_length_weighted_share": {str(k): round(float((h[vis] * (w[vis] == k)).sum()) / segs, 4) for k in range(1, 9)},
       "emission_blocks_of_4096": n_blocks, "segments_per_block_mean": round(float(seg_per_blk.mean()), 1), "segments_per_block_max": int(seg_per_blk.max()),
       "gaussians_per_block_mean": round(float(vis.sum()) / n_blocks, 1),
       "note": "ranking segments instead of instances would rank R / (instances per segment) items; the expansion (one packed word per instance) remains"}
print(json.dumps(out))
