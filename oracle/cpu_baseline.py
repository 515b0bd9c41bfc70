#!/usr/bin/env python
"""CPU baseline of bench.py (SURVEY.md 8(d) "CPU baseline in the same run"; task rule: the oracle timed on the GPU box's host
cores, bounded to ~10-30 s of CPU work, cores stated).  TEST / MEASUREMENT INFRASTRUCTURE -- never on the product path.

    python oracle/cpu_baseline.py --P 1000000 --width 1920 --height 1080 [--workers 0] [--budget-s 40]

The pure-PyTorch oracle (oracle/torch_oracle.py) renders the forward of the same synthetic frame bench.py times on the GPU:
  * preprocess + binning/sort on all P Gaussians in this process (torch intra-op threads, at most 16: beyond that the pool of
    many small tensor ops only adds overhead -- measured on the GPU box's 256-core host in round 2);
  * the blend -- independent per tile -- on ALL host cores: one forked, single-threaded worker per core, tiles dealt round-robin.
    A short probe sizes the run: if the whole frame would exceed the budget, every k-th tile is blended and the time is scaled
    by instance count (the sample is reported); otherwise the WHOLE frame is blended, no extrapolation.
One JSON line: {"value": Mpix/s, "unit", "cores", "host_cores", "kind": "port", "sample": "..."}."""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gaussian-splatting_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

_G = {}      # state inherited by the forked workers (copy-on-write): pre, bins, settings


def _init_worker():
    torch.set_num_threads(1)


def _blend_chunk(tiles):
    from oracle import torch_oracle as O
    t0 = time.perf_counter()
    with torch.no_grad():
        O.render_tiles(_G["pre"], _G["bins"], _G["s"], tiles=tiles)
    return len(tiles), time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--s-med", type=float, default=0.012)
    ap.add_argument("--workers", type=int, default=0, help="0 = all host cores")
    ap.add_argument("--budget-s", type=float, default=40.0, help="wall-clock budget of the blend leg")
    a = ap.parse_args()
    from gsr_synth import make_camera, make_scene
    from oracle import torch_oracle as O
    host_cores = os.cpu_count() or 1
    workers = host_cores if a.workers <= 0 else min(a.workers, host_cores)
    torch.set_num_threads(min(host_cores, 16))
    cam = make_camera(a.width, a.height)
    sc = make_scene(a.P, cam, seed=a.seed, s_med=a.s_med)
    s = O.settings_from_camera(cam, torch.zeros(3))
    with torch.no_grad():
        t0 = time.perf_counter()
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        t_pre = time.perf_counter() - t0
        t0 = time.perf_counter()
        bins = O.bin_and_sort(pre)
        t_bin = time.perf_counter() - t0
    _G.update(pre=pre, bins=bins, s=s)
    gx, gy = pre["grid"]
    ntile = gx * gy
    counts = (bins["ranges"][:, 1] - bins["ranges"][:, 0])
    R = int(bins["R"])
    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_init_worker) as pool:
        # probe: two tiles per worker, spread over the frame; a first round of one tile each absorbs the workers' start-up
        # (page faults of the forked state, torch's lazy initialisation), which the first version billed to the estimate and
        # therefore blended only every second tile on a box that had time for all of them
        pool.map(_blend_chunk, [[int(i * ntile / workers)] for i in range(workers)])
        probe = [int(i * ntile / (2 * workers)) + 1 for i in range(2 * workers)]
        probe = [min(t, ntile - 1) for t in probe]
        t0 = time.perf_counter()
        pool.map(_blend_chunk, [[t] for t in probe])
        t_probe = time.perf_counter() - t0
        inst_probe = int(counts[probe].sum())
        est_full = t_probe * (R / max(1, inst_probe))
        stride = 1 if est_full <= a.budget_s else int(est_full / a.budget_s) + 1
        tiles = list(range(0, ntile, stride))
        nchunk = workers * 4
        chunks = [tiles[k::nchunk] for k in range(nchunk) if tiles[k::nchunk]]
        t0 = time.perf_counter()
        pool.map(_blend_chunk, chunks, chunksize=1)
        t_blend = time.perf_counter() - t0
    inst = int(counts[tiles].sum())
    t_blend_full = t_blend * (R / max(1, inst))
    t_full = t_pre + t_bin + t_blend_full
    npix = a.width * a.height
    whole = stride == 1
    print(json.dumps({
        "value": round(npix / t_full / 1e6, 6), "unit": "Mpix/s", "cores": workers, "host_cores": host_cores, "kind": "port",
        "seconds_per_frame": round(t_full, 3),
        "sample": (f"pure-PyTorch oracle, same frame as the GPU run (P {a.P}, {a.width}x{a.height}, seed {a.seed}, s_med {a.s_med}): "
                   f"preprocess {t_pre:.2f}s + binning/sort {t_bin:.2f}s on all Gaussians (torch threads {min(host_cores, 16)}); blend of "
                   + ("ALL " if whole else f"every {stride}th of ") + f"{ntile} tiles ({inst} of {R} instances) in {t_blend:.2f}s on {workers} "
                   f"forked single-threaded workers" + ("" if whole else ", scaled by instance count")),
    }), flush=True)


if __name__ == "__main__":
    main()
