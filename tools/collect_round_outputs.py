#!/usr/bin/env python
"""Copies what a round's GPU calls left under gpurun_out/ (scratch) into profiles/ (tracked) under the round's names:
    python tools/collect_round_outputs.py r06
Only files that exist are copied; prints what it did.  (tools/collect_profiles.py makes gpurun_out/profiles_<tag>/ ON the box; this runs in the repo.)"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = [(f, os.path.join(P, os.path.basename(f))) for f in glob.glob(os.path.join(G, f"profiles_{tag}", "*"))]
for src, dst in (("train_run_sparse.json", "train_run_sparse"), ("train_run_sparse_growth.json", "train_run_sparse_growth"), ("train_run_sparse_fused_sh.json", "train_run_sparse_fused_sh"),
                 ("shard_model.json", "shard_model"), ("train_run_sparse_timeline_grown.json", "train_timeline_grown"), ("train_kernel_timeline_grown.json", "train_kernel_timeline_grown"),
                 ("depth_distribution_probe.json", "depth_distribution_probe"), ("parity_report.json", "parity_report"), ("host_cprofile.json", "host_cprofile"),
                 ("fuzz_bins.json", None), ("fuzz_render_71.json", "fuzz_render_seed71"), ("fuzz_render_7.json", "fuzz_render_seed7"),
                 ("fuzz_render_101.json", "fuzz_render_seed101"), ("fuzz_render_202.json", "fuzz_render_seed202")):
    s = os.path.join(G, src)
    if not os.path.exists(s):
        continue
    if dst is None:      # bins fuzz: named by its seed
        dst = f"fuzz_bins_seed{json.load(open(s)).get('seed', 0)}"
    pairs.append((s, os.path.join(P, f"{tag}_{dst}.json")))
for s, d in pairs:
    shutil.copyfile(s, d)
    print(f"{os.path.relpath(s, ROOT)} -> {os.path.relpath(d, ROOT)}")
# library A/B (tools/gpu_ab.sh LIBS=...): ab_<lib>_<rep>.log -> one json
ab = {}
for f in sorted(glob.glob(os.path.join(G, "ab_lib*_[0-9].log"))):
    lib = os.path.basename(f)[3:-6]
    try:
        d = json.loads([ln for ln in open(f) if ln.startswith("{")][-1])
    except Exception:      # noqa: BLE001
        continue
    ab.setdefault(lib, []).append({"ms_per_frame": d["ms_per_step"], "Mpix_s": d["value"], "train_iters_per_s": d["train_iters_per_s"], "stage_ms": d["stage_ms"]})
if len(ab) >= 2:
    out = {"what": "python bench.py (forward 50 steps + the train legs) with each library on ONE box, three interleaved runs each (tools/gpu_ab.sh LIBS=...); lib = this round's final library, lib_prev = round 5's (commit 57985bc)",
           "runs": ab}
    json.dump(out, open(os.path.join(P, f"{tag}_ab_libraries_final.json"), "w"), indent=1)
    print("A/B ->", f"profiles/{tag}_ab_libraries_final.json", {k: [r["ms_per_frame"] for r in v] for k, v in ab.items()})
