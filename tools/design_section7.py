#!/usr/bin/env python
"""Regenerates DESIGN.md section 7 (the table of measured results) from the committed profiles of a round:
    python tools/design_section7.py r06        (profiles/r06_bench_default.json, profiles/r06_shard_model.json)"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
d = json.load(open(os.path.join(ROOT, "profiles", f"{TAG}_bench_default.json")))
sm = json.load(open(os.path.join(ROOT, "profiles", f"{TAG}_shard_model.json")))
a = s.index("## 7. Measured")
b = s.index("## 8. Round")
sh, oc, st, fp = d["train_iters_per_s_sh_step_in_backward"], d["other_configs_forward"], d["stage_ms"], d["train_iters_per_s_fixed_P"]
wt = d["train_densify"]["where_the_time_goes"]
cl = oc["configs[1] clustered"]
g = json.load(open(os.path.join(ROOT, "profiles", f"{TAG}_train_run_sparse_growth.json")))
gw = [w["iters_per_s"] for w in g["windows"] if w["until_iter"] > 10000]


def model(k):
    v = sm[k]
    return "2 GPUs **%.2f×** (%.2f× / %.2f×); 4 GPUs **%.2f×** (%.2f× / %.2f×); 8 GPUs **%.2f×** (%.2f× / %.2f×)" % tuple(
        x for n in ("2", "4", "8") for x in (v[n]["speedup_back_to_back"], v[n]["speedup_pipelined"], v[n]["speedup_serial"]))


sp = lambda t: t.replace(",", " ")      # noqa: E731
rows = [
    ("**forward, configs[1] stand-in (`value`)**", f"**{d['value']:.0f} Mpix/s — {d['ms_per_step']:.4f} ms per frame** (HIP-event median {d['gpu_event_ms']['forward']['median_ms']:.4f}); tracking build {d['forward_builds_ms'][[k for k in d['forward_builds_ms'] if k.startswith('tracking')][0]]:.4f} ms"),
    ("stage times (library's HIP events, ms)", ", ".join(f"{n} {st[n]:.4f}" for n in ("preprocess", "depth_sort", "emit", "tile_sort", "render", "r_wait"))),
    ("**`value_reference_bins`**: the same frame with the REFERENCE's tile rectangles (`snug_tiles = 0`: the configuration whose bins are bit-exact against the oracle in reference mode)", sp(f"**{d['value_reference_bins']:.0f} Mpix/s** — {d['forward_reference_rectangles']['ms_per_frame']:.4f} ms per frame (R = {d['forward_reference_rectangles']['num_rendered']:,})")),
    ("32 cameras cycled / 3 parameter sets cycled", f"{d['forward_cycled_views']['ms_per_frame']:.4f} / {d['forward_cycled_scenes']['ms_per_frame']:.4f} ms per frame"),
    ("three independent frames in flight on 3 HIP streams (a camera-list loop; **not** `value`)", f"{d['forward_frames_in_flight']['ms_per_frame']:.4f} ms per frame = {d['forward_frames_in_flight']['Mpix_s']:.0f} Mpix/s"),
    ("train step (fused L1+SSIM loss, fused Adam)", f"**{d['train_iters_per_s']:.1f} it/s** ({d['train_ms_per_iter']:.4f} ms); SparseGaussianAdam + separate-SH form {d['train_iters_per_s_sparse_adam']:.1f}; L1 only {d['train_iters_per_s_l1']:.1f}; depth-supervised {d['train_iters_per_s_depth_supervised']:.1f}; SH step inside the backward {sh['dense_adam_sh_step_in_backward']:.1f} / {sh['sparse_adam_sh_step_in_backward']:.1f} (dense / sparse)"),
    ("**train step, the unchanged caller's op sequence** (`train.py:104-186` verbatim on the three drop-in packages: `render()` glue with torch activations, torch `l1_loss`, `fused_ssim`, `loss.item()` every iteration, torch boolean-index density statistics, exposure optimizer)", f"**{d['train_iters_per_s_unchanged_caller']:.1f} it/s** with `torch.optim.Adam` (train.py's default `optimizer_type`; torch's foreach Adam alone is ≈ 1.5 ms of the {d['train_unchanged_caller']['default_torch_adam']['ms_per_iter']:.2f} ms) — **{d['train_iters_per_s_unchanged_caller_sparse_adam']:.1f} it/s** with `--optimizer_type sparse_adam` (`SparseGaussianAdam.step(visible, N)`); with ONE added line, `gsr_scene.densify.attach(gaussians)` behind `training_setup()` (INTEGRATION.md §3): **{d.get('train_iters_per_s_unchanged_caller_one_added_line') or float('nan'):.1f} it/s** with the default optimizer (it becomes the fused kernel, same groups and state), {d.get('train_iters_per_s_unchanged_caller_sparse_adam_one_added_line') or float('nan'):.1f} with sparse_adam"),
    ("train step at fixed P (SURVEY 8(d); headline step)", sp(f"P = 1e5: {fp['100000']['iters_per_s']:.0f} it/s (R = {fp['100000']['num_rendered_view0']:,}); 1e6: {fp['1000000']['iters_per_s']:.0f}; 3e6: {fp['3000000']['iters_per_s']:.0f} (R = {fp['3000000']['num_rendered_view0']:,})")),
    ("backward stages (ms)", ", ".join(f"{n} {st[n]:.4f}" for n in ("render_bwd", "gather_bwd", "preprocess_bwd"))),
    ("train with density control every 100 it. (P 1.03 → 1.23 M)", f"{d['train_iters_per_s_densify']:.1f} it/s; {wt['clone_split_prune_events']} clone / split / prune events of {wt['ms_per_event']:.1f} ms, plain iterations {wt['plain_iteration_ms_median']:.2f} ms median, " + ("no stall beyond 20 ms outside the events" if not wt["stalls_over_20ms_outside_events"] else f"stalls outside the events: {wt['stalls_over_20ms_outside_events']} ms ({wt['iters_per_s_without_those_stalls']:.1f} it/s without them)")),
    ("configs[2]: the reference's 30 000-iteration schedule, P₀ = 100 K", f"{d['train_iters_per_s_full_loop_configs2']:.1f} it/s (final P {d['train_full_loop_configs2']['final_P']}; `profiles/{TAG}_train_run_sparse.json`); **growth settings (grad threshold 2e-5): P → {g['final_P'] / 1e6:.2f} M, {g['value']:.0f} it/s over the run, {min(gw):.0f}–{max(gw):.0f} it/s after iteration 10 000 — blend-bound, §8** (`profiles/{TAG}_train_run_sparse_growth.json`, `{TAG}_train_timeline_grown.json`)"),
    ("configs[3]: 1 M @ 3840×2160", sp(f"{oc['configs[3] 1M@4K']['ms_per_frame']:.4f} ms = {oc['configs[3] 1M@4K']['Mpix_s']:.0f} Mpix/s (R = {oc['configs[3] 1M@4K']['num_rendered']:,})")),
    ("configs[4]: 6 M @ 1080p", sp(f"{oc['configs[4] 6M@1080p']['ms_per_frame']:.4f} ms = {oc['configs[4] 6M@1080p']['Mpix_s']:.0f} Mpix/s (R = {oc['configs[4] 6M@1080p']['num_rendered']:,})")),
    ("configs[1], s̃ = 0.006", sp(f"{oc['configs[1] s_med 0.006']['ms_per_frame']:.4f} ms = {oc['configs[1] s_med 0.006']['Mpix_s']:.0f} Mpix/s (R = {oc['configs[1] s_med 0.006']['num_rendered']:,}; round 5: 0.4409)")),
    ("configs[1], clustered scene (longest tile list 8 487)", sp(f"{cl['ms_per_frame']:.4f} ms = {cl['Mpix_s']:.0f} Mpix/s (R = {cl['num_rendered']:,}; round 5: 0.4590 — the cluster crowded the uniform depth buckets)")),
    ("train step where 39 % of the Gaussians are visible (clustered scene): dense fused Adam / SparseGaussianAdam", f"{1e3 / cl['train_step_dense_adam_ms']:.1f} / {1e3 / cl['train_step_sparse_adam_ms']:.1f} it/s"),
    ("roofline (forward blend; kernel duration = mean / median of the launches of this run)", f"{d['roofline']['achieved']:.1f} TFLOP/s = {d['roofline']['frac']:.3f} of 157.3 (fp32 VALU) on the mean {d['roofline']['kernel_ms'] * 1e3:.1f} µs, {d['roofline']['frac_median']:.3f} on the median {d['roofline']['kernel_ms_median'] * 1e3:.1f} µs; " + (f"**PMC traffic {d['roofline']['traffic'] / 1e6:.0f} MB / launch measured by this run** " if d['roofline']['traffic'] else "PMC traffic not measured in this run ") + f"(committed passes of the full command: {(d['roofline']['traffic_from_committed_profile'] or 0) / 1e6:.0f} MB)"),
    ("roofline (blend backward)", f"{d['roofline_train']['achieved']:.1f} TFLOP/s = {d['roofline_train']['frac']:.3f} (mean {d['roofline_train']['kernel_ms'] * 1e3:.0f} µs); PMC traffic {(d['roofline_train']['traffic'] or 0) / 1e6:.0f} MB / launch (this run)"),
    ("CPU baseline (oracle, same frame, same run)", f"{d['cpu_baseline']['value']:.4f} Mpix/s on {d['cpu_baseline']['cores']} cores ({d['cpu_baseline']['seconds_per_frame']:.1f} s per frame)"),
]
for k in ("configs[1] 1M@1080p", "configs[3] 1M@4K", "configs[4] 6M@1080p", "configs[1] clustered"):
    rows.append((f"multi-GPU **model** from per-rank kernels measured on one GPU + modelled xGMI collectives, {k}: speed-up with every rank's frames queued back to back, as `bench.py` times them (in brackets: every piece timed on its own with a synchronisation — pipelined / serial)", model(k)))
new = f"## 7. Measured (round {int(TAG[1:])}, MI355X, ROCm 7.2; bench frame P = 1 M, 1920×1080; `profiles/{TAG}_bench_default.json` — one box; boxes differ by ± 4 %)\n\n| | |\n|---|---|\n" + "".join(f"| {x} | {y} |\n" for x, y in rows) + "\n"
open(p, "w").write(s[:a] + new + s[b:])
print("section 7 rewritten from", TAG)
