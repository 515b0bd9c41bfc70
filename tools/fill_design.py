#!/usr/bin/env python3
"""DESIGN.md = docs/DESIGN.template.md with the measured numbers of the round substituted from the committed profiles:
profiles/r04_bench_default.json (the default `python bench.py` line), profiles/r04_kernel_stats.csv (steady-state kernel medians of
the rocprofv3 --kernel-trace pass of the same command) and, when present, profiles/r04_ab_round3.json (the same-box comparison with
the round-3 library, tools/gpu_round_bundle.sh).  Run after every new measurement bundle:  python tools/fill_design.py"""
import csv
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"


def main():
    b = json.load(open(os.path.join(ROOT, "profiles", f"{TAG}_bench_default.json")))
    ks = {r["Name"]: float(r["MedianNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{TAG}_kernel_stats.csv")))}

    def k(prefix):
        for name, us in ks.items():
            if name.startswith(prefix):
                return us
        raise KeyError(prefix)

    ab = None
    rf, rt = b["roofline"], b["roofline_train"]
    fwd_steps, bwd_steps = b["blend_work"]["fwd_pair_steps_per_launch"], b["blend_work"]["bwd_pair_steps_per_launch"]
    chain = [k(n) for n in ("ds_hist", "ds_scan", "ds_scatter", "ds_segsort", "emit_hist", "rs_scan", "emit_scatter", "bucket_hist", "bucket_scatter")]
    cs = b["forward_cycled_scenes"]["preprocess_hbm"]
    sub = {
        "PRE_US": f"{k('preprocess_fwd_kernel<false>'):.1f}", "DSH_US": f"{k('ds_hist'):.1f}", "DSS_US": f"{k('ds_scan'):.1f}",
        "DSC_US": f"{k('ds_scatter'):.1f}", "DSG_US": f"{k('ds_segsort'):.1f}", "EH_US": f"{k('emit_hist'):.1f}", "RS_US": f"{k('rs_scan'):.1f}",
        "ES_US": f"{k('emit_scatter'):.1f}", "BH_US": f"{k('bucket_hist'):.1f}", "BS_US": f"{k('bucket_scatter'):.1f}",
        "RF_US": f"{k('render_fwd_wave_bf<true, 1, false>'):.1f}",
        "FRAME_US": f"{b['ms_per_step'] * 1e3:.1f}",
        "FWD_STEPS": f"{fwd_steps:,}".replace(",", " "), "FWD_GFLOP": f"{fwd_steps * 64 * 25 / 1e9:.2f}",
        "FWD_TF": f"{rf['achieved']:.1f}", "FWD_FRAC": f"{rf['frac']:.3f}", "TRAF_MB": f"{(rf['traffic'] or 0) / 1e6:.0f}",
        "BWD_STEPS": f"{bwd_steps:,}".replace(",", " "), "BWD_US": f"{rt['kernel_ms'] * 1e3:.0f}", "BWD_TF": f"{rt['achieved']:.1f}", "BWD_FRAC": f"{rt['frac']:.3f}",
        "WF_TBS": f"{b['whole_forward']['achieved_GBs'] / 1e3:.2f}", "WF_FRAC": f"{b['whole_forward']['frac_of_8TBs']:.3f}",
        "PREH_US": f"{cs['ms'] * 1e3:.1f}", "PREH_TBS": f"{cs['GBs'] / 1e3:.2f}", "PREH_FRAC": f"{cs['frac_of_8TBs']:.3f}",
        "CHAIN_US": f"{sum(chain):.0f}",
        "FWD_VALU_M": f"{(rf.get('valu_instructions_per_launch_from_committed_profile') or 0) / 1e6:.1f}",
        "FWD_VALU_PER_STEP": f"{(rf.get('valu_instructions_per_launch_from_committed_profile') or 0) / max(1, fwd_steps):.1f}",
        "FWD_ISSUE_FRAC": f"{rf.get('valu_issue_frac_of_measured_ceiling_from_committed_profile') or 0:.2f}",
        "FWD_ISSUE": f"{rf.get('valu_lane_ops_T_per_s_from_committed_profile') or 0:.1f}",
        "WHOLE6M": os.environ.get("GSR_WHOLE6M", "(see `profiles/r05_parity_report.json`)"),
    }
    oc = b["other_configs_forward"]
    fl = b["train_full_loop_configs2"]
    cpu = b["cpu_baseline"]
    sh = b["train_iters_per_s_sh_step_in_backward"]
    rows = [
        ("**forward, configs[1] stand-in (`value`)**", f"**{b['value']:.0f} Mpix/s — {b['ms_per_step']:.4f} ms per frame** (HIP-event median {b['gpu_event_ms']['forward']['median_ms']:.4f}); tracking build {b['forward_builds_ms'][[x for x in b['forward_builds_ms'] if x.startswith('tracking')][0]]:.4f} ms"),
        ("stage times (library's HIP events, ms)", ", ".join(f"{n} {b['stage_ms'][n]:.4f}" for n in ("preprocess", "depth_sort", "emit", "tile_sort", "render", "r_wait"))),
        ("the same frame with the REFERENCE's tile rectangles (`snug_tiles = 0`: the configuration whose bins are bit-exact against the oracle in reference mode)",
         f"{b['forward_reference_rectangles']['ms_per_frame']:.4f} ms per frame = {b['forward_reference_rectangles']['Mpix_s']:.0f} Mpix/s (R = {b['forward_reference_rectangles']['num_rendered']:,})".replace(",", " ")),
        ("32 cameras cycled / 3 parameter sets cycled", f"{b['forward_cycled_views']['ms_per_frame']:.4f} / {b['forward_cycled_scenes']['ms_per_frame']:.4f} ms per frame"),
    ] + ([("three independent frames in flight on 3 HIP streams (a camera-list loop; **not** `value`)",
            f"{b['forward_frames_in_flight']['ms_per_frame']:.4f} ms per frame = {b['forward_frames_in_flight']['Mpix_s']:.0f} Mpix/s")] if b.get("forward_frames_in_flight") else []) + [
        ("train step (fused L1+SSIM loss, fused Adam)", f"**{b['train_iters_per_s']:.1f} it/s** ({b['train_ms_per_iter']:.4f} ms); SparseGaussianAdam + separate-SH form {b['train_iters_per_s_sparse_adam']:.1f}; L1 only {b['train_iters_per_s_l1']:.1f}; depth-supervised {b['train_iters_per_s_depth_supervised']:.1f}; SH step inside the backward {sh['dense_adam_sh_step_in_backward']:.1f} / {sh['sparse_adam_sh_step_in_backward']:.1f} (dense / sparse)"),
        ("backward stages (ms)", ", ".join(f"{n} {b['stage_ms'][n]:.4f}" for n in ("render_bwd", "gather_bwd", "preprocess_bwd"))),
        ("train with density control every 100 it. (P 1.03 → 1.23 M)", f"{b['train_iters_per_s_densify']:.1f} it/s"),
        ("configs[2]: the reference's 30 000-iteration schedule, P₀ = 100 K", f"{fl['iters_per_s']:.1f} it/s ({fl['seconds']:.1f} s wall, final P {fl['final_P']}; `profiles/{TAG}_train_run_sparse.json`)"),
    ]
    for name, label in (("configs[3] 1M@4K", "configs[3]: 1 M @ 3840×2160"), ("configs[4] 6M@1080p", "configs[4]: 6 M @ 1080p"),
                        ("configs[1] s_med 0.006", "configs[1], s̃ = 0.006"), ("configs[1] clustered", "configs[1], clustered scene (longest tile list 8 487)")):
        o = oc[name]
        rows.append((label, f"{o['ms_per_frame']:.4f} ms = {o['Mpix_s']:.0f} Mpix/s (R = {o['num_rendered']:,})".replace(",", " ")))
    rows += [
        ("roofline (forward blend; kernel duration = mean / median of the launches of this run)", f"{rf['achieved']:.1f} TFLOP/s = {rf['frac']:.3f} of 157.3 (fp32 VALU) on the mean {rf['kernel_ms'] * 1e3:.1f} µs, {rf['frac_median']:.3f} on the median {rf['kernel_ms_median'] * 1e3:.1f} µs; issue rate {rf.get('valu_lane_ops_T_per_s_from_committed_profile') or 0:.1f} T lane-ops/s = {rf.get('valu_issue_frac_of_measured_ceiling_from_committed_profile') or 0:.2f} of the measured ceiling; PMC traffic {sub['TRAF_MB']} MB / launch (counters: the committed rocprofv3 --pmc passes of the same command)"),
        ("roofline (blend backward)", f"{rt['achieved']:.1f} TFLOP/s = {rt['frac']:.3f} (mean {rt['kernel_ms'] * 1e3:.0f} µs), {rt['frac_median']:.3f} (median {rt['kernel_ms_median'] * 1e3:.0f} µs); PMC traffic {(rt['traffic'] or 0) / 1e6:.0f} MB / launch"),
        ("CPU baseline (oracle, same frame, same run)", f"{cpu['value']:.4f} Mpix/s on {cpu['cores']} cores ({cpu['seconds_per_frame']:.1f} s per frame)"),
    ]
    lv = b.get("train_low_visibility")
    if lv:
        rows.insert(-3, (f"train step where {lv['visible_fraction'] * 100:.0f} % of the Gaussians are visible (clustered scene): dense fused Adam / SparseGaussianAdam", f"{lv['dense_adam_it_s']:.1f} / {lv['sparse_adam_it_s']:.1f} it/s"))
    sm_path = os.path.join(ROOT, "profiles", f"{TAG}_shard_model.json")
    if os.path.exists(sm_path):
        sm = json.load(open(sm_path))
        for name, cfg in sm.items():
            parts = []
            for G in ("2", "4", "8"):
                e = cfg.get(G)
                if not e:
                    continue
                t = (f"{G} GPUs **{e['speedup_back_to_back']:.2f}×**" if "speedup_back_to_back" in e else f"{G} GPUs") + f" ({e['speedup_pipelined']:.2f}× / {e['speedup_serial']:.2f}×)"
                parts.append(t)
            rows.append((f"multi-GPU **model** from per-rank kernels measured on one GPU + modelled xGMI collectives, {name}: speed-up with every rank's frames queued back to back, as `bench.py` times them (in brackets: every piece timed on its own with a synchronisation — pipelined / serial, round 4's method)", "; ".join(parts)))
    sub["RESULTS_TABLE"] = "| | |\n|---|---|\n" + "\n".join(f"| {a} | {c} |" for a, c in rows)
    text = open(os.path.join(ROOT, "docs", "DESIGN.template.md")).read()
    for key in sorted(sub, key=len, reverse=True):
        text = text.replace(key, sub[key])
    open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
    left = [w for w in ("_US", "_FRAC", "_TBS", "RESULTS_TABLE") if w in text]
    print("DESIGN.md written;", "placeholders left: " + str(left) if left else "no placeholder left")


if __name__ == "__main__":
    main()
