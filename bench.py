#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: "Mpix/s forward + train iters/s, 1 M Gaussians @1080p, 1/2/4/8 MI355X".

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one forward render of the whole frame (preprocess -> depth sort -> scan -> emit -> tile sort -> ranges
-> blend, including the 4-byte num_rendered read-back) of the synthetic stand-in for configs[1] ("garden", ~1 M
Gaussians, 1080p): P = 1e6, 1920x1080, SURVEY.md 8(d) generator, seed 0, s_med 0.012.  With N > 1 GPUs the SAME
frame is split into N bands of tile rows (strong scaling), every rank renders its band and the strips are
all-gathered over RCCL; value = W*H / (max-over-ranks time per frame).

The JSON line also carries
  train_iters_per_s : forward + L1 loss + backward + Adam on all parameters (same scene), steps/s
  roofline          : the dominant kernel (forward blend), algorithmic bytes / its mean duration from HIP events
                      recorded on the launch stream by the library (gsr_profile_*), against the 8 TB/s HBM peak
  cpu_baseline      : the pure-PyTorch CPU oracle on this host's cores, on a bounded sample of the same frame
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "gaussian-splatting_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)
HBM_ACHIEVABLE_GBS = 6290.0
FP32_VALU_PEAK_TF = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--P", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--s-med", type=float, default=0.012)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0, help="render_fwd_variant (0 wave + LDS broadcast, 1 workgroup-per-tile baseline, 2 wave + readlane)")
    ap.add_argument("--train-steps", type=int, default=-1, help="-1: same as --steps; 0 disables the train leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tiles", type=int, default=1200, help="tiles blended by the CPU baseline sample")
    ap.add_argument("--uniform-bands", action="store_true")
    ap.add_argument("--compare-torch-adam", action="store_true", help="also time the train step with torch.optim.Adam")
    ap.add_argument("--min-warm-seconds", type=float, default=1.5, help="the untimed warm-up lasts at least this long (wall clock)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the forward frames alternate on (N = 1 GPU): 1 = strictly "
                                                          "one frame after the other (default), 2 = double-buffered frames -- "
                                                          "measured 2.5x SLOWER on MI355X / ROCm 7.2 (1.29 vs 0.52 ms per frame), kept for A/B only")
    ap.add_argument("--sort-items", type=int, default=0, help="keys per workgroup of the large radix passes (1024 / 2048 / 4096; 0 = library default)")
    ap.add_argument("--bwd-variant", type=int, default=0, help="render_bwd_variant (0 default, 1 atomics baseline, 2 128-entry super-batches)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="gsr_set_option(NAME, VALUE) before timing (A/B switches)")
    return ap.parse_args()


def algorithmic_bytes(P, V, R, T, npix, M=16):
    """B_fwd, inference layout, sort counted as one read + one write of 12-byte pairs."""
    pre_in = 12 * P + V * (32 + 12 * M)
    pre_out = 8 * P + 40 * V
    scan = 8 * P
    dup_in = 8 * P + 16 * V
    dup_out = 12 * R
    sort = 24 * R
    ranges = 8 * R + 8 * T
    blend = 44 * R + 24 * npix
    return {"total": pre_in + pre_out + scan + dup_in + dup_out + sort + ranges + blend, "blend": blend,
            "preprocess": pre_in + pre_out, "sort": sort + dup_out}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback for the product path)"
    # Test hook (tests/test_gpu_next_rows.py): GSR_BENCH_SHARED_GPU=1 + GSR_BENCH_BACKEND=gloo run the N-rank code path
    # with all ranks on device 0 of a single-GPU box (RCCL refuses two ranks on one device).  Never set by the driver.
    shared_gpu = os.environ.get("GSR_BENCH_SHARED_GPU") == "1"
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from gsr_synth import make_camera, make_scene
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _lib, rasterize_gaussians
    from diff_gaussian_rasterization.debug import forward_with_views
    from diff_gaussian_rasterization.parallel import BandPlan, gather_strips_async, row_costs_from_ranges

    _lib.load()
    _lib.set_option("render_fwd_variant", a.variant)
    _lib.set_option("render_bwd_variant", a.bwd_variant)
    if a.sort_items:
        _lib.set_option("sort_items_large", a.sort_items)
    for kv in a.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    W, H, P = a.width, a.height, a.P
    cam = make_camera(W, H)
    scene_cpu = make_scene(P, cam, seed=a.seed, s_med=a.s_med)
    sc = scene_cpu.to(dev)
    camd = cam.to(dev)
    bg = torch.zeros(3, device=dev)
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg, 1.0, camd.world_view_transform,
                                       camd.full_proj_transform, 3, camd.camera_center, False, False, False)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    npix = W * H

    # ---- scene statistics (V, R) + band plan from a dry run ----
    with torch.no_grad():
        v0 = forward_with_views(rs, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        V = int((v0["radii"] > 0).sum())
        R = int(v0["R"])
        row_cost = row_costs_from_ranges(v0["ranges"].long(), gx, gy, banded=False)   # full frame on every rank
        del v0
    plan = BandPlan.uniform(gy, world) if (a.uniform_bands or world == 1) else BandPlan.balanced(row_cost, world)
    band = None if world == 1 else plan.band(rank)

    # N > 1: frames are pipelined two deep -- the strip all-gather of frame i (RCCL stream) overlaps the rasterization
    # of frame i+1; every frame is complete (assembled on every rank) before the closing barrier of the timed region.
    in_flight = []

    def forward_step():
        with torch.no_grad():
            color, radii, invd = rasterize_gaussians(sc.means3D, None, sc.shs, None, sc.opacities, sc.scales,
                                                     sc.rotations, None, rs, band)
            if world > 1:
                in_flight.append(gather_strips_async(color, plan, H))
                if len(in_flight) > 1:
                    color = in_flight.pop(0).wait()
        return color

    def sync_all():
        while in_flight:
            in_flight.pop(0).wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- timed forward ----
    # One-off stalls: on a fresh box a single ~75 ms pause of the whole GPU shows up once, a few hundred ms into the first
    # sustained activity of the process (seen at a fixed wall-clock offset, in whichever loop happened to be running; not
    # tied to any kernel or allocation of this code).  Two guards keep it out of the numbers without hiding anything:
    # the untimed warm-up runs for at least --min-warm-seconds of wall clock, and a timed loop whose host-side step stamps
    # (the host is paced by the per-frame R read-back) show a gap of more than STALL_MS is timed again, once; the JSON
    # line reports how often that happened ("retimed").
    STALL_MS = 20.0
    retimed = {}

    def timed_loop(step, n, name):
        for attempt in range(2):
            sync_all()
            t0 = time.perf_counter()
            stamps = [t0]
            for _ in range(n):
                step()
                stamps.append(time.perf_counter())
            sync_all()
            dt = time.perf_counter() - t0
            gaps = [b - a for a, b in zip(stamps[:-1], stamps[1:])]
            gmax = max(gaps) * 1e3 if gaps else 0.0
            flag = torch.tensor([1.0 if gmax > STALL_MS else 0.0], device=dev)
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)       # all ranks take the same decision
            if float(flag.item()) == 0.0 or attempt == 1:
                break
            retimed[name] = retimed.get(name, 0) + 1
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item()), gmax, int(max(range(len(gaps)), key=gaps.__getitem__)) if gaps else 0

    def timed_forward(step):
        t_w = time.perf_counter()
        for _ in range(a.warmup):
            step()
        while True:                                                # extended untimed warm-up (see above)
            more = torch.tensor([1.0 if time.perf_counter() - t_w < a.min_warm_seconds else 0.0], device=dev)
            if world > 1:
                dist.all_reduce(more, op=dist.ReduceOp.MAX)            # every rank must issue the same number of collectives
            if float(more.item()) == 0.0:
                break
            for _ in range(10):
                step()
        return timed_loop(step, a.steps, "forward")[0]

    dt = timed_forward(forward_step)                 # one frame after the other on one stream
    latency_ms = dt / a.steps * 1e3
    n_streams = 1
    if world == 1 and a.streams > 1:
        # Double-buffered frames: the pipeline of one frame is a strict chain of ~25 kernels, half of them small
        # latency-bound ones (sort passes, scans) that leave most CUs idle.  Consecutive frames are independent, so they
        # alternate between HIP streams: while the host waits for frame i+1's instance count, the GPU already holds
        # frame i's emit / tile sort / blend, and frame i+1's preprocess + depth sort run beside them.  Every frame is
        # rendered completely (same kernels, same outputs); all frames are complete before the closing synchronize.
        n_streams = a.streams
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
        counter = [0]

        def forward_step_streams():
            st = streams[counter[0] % n_streams]
            counter[0] += 1
            with torch.cuda.stream(st):
                forward_step()

        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        dt = timed_forward(forward_step_streams)
    ms_per_step = dt / a.steps * 1e3
    mpix_s = npix / (dt / a.steps) / 1e6

    # ---- per-stage HIP-event timing (separate pass: event pairs around every stage perturb the pipeline) ----
    _lib.profile_reset()
    _lib.profile_enable(True)
    for _ in range(a.steps):
        forward_step()
    while in_flight:
        in_flight.pop(0).wait()
    torch.cuda.synchronize()
    stages = _lib.profile_read()
    _lib.profile_enable(False)
    stage_ms = {k: (v["ms"] / max(1, v["launches"])) for k, v in stages.items() if v["launches"]}

    # ---- train leg: forward + loss + backward + Adam over all parameters (train.py:111-186 without densification) ----
    #   "ssim"   : the reference's loss, 0.8 L1 + 0.2 (1 - SSIM)  (train.py:119-126) with the fused HIP SSIM (fused_ssim package)
    #   "ssim_torch": same loss through utils/loss_utils.py-style torch conv2d ops (the reference's un-fused fallback)
    #   "l1"     : L1 only (isolates the rasterizer + optimizer)
    #   "l1_torch_adam": as "l1" but with torch.optim.Adam instead of the fused HIP Adam (gsr_optim.FusedAdam)
    train, train_gap = {}, {}
    tsteps = a.steps if a.train_steps < 0 else a.train_steps
    if tsteps > 0:
        from gsr_optim import FusedAdam
        from gsr_synth.losses import train_loss, l1_loss
        from fused_ssim import fused_ssim

        def fused_train_loss(image, gt_image, lambda_dssim=0.2):      # train.py:119-126 with FUSED_SSIM_AVAILABLE
            return (1.0 - lambda_dssim) * l1_loss(image, gt_image) + lambda_dssim * (1.0 - fused_ssim(image[None], gt_image[None]))
        gt = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        legs = [("ssim", FusedAdam, fused_train_loss), ("l1", FusedAdam, l1_loss)]
        if a.compare_torch_adam:
            legs.append(("l1_torch_adam", torch.optim.Adam, l1_loss))
            legs.append(("ssim_torch", FusedAdam, train_loss))
        from diff_gaussian_rasterization.parallel import render_two_axis, padded_shard_size
        # N > 1: two-axis sharding -- rank g owns Gaussians [lo, hi) (parameters + Adam state) and a band of
        # tile rows; 64-byte splat records are all-gathered forward, 48-byte gradient records reduce-scattered backward
        lo, hi = (P * rank) // world, (P * (rank + 1)) // world
        P_pad = padded_shard_size(hi - lo) if world > 1 else P
        for leg, opt_cls, loss_fn in legs:
            params = [t[lo:hi].detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
            opt = opt_cls(params, lr=1e-5, eps=1e-15)

            def train_step():
                opt.zero_grad(set_to_none=True)
                m, sh, o, s_, r_ = params
                if world > 1:
                    color, radii, invd = render_two_axis(rs, m, sh, o, s_, r_, plan, P_pad)
                else:
                    color, radii, invd = rasterize_gaussians(m, None, sh, None, o, s_, r_, None, rs, None)
                loss = loss_fn(color, gt)
                loss.backward()
                opt.step()

            for _ in range(max(2, a.warmup // 2)):
                train_step()
            tdt_s, gmax, gat = timed_loop(train_step, tsteps, "train_" + leg)
            train_gap[leg] = gmax
            train_gap[leg + "_at_step"] = gat
            train[leg] = tdt_s / tsteps * 1e3
            if leg == "l1":
                _lib.profile_reset()
                _lib.profile_enable(True)
                for _ in range(min(10, tsteps)):
                    train_step()
                torch.cuda.synchronize()
                tstages = _lib.profile_read()
                _lib.profile_enable(False)
                for k in ("render_bwd", "gather_bwd", "preprocess_bwd"):
                    if tstages[k]["launches"]:
                        stage_ms[k] = tstages[k]["ms"] / tstages[k]["launches"]
            del params, opt      # (no empty_cache(): the next leg re-uses the cached blocks instead of re-allocating)
    train_ms = train.get("ssim")
    train_ips = None if train_ms is None else 1e3 / train_ms

    # ---- CPU baseline (rank 0, N=1 only): pure-PyTorch oracle on a bounded sample of the same frame ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import torch_oracle as O   # cpu_baseline leg only
        # many small tensor ops: beyond ~16 threads torch's intra-op pool only adds overhead (measured: 256
        # threads on the GPU box's host were >10x slower than 16), so the baseline uses min(cores, 16) threads
        cores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(cores)
        s = O.settings_from_camera(cam, torch.zeros(3))
        with torch.no_grad():
            t0 = time.perf_counter()
            pre = O.preprocess(scene_cpu.means3D, scene_cpu.opacities, s, shs=scene_cpu.shs, scales=scene_cpu.scales,
                               rotations=scene_cpu.rotations)
            t_pre = time.perf_counter() - t0
            t0 = time.perf_counter()
            bins = O.bin_and_sort(pre)
            t_bin = time.perf_counter() - t0
            ntile = gx * gy
            k = max(1, min(a.cpu_tiles, ntile))
            sample = [int(i * ntile / k) for i in range(k)]
            t0 = time.perf_counter()
            O.render_tiles(pre, bins, s, tiles=sample)
            t_blend = time.perf_counter() - t0
        inst_sample = int((bins["ranges"][sample, 1] - bins["ranges"][sample, 0]).sum())
        # extrapolate the blend by instance count (the blend cost is proportional to list length)
        t_full = t_pre + t_bin + t_blend * (bins["R"] / max(1, inst_sample))
        cpu_baseline = {"value": round(npix / t_full / 1e6, 6), "host_cores": os.cpu_count(), "unit": "Mpix/s", "cores": cores, "kind": "port",
                        "sample": f"full preprocess ({t_pre:.1f}s) + full binning/sort ({t_bin:.1f}s) on all {P} Gaussians; "
                                  f"blend timed on {k} of {ntile} tiles ({inst_sample} of {bins['R']} instances, {t_blend:.1f}s) "
                                  f"and scaled by instance count; pure-PyTorch oracle, torch threads={cores}"}

    if rank == 0:
        ab = algorithmic_bytes(P, V, R, gx * gy, npix)
        render_ms = stage_ms.get("render")
        roof = None
        if render_ms:
            # per-launch algorithmic bytes of the blend kernel on THIS rank's band: 44 B per instance + 24 B per pixel
            frac_rows = 1.0 if world == 1 else (plan.band(0)[1] - plan.band(0)[0]) / gy
            blend_bytes = ab["blend"] if world == 1 else ab["blend"] * frac_rows
            ach = blend_bytes / (render_ms * 1e-3) / 1e9
            pairs = 256.0 * R * (frac_rows if world > 1 else 1.0)
            # HBM traffic of this kernel cannot be collected in-process: it comes from the committed rocprofv3 --pmc passes
            # of this same command (profiles/pmc_latest.json: FETCH_SIZE and WRITE_SIZE in separate runs, bytes =
            # (2*FETCH + WRITE)*1024 per MI355X_MICROARCH.md); null when the file or the kernel is missing
            pmc_traffic, pmc_note = None, None
            try:
                pk = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))["kernels"]
                kn = {0: "render_fwd_wave_bf<true, 1, false>", 1: "render_fwd_block", 2: "render_fwd_wave_bf<false, 1, false>"}.get(a.variant)
                if world == 1 and kn in pk and (P, W, H) == (1_000_000, 1920, 1080):
                    pmc_traffic = int(pk[kn]["hbm_bytes_corrected"])
                    pmc_note = "profiles/pmc_latest.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)"
            except Exception:
                pass
            roof = {"bound": "hbm", "kernel": {0: "render_fwd_wave_bf<LDS>", 1: "render_fwd_block", 2: "render_fwd_wave_bf<readlane>"}.get(a.variant, "?"),
                    "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                    "frac_of_measured_achievable_6.29TBs": round(ach / HBM_ACHIEVABLE_GBS, 5),
                    "traffic": pmc_traffic, "traffic_source": pmc_note, "kernel_ms": round(render_ms, 4),
                    "algorithmic_bytes_per_launch": int(blend_bytes),
                    "note": "blend is fp32-VALU/exp bound, not HBM bound (SURVEY 8(d)); listed (pixel,Gaussian) pairs = 256*R per "
                            "launch; most are never evaluated (early termination + exact box culling, DESIGN.md 3.3)",
                    "listed_pairs_per_s": round(pairs / (render_ms * 1e-3), 1),
                                        }
        whole = ab["total"] / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "Mpix/s forward (1 M Gaussians @1080p); train iters/s alongside",
            "value": round(mpix_s, 2), "unit": "Mpix/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "frame_latency_ms": round(latency_ms, 4),
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1] stand-in: 1M random Gaussians (SURVEY 8(d) generator, seed %d, s_med %.4g), "
                                   "%dx%d forward render, SH degree 3" % (a.seed, a.s_med, W, H),
                       "P": P, "visible": V, "num_rendered": R, "tiles": gx * gy,
                       "parallelism": "tile-row bands x%d%s" % (world, "" if world == 1 else
                                                                (" (uniform)" if a.uniform_bands else " (instance-balanced)") +
                                                                ", strip all-gather of frame i overlapped with frame i+1"),
                       "render_fwd_variant": a.variant,
                       "frame_streams": n_streams,
                       "frame_streams_note": "value = frames / time with consecutive frames alternating between HIP streams "
                                             "(double-buffered); frame_latency_ms = one frame after the other on one stream"
                                             if n_streams > 1 else "one frame after the other on one stream"},
            "train_iters_per_s": None if train_ips is None else round(train_ips, 3),
            "train_ms_per_iter": None if train_ms is None else round(train_ms, 4),
            "train_step": "forward + loss 0.8 L1 + 0.2 (1-SSIM) (train.py:119-126; fused HIP SSIM) + backward + fused HIP Adam over "
                          "all 59 floats/Gaussian; *_l1 = L1 loss only; *_ssim_torch = SSIM through torch conv2d ops" +
                          ("" if world == 1 else "; N > 1: Gaussians AND tile rows sharded (record all-gather / gradient "
                                                  "reduce-scatter), loss replicated on the all-gathered image"),
            "train_iters_per_s_ssim_torch": None if "ssim_torch" not in train else round(1e3 / train["ssim_torch"], 3),
            "train_iters_per_s_l1": None if "l1" not in train else round(1e3 / train["l1"], 3),
            "train_iters_per_s_l1_torch_adam": None if "l1_torch_adam" not in train else round(1e3 / train["l1_torch_adam"], 3),
            "whole_forward": {"algorithmic_bytes": int(ab["total"]), "achieved_GBs": round(whole, 2),
                              "frac_of_8TBs": round(whole / HBM_PEAK_GBS, 5),
                              "frac_of_6.29TBs": round(whole / HBM_ACHIEVABLE_GBS, 5),
                              "roofline_predicted_Mpix_s": round(npix / (ab["total"] / (HBM_PEAK_GBS * 1e9)) / 1e6, 1)},
            "train_max_step_gap_ms": {k: round(v, 3) for k, v in train_gap.items()},
            "retimed": retimed,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": roof,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
