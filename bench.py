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
  train_iters_per_s : forward + loss + backward + Adam on all parameters, a NEW CAMERA EVERY ITERATION (train.py:96-102:
                      --views synthetic cameras around the generating one, cycled), steps/s; legs for the dense fused Adam,
                      for the reference's accelerated call form (SparseGaussianAdam + separate SH tensors,
                      train.py:37-41,180-183) and for L1 only
  stages            : every pipeline stage with its mean HIP-event duration, algorithmic bytes (SURVEY.md 8(d) terms) and
                      the GB/s / fraction of 8 TB/s that gives
  roofline          : the dominant kernel of the forward (the blend).  It is fp32-VALU bound (SQ counters: VALU issue, not
                      memory, limits it), so `bound` = "valu": achieved = evaluated (pixel, Gaussian) pair-steps counted by
                      the kernel itself x 25 FLOP / its HIP-event duration, against the 157.3 TFLOP/s fp32
                      vector peak; the HBM view (algorithmic bytes, GB/s, PMC traffic) rides along under "hbm"
  roofline_train    : the same for the dominant kernel of the train step (the blend backward)
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
    ap.add_argument("--variant", type=int, default=None, help="render_fwd_variant (0 wave + LDS broadcast, 1 workgroup-per-tile baseline, 2 wave + readlane)")
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
    ap.add_argument("--bwd-variant", type=int, default=None, help="render_bwd_variant (0 default, 1 atomics baseline, 2 128-entry super-batches)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="gsr_set_option(NAME, VALUE) before timing (A/B switches)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short forward measurements of configs[3] (1 M @4K) and configs[4] (6 M @1080p)")
    ap.add_argument("--views", type=int, default=32, help="training views cycled by the train legs (1 = the static camera of round 1)")
    return ap.parse_args()


def algorithmic_bytes(P, V, R, T, npix, M=16, track=False):
    """SURVEY.md 8(d) B_fwd, term by term (inference layout unless track; sort counted as one read + one write of 12-byte
    pairs), mapped onto the stages of this pipeline."""
    pre = 12 * P + V * (32 + 12 * M) + 8 * P + 40 * V + (27 * V if track else 0)
    scan = 8 * P
    dup = 8 * P + 16 * V + 12 * R
    sort = 24 * R
    ranges = 8 * R + 8 * T
    blend = 44 * R + (24 if track else 16) * npix
    return {"total": pre + scan + dup + sort + ranges + blend, "blend": blend, "preprocess": pre, "scan": scan,
            "emit": dup, "tile_sort": sort + ranges, "depth_sort": 0}


def algorithmic_bytes_bwd(P, V, R, npix, M=16):
    """SURVEY.md 8(d) B_bwd terms + Adam."""
    return {"render_bwd": 44 * R + 32 * npix, "gather_bwd": 40 * R, "preprocess_bwd": 136 * V + V * (48 + 27 + 32 + 12 * M) + V * (40 + 12 * M),
            "adam": 7 * 4 * 59 * P}


FWD_FLOP_PER_PAIR = 25      # ~12 FLOP quadratic form + exp + ~8 FLOP blend per (pixel, Gaussian) pair
BWD_FLOP_PER_PAIR = 60      # bwd_step: ~49 FLOP per pair (alpha, T / accumulator recurrences, 10 gradient terms) + ~10 adds of the cross-lane sums


def make_views(make_camera, look_at_camera, W, H, n):
    """Camera 0 is the generating camera of the scene; the others sit on a small circle around it and look at
    the middle of the cloud, so every view sees most of the scene but R, the lists and the scratch sizes change per frame."""
    cams = [make_camera(W, H)]
    for k in range(1, n):
        ang = 2 * math.pi * k / max(1, n - 1)
        r = 0.25 + 0.35 * ((k * 7) % 5) / 4.0
        cams.append(look_at_camera(W, H, (r * math.cos(ang), r * math.sin(ang), -0.2 * ((k % 3) - 1)), (0.0, 0.0, 7.0)))
    return cams


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

    from gsr_synth import look_at_camera, make_camera, make_scene
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _lib, rasterize_gaussians
    from diff_gaussian_rasterization.debug import forward_with_views
    from diff_gaussian_rasterization.parallel import BandPlan, gather_strips_async, row_costs_from_ranges

    _lib.load()
    if a.variant is not None:
        _lib.set_option("render_fwd_variant", a.variant)
    if a.bwd_variant is not None:
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
    # work counters of the blend kernel in their OWN pass (two global words shared by 32 640 waves: the kernel runs several
    # times slower while they are on)
    _lib.profile_enable(False, counters=True)
    _lib.profile_counters(reset=True)
    ncount = min(3, a.steps)
    for _ in range(ncount):
        forward_step()
    while in_flight:
        in_flight.pop(0).wait()
    torch.cuda.synchronize()
    fwd_counters = _lib.profile_counters(reset=True)
    _lib.profile_enable(False)
    stage_ms = {k: (v["ms"] / max(1, v["launches"])) for k, v in stages.items() if v["launches"]}
    fwd_steps_per_launch = fwd_counters["fwd_steps"] / max(1, ncount)
    fwd_batches_per_launch = fwd_counters["fwd_batches"] / max(1, ncount)

    # the TRACKING build of the forward (what a training iteration runs: final_T / n_contrib / first-emission indices are
    # written for the backward), timed beside the inference build the headline uses
    track_ms = None
    if world == 1:
        req = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]

        def forward_track():
            rasterize_gaussians(req[0], None, req[1], None, req[2], req[3], req[4], None, rs, None)
        for _ in range(5):
            forward_track()
        track_ms = timed_loop(forward_track, a.steps, "forward_track")[0] / a.steps * 1e3
        _lib.profile_reset()
        _lib.profile_enable(True)
        for _ in range(min(10, a.steps)):
            forward_track()
        torch.cuda.synchronize()
        tr = _lib.profile_read()
        _lib.profile_enable(False)
        stage_ms["render_track"] = tr["render"]["ms"] / max(1, tr["render"]["launches"])
        del req

    # ---- train legs: forward + loss + backward + optimizer over all parameters (train.py:111-186 without densification; the
    # full loop with density control on the reference's schedule is tools/train_run.py).  A NEW CAMERA EVERY ITERATION
    # (train.py:96-102): --views synthetic cameras, cycled, each with its own target image.
    #   "ssim"        : the reference's loss, 0.8 L1 + 0.2 (1 - SSIM) (train.py:119-126) with the fused HIP SSIM, dense fused Adam
    #   "sparse_adam" : same loss, the reference's accelerated call form: dc= / shs= separate SH tensors + SparseGaussianAdam
    #                   stepping only the visible Gaussians (train.py:37-41,180-183; gaussian_renderer/__init__.py:82-100)
    #   "l1"          : L1 only with the dense fused Adam (isolates the rasterizer + optimizer)
    #   "l1_torch_adam" / "ssim_torch": torch.optim.Adam / un-fused torch SSIM (--compare-torch-adam)
    train, train_gap = {}, {}
    bwd_counters = None
    tsteps = a.steps if a.train_steps < 0 else a.train_steps
    if tsteps > 0:
        from gsr_optim import FusedAdam
        from gsr_synth.losses import train_loss, l1_loss
        from fused_ssim import fused_ssim
        from diff_gaussian_rasterization import SparseGaussianAdam

        def fused_train_loss(image, gt_image, lambda_dssim=0.2):      # train.py:119-126 with FUSED_SSIM_AVAILABLE
            return (1.0 - lambda_dssim) * l1_loss(image, gt_image) + lambda_dssim * (1.0 - fused_ssim(image[None], gt_image[None]))
        views = make_views(make_camera, look_at_camera, W, H, max(1, a.views))
        rs_views, gts = [], []
        for i, vc in enumerate(views):
            vd = vc.to(dev)
            rs_views.append(GaussianRasterizationSettings(H, W, vc.tanfovx, vc.tanfovy, bg, 1.0, vd.world_view_transform,
                                                          vd.full_proj_transform, 3, vd.camera_center, False, False, False))
            gts.append(torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1 + i)))
        legs = [("ssim", "dense", fused_train_loss), ("sparse_adam", "sparse", fused_train_loss), ("l1", "dense", l1_loss)]
        if a.compare_torch_adam:
            legs.append(("l1_torch_adam", "torch", l1_loss))
            legs.append(("ssim_torch", "dense", train_loss))
        if world > 1:
            legs = [l for l in legs if l[1] != "sparse"]      # the sharded path keeps the fused [P,16,3] SH tensor
        from diff_gaussian_rasterization.parallel import render_two_axis, padded_shard_size
        # N > 1: two-axis sharding -- rank g owns Gaussians [lo, hi) (parameters + Adam state) and a band of
        # tile rows; 64-byte splat records are all-gathered forward, 48-byte gradient records reduce-scattered backward
        lo, hi = (P * rank) // world, (P * (rank + 1)) // world
        P_pad = padded_shard_size(hi - lo) if world > 1 else P
        for leg, opt_kind, loss_fn in legs:
            if opt_kind == "sparse":
                src = (sc.means3D, sc.shs[:, :1].contiguous(), sc.shs[:, 1:].contiguous(), sc.opacities, sc.scales, sc.rotations)
            else:
                src = (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)
            params = [t[lo:hi].detach().clone().requires_grad_(True) for t in src]
            if opt_kind == "sparse":
                opt = SparseGaussianAdam([{"params": [p_], "lr": 1e-5} for p_ in params], lr=1e-5, eps=1e-15)
            elif opt_kind == "torch":
                opt = torch.optim.Adam(params, lr=1e-5, eps=1e-15)
            else:
                opt = FusedAdam(params, lr=1e-5, eps=1e-15)
            it_no = [0]

            def train_step():
                vi = it_no[0] % len(rs_views)
                it_no[0] += 1
                opt.zero_grad(set_to_none=True)
                if opt_kind == "sparse":
                    m, dc, rest, o, s_, r_ = params
                    color, radii, invd = rasterize_gaussians(m, None, rest, None, o, s_, r_, None, rs_views[vi], None, None, dc)
                elif world > 1:
                    m, sh, o, s_, r_ = params
                    color, radii, invd = render_two_axis(rs_views[vi], m, sh, o, s_, r_, plan, P_pad)
                else:
                    m, sh, o, s_, r_ = params
                    color, radii, invd = rasterize_gaussians(m, None, sh, None, o, s_, r_, None, rs_views[vi], None)
                loss = loss_fn(color, gts[vi])
                loss.backward()
                if opt_kind == "sparse":
                    opt.step(radii > 0, radii.shape[0])
                else:
                    opt.step()

            for _ in range(max(len(rs_views) if world == 1 else 2, a.warmup // 2)):     # every view once: scratch sizes settle
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
                it_no[0] = 0                      # counters on view 0 (the frame of the forward metric), own pass
                _lib.profile_enable(False, counters=True)
                _lib.profile_counters(reset=True)
                train_step()
                torch.cuda.synchronize()
                bwd_counters = _lib.profile_counters(reset=True)
                _lib.profile_enable(False)
            del params, opt      # (no empty_cache(): the next leg re-uses the cached blocks instead of re-allocating)
    train_ms = train.get("ssim")
    train_ips = None if train_ms is None else 1e3 / train_ms

    # ---- the other GPU configs of BASELINE.json, forward only, same sharding (short: they are not the headline metric) ----
    #   configs[3] stand-in: 1 M Gaussians @3840x2160;  configs[4] stand-in: 6 M Gaussians @1920x1080
    other = {}
    if not a.no_other_configs and (P, W, H) == (1_000_000, 1920, 1080):
        for oname, oP, oW, oH in (("configs[3] 1M@4K", 1_000_000, 3840, 2160), ("configs[4] 6M@1080p", 6_000_000, 1920, 1080)):
            ocam = make_camera(oW, oH)
            osc = make_scene(oP, ocam, seed=a.seed, s_med=a.s_med).to(dev)
            ocd = ocam.to(dev)
            ors = GaussianRasterizationSettings(oH, oW, ocam.tanfovx, ocam.tanfovy, bg, 1.0, ocd.world_view_transform,
                                                ocd.full_proj_transform, 3, ocd.camera_center, False, False, False)
            ogx, ogy = (oW + 15) // 16, (oH + 15) // 16
            with torch.no_grad():
                ov = forward_with_views(ors, osc.means3D, osc.opacities, shs=osc.shs, scales=osc.scales, rotations=osc.rotations)
                oR = int(ov["R"])
                orow = row_costs_from_ranges(ov["ranges"].long(), ogx, ogy, banded=False)
                del ov
            oplan = BandPlan.uniform(ogy, world) if (a.uniform_bands or world == 1) else BandPlan.balanced(orow, world)
            oband = None if world == 1 else oplan.band(rank)

            def ostep():
                with torch.no_grad():
                    color, _, _ = rasterize_gaussians(osc.means3D, None, osc.shs, None, osc.opacities, osc.scales, osc.rotations,
                                                      None, ors, oband)
                    if world > 1:
                        in_flight.append(gather_strips_async(color, oplan, oH))
                        if len(in_flight) > 1:
                            in_flight.pop(0).wait()
            for _ in range(5):
                ostep()
            osteps = max(5, min(20, a.steps))
            odt = timed_loop(ostep, osteps, "other_" + oname)[0]
            other[oname] = {"P": oP, "width": oW, "height": oH, "num_rendered": oR, "steps": osteps,
                            "ms_per_frame": round(odt / osteps * 1e3, 4), "Mpix_s": round(oW * oH / (odt / osteps) / 1e6, 1)}
            del osc
            torch.cuda.empty_cache()

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
        abb = algorithmic_bytes_bwd(P, V, R, npix)
        frac_rows = 1.0 if world == 1 else (plan.band(0)[1] - plan.band(0)[0]) / gy

        def pmc(kernel_names):
            """HBM traffic / VALU instruction counts cannot be collected in-process: they come from the committed rocprofv3
            --pmc passes of this same command (profiles/pmc_latest.json; FETCH_SIZE and WRITE_SIZE in separate runs, bytes =
            (2*FETCH + WRITE)*1024 per MI355X_MICROARCH.md; SQ_INSTS_VALU from the SQ pass).  None when unavailable."""
            try:
                pk = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))["kernels"]
                if world == 1 and (P, W, H) == (1_000_000, 1920, 1080):
                    for kn in kernel_names:
                        if kn in pk:
                            return pk[kn]
            except Exception:
                pass
            return None

        def blend_roofline(kernel, ms, steps, flop_per_pair, hbm_bytes, pmc_entry, what, pairs_per_step=64.0):
            if not ms:
                return None
            flops = steps * pairs_per_step * flop_per_pair
            ach = flops / (ms * 1e-3) / 1e12
            gbs = hbm_bytes / (ms * 1e-3) / 1e9
            r = {"bound": "valu", "kernel": kernel, "achieved": round(ach, 3), "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s",
                 "frac": round(ach / FP32_VALU_PEAK_TF, 5), "kernel_ms": round(ms, 4),
                 "evaluated_pair_steps_per_launch": int(steps * pairs_per_step), "flop_per_pair": flop_per_pair,
                 "counted": what,
                 "traffic": None if not pmc_entry else int(pmc_entry.get("hbm_bytes_corrected", 0)) or None,
                 "traffic_source": None if not pmc_entry else "profiles/pmc_latest.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)",
                 "hbm": {"algorithmic_bytes_per_launch": int(hbm_bytes), "achieved_GBs": round(gbs, 2),
                         "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 5), "frac_of_6.29TBs": round(gbs / HBM_ACHIEVABLE_GBS, 5)},
                 "note": "fp32-VALU bound, not HBM bound (SURVEY 8(d); SQ counters: waves wait for VALU issue, measured HBM traffic is a "
                         "fraction of the algorithmic bytes because the splat records stay in L2 / Infinity Cache and early "
                         "termination ends the walks); FLOPs counted = pairs the kernel actually evaluates (whole waves: 64 pixels "
                         "per surviving list entry), not the 256*R listed pairs"}
            if pmc_entry and pmc_entry.get("SQ_INSTS_VALU"):
                lane_ops = pmc_entry["SQ_INSTS_VALU"] * 64.0 / (ms * 1e-3) / 1e12
                r["valu_lane_ops_T_per_s"] = round(lane_ops, 2)
                r["valu_issue_frac_of_spec"] = round(lane_ops / (FP32_VALU_PEAK_TF / 2.0), 4)     # spec: one lane-op per lane and clock at 2.4 GHz
                r["valu_instructions_per_launch"] = int(pmc_entry["SQ_INSTS_VALU"])
            return r

        roof = blend_roofline("render_fwd_wave_bf<LDS, inference>", stage_ms.get("render"), fwd_steps_per_launch, FWD_FLOP_PER_PAIR,
                              ab["blend"] * frac_rows, pmc(["render_fwd_wave_bf<true, 1, false>"]),
                              "wave-level (8x8 pixel block, list entry) pairs that survive the exact box test and are blended, counted by the kernel")
        roof_train = None
        if bwd_counters:
            roof_train = blend_roofline("render_bwd_half", stage_ms.get("render_bwd"), bwd_counters["bwd_steps"], BWD_FLOP_PER_PAIR,
                                        abb["render_bwd"] * frac_rows, pmc(["render_bwd_half"]),
                                        "wave-level (16x8 half tile, list entry) steps of the backward walk x 128 pixels, counted by the kernel",
                                        pairs_per_step=128.0)
        # every stage: HIP-event ms, algorithmic bytes (SURVEY 8(d) term of the stage), GB/s
        stage_tab = {}
        sbytes = {"preprocess": ab["preprocess"], "scan": ab["scan"], "emit": ab["emit"], "tile_sort": ab["tile_sort"], "render": ab["blend"],
                  "render_track": algorithmic_bytes(P, V, R, gx * gy, npix, track=True)["blend"],
                  "render_bwd": abb["render_bwd"], "gather_bwd": abb["gather_bwd"], "preprocess_bwd": abb["preprocess_bwd"]}
        design = {"depth_sort": 80 * P, "emit": 16 * P + 4 * R, "tile_sort": 12 * R + 8 * gx * gy}
        for k, ms in stage_ms.items():
            e = {"ms": round(ms, 4)}
            if k in sbytes:
                e["algorithmic_bytes"] = int(sbytes[k])
                e["GBs"] = round(sbytes[k] / (ms * 1e-3) / 1e9, 1)
                e["frac_of_8TBs"] = round(sbytes[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if k in design:
                e["design_bytes"] = int(design[k])      # what this pipeline's kernels move by design for the stage
                e["design_GBs"] = round(design[k] / (ms * 1e-3) / 1e9, 1)
            stage_tab[k] = e
        whole = ab["total"] / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "Mpix/s forward (1 M Gaussians @1080p); train iters/s alongside",
            "value": round(mpix_s, 2), "unit": "Mpix/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "frame_latency_ms": round(latency_ms, 4),
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "parity": "oracle-only (the reference rasterizer is an un-vendored submodule)",
            "config": {"workload": "configs[1] stand-in: 1M random Gaussians (SURVEY 8(d) generator, seed %d, s_med %.4g), "
                                   "%dx%d forward render, SH degree 3" % (a.seed, a.s_med, W, H),
                       "P": P, "visible": V, "num_rendered": R, "tiles": gx * gy,
                       "parallelism": "tile-row bands x%d%s" % (world, "" if world == 1 else
                                                                (" (uniform)" if a.uniform_bands else " (instance-balanced)") +
                                                                ", strip all-gather of frame i overlapped with frame i+1"),
                       "render_fwd_variant": a.variant or 0, "options": os.environ.get("GSR_OPTIONS", ""),
                       "frame_streams": n_streams,
                       "frame_streams_note": "value = frames / time with consecutive frames alternating between HIP streams "
                                             "(double-buffered); frame_latency_ms = one frame after the other on one stream"
                                             if n_streams > 1 else "one frame after the other on one stream"},
            "forward_builds_ms": {"inference (value; torch.no_grad(): no final_T / n_contrib / first-emission writes)": round(ms_per_step, 4),
                                  "tracking (what a training iteration runs)": None if track_ms is None else round(track_ms, 4)},
            "train_iters_per_s": None if train_ips is None else round(train_ips, 3),
            "train_ms_per_iter": None if train_ms is None else round(train_ms, 4),
            "train_step": "forward + loss 0.8 L1 + 0.2 (1-SSIM) (train.py:119-126; fused HIP SSIM) + backward + fused HIP Adam over "
                          "all 59 floats/Gaussian, a new camera every iteration (%d views cycled, train.py:96-102); *_sparse_adam = the "
                          "reference's accelerated call form (separate dc / rest SH tensors + SparseGaussianAdam on visible rows, "
                          "train.py:180-183); *_l1 = L1 loss only; *_ssim_torch = SSIM through torch conv2d ops; full loop with density "
                          "control: tools/train_run.py" % max(1, a.views) +
                          ("" if world == 1 else "; N > 1: Gaussians AND tile rows sharded (record all-gather / gradient "
                                                  "reduce-scatter), loss replicated on the all-gathered image"),
            "train_iters_per_s_sparse_adam": None if "sparse_adam" not in train else round(1e3 / train["sparse_adam"], 3),
            "train_iters_per_s_ssim_torch": None if "ssim_torch" not in train else round(1e3 / train["ssim_torch"], 3),
            "train_iters_per_s_l1": None if "l1" not in train else round(1e3 / train["l1"], 3),
            "train_iters_per_s_l1_torch_adam": None if "l1_torch_adam" not in train else round(1e3 / train["l1_torch_adam"], 3),
            "whole_forward": {"algorithmic_bytes": int(ab["total"]), "achieved_GBs": round(whole, 2),
                              "frac_of_8TBs": round(whole / HBM_PEAK_GBS, 5),
                              "frac_of_6.29TBs": round(whole / HBM_ACHIEVABLE_GBS, 5),
                              "roofline_predicted_Mpix_s": round(npix / (ab["total"] / (HBM_PEAK_GBS * 1e9)) / 1e6, 1),
                              "note": "HBM roofline of SURVEY 8(d)'s B_fwd; structurally out of reach because the blend is VALU-bound "
                                      "(its %.3f ms alone exceed the %.3f ms the whole B_fwd takes at 8 TB/s)" %
                                      (stage_ms.get("render", 0.0), ab["total"] / (HBM_PEAK_GBS * 1e9) * 1e3)},
            "train_max_step_gap_ms": {k: round(v, 3) for k, v in train_gap.items()},
            "retimed": retimed,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stages": stage_tab,
            "blend_work": {"fwd_pair_steps_per_launch": int(fwd_steps_per_launch), "fwd_batches_per_launch": int(fwd_batches_per_launch),
                           "bwd_pair_steps_per_launch": None if not bwd_counters else int(bwd_counters["bwd_steps"]),
                           "bwd_batches_per_launch": None if not bwd_counters else int(bwd_counters["bwd_batches"]),
                           "listed_instance_blocks": 4 * R},
            "other_configs_forward": other,
            "roofline": roof,
            "roofline_train": roof_train,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
