"""Per-rank cost of the Gaussian-sharded forward (mode C of diff_gaussian_rasterization/parallel.py), measured on ONE GPU -- no
multi-GPU box is available to the builder, so this is the prediction the driver's SCALE run is to be checked against
("every term measured on one GPU").

For G = 2, 4, 8 and every rank g of the instance-balanced plan, exactly the kernels rank g of a G-GPU run executes are timed
with HIP events on this GPU:
    project   gsr_preprocess_forward on the rank's P/G Gaussians
    route     gsr_route_count + gsr_route_pack (per-band stable compaction into 48-byte records)
    band      gsr_rasterize_from_packed on the records rank g RECEIVES (built here by running all G shards' route once and
              concatenating the segments in rank order = the all-to-all), with the library's per-stage events: ingest, depth
              sort, scan, emission + level 1, level 2 + ranges, blend
and the two collectives are MODELLED from the measured byte counts (xGMI: one ~153 GB/s link to every peer, used
concurrently; HOP_US per collective for launch + first byte):
    all-to-all   max over peers of (records to that peer x 48 B) / 153 GB/s, send and receive side
    strips       every rank receives the other ranks' strips: (G-1)/G of 3 W H 4 B over min(G-1, 7) links
plus COUNT_SYNC_US for the G x G count exchange and its host read-back.
Reported per G: the slowest rank's GPU time, the collectives, and two speed-ups over the single-GPU frame:
    pipelined  = T1 / max(slowest rank's GPU time, slowest collective)   (bench.py keeps two frames in flight, so collectives
                 overlap the neighbouring frames' kernels)
    serial     = T1 / (GPU time + both collectives + count sync)         (one frame, nothing overlapped: the latency view)

    python tools/gpu_shard_model.py            # configs[1], [3], [4] stand-ins + the clustered scene
Round 4: the FIXED-CAPACITY form of the exchange (no count matrix on the host; segments of capacity + 1 rows, capacity = what
parallel.ExchangePolicy learns from the frame's own count matrix, x 1.25 rounded up to 256) is measured beside it: route_fixed /
band_fixed per rank (the padding rows enter the band's frame as tile-less Gaussians), the all-to-all modelled on capacity x 48 B per
peer, and the two speed-ups without COUNT_SYNC_US ("*_fixed").
Output: one JSON document on stdout (also gpurun_out/shard_model.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch
from gsr_synth import make_camera, make_scene, make_clustered_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, rasterize_gaussians, _lib
from diff_gaussian_rasterization.debug import forward_with_views
from diff_gaussian_rasterization.parallel import (BandPlan, ExchangePolicy, row_costs_from_ranges, hip_preprocess_shard, hip_route_count,
                                                 hip_route_pack, hip_route_pack_fixed, hip_render_packed, hip_render_segments)

dev = torch.device("cuda:0")
LINKS, LINK_GBS, HOP_US, COUNT_SYNC_US = 7, 153.0, 15.0, 25.0


def median_ms(fn, steps=12, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def throughput_ms(fn, steps=30, warm=6):
    """ms per call of fn when the calls follow each other on the stream WITHOUT a synchronisation in between (the host queues ahead of the GPU,
    as bench.py's timed loops and a pipelined multi-GPU run do) -- median_ms above times every call on its own and so adds the launch latency
    of its first kernel and the host's Python time to every piece, which inflates short pieces (a 250 K-Gaussian projection: 42 us that way,
    about half of it latency) much more than the single-GPU frame they are compared with."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


out = {}
for name, P, W, H, kind in (("configs[1] 1M@1080p", 1_000_000, 1920, 1080, "uniform"), ("configs[3] 1M@4K", 1_000_000, 3840, 2160, "uniform"),
                            ("configs[4] 6M@1080p", 6_000_000, 1920, 1080, "uniform"), ("configs[1] clustered", 1_000_000, 1920, 1080, "clustered")):
    cam = make_camera(W, H)
    sc = (make_clustered_scene(P, cam, seed=0) if kind == "clustered" else make_scene(P, cam, seed=0, s_med=0.012)).to(dev)
    camd = cam.to(dev)
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                       camd.full_proj_transform, 3, camd.camera_center, False, False, False)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        v0 = forward_with_views(rs, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    row_cost = row_costs_from_ranges(v0["ranges"].long(), gx, gy, banded=False)
    R, V = int(v0["R"]), int((v0["radii"] > 0).sum())
    del v0

    def full_frame():
        with torch.no_grad():
            rasterize_gaussians(sc.means3D, None, sc.shs, None, sc.opacities, sc.scales, sc.rotations, None, rs, None)
    t1 = median_ms(full_frame, steps=20, warm=8)
    t1_thr = throughput_ms(full_frame)
    res = {"P": P, "W": W, "H": H, "kind": kind, "visible": V, "R": R, "1": {"frame_ms": round(t1, 4), "frame_back_to_back_ms": round(t1_thr, 4)}}
    strip_bytes = 3 * W * H * 4
    for G in (2, 4, 8):
        plan = BandPlan.balanced(row_cost, G)
        bounds = list(plan.bounds)
        cuts = [(P * g) // G for g in range(G + 1)]
        shards, packs = [], []
        with torch.no_grad():
            for g in range(G):
                t = tuple(x[cuts[g]:cuts[g + 1]].contiguous() for x in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations))
                records, _, _, _ = hip_preprocess_shard(rs, *t)
                counts, scratch = hip_route_count(records, bounds)
                sc_counts = counts.cpu().tolist()
                packed, _, offs = hip_route_pack(records, bounds, sc_counts, scratch)
                shards.append(t)
                packs.append((packed, offs, sc_counts))
        pol = ExchangePolicy("fixed")
        pol.observe(max(max(pk[2]) for pk in packs))
        cap = int(pol.capacity)
        fixed_segs = []
        with torch.no_grad():
            for g in range(G):
                records, _, _, _ = hip_preprocess_shard(rs, *shards[g])
                counts, scratch = hip_route_count(records, bounds)
                fixed_segs.append(hip_route_pack_fixed(records, bounds, cap, scratch, counts)[0])
        per = []
        for g in range(G):
            t = shards[g]
            holder = {}

            def project():
                with torch.no_grad():
                    holder["rec"] = hip_preprocess_shard(rs, *t)[0]

            def route():
                with torch.no_grad():
                    c, scr = hip_route_count(holder["rec"], bounds)
                    hip_route_pack(holder["rec"], bounds, packs[g][2], scr)      # (counts known from the set-up pass: no sync in the timed region)
            recv = torch.cat([packs[src][0][packs[src][1][g]:packs[src][1][g + 1]] for src in range(G)], dim=0).contiguous()

            def band():
                with torch.no_grad():
                    hip_render_packed(rs, plan.band(g), recv, True)

            def route_fixed():
                with torch.no_grad():
                    c, scr = hip_route_count(holder["rec"], bounds)
                    hip_route_pack_fixed(holder["rec"], bounds, cap, scr, c)
            recv_fixed = torch.cat([fixed_segs[src][g * (cap + 1):(g + 1) * (cap + 1)] for src in range(G)], dim=0).contiguous()

            def band_fixed():
                with torch.no_grad():
                    hip_render_segments(rs, plan.band(g), recv_fixed, G, cap, True)
            ms_project = median_ms(project)
            ms_route = median_ms(route)
            ms_band = median_ms(band)
            ms_route_fixed = median_ms(route_fixed)
            ms_band_fixed = median_ms(band_fixed)

            def chain():          # what rank g queues per frame, back to back (the all-to-all sits between route and band: modelled)
                project(); route(); band()

            def chain_fixed():
                project(); route_fixed(); band_fixed()
            ms_chain = throughput_ms(chain)
            ms_chain_fixed = throughput_ms(chain_fixed)
            del recv_fixed
            _lib.profile_reset(); _lib.profile_enable(True)
            for _ in range(5):
                band()
            torch.cuda.synchronize()
            st = _lib.profile_read(); _lib.profile_enable(False)
            stage = {("ingest" if k == "preprocess" else k): round(v["ms"] / max(1, v["launches"]), 4) for k, v in st.items() if v["launches"]}
            sent = [packs[g][2][d] for d in range(G)]
            recvd = [packs[src][2][g] for src in range(G)]
            a2a_us = HOP_US + max(max((c for d, c in enumerate(sent) if d != g), default=0),
                                  max((c for s_, c in enumerate(recvd) if s_ != g), default=0)) * 48 / (LINK_GBS * 1e3)
            per.append({"rank": g, "band": plan.band(g), "P_shard": cuts[g + 1] - cuts[g], "P_received": int(recv.shape[0]),
                        "rows_sent_to_others": int(sum(sent) - sent[g]), "project_ms": round(ms_project, 4), "route_ms": round(ms_route, 4),
                        "band_ms": round(ms_band, 4), "gpu_ms": round(ms_project + ms_route + ms_band, 4), "band_stage_ms": stage,
                        "route_fixed_ms": round(ms_route_fixed, 4), "band_fixed_ms": round(ms_band_fixed, 4),
                        "gpu_fixed_ms": round(ms_project + ms_route_fixed + ms_band_fixed, 4),
                        "chain_back_to_back_ms": round(ms_chain, 4), "chain_fixed_back_to_back_ms": round(ms_chain_fixed, 4),
                        "all_to_all_model_us": round(a2a_us, 1)})
            del recv
        gpu = max(p["gpu_ms"] for p in per)
        a2a = max(p["all_to_all_model_us"] for p in per)
        strips_us = HOP_US + strip_bytes * (G - 1) / G / (min(G - 1, LINKS) * LINK_GBS * 1e3)
        total_rows = sum(sum(pk[2]) for pk in packs)
        gpu_f = max(p["gpu_fixed_ms"] for p in per)
        a2a_f = HOP_US + (cap + 1) * 48 / (LINK_GBS * 1e3)
        chain = max(p["chain_back_to_back_ms"] for p in per)
        chain_f = max(p["chain_fixed_back_to_back_ms"] for p in per)
        res[str(G)] = {"slowest_rank_gpu_ms": round(gpu, 4), "all_to_all_model_us": a2a, "strip_allgather_model_us": round(strips_us, 1),
                       "count_sync_us": COUNT_SYNC_US, "rows_exchanged_over_visible": round(total_rows / max(1, V), 3),
                       "speedup_pipelined": round(t1 / max(gpu, a2a * 1e-3, strips_us * 1e-3), 2),
                       "speedup_serial": round(t1 / (gpu + (a2a + strips_us + COUNT_SYNC_US) * 1e-3), 2),
                       # the throughput view: frames back to back on every rank (what bench.py times), against the single GPU's back-to-back frames
                       "slowest_rank_chain_back_to_back_ms": round(chain, 4),
                       "speedup_back_to_back": round(t1_thr / max(chain, a2a * 1e-3, strips_us * 1e-3), 2),
                       "speedup_back_to_back_fixed_exchange": round(t1_thr / max(chain_f, a2a_f * 1e-3, strips_us * 1e-3), 2),
                       "fixed_exchange": {"capacity": cap, "rows_padded_over_rows_exact": round(G * G * (cap + 1) / max(1, total_rows), 3),
                                          "slowest_rank_gpu_ms": round(gpu_f, 4), "all_to_all_model_us": round(a2a_f, 1),
                                          "speedup_pipelined": round(t1 / max(gpu_f, a2a_f * 1e-3, strips_us * 1e-3), 2),
                                          "speedup_serial": round(t1 / (gpu_f + (a2a_f + strips_us) * 1e-3), 2)},
                       "ranks": per}
        del shards, packs, fixed_segs
        torch.cuda.empty_cache()
    out[name] = res
    del sc
    torch.cuda.empty_cache()
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "shard_model.json"), "w"), indent=1)
