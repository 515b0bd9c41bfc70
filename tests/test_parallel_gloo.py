"""world_size-2 (and 3) gloo tests of the screen-tile sharding logic (band plan, strip all-gather, gradient
all-reduce) on CPU, with the oracle standing in for the band renderer -- the collectives and index math are the
code under test; on the GPU box the same functions get the HIP rasterizer (bench.py --gpus N)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import O, make_camera, make_scene, make_edge_scene, oracle_settings


def test_band_plan():
    from diff_gaussian_rasterization.parallel import BandPlan
    p = BandPlan.uniform(68, 4)
    assert p.bounds == [0, 17, 34, 51, 68]
    cost = [1.0] * 10 + [9.0] * 10                      # bottom half 9x as expensive
    q = BandPlan.balanced(cost, 2)
    assert q.bounds[0] == 0 and q.bounds[-1] == 20 and 14 <= q.bounds[1] <= 16
    loads = [sum(cost[q.bounds[g]:q.bounds[g + 1]]) for g in range(2)]
    assert max(loads) / (sum(cost) / 2) < 1.15
    r = BandPlan.balanced([0.0] * 5, 3)
    assert r.bounds[0] == 0 and r.bounds[-1] == 5 and sorted(r.bounds) == r.bounds
    e = BandPlan.balanced([5.0, 0, 0, 0], 4)             # empty bands allowed, monotone
    assert sorted(e.bounds) == e.bounds and e.bounds[-1] == 4
    assert q.pixel_rows(1, 310) == (q.bounds[1] * 16, 310)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, balanced, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from diff_gaussian_rasterization.parallel import BandPlan, render_sharded, row_costs_from_ranges
    cam = make_camera(112, 96)
    sc = make_edge_scene(500, cam, seed=33)
    s = oracle_settings(cam, bg=torch.tensor([0.2, 0.4, 0.6]))
    gy = 6
    plan = BandPlan.uniform(gy, world)
    if balanced:   # re-balance from the per-row instance histogram (each rank knows its own band only)
        with torch.no_grad():
            *_, aux = O.rasterize(sc.means3D, None, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations,
                                  tile_y0=plan.band(rank)[0], tile_y1=plan.band(rank)[1], return_aux=True)
        plan = BandPlan.balanced(row_costs_from_ranges(aux["ranges"], 7, gy), world)
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m2 = torch.zeros(sc.P, 3, requires_grad=True)

    def band_renderer(inp, rows):
        m, sh, o, scl, rot, m2_ = inp
        return O.rasterize(m, m2_, o, s, shs=sh, scales=scl, rotations=rot, tile_y0=rows[0], tile_y1=rows[1])

    color, radii, invd = render_sharded(band_renderer, leaves + [m2], plan)
    g = torch.Generator().manual_seed(7)
    wc, wd = torch.randn(3, 96, 112, generator=g), torch.randn(1, 96, 112, generator=g)
    ((color * wc).sum() + (invd * wd).sum()).backward()
    if rank == 0:
        torch.save({"color": color.detach(), "invd": invd.detach(), "radii": radii, "bounds": plan.bounds,
                    "grads": [t.grad for t in leaves] + [m2.grad]}, out_q)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,balanced", [(2, False), (2, True), (3, True)])
def test_sharded_render_equals_single_device(world, balanced):
    import tempfile
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rank0.pt")
        procs = [ctx.Process(target=_worker, args=(r, world, port, balanced, path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=900)
            assert p.exitcode == 0
        res = torch.load(path)
    # single-device reference
    cam = make_camera(112, 96)
    sc = make_edge_scene(500, cam, seed=33)
    s = oracle_settings(cam, bg=torch.tensor([0.2, 0.4, 0.6]))
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m2 = torch.zeros(sc.P, 3, requires_grad=True)
    color, radii, invd = O.rasterize(leaves[0], m2, leaves[2], s, shs=leaves[1], scales=leaves[3], rotations=leaves[4])
    g = torch.Generator().manual_seed(7)
    wc, wd = torch.randn(3, 96, 112, generator=g), torch.randn(1, 96, 112, generator=g)
    ((color * wc).sum() + (invd * wd).sum()).backward()
    assert res["bounds"][0] == 0 and res["bounds"][-1] == 6
    assert torch.equal(res["radii"], radii)
    assert torch.equal(res["color"], color.detach())          # strips tile the image exactly
    assert torch.equal(res["invd"], invd.detach())
    for a, b in zip(res["grads"], [t.grad for t in leaves] + [m2.grad]):
        # fp32 partial sums are combined in a different order across bands: compare against the gradient's scale
        assert (a - b).abs().max().item() <= 3e-5 * b.abs().max().item()


def _pipeline_worker(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diff_gaussian_rasterization.parallel import BandPlan, gather_strips_async
    H, W, gy = 70, 37, 5                              # last tile row is partial (70 = 4*16 + 6)
    plan = BandPlan([0, 1, 2, 5]) if world == 3 else BandPlan.uniform(gy, world)      # uneven strips incl. a short one
    frames = [torch.arange(3 * H * W, dtype=torch.float32).view(3, H, W) * (f + 1) for f in range(3)]
    a, b = plan.pixel_rows(rank, H)
    handles = []
    for f in frames:                                  # three gathers in flight before the first wait (bench.py keeps two)
        local = torch.full_like(f, -1.0)
        local[:, a:b] = f[:, a:b]
        handles.append(gather_strips_async(local, plan, H))
    outs = [h.wait() for h in handles]
    ok = all(torch.equal(o, f) for o, f in zip(outs, frames)) and torch.equal(handles[0].wait(), frames[0])
    if rank == 0:
        torch.save({"ok": ok}, path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_async_strip_gathers_in_flight(world):
    """Several strip all-gathers issued back to back (the frame pipeline of bench.py --gpus N) complete in order and
    assemble the right frames, with uneven band heights and a partial last tile row."""
    import tempfile
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ok.pt")
        procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert torch.load(path)["ok"]


def _oracle_two_axis(leaves, m2, s, plan, rank, world, P_pad, gy):
    """The two-axis flow with the oracle as both stages: local preprocess -> all_gather_rows (differentiable; backward
    = reduce-scatter) -> band clamp -> bin + blend on the own band -> strips."""
    from diff_gaussian_rasterization.parallel import all_gather_rows, _GatherStrips
    m, sh, o, scl, rot = leaves
    pre = O.preprocess(m, o, s, shs=sh, scales=scl, rotations=rot, means2D=m2)              # full frame
    P = m.shape[0]
    diff = torch.cat([pre["means2D"], pre["conic"], pre["opacity"][:, None], pre["rgb"], pre["depths"][:, None]], dim=1)
    ints = torch.cat([pre["rect"].to(diff.dtype), pre["tiles_touched"][:, None].to(diff.dtype),
                      pre["radii"][:, None].to(diff.dtype)], dim=1)
    rows = torch.cat([diff, ints], dim=1)
    rows = torch.cat([rows, rows.new_zeros(P_pad - P, rows.shape[1])], dim=0)               # padding: no tiles
    allr = all_gather_rows(rows)
    y0, y1 = plan.band(rank)
    rect = allr[:, 10:14].detach().to(torch.int64)
    full_tiles = allr[:, 14].detach().to(torch.int64)
    bminy, bmaxy = rect[:, 1].clamp(y0, y1), rect[:, 3].clamp(y0, y1)
    tiles = torch.where(full_tiles > 0, (rect[:, 2] - rect[:, 0]) * (bmaxy - bminy), torch.zeros_like(full_tiles))
    pre_all = {"means2D": allr[:, 0:2], "conic": allr[:, 2:5], "opacity": allr[:, 5], "rgb": allr[:, 6:9], "depths": allr[:, 9],
               "radii": allr[:, 15].detach().to(torch.int32), "tiles_touched": tiles,
               "rect": torch.stack([rect[:, 0], bminy, rect[:, 2], bmaxy], dim=1), "grid": pre["grid"], "band": (y0, y1)}
    # padding rows have depth 0 -> 1/depth = inf in the oracle's per-Gaussian table; they are never listed
    pre_all["depths"] = torch.where(full_tiles > 0, pre_all["depths"], torch.ones_like(pre_all["depths"]))
    bins = O.bin_and_sort(pre_all)
    color, invd, *_ = O.render_tiles(pre_all, bins, s)
    H = color.shape[1]
    both = _GatherStrips.apply(torch.cat([color, invd], dim=0), plan, H, None)
    return both[:3], pre["radii"], both[3:4]


def _two_axis_worker(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from diff_gaussian_rasterization.parallel import BandPlan, padded_shard_size
    cam = make_camera(112, 96)
    sc = make_edge_scene(500, cam, seed=33)
    s = oracle_settings(cam, bg=torch.tensor([0.2, 0.4, 0.6]))
    gy = 6
    plan = BandPlan([0, 2, 6]) if world == 2 else BandPlan([0, 1, 4, 6])
    cuts = [0, 230, 500] if world == 2 else [0, 100, 333, 500]           # uneven shards -> padding rows
    a, b = cuts[rank], cuts[rank + 1]
    leaves = [t[a:b].clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m2 = torch.zeros(b - a, 3, requires_grad=True)
    P_pad = padded_shard_size(b - a)
    color, radii, invd = _oracle_two_axis(leaves, m2, s, plan, rank, world, P_pad, gy)
    g = torch.Generator().manual_seed(7)
    wc, wd = torch.randn(3, 96, 112, generator=g), torch.randn(1, 96, 112, generator=g)
    ((color * wc).sum() + (invd * wd).sum()).backward()
    torch.save({"color": color.detach(), "invd": invd.detach(), "radii": radii, "P_pad": P_pad, "cut": (a, b),
                "grads": [t.grad for t in leaves] + [m2.grad]}, path % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_axis_sharding_equals_single_device(world):
    """Gaussian-sharded preprocess + record all-gather + band blend + gradient reduce-scatter against the
    single-device oracle: same image on every rank, and each rank's parameter gradients equal its slice of the
    single-device gradients."""
    import tempfile
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rank%d.pt")
        procs = [ctx.Process(target=_two_axis_worker, args=(r, world, port, path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        outs = [torch.load(path % r) for r in range(world)]
    cam = make_camera(112, 96)
    sc = make_edge_scene(500, cam, seed=33)
    s = oracle_settings(cam, bg=torch.tensor([0.2, 0.4, 0.6]))
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m2 = torch.zeros(sc.P, 3, requires_grad=True)
    color, radii, invd = O.rasterize(leaves[0], m2, leaves[2], s, shs=leaves[1], scales=leaves[3], rotations=leaves[4])
    g = torch.Generator().manual_seed(7)
    wc, wd = torch.randn(3, 96, 112, generator=g), torch.randn(1, 96, 112, generator=g)
    ((color * wc).sum() + (invd * wd).sum()).backward()
    ref = [t.grad for t in leaves] + [m2.grad]
    assert outs[0]["P_pad"] == max(o["cut"][1] - o["cut"][0] for o in outs)
    for o in outs:
        a, b = o["cut"]
        assert torch.equal(o["radii"], radii[a:b])
        assert (o["color"] - color.detach()).abs().max().item() <= 3e-6 and (o["invd"] - invd.detach()).abs().max().item() <= 3e-6
        for got, want in zip(o["grads"], ref):
            scale = want.abs().max().item() + 1e-30
            assert (got - want[a:b]).abs().max().item() <= 3e-5 * scale


# ------------------------------------------------------------------------------------------------------------------
# mode C: Gaussian-sharded forward with the destination-targeted all-to-all (parallel.render_gaussian_sharded)
# ------------------------------------------------------------------------------------------------------------------
def _oracle_gaussian_sharded(leaves, m2, s, plan, rank, gid0):
    """The mode-C flow with the oracle as both stages: local preprocess -> route (which band needs which record) ->
    count exchange -> differentiable variable-size all-to-all (backward = the reverse exchange + index_add = the HIP path's
    gsr_route_return) -> band clamp -> bin + blend on the own band -> strips.  Returns the image, the local radii and the
    band's bins expressed in GLOBAL Gaussian ids."""
    from diff_gaussian_rasterization.parallel import exchange_counts, exchange_rows, route_plan_torch, _GatherStrips
    m, sh, o, scl, rot = leaves
    pre = O.preprocess(m, o, s, shs=sh, scales=scl, rotations=rot, means2D=m2)              # full frame
    P = m.shape[0]
    diff = torch.cat([pre["means2D"], pre["conic"], pre["opacity"][:, None], pre["rgb"], pre["depths"][:, None]], dim=1)
    gid = (torch.arange(P, dtype=torch.int64) + gid0).to(diff.dtype)
    ints = torch.cat([pre["rect"].to(diff.dtype), gid[:, None]], dim=1)                       # full-frame rectangle + global id
    rows = torch.cat([diff, ints], dim=1)
    send_index, counts = route_plan_torch(pre["rect"][:, 1], pre["rect"][:, 3], pre["tiles_touched"], plan.bounds)
    send_counts, recv_counts = exchange_counts(torch.tensor(counts, dtype=torch.int64))
    assert send_counts == counts
    recv = exchange_rows(rows.index_select(0, send_index), send_counts, recv_counts)
    y0, y1 = plan.band(rank)
    rect = recv[:, 10:14].detach().to(torch.int64)
    bminy, bmaxy = rect[:, 1].clamp(y0, y1), rect[:, 3].clamp(y0, y1)
    tiles = (rect[:, 2] - rect[:, 0]) * (bmaxy - bminy)
    assert bool((tiles > 0).all())                                                           # only records that touch the band arrive
    pre_band = {"means2D": recv[:, 0:2], "conic": recv[:, 2:5], "opacity": recv[:, 5], "rgb": recv[:, 6:9], "depths": recv[:, 9],
                "tiles_touched": tiles, "rect": torch.stack([rect[:, 0], bminy, rect[:, 2], bmaxy], dim=1),
                "grid": pre["grid"], "band": (y0, y1)}
    bins = O.bin_and_sort(pre_band)
    color, invd, *_ = O.render_tiles(pre_band, bins, s)
    color = color + 0.0 * recv.sum()      # a rank with an empty band must still take part in the backward's collectives
    H = color.shape[1]
    both = _GatherStrips.apply(torch.cat([color, invd], dim=0), plan, H, None)
    gids = recv[:, 14].detach().to(torch.int64)
    return both[:3], pre["radii"], both[3:4], {"ranges": bins["ranges"], "point_gid": gids[bins["point_list"]],
                                                "n_recv": int(recv.shape[0]), "send_counts": send_counts}


def _gaussian_sharded_worker(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from diff_gaussian_rasterization.parallel import BandPlan
    cam = make_camera(112, 96)
    sc = make_edge_scene(500, cam, seed=33)
    s = oracle_settings(cam, bg=torch.tensor([0.2, 0.4, 0.6]))
    plan = {2: BandPlan([0, 2, 6]), 3: BandPlan([0, 1, 4, 6]), 4: BandPlan([0, 1, 3, 3, 6])}[world]      # world 4: an EMPTY band
    cuts = {2: [0, 230, 500], 3: [0, 100, 333, 500], 4: [0, 100, 100, 333, 500]}[world]                   # world 4: an EMPTY shard
    a, b = cuts[rank], cuts[rank + 1]
    leaves = [t[a:b].clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m2 = torch.zeros(b - a, 3, requires_grad=True)
    color, radii, invd, band_bins = _oracle_gaussian_sharded(leaves, m2, s, plan, rank, a)
    g = torch.Generator().manual_seed(7)
    wc, wd = torch.randn(3, 96, 112, generator=g), torch.randn(1, 96, 112, generator=g)
    ((color * wc).sum() + (invd * wd).sum()).backward()
    torch.save({"color": color.detach(), "invd": invd.detach(), "radii": radii, "cut": (a, b), "band": plan.band(rank),
                "bins": band_bins, "grads": [t.grad for t in leaves] + [m2.grad]}, path % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_gaussian_sharded_equals_single_device(world):
    """Mode C: Gaussian-sharded preprocess + destination-targeted all-to-all + band blend + reverse
    exchange of the gradient rows, against the single-device oracle: same image on every rank, each rank's parameter
    gradients equal its slice of the single-device gradients, and every band's bins (tile ranges + point list in global
    Gaussian ids) are BIT-EXACT the single-device bins of that band's tiles.  Uneven shards / bands, an empty shard and an
    empty band (world 4)."""
    import tempfile
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rank%d.pt")
        procs = [ctx.Process(target=_gaussian_sharded_worker, args=(r, world, port, path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        outs = [torch.load(path % r) for r in range(world)]
    cam = make_camera(112, 96)
    sc = make_edge_scene(500, cam, seed=33)
    s = oracle_settings(cam, bg=torch.tensor([0.2, 0.4, 0.6]))
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m2 = torch.zeros(sc.P, 3, requires_grad=True)
    color, radii, invd, aux = O.rasterize(leaves[0], m2, leaves[2], s, shs=leaves[1], scales=leaves[3], rotations=leaves[4],
                                          return_aux=True)
    g = torch.Generator().manual_seed(7)
    wc, wd = torch.randn(3, 96, 112, generator=g), torch.randn(1, 96, 112, generator=g)
    ((color * wc).sum() + (invd * wd).sum()).backward()
    ref = [t.grad for t in leaves] + [m2.grad]
    gx = 7
    visible = int((aux["tiles_touched"] > 0).sum())
    total_sent = sum(sum(o["bins"]["send_counts"]) for o in outs)
    assert visible <= total_sent < world * visible                 # targeted: fewer rows than an all-gather of the visible ones
    for o in outs:
        a, b = o["cut"]
        assert torch.equal(o["radii"], radii[a:b])
        assert (o["color"] - color.detach()).abs().max().item() <= 3e-6 and (o["invd"] - invd.detach()).abs().max().item() <= 3e-6
        for got, want in zip(o["grads"], ref):
            scale = want.abs().max().item() + 1e-30
            assert got.shape == want[a:b].shape
            if b > a:                                   # (world 4 has an empty shard)
                assert (got - want[a:b]).abs().max().item() <= 3e-5 * scale
        # bins of the band, bit-exact: same per-tile counts and the same Gaussians in the same order
        y0, y1 = o["band"]
        for t in range(y0 * gx, y1 * gx):
            ra, rb = (int(v) for v in aux["ranges"][t])
            ba, bb = (int(v) for v in o["bins"]["ranges"][t])
            assert rb - ra == bb - ba
            assert torch.equal(o["bins"]["point_gid"][ba:bb], aux["point_list"][ra:rb])


# ------------------------------------------------------------------------------------------------------------------
# mode C, fixed-capacity form of the exchange (round 4): no count matrix on the host, overflow agreed through an all-reduced flag
# ------------------------------------------------------------------------------------------------------------------
def _fixed_exchange_worker(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from diff_gaussian_rasterization.parallel import (BandPlan, ExchangePolicy, exchange_count_matrix, exchange_rows, exchange_rows_fixed,
                                                     route_plan_torch)
    cam = make_camera(112, 96)
    sc = make_edge_scene(500, cam, seed=33)
    s = oracle_settings(cam)
    plan = {2: BandPlan([0, 2, 6]), 3: BandPlan([0, 1, 4, 6]), 4: BandPlan([0, 1, 3, 3, 6])}[world]
    cuts = {2: [0, 230, 500], 3: [0, 100, 333, 500], 4: [0, 100, 100, 333, 500]}[world]
    a, b = cuts[rank], cuts[rank + 1]
    pre = O.preprocess(sc.means3D[a:b], sc.opacities[a:b], s, shs=sc.shs[a:b], scales=sc.scales[a:b], rotations=sc.rotations[a:b])
    rows = torch.cat([pre["means2D"], pre["conic"], pre["depths"][:, None], (torch.arange(a, b, dtype=torch.float32))[:, None]], dim=1)
    send_index, counts = route_plan_torch(pre["rect"][:, 1], pre["rect"][:, 3], pre["tiles_touched"], plan.bounds)
    send = rows.index_select(0, send_index).clone().requires_grad_(True)
    # exact frame: the policy learns the capacity from the all-gathered matrix (the same number on every rank)
    policy = ExchangePolicy("fixed", slack=1.25, granule=8)
    assert not policy.use_fixed()
    mat = exchange_count_matrix(torch.tensor(counts, dtype=torch.int64))
    policy.observe(int(mat.max()))
    exact = exchange_rows(send, mat[rank].tolist(), mat[:, rank].tolist())
    w = torch.randn(exact.shape, generator=torch.Generator().manual_seed(5 + rank))
    (exact * w).sum().backward()
    g_exact = send.grad.clone()
    send.grad = None
    # fixed frame with the learned capacity: same rows in the same order, same gradient rows back, no overflow
    assert policy.use_fixed() and policy.capacity >= int(mat.max()) and policy.capacity % 8 == 0
    fixed, ovf = exchange_rows_fixed(send, counts, policy.capacity)
    assert not ovf and torch.equal(fixed, exact)
    (fixed * w).sum().backward()
    assert torch.equal(send.grad, g_exact)
    # a capacity below the largest segment: EVERY rank must learn of the overflow, also the ones whose own segments fit
    small = max(1, int(mat.max()) - 1)
    _, ovf2 = exchange_rows_fixed(send.detach(), counts, small)
    own_fits = max(counts + [0]) <= small and int(mat[:, rank].max()) <= small
    torch.save({"capacity": policy.capacity, "max": int(mat.max()), "ovf2": ovf2, "own_fits": own_fits}, path % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_fixed_capacity_exchange_equals_exact_and_agrees_on_overflow(world):
    """The fixed-capacity exchange (segments of capacity + 1 rows with the count in a header row, equal-split all-to-all) delivers
    exactly the rows of the variable-size exchange, in the same order, forward and backward; the capacity a policy learns from the
    count matrix is the same on every rank; an overflow anywhere is seen by every rank (uneven shards / bands, an empty shard and
    an empty band at world 4)."""
    import tempfile
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rank%d.pt")
        procs = [ctx.Process(target=_fixed_exchange_worker, args=(r, world, port, path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        outs = [torch.load(path % r) for r in range(world)]
    assert len({o["capacity"] for o in outs}) == 1 and len({o["max"] for o in outs}) == 1
    assert all(o["ovf2"] for o in outs)
    assert any(o["own_fits"] for o in outs) or world == 2      # some rank only knows through the all-reduce


def test_exchange_policy_capacity_only_grows():
    from diff_gaussian_rasterization.parallel import ExchangePolicy
    p = ExchangePolicy("fixed", slack=1.25, granule=256)
    assert not p.use_fixed()
    p.observe(1000)
    assert p.capacity == 1280 and p.use_fixed()
    p.observe(10)
    assert p.capacity == 1280
    p.observe(2000)
    assert p.capacity == 2560
    q = ExchangePolicy("exact")
    q.observe(5)
    assert not q.use_fixed()
    with pytest.raises(ValueError):
        ExchangePolicy("sometimes")
