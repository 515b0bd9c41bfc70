"""The multi-GPU design's three modes as SHIPPED -- `diff_gaussian_rasterization.parallel` (band plan, autograd Functions, exchanges) over the real C ABI
(gsr_preprocess_forward, gsr_route_count / _pack / _pack_fixed / _return, gsr_rasterize_from_packed / _from_segments / _from_splats, gsr_backward_blend,
gsr_backward_preprocess) and the kernels behind it -- in world_size 2 and 3 `gloo` processes on the CPU: the library is the host build of the kernel
source (tests/simt, tests/_build/libgsr_simt.so; every lane a fiber), found through the package's own loader via GSR_LIB.  tests/test_parallel_gloo.py
covers the same flows with the oracle standing in for the kernels; here nothing stands in but the GPU.

  A  render_sharded + hip_band_renderer: every rank holds every Gaussian and blends a band; the [P,12] gradient records are all-reduced.
  B  render_two_axis: Gaussian shards -> record all-gather -> band blend -> gradient reduce-scatter.
  C  render_gaussian_sharded: Gaussian shards -> route -> destination-targeted all-to-all of packed records -> band blend -> reverse exchange; the exact
     form and the fixed-capacity form (first frame exact, second fixed; then a capacity that overflows and must fall back).

Each rank's image must be the single-device oracle's image and its gradients its slice of the single-device gradients, at the bars of the GPU suite.
Test infrastructure: the product never loads this library."""
import contextlib
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import O, make_camera, make_edge_scene, oracle_settings
import simt_build
from test_parallel_gloo import _free_port

W, H, GY, P = 112, 96, 6, 500
PLANS = {2: [0, 2, 6], 3: [0, 1, 4, 6]}
CUTS = {2: [0, 230, 500], 3: [0, 100, 333, 500]}


def _inputs():
    cam = make_camera(W, H)
    sc = make_edge_scene(P, cam, seed=33)
    s = oracle_settings(cam, bg=torch.tensor([0.2, 0.4, 0.6]))
    g = torch.Generator().manual_seed(7)
    return cam, sc, s, torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g)


def _worker(rank, world, port, lib_path, path):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"], os.environ["GSR_LIB"] = "127.0.0.1", str(port), lib_path
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import diff_gaussian_rasterization as pkg
    from diff_gaussian_rasterization import _lib, parallel
    from test_gpu_parity import gpu_settings
    # the three places that ask torch for a HIP device, and the pinned host word of the fixed exchange: host memory here
    pkg._require_cuda, pkg._stream_ptr = (lambda *a: None), (lambda d: None)
    torch.cuda.device = lambda d: contextlib.nullcontext()
    parallel._pinned_flag = lambda device: torch.zeros(1, dtype=torch.int32)
    assert _lib.load()._name == lib_path
    cam, sc, s, wc, wd = _inputs()
    rs = gpu_settings(s, torch.device("cpu"))
    plan = parallel.BandPlan(PLANS[world])
    a, b = CUTS[world][rank], CUTS[world][rank + 1]
    out = {"cut": (a, b), "band": plan.band(rank)}

    def leaves(lo, hi):
        return [t[lo:hi].clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]

    def finish(name, color, radii, invd, L, m2=None):
        ((color * wc).sum() + (invd * wd).sum()).backward()
        out[name] = {"color": color.detach(), "invd": invd.detach(), "radii": radii, "grads": [t.grad for t in L] + ([m2.grad] if m2 is not None else [])}

    # A: all Gaussians on every rank
    L = leaves(0, P)
    finish("A", *parallel.render_sharded(parallel.hip_band_renderer(rs), L, plan, reduce="records"), L)
    # B: two-axis
    L, m2 = leaves(a, b), torch.zeros(b - a, 3, requires_grad=True)
    out["P_pad"] = parallel.padded_shard_size(b - a)
    finish("B", *parallel.render_two_axis(rs, *L, plan, out["P_pad"], means2D=m2), L, m2)
    # C, exact exchange
    L, m2 = leaves(a, b), torch.zeros(b - a, 3, requires_grad=True)
    finish("C", *parallel.render_gaussian_sharded(rs, *L, plan, means2D=m2), L, m2)
    # C, fixed-capacity exchange: the first frame is exact and teaches the capacity, the second runs the fixed form
    policy = parallel.set_exchange_mode("fixed", granule=16)
    with torch.no_grad():
        parallel.render_gaussian_sharded(rs, *[t.detach() for t in L], plan)
    assert policy.frames_exact == 1 and policy.frames_fixed == 0 and policy.capacity is not None
    L, m2 = leaves(a, b), torch.zeros(b - a, 3, requires_grad=True)
    finish("C_fixed", *parallel.render_gaussian_sharded(rs, *L, plan, means2D=m2), L, m2)
    assert policy.frames_fixed == 1 and policy.overflows == 0
    out["capacity"] = int(policy.capacity)
    # ... and a capacity that is too small somewhere: every rank sees the flag, all repeat the frame in the exact form
    policy.capacity = 16
    L, m2 = leaves(a, b), torch.zeros(b - a, 3, requires_grad=True)
    finish("C_overflow", *parallel.render_gaussian_sharded(rs, *L, plan, means2D=m2), L, m2)
    out["overflows"], out["capacity_after"] = policy.overflows, int(policy.capacity)
    # forward-only pipelined form (bench.py's sharded forward): begin -> finish -> wait
    parallel.set_exchange_mode("exact")
    fr = parallel.sharded_forward_begin(rs, *[t.detach() for t in L], plan)
    out["pipelined"] = parallel.sharded_forward_finish(fr).wait()[:3].clone()
    torch.save(out, path % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_three_sharding_modes_with_the_real_kernels_equal_the_single_device_oracle(world):
    lib_path = simt_build.build_library()
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rank%d.pt")
        procs = [ctx.Process(target=_worker, args=(r, world, port, lib_path, path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(900)
            assert p.exitcode == 0
        outs = [torch.load(path % r) for r in range(world)]
    cam, sc, s, wc, wd = _inputs()
    L = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m2 = torch.zeros(P, 3, requires_grad=True)
    color, radii, invd = O.rasterize(L[0], m2, L[2], s, shs=L[1], scales=L[3], rotations=L[4])
    ((color * wc).sum() + (invd * wd).sum()).backward()
    ref = [t.grad for t in L] + [m2.grad]
    color, invd, radii = color.detach(), invd.detach(), radii.to(torch.int32)

    def image_ok(o, name):
        err = (o["color"] - color).abs().amax(0)      # (tests/test_gpu_parity.py's bars: a blend threshold within rounding noise may flip a pixel)
        assert float((err > 1e-5).float().mean()) < 0.01 and float(err.max()) < 1.1 / 255.0, name
        assert float(((o["invd"] - invd).abs() > 1e-5).float().mean()) < 0.01, name

    def grads_ok(o, name, lo, hi, n):
        for k, (got, want) in enumerate(zip(o["grads"][:n], ref)):
            scale = float(want.abs().max())
            assert got.shape == want[lo:hi].shape and scale > 0, (name, k)
            d = (got - want[lo:hi]).abs() / scale
            assert float(d.max()) < 2e-3 and float(torch.quantile(d.flatten()[:2_000_000], 0.999)) < 1e-4, (name, k, float(d.max()))

    assert outs[0]["P_pad"] == max(o["cut"][1] - o["cut"][0] for o in outs)
    for r, o in enumerate(outs):
        a, b = o["cut"]
        assert torch.equal(o["A"]["radii"].to(torch.int32), radii)
        image_ok(o["A"], "A")
        grads_ok(o["A"], "A", 0, P, 5)
        for name in ("B", "C", "C_fixed", "C_overflow"):
            assert torch.equal(o[name]["radii"].to(torch.int32), radii[a:b]), name
            image_ok(o[name], name)
            grads_ok(o[name], name, a, b, 6)
        # every rank holds the SAME image (strips are gathered, not recomputed), in every mode
        for name in ("A", "B", "C", "C_fixed", "C_overflow"):
            assert torch.equal(o[name]["color"], outs[0][name]["color"]) and torch.equal(o[name]["invd"], outs[0][name]["invd"]), name
        assert torch.equal(o["pipelined"], outs[0]["C"]["color"])
        # the fixed form is the exact form's arithmetic: same band bins, same blend -> same bits
        assert torch.equal(o["C_fixed"]["color"], o["C"]["color"]) and torch.equal(o["C_overflow"]["color"], o["C"]["color"])
        for x, y in zip(o["C_fixed"]["grads"], o["C"]["grads"]):
            assert torch.equal(x, y)
        assert o["overflows"] == 1 and o["capacity_after"] >= o["capacity"] > 16
