"""The operator's default FORWARD and BACKWARD, from the product's kernel source, against the oracle -- in the build container, without a GPU.
csrc/preprocess.hip (+ its frame statistics), depthsort.hip, tilesort.hip and render_fwd.hip are compiled with g++ against the SIMT shim of
tests/simt/ (fibers, wave64 ballots, DPP moves, readlane, LDS, barriers; -ffp-contract=off) and run through the product's own launchers in the
order of gsr_rasterize_forward: eleven launches, every lane executed.  The scenes, the oracle call and the bars are those of the GPU parity suite
(`tests/test_gpu_parity.py::test_forward_parity`: radii, tiles_touched, R, sorted point list, tile ranges and n_contrib bit-exact, image /
inverse depth 1e-5 off the oracle's fragile pixels, final_T 5e-6), both builds of the blend (tracking / inference) and both rectangle modes.
The backward -- blend backward (plan kernel, the two-pixels-per-lane walk, the three reduce kernels) + the fused per-Gaussian backward -- is held
to the bars of the GPU suite's `_backward_case` against autograd through the oracle.
Differences to the GPU build: no FMA contraction in the blend kernels, libm's exp2f for v_exp_f32, an exact 1/x for v_rcp_f32 -- inside the same bars.

Test infrastructure: a checker of the kernel SOURCE (tests/_build/libsimt_forward.so is never shipped); the product has no CPU path."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import reference_tiles
import test_gpu_parity as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libsimt_forward.so")
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


@pytest.fixture(scope="module")
def lib():
    from simt_build import build
    h = build("forward", fp_contract_off=True)
    h.simt_fwd_last_error.restype = C.c_char_p
    h.simt_forward.restype = C.c_int64
    return h


def f32(t):
    return np.ascontiguousarray(t.detach().to(torch.float32).numpy())


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def run_simt(lib, s, sc, track, snug=1, colors=None):
    from diff_gaussian_rasterization._lib import GsrRasterSettings
    H, W = int(s.image_height), int(s.image_width)
    bg, view, proj, campos = f32(s.bg), f32(s.viewmatrix), f32(s.projmatrix), f32(s.campos)
    rs = GsrRasterSettings(H, W, float(s.tanfovx), float(s.tanfovy), ptr(bg).value, float(s.scale_modifier), ptr(view).value, ptr(proj).value,
                           int(s.sh_degree), ptr(campos).value, 0, 0, 1 if s.antialiasing else 0, 0, 0, 0 if track else 1, None, None)
    P = sc.P
    m, op, scl, rot = f32(sc.means3D), f32(sc.opacities), f32(sc.scales), f32(sc.rotations)
    shs = None if colors is not None else f32(sc.shs)
    col_in = f32(colors) if colors is not None else None
    M = 0 if colors is not None else sc.shs.shape[1]
    radii = np.zeros(P, dtype=np.int32)
    tiles = np.zeros(P, dtype=np.uint32)
    color = np.zeros((3, H, W), dtype=np.float32)
    invd = np.zeros((1, H, W), dtype=np.float32)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    r_cap = 4_000_000
    point_list = np.zeros(r_cap, dtype=np.uint32)
    ranges = np.zeros((gx * gy, 2), dtype=np.uint32)
    final_T = np.zeros((H, W), dtype=np.float32)
    n_contrib = np.zeros((H, W), dtype=np.uint32)
    R = lib.simt_forward(C.byref(rs), snug, P, M, ptr(m), ptr(shs), ptr(col_in), ptr(op), ptr(scl), ptr(rot), ptr(radii), ptr(tiles), ptr(color),
                         ptr(invd), ptr(point_list), C.c_int64(r_cap), ptr(ranges), 1 if track else 0, ptr(final_T), ptr(n_contrib),
                         None, None, None, None, None, None, None, None)      # (no backward)
    assert R >= 0, lib.simt_fwd_last_error()
    out = {"radii": torch.from_numpy(radii), "tiles_touched": torch.from_numpy(tiles.astype(np.int64)), "R": int(R),
           "point_list": torch.from_numpy(point_list[:R].astype(np.int64)), "ranges": torch.from_numpy(ranges.astype(np.int64)),
           "color": torch.from_numpy(color), "invdepth": torch.from_numpy(invd)}
    if track:
        out["n_contrib"] = torch.from_numpy(n_contrib.astype(np.int64))
        out["final_T"] = torch.from_numpy(final_T)
    return out


@pytest.mark.parametrize("track", [True, False], ids=["track", "inference"])
@pytest.mark.parametrize("name", ["c1", "odd_aa", "edge_lookat", "edge_aa_scale", "deg1", "deg0_dense"])
def test_forward_of_the_kernel_source_on_the_cpu_against_the_oracle(lib, name, track):
    cam, sc, opts = G.mk(name)
    s, col, radii, invd, aux = G.run_oracle(cam, sc, opts)
    assert (radii > 0).sum() > 100
    out = run_simt(lib, s, sc, track)
    G.check_forward(s, col, radii, invd, aux, out)


def test_forward_of_the_kernel_source_with_the_reference_rectangles_and_precomputed_colours(lib):
    """snug = 0: the reference's own tile squares (Appendix A.2 step 8) -- bins against the oracle in reference mode; colors_precomp call form."""
    cam, sc, opts = G.mk("edge_lookat")
    colors = torch.rand(sc.P, 3, generator=torch.Generator().manual_seed(1))
    with reference_tiles():
        s, col, radii, invd, aux = G.run_oracle(cam, sc, opts, colors=colors)
    out = run_simt(lib, s, sc, True, snug=0, colors=colors)
    G.check_forward(s, col, radii, invd, aux, out)


def run_simt_backward(lib, s, sc, H, W, wc, wd, use_depth):
    """forward (tracking build) + blend backward + per-Gaussian backward of the kernel source -> (radii, {input name: gradient array})"""
    from diff_gaussian_rasterization._lib import GsrRasterSettings
    bg, view, proj, campos = f32(s.bg), f32(s.viewmatrix), f32(s.projmatrix), f32(s.campos)
    rs = GsrRasterSettings(H, W, float(s.tanfovx), float(s.tanfovy), ptr(bg).value, float(s.scale_modifier), ptr(view).value, ptr(proj).value,
                           int(s.sh_degree), ptr(campos).value, 0, 0, 1 if s.antialiasing else 0, 0, 0, 0, None, None)
    P, M = sc.P, sc.shs.shape[1]
    m, op, scl, rot, shs = f32(sc.means3D), f32(sc.opacities), f32(sc.scales), f32(sc.rotations), f32(sc.shs)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    r_cap = 4_000_000
    bufs = dict(radii=np.zeros(P, np.int32), tiles=np.zeros(P, np.uint32), color=np.zeros((3, H, W), np.float32), invd=np.zeros((1, H, W), np.float32),
                pl=np.zeros(r_cap, np.uint32), ranges=np.zeros((gx * gy, 2), np.uint32), final_T=np.zeros((H, W), np.float32), n_contrib=np.zeros((H, W), np.uint32))
    g = dict(means2D=np.zeros((P, 3), np.float32), means3D=np.zeros((P, 3), np.float32), opacities=np.zeros((P, 1), np.float32),
             shs=np.zeros((P, M, 3), np.float32), scales=np.zeros((P, 3), np.float32), rotations=np.zeros((P, 4), np.float32))
    dcol, dinv = f32(wc), (f32(wd) if use_depth else None)
    R = lib.simt_forward(C.byref(rs), 1, P, M, ptr(m), ptr(shs), None, ptr(op), ptr(scl), ptr(rot), ptr(bufs["radii"]), ptr(bufs["tiles"]), ptr(bufs["color"]),
                         ptr(bufs["invd"]), ptr(bufs["pl"]), C.c_int64(r_cap), ptr(bufs["ranges"]), 1, ptr(bufs["final_T"]), ptr(bufs["n_contrib"]),
                         ptr(dcol), ptr(dinv), ptr(g["means2D"]), ptr(g["means3D"]), ptr(g["opacities"]), ptr(g["shs"]), ptr(g["scales"]), ptr(g["rotations"]))
    assert R > 0, lib.simt_fwd_last_error()
    return bufs["radii"], g


@pytest.mark.parametrize("name,n,seed,use_depth", [("c1", 1000, 0, True), ("edge_aa_scale", 700, 3, True), ("odd_aa", 1200, 4, False)])
def test_backward_of_the_kernel_source_on_the_cpu_against_the_oracles_autograd(lib, name, n, seed, use_depth):
    """The blend backward (plan kernel, the two-pixels-per-lane walk with its DPP / permlane transpose-reduce, the three reduce kernels) and the
    fused per-Gaussian backward, run from the kernel source through the shim, against autograd through the oracle: the scenes, the loss and the
    bars of tests/test_gpu_parity.py::_backward_case (max 1e-4, 99.9th percentile 1e-5 of max |grad|), with and without a gradient on the
    inverse-depth image (the HAS_DEPTH build of the walk)."""
    import copy
    from helpers import O, oracle_settings
    cam, sc, opts = G.mk(name)
    idx = torch.arange(min(n, sc.P))
    sc = copy.copy(sc)
    sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs = sc.means3D[idx], sc.scales[idx], sc.rotations[idx], sc.opacities[idx], sc.shs[idx]
    s = oracle_settings(cam, bg=opts.get("bg"), sh_degree=opts.get("sh_degree", 3), scale_modifier=opts.get("scale_modifier", 1.0),
                        antialiasing=opts.get("antialiasing", False))
    H, W = cam.image_height, cam.image_width
    wc, wd = G._loss_weights(H, W, seed)
    L = {k: v.detach().clone().requires_grad_(True) for k, v in dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations).items()}
    L["means2D"] = torch.zeros(sc.P, 3, requires_grad=True)
    col, radii, invd = O.rasterize(L["means3D"], L["means2D"], L["opacities"], s, shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
    ((col * wc).sum() + ((invd * wd).sum() if use_depth else 0.0)).backward()
    radii_k, g = run_simt_backward(lib, s, sc, H, W, wc, wd, use_depth)
    assert np.array_equal(radii_k, radii.numpy().astype(np.int32))
    for k in L:
        a, b = torch.from_numpy(g[k]).double(), L[k].grad.double()
        scale = b.abs().max().item() + 1e-30
        d = (a - b).abs() / scale
        assert b.abs().max().item() > 0, f"{k}: oracle gradient is identically zero"
        assert d.max().item() < 1e-4, f"{k}: max err {d.max().item():.3e} (rel. to max |grad|)"
        assert torch.quantile(d.flatten()[:4_000_000], 0.999).item() < 1e-5, f"{k}: 99.9th pct err too large"


@pytest.mark.parametrize("name,bounds", [("c1", [0, 5, 11, 16]), ("edge_aa_scale", [0, 3, 3, 9]), ("odd_aa", [0, 9]), ("c1", [0, 1, 2, 3, 16])])
def test_gaussian_sharded_forward_of_the_kernel_source_equals_the_single_device_forward(lib, name, bounds):
    """Mode C of the multi-GPU renderer (parallel.py; no multi-GPU hardware in any round): G contiguous shards of Gaussians and G bands of tile
    rows, every rank's kernels run one after the other -- per-Gaussian kernel with full-frame rectangles, route_count + scan, the stable
    48-byte pack, then per band the records of all shards in rank order through ingest_packed (which recomputes tau and 1 / depth), depth sort,
    tile sort and the blend of the band's rows.  The assembled frame must be THE SAME BITS as the single-device forward of the same source,
    the per-(shard, band) counts those of parallel.route_plan_torch; an empty band (bounds 3, 3) shows the background."""
    from diff_gaussian_rasterization._lib import GsrRasterSettings
    from diff_gaussian_rasterization.parallel import route_plan_torch
    from helpers import O
    cam, sc, opts = G.mk(name)
    s, col, radii, invd, aux = G.run_oracle(cam, sc, opts)
    H, W = int(s.image_height), int(s.image_width)
    gy = (H + 15) // 16
    bounds = [min(b, gy) for b in bounds[:-1]] + [gy]
    single = run_simt(lib, s, sc, False)
    bg, view, proj, campos = f32(s.bg), f32(s.viewmatrix), f32(s.projmatrix), f32(s.campos)
    rs = GsrRasterSettings(H, W, float(s.tanfovx), float(s.tanfovy), ptr(bg).value, float(s.scale_modifier), ptr(view).value, ptr(proj).value,
                           int(s.sh_degree), ptr(campos).value, 0, 0, 1 if s.antialiasing else 0, 0, 0, 1, None, None)
    P, M = sc.P, sc.shs.shape[1]
    m, op, scl, rot, shs = f32(sc.means3D), f32(sc.opacities), f32(sc.scales), f32(sc.rotations), f32(sc.shs)
    ng = len(bounds) - 1
    b32 = np.array(bounds, dtype=np.int32)
    radii_s = np.zeros(P, dtype=np.int32)
    color = np.zeros((3, H, W), dtype=np.float32)
    invd_s = np.zeros((1, H, W), dtype=np.float32)
    counts = np.zeros(ng * ng, dtype=np.uint32)
    lib.simt_forward_sharded.restype = C.c_int64
    R = lib.simt_forward_sharded(C.byref(rs), 1, ng, ptr(b32), P, M, ptr(m), ptr(shs), ptr(op), ptr(scl), ptr(rot), ptr(radii_s), ptr(color), ptr(invd_s), ptr(counts))
    assert R >= 0, lib.simt_fwd_last_error()
    assert np.array_equal(radii_s, single["radii"].numpy())
    assert np.array_equal(color, single["color"].numpy()), "sharded frame differs from the single-device frame"
    assert np.array_equal(invd_s, single["invdepth"].numpy())
    # the counts of the exchange = the torch restatement on the oracle's full-frame rectangles
    with torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    for g in range(ng):
        lo, hi = P * g // ng, P * (g + 1) // ng
        _, want = route_plan_torch(pre["rect"][lo:hi, 1], pre["rect"][lo:hi, 3], pre["tiles_touched"][lo:hi], bounds)
        assert counts[g * ng:(g + 1) * ng].tolist() == want, (g, counts[g * ng:(g + 1) * ng].tolist(), want)


def test_separate_sh_call_form_of_the_kernel_source_equals_the_fused_form(lib):
    """The reference's accelerated call form (gaussian_renderer/__init__.py:82-100: dc [P,1,3] and rest [P,15,3] as separate tensors) through the
    split-SH instantiations of the per-Gaussian kernels: same image and radii as the fused [P,16,3] tensor bit for bit, the same gradients, with
    dL/ddc and dL/drest the two parts of dL/dshs."""
    from diff_gaussian_rasterization._lib import GsrRasterSettings
    from helpers import oracle_settings
    cam, sc, opts = G.mk("c1")
    s = oracle_settings(cam)
    H, W = cam.image_height, cam.image_width
    wc, _ = G._loss_weights(H, W, 7)
    bg, view, proj, campos = f32(s.bg), f32(s.viewmatrix), f32(s.projmatrix), f32(s.campos)
    rs = GsrRasterSettings(H, W, float(s.tanfovx), float(s.tanfovy), ptr(bg).value, float(s.scale_modifier), ptr(view).value, ptr(proj).value,
                           int(s.sh_degree), ptr(campos).value, 0, 0, 0, 0, 0, 0, None, None)
    P, M = sc.P, 16
    m, op, scl, rot, shs = f32(sc.means3D), f32(sc.opacities), f32(sc.scales), f32(sc.rotations), f32(sc.shs)
    dc, rest = np.ascontiguousarray(shs[:, :1]), np.ascontiguousarray(shs[:, 1:])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    r_cap = 4_000_000
    dcol = f32(wc)

    def run(split):
        b = dict(radii=np.zeros(P, np.int32), tiles=np.zeros(P, np.uint32), color=np.zeros((3, H, W), np.float32), invd=np.zeros((1, H, W), np.float32),
                 pl=np.zeros(r_cap, np.uint32), ranges=np.zeros((gx * gy, 2), np.uint32), final_T=np.zeros((H, W), np.float32), n_contrib=np.zeros((H, W), np.uint32))
        g = dict(means2D=np.zeros((P, 3), np.float32), means3D=np.zeros((P, 3), np.float32), opacities=np.zeros((P, 1), np.float32),
                 shs=np.zeros((P, 15 if split else 16, 3), np.float32), scales=np.zeros((P, 3), np.float32), rotations=np.zeros((P, 4), np.float32))
        ddc = np.zeros((P, 1, 3), np.float32)
        lib.simt_set_split_sh(ptr(dc) if split else None, ptr(ddc) if split else None)
        try:
            R = lib.simt_forward(C.byref(rs), 1, P, M, ptr(m), ptr(rest if split else shs), None, ptr(op), ptr(scl), ptr(rot), ptr(b["radii"]), ptr(b["tiles"]),
                                 ptr(b["color"]), ptr(b["invd"]), ptr(b["pl"]), C.c_int64(r_cap), ptr(b["ranges"]), 1, ptr(b["final_T"]), ptr(b["n_contrib"]),
                                 ptr(dcol), None, ptr(g["means2D"]), ptr(g["means3D"]), ptr(g["opacities"]), ptr(g["shs"]), ptr(g["scales"]), ptr(g["rotations"]))
        finally:
            lib.simt_set_split_sh(None, None)
        assert R > 0, lib.simt_fwd_last_error()
        return b, g, ddc

    bf, gf, _ = run(False)
    bs, gs, ddc = run(True)
    assert np.array_equal(bf["color"], bs["color"]) and np.array_equal(bf["radii"], bs["radii"]) and np.array_equal(bf["invd"], bs["invd"])
    for k in ("means2D", "means3D", "opacities", "scales", "rotations"):
        assert np.array_equal(gf[k], gs[k]), k
    assert np.array_equal(gf["shs"][:, :1], ddc) and np.array_equal(gf["shs"][:, 1:], gs["shs"])
    assert np.abs(gf["shs"]).max() > 0
