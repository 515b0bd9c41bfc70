"""GPU parity at the BASELINE.json sizes, against the CPU oracle (not just invariants):

  configs[1]  1 M Gaussians @1920x1080  -- exactly the frame bench.py times (seed 0, s_med 0.012)
  configs[3]  1 M Gaussians @3840x2160
  configs[4]  6 M Gaussians @1920x1080

For every config, BOTH builds of the forward (TRACK = the training build, and the INFERENCE build that
`torch.no_grad()` / bench.py's forward metric run): radii, tiles_touched, R, the sorted point list and the tile
ranges are compared BIT-EXACT with `O.preprocess` + `O.bin_and_sort` on ALL Gaussians; the image / inverse depth
(and n_contrib / final_T in the TRACK build) are compared on a sample of busy tiles blended by the oracle
(`O.render_tiles(tiles=...)`), bar 1e-5 outside the oracle's "fragile" pixels (a hard blend threshold within rounding
noise), one alpha quantum inside.  configs[1] additionally gets the BACKWARD: dL/dpixel is put on sampled tiles only
and every input gradient is compared with the oracle's autograd through the same tiles.

The measured numbers (max error, fragile fraction, gradient errors) are printed and collected into
gpurun_out/parity_report.json (copied to profiles/ by the builder), so the tolerances are visible, not hidden in asserts.
Reference boundary being matched: gaussian_renderer/__init__.py:91-110 (forward), train.py:142 (backward).
"""
import json
import os
import time

import pytest
import torch

from helpers import O, ROOT, make_camera, make_scene, make_clustered_scene, oracle_settings, parity_report, reference_tiles

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-5


_report = parity_report


def _gpu_settings(s, dev, debug=False):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.bg.to(dev),
                                         s.scale_modifier, s.viewmatrix.to(dev), s.projmatrix.to(dev), s.sh_degree,
                                         s.campos.to(dev), False, debug, s.antialiasing)


_cache = {}


def _config(P, W, H, kind="uniform"):
    """Scene + oracle bins of a config (cached across the tests of this module: the 6 M scene takes a while to make)."""
    key = (P, W, H, kind)
    if key not in _cache:
        _cache.clear()             # keep one config resident at a time (the 6 M scene + its bins are several GB)
        t0 = time.perf_counter()
        cam = make_camera(W, H)
        # bench.py's generator calls
        sc = make_clustered_scene(P, cam, seed=0, s_med=0.012) if kind == "clustered" else make_scene(P, cam, seed=0, s_med=0.012)
        s = oracle_settings(cam)
        with torch.no_grad():
            pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
            bins = O.bin_and_sort(pre)
        _cache[key] = (cam, sc, s, pre, bins, time.perf_counter() - t0)
    return _cache[key]


def _busy_sample(bins, n):
    busy = torch.nonzero(bins["tile_counts"] > 0).flatten()
    step = max(1, busy.numel() // n)
    return busy[::step][:n].tolist()


def _check_bins(out, pre, bins):
    assert torch.equal(out["radii"].cpu(), pre["radii"].to(torch.int32)), "radii differ"
    assert torch.equal(out["tiles_touched"].cpu().to(torch.int64), pre["tiles_touched"]), "tiles_touched differ"
    assert out["R"] == bins["R"], f"R {out['R']} != {bins['R']}"
    assert torch.equal(out["point_list"].cpu().to(torch.int64), bins["point_list"]), "sorted point list differs"
    assert torch.equal(out["ranges"].cpu().to(torch.int64), bins["ranges"]), "tile ranges differ"


def _check_sampled_tiles(out, pre, bins, s, sample, W, H):
    """Image / invdepth (+ n_contrib, final_T when present) on the sampled tiles.  Returns the measured numbers."""
    gx = pre["grid"][0]
    with torch.no_grad():
        col, invd, fT, ncon, frag = O.render_tiles(pre, bins, s, tiles=sample, want_fragile=True)
    gcol, ginv = out["color"].cpu(), out["invdepth"].cpu()
    gnc = out["n_contrib"].cpu().to(torch.int64) if "n_contrib" in out else None
    gft = out["final_T"].cpu() if "final_T" in out else None
    cmax = max(1.0, float(pre["rgb"].abs().max()))
    max_ok, max_frag, n_frag, n_pix, max_T = 0.0, 0.0, 0, 0, 0.0
    for t in sample:
        y0, x0 = (t // gx) * 16, (t % gx) * 16
        ys, xs = slice(y0, min(y0 + 16, H)), slice(x0, min(x0 + 16, W))
        fr = frag[ys, xs]
        ok = ~fr
        d = (gcol[:, ys, xs] - col[:, ys, xs]).abs().max(0).values
        di = (ginv[0, ys, xs] - invd[0, ys, xs]).abs()
        n_pix += fr.numel()
        n_frag += int(fr.sum())
        if ok.any():
            max_ok = max(max_ok, float(d[ok].max()), float(di[ok].max()))
        if fr.any():
            max_frag = max(max_frag, float(d[fr].max()))
        if gnc is not None:
            assert torch.equal(gnc[ys, xs][ok], ncon[ys, xs][ok]), f"n_contrib differs on tile {t}"
            max_T = max(max_T, float((gft[ys, xs] - fT[ys, xs]).abs()[ok].max()) if ok.any() else 0.0)
    assert max_ok <= IMG_TOL, f"image error {max_ok:.3e} on non-fragile pixels"
    assert max_frag <= cmax / 255.0 * 1.01 + IMG_TOL, f"fragile-pixel error {max_frag:.3e} exceeds one alpha quantum"
    assert n_frag / max(1, n_pix) < 0.02, "too many fragile pixels for the exclusion to be meaningful"
    assert max_T <= 5e-6
    return {"tiles": len(sample), "pixels": n_pix, "max_err_nonfragile": max_ok, "fragile_fraction": n_frag / max(1, n_pix),
            "max_err_fragile": max_frag, "max_final_T_err": max_T}


def _forward_case(P, W, H, n_tiles, name, kind="uniform"):
    from diff_gaussian_rasterization import rasterize_gaussians
    from diff_gaussian_rasterization.debug import forward_with_views
    dev = torch.device("cuda:0")
    cam, sc, s, pre, bins, t_oracle = _config(P, W, H, kind)
    d = sc.to(dev)
    rs = _gpu_settings(s, dev)
    sample = _busy_sample(bins, n_tiles)
    images = {}
    for build, nb in (("track", False), ("inference", True)):
        out = forward_with_views(rs, d.means3D, d.opacities, shs=d.shs, scales=d.scales, rotations=d.rotations, no_backward=nb)
        torch.cuda.synchronize()
        _check_bins(out, pre, bins)
        m = _check_sampled_tiles(out, pre, bins, s, sample, W, H)
        images[build] = out["color"].clone()
        _report(f"{name}/forward/{build}", P=P, W=W, H=H, visible=int((pre["radii"] > 0).sum()), R=bins["R"],
                bins="bit-exact (radii, tiles_touched, R, point_list, ranges on all Gaussians)", **m)
        del out
    # the two builds produce the same image bit for bit, and the operator under torch.no_grad() (bench.py's timed call) is
    # the inference build
    assert torch.equal(images["track"], images["inference"]), "inference build image != tracking build image"
    with torch.no_grad():
        color, radii, invd = rasterize_gaussians(d.means3D, None, d.shs, None, d.opacities, d.scales, d.rotations, None, rs, None)
    torch.cuda.synchronize()
    assert torch.equal(color, images["inference"])
    assert torch.equal(radii.cpu(), pre["radii"].to(torch.int32))


def test_config1_1M_1080p_forward_both_builds():
    """BASELINE configs[1] stand-in, exactly as bench.py builds it (P = 1e6, 1920x1080, seed 0, s_med 0.012)."""
    _forward_case(1_000_000, 1920, 1080, 240, "configs[1] 1M@1080p")
    cam, sc, s, pre, bins, _ = _config(1_000_000, 1920, 1080)
    # SURVEY 8(d) probe numbers of this generator (they identify the frame the bench line is quoted on): 11 330 172 instances in
    # the reference's tile squares, 7 916 318 of them in the snug rectangles the product bins
    assert int((pre["radii"] > 0).sum()) == 876281 and bins["R"] == 7916318
    with torch.no_grad():
        ref = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations, snug=False)
    assert int(ref["tiles_touched"].sum()) == 11330172 and torch.equal(ref["radii"], pre["radii"])


def _backward_case(P, W, H, n_tiles, name, kind="uniform"):
    """Backward at a BASELINE size: dL/dpixel (colour and inverse depth) is non-zero on sampled busy tiles only; every
    input gradient of the HIP backward is compared with the oracle's autograd through O.render_tiles(tiles=sample)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    cam, sc, s, pre0, bins, _ = _config(P, W, H, kind)
    gx = pre0["grid"][0]
    sample = _busy_sample(bins, n_tiles)
    mask = torch.zeros(H, W, dtype=torch.bool)
    for t in sample:
        y0, x0 = (t // gx) * 16, (t % gx) * 16
        mask[y0:y0 + 16, x0:x0 + 16] = True
    g = torch.Generator().manual_seed(11)
    wc = torch.randn(3, H, W, generator=g) * mask
    wd = torch.randn(1, H, W, generator=g) * 0.3 * mask

    names = ("means3D", "shs", "opacities", "scales", "rotations")

    def leaves(device):
        L = {k: getattr(sc, k).detach().clone().to(device).requires_grad_(True) for k in names}
        L["means2D"] = torch.zeros(P, 3, device=device, requires_grad=True)
        return L

    t0 = time.perf_counter()
    Lc = leaves("cpu")
    pre = O.preprocess(Lc["means3D"], Lc["opacities"], s, shs=Lc["shs"], scales=Lc["scales"], rotations=Lc["rotations"],
                       means2D=Lc["means2D"])
    col, invd, _, _, _ = O.render_tiles(pre, bins, s, tiles=sample)
    ((col * wc).sum() + (invd * wd).sum()).backward()
    t_oracle = time.perf_counter() - t0
    del pre, col, invd

    Lg = leaves(dev)
    rast = GaussianRasterizer(raster_settings=_gpu_settings(s, dev))
    gcol, gradii, ginvd = rast(means3D=Lg["means3D"], means2D=Lg["means2D"], opacities=Lg["opacities"], shs=Lg["shs"],
                               scales=Lg["scales"], rotations=Lg["rotations"])
    ((gcol * wc.to(dev)).sum() + (ginvd * wd.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    assert torch.equal(gradii.cpu(), pre0["radii"].to(torch.int32))
    res = {}
    for k in Lc:
        a, b = Lg[k].grad.cpu().double(), Lc[k].grad.double()
        scale = b.abs().max().item()
        assert scale > 0, f"{k}: oracle gradient is identically zero"
        dd = (a - b).abs() / scale
        # Gaussians that touch no sampled tile get exactly zero from both sides
        zero_b = (b.reshape(P, -1).abs().sum(1) == 0)
        assert float(a.reshape(P, -1)[zero_b].abs().max()) == 0.0, f"{k}: non-zero gradient for an untouched Gaussian"
        nz = dd.reshape(P, -1)[~zero_b].flatten()
        res[k] = {"max": float(dd.max()), "p99.9": float(torch.quantile(nz[:8_000_000], 0.999)) if nz.numel() else 0.0}
        # bar: 2e-4 of max |grad| on every entry, 1e-5 at the 99.9th percentile (measured: <= 8e-6 / 4e-7)
        assert res[k]["max"] < 2e-4, f"{k}: max err {res[k]['max']:.3e} (rel. to max |grad|)"
        assert res[k]["p99.9"] < 1e-5, f"{k}: 99.9th pct err {res[k]['p99.9']:.3e}"
    _report(f"{name}/backward", tiles=len(sample), oracle_seconds=round(t_oracle, 1),
            **{f"{k}_{m}": f"{v:.2e}" for k, r in res.items() for m, v in r.items()})


def test_config1_1M_1080p_backward_on_sampled_tiles():
    _backward_case(1_000_000, 1920, 1080, 96, "configs[1] 1M@1080p")


MAX_ADJUDICATED = 8      # pixels per frame that may differ from the fp32 oracle by more than 1e-5 OUTSIDE its fragile mask; each must be within 1e-5 of its fp64 blend


def _fp64_adjudicate(pre, bins, pixels, hip_color, oracle_color, s):
    """Blend the tiles of `pixels` ([k, 2] rows of (y, x)) in fp64 from the oracle's fp32 per-Gaussian outputs; per pixel the distance
    of the kernel's and of the fp32 oracle's colour from the fp64 colour (max over channels)."""
    gx = pre["grid"][0]
    res = []
    bg = s.bg.detach().cpu().double().reshape(3)
    for y, x in pixels.tolist():
        t = (y // 16) * gx + x // 16
        a, b = int(bins["ranges"][t, 0]), int(bins["ranges"][t, 1])
        ids = bins["point_list"][a:b]
        px = torch.tensor([float(x)], dtype=torch.float64)
        py = torch.tensor([float(y)], dtype=torch.float64)
        with torch.no_grad():
            C, _, fT, nc, _ = O._blend_tile(px, py, pre["means2D"][ids].double(), pre["conic"][ids].double(), pre["opacity"][ids].double(),
                                            pre["rgb"][ids].double(), (1.0 / pre["depths"][ids]).double())
        c64 = C[0] + fT[0] * bg
        res.append({"pixel": [y, x], "list_length": b - a, "n_contrib_fp64": int(nc[0]),
                    "hip_minus_oracle32": float((hip_color[:, y, x].double() - oracle_color[:, y, x].double()).abs().max()),
                    "hip_minus_fp64": float((hip_color[:, y, x].double() - c64).abs().max()),
                    "oracle32_minus_fp64": float((oracle_color[:, y, x].double() - c64).abs().max())})
    return res


def _whole_frame_case(P, W, H, name, kind="uniform"):
    """The oracle blends the WHOLE frame (every tile), and the comparison is reported twice: with the oracle's fragile mask
    (bar 1e-5 on every other pixel) and WITHOUT any mask -- the count of pixels whose error exceeds 1e-5 over the full frame and
    the largest error among them (a flipped hard threshold moves a pixel by at most one alpha quantum of the brightest colour)."""
    from diff_gaussian_rasterization.debug import forward_with_views
    dev = torch.device("cuda:0")
    cam, sc, s, pre, bins, _ = _config(P, W, H, kind)
    d = sc.to(dev)
    out = forward_with_views(_gpu_settings(s, dev), d.means3D, d.opacities, shs=d.shs, scales=d.scales, rotations=d.rotations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        col, invd, fT, ncon, frag = O.render_tiles(pre, bins, s, want_fragile=True)
    t_oracle = time.perf_counter() - t0
    err = (out["color"].cpu() - col).abs().max(0).values
    erri = (out["invdepth"].cpu()[0] - invd[0]).abs()
    over = err > IMG_TOL
    cmax = max(1.0, float(pre["rgb"].abs().max()))
    m = {"pixels": W * H, "oracle_seconds": round(t_oracle, 1), "max_err_nonfragile": float(err[~frag].max()),
         "max_invdepth_err_nonfragile": float(erri[~frag].max()), "fragile_fraction": float(frag.float().mean()),
         "pixels_over_1e-5_no_mask": int(over.sum()), "pixels_over_1e-5_outside_fragile_mask": int((over & ~frag).sum()),
         "max_err_no_mask": float(err.max()), "n_contrib_mismatches_no_mask": int((out["n_contrib"].cpu().long() != ncon).sum())}
    # a pixel beyond 1e-5 OUTSIDE the fragile mask is adjudicated by an fp64 blend of its tile (the same fp32
    # preprocess outputs and sorted list, the blend arithmetic of Appendix A.4 in double): whichever of the kernel (sequential fp32 FMA chain)
    # and the fp32 oracle (torch's blocked .sum() over the list) is closer to it is the better fp32 evaluation; the 1e-5 bar is held
    # against the fp64 value.
    outside = torch.nonzero(over & ~frag)
    # (the budget is on the pixels FOUND, and every one of them is adjudicated -- round 5 looked at the first 8 in raster
    # order and asserted the length of that list, which let any number of others through.  Measured: 0 / 0 / 0 / 1 on the four configs.)
    assert outside.shape[0] <= MAX_ADJUDICATED, f"{outside.shape[0]} pixels beyond 1e-5 outside the fragile mask (budget {MAX_ADJUDICATED})"
    if outside.numel():
        m["fp64_adjudication"] = _fp64_adjudicate(pre, bins, outside, out["color"].cpu(), col, s)
    _report(f"{name}/whole frame", **m)
    if outside.numel():
        assert len(m["fp64_adjudication"]) == outside.shape[0]
        for a in m["fp64_adjudication"]:
            assert a["hip_minus_fp64"] <= IMG_TOL, f"pixel {a['pixel']}: kernel is {a['hip_minus_fp64']:.3e} from the fp64 blend"
    else:
        assert m["max_err_nonfragile"] <= IMG_TOL
    assert m["max_err_no_mask"] <= cmax / 255.0 * 1.01 + IMG_TOL
    assert m["pixels_over_1e-5_no_mask"] <= 1e-4 * W * H


def test_config1_1M_1080p_whole_frame_without_a_mask():
    """configs[1], all 8 160 tiles blended by the oracle."""
    _whole_frame_case(1_000_000, 1920, 1080, "configs[1] 1M@1080p")


def _reference_bins_case(P, W, H, n_tiles, name, kind="uniform", expect_R=None):
    """north_star: "tile bin counts bit-exact" -- on the REFERENCE's own tile rectangles (consumer
    gaussian_renderer/__init__.py:91-110).  The library runs with snug_tiles = 0 (it bins the square of radius ceil(3 sqrt(lambda_max))
    exactly as the reference does) and is compared with the oracle in reference mode: tiles_touched, R, the sorted point list
    and the tile ranges bit for bit on all Gaussians, n_contrib / image on sampled tiles (round 3 had moved
    every integer comparison to the snug restatement, which was written from the product's own header)."""
    from diff_gaussian_rasterization import _lib
    from diff_gaussian_rasterization.debug import forward_with_views
    dev = torch.device("cuda:0")
    cam, sc, s, pre_snug, bins_snug, _ = _config(P, W, H, kind)
    t0 = time.perf_counter()
    with reference_tiles(), torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        bins = O.bin_and_sort(pre)
    t_oracle = time.perf_counter() - t0
    assert bins["R"] > bins_snug["R"] and torch.equal(pre["radii"], pre_snug["radii"])
    if expect_R is not None:
        assert bins["R"] == expect_R, (bins["R"], expect_R)
    d = sc.to(dev)
    sample = _busy_sample(bins, n_tiles)
    _lib.set_option("snug_tiles", 0)
    try:
        out = forward_with_views(_gpu_settings(s, dev), d.means3D, d.opacities, shs=d.shs, scales=d.scales, rotations=d.rotations)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("snug_tiles", 1)
    _check_bins(out, pre, bins)
    with reference_tiles():
        m = _check_sampled_tiles(out, pre, bins, s, sample, W, H)
    _report(f"{name}/reference tile rectangles", P=P, W=W, H=H, R_reference=bins["R"], R_snug=bins_snug["R"],
            oracle_seconds=round(t_oracle, 1), bins="bit-exact vs the oracle in reference mode (tiles_touched, R, point_list, ranges; "
            "n_contrib on the sampled tiles)", **m)
    del out, pre, bins


def test_config1_reference_tile_rectangles_bit_exact():
    _reference_bins_case(1_000_000, 1920, 1080, 60, "configs[1] 1M@1080p", expect_R=11330172)      # SURVEY 8(d)'s probe count


def test_config1_clustered_reference_tile_rectangles_bit_exact():
    _reference_bins_case(1_000_000, 1920, 1080, 40, "configs[1] clustered", kind="clustered")


def test_config1_clustered_whole_frame_without_a_mask():
    """the clustered stand-in, whole frame, no mask.  Measured (round 4): 10 of 2 073 600 pixels beyond 1e-5
    with no mask, ONE of them outside the oracle's fragile mask, at 1.13e-5 from the fp32 oracle (tile lists of up to 8 487 entries).
    Round 5: such a pixel is adjudicated by an fp64 blend of its tile (`_fp64_adjudicate`) and the 1e-5 bar is
    held against THAT value -- no relaxed bar for this scene any more."""
    _whole_frame_case(1_000_000, 1920, 1080, "configs[1] clustered", kind="clustered")


def test_config1_clustered_forward_and_backward():
    """The clustered stand-in (gsr_synth.make_clustered_scene, the scene bench.py reports beside the uniform one): heavy-tailed
    tile lists, splats that cover the whole frame, ~40 % of the Gaussians visible -- bins bit-exact, sampled image and sampled
    gradients against the oracle."""
    _forward_case(1_000_000, 1920, 1080, 120, "configs[1] clustered", kind="clustered")
    _backward_case(1_000_000, 1920, 1080, 48, "configs[1] clustered", kind="clustered")


def test_config3_1M_4K_forward_both_builds():
    """BASELINE configs[3] stand-in: 1 M Gaussians @3840x2160 (32 400 tiles)."""
    _forward_case(1_000_000, 3840, 2160, 160, "configs[3] 1M@4K")


def test_config3_1M_4K_backward_on_sampled_tiles():
    _backward_case(1_000_000, 3840, 2160, 64, "configs[3] 1M@4K")


def test_config3_1M_4K_reference_tile_rectangles_bit_exact():
    _reference_bins_case(1_000_000, 3840, 2160, 40, "configs[3] 1M@4K")


def test_config3_1M_4K_whole_frame_without_a_mask():
    """configs[3], all 32 400 tiles blended by the oracle (about a minute of host time)."""
    _whole_frame_case(1_000_000, 3840, 2160, "configs[3] 1M@4K")


def test_config4_6M_1080p_forward_both_builds():
    """BASELINE configs[4] stand-in: 6 M Gaussians @1920x1080 (R = 68 M instances)."""
    _forward_case(6_000_000, 1920, 1080, 60, "configs[4] 6M@1080p")


def test_config4_6M_1080p_backward_on_sampled_tiles():
    _backward_case(6_000_000, 1920, 1080, 24, "configs[4] 6M@1080p")


def test_config4_6M_1080p_whole_frame_without_a_mask():
    """configs[4] (6 M Gaussians, R = 47.6 M), all 8 160 tiles blended by the oracle -- the last full-size config that was
    only sampled."""
    _whole_frame_case(6_000_000, 1920, 1080, "configs[4] 6M@1080p")


def test_config4_6M_1080p_reference_tile_rectangles_bit_exact():
    _reference_bins_case(6_000_000, 1920, 1080, 16, "configs[4] 6M@1080p")
    _cache.clear()
