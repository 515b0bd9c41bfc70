"""Bit-exact bins across the structural regimes that select different code paths of the binning chain.

Every case compares radii, tiles_touched, R, the sorted point list and the tile ranges of BOTH forward builds with
`O.preprocess` + `O.bin_and_sort` on all Gaussians (the same contract as tests/test_gpu_fullsize.py, at sizes the oracle
finishes in a second or two).  What each case is there for:

  single_block      fewer instances than one emission workgroup (4096) and fewer Gaussians than one sort workgroup
  tiny_splats       ~1 tile per Gaussian: more than 1024 Gaussians per emission workgroup -> emit_scatter's global-fetch loop
                    (the owner records no longer fit the LDS table)
  huge_splats       > 64 tiles per Gaussian on average -> the per-block "first Gaussian" table is rebuilt from R
                    (gsr_launch_fill_block_first) instead of coming from the scan kernel
  sort_1024_tier    P just below 512 K  -> 1024-key radix workgroups for the depth sort
  sort_2048_tier    P just above 512 K  -> 2048-key workgroups (ragged last workgroup)
  odd_frame         width / height not multiples of 16, partial edge tiles, a one-tile-high last row
  tiles_65536       exactly 65536 tiles: the largest frame of the fused two-level tile sort (one more tile takes the
                    32-bit-key LSD path, covered by test_gpu_parity.py)
Reference boundary: gaussian_renderer/__init__.py:91-110; bins = SURVEY 8(c)(3) "tile bin counts bit-exact"."""
import pytest
import torch

from helpers import O, make_camera, make_scene, oracle_settings

pytestmark = pytest.mark.gpu

CASES = {
    # name: (P, W, H, s_med, seed)
    "single_block": (300, 80, 48, 0.03, 1),
    "tiny_splats": (40_000, 640, 368, 0.0015, 2),
    "huge_splats": (1_500, 800, 608, 0.6, 3),
    "sort_1024_tier": (524_288 - 3, 320, 240, 0.004, 4),
    "sort_2048_tier": (524_288 + 1061, 320, 240, 0.004, 5),
    "odd_frame": (50_000, 1001, 337, 0.02, 6),
    "tiles_65536": (6_000, 4096, 4096, 0.03, 7),
}


@pytest.mark.parametrize("name", list(CASES))
def test_bins_bit_exact_in_every_structural_regime(name):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    from diff_gaussian_rasterization.debug import forward_with_views
    P, W, H, s_med, seed = CASES[name]
    dev = torch.device("cuda:0")
    cam = make_camera(W, H)
    sc = make_scene(P, cam, seed=seed, s_med=s_med)
    s = oracle_settings(cam)
    with torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        bins = O.bin_and_sort(pre)
    V = int((pre["radii"] > 0).sum())
    R = int(bins["R"])
    assert V > 0 and R > 0, "degenerate case: nothing visible"
    # the case really is in the regime it is named after
    if name == "tiny_splats":
        assert R / V < 3.0 and R > 3 * 4096, (R, V)        # > 1024 Gaussians per 4096 instances, several workgroups
    if name == "huge_splats":
        assert R / P > 64.0, (R, P)                         # beyond the scan kernel's block-table capacity
    if name == "tiles_65536":
        assert ((W + 15) // 16) * ((H + 15) // 16) == 65536
    d = sc.to(dev)
    rs = GaussianRasterizationSettings(H, W, s.tanfovx, s.tanfovy, s.bg.to(dev), s.scale_modifier, s.viewmatrix.to(dev),
                                       s.projmatrix.to(dev), s.sh_degree, s.campos.to(dev), False, False, s.antialiasing)
    for no_backward in (False, True):
        out = forward_with_views(rs, d.means3D, d.opacities, shs=d.shs, scales=d.scales, rotations=d.rotations,
                                 no_backward=no_backward)
        torch.cuda.synchronize()
        assert torch.equal(out["radii"].cpu(), pre["radii"].to(torch.int32)), "radii differ"
        assert torch.equal(out["tiles_touched"].cpu().to(torch.int64), pre["tiles_touched"]), "tiles_touched differ"
        assert out["R"] == R, f"R {out['R']} != {R}"
        assert torch.equal(out["point_list"].cpu().to(torch.int64), bins["point_list"]), "sorted point list differs"
        assert torch.equal(out["ranges"].cpu().to(torch.int64), bins["ranges"]), "tile ranges differ"
        del out


# ------------------------------------------------------------------------------------------------------------------
# The backward in the two extreme segment regimes of its gradient reduce (bwd_reduce_instances: a segmented scan over the
# instances of every Gaussian): one or two instances per Gaussian -- most segments are single lanes, several Gaussians of a
# group have none -- and hundreds of instances per Gaussian -- segments span many 64-instance chunks and all four waves.
# Oracle: autograd through O.rasterize (train.py:142 boundary).  Bars as in test_gpu_parity.py.
# ------------------------------------------------------------------------------------------------------------------
BWD_CASES = {
    # name: (P, W, H, s_med, seed)
    "segments_of_one": (6_000, 320, 192, 0.002, 11),
    "segments_of_hundreds": (60, 320, 192, 0.6, 12),
}


@pytest.mark.parametrize("name", list(BWD_CASES))
def test_backward_in_extreme_segment_regimes(name):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    P, W, H, s_med, seed = BWD_CASES[name]
    dev = torch.device("cuda:0")
    cam = make_camera(W, H)
    sc = make_scene(P, cam, seed=seed, s_med=s_med)
    s = oracle_settings(cam, bg=torch.tensor([0.2, 0.1, 0.3]))
    g = torch.Generator().manual_seed(seed)
    wc, wd = torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g) * 0.3

    def leaves(device):
        L = {k: getattr(sc, k).detach().clone().to(device).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        L["means2D"] = torch.zeros(P, 3, device=device, requires_grad=True)
        return L

    Lc = leaves("cpu")
    col, radii, invd = O.rasterize(Lc["means3D"], Lc["means2D"], Lc["opacities"], s, shs=Lc["shs"], scales=Lc["scales"],
                                   rotations=Lc["rotations"])
    ((col * wc).sum() + (invd * wd).sum()).backward()
    R = int(O.bin_and_sort(O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations))["R"])
    V = int((radii > 0).sum())
    if name == "segments_of_one":
        assert R / V < 2.5, (R, V)
    else:
        assert R / V > 100.0, (R, V)
    Lg = leaves(dev)
    rs = GaussianRasterizationSettings(H, W, s.tanfovx, s.tanfovy, s.bg.to(dev), s.scale_modifier, s.viewmatrix.to(dev),
                                       s.projmatrix.to(dev), s.sh_degree, s.campos.to(dev), False, False, s.antialiasing)
    gcol, gradii, ginvd = GaussianRasterizer(rs)(means3D=Lg["means3D"], means2D=Lg["means2D"], opacities=Lg["opacities"],
                                                 shs=Lg["shs"], scales=Lg["scales"], rotations=Lg["rotations"])
    ((gcol * wc.to(dev)).sum() + (ginvd * wd.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    assert torch.equal(gradii.cpu(), radii)
    for k in Lc:
        a, b = Lg[k].grad.cpu().double(), Lc[k].grad.double()
        assert b.abs().max().item() > 0, f"{k}: oracle gradient is identically zero"
        d = (a - b).abs() / (b.abs().max().item() + 1e-30)
        assert d.max().item() < 1e-4, f"{name}/{k}: max err {d.max().item():.3e} (rel. to max |grad|)"
        assert torch.quantile(d.flatten()[: 4_000_000], 0.999).item() < 1e-5, f"{name}/{k}: 99.9th percentile too large"
