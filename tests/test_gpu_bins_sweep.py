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
  bucket_tables_beyond_1M   P = 1.3 M: more than 256 workgroups in the bucket depth sort's tables (ds_scan's re-reading sweep)
  odd_frame         width / height not multiples of 16, partial edge tiles, a one-tile-high last row
  tiles_65536       exactly 65536 tiles: the largest frame of the fused two-level tile sort (one more tile takes the
                    32-bit-key LSD path, covered by test_gpu_parity.py)
  huge_tiles_65536  the same frame with splats of thousands of tiles each
  depth_*           depth distributions that select the paths of the bucket depth sort (ties, a crowd, a 4-decade gap, a single key;
                    round 6: heavy tails, a wall, a thin wall, > 4096 ties per value -- the equalised bucket tables)
Reference boundary: gaussian_renderer/__init__.py:91-110; bins = SURVEY 8(c)(3) "tile bin counts bit-exact"."""
import pytest
import torch

from helpers import O, make_camera, make_scene, oracle_settings, reference_tiles

pytestmark = pytest.mark.gpu

CASES = {
    # name: (P, W, H, s_med, seed)
    "single_block": (300, 80, 48, 0.03, 1),
    "tiny_splats": (40_000, 640, 368, 0.0015, 2),
    "huge_splats": (1_500, 800, 608, 0.6, 3),
    "sort_1024_tier": (524_288 - 3, 320, 240, 0.004, 4),
    "sort_2048_tier": (524_288 + 1061, 320, 240, 0.004, 5),
    # more than 1 M Gaussians: ds_scan walks more than 16 table rows per thread (its second sweep re-reads instead of keeping them)
    "bucket_tables_beyond_1M": (1_300_000, 320, 240, 0.003, 15),
    "odd_frame": (50_000, 1001, 337, 0.02, 6),
    "tiles_65536": (6_000, 4096, 4096, 0.03, 7),
    # thousands of tiles per Gaussian on a 65536-tile frame (256 x 256 buckets, Gaussians that own dozens of emission blocks)
    "huge_tiles_65536": (2_000, 4096, 4096, 0.35, 14),
    # depth distributions that select the code paths of the BUCKET depth sort (depthsort.hip):
    "depth_ties": (30_000, 320, 240, 0.01, 8),        # depths quantised to 64 values: tie order = Gaussian index, in every segment
    "depth_crowd": (20_000, 320, 240, 0.01, 9),       # 3/4 of the Gaussians inside 1e-5 of the depth range: oversized segments
    "depth_gap": (30_000, 320, 240, 0.01, 10),        # two clusters 4 orders of magnitude apart: shift 16, sparse buckets
    "depth_one_key": (9_000, 320, 240, 0.01, 13),     # every visible Gaussian at the SAME depth bits: zero sort passes
    # round 5: the bulk inside 5 % of an octave + a few floaters 100x farther / 10x nearer -- the buckets span ds_hist's ROBUST key
    # range, the floaters share the end buckets, whose segments sort on the full keys; 300 K Gaussians = 293 workgroups of the projection kernel,
    # so the robust range has groups of two workgroups to work with
    "depth_outliers": (300_000, 320, 240, 0.004, 16),
    # round 6: the histogram-equalised buckets (csrc/depthsort.hip ds_hist) -- 3 % of the Gaussians 0.1-300x nearer / farther than a narrow bulk (every
    # workgroup of the projection kernel holds some); half of them on a slab 0.3 % thick; 60 % on a slab 2e-4 thick (narrower than a first-level bucket:
    # the second-level table); and a crowd whose 48 depth values hold > 4096 ties each (a single bucket of equal keys: chunked output, no pass)
    "depth_heavy_tails": (300_000, 320, 240, 0.004, 17),
    "depth_wall": (300_000, 320, 240, 0.004, 18),
    "depth_wall_thin": (300_000, 320, 240, 0.004, 19),
    "depth_crowd_big": (400_000, 320, 240, 0.003, 20),
}


def _reshape_depths(name, sc, cam, seed):
    """Move the Gaussians along their view rays (camera at the origin looking down +z: x, y, z scale together, the screen
    position stays) so that the depth distribution is the one the case is named after."""
    g = torch.Generator().manual_seed(seed)
    z = sc.means3D[:, 2].clone()
    P = z.numel()
    if name == "depth_ties":
        znew = 2.0 + torch.randint(0, 64, (P,), generator=g).float() * 0.125
    elif name == "depth_crowd":
        znew = z.clone()
        crowd = torch.rand(P, generator=g) < 0.75
        znew[crowd] = 5.0 + torch.randint(0, 48, (int(crowd.sum()),), generator=g).float() * 4.76837158203125e-07   # 1 ulp steps at 5.0
    elif name == "depth_gap":
        near = torch.rand(P, generator=g) < 0.5
        znew = torch.where(near, 0.25 + 0.05 * torch.rand(P, generator=g), 3000.0 + 6000.0 * torch.rand(P, generator=g))
    elif name == "depth_one_key":
        znew = torch.full_like(z, 4.0)
    elif name == "depth_outliers":
        znew = 4.0 + 0.15 * torch.rand(P, generator=g)
        far, near = torch.randperm(P, generator=g)[:9], torch.randperm(P, generator=g)[:4]
        znew[far] = 400.0 + 2000.0 * torch.rand(9, generator=g)
        znew[near] = 0.3 + 0.2 * torch.rand(4, generator=g)
    elif name == "depth_heavy_tails":
        g = torch.Generator().manual_seed(seed + 1000)      # (make_scene draws its screen positions from the SAME seed: a mask from `g` would select a screen edge)
        znew = 4.0 + 0.4 * torch.rand(P, generator=g)
        far = torch.rand(P, generator=g) < 0.03
        znew[far] = torch.exp(torch.empty(int(far.sum())).uniform_(-1.2, 7.0, generator=g))      # 0.3 .. 1100
    elif name == "depth_wall":
        g = torch.Generator().manual_seed(seed + 1000)
        znew = torch.where(torch.rand(P, generator=g) < 0.5, 6.0 + 0.009 * torch.randn(P, generator=g), 1.0 + 39.0 * torch.rand(P, generator=g))
    elif name == "depth_wall_thin":
        g = torch.Generator().manual_seed(seed + 1000)
        znew = torch.where(torch.rand(P, generator=g) < 0.6, 5.0 * (1.0 + 2e-4 * torch.rand(P, generator=g)), 1.0 + 39.0 * torch.rand(P, generator=g))
    elif name == "depth_crowd_big":
        g = torch.Generator().manual_seed(seed + 1000)
        znew = z.clone()
        crowd = torch.rand(P, generator=g) < 0.75
        znew[crowd] = 5.0 + torch.randint(0, 48, (int(crowd.sum()),), generator=g).float() * 4.76837158203125e-07
    else:
        return
    f = (znew / z).unsqueeze(1)
    sc.means3D.mul_(f)
    if name in ("depth_gap", "depth_outliers", "depth_heavy_tails", "depth_wall", "depth_wall_thin"):
        sc.scales.mul_(f)          # keep the far cluster's / the floaters' splats visible on screen


@pytest.mark.parametrize("tiles_mode", ["snug", "reference"])
@pytest.mark.parametrize("depth_sort_mode", [2, 1], ids=["bucket", "lsd"])
@pytest.mark.parametrize("name", list(CASES))
def test_bins_bit_exact_in_every_structural_regime(name, depth_sort_mode, tiles_mode):
    """`reference`: the library bins the reference's own tile rectangles (option snug_tiles = 0)
    and is compared with the oracle in reference mode -- north_star's "tile bin counts bit-exact" on the reference's bins
    (VERDICT r03 item 2); `snug`: the product's default rectangles against the oracle's restatement of them."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib
    from diff_gaussian_rasterization.debug import forward_with_views
    import contextlib
    if tiles_mode == "reference" and depth_sort_mode == 1 and name not in ("single_block", "odd_frame", "depth_ties"):
        pytest.skip("reference rectangles x LSD sort: three representative cases are enough")
    if name == "bucket_tables_beyond_1M" and (tiles_mode == "reference" or depth_sort_mode == 1):
        pytest.skip("a case for the bucket sort's tables only")
    P, W, H, s_med, seed = CASES[name]
    dev = torch.device("cuda:0")
    cam = make_camera(W, H)
    sc = make_scene(P, cam, seed=seed, s_med=s_med)
    _reshape_depths(name, sc, cam, seed)
    s = oracle_settings(cam)
    with (reference_tiles() if tiles_mode == "reference" else contextlib.nullcontext()), torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        bins = O.bin_and_sort(pre)
    V = int((pre["radii"] > 0).sum())
    R = int(bins["R"])
    assert V > 0 and R > 0, "degenerate case: nothing visible"
    # the case really is in the regime it is named after
    if name == "tiny_splats":
        assert R / V < 3.0 and R > 3 * 4096, (R, V)        # > 1024 Gaussians per 4096 instances, several workgroups
    if name == "huge_splats" and tiles_mode == "snug":
        assert R / P > 64.0, (R, P)                         # beyond the scan kernel's block-table capacity
    if name == "tiles_65536":
        assert ((W + 15) // 16) * ((H + 15) // 16) == 65536
    if name == "depth_crowd":
        assert V > 12_000
    d = sc.to(dev)
    rs = GaussianRasterizationSettings(H, W, s.tanfovx, s.tanfovy, s.bg.to(dev), s.scale_modifier, s.viewmatrix.to(dev),
                                       s.projmatrix.to(dev), s.sh_degree, s.campos.to(dev), False, False, s.antialiasing)
    _lib.set_option("depth_sort_mode", depth_sort_mode)
    _lib.set_option("snug_tiles", 0 if tiles_mode == "reference" else 1)
    # the level-2 scan rides along: folded into the scatter with the bucket sort, its own launch with the LSD passes
    _lib.set_option("level2_scan_mode", 2 if depth_sort_mode == 2 else 1)
    try:
        for no_backward in (False, True):
            out = forward_with_views(rs, d.means3D, d.opacities, shs=d.shs, scales=d.scales, rotations=d.rotations,
                                     no_backward=no_backward)
            torch.cuda.synchronize()
            assert torch.equal(out["radii"].cpu(), pre["radii"].to(torch.int32)), "radii differ"
            assert torch.equal(out["tiles_touched"].cpu().to(torch.int64), pre["tiles_touched"]), "tiles_touched differ"
            assert out["R"] == R, f"R {out['R']} != {R}"
            pl = out["point_list"].cpu().to(torch.int64)
            if not torch.equal(pl, bins["point_list"]):
                bad = (pl != bins["point_list"]).nonzero().flatten()
                raise AssertionError(f"sorted point list differs at {bad.numel()} of {pl.numel()} positions, first {bad[:8].tolist()}: "
                                     f"got {pl[bad[:8]].tolist()} want {bins['point_list'][bad[:8]].tolist()}")
            assert torch.equal(out["ranges"].cpu().to(torch.int64), bins["ranges"]), "tile ranges differ"
            del out
    finally:
        _lib.set_option("depth_sort_mode", 0)
        _lib.set_option("snug_tiles", 1)
        _lib.set_option("level2_scan_mode", 0)


def test_depth_order_and_scan_of_the_bucket_sort_equal_the_lsd_sort():
    """The two depth sorts leave the same internal state: depth order, rectangles in depth order (listed Gaussians) and
    the inclusive scan of the tile counts -- the arrays the emission and the backward's reduce read -- on a frame with ties."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib
    from diff_gaussian_rasterization.debug import forward_with_views
    dev = torch.device("cuda:0")
    cam = make_camera(640, 368)
    sc = make_scene(200_000, cam, seed=21, s_med=0.006)
    sc.means3D.mul_((torch.round(sc.means3D[:, 2] * 64) / 64 / sc.means3D[:, 2]).unsqueeze(1))      # depth ties
    s = oracle_settings(cam)
    d = sc.to(dev)
    rs = GaussianRasterizationSettings(368, 640, s.tanfovx, s.tanfovy, s.bg.to(dev), s.scale_modifier, s.viewmatrix.to(dev),
                                       s.projmatrix.to(dev), s.sh_degree, s.campos.to(dev), False, False, s.antialiasing)
    got = {}
    try:
        for mode in (1, 2):
            _lib.set_option("depth_sort_mode", mode)
            out = forward_with_views(rs, d.means3D, d.opacities, shs=d.shs, scales=d.scales, rotations=d.rotations)
            torch.cuda.synchronize()
            got[mode] = {k: out[k].cpu().clone() for k in ("depth_order", "offsets", "point_list", "ranges")}
            got[mode]["R"] = out["R"]
            del out
    finally:
        _lib.set_option("depth_sort_mode", 0)
    assert got[1]["R"] == got[2]["R"]
    for k in ("depth_order", "offsets", "point_list", "ranges"):
        assert torch.equal(got[1][k], got[2][k]), f"{k} differs between the LSD and the bucket depth sort"


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
