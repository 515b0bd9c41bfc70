"""GPU parity tests proper: the HIP path (through the C ABI, via the drop-in package) against the CPU oracle on
identical seeded inputs.  Bars (BASELINE.json north_star): integer outputs -- radii, tiles_touched, R, sorted
point list, tile ranges, n_contrib -- BIT-EXACT; image / invdepth within 1e-5 (fp32), except at pixels the oracle
flags "fragile" (a hard blend threshold within rounding noise, where either branch is a correct fp32 result;
there the bound is one alpha quantum, 1/255 x colour range)."""
import math

import numpy as np
import pytest
import torch

from helpers import O, make_camera, look_at_camera, make_scene, make_edge_scene, oracle_settings, parity_report

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-5


def gpu_settings(s, dev):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.bg.to(dev),
                                         s.scale_modifier, s.viewmatrix.to(dev), s.projmatrix.to(dev), s.sh_degree,
                                         s.campos.to(dev), False, True, s.antialiasing)


def mk(name):
    if name == "c1":
        cam = make_camera(256, 256)
        return cam, make_scene(1000, cam, seed=0), {}
    if name == "odd_aa":
        cam = make_camera(250, 131)
        return cam, make_scene(3000, cam, seed=3, s_med=0.02), {"antialiasing": True, "bg": torch.tensor([0.3, 0.6, 0.1])}
    if name == "edge_lookat":
        cam = look_at_camera(333, 200, (0.3, -0.2, -1.0), (0.1, 0.0, 3.0))
        return cam, make_edge_scene(4000, cam, seed=5), {"bg": torch.tensor([1.0, 1.0, 1.0])}
    if name == "edge_aa_scale":
        cam = make_camera(480, 270)
        return cam, make_edge_scene(4000, cam, seed=7), {"antialiasing": True, "scale_modifier": 1.7}
    if name == "deg1":
        cam = make_camera(320, 200)
        sc = make_scene(2500, cam, seed=9, s_med=0.03)
        return cam, sc, {"sh_degree": 1}
    if name == "deg0_dense":
        cam = make_camera(192, 128)
        return cam, make_scene(20000, cam, seed=13, s_med=0.03), {"sh_degree": 0}
    raise KeyError(name)


def run_oracle(cam, sc, opts, colors=None, cov=None, tile_rows=None):
    s = oracle_settings(cam, bg=opts.get("bg"), sh_degree=opts.get("sh_degree", 3),
                        scale_modifier=opts.get("scale_modifier", 1.0), antialiasing=opts.get("antialiasing", False))
    kw = {}
    if tile_rows is not None:
        kw = dict(tile_y0=tile_rows[0], tile_y1=tile_rows[1])
    with torch.no_grad():
        col, radii, invd, aux = O.rasterize(
            sc.means3D, None, sc.opacities, s, shs=None if colors is not None else sc.shs, colors_precomp=colors,
            scales=None if cov is not None else sc.scales, rotations=None if cov is not None else sc.rotations,
            cov3D_precomp=cov, want_fragile=True, return_aux=True, **kw)
    return s, col, radii, invd, aux


def _set_variant(name, value):
    """Non-default kernel variants exist only in the measurement build (GSR_AB=1 python build.py -> lib_ab/, selected
    with GSR_LIB); the product library rejects them."""
    from diff_gaussian_rasterization import _lib
    try:
        _lib.set_option(name, value)
    except _lib.GsrError:
        assert value != 0, "the default variant must always be available"
        pytest.skip(f"{name}={value} is an A/B variant that is not compiled into this library")


def run_gpu(s, sc, colors=None, cov=None, tile_rows=None, variant=0, no_backward=False):
    from diff_gaussian_rasterization import _lib
    from diff_gaussian_rasterization.debug import forward_with_views
    dev = torch.device("cuda:0")
    _set_variant("render_fwd_variant", variant)
    d = sc.to(dev)
    out = forward_with_views(gpu_settings(s, dev), d.means3D, d.opacities, shs=None if colors is not None else d.shs,
                             colors_precomp=None if colors is None else colors.to(dev),
                             scales=None if cov is not None else d.scales,
                             rotations=None if cov is not None else d.rotations,
                             cov3D_precomp=None if cov is None else cov.to(dev), tile_rows=tile_rows,
                             no_backward=no_backward)
    torch.cuda.synchronize()
    _lib.set_option("render_fwd_variant", 0)
    return out


def check_forward(s, col, radii, invd, aux, out, band=None):
    H, W = int(s.image_height), int(s.image_width)
    # ---- integer outputs: bit exact ----
    assert torch.equal(out["radii"].cpu(), radii), "radii differ"
    assert torch.equal(out["tiles_touched"].cpu().to(torch.int64), aux["tiles_touched"]), "tiles_touched differ"
    assert out["R"] == aux["R"], f"R {out['R']} != {aux['R']}"
    assert torch.equal(out["point_list"].cpu().to(torch.int64), aux["point_list"]), "sorted point list differs"
    assert torch.equal(out["ranges"].cpu().to(torch.int64), aux["ranges"]), "tile ranges differ"
    # ---- float outputs ----
    frag = aux["fragile"]
    rows = slice(0, H) if band is None else slice(band[0] * 16, min(band[1] * 16, H))
    g_col, g_inv = out["color"].cpu(), out["invdepth"].cpu()
    err = (g_col - col).abs().amax(dim=0)
    ok = ~frag
    assert err[rows][ok[rows]].max().item() <= IMG_TOL, f"image error {err[rows][ok[rows]].max().item():.3e}"
    cmax = max(1.0, float(aux["rgb"].abs().max()), float(s.bg.abs().max()))
    if frag.any():
        assert err[frag].max().item() <= cmax / 255.0 * 1.01 + IMG_TOL
    assert frag.float().mean().item() < 0.02, "too many fragile pixels for the exclusion to be meaningful"
    ierr = (g_inv - invd).abs()[0]
    assert ierr[rows][ok[rows]].max().item() <= IMG_TOL * max(1.0, float(invd.abs().max()))
    # blend state kept for backward (absent in the inference build, no_backward = 1)
    if "n_contrib" in out:
        assert torch.equal(out["n_contrib"].cpu().to(torch.int64)[rows][ok[rows]], aux["n_contrib"][rows][ok[rows]])
        terr = (out["final_T"].cpu() - aux["final_T"]).abs()
        # T is a running product of up to hundreds of (1 - alpha) factors; T(1-alpha) vs fma(-alpha,T,T) differ by an ulp each
        assert terr[rows][ok[rows]].max().item() <= 5e-6
    if band is not None:   # rows outside the band untouched (zeros)
        mask = torch.ones(H, dtype=torch.bool)
        mask[rows] = False
        assert g_col[:, mask].abs().max().item() == 0.0


@pytest.mark.parametrize("no_backward", [False, True], ids=["track", "inference"])
@pytest.mark.parametrize("variant", [0, 1, 3])
@pytest.mark.parametrize("name", ["c1", "odd_aa", "edge_lookat", "edge_aa_scale", "deg1", "deg0_dense"])
def test_forward_parity(name, variant, no_backward):
    """no_backward = True is the INFERENCE instantiation (render_fwd_wave_bf<.., TRACK=false>, emit without the
    first-emission write, final_T / n_contrib not produced): the build a torch.no_grad() render and bench.py's forward
    metric run.  Both builds are held to the same bar against the oracle."""
    cam, sc, opts = mk(name)
    s, col, radii, invd, aux = run_oracle(cam, sc, opts)
    out = run_gpu(s, sc, variant=variant, no_backward=no_backward)
    assert (radii > 0).sum() > 100
    check_forward(s, col, radii, invd, aux, out)


def test_forward_colors_and_cov_precomp():
    cam, sc, opts = mk("edge_lookat")
    colors = torch.rand(sc.P, 3, generator=torch.Generator().manual_seed(1))
    cov = O.compute_cov3d(sc.scales, sc.rotations, 1.0, torch.float32)
    s, col, radii, invd, aux = run_oracle(cam, sc, opts, colors=colors, cov=cov)
    out = run_gpu(s, sc, colors=colors, cov=cov)
    check_forward(s, col, radii, invd, aux, out)
    # cov3D computed inside == cov3D handed in (same expression tree): identical radii and image
    out2 = run_gpu(s, sc, colors=colors)
    assert torch.equal(out2["radii"], out["radii"])
    assert torch.equal(out2["color"], out["color"])


@pytest.mark.parametrize("no_backward", [False, True], ids=["track", "inference"])
def test_forward_tile_band(no_backward):
    cam, sc, opts = mk("edge_aa_scale")
    band = (5, 11)
    s, col, radii, invd, aux = run_oracle(cam, sc, opts, tile_rows=band)
    out = run_gpu(s, sc, tile_rows=band, no_backward=no_backward)
    check_forward(s, col, radii, invd, aux, out, band=band)


def test_tile_bands_tile_the_frame_forward_and_backward():
    """What every rank of a screen-sharded run computes (HIP path, tile_rows extension): the bands' strips tile the
    full image bit-exactly and the bands' gradients sum to the full-frame gradients (the all-reduce of parallel.py)."""
    from diff_gaussian_rasterization import rasterize_gaussians
    dev = torch.device("cuda:0")
    cam = make_camera(400, 300)
    sc = make_edge_scene(6000, cam, seed=41).to(dev)
    s = oracle_settings(cam, bg=torch.tensor([0.3, 0.2, 0.1]))
    rs = gpu_settings(s, dev)
    H, W = 300, 400
    wc = torch.randn(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
    wd = torch.randn(1, H, W, generator=torch.Generator().manual_seed(4)).to(dev)

    def run(band):
        L = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
        m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
        col, radii, invd = rasterize_gaussians(L[0], m2, L[1], None, L[2], L[3], L[4], None, rs, band)
        rows = slice(0, H) if band is None else slice(band[0] * 16, min(band[1] * 16, H))
        ((col[:, rows] * wc[:, rows]).sum() + (invd[:, rows] * wd[:, rows]).sum()).backward()
        torch.cuda.synchronize()
        return col.detach(), radii, [t.grad for t in L] + [m2.grad]

    full_col, full_radii, full_g = run(None)
    bands = [(0, 5), (5, 6), (6, 14), (14, 19)]
    parts = [run(b) for b in bands]
    assert torch.equal(sum(p[0] for p in parts), full_col)
    for p in parts:
        assert torch.equal(p[1], full_radii)
    for i, fg in enumerate(full_g):
        tot = sum(p[2][i] for p in parts)
        assert (tot - fg).abs().max().item() <= 3e-5 * fg.abs().max().item()


def test_two_phase_backward_with_record_hook():
    """gsr_backward_blend + hook + gsr_backward_preprocess (the multi-GPU path) == the fused backward; the hook sees
    the [P,12] record tensor and its edits propagate linearly (what an all-reduce over ranks does)."""
    from diff_gaussian_rasterization import rasterize_gaussians
    from diff_gaussian_rasterization.parallel import render_sharded, hip_band_renderer, BandPlan
    dev = torch.device("cuda:0")
    cam = make_camera(320, 240)
    sc = make_scene(8000, cam, seed=23, s_med=0.03).to(dev)
    s = oracle_settings(cam, bg=torch.tensor([0.2, 0.2, 0.2]))
    rs = gpu_settings(s, dev)
    wc = torch.randn(3, 240, 320, generator=torch.Generator().manual_seed(5)).to(dev)
    seen = {}

    def run(sync):
        L = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
        col, _, invd = rasterize_gaussians(L[0], None, L[1], None, L[2], L[3], L[4], None, rs, None, sync)
        ((col * wc).sum() + invd.sum()).backward()
        torch.cuda.synchronize()
        return [t.grad for t in L]

    def double(rec):
        seen["shape"] = tuple(rec.shape)
        rec.mul_(2.0)

    fused, ident, twice = run(None), run(lambda rec: None), run(double)
    assert seen["shape"] == (sc.P, 12)
    for a, b, c in zip(fused, ident, twice):
        assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item()
        assert (c - 2 * a).abs().max().item() <= 2e-5 * a.abs().max().item()
    # the single-process (world = 1) path of the sharded renderer is the plain renderer
    L = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    col, radii, invd = render_sharded(hip_band_renderer(rs), L, BandPlan.uniform(15, 1), reduce="records")
    ((col * wc).sum() + invd.sum()).backward()
    for a, t in zip(fused, L):
        assert (a - t.grad).abs().max().item() <= 1e-5 * a.abs().max().item()


def _loss_weights(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g) * 0.3


def _backward_case(name, n, colors_cov=False, use_depth=True, seed=0):
    from diff_gaussian_rasterization import GaussianRasterizer
    cam, sc, opts = mk(name)
    # shrink so the per-tile autograd oracle stays fast
    idx = torch.arange(min(n, sc.P))
    import copy
    sc = copy.copy(sc)
    sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs = (sc.means3D[idx], sc.scales[idx], sc.rotations[idx],
                                                                   sc.opacities[idx], sc.shs[idx])
    s = oracle_settings(cam, bg=opts.get("bg"), sh_degree=opts.get("sh_degree", 3),
                        scale_modifier=opts.get("scale_modifier", 1.0), antialiasing=opts.get("antialiasing", False))
    H, W = cam.image_height, cam.image_width
    wc, wd = _loss_weights(H, W, seed)
    if not use_depth:
        wd = None

    def leaves(dev):
        L = {"means3D": sc.means3D, "opacities": sc.opacities}
        if colors_cov:
            L["colors_precomp"] = torch.rand(sc.P, 3, generator=torch.Generator().manual_seed(2))
            L["cov3D_precomp"] = O.compute_cov3d(sc.scales, sc.rotations, s.scale_modifier, torch.float32)
        else:
            L["shs"], L["scales"], L["rotations"] = sc.shs, sc.scales, sc.rotations
        L = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in L.items()}
        L["means2D"] = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
        return L

    # oracle
    Lc = leaves("cpu")
    kw = {k: v for k, v in Lc.items() if k not in ("means3D", "means2D", "opacities")}
    col, radii, invd = O.rasterize(Lc["means3D"], Lc["means2D"], Lc["opacities"], s, **kw)
    loss = (col * wc).sum() + ((invd * wd).sum() if use_depth else 0.0)
    loss.backward()
    # HIP
    dev = torch.device("cuda:0")
    Lg = leaves(dev)
    kwg = {k: v for k, v in Lg.items() if k not in ("means3D", "means2D", "opacities")}
    rast = GaussianRasterizer(raster_settings=gpu_settings(s, dev))
    gcol, gradii, ginvd = rast(means3D=Lg["means3D"], means2D=Lg["means2D"], opacities=Lg["opacities"], **kwg)
    gloss = (gcol * wc.to(dev)).sum() + ((ginvd * wd.to(dev)).sum() if use_depth else 0.0)
    gloss.backward()
    torch.cuda.synchronize()
    assert torch.equal(gradii.cpu(), radii)
    measured = {}
    for k in Lc:
        a, b = Lg[k].grad.cpu().double(), Lc[k].grad.double()
        scale = b.abs().max().item() + 1e-30
        d = (a - b).abs() / scale
        measured[k] = f"{d.max().item():.2e}/{torch.quantile(d.flatten()[: 4_000_000], 0.999).item():.2e}"
        # measured on MI355X (profiles/r02_parity_report.json): max <= 9.5e-6, p99.9 <= 3.3e-6 over the three scenes
        assert d.max().item() < 1e-4, f"{k}: max err {d.max().item():.3e} (rel. to max |grad|)"
        assert torch.quantile(d.flatten()[: 4_000_000], 0.999).item() < 1e-5, f"{k}: 99.9th pct err too large"
        assert b.abs().max().item() > 0, f"{k}: oracle gradient is identically zero"
    parity_report(f"small scene {name}/backward (max / p99.9 of |d grad| / max |grad|)", n=int(sc.P), **measured)


@pytest.mark.parametrize("bwd_variant", [0, 1, 4, 5])
def test_backward_parity_sh_scale_rot(bwd_variant):
    from diff_gaussian_rasterization import _lib
    # 0: wave per 16x8 half tile, two pixels per lane (default); A/B builds only: 1 = global atomics, 4 = round 1's
    # workgroup-per-tile kernel, 5 = wave per 8x8 quadrant
    _set_variant("render_bwd_variant", bwd_variant)
    try:
        _backward_case("c1", 1000)
    finally:
        _lib.set_option("render_bwd_variant", 0)


def test_backward_is_bit_reproducible_and_variants_agree():
    """The default backward has no atomics and every sum has a fixed association order: two runs are BIT-IDENTICAL.
    The A/B kernels (measurement build only) agree with it to fp32 summation noise."""
    from diff_gaussian_rasterization import GaussianRasterizer, _lib
    dev = torch.device("cuda:0")
    cam = make_camera(640, 360)
    sc = make_scene(60000, cam, seed=17, s_med=0.02).to(dev)
    s = oracle_settings(cam, bg=torch.tensor([0.1, 0.2, 0.3]))
    wc = torch.randn(3, 360, 640, generator=torch.Generator().manual_seed(0)).to(dev)

    def grads(variant):
        _lib.set_option("render_bwd_variant", variant)
        L = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
        m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
        col, _, invd = GaussianRasterizer(gpu_settings(s, dev))(means3D=L[0], means2D=m2, opacities=L[2], shs=L[1],
                                                                scales=L[3], rotations=L[4])
        ((col * wc).sum() + invd.sum()).backward()
        torch.cuda.synchronize()
        _lib.set_option("render_bwd_variant", 0)
        return [t.grad for t in L] + [m2.grad]

    a, b = grads(0), grads(0)
    for x, y in zip(a, b):
        assert torch.equal(x, y), "default backward is not bit-reproducible"
    # the order in which the blend backward starts its tiles (heaviest first, planned from the forward's step counts) is
    # scheduling only: every wave writes its own slots, so the gradients are the same bits in either order
    # (0 index order, 1 tiles by the sum of their blocks, 2 tiles by their heaviest half -- the default --, 3 every half tile on its own)
    for order in (0, 1, 3):
        _lib.set_option("bwd_heavy_first", order)
        try:
            c = grads(0)
        finally:
            _lib.set_option("bwd_heavy_first", 2)
        for x, z in zip(a, c):
            assert torch.equal(x, z), f"launch order {order} of the blend backward changed a gradient"
    for variant, tol in ((1, 2e-4), (4, 5e-5), (5, 5e-5)):
        try:
            c = grads(variant)
        except _lib.GsrError:
            continue               # product build: A/B variants not compiled in
        for x, z in zip(a, c):
            assert (x - z).abs().max().item() <= tol * z.abs().max().item()


@pytest.mark.parametrize("scene", ["uniform", "edge_aa"])
def test_snug_tiles_change_no_bit(scene):
    """The product bins a Gaussian only into the tiles its alpha >= 1/255 ellipse reaches (csrc/gsr_math.h, `snug_tiles` = 1, the
    default); with `snug_tiles` = 0 it bins the reference's tile square.  Every instance the snug rectangle drops is skipped pixel
    by pixel by the blend, and the blend adds a pixel's terms one entry at a time: image, inverse depth and radii are THE SAME BITS
    either way, from lists a third shorter.  The gradients are the same sums of the same per-instance records; the reduce adds them
    in units of 256 records whose boundaries move with the emission indices, so they agree to fp32 summation order (1e-5 of the
    largest entry), not bit for bit."""
    from diff_gaussian_rasterization import GaussianRasterizer, _lib
    dev = torch.device("cuda:0")
    cam = make_camera(640, 360)
    aa = scene == "edge_aa"
    sc_cpu = make_edge_scene(20000, cam, seed=11) if aa else make_scene(60000, cam, seed=17, s_med=0.02)
    sc = sc_cpu.to(dev)
    s = oracle_settings(cam, antialiasing=aa, bg=torch.tensor([0.1, 0.2, 0.3]))
    wc = torch.randn(3, 360, 640, generator=torch.Generator().manual_seed(0)).to(dev)

    def run(snug):
        _lib.set_option("snug_tiles", snug)
        try:
            L = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
            m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
            col, radii, invd = GaussianRasterizer(gpu_settings(s, dev))(means3D=L[0], means2D=m2, opacities=L[2], shs=L[1],
                                                                        scales=L[3], rotations=L[4])
            ((col * wc).sum() + invd.sum()).backward()
            torch.cuda.synchronize()
            R = run_gpu(s, sc_cpu)["R"]
        finally:
            _lib.set_option("snug_tiles", 1)
        return [col.detach(), radii, invd.detach()] + [t.grad for t in L] + [m2.grad], R

    a, Ra = run(1)
    b, Rb = run(0)
    assert Ra < 0.85 * Rb, (Ra, Rb)
    for k, (x, y) in enumerate(zip(a[:3], b[:3])):
        assert torch.equal(x, y), f"forward output {k} changed with the tile rectangle ({(x.float() - y.float()).abs().max().item():.3e})"
    for k, (x, y) in enumerate(zip(a[3:], b[3:])):
        assert (x - y).abs().max().item() <= 1e-4 * y.abs().max().item(), f"gradient {k}: {(x - y).abs().max().item():.3e} of {y.abs().max().item():.3e}"




def test_backward_parity_edge_aa():
    _backward_case("edge_aa_scale", 1500, seed=3)


def test_backward_parity_colors_cov_nodepth():
    _backward_case("edge_lookat", 1500, colors_cov=True, use_depth=False, seed=5)


# ---------------------------------------------------------------------------------------------------
# known-answer tests straight against closed forms
# ---------------------------------------------------------------------------------------------------
def _single(dev, W=64, H=64, z=4.0, opacity=0.8, scale=0.05, color=(0.2, 0.5, 0.9), bg=(0.1, 0.2, 0.3), xy=(0.0, 0.0)):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    cam = make_camera(W, H).to(dev)
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.tensor(bg, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center,
                                       False, False, False)
    means = torch.tensor([[xy[0], xy[1], z]], device=dev)
    out = GaussianRasterizer(rs)(means3D=means, means2D=torch.zeros_like(means),
                                 opacities=torch.tensor([[opacity]], device=dev),
                                 colors_precomp=torch.tensor([color], device=dev),
                                 scales=torch.full((1, 3), scale, device=dev),
                                 rotations=torch.tensor([[1.0, 0, 0, 0]], device=dev))
    return cam, out


def test_kat_single_gaussian_centre_pixel():
    dev = torch.device("cuda:0")
    W = H = 64
    # a Gaussian at NDC (1/64, 1/64)*... choose xy so that it projects exactly on pixel centre (32, 32):
    # pix = ((ndc+1)*W-1)/2 = 32  =>  ndc = 1/64 ; ndc = x / (z tanfov)
    cam0 = make_camera(W, H)
    z = 4.0
    x = (1.0 / W) * z * cam0.tanfovx
    y = (1.0 / H) * z * cam0.tanfovy
    cam, (color, radii, invd) = _single(dev, W, H, z=z, xy=(x, y))
    c = color[:, 32, 32].cpu()
    alpha = min(0.99, 0.8)
    exp_c = torch.tensor([0.2, 0.5, 0.9]) * alpha + (1 - alpha) * torch.tensor([0.1, 0.2, 0.3])
    assert torch.allclose(c, exp_c, atol=2e-5), (c, exp_c)
    assert abs(invd[0, 32, 32].item() - alpha / z) < 1e-5
    assert radii.item() > 0
    # far corner: background only
    assert torch.allclose(color[:, 0, 0].cpu(), torch.tensor([0.1, 0.2, 0.3]), atol=1e-6)


def test_kat_culling_and_low_opacity():
    dev = torch.device("cuda:0")
    _, (color, radii, _) = _single(dev, z=0.15)          # behind the 0.2 near plane
    assert radii.item() == 0
    bg = torch.tensor([0.1, 0.2, 0.3])
    assert torch.allclose(color.cpu(), bg[:, None, None].expand(3, 64, 64), atol=0)
    _, (color, radii, _) = _single(dev, opacity=0.5 / 255.0)   # alpha < 1/255 everywhere -> image = bg
    assert radii.item() > 0
    assert torch.allclose(color.cpu(), bg[:, None, None].expand(3, 64, 64), atol=0)


def test_kat_depth_order_two_gaussians():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    W = H = 64
    cam = make_camera(W, H).to(dev)
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center,
                                       False, False, False)
    cam0 = make_camera(W, H)
    outs = []
    for order in ([0, 1], [1, 0]):   # the far one first in memory, then the near one first: same image
        z = torch.tensor([3.0, 6.0])[order]
        means = torch.stack([(1.0 / W) * z * cam0.tanfovx, (1.0 / H) * z * cam0.tanfovy, z], dim=1).to(dev)
        cols = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]])[order].to(dev)
        color, _, _ = GaussianRasterizer(rs)(means3D=means, means2D=torch.zeros_like(means),
                                             opacities=torch.full((2, 1), 0.6, device=dev), colors_precomp=cols,
                                             scales=torch.full((2, 3), 0.08, device=dev),
                                             rotations=torch.tensor([[1.0, 0, 0, 0]] * 2, device=dev))
        outs.append(color.cpu())
    assert torch.allclose(outs[0], outs[1], atol=1e-7)
    c = outs[0][:, 32, 32]
    assert abs(c[0].item() - 0.6) < 1e-5 and abs(c[1].item() - 0.4 * 0.6) < 1e-5


def test_degenerate_sizes_forward_and_backward():
    """P = 1, everything culled (R = 0: image = background, all gradients zero), and a 17x9 image smaller than one tile
    row -- forward and backward must not touch out-of-range memory or hang."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    for (W, H, z_vals) in ((64, 48, [4.0]), (64, 48, [0.1, -2.0, 0.05]), (17, 9, [3.0, 5.0]), (300, 2, [3.0, 5.0])):
        cam = make_camera(W, H).to(dev)
        rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.tensor([0.2, 0.4, 0.6], device=dev), 1.0,
                                           cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                           False, True, False)
        n = len(z_vals)
        means = torch.tensor([[0.0, 0.0, z] for z in z_vals], device=dev, requires_grad=True)
        shs = (torch.randn(n, 16, 3, device=dev) * 0.3).requires_grad_(True)
        op = torch.full((n, 1), 0.7, device=dev, requires_grad=True)
        sc = torch.full((n, 3), 0.1, device=dev, requires_grad=True)
        rot = torch.tensor([[1.0, 0, 0, 0]] * n, device=dev, requires_grad=True)
        m2 = torch.zeros(n, 3, device=dev, requires_grad=True)
        col, radii, invd = GaussianRasterizer(rs)(means3D=means, means2D=m2, opacities=op, shs=shs, scales=sc, rotations=rot)
        (col.sum() + invd.sum()).backward()
        torch.cuda.synchronize()
        assert col.shape == (3, H, W) and torch.isfinite(col).all()
        for t in (means, shs, op, sc, rot, m2):
            assert t.grad is not None and torch.isfinite(t.grad).all()
        if all(z <= 0.2 for z in z_vals):
            assert int(radii.abs().sum()) == 0
            assert torch.equal(col.cpu(), torch.tensor([0.2, 0.4, 0.6])[:, None, None].expand(3, H, W))
            assert all(float(t.grad.abs().max()) == 0.0 for t in (means, shs, op, sc, rot, m2))
        else:
            assert int((radii > 0).sum()) == n and float(op.grad.abs().sum()) > 0


def test_api_errors_empty_and_mark_visible():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    cam = make_camera(64, 48).to(dev)
    rs = GaussianRasterizationSettings(48, 64, cam.tanfovx, cam.tanfovy, torch.ones(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                       False, False, False)
    r = GaussianRasterizer(rs)
    m = torch.zeros(0, 3, device=dev)
    color, radii, invd = r(means3D=m, means2D=m, opacities=torch.zeros(0, 1, device=dev),
                           shs=torch.zeros(0, 16, 3, device=dev), scales=m, rotations=torch.zeros(0, 4, device=dev))
    assert color.shape == (3, 48, 64) and color.abs().max().item() == 0 and radii.numel() == 0   # zero, not bg
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=m, scales=m, rotations=m)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=m, shs=m)
    pts = torch.tensor([[0.0, 0, 1.0], [0, 0, 0.1], [0, 0, -3.0], [5.0, 5.0, 0.21]], device=dev)
    assert r.markVisible(pts).cpu().tolist() == [True, False, False, True]
    assert O.mark_visible(pts.cpu(), cam.world_view_transform.cpu()).tolist() == [True, False, False, True]


def test_debug_mode_writes_snapshots_on_failure(tmp_path, monkeypatch):
    """README.md:156-157 of the reference (`--debug`): with debug=True the inputs are copied to the CPU before the call and
    written to snapshot_fw.dump / snapshot_bw.dump when the rasterizer raises, then the error is re-raised
    (VERDICT r03 missing #7: the path existed, no test triggered it)."""
    import diff_gaussian_rasterization as D
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _lib
    monkeypatch.chdir(tmp_path)
    dev = torch.device("cuda:0")
    cam = make_camera(64, 48)
    sc = make_scene(50, cam, seed=3).to(dev)
    cd = cam.to(dev)

    def settings(debug):
        return GaussianRasterizationSettings(48, 64, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, cd.world_view_transform,
                                             cd.full_proj_transform, 3, cd.camera_center, False, debug, False)
    # forward: active degree 3 with only four coefficients per Gaussian -- refused by the library (GSR_ERR_INVALID_ARG)
    bad_shs = sc.shs[:, :4].contiguous()
    with pytest.raises(D.GsrError, match="coefficients"):
        GaussianRasterizer(settings(False))(means3D=sc.means3D, means2D=None, opacities=sc.opacities, shs=bad_shs, scales=sc.scales,
                                            rotations=sc.rotations)
    assert not (tmp_path / "snapshot_fw.dump").exists(), "no snapshot without debug=True"
    with pytest.raises(D.GsrError, match="coefficients"):
        GaussianRasterizer(settings(True))(means3D=sc.means3D, means2D=None, opacities=sc.opacities, shs=bad_shs, scales=sc.scales,
                                           rotations=sc.rotations)
    snap = torch.load(tmp_path / "snapshot_fw.dump", weights_only=False)
    assert torch.equal(snap[0], sc.means3D.cpu()) and torch.equal(snap[1], bad_shs.cpu()) and snap[0].device.type == "cpu"
    assert snap[-1].image_height == 48 and snap[-1].debug is True
    # backward: the library call is made to fail (an injected error: nothing a valid forward state can provoke)
    m = sc.means3D.clone().requires_grad_(True)
    color, radii, invd = GaussianRasterizer(settings(True))(means3D=m, means2D=None, opacities=sc.opacities, shs=sc.shs, scales=sc.scales,
                                                            rotations=sc.rotations)
    real_check = _lib.check

    def failing_check(rc, what=""):
        if "backward" in what:
            raise D.GsrError("injected failure in " + what)
        return real_check(rc, what)
    monkeypatch.setattr(_lib, "check", failing_check)
    with pytest.raises(D.GsrError, match="injected failure"):
        color.sum().backward()
    monkeypatch.setattr(_lib, "check", real_check)
    snap = torch.load(tmp_path / "snapshot_bw.dump", weights_only=False)
    assert torch.equal(snap[0], sc.means3D.cpu()) and torch.equal(snap[1], radii.cpu())
    assert snap[7].shape == (3, 48, 64) and float(snap[7].min()) == 1.0        # the incoming image gradient (d sum / d color = 1)


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[1] size: 1 M Gaussians @1080p -- size-independent properties (no oracle at this size)
# ---------------------------------------------------------------------------------------------------
def test_full_size_invariants_and_determinism():
    from diff_gaussian_rasterization import _lib
    dev = torch.device("cuda:0")
    cam = make_camera(1920, 1080)
    sc = make_scene(1_000_000, cam, seed=0)
    s = oracle_settings(cam)
    out = run_gpu(s, sc)
    P = sc.P
    R = out["R"]
    tt = out["tiles_touched"].long()
    assert int(tt.sum()) == R and R > 5_000_000
    assert not ((tt > 0) & ~(out["radii"] > 0)).any()      # (snug rectangles: a visible Gaussian may reach no tile)
    rng = out["ranges"].long()
    cnt = rng[:, 1] - rng[:, 0]
    nz = cnt > 0
    assert int(cnt.sum()) == R                          # ranges partition [0, R)
    st = rng[nz, 0]
    assert torch.equal(st[1:], rng[nz, 1][:-1]) and int(st[0]) == 0
    pl = out["point_list"].long()
    assert int(pl.min()) >= 0 and int(pl.max()) < P
    depth = out["splats"][:, 9]
    tile_of = torch.repeat_interleave(torch.arange(rng.shape[0], device=dev), cnt)
    d = depth[pl]
    same = tile_of[1:] == tile_of[:-1]
    assert bool((d[1:][same] >= d[:-1][same]).all())    # near-to-far inside every tile
    tie = same & (d[1:] == d[:-1])
    assert bool((pl[1:][tie] > pl[:-1][tie]).all())     # ties broken by Gaussian index (stable)
    # per-Gaussian multiplicity in the list equals tiles_touched
    assert torch.equal(torch.bincount(pl, minlength=P), tt)
    fT = out["final_T"]
    assert float(fT.min()) >= 0.0 and float(fT.max()) <= 1.0
    H, W = 1080, 1920
    ncb = out["n_contrib"].long()
    tid = (torch.arange(H, device=dev)[:, None] // 16) * ((W + 15) // 16) + torch.arange(W, device=dev)[None, :] // 16
    assert bool((ncb <= cnt[tid]).all())
    assert torch.isfinite(out["color"]).all()
    # determinism: bit-identical second run
    out2 = run_gpu(s, sc)
    assert torch.equal(out2["color"], out["color"]) and torch.equal(out2["point_list"], out["point_list"])


def test_full_size_workgroup_per_tile_variant_agrees():
    """Measurement build only (run_gpu skips when the A/B kernels are not compiled in): the workgroup-per-tile baseline blend
    (render_fwd_variant 1) agrees with the default kernel to fp32 noise at configs[1]."""
    cam = make_camera(1920, 1080)
    sc = make_scene(1_000_000, cam, seed=0)
    s = oracle_settings(cam)
    out3 = run_gpu(s, sc, variant=1)
    out = run_gpu(s, sc)
    diff = (out3["color"] - out["color"]).abs()
    assert float((diff > 1e-5).float().mean()) < 1e-3


def test_fused_adam_matches_torch_adam():
    """gsr_adam_step == torch.optim.Adam (eps=1e-15, per-group lr as in scene/gaussian_model.py:183-190)."""
    from gsr_optim import FusedAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    shapes = [(10007, 3), (10007, 16, 3), (10007, 1), (10007, 4), (5,)]
    lrs = [1.6e-4, 2.5e-3, 0.025, 1e-3, 0.01]
    a = [torch.randn(s, generator=g).to(dev).requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    oa = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(a, lrs)], lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=0.0, eps=1e-15)
    for it in range(7):
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, generator=g).to(dev) * (10.0 ** (it % 3 - 1))
            if it == 3:
                gr[::2] = 0          # zero gradients (invisible Gaussians): v stays, denominator ~ eps
            p.grad = gr.clone()
            q.grad = gr.clone()
        if it == 4:
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 8e-5      # update_learning_rate (train.py:91)
        oa.step()
        ob.step()
    torch.cuda.synchronize()
    for p, q, lr in zip(a, b, lrs):
        assert (p - q).abs().max().item() <= 2e-6 * max(1.0, lr / 1e-3), (p - q).abs().max().item()
    for p, q in zip(a, b):
        sa, sb = oa.state[p], ob.state[q]
        assert int(sa["step"]) == int(sb["step"])
        for key in ("exp_avg", "exp_avg_sq"):      # fp32 round-off of different but equivalent update forms (lerp / fma)
            assert (sa[key] - sb[key]).abs().max().item() <= 2e-6 * sb[key].abs().max().item()


def test_fused_adam_one_launch_equals_one_launch_per_tensor():
    """gsr_optim.FusedAdam steps all tensors of a model in one launch (gsr_adam_step_multi); every tensor must come out exactly as
    from its own gsr_adam_step launch -- odd sizes, a tensor that is not 16-byte aligned, different steps and learning rates."""
    import ctypes as C
    from diff_gaussian_rasterization import _lib
    from gsr_optim import FusedAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    shapes = [(1000, 3), (1000, 16, 3), (1000, 1), (1000, 3), (1000, 4), (777,), (5,), (4099, 3), (33, 16, 3), (12,)]      # > 8 tensors: two launches
    params = [torch.randn(sh, generator=g).to(dev) for sh in shapes]
    lrs = [1e-3 * (k + 1) for k in range(len(shapes))]
    a = [p.clone().requires_grad_(True) for p in params]
    b = [p.clone() for p in params]
    pad_a, pad_b = torch.empty(4 + 777, device=dev), torch.empty(4 + 777, device=dev)      # tensor 5: 4-byte aligned only, both ways
    pad_a[1:778] = params[5]
    pad_b[1:778] = params[5]
    a[5] = pad_a[1:778].requires_grad_(True)
    b[5] = pad_b[1:778]
    opt = FusedAdam([{"params": [q], "lr": lr} for q, lr in zip(a, lrs)], eps=1e-15)
    m = [torch.zeros_like(p) for p in b]
    v = [torch.zeros_like(p) for p in b]
    lib = _lib.load()
    for it in range(1, 4):
        grads = [torch.randn(p.shape, generator=g).to(dev) for p in params]
        for q, gr in zip(a, grads):
            q.grad = gr.clone()
        opt.step()
        for k in range(len(b)):
            _lib.check(lib.gsr_adam_step(C.c_void_p(b[k].data_ptr()), C.c_void_p(grads[k].data_ptr()), C.c_void_p(m[k].data_ptr()),
                                         C.c_void_p(v[k].data_ptr()), b[k].numel(), lrs[k], 0.9, 0.999, 1e-15, it,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gsr_adam_step")
    torch.cuda.synchronize()
    for k in range(len(b)):
        assert torch.equal(a[k].detach(), b[k]), k
        assert torch.equal(opt.state[a[k]]["exp_avg"], m[k]) and torch.equal(opt.state[a[k]]["exp_avg_sq"], v[k]), k


@pytest.fixture(params=[0, 1], ids=["marching", "tiled"])
def ssim_variant(request):
    """Both forms of the SSIM kernels (ssim.hip): marching waves (the default) and the LDS-tiled A/B form."""
    from diff_gaussian_rasterization import _lib
    _lib.set_option("ssim_variant", request.param)
    yield request.param
    _lib.set_option("ssim_variant", 0)


@pytest.mark.parametrize("shape", [(1, 3, 67, 93), (2, 3, 128, 160), (1, 1, 16, 16), (1, 3, 11, 300), (1, 3, 5, 7), (1, 2, 150, 55)])
def test_fused_ssim_matches_reference_formula(shape, ssim_variant):
    """fused_ssim (HIP) == utils/loss_utils.py:56-87 (restated in oracle.losses, which is pinned to the
    reference by tests/test_oracle.py::test_train_loss_matches_reference_loss_utils) -- value and gradient."""
    from fused_ssim import fused_ssim
    from oracle.losses import ssim as torch_ssim
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(shape[2])
    a = torch.rand(shape, generator=g)
    b = (a + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1)
    a1 = a.clone().to(dev).requires_grad_(True)
    a2 = a.clone().double().requires_grad_(True)
    v1 = fused_ssim(a1, b.to(dev))
    v2 = torch_ssim(a2, b.double())           # fp64 CPU reference of the same formula
    assert abs(v1.item() - v2.item()) < 2e-6
    (v1 * 3.0).backward()
    (v2 * 3.0).backward()
    torch.cuda.synchronize()
    d = (a1.grad.cpu().double() - a2.grad).abs().max().item()
    assert d <= 2e-5 * a2.grad.abs().max().item(), d
    with torch.no_grad():                      # inference form: no derivative maps kept
        assert abs(fused_ssim(a.to(dev), b.to(dev), train=False).item() - v2.item()) < 2e-6
    # the map-returning form (gsr_ssim_forward / backward) agrees with the mean form (gsr_ssim_mean_*)
    from fused_ssim import FusedSSIMMap
    a3 = a.clone().to(dev).requires_grad_(True)
    m3 = FusedSSIMMap.apply(a3, b.to(dev)).mean()
    assert abs(m3.item() - v1.item()) < 1e-6
    (m3 * 3.0).backward()
    assert (a3.grad - a1.grad).abs().max().item() <= 1e-6 * a1.grad.abs().max().item()


@pytest.mark.parametrize("shape,lam", [((3, 67, 93), 0.2), ((1, 3, 128, 160), 0.2), ((3, 16, 16), 0.5), ((3, 11, 300), 0.0), ((3, 40, 40), 1.0)])
def test_fused_train_loss_matches_reference_formula(shape, lam, ssim_variant):
    """SURVEY 8(f) N1 "fused SSIM + L1": fused_train_loss (ONE HIP kernel pair) == (1 - lambda) l1_loss + lambda (1 - ssim) of
    utils/loss_utils.py:40-87 / train.py:119-126 (restated in oracle.losses, pinned to the reference by the golden vector of
    tests/test_oracle.py::test_train_loss_matches_reference_loss_utils) -- value, the two read-outs and the gradient, incl.
    pixels where image == gt (torch's sign(0) = 0) and the pure-L1 / pure-SSIM mixes."""
    from fused_ssim import fused_train_loss
    from oracle.losses import train_loss, l1_loss, ssim as torch_ssim
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(shape[-1])
    a = torch.rand(shape, generator=g)
    b = (a + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1)
    b[..., :3, :5] = a[..., :3, :5]                    # exact ties: zero L1 gradient there
    a1 = a.clone().to(dev).requires_grad_(True)
    a2 = a.clone().double().requires_grad_(True)
    v1, l1_part, ssim_part = fused_train_loss(a1, b.to(dev), lam, return_parts=True)
    v2 = train_loss(a2, b.double(), lam)               # fp64 CPU reference of the same formula
    assert abs(v1.item() - v2.item()) < 2e-6
    assert abs(l1_part.item() - l1_loss(a.double(), b.double()).item()) < 1e-6
    assert abs(ssim_part.item() - torch_ssim(a.double(), b.double()).item()) < 2e-6
    assert not l1_part.requires_grad and not ssim_part.requires_grad
    (v1 * 3.0).backward()
    (v2 * 3.0).backward()
    torch.cuda.synchronize()
    d = (a1.grad.cpu().double() - a2.grad).abs().max().item()
    assert d <= 2e-5 * a2.grad.abs().max().item(), d
    with torch.no_grad():                               # inference form: no derivative maps kept
        assert abs(fused_train_loss(a.to(dev), b.to(dev), lam).item() - v2.item()) < 2e-6


@pytest.mark.parametrize("no_backward", [False, True], ids=["track", "inference"])
def test_more_than_65536_tiles_uses_32bit_tile_keys(no_backward):
    """4112 x 4112 pixels = 257 x 257 = 66 049 tiles: tile ids no longer fit 16 bits, so the binning switches to
    32-bit keys and a 3-pass tile sort.  Bins against the oracle bit-for-bit; the image on the tiles that hold splats."""
    W = H = 4112
    cam = make_camera(W, H)
    sc = make_scene(3000, cam, seed=41, s_med=0.004)
    s = oracle_settings(cam)
    out = run_gpu(s, sc, no_backward=no_backward)
    with torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        bins = O.bin_and_sort(pre)
    assert pre["grid"] == (257, 257)
    assert torch.equal(out["radii"].cpu(), pre["radii"].to(torch.int32))
    assert out["R"] == bins["R"] and bins["R"] > 3000
    assert torch.equal(out["point_list"].cpu().long(), bins["point_list"])
    assert torch.equal(out["ranges"].cpu().long(), bins["ranges"])
    # tiles beyond id 65535 must be populated, otherwise the case proves nothing
    assert int(bins["tile_counts"][65536:].sum()) > 0
    busy = torch.nonzero(bins["tile_counts"] > 0).flatten()
    sample = busy[:: max(1, busy.numel() // 40)].tolist()
    col, invd, fT, ncon, frag = O.render_tiles(pre, bins, s, tiles=sample, want_fragile=True)
    gx = 257
    gcol = out["color"].cpu()
    for t in sample:
        y0, x0 = (t // gx) * 16, (t % gx) * 16
        ys, xs = slice(y0, min(y0 + 16, H)), slice(x0, min(x0 + 16, W))
        ok = ~frag[ys, xs]
        d = (gcol[:, ys, xs] - col[:, ys, xs]).abs().max(0).values
        assert float(d[ok].max() if ok.any() else 0.0) <= IMG_TOL
    assert torch.isfinite(out["color"]).all()


def test_a_frame_counter_that_is_not_zero_costs_one_frame():
    """csrc/gsr_frame.h sums R in a device counter that must be zero when a frame begins.  Round 6 met one that was not (the control block of a second
    concurrent caller): it published a partial R early, was left non-zero again by the workgroups behind, and every later frame on that slot was wrong.
    The test hook `debug_dirty_control_block` preloads tickets into the NEXT frame's counter: after ONE stray ticket that frame may be wrong, the frames
    after it are the scene's own again (a slot alternates between two counters; the last workgroup clears both); with more tickets than workgroups no
    workgroup is the last, and the host -- stream idle, frame words unpublished -- clears the slot and refuses the frame."""
    from diff_gaussian_rasterization import GaussianRasterizer, _lib
    dev = torch.device("cuda:0")
    cam = make_camera(320, 240)
    scenes = [make_scene(30_000, cam, seed=41, s_med=0.02).to(dev), make_scene(9_000, cam, seed=42, s_med=0.05).to(dev)]
    rs = gpu_settings(oracle_settings(cam, bg=torch.tensor([0.1, 0.2, 0.3])), dev)

    def render(i):
        sc = scenes[i]
        with torch.no_grad():
            out = GaussianRasterizer(rs)(means3D=sc.means3D, means2D=None, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        torch.cuda.synchronize()
        return out
    want = render(0)
    _lib.set_option("debug_dirty_control_block", 1)
    render(1)                                               # (another scene: the leftover of a frame is exactly what the SAME frame would miss next time)
    for i in range(4):
        assert all(torch.equal(a, b) for a, b in zip(render(0), want)), f"frame {i + 1} after a stray ticket"
    _lib.set_option("debug_dirty_control_block", 1500)
    with pytest.raises(_lib.GsrError, match="never published"):
        render(1)
    for i in range(2):
        assert all(torch.equal(a, b) for a, b in zip(render(0), want)), f"frame {i + 1} after a refused frame"


def test_concurrent_forward_calls_from_two_host_threads():
    """Round 4 moved R (and the depth-key range) into a per-call device counter + mapped host word that the preprocess kernel's
    last workgroup publishes (csrc/gsr_frame.h); both are LEASED per call.  Two host threads rendering different scenes on their
    own streams at the same time must never see each other's counts: every frame of either thread equals the frame the same
    scene gives when rendered alone (bit for bit), over enough iterations for the leases to be recycled many times."""
    import threading
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    cams = [make_camera(320, 240), make_camera(400, 208)]
    scenes = [make_scene(30_000, cams[0], seed=41, s_med=0.02).to(dev), make_scene(9_000, cams[1], seed=42, s_med=0.05).to(dev)]
    sets = [gpu_settings(oracle_settings(c, bg=torch.tensor([0.1, 0.2, 0.3])), dev) for c in cams]

    def render(i):
        sc = scenes[i]
        with torch.no_grad():
            return GaussianRasterizer(sets[i])(means3D=sc.means3D, means2D=None, opacities=sc.opacities, shs=sc.shs, scales=sc.scales,
                                               rotations=sc.rotations)
    want = [tuple(t.clone() for t in render(i)) for i in range(2)]
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(60):
                    got = render(i)
                    st.synchronize()
                    for a, b in zip(got, want[i]):
                        if not torch.equal(a, b):
                            errors.append(f"thread {i}: a frame differs from the scene rendered alone")
                            return
        except Exception as ex:      # noqa: BLE001
            errors.append(f"thread {i}: {ex!r}")
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_wave_trace_of_the_blend_launches_accounts_for_every_step():
    """gsr_profile_enable(4) / gsr_profile_trace (measurement only): one record per wave of the blend kernels.  The records' steps add up to the
    work counters' totals, every traced wave ends after it starts, and tracing changes no output bit."""
    from diff_gaussian_rasterization import _lib, rasterize_gaussians
    from diff_gaussian_rasterization.debug import wave_timeline
    dev = torch.device("cuda:0")
    cam = make_camera(640, 360)
    sc = make_scene(60000, cam, seed=17, s_med=0.02).to(dev)
    s = oracle_settings(cam, bg=torch.tensor([0.1, 0.2, 0.3]))
    rs = gpu_settings(s, dev)
    wc = torch.randn(3, 360, 640, generator=torch.Generator().manual_seed(0)).to(dev)

    def run():
        L = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
        col, _, _ = rasterize_gaussians(L[0], None, L[1], None, L[2], L[3], L[4], None, rs)
        fw = _lib.profile_trace() if tracing[0] else None
        col.backward(wc)
        torch.cuda.synchronize()
        bw = _lib.profile_trace() if tracing[0] else None
        return col.detach(), [t.grad for t in L], fw, bw

    tracing = [False]
    _lib.profile_enable(False, counters=True)
    _lib.profile_counters(reset=True)
    col0, g0, _, _ = run()
    cnt = _lib.profile_counters(reset=True)
    tracing[0] = True
    _lib.profile_enable(False, trace=True)
    try:
        _lib.profile_trace()
        col1, g1, fw, bw = run()
    finally:
        _lib.profile_enable(False)
    assert torch.equal(col0, col1) and all(torch.equal(a, b) for a, b in zip(g0, g1)), "tracing changed an output"
    for tr, kernel, total in ((fw, 1, cnt["fwd_steps"]), (bw, 2, cnt["bwd_steps"])):
        t = tr[(tr[:, 1] != 0) & (((tr[:, 2] >> np.uint64(40)) & np.uint64(3)) == np.uint64(kernel))]
        assert len(t) > 0 and (t[:, 1] >= t[:, 0]).all()
        assert int((t[:, 3] & np.uint64(0xFFFF)).sum()) == total, (kernel, int((t[:, 3] & np.uint64(0xFFFF)).sum()), total)
        tl = wave_timeline(tr, kernel)
        assert tl["waves"] == len(t) and tl["span_us"] > 0 and -0.5 < tl["tail_loss"] < 2.0


def test_first_frames_of_concurrent_callers_in_fresh_processes():
    """The moment the library creates one control block per concurrent caller (gsr_api.cpp lease_host_word) only exists once per process per caller, so it is
    exercised in FRESH processes: four host threads render their first frames at the same time on their own non-blocking streams; every frame of every thread
    equals the frame of the same scene rendered alone, and nothing is left dirty behind them (tools/gpu_first_frame_stress.py; round 6: a block cleared with
    hipMemset -- the null stream, asynchronous on this runtime -- could lose its counter to the first kernel that counted in it)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_first_frame_stress.py"), "--trials", "3", "--threads", "4", "--frames", "6"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["trials"] == 3 and d["failed_trials"] == 0, d
