"""Pins the CPU oracle (oracle/torch_oracle.py) -- CPU only.

 * against the golden vectors generated from the REFERENCE's in-tree Python fragments
   (tests/golden/reference_fragments.npz, see tests/golden/make_golden.py for provenance),
 * against closed-form known answers,
 * against fp64 central finite differences of itself (gradient semantics),
 * against structural invariants,
 * against its own frozen outputs on BASELINE configs[0] (tests/golden/oracle_c1.npz).
The rasterizer's own source is absent from the reference, so parity is otherwise unpinned."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import O, make_camera, look_at_camera, make_scene, make_edge_scene, oracle_settings, reference_tiles
from gsr_synth import Camera

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def frag():
    return np.load(os.path.join(GOLD, "reference_fragments.npz"))


# ----------------------------------------------------------------------------- reference fragments
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_matches_reference_eval_sh(frag, deg):
    shs, xyz, campos = (torch.tensor(frag[k]) for k in ("sh_shs", "sh_xyz", "sh_campos"))
    rgb, clamped = O.eval_sh_colors(deg, shs, xyz, campos, torch.float32)
    ref = torch.tensor(frag[f"sh_rgb_deg{deg}"])
    assert torch.allclose(rgb, ref, atol=2e-6, rtol=0)
    assert torch.equal(clamped.any(dim=1), (ref == 0).any(dim=1)) or (ref == 0).sum() == clamped.sum()
    rgb64, _ = O.eval_sh_colors(deg, shs.double(), xyz.double(), campos.double(), torch.float64)
    assert torch.allclose(rgb64.float(), ref, atol=2e-6, rtol=0)


@pytest.mark.parametrize("mod", [1.0, 1.7])
def test_cov3d_matches_reference_build_covariance(frag, mod):
    scales, rots = torch.tensor(frag["cov_scales"]), torch.tensor(frag["cov_rots"])
    cov = O.compute_cov3d(scales, rots, mod, torch.float32)
    ref = torch.tensor(frag[f"cov_mod{mod}"])
    # the reference normalises q again and uses a batched matmul; off-diagonals cancel, so compare each
    # matrix against its own scale (fp32 round-off), not element-relative
    scale = ref.abs().amax(dim=1, keepdim=True)
    assert ((cov - ref).abs() <= 4e-6 * scale).all(), ((cov - ref).abs() / scale).max()
    # packing order [xx,xy,xz,yy,yz,zz] (utils/general_utils.py:64-73): diagonals are positive
    assert (cov[:, [0, 3, 5]] > 0).all()


def test_camera_conventions_match_reference(frag):
    from gsr_synth import make_camera as mk
    R, T = torch.tensor(frag["cam_R"]), torch.tensor(frag["cam_T"])
    fovx, fovy = frag["cam_fov"]
    # the reference builds W2C with R transposed (utils/graphics_utils.py:40): W2C[:3,:3] = R^T
    w, h = 640, int(round(640 * math.tan(fovy / 2) / math.tan(fovx / 2)))
    cam = mk(w, h, math.degrees(fovx), R=R.t().float(), t=T.float())
    assert np.allclose(cam.world_view_transform.numpy(), frag["cam_wvt"], atol=1e-6)
    assert np.allclose(cam.camera_center.numpy(), frag["cam_center"], atol=1e-5)
    # projection entries (fovy of `cam` is derived from square pixels, so compare the x column + z rows)
    from gsr_synth import projection_matrix
    P = projection_matrix(0.01, 100.0, float(fovx), float(fovy)).transpose(0, 1)
    assert np.allclose(P.numpy(), frag["cam_proj"], atol=1e-6)
    full = torch.tensor(frag["cam_wvt"]).unsqueeze(0).bmm(P.unsqueeze(0)).squeeze(0)
    assert np.allclose(full.numpy(), frag["cam_full"], atol=1e-5)
    # the oracle reads the flat matrices column-major: view-space z of a point == (W2C @ p)[2]
    p = torch.tensor([[0.3, -0.4, 2.0]])
    w2c = torch.tensor(frag["cam_wvt"]).t()
    z_ref = (w2c[:3, :3] @ p[0] + w2c[:3, 3])[2]
    s = O.Settings(h, w, math.tan(fovx / 2), math.tan(fovy / 2), torch.zeros(3), 1.0, torch.tensor(frag["cam_wvt"]),
                   torch.tensor(frag["cam_full"]), 0, torch.tensor(frag["cam_center"]), False, False, False)
    pre = O.preprocess(p, torch.ones(1, 1), s, colors_precomp=torch.ones(1, 3), scales=torch.full((1, 3), 0.1),
                       rotations=torch.tensor([[1.0, 0, 0, 0]]))
    assert abs(pre["depths"][0].item() - z_ref.item()) < 1e-6
    # and the pixel centre equals the reference projection convention ((ndc+1)*W-1)/2
    hom = torch.cat([p[0], torch.ones(1)]) @ torch.tensor(frag["cam_full"])
    ndc = hom[:2] / (hom[3] + 1e-7)
    assert abs(pre["means2D"][0, 0].item() - ((ndc[0].item() + 1) * w - 1) / 2) < 1e-3


def test_sh_python_path_equals_internal_sh_path():
    """colors_precomp from the reference-style SH evaluation == shs path (image and radii identical)."""
    cam = make_camera(128, 96)
    sc = make_scene(600, cam, seed=4, s_med=0.04)
    s = oracle_settings(cam)
    with torch.no_grad():
        a = O.rasterize(sc.means3D, None, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        rgb, _ = O.eval_sh_colors(3, sc.shs, sc.means3D, cam.camera_center, torch.float32)
        b = O.rasterize(sc.means3D, None, sc.opacities, s, colors_precomp=rgb, scales=sc.scales, rotations=sc.rotations)
        cov = O.compute_cov3d(sc.scales, sc.rotations, 1.0, torch.float32)
        c = O.rasterize(sc.means3D, None, sc.opacities, s, shs=sc.shs, cov3D_precomp=cov)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])


# ----------------------------------------------------------------------------- known answers
def _one(W=64, H=64, z=4.0, opacity=0.8, scale=0.05, color=(0.2, 0.5, 0.9), bg=(0.1, 0.2, 0.3), centre=True,
         scale_modifier=1.0, dtype=torch.float32):
    cam = make_camera(W, H)
    x = (1.0 / W) * z * cam.tanfovx if centre else 0.0
    y = (1.0 / H) * z * cam.tanfovy if centre else 0.0
    s = O.settings_from_camera(cam, torch.tensor(bg), 0, scale_modifier)
    means = torch.tensor([[x, y, z]], dtype=dtype)
    return O.rasterize(means, None, torch.tensor([[opacity]], dtype=dtype), s,
                       colors_precomp=torch.tensor([color], dtype=dtype),
                       scales=torch.full((1, 3), scale, dtype=dtype),
                       rotations=torch.tensor([[1.0, 0, 0, 0]], dtype=dtype), return_aux=True)


def test_kat_single_gaussian_on_pixel_centre():
    col, radii, invd, aux = _one()
    assert abs(aux["means2D"][0, 0].item() - 32.0) < 1e-4 and abs(aux["means2D"][0, 1].item() - 32.0) < 1e-4
    alpha = 0.8
    exp = torch.tensor([0.2, 0.5, 0.9]) * alpha + (1 - alpha) * torch.tensor([0.1, 0.2, 0.3])
    assert torch.allclose(col[:, 32, 32], exp, atol=2e-5)
    assert abs(aux["final_T"][32, 32].item() - (1 - alpha)) < 1e-6
    assert abs(invd[0, 32, 32].item() - alpha / 4.0) < 1e-6
    assert aux["n_contrib"][32, 32].item() == 1
    # closed-form covariance on the optical axis: isotropic sigma -> Sigma2D = (f s / z)^2 + 0.3 on the diagonal
    col, radii, invd, aux = _one(centre=False)
    assert abs(aux["means2D"][0, 0].item() - 31.5) < 1e-4
    cam = make_camera(64, 64)
    f = 64 / (2 * cam.tanfovx)
    v = (f * 0.05 / 4.0) ** 2 + 0.3
    assert abs(aux["conic"][0, 0].item() - 1 / v) < 1e-4 and abs(aux["conic"][0, 1].item()) < 1e-6
    # lambda_max = mid + sqrt(max(0.1, mid^2 - det)) and the discriminant is 0 here -> the 0.1 floor applies
    assert radii.item() == math.ceil(3 * math.sqrt(v + math.sqrt(0.1)))


def test_kat_alpha_cap_and_cull_and_threshold():
    col, radii, invd, aux = _one(opacity=1.0)
    assert abs(aux["final_T"][32, 32].item() - 0.01) < 1e-6            # alpha capped at 0.99
    col, radii, _, aux = _one(z=0.15)
    assert radii.item() == 0 and aux["R"] == 0                        # near-plane cull (z <= 0.2)
    assert torch.equal(col, torch.tensor([0.1, 0.2, 0.3])[:, None, None].expand(3, 64, 64))
    col, radii, _, aux = _one(opacity=0.9 / 255)
    assert radii.item() > 0 and aux["n_contrib"].max().item() == 0     # alpha < 1/255 everywhere
    assert torch.equal(col, torch.tensor([0.1, 0.2, 0.3])[:, None, None].expand(3, 64, 64))


def test_kat_scale_modifier_equals_scaled_scales():
    a = _one(scale=0.05, scale_modifier=2.0)
    b = _one(scale=0.10, scale_modifier=1.0)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_kat_sh_degree0_is_view_independent():
    cam = make_camera(64, 64)
    s = O.settings_from_camera(cam, torch.zeros(3), 0)
    shs = torch.zeros(2, 16, 3)
    shs[:, 0] = torch.tensor([[0.7, -0.2, -5.0], [0.7, -0.2, -5.0]])
    shs[:, 1:] = 3.0                                                     # must be ignored at degree 0
    means = torch.tensor([[0.5, 0.2, 3.0], [-1.0, 0.4, 6.0]])
    rgb, clamped = O.eval_sh_colors(0, shs, means, cam.camera_center, torch.float32)
    exp = torch.clamp_min(0.28209479177387814 * shs[:, 0] + 0.5, 0)
    assert torch.allclose(rgb, exp, atol=1e-7) and torch.equal(rgb[0], rgb[1])
    assert clamped[:, 2].all() and not clamped[:, :2].any()


def test_kat_early_termination_and_order():
    """Stack of near-opaque Gaussians on one pixel: blending stops once T would drop below 1e-4 and the
    terminating Gaussian is NOT blended."""
    cam = make_camera(64, 64)
    s = O.settings_from_camera(cam, torch.zeros(3), 0)
    n = 6
    z = torch.linspace(3.0, 8.0, n)
    means = torch.stack([(1.0 / 64) * z * cam.tanfovx, (1.0 / 64) * z * cam.tanfovy, z], dim=1)
    perm = torch.tensor([3, 0, 5, 1, 4, 2])                               # memory order != depth order
    cols = torch.eye(3).repeat(2, 1)
    out = O.rasterize(means[perm], None, torch.full((n, 1), 0.95), s, colors_precomp=cols[perm],
                      scales=torch.full((n, 3), 0.2), rotations=torch.tensor([[1.0, 0, 0, 0]] * n), return_aux=True)
    col, _, _, aux = out
    # T after k contributors = 0.05^k ; 0.05^3 = 1.25e-4 ok, 0.05^4 = 6.25e-6 < 1e-4 -> 3 contributors
    assert aux["n_contrib"][32, 32].item() == 3
    assert abs(aux["final_T"][32, 32].item() - 0.05 ** 3) < 1e-9
    w = [0.95, 0.95 * 0.05, 0.95 * 0.05 ** 2]
    assert torch.allclose(col[:, 32, 32], torch.tensor(w), atol=1e-6)     # colours e0,e1,e2 in depth order


def test_tie_order_follows_gaussian_index():
    cam = make_camera(64, 64)
    s = O.settings_from_camera(cam, torch.zeros(3), 0)
    means = torch.tensor([[0.0, 0.0, 4.0]] * 3)
    cols = torch.eye(3)
    _, _, _, aux = O.rasterize(means, None, torch.full((3, 1), 0.5), s, colors_precomp=cols,
                               scales=torch.full((3, 3), 0.05), rotations=torch.tensor([[1.0, 0, 0, 0]] * 3),
                               return_aux=True)
    pl = aux["point_list"]
    rng = aux["ranges"]
    for t in range(rng.shape[0]):
        a, b = int(rng[t, 0]), int(rng[t, 1])
        if b > a:
            assert pl[a:b].tolist() == sorted(pl[a:b].tolist())


# ----------------------------------------------------------------------------- snug tile rectangles
@pytest.mark.parametrize("maker,aa,size", [(lambda c: make_scene(3000, c, seed=2, s_med=0.03), False, (250, 131)),
                                           (lambda c: make_edge_scene(3000, c, seed=8), True, (250, 131)),
                                           (lambda c: make_edge_scene(2000, c, seed=21), False, (333, 200)),
                                           (lambda c: make_scene(4000, c, seed=5, s_med=0.006), True, (640, 360))])
def test_snug_tiles_change_no_output(maker, aa, size):
    """The product bins a Gaussian only into the tiles its alpha >= 1/255 ellipse can reach (csrc/gsr_math.h, restated in
    O.preprocess(snug=True)); the reference bins the square of radius 3 sqrt(lambda_max).  Every instance the snug rectangle
    drops is one the reference's blend skips pixel by pixel, so colour, inverse depth, final transmittance and radii are the SAME
    BITS either way; only the lists get shorter (and the contributor positions move with them)."""
    cam = look_at_camera(size[0], size[1], (0.2, 0.1, -0.6), (0.0, 0.0, 4.0))
    sc = maker(cam)
    s = oracle_settings(cam, antialiasing=aa, bg=torch.tensor([0.2, 0.3, 0.4]))
    kw = dict(shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    outs = {}
    for snug in (False, True):
        with torch.no_grad():
            pre = O.preprocess(sc.means3D, sc.opacities, s, snug=snug, **kw)
            bins = O.bin_and_sort(pre)
            col, invd, fT, ncon, _ = O.render_tiles(pre, bins, s, False)
        outs[snug] = (pre, bins, col, invd, fT, ncon)
    (pre0, bins0, col0, invd0, fT0, n0), (pre1, bins1, col1, invd1, fT1, n1) = outs[False], outs[True]
    # (the oracle adds a pixel's terms with a vectorised sum whose pairing depends on the list length: a few ulp; the product adds
    #  them one entry at a time, and tests/test_gpu_parity.py::test_snug_tiles_change_no_bit compares ITS two ways bit for bit)
    assert torch.equal(fT0, fT1)
    assert (col0 - col1).abs().max().item() <= 1e-6 and (invd0 - invd1).abs().max().item() <= 1e-6 * max(1.0, float(invd0.max()))
    assert torch.equal(pre0["radii"], pre1["radii"])
    R0, R1 = int(bins0["R"]), int(bins1["R"])
    assert R1 < 0.9 * R0, (R0, R1)
    # the snug rectangle lies inside the reference's, and the snug list of every tile is a subsequence of the reference's
    r0, r1 = pre0["rect"], pre1["rect"]
    listed = pre1["tiles_touched"] > 0
    assert (r1[listed, 0] >= r0[listed, 0]).all() and (r1[listed, 2] <= r0[listed, 2]).all()
    assert (r1[listed, 1] >= r0[listed, 1]).all() and (r1[listed, 3] <= r0[listed, 3]).all()
    for t in range(0, bins0["ranges"].shape[0], 7):
        a = bins0["point_list"][bins0["ranges"][t, 0]:bins0["ranges"][t, 1]].tolist()
        b = bins1["point_list"][bins1["ranges"][t, 0]:bins1["ranges"][t, 1]].tolist()
        it = iter(a)
        assert all(x in it for x in b), t
    # the contributor count of a pixel never grows, and the last contributor is the same Gaussian
    assert (n1 <= n0).all()
    gx = pre0["grid"][0]
    H, W = size[1], size[0]
    tid = (torch.arange(H)[:, None] // 16) * gx + torch.arange(W)[None, :] // 16
    has = n1 > 0
    g0 = bins0["point_list"][(bins0["ranges"][tid, 0] + n0 - 1)[has]]
    g1 = bins1["point_list"][(bins1["ranges"][tid, 0] + n1 - 1)[has]]
    assert torch.equal(g0, g1)
    assert torch.equal(n0 > 0, n1 > 0)


def test_deterministic_logarithm_is_a_logarithm():
    """det_log (frexp + atanh series in fp64, the restatement of csrc/gsr_math.h gsr_log_det) against libm over 90 decades, and
    tau = 2 ln(255 opacity) + 0.01 at the edges of its domain."""
    v = torch.exp(torch.linspace(-103.0, 103.0, 200001, dtype=torch.float64))
    v = torch.cat([v, torch.tensor([1.0, 2.0, 0.5, 255.0, 1.0 + 2 ** -52, 2.0 ** -149 * 255.0], dtype=torch.float64)])
    err = (O.det_log(v) - torch.log(v)).abs()
    assert float(err.max()) <= 3e-14 * max(1.0, float(torch.log(v).abs().max())), float(err.max())
    assert float(O.det_log(torch.tensor([1.0], dtype=torch.float64))) == 0.0
    op = torch.tensor([0.0, 1.0 / 255.0, 0.5, 1.0, float("inf"), float("nan"), -0.1, 1e-45], dtype=torch.float32)
    tau = O.tau_of_opacity(op)
    assert tau[0] == -math.inf and abs(float(tau[1]) - 0.01) < 1e-6 and abs(float(tau[2]) - (2 * math.log(127.5) + 0.01)) < 1e-6
    assert abs(float(tau[3]) - (2 * math.log(255.0) + 0.01)) < 1e-6 and tau[4] == math.inf and math.isnan(float(tau[5])) and math.isnan(float(tau[6]))
    assert float(tau[7]) < -190.0          # a denormal opacity: finite, far below every q


@pytest.mark.parametrize("seed,aa", [(3, False), (4, True)])
def test_snug_rectangle_holds_every_pixel_that_can_contribute(seed, aa):
    """Pixel by pixel, for adversarial splats -- needles with aspect ratios up to 1e4 at every angle, splats larger than the frame,
    opacities from far below 1/255 to 1, centres off screen -- with the blend's own fp32 arithmetic: wherever
    alpha = opacity exp(-q/2) reaches 1/255 inside the reference's tile square, the pixel's tile lies inside the snug rectangle."""
    W, H = 208, 144
    cam = look_at_camera(W, H, (0.3, -0.2, -0.5), (0.0, 0.0, 4.0))
    g = torch.Generator().manual_seed(seed)
    P = 800
    sc = make_scene(P, cam, seed=seed, s_med=0.05, overscan=1.5)
    logs = torch.rand(P, 3, generator=g) * 9.2 - 6.9                  # scales 1e-3 .. 10, independent per axis
    sc.scales[:] = torch.exp(logs)
    sc.rotations[:] = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=1)
    sc.opacities[:] = torch.exp(torch.rand(P, 1, generator=g) * 7.0 - 7.0)      # 9e-4 .. 1
    sc.opacities[::7] = (1.0 / 255.0) * (1.0 + (torch.rand(sc.opacities[::7].shape, generator=g) - 0.5) * 0.02)   # on the threshold
    s = oracle_settings(cam, antialiasing=aa)
    with torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations, snug=True)
        ref = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations, snug=False)
    vis = ref["tiles_touched"] > 0
    assert int(vis.sum()) > 150
    m2, con, op, rect, rrect = pre["means2D"][vis], pre["conic"][vis], pre["opacity"][vis].reshape(-1), pre["rect"][vis], ref["rect"][vis]
    px = torch.arange(W, dtype=torch.float32)[None, None, :]
    py = torch.arange(H, dtype=torch.float32)[None, :, None]
    dx = m2[:, 0, None, None] - px
    dy = m2[:, 1, None, None] - py
    # the reference's expression (forward.cu renderCUDA) and the product's (mul, fma, fma in log2 units): either may decide a pixel
    power = -0.5 * (con[:, 0, None, None] * dx * dx + con[:, 2, None, None] * dy * dy) - con[:, 1, None, None] * dx * dy
    alpha = torch.minimum(torch.tensor(0.99), op[:, None, None] * torch.exp(power))
    hit = (power <= 0) & (alpha >= 1.0 / 255.0)
    tx = (torch.arange(W) // 16)[None, None, :]
    ty = (torch.arange(H) // 16)[None, :, None]
    inside = (tx >= rect[:, 0, None, None]) & (tx < rect[:, 2, None, None]) & (ty >= rect[:, 1, None, None]) & (ty < rect[:, 3, None, None])
    # (the reference itself never blends a splat outside its 3-sigma tile square, however opaque: only pixels in there count)
    listed = (tx >= rrect[:, 0, None, None]) & (tx < rrect[:, 2, None, None]) & (ty >= rrect[:, 1, None, None]) & (ty < rrect[:, 3, None, None])
    hit = hit & listed
    assert int(hit.sum()) > 5000
    missed = hit & ~inside
    assert not missed.any(), f"{int(missed.sum())} contributing pixels outside the snug rectangle (Gaussians {torch.nonzero(missed.flatten(1).any(1)).flatten()[:5].tolist()})"
    # and the rectangles do shrink
    assert int(pre["tiles_touched"].sum()) < 0.8 * int(ref["tiles_touched"].sum())


# ----------------------------------------------------------------------------- invariants
@pytest.mark.parametrize("maker,aa", [(lambda c: make_scene(1500, c, seed=2, s_med=0.03), False),
                                      (lambda c: make_edge_scene(1500, c, seed=8), True)])
def test_invariants(maker, aa):
    cam = look_at_camera(250, 131, (0.2, 0.1, -0.6), (0.0, 0.0, 4.0))
    sc = maker(cam)
    s = oracle_settings(cam, antialiasing=aa, bg=torch.tensor([0.2, 0.3, 0.4]))
    with torch.no_grad():
        col, radii, invd, aux = O.rasterize(sc.means3D, None, sc.opacities, s, shs=sc.shs, scales=sc.scales,
                                            rotations=sc.rotations, return_aux=True)
    R = aux["R"]
    assert int(aux["tiles_touched"].sum()) == R
    rng = aux["ranges"]
    cnt = rng[:, 1] - rng[:, 0]
    assert int(cnt.sum()) == R
    nz = cnt > 0
    assert torch.equal(rng[nz, 0][1:], rng[nz, 1][:-1])
    gx = aux["grid"][0]
    H, W = 131, 250
    tid = (torch.arange(H)[:, None] // 16) * gx + torch.arange(W)[None, :] // 16
    assert (aux["n_contrib"] <= cnt[tid]).all()
    assert (aux["final_T"] >= 0).all() and (aux["final_T"] <= 1).all()
    assert not ((aux["tiles_touched"] > 0) & ~(radii > 0)).any()      # (snug rectangles: a visible Gaussian may reach no tile)
    # permuting the Gaussians changes nothing except tie order among bit-identical depths
    dv = aux["depths"][radii > 0]
    if torch.unique(dv).numel() != dv.numel():
        return
    perm = torch.randperm(sc.P, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        col2, radii2, _ = O.rasterize(sc.means3D[perm], None, sc.opacities[perm], s, shs=sc.shs[perm],
                                      scales=sc.scales[perm], rotations=sc.rotations[perm])
    assert torch.equal(radii2, radii[perm])
    assert torch.allclose(col2, col, atol=1e-6)


def test_band_rendering_tiles_the_full_image():
    cam = make_camera(200, 150)
    sc = make_edge_scene(1200, cam, seed=12)
    s = oracle_settings(cam, bg=torch.tensor([0.5, 0.5, 0.5]))
    with torch.no_grad():
        full = O.rasterize(sc.means3D, None, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        parts = [O.rasterize(sc.means3D, None, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations,
                             tile_y0=a, tile_y1=b) for a, b in ((0, 3), (3, 7), (7, 10))]
    img = sum(p[0] for p in parts)
    assert torch.equal(img, full[0])
    for p in parts:
        assert torch.equal(p[1], full[1])          # radii unaffected by the band


def test_frozen_c1_outputs():
    g = np.load(os.path.join(GOLD, "oracle_c1.npz"))
    cam = make_camera(256, 256)
    sc = make_scene(1000, cam, seed=0)
    s = oracle_settings(cam)
    with torch.no_grad(), reference_tiles():      # the frozen lists are the reference's (R = 1711 is SURVEY 8(d)'s probe value)
        col, radii, invd, aux = O.rasterize(sc.means3D, None, sc.opacities, s, shs=sc.shs, scales=sc.scales,
                                            rotations=sc.rotations, return_aux=True)
    assert aux["R"] == int(g["R"]) == 1711 and int((radii > 0).sum()) == 874     # SURVEY 8(d) probe values
    assert np.array_equal(radii.numpy(), g["radii"])
    assert np.array_equal(aux["tiles_touched"].numpy(), g["tiles_touched"])
    assert np.array_equal(aux["point_list"].numpy(), g["point_list"])
    assert np.array_equal(aux["ranges"].numpy(), g["ranges"])
    assert np.array_equal(aux["n_contrib"].numpy(), g["n_contrib"])
    assert np.allclose(col.double().sum(dim=(0, 2)).numpy(), g["color_rowsum"], atol=1e-4)
    assert np.allclose(col.numpy(), g["color"].astype(np.float32), atol=2e-3)


# ----------------------------------------------------------------------------- gradients vs fp64 finite differences
def _fd_scene(aa):
    cam = look_at_camera(48, 40, (0.1, 0.05, -0.3), (0.0, 0.0, 3.0))
    g = torch.Generator().manual_seed(5)
    P = 14
    z = torch.rand(P, generator=g) * 2 + 2.0
    means_c = torch.stack([(torch.rand(P, generator=g) - 0.5) * z * 0.8, (torch.rand(P, generator=g) - 0.5) * z * 0.6, z], 1)
    w2c = cam.world_view_transform.t()
    means = ((means_c - w2c[:3, 3]) @ w2c[:3, :3]).double()
    scales = (torch.rand(P, 3, generator=g) * 0.25 + 0.08).double()
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g)).double()
    opac = (torch.rand(P, 1, generator=g) * 0.6 + 0.2).double()
    shs = (torch.randn(P, 16, 3, generator=g) * 0.2).double()
    shs[:, 0] += 1.0   # keep colours away from the clamp
    s = O.Settings(40, 48, cam.tanfovx, cam.tanfovy, torch.tensor([0.2, 0.1, 0.3]), 1.0, cam.world_view_transform,
                   cam.full_proj_transform, 3, cam.camera_center, False, False, aa)
    wc = torch.randn(3, 40, 48, generator=g).double()
    wd = torch.randn(1, 40, 48, generator=g).double()
    return s, means, scales, rots, opac, shs, wc, wd


@pytest.mark.parametrize("aa", [False, True])
def test_gradients_match_fp64_finite_differences(aa):
    s, means, scales, rots, opac, shs, wc, wd = _fd_scene(aa)

    def f(m, sc, r, o, sh):
        col, _, invd = O.rasterize(m, None, o, s, shs=sh, scales=sc, rotations=r)
        return (col * wc).sum() + (invd * wd).sum()

    leaves = [t.clone().requires_grad_(True) for t in (means, scales, rots, opac, shs)]
    f(*leaves).backward()
    rng = np.random.default_rng(0)
    eps = 1e-6
    for li, name in enumerate(["means3D", "scales", "rotations", "opacities", "shs"]):
        base = [means, scales, rots, opac, shs]
        g = leaves[li].grad
        flat_n = base[li].numel()
        for k in rng.choice(flat_n, size=min(10, flat_n), replace=False):
            d = torch.zeros(flat_n, dtype=torch.float64)
            d[k] = eps
            d = d.view_as(base[li])
            with torch.no_grad():
                args_p = [b + d if i == li else b for i, b in enumerate(base)]
                args_m = [b - d if i == li else b for i, b in enumerate(base)]
                fd = (f(*args_p) - f(*args_m)).item() / (2 * eps)
            an = g.view(-1)[k].item()
            # the conic backward deliberately uses 1/(det^2+1e-7) (reference convention): <= ~2e-5 relative
            assert abs(fd - an) <= 5e-5 * max(1.0, abs(fd), g.abs().max().item()), (name, int(k), fd, an)


def test_means2D_gradient_is_ndc_scaled_pixel_gradient():
    s, means, scales, rots, opac, shs, wc, wd = _fd_scene(False)
    m2 = torch.zeros(means.shape[0], 3, dtype=torch.float64, requires_grad=True)
    col, _, invd, aux = O.rasterize(means, m2, opac, s, shs=shs, scales=scales, rotations=rots, return_aux=True)
    ((col * wc).sum()).backward()
    # finite difference on the pixel centre itself
    pre = O.preprocess(means, opac, s, shs=shs, scales=scales, rotations=rots)
    bins = O.bin_and_sort(pre)

    def img_loss(dxy):
        p2 = dict(pre)
        p2["means2D"] = pre["means2D"] + dxy
        c, *_ = O.render_tiles(p2, bins, s)
        return (c * wc).sum().item()

    k = int(torch.argmax(m2.grad[:, 0].abs()))
    eps = 1e-6
    d = torch.zeros_like(pre["means2D"])
    d[k, 0] = eps
    with torch.no_grad():
        fd = (img_loss(d) - img_loss(-d)) / (2 * eps)
    assert abs(m2.grad[k, 0].item() - fd * 0.5 * 48) <= 1e-5 * max(1.0, abs(fd * 24))
    assert m2.grad[:, 2].abs().max().item() == 0


def test_train_loss_matches_reference_loss_utils(frag):
    """oracle.losses (the checker of the fused HIP loss kernels) == utils/loss_utils.py l1_loss / ssim of the reference."""
    from oracle.losses import ssim, l1_loss, train_loss
    img, gt = torch.tensor(frag["loss_img"]), torch.tensor(frag["loss_gt"])
    assert abs(ssim(img, gt).item() - float(frag["loss_ssim"])) < 1e-6
    assert abs(l1_loss(img, gt).item() - float(frag["loss_l1"])) < 1e-7
    assert abs(train_loss(img, gt).item() - (0.8 * float(frag["loss_l1"]) + 0.2 * (1 - float(frag["loss_ssim"])))) < 1e-6
