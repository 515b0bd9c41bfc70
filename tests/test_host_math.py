"""CPU test of the PRODUCT's per-Gaussian math (csrc/gsr_math.h, compiled for the host by g++) against the
oracle: forward integer outputs bit-exact, float outputs bit-exact (same fp32 expression trees, no FMA),
backward against oracle autograd.  Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import O, host_cam, host_math_lib, fptr, np32, make_camera, look_at_camera, make_scene, make_edge_scene, oracle_settings


def run_host_fwd(s, sc, scale_mod=1.0, colors=None, cov=None, tile_rows=(0, 0)):
    lib = host_math_lib()
    P = sc.P
    M = sc.shs.shape[1]
    hc = host_cam(s, M, *tile_rows)
    out_f = np.zeros((P, 12), np.float32)
    out_i = np.zeros((P, 8), np.int32)
    out_cov = np.zeros((P, 6), np.float32)
    lib.host_preprocess(C.byref(hc), P, fptr(np32(sc.means3D)), fptr(np32(sc.scales)) if cov is None else None,
                        fptr(np32(sc.rotations)) if cov is None else None, fptr(np32(cov)), fptr(np32(sc.opacities)),
                        fptr(np32(sc.shs)) if colors is None else None, fptr(np32(colors)), fptr(out_f), fptr(out_i),
                        fptr(out_cov))
    return out_f, out_i, out_cov


def _needle_scene(cam, seed=3, P=4000):
    """Adversarial footprints for the snug tile rectangle: per-axis scales from 1e-3 to 10 (aspect ratios up to 1e4 at every
    angle, splats larger than the frame), opacities from 9e-4 to 1 and a seventh of them within 1 % of the 1/255 threshold."""
    g = torch.Generator().manual_seed(seed)
    sc = make_scene(P, cam, seed=seed, s_med=0.05, overscan=1.5)
    sc.scales[:] = torch.exp(torch.rand(P, 3, generator=g) * 9.2 - 6.9)
    sc.rotations[:] = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=1)
    sc.opacities[:] = torch.exp(torch.rand(P, 1, generator=g) * 7.0 - 7.0)
    sc.opacities[::7] = (1.0 / 255.0) * (1.0 + (torch.rand(sc.opacities[::7].shape, generator=g) - 0.5) * 0.02)
    return sc


CASES = [
    ("c1", lambda: make_camera(256, 256), lambda cam: make_scene(1000, cam, seed=0), False),
    ("needles", lambda: look_at_camera(640, 360, (0.3, -0.2, -0.5), (0.0, 0.0, 4.0)), lambda cam: _needle_scene(cam), False),
    ("needles_aa", lambda: look_at_camera(333, 200, (0.1, 0.2, -0.5), (0.0, 0.0, 4.0)), lambda cam: _needle_scene(cam, seed=9), True),
    ("odd_aa", lambda: make_camera(250, 131), lambda cam: make_scene(3000, cam, seed=3, s_med=0.02), True),
    ("edge_lookat", lambda: look_at_camera(333, 200, (0.3, -0.2, -1.0), (0.1, 0.0, 3.0)),
     lambda cam: make_edge_scene(4000, cam, seed=5), False),
    ("edge_aa", lambda: make_camera(480, 270), lambda cam: make_edge_scene(4000, cam, seed=7), True),
]


@pytest.mark.parametrize("name,mkcam,mkscene,aa", CASES)
def test_forward_matches_oracle_bitwise(name, mkcam, mkscene, aa):
    cam = mkcam()
    sc = mkscene(cam)
    s = oracle_settings(cam, antialiasing=aa)
    with torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    out_f, out_i, out_cov = run_host_fwd(s, sc)
    vis = pre["visible"].numpy()
    assert vis.sum() > 50
    # integer-deciding outputs: bit exact
    np.testing.assert_array_equal(out_i[:, 0], pre["radii"].numpy())
    np.testing.assert_array_equal(out_i[:, 5], pre["tiles_touched"].numpy())
    np.testing.assert_array_equal(out_i[:, 7].astype(bool), vis)
    rect = pre["rect"].numpy()
    np.testing.assert_array_equal(out_i[vis][:, 1:5], rect[vis])
    # float outputs of visible Gaussians: same expression trees -> bit exact
    np.testing.assert_array_equal(out_cov, pre["cov3D"].numpy())
    np.testing.assert_array_equal(out_f[vis][:, 0:2], pre["means2D"].numpy()[vis])
    np.testing.assert_array_equal(out_f[vis][:, 2:5], pre["conic"].numpy()[vis])
    np.testing.assert_array_equal(out_f[vis][:, 5], pre["opacity"].numpy()[vis])
    np.testing.assert_array_equal(out_f[vis][:, 9], pre["depths"].numpy()[vis])
    # tau = 2 ln(255 opacity) + 0.01 decides the snug tile rectangle: its logarithm is a fixed sequence of IEEE operations
    np.testing.assert_array_equal(out_f[vis][:, 10], O.tau_of_opacity(pre["opacity"]).numpy()[vis])
    assert int(pre["tiles_touched"].sum()) < int(O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales,
                                                                rotations=sc.rotations, snug=False)["tiles_touched"].sum())
    # SH colours go through one more op chain; x86 and torch evaluate them identically as well
    np.testing.assert_allclose(out_f[vis][:, 6:9], pre["rgb"].numpy()[vis], rtol=0, atol=1e-6)
    cl = pre["clamped"].numpy()
    bits = (cl[:, 0].astype(np.int32) | (cl[:, 1].astype(np.int32) << 1) | (cl[:, 2].astype(np.int32) << 2))
    np.testing.assert_array_equal(out_i[vis][:, 6], bits[vis])


def test_band_restriction_matches_oracle():
    cam = make_camera(320, 240)
    sc = make_scene(2000, cam, seed=11, s_med=0.03)
    s = oracle_settings(cam)
    with torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations,
                           tile_y0=4, tile_y1=9)
    _, out_i, _ = run_host_fwd(s, sc, tile_rows=(4, 9))
    np.testing.assert_array_equal(out_i[:, 5], pre["tiles_touched"].numpy())
    np.testing.assert_array_equal(out_i[:, 0], pre["radii"].numpy())
    assert 0 < out_i[:, 5].sum() < (out_i[:, 0] > 0).sum() * 50


@pytest.mark.parametrize("aa,use_cov", [(False, False), (True, False), (True, True)])
def test_backward_matches_oracle_autograd(aa, use_cov):
    torch.manual_seed(0)
    cam = look_at_camera(200, 150, (0.2, 0.1, -0.5), (0.0, 0.0, 4.0))
    sc = make_edge_scene(1500, cam, seed=21)
    s = oracle_settings(cam, antialiasing=aa)
    P, M = sc.P, sc.shs.shape[1]
    m = sc.means3D.clone().requires_grad_(True)
    op = sc.opacities.clone().requires_grad_(True)
    sh = sc.shs.clone().requires_grad_(True)
    scl = sc.scales.clone().requires_grad_(True)
    rot = sc.rotations.clone().requires_grad_(True)
    cov = None
    if use_cov:
        cov = O.compute_cov3d(sc.scales, sc.rotations, 1.0, torch.float32).detach().clone().requires_grad_(True)
        pre = O.preprocess(m, op, s, shs=sh, cov3D_precomp=cov)
    else:
        pre = O.preprocess(m, op, s, shs=sh, scales=scl, rotations=rot)
    vis = pre["visible"]
    g = torch.randn(P, 12) * vis[:, None]
    g[:, 10:] = 0
    loss = ((g[:, 0:2] * pre["means2D"]).sum() + (g[:, 2:5] * pre["conic"]).sum() + (g[:, 5] * pre["opacity"]).sum()
            + (g[:, 6:9] * pre["rgb"]).sum() + (g[:, 9] * (1.0 / pre["depths"]))[vis].sum())
    loss.backward()

    lib = host_math_lib()
    hc = host_cam(s, M)
    cl = pre["clamped"].numpy()
    bits = (cl[:, 0].astype(np.uint32) | (cl[:, 1].astype(np.uint32) << 1) | (cl[:, 2].astype(np.uint32) << 2)).astype(np.uint32)
    radii = np.ascontiguousarray(pre["radii"].numpy().astype(np.int32))
    o = {k: np.zeros(shape, np.float32) for k, shape in dict(m2=(P, 3), col=(P, 3), op=(P,), m3=(P, 3), cov=(P, 6),
                                                               sh=(P, M, 3), sc=(P, 3), rot=(P, 4)).items()}
    lib.host_preprocess_backward(C.byref(hc), P, fptr(np32(sc.means3D)), None if use_cov else fptr(np32(sc.scales)),
                                 None if use_cov else fptr(np32(sc.rotations)), fptr(np32(cov)) if use_cov else None,
                                 fptr(np32(sc.opacities)), fptr(np32(sc.shs)), fptr(radii), fptr(bits), fptr(np32(g)),
                                 fptr(o["m2"]), fptr(o["col"]), fptr(o["op"]), fptr(o["m3"]), fptr(o["cov"]), fptr(o["sh"]),
                                 None if use_cov else fptr(o["sc"]), None if use_cov else fptr(o["rot"]))

    def close(a, b, name, rtol=2e-4):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        scale = np.abs(b).max() + 1e-30
        err = np.abs(a - b).max() / scale
        assert err < rtol, f"{name}: max rel-to-max error {err:.3e}"

    close(o["m3"], m.grad.numpy(), "dL_dmeans3D")
    close(o["op"], op.grad.numpy().reshape(-1), "dL_dopacity")
    close(o["sh"], sh.grad.numpy(), "dL_dsh")
    if use_cov:
        close(o["cov"], cov.grad.numpy(), "dL_dcov3D")
    else:
        close(o["sc"], scl.grad.numpy(), "dL_dscales")
        close(o["rot"], rot.grad.numpy(), "dL_drotations")
    # means2D gradient is the pixel gradient in NDC-scaled units
    np.testing.assert_allclose(o["m2"][:, 0], g[:, 0].numpy() * 0.5 * cam.image_width, rtol=1e-6)
    np.testing.assert_allclose(o["m2"][:, 1], g[:, 1].numpy() * 0.5 * cam.image_height, rtol=1e-6)


@pytest.mark.parametrize("deg,M", [(0, 16), (1, 16), (2, 16), (1, 4), (2, 9), (0, 1)])
def test_sh_backward_all_degrees_and_ragged_blocks(deg, M):
    """The grouped (4 coefficients = 48 bytes at a time) SH backward for every active degree and for SH blocks whose
    coefficient count is not a multiple of 4 (max degree 0 / 2)."""
    cam = look_at_camera(160, 120, (0.1, 0.2, -0.4), (0.0, 0.0, 3.0))
    sc = make_scene(400, cam, seed=31, s_med=0.05)
    sc.shs = sc.shs[:, :M].contiguous()
    s = oracle_settings(cam, sh_degree=deg)
    P = sc.P
    m = sc.means3D.clone().requires_grad_(True)
    sh = sc.shs.clone().requires_grad_(True)
    pre = O.preprocess(m, sc.opacities, s, shs=sh, scales=sc.scales, rotations=sc.rotations)
    vis = pre["visible"]
    g = torch.zeros(P, 12)
    g[:, 6:9] = torch.randn(P, 3, generator=torch.Generator().manual_seed(1)) * vis[:, None]
    (g[:, 6:9] * pre["rgb"]).sum().backward()
    lib = host_math_lib()
    hc = host_cam(s, M)
    cl = pre["clamped"].numpy()
    bits = (cl[:, 0].astype(np.uint32) | (cl[:, 1].astype(np.uint32) << 1) | (cl[:, 2].astype(np.uint32) << 2)).astype(np.uint32)
    radii = np.ascontiguousarray(pre["radii"].numpy().astype(np.int32))
    o = {k: np.full(shape, 7.0, np.float32) for k, shape in dict(m2=(P, 3), col=(P, 3), op=(P,), m3=(P, 3), cov=(P, 6),
                                                                   sh=(P, M, 3), sc=(P, 3), rot=(P, 4)).items()}
    lib.host_preprocess_backward(C.byref(hc), P, fptr(np32(sc.means3D)), fptr(np32(sc.scales)), fptr(np32(sc.rotations)), None,
                                 fptr(np32(sc.opacities)), fptr(np32(sc.shs)), fptr(radii), fptr(bits), fptr(np32(g)),
                                 fptr(o["m2"]), fptr(o["col"]), fptr(o["op"]), fptr(o["m3"]), fptr(o["cov"]), fptr(o["sh"]),
                                 fptr(o["sc"]), fptr(o["rot"]))
    ref = sh.grad.numpy()
    assert np.abs(o["sh"] - ref).max() <= 2e-6 * max(1e-6, np.abs(ref).max())
    assert np.abs(o["sh"][:, (deg + 1) ** 2:]).max() == 0 if M > (deg + 1) ** 2 else True
    mref = m.grad.numpy() if m.grad is not None else np.zeros((P, 3), np.float32)   # degree 0 is view independent
    assert np.abs(o["m3"] - mref).max() <= 2e-4 * max(1e-9, np.abs(mref).max())


def test_tau_bitwise_over_the_whole_opacity_range():
    """gsr_tau (host build of csrc/gsr_math.h) == the oracle's restatement, bit for bit: 200 000 opacities from denormal to 1,
    the threshold 1/255 and its neighbours, 0, infinities and NaN."""
    lib = host_math_lib()
    g = torch.Generator().manual_seed(11)
    op = torch.cat([torch.exp(torch.rand(200000, generator=g) * 100.0 - 100.0), torch.rand(20000, generator=g),
                    torch.tensor([0.0, 1.0, 1.0 / 255.0, float(np.nextafter(np.float32(1 / 255), np.float32(1))),
                                  float(np.nextafter(np.float32(1 / 255), np.float32(0))), 1e-45, float("inf"), -0.5])]).to(torch.float32)
    a = np32(op)
    out = np.empty_like(a)
    lib.host_tau(len(a), fptr(a), fptr(out))
    ref = O.tau_of_opacity(op).numpy()
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))
    nan_in = np.array([np.nan], dtype=np.float32)
    nan_out = np.empty(1, dtype=np.float32)
    lib.host_tau(1, fptr(nan_in), fptr(nan_out))
    assert np.isnan(nan_out[0]) and np.isnan(O.tau_of_opacity(torch.tensor([float("nan")])).numpy()[0])
