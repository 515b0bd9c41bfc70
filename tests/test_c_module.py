"""`diff_gaussian_rasterization._C` (part of the import surface) -- the extension-module names of the reference,
answered by the C ABI.

* `fusedssim` / `fusedssim_backward`: the reference's OWN utils/loss_utils.py is imported unchanged (where /root/reference exists) and its
  `fast_ssim` (loss_utils.py:24-38,89-91 -- which calls `_C.fusedssim(C1, C2, img1, img2)` and `_C.fusedssim_backward(C1, C2, img1, img2, dL)`)
  must equal its `ssim` (loss_utils.py:56-87), value and gradient.  Elsewhere (the GPU box) the same two are restated here.
* `rasterize_gaussians` / `rasterize_gaussians_backward` / `mark_visible`: the positional forms against the `GaussianRasterizer` module.
* `adamUpdate`: against `SparseGaussianAdam.step`.

CPU leg: the library is the kernel source run through tests/simt; GPU leg: the shipped library on cuda:0."""
import importlib
import os
import sys
from unittest import mock

import pytest
import torch
import torch.nn.functional as F

from helpers import make_camera, make_scene

REF = "/root/reference"
HAVE_REF = os.path.exists(os.path.join(REF, "utils", "loss_utils.py"))


def _reference_loss_utils(pkg):
    """utils/loss_utils.py of the reference, imported with `diff_gaussian_rasterization` resolved to `pkg`."""
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        sys.modules.pop(k, None)
    sys.path.insert(0, REF)
    try:
        c_mod = importlib.import_module(pkg.__name__ + "._C")
        with mock.patch.dict(sys.modules, {"diff_gaussian_rasterization": pkg, "diff_gaussian_rasterization._C": c_mod}):
            mod = importlib.import_module("utils.loss_utils")
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
            sys.modules.pop(k, None)
    assert hasattr(mod, "fusedssim") and hasattr(mod, "fusedssim_backward"), "the reference's try/except import of _C failed"
    return mod


class _Restated:
    """utils/loss_utils.py:21-38,56-91 restated (the GPU box has no /root/reference); pinned to the import above by the CPU test."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2

    def __init__(self, c_mod):
        outer = self

        class FusedSSIMMap(torch.autograd.Function):
            @staticmethod
            def forward(ctx, C1, C2, img1, img2):
                ctx.save_for_backward(img1.detach(), img2)
                ctx.C1, ctx.C2 = C1, C2
                return c_mod.fusedssim(C1, C2, img1, img2)

            @staticmethod
            def backward(ctx, g):
                img1, img2 = ctx.saved_tensors
                return None, None, c_mod.fusedssim_backward(ctx.C1, ctx.C2, img1, img2, g), None
        self._map = FusedSSIMMap
        self.outer = outer

    def fast_ssim(self, a, b):
        return self._map.apply(self.C1, self.C2, a, b).mean()

    @staticmethod
    def ssim(img1, img2):
        ch = img1.size(-3)
        g = torch.tensor([pow(2.718281828459045, -(x - 5) ** 2 / 4.5) for x in range(11)], dtype=torch.float64)
        g = (g / g.sum()).float().unsqueeze(1)
        w = (g @ g.t()).unsqueeze(0).unsqueeze(0).expand(ch, 1, 11, 11).contiguous().to(img1)
        mu1, mu2 = F.conv2d(img1, w, padding=5, groups=ch), F.conv2d(img2, w, padding=5, groups=ch)
        s1 = F.conv2d(img1 * img1, w, padding=5, groups=ch) - mu1 * mu1
        s2 = F.conv2d(img2 * img2, w, padding=5, groups=ch) - mu2 * mu2
        s12 = F.conv2d(img1 * img2, w, padding=5, groups=ch) - mu1 * mu2
        m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
        return m.mean()


def _images(dev, shape=(1, 3, 67, 93)):
    g = torch.Generator().manual_seed(3)
    a = torch.rand(*shape, generator=g)
    b = (a + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
    return a.to(dev), b.to(dev)


def _fast_ssim_equals_ssim(L, dev):
    a, b = _images(dev)
    a1 = a.clone().requires_grad_(True)
    v1 = L.fast_ssim(a1, b)
    v1.backward()
    a2 = a.clone().requires_grad_(True)
    v2 = L.ssim(a2, b)
    v2.backward()
    assert abs(float(v1) - float(v2)) < 2e-6, (float(v1), float(v2))
    scale = float(a2.grad.abs().max())
    assert float((a1.grad - a2.grad).abs().max()) <= 2e-5 * scale
    with pytest.raises(Exception, match="C1"):      # other constants are refused loudly, not silently ignored
        L_c = sys.modules.get("diff_gaussian_rasterization._C") or importlib.import_module("diff_gaussian_rasterization._C")
        L_c.fusedssim(0.5, 0.03 ** 2, a, b)


def _positional_forms_equal_the_module(pkg, dev):
    C_ = importlib.import_module(pkg.__name__ + "._C")
    cam = make_camera(96, 64)
    sc = make_scene(400, cam, seed=4, s_med=0.05).to(dev)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    rs = pkg.GaussianRasterizationSettings(64, 96, cam.tanfovx, cam.tanfovy, bg, 1.0, cam.world_view_transform.to(dev),
                                           cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False, False)
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    L = {k: getattr(sc, k).detach().clone().requires_grad_(True) for k in names}
    m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
    col, radii, invd = pkg.GaussianRasterizer(rs)(means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"],
                                                  rotations=L["rotations"])
    g = torch.Generator().manual_seed(8)
    wc, wd = torch.randn(3, 64, 96, generator=g).to(dev), torch.randn(1, 64, 96, generator=g).to(dev)
    ((col * wc).sum() + (invd * wd).sum()).backward()
    e = torch.empty(0, device=dev)
    nr, col2, radii2, geom, binning, img, invd2 = C_.rasterize_gaussians(bg, sc.means3D, e, sc.opacities, sc.scales, sc.rotations, 1.0, e, rs.viewmatrix,
                                                                         rs.projmatrix, rs.tanfovx, rs.tanfovy, 64, 96, sc.shs, 3, rs.campos, False, False, False)
    assert nr > 0 and torch.equal(col2, col.detach()) and torch.equal(radii2, radii) and torch.equal(invd2, invd.detach())
    dm2, dcol, dop, dm3, dcov, dsh, dsc, drot = C_.rasterize_gaussians_backward(bg, sc.means3D, radii2, e, sc.opacities, sc.scales, sc.rotations, 1.0, e,
                                                                                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, wc, wd, sc.shs, 3,
                                                                                rs.campos, geom, nr, binning, img, False, False)
    assert dcol.numel() == 0 and dcov.numel() == 0
    for a, b in ((dm2, m2.grad), (dop, L["opacities"].grad), (dm3, L["means3D"].grad), (dsh, L["shs"].grad), (dsc, L["scales"].grad),
                 (drot, L["rotations"].grad)):
        assert torch.equal(a.reshape(b.shape), b)      # the same kernels on the same inputs: the same bits
    assert torch.equal(C_.mark_visible(sc.means3D, rs.viewmatrix, rs.projmatrix), pkg.GaussianRasterizer(rs).markVisible(sc.means3D))
    # adamUpdate == SparseGaussianAdam.step on one tensor
    p = torch.nn.Parameter(sc.means3D.clone())
    p.grad = torch.randn(sc.P, 3, generator=g).to(dev)
    vis = radii > 0
    opt = pkg.SparseGaussianAdam([{"params": [p], "lr": 0.01, "name": "xyz"}], lr=0.0, eps=1e-15)
    q, m, v = sc.means3D.clone(), torch.zeros_like(sc.means3D), torch.zeros_like(sc.means3D)
    for _ in range(2):
        opt.step(vis, sc.P)
        C_.adamUpdate(q, p.grad, m, v, vis, 0.01, 0.9, 0.999, 1e-15, sc.P, 3)
    assert torch.equal(q, p.detach()) and torch.equal(m, opt.state[p]["exp_avg"]) and torch.equal(v, opt.state[p]["exp_avg_sq"])
    assert not torch.equal(q, sc.means3D) and torch.equal(q[~vis], sc.means3D[~vis])


# ---------------------------------------------------------------- CPU: the kernel source through tests/simt
def test_reference_fast_ssim_imported_unchanged_equals_ssim_on_the_cpu():
    import simt_build
    from test_simt_package_cpu import package_on_the_cpu
    with package_on_the_cpu(simt_build.build_library()) as pkg:
        if HAVE_REF:
            L = _reference_loss_utils(pkg)
            a, b = _images("cpu")
            assert abs(float(L.ssim(a, b)) - float(_Restated.ssim(a, b))) < 1e-6      # pins the restatement the GPU leg uses
        else:
            L = _Restated(importlib.import_module(pkg.__name__ + "._C"))
        _fast_ssim_equals_ssim(L, "cpu")


def test_positional_extension_entry_points_on_the_cpu():
    import simt_build
    from test_simt_package_cpu import package_on_the_cpu
    with package_on_the_cpu(simt_build.build_library()) as pkg:
        _positional_forms_equal_the_module(pkg, "cpu")


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_fast_ssim_through_the_c_module_equals_ssim_on_the_gpu():
    import diff_gaussian_rasterization as pkg
    _fast_ssim_equals_ssim(_Restated(importlib.import_module(pkg.__name__ + "._C")), torch.device("cuda:0"))


@pytest.mark.gpu
def test_positional_extension_entry_points_on_the_gpu():
    import diff_gaussian_rasterization as pkg
    _positional_forms_equal_the_module(pkg, torch.device("cuda:0"))
