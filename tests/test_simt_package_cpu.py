"""The drop-in as a whole, without a GPU: the reference's render glue -> THIS repo's `diff_gaussian_rasterization` package exactly as shipped
(settings record, `GaussianRasterizer`, the autograd Function, the resize callbacks, `_lib.load()` with its own argument types) -> the real C ABI
-> gsr_api.cpp -> every kernel of csrc/, the last two compiled for the host through tests/simt (each lane a fiber).  `GSR_LIB` points the package's
own loader at tests/_build/libgsr_simt.so; the only things replaced in the package are the three places that ask torch for a HIP device
(`_require_cuda`, `_stream_ptr`, `torch.cuda.device`), because the tensors behind the pointers live in host memory here.

The glue is the reference's own `gaussian_renderer.render()` where /root/reference exists (the build container) and the restated `_render()` of
tests/test_gpu_reference_glue.py elsewhere (pinned to it bit for bit by tests/test_reference_render_cpu.py).  The bars are those of the GPU test.
Test infrastructure: the product never loads this library."""
import contextlib
import os
from unittest import mock

import pytest
import torch

from helpers import O
import simt_build
import test_gpu_reference_glue as G
import test_reference_render_cpu as RR

HAVE_REF = os.path.exists(os.path.join(RR.REF, "gaussian_renderer", "__init__.py"))


@pytest.fixture(scope="module")
def simt_lib():
    return simt_build.build_library()


@contextlib.contextmanager
def package_on_the_cpu(path):
    import diff_gaussian_rasterization as pkg
    from diff_gaussian_rasterization import _lib
    saved, saved_env = _lib._lib, os.environ.get("GSR_LIB")
    _lib._lib = None
    os.environ["GSR_LIB"] = path
    real_cdll = _lib.C.CDLL
    try:
        # (RTLD_LOCAL: the host build must not enter the process's global symbol scope, where a product library loaded later would bind to it)
        with mock.patch.object(_lib.C, "CDLL", lambda p, mode=0: real_cdll(p)), mock.patch.object(pkg, "_require_cuda", lambda *a: None), mock.patch.object(pkg, "_stream_ptr", lambda d: None), \
                mock.patch.object(torch.cuda, "device", lambda d: contextlib.nullcontext()), \
                mock.patch.object(torch, "zeros_like", RR._cpu_zeros_like(torch.zeros_like)):
            lib = _lib.load()      # the package's own loader: ABI version, export list, argument types
            assert lib._name == path
            yield pkg
    finally:
        _lib._lib = saved
        if saved_env is None:
            os.environ.pop("GSR_LIB", None)
        else:
            os.environ["GSR_LIB"] = saved_env


@pytest.mark.parametrize("glue", ["reference", "restated"])
@pytest.mark.parametrize("case", list(G.CASES))
def test_render_glue_through_the_package_and_the_kernels_on_the_cpu(simt_lib, case, glue):
    if glue == "reference" and not HAVE_REF:
        pytest.skip("reference tree not present")
    opt, cam, sc, gt, mono, dmask, col, pipe, bg = RR._case_inputs(case)
    kw = dict(scaling_modifier=opt.get("scaling_modifier", 1.0), separate_sh=opt.get("separate_sh", False), use_trained_exp=opt.get("use_trained_exp", False))
    oc = col if opt.get("override_color") else None
    with package_on_the_cpu(simt_lib) as pkg:
        pc = G._Model(sc, torch.device("cpu"), opt.get("active_sh_degree", 3))
        if glue == "reference":
            out = RR._import_reference_render(pkg).render(RR._Camera(cam), pc, pipe, bg, override_color=oc, **kw)
        else:
            out = G._render(cam, pc, pipe, bg, pkg.GaussianRasterizationSettings, pkg.GaussianRasterizer, override_color=oc, **kw)
        got = RR._finish(out, pc, gt, mono, dmask)
    pc2 = G._Model(sc, torch.device("cpu"), opt.get("active_sh_degree", 3))
    want = RR._finish(G._render(cam, pc2, pipe, bg, O.Settings, G._OracleRasterizer, override_color=oc, **kw), pc2, gt, mono, dmask)
    (ch, dh, rh, vh, lh, gh), (ci, di, ri, vi, li, gi) = got, want
    assert rh.dtype == torch.int32 and torch.equal(rh, ri.to(torch.int32)) and torch.equal(vh, vi) and int((ri > 0).sum()) > 50
    err = (ch - ci).abs().amax(0)
    assert float((err > 1e-5).float().mean()) < 0.01 and float(err.max()) < 1.1 / 255.0      # (tests/test_gpu_reference_glue.py's bars from here on)
    assert float(((dh - di).abs() > 1e-5).float().mean()) < 0.01
    assert abs(lh - li) < 1e-5
    for k, a in gi.items():
        b = gh[k]
        if a is None:
            assert b is None or float(b.abs().max()) == 0.0, k
            continue
        assert b is not None, f"{k}: no gradient from the package"
        scale = float(a.abs().max())
        if scale == 0.0:
            assert float(b.abs().max()) == 0.0, k
            continue
        d = (a.double() - b.double()).abs() / scale
        assert float(d.max()) < 2e-3 and float(torch.quantile(d.flatten()[:2_000_000], 0.999)) < 1e-4, (k, float(d.max()))
    na = torch.norm(gi["viewspace_points"][vi[:, 0] if vi.dim() > 1 else vi, :2], dim=-1)      # the statistic the densifier reads
    nb = torch.norm(gh["viewspace_points"][vh[:, 0] if vh.dim() > 1 else vh, :2], dim=-1)
    assert float((na - nb).abs().max()) <= 2e-3 * float(na.max())


def test_inference_call_and_mark_visible_through_the_package_on_the_cpu(simt_lib):
    """`torch.no_grad()` rendering takes the no-tracking build (render.py of the reference) and `markVisible` is the frustum test of
    cuda_rasterizer/auxiliary.h:141-166 -- both through the package."""
    opt, cam, sc, gt, mono, dmask, col, pipe, bg = RR._case_inputs("default")
    args = lambda S: S(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, bg, 1.0, cam.world_view_transform, cam.full_proj_transform, 3,      # noqa: E731
                       cam.camera_center, False, False, False)
    call = dict(means3D=sc.means3D, means2D=torch.zeros(sc.P, 3), opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    with package_on_the_cpu(simt_lib) as pkg, torch.no_grad():
        rast = pkg.GaussianRasterizer(args(pkg.GaussianRasterizationSettings))
        color, radii, invd = rast(**call)
        vis = rast.markVisible(sc.means3D)
    with torch.no_grad():
        color_o, radii_o, invd_o = O.rasterize(call["means3D"], call["means2D"], call["opacities"], args(O.Settings), shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    assert torch.equal(radii, radii_o.to(torch.int32)) and int((radii > 0).sum()) > 50
    err = (color - color_o).abs().amax(0)
    assert float((err > 1e-5).float().mean()) < 0.01 and float(err.max()) < 1.1 / 255.0
    assert float(((invd - invd_o).abs() > 1e-5).float().mean()) < 0.01
    p = torch.cat((sc.means3D, torch.ones(sc.P, 1)), dim=1) @ cam.world_view_transform
    assert torch.equal(vis.bool(), p[:, 2] > 0.2) and 0 < int(vis.sum()) <= sc.P
