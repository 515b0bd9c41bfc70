"""The reference's OWN render glue, unchanged, in the build container (CPU): `/root/reference/gaussian_renderer/__init__.py::render` is
imported as it stands and executed -- screen-space tensor, settings record, every branch of the call, exposure, clamp, returned dict, then
the L1 + inverse-depth loss of train.py:112-142 and `backward()` -- against (1) the CPU oracle behind the operator's call interface and
(2) THIS repo's `diff_gaussian_rasterization` package (its settings record, `GaussianRasterizer` module and autograd Function run as shipped;
only the native calls underneath, which need a GPU, are answered by the oracle through the same C-ABI argument lists).

What this pins: the step-by-step restatement `_render()` that the GPU tests drive (`tests/test_gpu_reference_glue.py`, the reference tree does
not travel to the GPU box) IS the reference's glue -- same image, radii, visibility filter and gradients bit for bit through the same operator --
and the package accepts exactly the calls the reference makes and hands back what `train.py` reads (`viewspace_points.grad`, `radii`,
`visibility_filter`, `depth`).  Test infrastructure: the oracle stands in for the GPU here and nowhere else."""
import ctypes as C
import importlib
import math
import os
import sys
import types
from unittest import mock

import numpy as np
import pytest
import torch

from helpers import O, look_at_camera, make_camera, make_edge_scene, make_scene
import simple_knn._C      # noqa: F401  (resolved with the real package before `diff_gaussian_rasterization` is swapped for a stub below)
import test_gpu_reference_glue as G

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "gaussian_renderer", "__init__.py")), reason="reference tree not present")


def _import_reference_render(operator_module):
    """gaussian_renderer/__init__.py with `diff_gaussian_rasterization` resolved to `operator_module`; scene/__init__.py (dataset readers, PIL)
    is not executed -- only scene/gaussian_model.py, which the glue imports for a type annotation."""
    for k in [k for k in sys.modules if k == "gaussian_renderer" or k == "utils" or k.startswith("utils.") or k == "scene" or k.startswith("scene.")]:
        sys.modules.pop(k, None)
    pkg = types.ModuleType("scene")
    pkg.__path__ = [os.path.join(REF, "scene")]
    sys.path.insert(0, REF)
    try:
        with mock.patch.dict(sys.modules, {"scene": pkg, "diff_gaussian_rasterization": operator_module}):
            mod = importlib.import_module("gaussian_renderer")
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "gaussian_renderer" or k == "utils" or k.startswith("utils.") or k.startswith("scene.")]:
            sys.modules.pop(k, None)
    return mod


def _cpu_zeros_like(real):
    def f(*a, **k):
        if k.get("device") == "cuda":      # the glue asks for "cuda" by name (gaussian_renderer/__init__.py:26); there is none here
            k.pop("device")
        return real(*a, **k)
    return f


class _Camera:
    def __init__(self, cam):
        self.FoVx, self.FoVy = cam.FoVx, cam.FoVy
        self.image_height, self.image_width = cam.image_height, cam.image_width
        self.world_view_transform, self.full_proj_transform, self.camera_center = cam.world_view_transform, cam.full_proj_transform, cam.camera_center
        self.image_name = "view0"


def _case_inputs(case):
    opt = G.CASES[case]
    W, H = 152, 100
    cam = look_at_camera(W, H, (0.4, -0.3, -1.0), (0.0, 0.1, 4.0)) if "aa" in case else make_camera(W, H)
    sc = make_edge_scene(500, cam, seed=11) if "aa" in case else make_scene(500, cam, seed=12, s_med=0.05)
    g = torch.Generator().manual_seed(5)
    gt = torch.rand(3, H, W, generator=g)
    mono = torch.rand(1, H, W, generator=g) * 0.5
    dmask = (torch.rand(1, H, W, generator=g) > 0.3).float()
    col = torch.rand(sc.P, 3, generator=g)
    pipe = G._Pipe(convert_SHs_python=opt.get("convert_SHs_python", False), compute_cov3D_python=opt.get("compute_cov3D_python", False),
                   antialiasing=opt.get("antialiasing", False))
    return opt, cam, sc, gt, mono, dmask, col, pipe, torch.tensor([0.1, 0.0, 0.3])


def _finish(out, pc, gt, mono, dmask):
    image, inv = out["render"], out["depth"]
    loss = (image - gt).abs().mean() + 0.5 * torch.abs((inv - mono) * dmask).mean()      # train.py:112-142
    loss.backward()
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in pc.params().items()}
    grads["viewspace_points"] = out["viewspace_points"].grad.detach().clone()
    return image.detach(), inv.detach(), out["radii"], out["visibility_filter"], float(loss.detach()), grads


@pytest.mark.parametrize("case", list(G.CASES))
def test_the_restated_glue_is_the_reference_glue(case):
    opt, cam, sc, gt, mono, dmask, col, pipe, bg = _case_inputs(case)
    stub = types.ModuleType("diff_gaussian_rasterization")
    stub.GaussianRasterizationSettings, stub.GaussianRasterizer = O.Settings, G._OracleRasterizer
    ref = _import_reference_render(stub)
    kw = dict(scaling_modifier=opt.get("scaling_modifier", 1.0), separate_sh=opt.get("separate_sh", False),
              use_trained_exp=opt.get("use_trained_exp", False))
    res = []
    for which in ("reference", "restated"):
        pc = G._Model(sc, torch.device("cpu"), opt.get("active_sh_degree", 3))
        oc = col if opt.get("override_color") else None
        if which == "reference":
            with mock.patch.object(torch, "zeros_like", _cpu_zeros_like(torch.zeros_like)):
                out = ref.render(_Camera(cam), pc, pipe, bg, override_color=oc, **kw)
        else:
            out = G._render(cam, pc, pipe, bg, O.Settings, G._OracleRasterizer, override_color=oc, **kw)
        assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii", "depth"}
        res.append(_finish(out, pc, gt, mono, dmask))
    (ia, da, ra, va, la, ga), (ib, db, rb, vb, lb, gb) = res
    assert torch.equal(ra, rb) and torch.equal(va, vb) and int((ra > 0).sum()) > 50
    # the python-SH branch calls the reference's own eval_sh there and the oracle's pinned restatement here: same expression order
    tol = 1e-6 if pipe.convert_SHs_python else 0.0
    assert float((ia - ib).abs().max()) <= tol and float((da - db).abs().max()) <= tol and abs(la - lb) <= tol
    for k, a in ga.items():
        b = gb[k]
        assert (a is None) == (b is None), k
        if a is not None:
            assert float((a - b).abs().max()) <= tol * max(1.0, float(a.abs().max())), k


# -------------------------------------------------------------------------------------------------------------------------------
# (2) the reference's glue through THIS repo's package.  The native library needs a GPU; here a stand-in object with the same
# entry points (gsr_rasterize_forward / gsr_rasterize_backward / gsr_backward_scratch_bytes, argument lists of include/gsr.h) reads the
# host pointers the package hands over, lets the oracle compute, and writes the results where the package asked for them.
# -------------------------------------------------------------------------------------------------------------------------------
def _arr(ptr, shape, dtype=np.float32):
    if ptr is None:
        return None
    addr = ptr.value if isinstance(ptr, C.c_void_p) else int(ptr)
    if not addr:
        return None
    n = int(np.prod(shape))
    ct = {np.float32: C.c_float, np.int32: C.c_int32}[dtype]
    return np.ctypeslib.as_array((ct * n).from_address(addr)).reshape(shape)


def _t(ptr, shape):
    a = _arr(ptr, shape)
    return None if a is None else torch.from_numpy(a.copy())


class _OracleBehindTheCAbi:
    """Answers the three native calls of a single-GPU training step from the oracle (CPU tensors behind the pointers)."""

    def __init__(self):
        self.calls = []

    def _settings(self, sp):
        s = sp._obj
        H, W = s.image_height, s.image_width
        assert s.tile_y0 == 0 and s.tile_y1 == 0 and s.prefiltered == 0
        return O.Settings(H, W, s.tanfovx, s.tanfovy, _t(s.bg, (3,)), s.scale_modifier, _t(s.viewmatrix, (4, 4)), _t(s.projmatrix, (4, 4)),
                          s.sh_degree, _t(s.campos, (3,)), False, False, bool(s.antialiasing)), s

    def _inputs(self, s, P, M, means3D, shs, colors, opac, scales, rots, cov):
        dc = _t(s.sh_dc, (P, 1, 3)) if s.sh_dc else None
        sh = _t(shs, (P, M - 1 if dc is not None else M, 3)) if (shs is not None and M > 0) else None
        L = {"means3D": _t(means3D, (P, 3)), "opacities": _t(opac, (P, 1)), "dc": dc, "shs": sh, "colors_precomp": _t(colors, (P, 3)),
             "scales": _t(scales, (P, 3)), "rotations": _t(rots, (P, 4)), "cov3D_precomp": _t(cov, (P, 6))}
        return {k: v for k, v in L.items() if v is not None}

    def _run(self, L, st, grad=False):
        L = {k: v.clone().requires_grad_(grad) for k, v in L.items()}
        m2 = torch.zeros(L["means3D"].shape[0], 3, requires_grad=grad)
        shs = torch.cat((L["dc"], L["shs"]), dim=1) if "dc" in L else L.get("shs")
        col, radii, inv = O.rasterize(L["means3D"], m2, L["opacities"], st, shs=shs, colors_precomp=L.get("colors_precomp"), scales=L.get("scales"),
                                      rotations=L.get("rotations"), cov3D_precomp=L.get("cov3D_precomp"))
        return L, m2, col, radii, inv

    def gsr_rasterize_forward(self, sp, P, M, means3D, shs, colors, opac, scales, rots, cov, geom_cb, geom_u, bin_cb, bin_u, img_cb, img_u,
                              out_color, out_inv, radii, nr, stream):
        st, s = self._settings(sp)
        self.calls.append(("forward", P, M, bool(s.sh_dc), bool(s.no_backward)))
        for cb in (geom_cb, bin_cb, img_cb):      # the three caller-owned scratch buffers are requested like the library would
            assert cb(None, 256)
        with torch.no_grad():
            _, _, col, rad, inv = self._run(self._inputs(s, P, M, means3D, shs, colors, opac, scales, rots, cov), st)
        _arr(out_color, (3, st.image_height, st.image_width))[...] = col.numpy()
        _arr(out_inv, (1, st.image_height, st.image_width))[...] = inv.numpy()
        _arr(radii, (P,), np.int32)[...] = rad.numpy()
        nr._obj.value = 1234
        return 0

    def gsr_backward_scratch_bytes(self, P, R):
        assert R == 1234
        return 4096

    def gsr_rasterize_backward(self, sp, P, M, R, means3D, shs, colors, opac, scales, rots, cov, radii, geom, binning, img, g_color, g_depth,
                               d_m2, d_col, d_op, d_m3, d_cov, d_sh, d_sc, d_rot, scratch, rec, stream):
        st, s = self._settings(sp)
        self.calls.append(("backward", P, M, bool(s.sh_dc), g_depth is not None and bool(getattr(g_depth, "value", g_depth))))
        H, W = st.image_height, st.image_width
        with torch.enable_grad():      # (autograd.Function.backward runs with grad mode off)
            L, m2, col, rad, inv = self._run(self._inputs(s, P, M, means3D, shs, colors, opac, scales, rots, cov), st, grad=True)
            loss = (col * _t(g_color, (3, H, W))).sum()
            gd = _t(g_depth, (1, H, W))
            if gd is not None:
                loss = loss + (inv * gd).sum()
            if loss.requires_grad:
                loss.backward()
        z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad      # noqa: E731
        _arr(d_m2, (P, 3))[...] = z(m2).numpy()
        _arr(d_op, (P, 1))[...] = z(L["opacities"]).numpy()
        _arr(d_m3, (P, 3))[...] = z(L["means3D"]).numpy()
        for ptr, key, shape in ((d_col, "colors_precomp", (P, 3)), (d_cov, "cov3D_precomp", (P, 6)), (d_sc, "scales", (P, 3)), (d_rot, "rotations", (P, 4))):
            if key in L:
                _arr(ptr, shape)[...] = z(L[key]).numpy()
        if "shs" in L:
            _arr(d_sh, tuple(L["shs"].shape))[...] = z(L["shs"]).numpy()
        if "dc" in L:
            _arr(s.dL_dsh_dc, (P, 1, 3))[...] = z(L["dc"]).numpy()
        return 0


@pytest.mark.parametrize("case", list(G.CASES))
def test_reference_glue_through_this_package(case):
    import contextlib
    import diff_gaussian_rasterization as pkg
    from diff_gaussian_rasterization import _lib
    opt, cam, sc, gt, mono, dmask, col, pipe, bg = _case_inputs(case)
    ref = _import_reference_render(pkg)
    kw = dict(scaling_modifier=opt.get("scaling_modifier", 1.0), separate_sh=opt.get("separate_sh", False),
              use_trained_exp=opt.get("use_trained_exp", False))
    oc = col if opt.get("override_color") else None
    fake = _OracleBehindTheCAbi()
    pc = G._Model(sc, torch.device("cpu"), opt.get("active_sh_degree", 3))
    with mock.patch.object(torch, "zeros_like", _cpu_zeros_like(torch.zeros_like)), mock.patch.object(_lib, "load", lambda: fake), \
            mock.patch.object(pkg, "_require_cuda", lambda *a: None), mock.patch.object(pkg, "_stream_ptr", lambda d: None), \
            mock.patch.object(torch.cuda, "device", lambda d: contextlib.nullcontext()):
        out = ref.render(_Camera(cam), pc, pipe, bg, override_color=oc, **kw)
        got = _finish(out, pc, gt, mono, dmask)
    pc2 = G._Model(sc, torch.device("cpu"), opt.get("active_sh_degree", 3))
    want = _finish(G._render(cam, pc2, pipe, bg, O.Settings, G._OracleRasterizer, override_color=oc, **kw), pc2, gt, mono, dmask)
    # what the package made of the reference's call: one forward, one backward, the split-SH form natively when the glue passes dc= / shs=
    kinds = [c[0] for c in fake.calls]
    assert kinds == ["forward", "backward"], kinds
    split_expected = bool(opt.get("separate_sh", False)) and not pipe.convert_SHs_python and not opt.get("override_color")
    assert fake.calls[0][3] == split_expected, "dc= / shs= of the separate_sh form must reach the library as two tensors (no concatenation)"
    assert fake.calls[0][4] is False, "a call that will be differentiated must run the tracking build"
    assert fake.calls[1][4] is True, "the inverse-depth loss sends a gradient for the depth image"
    (ia, da, ra, va, la, ga), (ib, db, rb, vb, lb, gb) = got, want
    assert ra.dtype == torch.int32 and torch.equal(ra, rb.to(torch.int32)) and torch.equal(va, vb)
    tol = 2e-6
    assert float((ia - ib).abs().max()) <= tol and float((da - db).abs().max()) <= tol and abs(la - lb) <= tol
    for k, a in ga.items():
        b = gb[k]
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, k
            continue
        assert a is not None, f"{k}: the package returned no gradient"
        assert a.shape == b.shape, k
        assert float((a - b).abs().max()) <= 1e-5 * max(1e-30, float(b.abs().max())) + 1e-12, k
