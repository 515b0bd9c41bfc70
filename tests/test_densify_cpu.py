"""Adaptive density control: gsr_scene.densify against its specification and -- in the build
container, where /root/reference exists -- against the reference's OWN GaussianModel.densify_and_prune / reset_opacity
(scene/gaussian_model.py:258-261, 316-469) executed on CPU tensors with the same random seed."""
import os
import sys
import types
from unittest import mock

import pytest
import torch
import torch.nn as nn

import helpers  # noqa: F401
from gsr_scene.densify import DensifyStats, densify_and_prune, reset_opacity, quaternion_to_rotation, GROUPS

REF = "/root/reference"


def _scene(P, seed):
    g = torch.Generator().manual_seed(seed)
    return {"xyz": torch.randn(P, 3, generator=g), "f_dc": torch.randn(P, 1, 3, generator=g), "f_rest": torch.randn(P, 15, 3, generator=g),
            "opacity": torch.randn(P, 1, generator=g) * 2.0, "scaling": torch.randn(P, 3, generator=g) * 0.8 - 3.6,
            "rotation": torch.randn(P, 4, generator=g)}


def _optimizer(t):
    params = {k: nn.Parameter(v.clone().requires_grad_(True)) for k, v in t.items()}
    opt = torch.optim.Adam([{"params": [params[k]], "lr": 1e-3, "name": k} for k in GROUPS], lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(99)
    for k in GROUPS:                                  # one real step so that the moments are populated
        params[k].grad = torch.randn(params[k].shape, generator=g)
    opt.step()
    opt.zero_grad(set_to_none=True)
    return params, opt


def _stats(P, seed):
    g = torch.Generator().manual_seed(seed)
    st = DensifyStats.zeros(P, "cpu")
    st.denom[:] = torch.randint(0, 5, (P, 1), generator=g).float()           # zeros -> NaN grads -> treated as 0
    st.xyz_gradient_accum[:] = torch.rand(P, 1, generator=g) * 0.002 * st.denom.clamp_min(1)
    st.max_radii2D[:] = torch.rand(P, generator=g) * 50
    return st


def test_single_repack_semantics():
    P = 400
    t = _scene(P, 1)
    params, opt = _optimizer(t)
    before = {k: (params[k].detach().clone(), opt.state[params[k]]["exp_avg"].clone()) for k in GROUPS}
    st = _stats(P, 2)
    grads = st.xyz_gradient_accum / st.denom
    grads[grads.isnan()] = 0
    smax = torch.exp(before["scaling"][0]).max(1).values
    extent, thr = 4.0, 0.0006
    clone = (grads.squeeze() >= thr) & (smax <= 0.01 * extent)
    split = (grads.squeeze() >= thr) & (smax > 0.01 * extent)
    low = torch.sigmoid(before["opacity"][0]).squeeze() < 0.005
    assert clone.any() and split.any() and low.any()
    torch.manual_seed(5)
    new, nst, _ = densify_and_prune(opt, st, thr, 0.005, extent, max_screen_size=20, radii=torch.arange(P))
    keep = ~split & ~low & ~(smax > 0.1 * extent)
    n_keep = int(keep.sum())
    n_clone = int((clone & ~low & ~(smax > 0.1 * extent)).sum())
    assert new["xyz"].shape[0] >= n_keep + n_clone and nst.denom.shape[0] == new["xyz"].shape[0]
    for k in GROUPS:
        assert new[k].shape[0] == new["xyz"].shape[0] and new[k].requires_grad and opt.param_groups[GROUPS.index(k)]["params"][0] is new[k]
        assert torch.equal(new[k].detach()[:n_keep], before[k][0][keep])                    # survivors: value and moments kept
        assert torch.equal(opt.state[new[k]]["exp_avg"][:n_keep], before[k][1][keep])
        assert not opt.state[new[k]]["exp_avg"][n_keep:].any() and not opt.state[new[k]]["exp_avg_sq"][n_keep:].any()
    assert torch.equal(new["xyz"].detach()[n_keep:n_keep + n_clone], before["xyz"][0][clone & ~low & ~(smax > 0.1 * extent)])
    kids = new["scaling"].detach()[n_keep + n_clone:]
    assert kids.shape[0] > 0 and kids.shape[0] % 1 == 0
    # children: scale = parent scale / 1.6, position within a few sigma of the parent
    par_scale = torch.exp(before["scaling"][0])[split & ~low]
    assert torch.allclose(torch.exp(kids)[: par_scale.shape[0]] * 1.6, par_scale[: kids.shape[0]], rtol=1e-5) or kids.shape[0] < par_scale.shape[0]
    R = quaternion_to_rotation(torch.tensor([[2.0, 0.0, 0.0, 0.0], [0.7071068, 0.7071068, 0.0, 0.0]]))
    assert torch.allclose(R[0], torch.eye(3)) and torch.allclose(R[1], torch.tensor([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]]), atol=1e-6)
    p = reset_opacity(opt, 0.01)
    assert float(torch.sigmoid(p).max()) <= 0.01 + 1e-6 and not opt.state[p]["exp_avg"].any()


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "scene", "gaussian_model.py")), reason="reference tree not present")
@pytest.mark.parametrize("max_screen_size", [None, 20])
def test_matches_the_reference_gaussian_model(max_screen_size):
    sys.path.insert(0, REF)
    try:
        sys.modules.pop("scene", None)
        pkg = types.ModuleType("scene")
        pkg.__path__ = [os.path.join(REF, "scene")]
        with mock.patch.dict(sys.modules, {"scene": pkg}):
            import importlib
            gm = importlib.import_module("scene.gaussian_model")
            real_zeros = torch.zeros

            def cpu_zeros(*a, **k):
                k.pop("device", None)
                return real_zeros(*a, **k)

            P, extent, thr, min_op = 700, 4.0, 0.0006, 0.005
            t = _scene(P, 11)
            # reference model
            m = gm.GaussianModel(3)
            rp, ropt = _optimizer(t)
            m._xyz, m._features_dc, m._features_rest = rp["xyz"], rp["f_dc"], rp["f_rest"]
            m._opacity, m._scaling, m._rotation = rp["opacity"], rp["scaling"], rp["rotation"]
            m.optimizer, m.percent_dense = ropt, 0.01
            st = _stats(P, 12)
            m.xyz_gradient_accum, m.denom, m.max_radii2D = st.xyz_gradient_accum.clone(), st.denom.clone(), st.max_radii2D.clone()
            radii = torch.arange(P, dtype=torch.float32)
            with mock.patch.object(torch, "zeros", cpu_zeros):
                torch.manual_seed(123)
                m.densify_and_prune(thr, min_op, extent, max_screen_size, radii.clone())
                m.reset_opacity()
            # this repo
            mp, mopt = _optimizer(t)
            torch.manual_seed(123)
            new, nst, tmp = densify_and_prune(mopt, _stats(P, 12), thr, min_op, extent, max_screen_size, radii=radii.clone())
            new["opacity"] = reset_opacity(mopt, 0.01)
            ref = {"xyz": m._xyz, "f_dc": m._features_dc, "f_rest": m._features_rest, "opacity": m._opacity, "scaling": m._scaling,
                   "rotation": m._rotation}
            assert ref["xyz"].shape[0] != P                               # the step really changed the set
            for k in GROUPS:
                assert new[k].shape == ref[k].shape, k
                assert torch.equal(new[k].detach(), ref[k].detach()), k
                for s in ("exp_avg", "exp_avg_sq"):
                    assert torch.equal(mopt.state[new[k]][s], ropt.state[ref[k]][s]), (k, s)
            assert nst.denom.shape == m.denom.shape and not nst.xyz_gradient_accum.any() and not m.xyz_gradient_accum.any()
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k.startswith("scene.")]:
            sys.modules.pop(k, None)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "scene", "gaussian_model.py")), reason="reference tree not present")
@pytest.mark.parametrize("max_screen_size", [None, 20])
def test_attach_drives_the_reference_gaussian_model_through_its_own_method_names(max_screen_size):
    """gsr_scene.densify.attach(gaussians): the reference's OWN GaussianModel instance, once with its own methods and once with the three methods the
    adapter binds, through the calls train.py:164-174 makes (add_densification_stats, densify_and_prune, reset_opacity) -- every attribute the
    caller reads afterwards and the optimizer's state are the same bits."""
    from gsr_scene.densify import attach
    sys.path.insert(0, REF)
    try:
        sys.modules.pop("scene", None)
        pkg = types.ModuleType("scene")
        pkg.__path__ = [os.path.join(REF, "scene")]
        with mock.patch.dict(sys.modules, {"scene": pkg}):
            import importlib
            gm = importlib.import_module("scene.gaussian_model")
            real_zeros = torch.zeros

            def cpu_zeros(*a, **k):
                k.pop("device", None)
                return real_zeros(*a, **k)

            P, extent, thr, min_op = 600, 4.0, 0.0006, 0.005
            t = _scene(P, 21)

            def model():
                m = gm.GaussianModel(3)
                rp, ropt = _optimizer(t)
                m._xyz, m._features_dc, m._features_rest = rp["xyz"], rp["f_dc"], rp["f_rest"]
                m._opacity, m._scaling, m._rotation = rp["opacity"], rp["scaling"], rp["rotation"]
                m.optimizer, m.percent_dense = ropt, 0.01
                st = _stats(P, 22)
                m.xyz_gradient_accum, m.denom, m.max_radii2D = st.xyz_gradient_accum.clone(), st.denom.clone(), st.max_radii2D.clone()
                return m
            g = torch.Generator().manual_seed(5)
            view = types.SimpleNamespace(grad=torch.randn(P, 3, generator=g) * 1e-3)
            seen = torch.rand(P, generator=g) < 0.7
            radii = torch.arange(P, dtype=torch.float32)
            a, b = model(), attach(model())
            assert b.densify_and_prune.__func__ is not gm.GaussianModel.densify_and_prune
            with mock.patch.object(torch, "zeros", cpu_zeros):
                for m in (a, b):
                    m.add_densification_stats(view, seen)
                assert torch.equal(a.xyz_gradient_accum, b.xyz_gradient_accum) and torch.equal(a.denom, b.denom)
                for m in (a, b):
                    torch.manual_seed(77)
                    m.densify_and_prune(thr, min_op, extent, max_screen_size, radii.clone())
                    m.reset_opacity()
            assert a._xyz.shape[0] != P and a.tmp_radii is None and b.tmp_radii is None
            for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
                pa, pb = getattr(a, name), getattr(b, name)
                assert isinstance(pb, nn.Parameter) and pb.requires_grad and torch.equal(pa.detach(), pb.detach()), name
                for s in ("exp_avg", "exp_avg_sq"):
                    assert torch.equal(a.optimizer.state[pa][s], b.optimizer.state[pb][s]), (name, s)
            for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
                assert torch.equal(getattr(a, name), getattr(b, name)), name
            assert torch.equal(a.get_opacity, b.get_opacity) and torch.equal(a.get_scaling, b.get_scaling)      # the model's own accessors
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k.startswith("scene.")]:
            sys.modules.pop(k, None)


def test_attach_on_a_duck_typed_model():
    """The same adapter on an object that only has the attributes (the GPU box has no reference tree): the bound methods leave the attributes the caller
    reads in place and consistent with the optimizer."""
    from gsr_scene.densify import attach
    P = 300
    t = _scene(P, 31)
    rp, ropt = _optimizer(t)
    st = _stats(P, 32)
    m = types.SimpleNamespace(_xyz=rp["xyz"], _features_dc=rp["f_dc"], _features_rest=rp["f_rest"], _opacity=rp["opacity"], _scaling=rp["scaling"],
                              _rotation=rp["rotation"], optimizer=ropt, percent_dense=0.01, xyz_gradient_accum=st.xyz_gradient_accum, denom=st.denom,
                              max_radii2D=st.max_radii2D, tmp_radii=None, scaling_activation=torch.exp, scaling_inverse_activation=torch.log,
                              opacity_activation=torch.sigmoid, inverse_opacity_activation=lambda p: torch.log(p / (1 - p)))
    attach(m)
    view = types.SimpleNamespace(grad=torch.full((P, 3), 1e-3))
    before = m.denom.clone()
    m.add_densification_stats(view, torch.ones(P, dtype=torch.bool))
    assert torch.equal(m.denom, before + 1)
    torch.manual_seed(3)
    m.densify_and_prune(0.0006, 0.005, 4.0, 20, torch.arange(P, dtype=torch.float32))
    n = m._xyz.shape[0]
    assert n != P and all(getattr(m, k).shape[0] == n for k in ("_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "xyz_gradient_accum", "denom", "max_radii2D"))
    assert all(g["params"][0] is getattr(m, a) for g, a in zip(m.optimizer.param_groups, ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")))
    m.reset_opacity()
    assert float(torch.sigmoid(m._opacity).max()) <= 0.01 + 1e-6 and m.optimizer.param_groups[3]["params"][0] is m._opacity
