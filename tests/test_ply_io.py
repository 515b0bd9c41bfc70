"""PLY container + Gaussian attribute layout: known-answer bytes against the PLY specification,
round trips, and -- in the build container, where /root/reference exists -- the reference's OWN save_ply / load_ply
(scene/gaussian_model.py:239-314) running on the in-tree `plyfile` stand-in."""
import io
import os
import struct
import sys
import types
from unittest import mock

import numpy as np
import pytest
import torch

import helpers  # noqa: F401
from plyfile import PlyData, PlyElement, PlyParseError
from gsr_scene import (gaussian_attribute_names, save_gaussians_ply, load_gaussians_ply, store_points_ply, fetch_points_ply)


def _model(P, deg, seed=0):
    g = torch.Generator().manual_seed(seed)
    K = (deg + 1) ** 2 - 1
    return dict(xyz=torch.randn(P, 3, generator=g), features_dc=torch.randn(P, 1, 3, generator=g),
                features_rest=torch.randn(P, K, 3, generator=g), opacity=torch.randn(P, 1, generator=g),
                scaling=torch.randn(P, 3, generator=g), rotation=torch.randn(P, 4, generator=g))


def test_known_answer_bytes(tmp_path):
    """Two Gaussians, degree 1: the file must be exactly the spec's header + packed little-endian floats."""
    m = _model(2, 1)
    path = str(tmp_path / "point_cloud.ply")
    save_gaussians_ply(path, **m)
    names = gaussian_attribute_names(9)
    assert names[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] and names[-8:] == \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex 2\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    body = b""
    for i in range(2):
        row = list(m["xyz"][i]) + [0.0, 0.0, 0.0] + list(m["features_dc"][i, 0])
        row += [float(m["features_rest"][i, k, c]) for c in range(3) for k in range(3)]      # channel-major
        row += list(m["opacity"][i]) + list(m["scaling"][i]) + list(m["rotation"][i])
        body += struct.pack("<%df" % len(row), *[float(v) for v in row])
    assert open(path, "rb").read() == header.encode("ascii") + body


@pytest.mark.parametrize("P,deg", [(0, 3), (1, 0), (257, 3), (1000, 2)])
def test_round_trip(tmp_path, P, deg):
    m = _model(P, deg, seed=P)
    path = str(tmp_path / "sub" / "pc.ply")
    save_gaussians_ply(path, **m)
    back = load_gaussians_ply(path, max_sh_degree=deg)
    for k in m:
        assert back[k].shape == m[k].shape and torch.equal(back[k], m[k]), k
    with pytest.raises(ValueError):
        load_gaussians_ply(path, max_sh_degree=deg + 1)


def test_container_ascii_big_endian_and_errors(tmp_path):
    el = np.zeros(3, dtype=[("x", "f4"), ("n", "i4"), ("c", "u1")])
    el["x"], el["n"], el["c"] = [0.5, -1.25, 3.0], [-7, 0, 9], [0, 128, 255]
    for kw in (dict(text=True), dict(byte_order=">"), dict()):
        buf = io.BytesIO()
        PlyData([PlyElement.describe(el, "vertex")], comments=["made by test"], **kw).write(buf)
        buf.seek(0)
        back = PlyData.read(buf)
        v = back["vertex"]
        assert [p.name for p in v.properties] == ["x", "n", "c"] and back.comments == ["made by test"]
        assert np.array_equal(v["x"], el["x"]) and np.array_equal(v["n"], el["n"]) and np.array_equal(v["c"], el["c"])
    with pytest.raises(PlyParseError):
        PlyData.read(io.BytesIO(b"plx\n"))
    with pytest.raises(PlyParseError):
        PlyData.read(io.BytesIO(b"ply\nformat ascii 1.0\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n3 0 1 2\n"))
    with pytest.raises(PlyParseError):      # truncated body
        PlyData.read(io.BytesIO(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nend_header\n\x00\x00"))
    # sized type names are accepted on read
    b = b"ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float32 x\nproperty uint8 r\nend_header\n" + struct.pack("<fB", 1.5, 7)
    v = PlyData.read(io.BytesIO(b)).elements[0]
    assert float(v["x"][0]) == 1.5 and int(v["r"][0]) == 7


def test_points3d_store_fetch(tmp_path):
    rng = np.random.default_rng(0)
    xyz = rng.normal(size=(500, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, size=(500, 3))
    path = str(tmp_path / "points3D.ply")
    store_points_ply(path, xyz, rgb)
    pos, col, nrm = fetch_points_ply(path)
    assert np.array_equal(pos, xyz) and np.allclose(col, rgb / 255.0) and not nrm.any()


REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "scene", "gaussian_model.py")), reason="reference tree not present")
def test_reference_gaussian_model_save_and_load_through_the_stand_in(tmp_path):
    """The reference's GaussianModel.save_ply / load_ply, unmodified, on the in-tree plyfile + simple_knn +
    diff_gaussian_rasterization packages; files interchange with gsr_scene in both directions."""
    sys.path.insert(0, REF)
    try:
        sys.modules.pop("scene", None)
        pkg = types.ModuleType("scene")          # avoid scene/__init__.py (needs the dataset readers' extra dependencies)
        pkg.__path__ = [os.path.join(REF, "scene")]
        with mock.patch.dict(sys.modules, {"scene": pkg}):
            import importlib
            gm = importlib.import_module("scene.gaussian_model")
            real_tensor = torch.tensor

            def cpu_tensor(*a, **k):
                k.pop("device", None)
                return real_tensor(*a, **k)

            m = _model(300, 3, seed=4)
            model = gm.GaussianModel(3)
            model._xyz, model._features_dc, model._features_rest = m["xyz"], m["features_dc"], m["features_rest"]
            model._opacity, model._scaling, model._rotation = m["opacity"], m["scaling"], m["rotation"]
            ref_path = str(tmp_path / "ref" / "point_cloud.ply")
            model.save_ply(ref_path)                                     # reference writer -> our reader
            back = load_gaussians_ply(ref_path, 3)
            for k in m:
                assert torch.equal(back[k], m[k]), k
            ours = str(tmp_path / "ours" / "point_cloud.ply")
            save_gaussians_ply(ours, **m)
            assert open(ours, "rb").read() == open(ref_path, "rb").read()   # byte-identical files
            model2 = gm.GaussianModel(3)
            with mock.patch.object(torch, "tensor", cpu_tensor):          # load_ply asks for device="cuda"
                model2.load_ply(ours)                                     # our writer -> reference reader
            assert torch.equal(model2._xyz.detach(), m["xyz"]) and torch.equal(model2._features_rest.detach(), m["features_rest"])
            assert torch.equal(model2._features_dc.detach(), m["features_dc"]) and torch.equal(model2._rotation.detach(), m["rotation"])
            assert torch.equal(model2._opacity.detach(), m["opacity"]) and torch.equal(model2._scaling.detach(), m["scaling"])
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k.startswith("scene.")]:
            sys.modules.pop(k, None)


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def activated_from_ply(path, device="cpu"):
    """What gaussian_renderer/__init__.py:56-110 hands to the rasterizer, from a point_cloud.ply, without the reference:
    gsr_scene.load_gaussians_ply + the activations of scene/gaussian_model.py:39-47 (exp, sigmoid, normalize)."""
    raw = load_gaussians_ply(path, 3, device=device)
    return dict(means3D=raw["xyz"], shs=torch.cat((raw["features_dc"], raw["features_rest"]), dim=1),
                opacities=torch.sigmoid(raw["opacity"]), scales=torch.exp(raw["scaling"]),
                rotations=torch.nn.functional.normalize(raw["rotation"]))


def test_file_loaded_by_the_reference_equals_file_loaded_by_gsr_scene():
    """tests/golden/scene_small.ply went through the REFERENCE's GaussianModel.load_ply and activations once, in the build
    container (tests/golden/make_golden_ply.py); the frozen tensors must equal what this repo's loader produces from the same
    file -- on any box, with no reference tree."""
    ref = np.load(os.path.join(GOLD, "scene_small_reference_loaded.npz"))
    ours = activated_from_ply(os.path.join(GOLD, "scene_small.ply"))
    for k, v in ours.items():
        assert v.shape == ref[k].shape, k
        assert np.array_equal(v.numpy(), ref[k]), f"{k}: max diff {np.abs(v.numpy() - ref[k]).max()}"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "scene", "gaussian_model.py")), reason="reference tree not present")
def test_the_ply_fixture_is_reproducible_from_the_reference(tmp_path):
    """The committed fixture is what the generating script produces today (build container only)."""
    import subprocess
    import shutil
    work = tmp_path / "golden"
    work.mkdir()
    shutil.copy(os.path.join(GOLD, "make_golden_ply.py"), work / "make_golden_ply.py")
    # the script writes next to itself; point its ROOT at the repo by running a patched copy
    src = (work / "make_golden_ply.py").read_text().replace('ROOT = os.path.dirname(os.path.dirname(HERE))', f'ROOT = {helpers.ROOT!r}')
    (work / "make_golden_ply.py").write_text(src)
    subprocess.check_call([sys.executable, str(work / "make_golden_ply.py")], stdout=subprocess.DEVNULL)
    assert (work / "scene_small.ply").read_bytes() == open(os.path.join(GOLD, "scene_small.ply"), "rb").read()
    new, old = np.load(work / "scene_small_reference_loaded.npz"), np.load(os.path.join(GOLD, "scene_small_reference_loaded.npz"))
    for k in old.files:
        assert np.array_equal(new[k], old[k]), k
