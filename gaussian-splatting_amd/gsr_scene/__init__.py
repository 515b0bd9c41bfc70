"""Scene-side data formats next to the operator: the trained-model file `point_cloud.ply`
(scene/gaussian_model.py:225-256 writer, :263-314 reader) and the SfM seed cloud `points3D.ply`
(scene/dataset_readers.py fetchPly / storePly), read and written without the third-party `plyfile` package
(the in-tree `plyfile.py` stand-in parses the container; this module knows the Gaussian attribute layout).

Attribute layout of point_cloud.ply -- one `vertex` element, every property `float`:
    x y z | nx ny nz (zeros) | f_dc_0..2 | f_rest_0..(3K-1) | opacity | scale_0..2 | rot_0..3
with K = (max_sh_degree+1)^2 - 1.  SH features are stored CHANNEL-MAJOR: the writer flattens
features.transpose(1, 2), i.e. f_rest_{c*K + k} = features_rest[:, k, c].  All values are the raw (pre-activation)
parameters: log-scales, logit-opacities, un-normalised quaternions."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch

from plyfile import PlyData, PlyElement

__all__ = ["gaussian_attribute_names", "save_gaussians_ply", "load_gaussians_ply", "store_points_ply", "fetch_points_ply"]
# adaptive density control (clone / split / prune as one repack): gsr_scene.densify


def gaussian_attribute_names(n_rest: int, n_scale: int = 3, n_rot: int = 4):
    """construct_list_of_attributes (scene/gaussian_model.py:225-237)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def _np(t) -> np.ndarray:
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save_gaussians_ply(path: str, xyz, features_dc, features_rest, opacity, scaling, rotation) -> None:
    """xyz[P,3], features_dc[P,1,3], features_rest[P,K,3], opacity[P,1], scaling[P,3], rotation[P,4] (raw parameters)."""
    xyz = _np(xyz).astype(np.float32)
    P = xyz.shape[0]
    fd, fr = _np(features_dc).astype(np.float32), _np(features_rest).astype(np.float32)
    f_dc = np.ascontiguousarray(fd.transpose(0, 2, 1)).reshape(P, fd.shape[1] * fd.shape[2])
    f_rest = np.ascontiguousarray(fr.transpose(0, 2, 1)).reshape(P, fr.shape[1] * fr.shape[2])
    cols = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, _np(opacity).astype(np.float32).reshape(P, 1),
                           _np(scaling).astype(np.float32).reshape(P, 3), _np(rotation).astype(np.float32).reshape(P, 4)], axis=1)
    names = gaussian_attribute_names(f_rest.shape[1], 3, 4)
    assert cols.shape[1] == len(names)
    # a structured array of equal-width f4 fields is the same bytes as the [P, n] float32 matrix
    elements = np.ascontiguousarray(cols).view(np.dtype([(n, "<f4") for n in names])).reshape(P)
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    PlyData([PlyElement.describe(elements, "vertex")]).write(path)


def load_gaussians_ply(path: str, max_sh_degree: int = 3, device="cpu") -> Dict[str, torch.Tensor]:
    """Inverse of save_gaussians_ply / the reference's load_ply: tensors in the parameter layout of GaussianModel."""
    v = PlyData.read(path).elements[0]
    P = v.count

    def col(name):
        return np.asarray(v[name], dtype=np.float32)

    def numbered(prefix):
        names = [p.name for p in v.properties if p.name.startswith(prefix)]
        return sorted(names, key=lambda n: int(n.split("_")[-1]))

    xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
    f_dc = np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], axis=1).reshape(P, 3, 1)
    rest_names = numbered("f_rest_")
    K = (max_sh_degree + 1) ** 2 - 1
    if len(rest_names) != 3 * K:
        raise ValueError(f"{path}: {len(rest_names)} f_rest_* properties, max_sh_degree={max_sh_degree} needs {3 * K}")
    f_rest = (np.stack([col(n) for n in rest_names], axis=1) if rest_names else np.zeros((P, 0), np.float32)).reshape(P, 3, K)
    scales = np.stack([col(n) for n in numbered("scale_")], axis=1)
    rots = np.stack([col(n) for n in numbered("rot")], axis=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    return {"xyz": t(xyz), "features_dc": t(f_dc.transpose(0, 2, 1)), "features_rest": t(f_rest.transpose(0, 2, 1)),
            "opacity": t(col("opacity").reshape(P, 1)), "scaling": t(scales), "rotation": t(rots)}


def store_points_ply(path: str, xyz, rgb) -> None:
    """storePly (scene/dataset_readers.py): x y z nx ny nz as float, red green blue as uchar (0..255)."""
    xyz = _np(xyz).astype(np.float32)
    rgb = _np(rgb)
    el = np.empty(xyz.shape[0], dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"),
                                       ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    el["x"], el["y"], el["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    el["nx"] = el["ny"] = el["nz"] = 0.0
    el["red"], el["green"], el["blue"] = rgb[:, 0].astype(np.uint8), rgb[:, 1].astype(np.uint8), rgb[:, 2].astype(np.uint8)
    PlyData([PlyElement.describe(el, "vertex")]).write(path)


def fetch_points_ply(path: str):
    """fetchPly: (positions[N,3] float32, colors[N,3] in 0..1, normals[N,3])."""
    v = PlyData.read(path)["vertex"]
    pos = np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32)
    col = np.stack([v["red"], v["green"], v["blue"]], axis=1).astype(np.float32) / 255.0
    names = {p.name for p in v.properties}
    nrm = np.stack([v["nx"], v["ny"], v["nz"]], axis=1).astype(np.float32) if {"nx", "ny", "nz"} <= names else np.zeros_like(pos)
    return pos, col, nrm
