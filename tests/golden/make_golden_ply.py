"""Generates tests/golden/scene_small.ply and tests/golden/scene_small_reference_loaded.npz.  Run in the BUILD container
(needs /root/reference, read-only):

    python tests/golden/make_golden_ply.py

The closest thing to the reference's render.py:48-60 this environment allows: a trained-model file in
the reference's on-disk layout (scene/gaussian_model.py:239-256), loaded by the REFERENCE's own GaussianModel.load_ply
(scene/gaussian_model.py:263-314) and passed through the REFERENCE's own activations (get_xyz / get_features / get_opacity /
get_scaling / get_rotation, scene/gaussian_model.py:39-47,102-130) -- exactly what gaussian_renderer/__init__.py:56-110 hands
to the rasterizer.  The result is frozen as a fixture; the GPU test (tests/test_gpu_reference_glue.py) loads the same file
through gsr_scene.load_gaussians_ply on the GPU box, where /root/reference does not exist, and must arrive at the same
tensors and the same image.
"""
import importlib
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
REF = "/root/reference"

P, W, H, SEED = 700, 320, 200, 17


def raw_model():
    """A synthetic scene of the SURVEY 8(d) recipe, as RAW parameters (log-scales, logit-opacities, un-normalised quaternions)."""
    from gsr_synth import make_camera, make_scene
    cam = make_camera(W, H)
    sc = make_scene(P, cam, seed=SEED, s_med=0.03)
    g = torch.Generator().manual_seed(SEED)
    op = sc.opacities.clamp(1e-4, 1 - 1e-4)
    return dict(xyz=sc.means3D.clone(), features_dc=sc.shs[:, :1].contiguous(), features_rest=sc.shs[:, 1:].contiguous(),
                opacity=torch.log(op / (1 - op)), scaling=torch.log(sc.scales),
                rotation=sc.rotations * (0.5 + torch.rand(P, 1, generator=g)))          # un-normalised on disk, like a trained model


def main():
    from gsr_scene import save_gaussians_ply
    ply = os.path.join(HERE, "scene_small.ply")
    save_gaussians_ply(ply, **raw_model())
    sys.path.insert(0, REF)
    pkg = types.ModuleType("scene")          # avoid scene/__init__.py (needs the dataset readers' extra dependencies)
    pkg.__path__ = [os.path.join(REF, "scene")]
    with mock.patch.dict(sys.modules, {"scene": pkg}):
        gm = importlib.import_module("scene.gaussian_model")
        real_tensor = torch.tensor

        def cpu_tensor(*a, **k):
            k.pop("device", None)
            return real_tensor(*a, **k)

        model = gm.GaussianModel(3)
        with mock.patch.object(torch, "tensor", cpu_tensor):          # load_ply asks for device="cuda"
            model.load_ply(ply)
        with torch.no_grad():
            out = dict(means3D=model.get_xyz, shs=model.get_features, opacities=model.get_opacity, scales=model.get_scaling,
                       rotations=model.get_rotation)
            out = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, "scene_small_reference_loaded.npz"), width=W, height=H, **out)
    print({k: v.shape for k, v in out.items()}, os.path.getsize(ply), "bytes of ply")


if __name__ == "__main__":
    main()
