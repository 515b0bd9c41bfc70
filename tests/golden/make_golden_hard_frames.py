"""Generates tests/golden/hard_frames.npz: the frames the randomised whole-operator cross-check (tools/gpu_fuzz_render.py) reported as
hard in round 5, frozen as fixtures so that `pytest -m gpu` (and the kernel source run on the CPU) holds them, not a tool.

    python tests/golden/make_golden_hard_frames.py

 (seed 71, frame 109)  20 needles, 31x257, depth loss, scale modifier 1.7 -- round 5's one recorded parity failure: rotation gradient
                       2.4e-3 of max |grad| from the fp64 oracle (bar 2e-3) -- the fp32 covariance chain of the per-Gaussian backward
 (seed 53, frame 92)   130 needles, 480x270     } frames where fp32 stops resolving the blend: the fp32 ORACLE is 2.4e-3 .. 1.3e-2 of
 (seed 53, frame 137)  1524 needles, 31x257     } max |grad| from its own fp64 evaluation
 (seed 71, frame 32)   48 needles, cov3D_precomp call form (adjudicated by fp64 in round 5)
 (seed 71, frame 78)   1494 extreme needles (condition numbers >= 1e5), 480x270: finite gradients; a sign flip of the exponent
 (seed 101, frame 172) 1670 needles: round 6's worst kernel-vs-fp64 distance on the final binaries (rotations 6.1e-3 where the fp32 oracle is 3.6e-2)
 (seed 202, frame 263) 134 needles: scales 2.1e-3 from fp64 where the fp32 oracle is 3.6e-3 (the closest the two came in 1 100 frames)
 (seed 404, frame 72)  55 Gaussians at the frustum's edge, antialiased, depth loss: ONE pixel whose alpha sits on 1/255 takes the other branch in the kernel
                       (image error there 3.5e-4 = one alpha quantum) and moves the rotation gradient by 4.1e-3 of max |grad|; without that pixel 4.8e-6
 (seed 303, frame 189) 9 needles, five pixels with a threshold inside the exponent's cancellation noise: rotations 1.6e-2 from fp64 with them, 4.2e-5 without
                       (the fp32 oracle: 1.3e-2).  Both frames are stored with the loss weights zeroed at those pixels: what is pinned is "the gradients
                       agree wherever no threshold is within noise".

Per frame the file holds the INPUTS of the operator call (settings record, every tensor of the call form, the loss weights) and the
EXPECTED outputs: radii, and the gradient of the frame's loss with respect to every input from the oracle's autograd evaluated in
fp64 on the same fp32 inputs (`<frame>/grad64/<name>`), with the distance of the fp32 oracle's gradients from them
(`<frame>/o32_err/<name>`, in units of max |grad64|) -- the measure of what fp32 can resolve on that frame.

The frames are rebuilt by tests/fuzz_frames.py (the fuzzer's own generator, same draws).  Test infrastructure."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fuzz_frames as F  # noqa: E402

FRAMES = [(71, 109), (53, 92), (53, 137), (71, 32), (71, 78), (101, 172), (202, 263), (404, 72), (303, 189)]
# frames whose ONLY failure was a pixel with a hard blend threshold inside fp32's rounding noise (either branch is a correct fp32 result: the pixel's
# branch feeds the gradients too): their loss weights are stored with zeros at those pixels (tests/fuzz_frames.py threshold_pixels)
WITHOUT_THRESHOLD_PIXELS = {(404, 72), (303, 189)}
SETTINGS_FIELDS = ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
                   "campos", "prefiltered", "debug", "antialiasing")


def main():
    out = {"frames": np.array([f"{s}_{i}" for s, i in FRAMES])}
    for seed, it in FRAMES:
        f = F.frame(seed, it)
        key = f"{seed}_{it}"
        if (seed, it) in WITHOUT_THRESHOLD_PIXELS:
            near = F.threshold_pixels(f)
            keep = (~near).to(f.wc.dtype)
            f.wc = f.wc * keep
            if f.use_depth:
                f.wd = f.wd * keep
            out[f"{key}/threshold_pixels"] = torch.nonzero(near).numpy().astype(np.int32)
            print(f"{key}: {int(near.sum())} pixel(s) with a threshold within noise, loss weights zeroed there", flush=True)
        for name in SETTINGS_FIELDS:
            v = getattr(f.s, name)
            out[f"{key}/settings/{name}"] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
        out[f"{key}/kind"] = np.asarray(f.kind)
        out[f"{key}/form"] = np.asarray(f.form)
        out[f"{key}/use_depth"] = np.asarray(f.use_depth)
        L = f.leaves("cpu")
        for k, v in L.items():
            if k != "means2D":
                out[f"{key}/in/{k}"] = v.detach().numpy()
        out[f"{key}/wc"] = f.wc.numpy()
        if f.use_depth:
            out[f"{key}/wd"] = f.wd.numpy()
        L32, _, r32, _ = f.oracle_backward(torch.float32)
        out[f"{key}/radii"] = r32.numpy().astype(np.int32)
        L64, _, r64, _ = f.oracle_backward(torch.float64)
        if not torch.equal(r32, r64):
            # a radius is decided differently in fp64: not the same frame (the extreme needles) -- inputs and fp32 radii only; the test holds
            # the operator to the fuzzer's contract for this kind: radii bit-exact, every gradient finite
            assert f.kind == "extreme_needles"
            print(f"{key} {f.kind}: {int((r32 != r64).sum())} radii differ in fp64 -- no fp64 gradients stored", flush=True)
            continue
        for k in L64:
            g64 = L64[k].grad
            if g64 is None:
                continue
            out[f"{key}/grad64/{k}"] = g64.numpy()
            scale = g64.abs().max().item()
            g32 = L32[k].grad
            e = 0.0 if scale == 0.0 else ((g32.double() - g64).abs().max().item() / scale)
            out[f"{key}/o32_err/{k}"] = np.asarray(e)
            print(f"{key} {f.kind:16s} P={f.P:5d} {f.W}x{f.H} {f.form:12s} {k:14s} max|grad64| {scale:.4e}   fp32 oracle vs fp64 {e:.3e}", flush=True)
    path = os.path.join(HERE, "hard_frames.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
