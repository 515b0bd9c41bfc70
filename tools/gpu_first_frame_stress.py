"""First frames of concurrent callers (GPU; measurement / regression hunt, not a test): a fresh process per trial, T host threads that each
render their own scene on their own NON-BLOCKING stream for the first time at the same moment -- the moment the library creates one
control block per caller (gsr_api.cpp lease_host_word).  Every frame of every thread must equal the frame of the same scene rendered alone.

    python tools/gpu_first_frame_stress.py [--trials 20] [--threads 4] [--frames 8]          (GSR_LIB selects the library)

Prints one JSON line: trials, failed trials, the first messages."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gaussian-splatting_amd")]

CHILD = r"""
import sys, threading, torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings
T, FRAMES = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
cams = [make_camera(256 + 32 * i, 160 + 16 * i) for i in range(T)]
scenes = [make_scene(6000 + 9000 * i, cams[i], seed=40 + i, s_med=0.03).to(dev) for i in range(T)]
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
sets = [GaussianRasterizationSettings(c.image_height, c.image_width, c.tanfovx, c.tanfovy, bg, 1.0, c.world_view_transform.to(dev),
                                      c.full_proj_transform.to(dev), 3, c.camera_center.to(dev), False, False, False) for c in cams]
streams = [torch.cuda.Stream(device=dev) for _ in range(T)]
torch.cuda.synchronize()
def render(i):
    sc = scenes[i]
    with torch.no_grad():
        return GaussianRasterizer(sets[i])(means3D=sc.means3D, means2D=None, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
got = [[] for _ in range(T)]
errors = []
gate = threading.Barrier(T)
def worker(i):
    try:
        with torch.cuda.stream(streams[i]):
            gate.wait()
            for _ in range(FRAMES):
                got[i].append(tuple(t.clone() for t in render(i)))
            streams[i].synchronize()
    except Exception as ex:
        errors.append(f"thread {i}: {ex!r}")
th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
[t.start() for t in th]
[t.join() for t in th]
torch.cuda.synchronize()
# the frames alone, afterwards (one caller: one control block, long since created)
for i in range(T):
    try:
        want = render(i)
        torch.cuda.synchronize()
    except Exception as ex:
        errors.append(f"alone {i}: {ex!r}")
        continue
    for f, g in enumerate(got[i]):
        if not all(torch.equal(a, b) for a, b in zip(g, want)):
            errors.append(f"thread {i} frame {f} differs")
            break
# and twice more alone: a control block left dirty shows here
for i in range(T):
    try:
        a, b = render(i), render(i)
        torch.cuda.synchronize()
        if not all(torch.equal(x, y) for x, y in zip(a, b)):
            errors.append(f"alone {i}: two frames of the same scene differ")
    except Exception as ex:
        errors.append(f"alone-again {i}: {ex!r}")
print("ERRORS " + repr(errors) if errors else "CLEAN")
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=20)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--frames", type=int, default=8)
    a = ap.parse_args()
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(sys.path[:1] + [os.environ.get("PYTHONPATH", "")]))
    failed, msgs = 0, []
    for t in range(a.trials):
        r = subprocess.run([sys.executable, "-c", CHILD, str(a.threads), str(a.frames)], env=env, capture_output=True, text=True, timeout=300)
        out = r.stdout.strip().splitlines()
        ok = r.returncode == 0 and out and out[-1] == "CLEAN"
        if not ok:
            failed += 1
            if len(msgs) < 4:
                msgs.append((out[-1] if out else "") + (" | " + r.stderr.strip().splitlines()[-1] if r.returncode else ""))
    print(json.dumps({"library": os.environ.get("GSR_LIB", "lib"), "trials": a.trials, "threads": a.threads, "frames": a.frames, "failed_trials": failed,
                      "first_messages": msgs}))


if __name__ == "__main__":
    main()
