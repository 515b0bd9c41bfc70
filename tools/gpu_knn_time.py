"""Times simple_knn.distCUDA2 (csrc/knn.hip) on synthetic clouds; prints one line per size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402

for n in (100_000, 1_000_000):
    g = torch.Generator(device="cuda").manual_seed(0)
    pts = torch.randn(n, 3, device="cuda", generator=g) * torch.tensor([3.0, 1.0, 0.3], device="cuda")
    distCUDA2(pts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        d = distCUDA2(pts)
    torch.cuda.synchronize()
    print(f"distCUDA2 N={n}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms  mean={float(d.mean()):.3e}")
