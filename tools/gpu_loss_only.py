"""The fused training loss alone (fused_ssim.fused_train_loss forward + backward, 3 x 1080 x 1920), N times: the workload of
tools/gpu_loss_pmc.sh (counter passes on the two SSIM kernels without the rest of the bench around them)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import torch
from diff_gaussian_rasterization import _lib
from fused_ssim import fused_train_loss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for o in sys.argv[2:]:
    k, v = o.split("=")
    _lib.set_option(k, int(v))
dev = torch.device("cuda:0")
H, W = (int(os.environ.get("LOSS_H", 1080)), int(os.environ.get("LOSS_W", 1920)))
g = torch.Generator(device=dev).manual_seed(0)
gt = torch.rand(3, H, W, device=dev, generator=g)
img = (gt + 0.1 * torch.randn(3, H, W, device=dev, generator=g)).clamp(0, 1).requires_grad_(True)
for _ in range(3):
    fused_train_loss(img, gt).backward()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    img.grad = None
    fused_train_loss(img, gt).backward()
e1.record()
torch.cuda.synchronize()
print(f"loss forward + backward: {e0.elapsed_time(e1) / n * 1e3:.1f} us per iteration ({n} iterations, {H}x{W}, options {sys.argv[2:]})")
