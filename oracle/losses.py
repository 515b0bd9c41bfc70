"""The training loss of the reference's train step, restated with plain torch ops for bench.py:
    loss = (1 - lambda_dssim) * L1(image, gt) + lambda_dssim * (1 - SSIM(image, gt))       (train.py:119-126)
with L1 = mean |a - b| (utils/loss_utils.py:40-41), SSIM = 11x11 Gaussian window (sigma 1.5), grouped conv2d with
padding 5, C1 = 0.01^2, C2 = 0.03^2, mean over the map (utils/loss_utils.py:43-87), lambda_dssim = 0.2
(arguments/__init__.py:88).

TEST INFRASTRUCTURE (moved here from the product package in round 3): the checker of the fused HIP loss
kernels (fused_ssim.fused_ssim / fused_train_loss).  Pinned to the reference's own utils/loss_utils.py by the golden vector of
tests/golden/reference_fragments.npz (tests/test_oracle.py::test_train_loss_matches_reference_loss_utils).  Imported only
by tests/ and by bench.py's optional --compare-torch-adam comparison leg; the product never routes through it."""
import math

import torch
import torch.nn.functional as F

_window_cache = {}


def _window(channels: int, device, dtype):
    key = (channels, str(device), dtype)
    if key not in _window_cache:
        g = torch.tensor([math.exp(-((x - 5) ** 2) / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float64)
        g = (g / g.sum()).to(dtype)
        w2 = (g[:, None] @ g[None, :])[None, None].expand(channels, 1, 11, 11).contiguous().to(device)
        _window_cache[key] = w2
    return _window_cache[key]


def ssim(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """img: [3,H,W] or [B,3,H,W]."""
    if img1.dim() == 3:
        img1, img2 = img1[None], img2[None]
    c = img1.shape[1]
    w = _window(c, img1.device, img1.dtype)
    mu1 = F.conv2d(img1, w, padding=5, groups=c)
    mu2 = F.conv2d(img2, w, padding=5, groups=c)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=5, groups=c) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=5, groups=c) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=5, groups=c) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean()


def l1_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return (a - b).abs().mean()


def train_loss(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2) -> torch.Tensor:
    return (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))
