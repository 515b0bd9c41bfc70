"""The fused training loss of csrc/ssim.hip -- (1 - lambda) L1 + lambda (1 - SSIM) computed inside the marching-wave SSIM kernels, forward and
backward -- and the mean-SSIM form, executed on the CPU from the source through the SIMT shim (DPP wave shifts, raw buffer loads / stores with
the hardware's out-of-range behaviour) against the fp64 formula of the reference's utils/loss_utils.py:40-87 / train.py:119-126 (restated in
oracle/losses.py, pinned to the reference by a golden vector): the shapes, mixes and bars of the GPU tests
(tests/test_gpu_parity.py::test_fused_train_loss_matches_reference_formula, ::test_fused_ssim_matches_reference_formula).
Test infrastructure: tests/_build/libsimt_loss.so is never part of the product."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libsimt_loss.so")
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


@pytest.fixture(scope="module")
def lib():
    from simt_build import build
    h = build("loss", fp_contract_off=True)
    h.simt_loss_last_error.restype = C.c_char_p
    return h


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _images(shape):
    g = torch.Generator().manual_seed(shape[-1])
    a = torch.rand(shape, generator=g)
    b = (a + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1)
    b[..., :3, :5] = a[..., :3, :5]                    # exact ties: zero L1 gradient there
    return a, b


@pytest.mark.parametrize("shape,lam", [((3, 67, 93), 0.2), ((3, 128, 160), 0.2), ((3, 16, 16), 0.5), ((3, 11, 300), 0.0), ((3, 40, 40), 1.0), ((2, 150, 55), 0.2)])
def test_fused_train_loss_source_on_the_cpu(lib, shape, lam):
    from oracle.losses import train_loss
    a, b = _images(shape)
    a2 = a.clone().double().requires_grad_(True)
    v2 = train_loss(a2, b.double(), lam)
    (v2 * 3.0).backward()
    a_np, b_np = np.ascontiguousarray(a.numpy()), np.ascontiguousarray(b.numpy())
    loss = np.zeros(4, dtype=np.float32)
    grad = np.zeros(shape, dtype=np.float32)
    assert lib.simt_train_loss(shape[0], shape[1], shape[2], ptr(a_np), ptr(b_np), C.c_float(lam), C.c_float(3.0), ptr(loss), ptr(grad)) == 0, lib.simt_loss_last_error()
    assert abs(float(loss[0]) - v2.item()) < 2e-6
    d = np.abs(grad.astype(np.float64) - a2.grad.numpy()).max()
    assert d <= 2e-5 * a2.grad.abs().max().item(), d


@pytest.mark.parametrize("shape", [(3, 67, 93), (1, 16, 16), (3, 5, 7), (3, 11, 300)])
def test_fused_mean_ssim_source_on_the_cpu(lib, shape):
    from oracle.losses import ssim as torch_ssim
    a, b = _images(shape)
    a2 = a.clone().double().requires_grad_(True)
    v2 = torch_ssim(a2, b.double())
    (v2 * 3.0).backward()
    a_np, b_np = np.ascontiguousarray(a.numpy()), np.ascontiguousarray(b.numpy())
    mean = np.zeros(4, dtype=np.float32)
    grad = np.zeros(shape, dtype=np.float32)
    assert lib.simt_ssim_mean(shape[0], shape[1], shape[2], ptr(a_np), ptr(b_np), C.c_float(3.0), ptr(mean), ptr(grad)) == 0, lib.simt_loss_last_error()
    assert abs(float(mean[0]) - v2.item()) < 2e-6
    d = np.abs(grad.astype(np.float64) - a2.grad.numpy()).max()
    assert d <= 2e-5 * a2.grad.abs().max().item(), d
