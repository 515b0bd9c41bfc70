"""Self-test of the SIMT-on-CPU shim (tests/simt/) that the test_simt_*_cpu.py suites run the kernel source on: every wave-level primitive the
csrc/ kernels use, against plain loops (tests/simt/selftest.cpp)."""
import ctypes as C
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


def test_shim_primitives():
    out = os.path.join(ROOT, "tests", "_build", "libsimt_selftest.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "tests", "simt"),
                           "-I" + os.path.join(ROOT, "gaussian-splatting_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-x", "c++",
                           os.path.join(ROOT, "tests", "simt", "selftest.cpp"), "-o", out])
    assert C.CDLL(out).simt_selftest() == 0
