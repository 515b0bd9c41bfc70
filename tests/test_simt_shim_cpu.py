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
    src = os.path.join(ROOT, "tests", "simt", "selftest_harness.cpp")
    from simt_build import build
    assert build("selftest", fp_contract_off=True).simt_selftest() == 0
