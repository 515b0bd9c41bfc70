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
    lib = build("selftest", fp_contract_off=True)
    assert lib.simt_selftest() == 0
    assert lib.simt_selftest_lds_is_garbage() == 0, "a workgroup must not see zeros or its predecessor's values in LDS"


def test_shuffled_schedule_exposes_a_missing_barrier_and_a_block_order_assumption():
    """SIMT_SCHEDULE=<seed> sweeps the waves of a workgroup in a random order with a random half sitting out, and runs the workgroups of a grid in a random
    order (tests/simt/simt_runtime.h).  Two kernels that are wrong on a GPU -- an LDS hand-over without its barrier, a block that reads what its
    predecessor wrote -- pass in the default order and must be caught under the shuffled one; the primitives' self-test must not care."""
    import sys
    from simt_build import build
    build("selftest", fp_contract_off=True)
    code = ("import ctypes, sys; lib = ctypes.CDLL(sys.argv[1]); "
            "print(lib.simt_selftest(), lib.simt_selftest_detects_order_dependence())")
    path = os.path.join(ROOT, "tests", "_build", "libsimt_selftest.so")

    def run(seed):
        env = dict(os.environ)
        env.pop("SIMT_SCHEDULE", None)
        if seed:
            env["SIMT_SCHEDULE"] = str(seed)
        return tuple(int(v) for v in subprocess.check_output([sys.executable, "-c", code, path], env=env).split())

    assert run(0) == (0, 0)
    seen = 0
    for seed in range(1, 9):
        bad, flags = run(seed)
        assert bad == 0, f"seed {seed}: a primitive depends on the schedule"
        seen |= flags
    assert seen == 3, f"the shuffled schedule missed {'the missing barrier' if not seen & 1 else 'the block-order assumption'}"
