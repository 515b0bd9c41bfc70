"""Builds a harness of tests/simt/ (kernel source of csrc/ compiled for the host against the SIMT-on-CPU shim) into tests/_build/ and loads it.
GSR_SIMT_EXTRA_FLAGS adds compiler flags -- e.g. the kernel source under AddressSanitizer:
    GSR_SIMT_EXTRA_FLAGS="-fsanitize=address -g" LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \\
        python -m pytest tests -q -k simt
(round 4: the whole shim suite is clean under it).  Test infrastructure."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(name: str, fp_contract_off: bool = False) -> C.CDLL:
    out = os.path.join(ROOT, "tests", "_build", f"libsimt_{name}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "tests", "simt"), "-I" + os.path.join(ROOT, "gaussian-splatting_amd", "csrc"),
           "-I" + os.path.join(ROOT, "include")] + (["-ffp-contract=off"] if fp_contract_off else []) + os.environ.get("GSR_SIMT_EXTRA_FLAGS", "").split()
    subprocess.check_call(cmd + ["-x", "c++", os.path.join(ROOT, "tests", "simt", f"{name}_harness.cpp"), "-o", out])
    return C.CDLL(out)
