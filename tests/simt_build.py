"""Builds a harness of tests/simt/ (kernel source of csrc/ compiled for the host against the SIMT-on-CPU shim) into tests/_build/ and loads it.
GSR_SIMT_EXTRA_FLAGS adds compiler flags -- e.g. the kernel source under AddressSanitizer:
    GSR_SIMT_EXTRA_FLAGS="-fsanitize=address -g" LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \\
        python -m pytest tests -q -k simt
(round 4: the whole shim suite -- kernels, the C-ABI host code of gsr_api.cpp, the Python package and the gloo workers above it -- is clean under it, and
under "-fsanitize=undefined -fno-sanitize-recover=undefined" and "-fsanitize=float-cast-overflow,float-divide-by-zero" (run those with pytest -s: the
reports go to stderr): no out-of-range float -> int conversion, shift or signed overflow anywhere on the tested paths).
SIMT_SCHEDULE=<seed> shuffles the wave interleaving inside a workgroup and the order of the workgroups of a grid (tests/simt/simt_runtime.h): the suite
passes under it, i.e. no result depends on either.  Test infrastructure."""
import ctypes as C
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Every build here defines the same exported names as libgsr_hip.so (gsr_launch_*, the C ABI).  The package loads the product library RTLD_GLOBAL, so without
# this flag a host build loaded later in the same process would have its INTERNAL calls resolved to the product library's functions (ELF interposition).
BSYM = "-Wl,-Bsymbolic"
# Host compiler: the ROCm tree's clang++ as a plain x86 compiler where it exists (it accepts the LDS section of tests/simt/hip/hip_runtime.h, so LDS is
# poisoned before every workgroup), else g++.  GSR_SIMT_CXX=g++ runs the suite with the other compiler (round 4: same results, bit-exact comparisons included).
_CLANG = "/opt/rocm/lib/llvm/bin/clang++"
CXX = os.environ.get("GSR_SIMT_CXX", _CLANG if os.path.exists(_CLANG) else "g++")


def build(name: str, fp_contract_off: bool = False, defines=(), tag: str = "") -> C.CDLL:
    """tests/simt/<name>_harness.cpp -> tests/_build/libsimt_<name><tag>.so.  `defines` (-D flags, with a `tag` for the file name) builds the same
    harness over a candidate form of the kernel source (the macros of csrc/ that are off in the product)."""
    out = os.path.join(ROOT, "tests", "_build", f"libsimt_{name}{tag}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [CXX, "-O1", "-std=c++17", "-shared", "-fPIC", BSYM, "-I" + os.path.join(ROOT, "tests", "simt"), "-I" + os.path.join(ROOT, "gaussian-splatting_amd", "csrc"),
           "-I" + os.path.join(ROOT, "include")] + (["-ffp-contract=off"] if fp_contract_off else []) + list(defines) + os.environ.get("GSR_SIMT_EXTRA_FLAGS", "").split()
    tmp = f"{out}.{os.getpid()}.tmp"      # written aside and renamed: a process that has the previous file mapped keeps its own copy
    subprocess.check_call(cmd + ["-x", "c++", os.path.join(ROOT, "tests", "simt", f"{name}_harness.cpp"), "-o", tmp])
    os.replace(tmp, out)
    return C.CDLL(out)


_library = None


def build_library() -> str:
    """The WHOLE library -- every translation unit of gaussian-splatting_amd/build.py incl. the C-ABI host code of csrc/gsr_api.cpp -- compiled for the
    host against the shim (its slice of the HIP runtime API included: one device whose memory is host memory, launches complete on return) into
    tests/_build/libgsr_simt.so: the same exported C ABI as libgsr_hip.so, every kernel lane a fiber.  Returns the path."""
    global _library
    if _library is not None:      # once per test process
        return _library
    import importlib.util
    spec = importlib.util.spec_from_file_location("gsr_build_units", os.path.join(ROOT, "gaussian-splatting_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out_dir = os.path.join(ROOT, "tests", "_build", f"simt_lib.{os.getpid()}")
    os.makedirs(out_dir, exist_ok=True)
    csrc = os.path.join(ROOT, "gaussian-splatting_amd", "csrc")
    common = [CXX, "-O1", "-std=c++17", "-fPIC", "-ffp-contract=off", "-D__HIPCC__=1", "-I" + os.path.join(ROOT, "tests", "simt"), "-I" + csrc,
              "-I" + os.path.join(ROOT, "include")] + os.environ.get("GSR_SIMT_EXTRA_FLAGS", "").split()
    objs, procs = [], []
    rt = os.path.join(out_dir, "simt_rt.cpp")
    open(rt, "w").write('#include "hip/hip_runtime.h"\n#include "simt_runtime.h"\n')
    for src in [u[0] for u in mod.UNITS] + [rt]:
        path = src if os.path.isabs(src) else os.path.join(csrc, src)
        obj = os.path.join(out_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        procs.append(subprocess.Popen(common + ["-x", "c++", "-c", path, "-o", obj]))
    for pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("g++ failed on a translation unit of the shim build of the library")
    lib = os.path.join(ROOT, "tests", "_build", "libgsr_simt.so")
    subprocess.check_call([CXX, "-shared", BSYM, "-o", lib + f".{os.getpid()}.tmp"] + objs + os.environ.get("GSR_SIMT_EXTRA_FLAGS", "").split())
    os.replace(lib + f".{os.getpid()}.tmp", lib)      # (see build())
    shutil.rmtree(out_dir, ignore_errors=True)
    _library = lib
    return lib
