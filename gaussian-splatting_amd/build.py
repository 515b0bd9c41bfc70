"""Build libgsr_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python gaussian-splatting_amd/build.py [--force]

Translation units and their flags:
  preprocess.hip   -ffp-contract=off   (bit-exact radii / tile counts, see csrc/gsr_math.h)
  sort.hip, depthsort.hip, binning.hip (integer)
  render_fwd.hip, render_bwd.hip       (FMA contraction allowed; image tolerance 1e-5)
  gsr_api.cpp                          (host glue, C ABI)
The library is built IN-TREE (gaussian-splatting_amd/lib/) so it travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# GSR_AB=1: measurement build with the A/B kernel variants compiled in (-DGSR_AB_VARIANTS) -> lib_ab/libgsr_hip.so;
# select it at run time with GSR_LIB=<path>.  The default (product) build contains the default kernels only.
AB = os.environ.get("GSR_AB") == "1"
# tuning builds: GSR_OUT=<dir name under gaussian-splatting_amd/> GSR_EXTRA_FLAGS="-DGSR_TS_ITEMS=2048" python build.py
OUT = os.path.join(HERE, os.environ.get("GSR_OUT") or ("lib_ab" if AB else "lib"))
LIB = os.path.join(OUT, "libgsr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

AB_SRC = os.path.join(ROOT, "tools", "ab_variants")      # measured-and-rejected kernel variants: NOT part of the product tree, compiled into lib_ab/ only
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
          "-Wall", "-Wno-unused-function"] + (["-DGSR_AB_VARIANTS", "-I" + AB_SRC] if AB else []) + os.environ.get("GSR_EXTRA_FLAGS", "").split()
UNITS = [
    ("preprocess.hip", ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]),
    ("sort.hip", []),
    ("depthsort.hip", []),
    ("binning.hip", []),
    ("tilesort.hip", []),
    ("route.hip", []),
    # -fno-slp-vectorize: automatic v_pk_*_f32 packing costs more issue slots than it saves on gfx950 (forward blend -3 %)
    ("render_fwd.hip", ["-ffp-contract=fast", "-fno-slp-vectorize"]),
    ("render_bwd.hip", ["-ffp-contract=fast", "-fno-slp-vectorize"]),
    ("adam.hip", ["-ffp-contract=off"]),
    # no SLP vectorisation: the auto-packed v_pk_fma_f32 and the v_mov shuffles that assemble their operand pairs cost more
    # issue slots than the scalar FMAs they replace (forward 60.2 -> 55.3 us on one box); the per-Gaussian kernels were
    # A/B'd too (fused forms -1..2 %, split-SH forward +27 %) and keep the default
    ("ssim.hip", ["-ffp-contract=fast", "-fno-slp-vectorize"]),
    ("knn.hip", ["-ffp-contract=off"]),
    ("density.hip", ["-ffp-contract=off"]),
    ("gsr_api.cpp", ["-x", "hip"]),
]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    ab = AB_SRC                        # measured-and-rejected kernel variants, included by the sources under -DGSR_AB_VARIANTS only
    if AB and os.path.isdir(ab):
        hs += [os.path.join(ab, f) for f in os.listdir(ab) if f.endswith(".inc")]
    hs.append(os.path.join(ROOT, "include", "gsr.h"))
    hs.append(os.path.abspath(__file__))
    return hs


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OUT, exist_ok=True)
    headers = _headers()
    jobs = []
    objs = []
    for src, flags in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            if "-x" in flags:
                cmd = [HIPCC] + COMMON + flags + ["-c", s, "-o", o]
            else:
                cmd = [HIPCC] + COMMON + flags + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    linked = bool(jobs or force or _stale(LIB, objs))
    if linked:
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs +
            ["-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"])
    if verbose:
        print(f"[build] {os.path.relpath(LIB, ROOT)}: compiled {len(jobs)} of {len(UNITS)} translation units for {ARCH}"
              f" ({', '.join(os.path.basename(c[-3]) for c in jobs) or 'all objects up to date'}); {'linked' if linked else 'library up to date'}", flush=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
