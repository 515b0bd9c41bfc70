#!/usr/bin/env python
"""Static audit of the gfx950 ISA of every kernel in csrc/ (no GPU needed): registers, LDS, spills, and the pattern that cost
this project the most wall-clock per line of source -- SERIAL LOAD CHAINS.

    python tools/isa_audit.py [file.hip ...]          # default: every translation unit of build.py

A "chain" is a run of  global_load -> s_waitcnt vmcnt(0)  pairs: every load is waited for before the next one is issued, so the
kernel pays one full memory round trip (~1-2 us under load on MI355X) per element instead of one per batch.  hipcc produces
it from perfectly innocent source:

    for (k...) if (in_range(k)) lds[k] = global[k];           // load inside a branch      -> load, wait, ds_write, repeat
    for (k...) v[k] = global[k];  for (k...) if (c(k)) lds[k] = v[k];   // conditional USE -> the load is sunk into the branch

The cure used throughout csrc/: unconditional loads from a clamped (always valid) address into registers, THEN unconditional
LDS writes (surplus lanes write into spare rows).  Round 2 found and removed chains of 12 (SH block of the per-Gaussian
backward: 141 -> 126 us), 15 (SSIM halo staging), 4-6 (scan kernels) and IPT (rectangle gather of the depth sort's last pass).
The output lists, per kernel: VGPRs, LDS bytes, spilled VGPRs, number of vector loads, flat_* instructions (an LDS access and a
global access merged into one generic-pointer access: emit_scatter read its LDS records with four flat_load_dword per instance
until the two sources got separate loops) and the lengths of all chains >= 3."""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussian-splatting_amd", "csrc")
UNITS = {"preprocess.hip": ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"], "sort.hip": [], "depthsort.hip": [], "binning.hip": [],
         "tilesort.hip": [], "route.hip": [], "render_fwd.hip": ["-ffp-contract=fast", "-fno-slp-vectorize"], "render_bwd.hip": ["-ffp-contract=fast", "-fno-slp-vectorize"],
         "adam.hip": ["-ffp-contract=off"], "ssim.hip": ["-ffp-contract=fast", "-fno-slp-vectorize"], "knn.hip": ["-ffp-contract=off"],
         "density.hip": ["-ffp-contract=off"]}


def demangle_short(name: str) -> str:
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    name = re.sub(r"^_Z\d+", "", name)
    return name[:46]


def audit(path: str, flags, extra):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
               "-S", "--cuda-device-only", "-o", out, path] + flags + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:])
            raise SystemExit(1)
        s = open(out).read()
    meta = {}
    for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.vgpr_count:\s*(\d+)\s*\n\s*\.vgpr_spill_count:\s*(\d+)", s, re.S):
        meta[m.group(2)] = (int(m.group(3)), int(m.group(1)), int(m.group(4)))
    rows = []
    for m in re.finditer(r"^(_Z\w+): ; @.*?\n(.*?)^\.Lfunc_end", s, re.S | re.M):
        seq = []
        for line in m.group(2).split("\n"):
            t = line.strip().split()
            if not t:
                continue
            if t[0].startswith(("global_load", "flat_load", "buffer_load")):
                seq.append("L")
            elif t[0] == "s_waitcnt" and "vmcnt(0)" in line:
                seq.append("0")
            elif t[0] == "s_waitcnt" and "vmcnt" in line:
                seq.append("w")
        st = "".join(seq)
        chains = [len(c) // 2 for c in re.findall(r"(?:L0){3,}", st)]
        v, lds, sp = meta.get(m.group(1), (-1, -1, -1))
        # flat_* = the compiler lost the address space (e.g. it merged an LDS read and a global read into one pointer
        # select): such a load goes down both memory paths and waits on both counters
        nflat = len(re.findall(r"^\s+flat_(?:load|store|atomic)", m.group(2), re.M))
        rows.append((demangle_short(m.group(1)), v, lds, sp, st.count("L"), chains, nflat))
    return rows


def forward_walk_step(extra=()):
    """Instruction mix of ONE step of the forward blend's survivor walk in the inference build (render_fwd_wave_bf<true, 1, false>): the first unrolled
    copy of the inner loop's body, from the loop header to its closing branch -> {"valu": n, "salu": n, "ds": n}.  The kernel is bound by issue slots
    (DESIGN 4), so this count IS its cost model; tests/test_isa_audit_cpu.py pins it for the shipped form and for the candidate forms."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only",
               "-o", out, os.path.join(CSRC, "render_fwd.hip")] + UNITS["render_fwd.hip"] + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        s = open(out).read()
    m = re.search(r"^_ZN12_GLOBAL__N_118render_fwd_wave_bfILb1ELi1ELb0EEE\w+: ; @.*?\n(.*?)^\.Lfunc_end", s, re.S | re.M)
    if not m:
        raise RuntimeError("render_fwd_wave_bf<true, 1, false> not found")
    body = m.group(1)
    body = body[body.index("This Inner Loop Header"):]
    mix = {"valu": 0, "salu": 0, "ds": 0}
    seen_fma = 0
    for line in body.split("\n")[1:]:
        t = line.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        if op.startswith("v_"):
            mix["valu"] += 1
            seen_fma += op.startswith("v_fmac_f32")
        elif op.startswith("ds_"):
            mix["ds"] += 1
        elif op.startswith("s_") and op not in ("s_waitcnt", "s_nop"):
            mix["salu"] += 1
            if op.startswith("s_cbranch") and seen_fma >= 6:      # the branch that closes the step (two FMAs of the exponent, four of the accumulators)
                break
    return mix


def main():
    files = [a for a in sys.argv[1:] if a.endswith(".hip")]
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    if not files:
        files = list(UNITS)
    bad = 0
    for f in files:
        base = os.path.basename(f)
        print(f"== {base}")
        for name, v, lds, sp, nl, chains, nflat in audit(os.path.join(CSRC, base), UNITS.get(base, []), extra):
            flag = "   <-- serial load chain" if chains else ""
            spill = f" SPILLS {sp}" if sp > 0 else ""
            flat = f"  FLAT {nflat}" if nflat else ""
            print(f"  {name:46s} vgpr {v:3d}  lds {lds:6d}{spill}  loads {nl:3d}{flat}  chains {chains}{flag}")
            bad += len(chains)
    print(f"{bad} serial load chain(s) of length >= 3")


if __name__ == "__main__":
    main()
