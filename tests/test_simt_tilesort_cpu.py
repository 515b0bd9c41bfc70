"""The fused emission + two-level tile sort of csrc/tilesort.hip EXECUTED on the CPU, lane by lane, through the SIMT shim (tests/simt/: 256 fibers
per workgroup, wave64 ballots, DPP moves, LDS, barriers; the shim's header states what it does not model) and through ITS OWN LAUNCHERS
(`hipLaunchKernelGGL` runs the grid): `fill_block_first`, `emit_hist`, `emit_scatter`, `bucket_hist`, `bucket_scan`, `bucket_scatter`
against plain stable sorts of the frame's instances: packed words by level-1 bucket, then the reference's (tile, depth, index) list and the
tile ranges.  The kernel SOURCE is what is compiled here (g++, -Itests/simt ahead of the real HIP headers).

Test infrastructure: tests/_build/libsimt_tilesort.so is never part of the product."""
import ctypes as C
import os
import shutil
import subprocess
import zlib

import numpy as np
import pytest


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libsimt_tilesort.so")
TS_ITEMS = 4096

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


@pytest.fixture(scope="module")
def lib():
    from simt_build import build
    h = build("tilesort")
    h.simt_last_error.restype = C.c_char_p
    return h


def make_frame(rng, P, gx, gy, kind):
    """Depth-ordered rectangles (minx, maxx, miny, maxy) of P Gaussians; the tile-less ones come last, as the depth sort leaves them."""
    rects = []
    for _ in range(P):
        if kind == "small":
            w, h = rng.integers(1, 5), rng.integers(1, 5)
        elif kind == "ones":
            w, h = 1, 1
        elif kind == "columns":      # one tile wide, four high: 1024 Gaussians per block, 4096 one-tile pieces (> SEG_PCAP with nG <= TS_NGCAP)
            w, h = 1, 4
        elif kind == "wide":
            w, h = rng.integers(1, gx + 1), rng.integers(1, 4)
        elif kind == "huge":
            w, h = (gx, gy) if rng.random() < 0.02 else (rng.integers(1, 9), rng.integers(1, 9))
        else:
            w, h = rng.integers(1, 9), rng.integers(1, 9)
        w, h = min(int(w), gx), min(int(h), gy)
        minx, miny = int(rng.integers(0, gx - w + 1)), int(rng.integers(0, gy - h + 1))
        rects.append((minx, minx + w, miny, miny + h))
    n_dead = int(P * 0.1)
    rects = rects[: P - n_dead] + [(3, 3, 2, 2)] * n_dead       # zero tiles
    return rects


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


CASES = [("mixed", 3000, 120, 68), ("small", 6000, 120, 68), ("wide", 900, 120, 68), ("huge", 700, 120, 68), ("ones", 9000, 37, 21),
         ("columns", 4000, 120, 68), ("mixed", 2500, 250, 131), ("small", 5, 8, 8)]


@pytest.mark.parametrize("kind,P,gx,gy", CASES)
def test_level1_kernels_on_the_cpu_equal_a_stable_sort_by_bucket(lib, kind, P, gx, gy):
    rng = np.random.default_rng(zlib.crc32(f"simt-{kind}-{P}-{gx}-{gy}".encode()))
    rects4 = make_frame(rng, P, gx, gy, kind)
    n_tiles = gx * gy
    nbits = 1
    while (1 << nbits) < n_tiles:
        nbits += 1
    lb = (nbits + 1) // 2
    hb = nbits - lb
    nb1 = 1 << hb
    tiles = np.array([(r[1] - r[0]) * (r[3] - r[2]) for r in rects4], dtype=np.int64)
    incl = np.cumsum(tiles)
    R = int(incl[-1])
    nblk = (R + TS_ITEMS - 1) // TS_ITEMS
    order = rng.permutation(P).astype(np.uint32)
    rect_sorted = np.array([[r[0] | (r[1] << 16), r[2] | (r[3] << 16)] for r in rects4], dtype=np.uint32)
    offsets = incl.astype(np.uint32)
    # ---- reference: every instance in emission order, stable sort by bucket ----
    inst_tile, inst_id, first_emission = [], [], {}
    for j, (minx, maxx, miny, maxy) in enumerate(rects4):
        if (maxx - minx) * (maxy - miny):
            first_emission[int(order[j])] = len(inst_tile)
        for y in range(miny, maxy):
            for x in range(minx, maxx):
                inst_tile.append(y * gx + x)
                inst_id.append(int(order[j]))
    inst_tile, inst_id = np.array(inst_tile, dtype=np.int64), np.array(inst_id, dtype=np.int64)
    bucket = inst_tile >> lb
    ref_words = ((inst_id << lb) | (inst_tile & ((1 << lb) - 1)))[np.argsort(bucket, kind="stable")].astype(np.uint32)
    ref_hist = np.zeros((nb1, nblk), dtype=np.int64)
    np.add.at(ref_hist, (bucket, np.arange(R) // TS_ITEMS), 1)
    # ---- the per-block table of first Gaussians: the shipped fill_block_first ----
    block_first = np.zeros((nblk + 2, 2), dtype=np.uint32)
    assert lib.simt_fill_block_first(P, ptr(offsets), ptr(block_first), nblk + 2) == 0, lib.simt_last_error()
    for b in range(nblk):
        j = int(np.searchsorted(incl, b * TS_ITEMS, side="right"))
        assert tuple(block_first[b]) == (j, int(incl[j] - tiles[j])), (b, block_first[b], j)
    # the launcher's own plan for this frame
    plb, phb, pw64 = C.c_int(), C.c_int(), C.c_int()
    lib.simt_tile_sort_plan(n_tiles, P, C.byref(plb), C.byref(phb), C.byref(pw64))
    assert (plb.value, phb.value, pw64.value) == (lb, hb, 0)
    # ---- level 1 through gsr_launch_tile_sort_level1 (emit_hist, the scan of the block histograms, the scatter), both scatters ----
    ref_before = (np.cumsum(ref_hist, axis=1) - ref_hist).astype(np.uint32)
    total = ref_hist.sum(axis=1).astype(np.uint32)
    res = {}
    for mode in (0,):
        words = np.full(R, 0xFFFFFFFF, dtype=np.uint32)
        hist1 = np.zeros(nb1 * nblk, dtype=np.uint32)
        digit_total = np.zeros(nb1, dtype=np.uint32)
        bucket_base = np.zeros(nb1 + 1, dtype=np.uint32)
        blk2_start = np.zeros(nb1 + 1, dtype=np.uint32)
        splats = np.zeros((P, 16), dtype=np.float32)
        rc = lib.simt_level1(0, C.c_int64(R), gx, lb, hb, ptr(block_first), ptr(offsets), ptr(rect_sorted), ptr(order), ptr(words), ptr(hist1),
                             ptr(digit_total), ptr(bucket_base), ptr(blk2_start), ptr(splats))
        assert rc == 0, lib.simt_last_error()
        assert np.array_equal(hist1.reshape(nb1, nblk), ref_before), "emit_hist (+ scan) differs from the reference histogram"
        assert np.array_equal(digit_total, total)
        if not np.array_equal(words, ref_words):
            bad = np.nonzero(words != ref_words)[0]
            raise AssertionError(f"mode {mode}: {bad.size} of {R} packed words differ, first at {bad[:6].tolist()}: got {words[bad[:6]].tolist()} want {ref_words[bad[:6]].tolist()}")
        assert np.array_equal(bucket_base[:-1].astype(np.int64), np.concatenate([[0], np.cumsum(total.astype(np.int64))])[:-1]) and int(bucket_base[-1]) == R
        assert np.array_equal(np.diff(blk2_start.astype(np.int64)), (total.astype(np.int64) + TS_ITEMS - 1) // TS_ITEMS)
        fe = splats.view(np.uint32)[:, 14]                    # 4th quad, word 2: first emission index (the backward's record address)
        for gid, k in first_emission.items():
            assert int(fe[gid]) == k, (mode, gid, int(fe[gid]), k)
        res[mode] = (words, fe.copy())
    # ---- level 2 on the CPU (bucket_hist, bucket_scan, bucket_scatter, both forms of the scan): the reference's (tile, depth, index) order ----
    o2 = np.argsort(inst_tile, kind="stable")
    ref_list = inst_id[o2].astype(np.uint32)
    cnt = np.bincount(inst_tile, minlength=n_tiles)
    starts = np.concatenate([[0], np.cumsum(cnt)])[:-1]
    ref_ranges = np.stack([np.where(cnt > 0, starts, 0), np.where(cnt > 0, starts + cnt, 0)], axis=1).astype(np.uint32)
    for scan_mode in (2, 1, 0):      # folded into the scatter / its own launch / the launcher's choice
        hist2 = np.zeros((nblk + 256 + nb1) * 256, dtype=np.uint32)
        tile_base = np.zeros(65536, dtype=np.uint32)
        point_list = np.full(R, 0xFFFFFFFF, dtype=np.uint32)
        ranges = np.full((n_tiles, 2), 0xFFFFFFFF, dtype=np.uint32)
        rc = lib.simt_level2(0, scan_mode, C.c_int64(R), n_tiles, lb, hb, ptr(res[0][0]), ptr(bucket_base), ptr(blk2_start), ptr(hist2), ptr(tile_base), ptr(point_list), ptr(ranges))
        assert rc == 0, lib.simt_last_error()
        assert np.array_equal(point_list, ref_list), f"level 2 (scan mode {scan_mode}): sorted point list differs"
        assert np.array_equal(ranges, ref_ranges), f"level 2 (scan mode {scan_mode}): tile ranges differ"


def test_level1_scatter_with_64_bit_words_on_the_cpu(lib):
    """The 8-byte word form (P << lb beyond 32 bits) of both level-1 scatters, on a frame whose Gaussian ids are large."""
    kind, P, gx, gy = "mixed", 1500, 120, 68
    rng = np.random.default_rng(7)
    rects4 = make_frame(rng, P, gx, gy, kind)
    lb, hb = 7, 6
    nb1 = 1 << hb
    tiles = np.array([(r[1] - r[0]) * (r[3] - r[2]) for r in rects4], dtype=np.int64)
    incl = np.cumsum(tiles)
    R = int(incl[-1])
    nblk = (R + TS_ITEMS - 1) // TS_ITEMS
    order = (rng.permutation(P).astype(np.uint32) + np.uint32(40_000_000))          # ids that need the wide word
    rect_sorted = np.array([[r[0] | (r[1] << 16), r[2] | (r[3] << 16)] for r in rects4], dtype=np.uint32)
    offsets = incl.astype(np.uint32)
    inst_tile, inst_id = [], []
    for j, (minx, maxx, miny, maxy) in enumerate(rects4):
        for y in range(miny, maxy):
            for x in range(minx, maxx):
                inst_tile.append(y * gx + x)
                inst_id.append(int(order[j]))
    inst_tile, inst_id = np.array(inst_tile, dtype=np.int64), np.array(inst_id, dtype=np.int64)
    bucket = inst_tile >> lb
    ref_words = ((inst_id << 32) | (inst_tile & ((1 << lb) - 1)))[np.argsort(bucket, kind="stable")].astype(np.uint64)
    ref_hist = np.zeros((nb1, nblk), dtype=np.int64)
    np.add.at(ref_hist, (bucket, np.arange(R) // TS_ITEMS), 1)
    block_first = np.zeros((nblk + 2, 2), dtype=np.uint32)
    assert lib.simt_fill_block_first(P, ptr(offsets), ptr(block_first), nblk + 2) == 0
    for mode in (0,):
        words = np.zeros(R, dtype=np.uint64)
        hist1 = np.zeros(nb1 * nblk, dtype=np.uint32)
        digit_total = np.zeros(nb1, dtype=np.uint32)
        bucket_base = np.zeros(nb1 + 1, dtype=np.uint32)
        blk2_start = np.zeros(nb1 + 1, dtype=np.uint32)
        rc = lib.simt_level1(1, C.c_int64(R), gx, lb, hb, ptr(block_first), ptr(offsets), ptr(rect_sorted), ptr(order), ptr(words), ptr(hist1),
                             ptr(digit_total), ptr(bucket_base), ptr(blk2_start), None)
        assert rc == 0, lib.simt_last_error()
        assert np.array_equal(words, ref_words), f"mode {mode}"
    # level 2 on the wide words
    o2 = np.argsort(inst_tile, kind="stable")
    hist2 = np.zeros((nblk + 256 + nb1) * 256, dtype=np.uint32)
    tile_base = np.zeros(65536, dtype=np.uint32)
    point_list = np.zeros(R, dtype=np.uint32)
    ranges = np.zeros((gx * gy, 2), dtype=np.uint32)
    assert lib.simt_level2(1, 0, C.c_int64(R), gx * gy, lb, hb, ptr(words), ptr(bucket_base), ptr(blk2_start), ptr(hist2), ptr(tile_base), ptr(point_list), ptr(ranges)) == 0, lib.simt_last_error()
    assert np.array_equal(point_list, inst_id[o2].astype(np.uint32))
