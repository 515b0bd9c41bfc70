"""The bucket depth sort of csrc/depthsort.hip (ds_hist, ds_scan, ds_scatter, ds_segsort -- the four launches that replaced eleven in round 4)
EXECUTED on the CPU, lane by lane, through the SIMT shim of tests/simt/ (512-thread workgroups, wave64 ballots, LDS radix passes): depth order,
gathered rectangles, the inclusive scan of the tile counts and the emission's per-block table against a stable numpy sort -- on the key
distributions that select its paths (uniform, ties, a crowded bucket that overflows the LDS segment and goes through global memory, a gap of
four decades, one key, tile-less Gaussians, ragged sizes; round 6: far outliers, heavy tails and a wall, which the histogram-equalised buckets
must keep inside the LDS segments).  The kernel SOURCE is what runs (g++ -Itests/simt); the GPU tests of the same
properties are tests/test_gpu_bins_sweep.py.  Test infrastructure: tests/_build/libsimt_depthsort.so is never part of the product."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libsimt_depthsort.so")
CULLED = (1 << 27) - 1
TS_ITEMS = 4096

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


@pytest.fixture(scope="module")
def lib():
    from simt_build import build
    h = build("depthsort")
    h.simt_ds_last_error.restype = C.c_char_p
    return h


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_keys(rng, P, kind):
    base = 0x00C00000
    if kind == "uniform":
        k = base + rng.integers(0, 1 << 22, P)
    elif kind == "ties":
        k = base + rng.integers(0, 64, P) * 4099
    elif kind == "crowd":                     # 3/4 of the keys inside 48 consecutive values: one bucket far beyond the LDS segment
        k = np.where(rng.random(P) < 0.75, base + 777_000 + rng.integers(0, 48, P), base + rng.integers(0, 1 << 22, P))
    elif kind == "crowd_16":                  # 3/4 of the keys on 16 values: buckets of equal keys several LDS capacities long
        k = np.where(rng.random(P) < 0.75, base + 777_000 + rng.integers(0, 16, P), base + rng.integers(0, 1 << 22, P))
    elif kind == "gap":
        k = np.where(rng.random(P) < 0.5, base + rng.integers(0, 4096, P), base + (1 << 25) + rng.integers(0, 1 << 12, P))
    elif kind == "one_key":
        k = np.full(P, base + 12345)
    elif kind == "outliers":                  # the bulk in a narrow band, a handful of keys far below / above it
        k = base + (1 << 24) + rng.integers(0, 1 << 16, P)
        k[rng.integers(0, P, 9)] = base + (1 << 26) + rng.integers(0, 1 << 25, 9)
        k[rng.integers(0, P, 4)] = rng.integers(1, 4096, 4)
    elif kind == "heavy_tails":               # round 6: 3 % of the keys spread over the whole key space around a narrow bulk (every workgroup has some)
        k = base + (1 << 24) + rng.integers(0, 1 << 16, P)
        far = rng.random(P) < 0.03
        k[far] = rng.integers(1, (1 << 27) - 2, int(far.sum()))
    elif kind == "wall_thin":                 # 60 % of the keys within 1700 consecutive values (narrower than a bucket of the first-level table)
        k = np.where(rng.random(P) < 0.6, base + (1 << 25) + 4321 + rng.integers(0, 1700, P), base + rng.integers(0, 5 << 23, P))
    elif kind == "wall":                      # half of the keys within 2^13 consecutive values, the rest over five octaves
        k = np.where(rng.random(P) < 0.5, base + (1 << 25) + rng.integers(0, 1 << 13, P), base + rng.integers(0, 5 << 23, P))
    else:
        raise ValueError(kind)
    return k.astype(np.int64)


@pytest.mark.parametrize("kind,P", [("uniform", 20_000), ("uniform", 4_097), ("uniform", 300), ("ties", 12_000), ("crowd", 16_000), ("gap", 9_000), ("one_key", 5_000),
                                    ("outliers", 24_000), ("heavy_tails", 40_000), ("wall", 40_000), ("wall_thin", 40_000), ("crowd", 400_000), ("crowd_16", 400_000), ("one_key", 80_000), ("one_key", 2_200_000)])
def test_bucket_depth_sort_on_the_cpu_equals_a_stable_sort(lib, kind, P):
    rng = np.random.default_rng(1000 + P + len(kind))
    keys = make_keys(rng, P, kind)
    # rectangles (minx | maxx << 16, miny | maxy << 16) and their tile counts = areas (the segment kernel scans the areas of the gathered rectangles)
    w, h = rng.integers(1, 6, P), rng.integers(1, 4, P)
    dead = rng.random(P) < 0.12                                       # no tile in the band: the key-producing kernel gives them the last key
    w[dead] = 0
    minx, miny = rng.integers(0, 100, P), rng.integers(0, 60, P)
    rect = np.stack([minx | ((minx + w) << 16), miny | ((miny + h) << 16)], axis=1).astype(np.uint32)
    tiles = (w * h).astype(np.int64)
    keys[tiles == 0] = CULLED
    listed = tiles > 0
    # the key-producing kernel's per-workgroup key ranges: workgroup c of n_range 256-thread workgroups owns the keys c * 256 + t + k * n_range * 256
    n_range = min(1024, (P + 255) // 256) if kind in ("outliers", "heavy_tails", "wall", "wall_thin") or P > 100_000 else 5
    wg = np.zeros((n_range, 2), dtype=np.uint32)
    parts = []
    for c in range(n_range):
        idx = (np.arange(c * 256, P, n_range * 256)[:, None] + np.arange(256)[None, :]).reshape(-1)
        parts.append(idx[idx < P])
    for c, idx in enumerate(parts):
        kk = keys[idx][listed[idx]]
        if kk.size:
            wg[c] = ((~np.uint32(kk.min())) & np.uint32(0xFFFFFFFF), kk.max())
    R = int(tiles.sum())
    frame = np.zeros(32, dtype=np.uint32)
    frame[0] = R
    keys32, tiles32 = keys.astype(np.uint32), tiles.astype(np.uint32)
    order = np.full(P, 0xFFFFFFFF, dtype=np.uint32)
    rect_sorted = np.zeros((P, 2), dtype=np.uint32)
    offsets = np.zeros(P, dtype=np.uint32)
    nblk = (R + TS_ITEMS - 1) // TS_ITEMS
    bf_cap = P // 64 + 66
    block_first = np.full((bf_cap, 2), 0xFFFFFFFF, dtype=np.uint32)
    slow = np.zeros(16, dtype=np.uint32)
    rc = lib.simt_depth_bucket_sort(P, ptr(keys32), ptr(tiles32), ptr(rect), ptr(frame), ptr(wg), n_range, ptr(order), ptr(rect_sorted), ptr(offsets),
                                    ptr(block_first), bf_cap, ptr(slow))
    assert rc == 0, lib.simt_ds_last_error()
    ref = np.argsort(keys, kind="stable")
    assert np.array_equal(order.astype(np.int64), ref), f"depth order differs at {np.nonzero(order.astype(np.int64) != ref)[0][:8].tolist()}"
    V = int(listed.sum())
    assert np.array_equal(rect_sorted[:V], rect[ref][:V]), "gathered rectangles differ"          # (the tile-less tail's rectangles are never read)
    incl = np.cumsum(tiles[ref])
    assert np.array_equal(offsets.astype(np.int64), incl), "inclusive scan of the tile counts differs"
    assert int(frame[6]) == int(keys[listed].min()) and int(frame[7]) == int(keys[listed].max()), "true key range of the frame"
    assert (int(frame[2]), int(frame[3])) == (int(frame[6]), int(frame[7]))
    if kind in ("outliers", "heavy_tails", "wall", "wall_thin", "gap") or (kind == "crowd" and P < 100_000):
        assert int(slow[0]) == 0, "with the equalised buckets no segment of this distribution may overflow the LDS capacity"
    for b in range(nblk):                                             # the Gaussian that holds the first instance of every emission block
        j = int(np.searchsorted(incl, b * TS_ITEMS, side="right"))
        assert tuple(int(v) for v in block_first[b]) == (j, int(incl[j] - tiles[ref][j])), (b, block_first[b].tolist(), j)
    last = int(np.nonzero(tiles[ref])[0][-1])
    assert int(block_first[nblk][0]) == last, (block_first[nblk].tolist(), last)
    if kind == "crowd" and P > 100_000:
        # 300 000 keys on 48 values = 6 250 equal keys per bucket: oversized segments remain, but of ONE key each -- no pass, only the chunked output
        assert int(slow[0]) == 0, "a bucket of a few thousand equal keys takes the chunked output path and is NOT reported as slow"
    elif kind == "crowd_16":
        # 300 000 keys on 16 values = 18 750 equal keys per bucket, ONE key value per bucket by the table: five chunks each, the first written by the segment's
        # own workgroup, the others by the workgroups of the windows the bucket covers
        assert int(slow[0]) == 0, "buckets of a few capacities of one key value are written by several workgroups and NOT reported as slow"
    elif kind == "one_key" and P > 16 * 4096:
        assert int(slow[0]) == 0, "one bucket of 70 000 equal keys: one key value by the table, 18 chunks written by 18 workgroups in constant time each"
    elif kind == "uniform":
        assert int(slow[0]) == 0
