"""The kernels AROUND the default path, executed on the CPU from their source through the SIMT shim (tests/simt/rows_harness.cpp) and their own
launchers: the LSD radix sort of csrc/sort.hip (32-bit keys on 27 / 32 bits with the rectangle gather of its last pass, 16-bit keys; three
workgroup sizes) against stable numpy sorts; the legacy binning path of csrc/binning.hip (tile scan, 32- and 16-bit instance emission, tile
ranges) against the reference's (tile, depth, index) order; `distCUDA2` (csrc/knn.hip: Morton sort, boxes, exact 3-NN query) against brute force;
the density-control statistics (csrc/density.hip) against the reference's expression (scene/gaussian_model.py:471-473, train.py:166).
Test infrastructure: tests/_build/libsimt_rows.so is never part of the product."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libsimt_rows.so")
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


@pytest.fixture(scope="module")
def lib():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "tests", "simt"),
                           "-I" + os.path.join(ROOT, "gaussian-splatting_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-x", "c++",
                           os.path.join(ROOT, "tests", "simt", "rows_harness.cpp"), "-o", OUT])
    h = C.CDLL(OUT)
    h.simt_rows_last_error.restype = C.c_char_p
    h.simt_legacy_bins.restype = C.c_int64
    return h


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("n,nbits,digit,items", [(10_000, 27, 9, 1024), (9_001, 27, 9, 2048), (20_000, 27, 9, 4096), (5_000, 32, 8, 1024), (300, 13, 8, 1024), (1, 27, 9, 1024)])
def test_lsd_radix_sort_pairs_on_the_cpu(lib, n, nbits, digit, items):
    rng = np.random.default_rng(n + nbits)
    keys = rng.integers(0, 1 << nbits, n, dtype=np.uint64).astype(np.uint32)
    keys[rng.random(n) < 0.3] = keys[0]                        # ties: stability
    vals = np.arange(n, dtype=np.uint32)
    rect = rng.integers(0, 1 << 31, (n, 2)).astype(np.uint32)
    rect_sorted = np.zeros_like(rect)
    k, v = keys.copy(), vals.copy()
    assert lib.simt_radix_sort(0, C.c_int64(n), nbits, digit, items, ptr(k), ptr(v), ptr(rect), ptr(rect_sorted)) == 0, lib.simt_rows_last_error()
    ref = np.argsort(keys, kind="stable")
    assert np.array_equal(v, ref.astype(np.uint32)) and np.array_equal(k, keys[ref])
    assert np.array_equal(rect_sorted, rect[ref]), "the last pass's rectangle gather"


def test_lsd_radix_sort_with_16_bit_keys_on_the_cpu(lib):
    rng = np.random.default_rng(3)
    n = 30_000
    keys = rng.integers(0, 8160, n).astype(np.uint16)
    vals = rng.permutation(n).astype(np.uint32)
    k, v = keys.copy(), vals.copy()
    assert lib.simt_radix_sort(1, C.c_int64(n), 13, 8, 4096, ptr(k), ptr(v), None, None) == 0, lib.simt_rows_last_error()
    ref = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[ref]) and np.array_equal(v, vals[ref])


@pytest.mark.parametrize("key16", [0, 1])
def test_legacy_binning_path_on_the_cpu(lib, key16):
    """Scan + emission + LSD tile sort + ranges: what frames with more than 65536 tiles (32-bit tile ids) and the A/B option tile_sort_mode = 1
    (16-bit ids) run instead of the fused two-level sort."""
    rng = np.random.default_rng(11 + key16)
    P, gx, gy = 3000, 120, 68
    w, h = rng.integers(1, 7, P), rng.integers(1, 5, P)
    w[rng.random(P) < 0.1] = 0
    minx, miny = rng.integers(0, gx - 6, P), rng.integers(0, gy - 4, P)
    rect = np.stack([minx | ((minx + w) << 16), miny | ((miny + h) << 16)], axis=1).astype(np.uint32)
    order = rng.permutation(P).astype(np.uint32)                         # the depth order
    offsets = np.zeros(P, dtype=np.uint32)
    rect_sorted = np.zeros((P, 2), dtype=np.uint32)
    r_cap = 200_000
    point_list = np.zeros(r_cap, dtype=np.uint32)
    ranges = np.zeros((gx * gy, 2), dtype=np.uint32)
    R = lib.simt_legacy_bins(P, gx, gy, ptr(order), ptr(rect), ptr(offsets), ptr(rect_sorted), ptr(point_list), C.c_int64(r_cap), ptr(ranges), key16)
    assert R >= 0, lib.simt_rows_last_error()
    tiles = (w * h)[order]
    assert R == int(tiles.sum()) and np.array_equal(offsets.astype(np.int64), np.cumsum(tiles))
    assert np.array_equal(rect_sorted, rect[order])
    it, ii = [], []
    for j in order:
        if w[j]:
            ys, xs = np.meshgrid(np.arange(miny[j], miny[j] + h[j]), np.arange(minx[j], minx[j] + w[j]), indexing="ij")
            it.append((ys * gx + xs).reshape(-1))
            ii.append(np.full(int(w[j] * h[j]), j))
    it, ii = np.concatenate(it), np.concatenate(ii)
    assert np.array_equal(point_list[:R], ii[np.argsort(it, kind="stable")].astype(np.uint32)), "sorted point list differs"
    cnt = np.bincount(it, minlength=gx * gy)
    st = np.concatenate([[0], np.cumsum(cnt)])[:-1]
    assert np.array_equal(ranges, np.stack([np.where(cnt > 0, st, 0), np.where(cnt > 0, st + cnt, 0)], axis=1).astype(np.uint32)), "tile ranges differ"


@pytest.mark.parametrize("kind,N", [("cloud", 3000), ("duplicates", 1200), ("plane", 2000), ("tiny", 3), ("line", 700)])
def test_knn_mean_dist2_on_the_cpu_equals_brute_force(lib, kind, N):
    from oracle.knn_oracle import dist2_mean3
    rng = np.random.default_rng(N)
    pts = rng.normal(size=(N, 3)).astype(np.float32)
    if kind == "duplicates":
        pts[N // 2:] = pts[: N - N // 2]
    elif kind == "plane":
        pts[:, 2] = 0.25
    elif kind == "line":
        pts[:, 1:] = 0.0
    out = np.zeros(N, dtype=np.float32)
    assert lib.simt_knn(N, ptr(pts), ptr(out)) == 0, lib.simt_rows_last_error()
    ref = dist2_mean3(pts)
    assert np.allclose(out.astype(np.float64), ref, rtol=2e-6, atol=1e-12), float(np.abs(out - ref).max())


def test_density_statistics_on_the_cpu(lib):
    rng = np.random.default_rng(5)
    P = 5000
    grad = rng.normal(size=(P, 3)).astype(np.float32)
    radii = rng.integers(0, 40, P).astype(np.int32)
    radii[rng.random(P) < 0.4] = 0
    accum = rng.random((P, 1)).astype(np.float32)
    denom = rng.integers(0, 5, (P, 1)).astype(np.float32)
    maxr = rng.integers(0, 30, P).astype(np.float32)
    a, d, m = accum.copy(), denom.copy(), maxr.copy()
    assert lib.simt_density_stats(P, ptr(grad), None, ptr(radii), ptr(a), ptr(d), ptr(m)) == 0, lib.simt_rows_last_error()
    vis = radii > 0
    norm = np.sqrt((grad[:, 0].astype(np.float32) ** 2 + grad[:, 1].astype(np.float32) ** 2).astype(np.float32)).astype(np.float32)
    assert np.allclose(a[:, 0], np.where(vis, accum[:, 0] + norm, accum[:, 0]), rtol=1e-6)
    assert np.array_equal(d[:, 0], np.where(vis, denom[:, 0] + 1, denom[:, 0]))
    assert np.array_equal(m, np.where(vis, np.maximum(maxr, radii.astype(np.float32)), maxr))
