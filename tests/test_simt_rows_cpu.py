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
    from simt_build import build
    h = build("rows", fp_contract_off=True)
    h.simt_rows_last_error.restype = C.c_char_p
    h.simt_legacy_bins.restype = C.c_int64
    return h


@pytest.fixture
def sort_lib(lib):
    return lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("n,nbits,digit,items", [(10_000, 27, 9, 1024), (9_001, 27, 9, 2048), (20_000, 27, 9, 4096), (5_000, 32, 8, 1024), (300, 13, 8, 1024), (1, 27, 9, 1024)])
def test_lsd_radix_sort_pairs_on_the_cpu(sort_lib, n, nbits, digit, items):
    lib = sort_lib
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


def test_lsd_radix_sort_with_16_bit_keys_on_the_cpu(sort_lib):
    lib = sort_lib
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


def test_fused_adam_source_on_the_cpu_equals_torch_adam(lib):
    """gsr_adam_step_multi (one launch over the tensors of a 3DGS model: ragged sizes, an unaligned view) and gsr_adam_step against
    torch.optim.Adam with the reference's settings (eps 1e-15, scene/gaussian_model.py:178-211), three steps."""
    import torch

    class T(C.Structure):
        _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64),
                    ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("step", C.c_int32), ("reserved", C.c_int32)]
    g = torch.Generator().manual_seed(3)
    shapes, lrs = [(1000, 3), (1000, 15, 3), (1000, 1), (1000, 4), (7,), (4099,)], [1.6e-4, 1.25e-4, 2.5e-2, 1e-3, 5e-3, 1e-2]
    params = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in params]
    opt = torch.optim.Adam([{"params": [r], "lr": lr} for r, lr in zip(ref, lrs)], lr=0.0, eps=1e-15)
    mine = [np.ascontiguousarray(p.numpy().copy()) for p in params]
    m = [np.zeros_like(x) for x in mine]
    v = [np.zeros_like(x) for x in mine]
    for step in (1, 2, 3):
        grads = [torch.randn(s, generator=g) * (0.1 if i % 2 else 10.0) for i, s in enumerate(shapes)]
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        opt.step()
        gn = [np.ascontiguousarray(gr.numpy()) for gr in grads]
        arr = (T * len(mine))(*[T(ptr(mine[i]).value, ptr(gn[i]).value, ptr(m[i]).value, ptr(v[i]).value, mine[i].size, lrs[i], 0.9, 0.999, 1e-15, step, 0) for i in range(len(mine))])
        assert lib.simt_adam_multi(arr, len(mine)) == 0, lib.simt_rows_last_error()
        for x, r in zip(mine, ref):
            assert np.allclose(x, r.detach().numpy(), rtol=2e-6, atol=1e-9), float(np.abs(x - r.detach().numpy()).max())
    # single-tensor entry point, starting at an address that is not 16-byte aligned (scalar path)
    base = np.zeros(1001, dtype=np.float32)
    p1, g1, m1, v1 = base[1:].copy(), np.random.default_rng(0).normal(size=1000).astype(np.float32), np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    r1 = torch.from_numpy(p1.copy()).requires_grad_(True)
    o1 = torch.optim.Adam([r1], lr=1e-3, eps=1e-15)
    r1.grad = torch.from_numpy(g1.copy())
    o1.step()
    assert lib.simt_adam(ptr(p1), ptr(g1), ptr(m1), ptr(v1), C.c_int64(1000), C.c_double(1e-3), C.c_double(0.9), C.c_double(0.999), C.c_double(1e-15), 1) == 0
    assert np.allclose(p1, r1.detach().numpy(), rtol=2e-6, atol=1e-9)


def test_sparse_adam_source_on_the_cpu(lib):
    """SparseGaussianAdam.step(visible, N) [RECALLED semantics, csrc/adam.hip]: rows of invisible Gaussians keep parameter AND moments; visible
    rows take m = b1 m + (1 - b1) g, v = b2 v + (1 - b2) g^2, p -= lr m / (sqrt(v) + eps) -- no bias correction."""
    rng = np.random.default_rng(9)
    for N, M in ((3000, 3), (3000, 45), (777, 1), (513, 4)):
        p, g = rng.normal(size=(N, M)).astype(np.float32), rng.normal(size=(N, M)).astype(np.float32)
        m, v = rng.normal(size=(N, M)).astype(np.float32) * 0.1, rng.random((N, M)).astype(np.float32) * 0.01
        vis = (rng.random(N) < 0.6).astype(np.uint8)
        p0, m0, v0 = p.copy(), m.copy(), v.copy()
        lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-15
        assert lib.simt_sparse_adam(ptr(p), ptr(g), ptr(m), ptr(v), ptr(vis), C.c_int64(N), C.c_int64(M), C.c_double(lr), C.c_double(b1), C.c_double(b2), C.c_double(eps)) == 0
        mm = (np.float32(b1) * m0 + np.float32(1 - b1) * g).astype(np.float32)
        vv = (np.float32(b2) * v0 + np.float32(1 - b2) * g * g).astype(np.float32)
        pp = (p0 - np.float32(lr) * mm / (np.sqrt(vv) + np.float32(eps))).astype(np.float32)
        sel = vis.astype(bool)[:, None]
        assert np.allclose(m, np.where(sel, mm, m0), rtol=1e-6, atol=1e-12) and np.allclose(v, np.where(sel, vv, v0), rtol=1e-6, atol=1e-12)
        assert np.allclose(p, np.where(sel, pp, p0), rtol=2e-6, atol=1e-9)
        assert np.array_equal(p[~vis.astype(bool)], p0[~vis.astype(bool)])


def test_sparse_adam_multi_tensor_launch_equals_the_single_tensor_launches(lib):
    """gsr_sparse_adam_step_multi (round 5: the six parameter groups of SparseGaussianAdam.step in ONE launch) leaves every tensor the bits the
    single-tensor launch leaves -- the 3DGS row widths (3, 3, 45, 1, 3, 4), an unaligned view (scalar walk), per-tensor lr / eps, one shared mask."""
    class T(C.Structure):
        _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("M", C.c_int64),
                    ("lr", C.c_double), ("eps", C.c_double)]
    rng = np.random.default_rng(11)
    N = 2500
    vis = (rng.random(N) < 0.4).astype(np.uint8)
    widths, lrs = (3, 3, 45, 1, 3, 4), (1.6e-4, 2.5e-3, 1.25e-4, 0.025, 5e-3, 1e-3)
    a, b = [], []
    for k, M in enumerate(widths):
        base = [rng.normal(size=N * M + 1).astype(np.float32) for _ in range(2)] + [rng.random(N * M + 1).astype(np.float32) * 0.01 for _ in range(2)]
        off = 1 if k == 4 else 0      # one tensor starts 4 bytes off the 16-byte grid: the scalar walk
        a.append([x[off:off + N * M].copy() if not off else x[off:off + N * M] for x in [y.copy() for y in base]])
        b.append([x[off:off + N * M].copy() if not off else x[off:off + N * M] for x in [y.copy() for y in base]])
    arr = (T * len(widths))()
    for k, M in enumerate(widths):
        p, g, m, v = a[k]
        arr[k] = T(p.ctypes.data, g.ctypes.data, m.ctypes.data, v.ctypes.data, M, lrs[k], 1e-15)
    assert lib.simt_sparse_adam_multi(arr, len(widths), ptr(vis), C.c_int64(N), C.c_double(0.9), C.c_double(0.999)) == 0, lib.simt_rows_last_error()
    for k, M in enumerate(widths):
        p, g, m, v = b[k]
        assert lib.simt_sparse_adam(ptr(p), ptr(g), ptr(m), ptr(v), ptr(vis), C.c_int64(N), C.c_int64(M), C.c_double(lrs[k]), C.c_double(0.9), C.c_double(0.999),
                                    C.c_double(1e-15)) == 0
        for x, y, name in zip(a[k], b[k], ("param", "grad", "exp_avg", "exp_avg_sq")):
            assert np.array_equal(x, y), f"tensor {k} ({M} per row): {name} differs between the multi-tensor and the single-tensor launch"
        assert not np.array_equal(a[k][0].reshape(N, M)[vis.astype(bool)], 0 * a[k][0].reshape(N, M)[vis.astype(bool)])
