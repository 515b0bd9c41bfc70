"""TEST INFRASTRUCTURE ONLY -- CPU oracle for `simple_knn._C.distCUDA2`.

The source of simple-knn is an un-vendored submodule (.gitmodules:1-3), so this restates its contract from the one
call site, scene/gaussian_model.py:159-160: per point, the mean of the squared Euclidean distances to the three
nearest OTHER points (self excluded by index; a coincident duplicate is a neighbour at distance 0).
PARITY UNPINNED: no reference vectors exist for this function; the kernel is checked for exactness of the k-NN
set, not for the reference's floating-point summation order.

Only tests/ may import this module."""
from __future__ import annotations

import numpy as np


def dist2_mean3(points: np.ndarray) -> np.ndarray:
    """Brute force in float64 on the float32 coordinates: [N] mean of the three smallest squared distances."""
    p = np.asarray(points, dtype=np.float32).astype(np.float64)
    n = p.shape[0]
    out = np.empty(n, dtype=np.float64)
    chunk = max(1, min(1024, (1 << 22) // max(1, n)))
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        d = ((p[a:b, None, :] - p[None, :, :]) ** 2).sum(-1)
        d[np.arange(b - a), np.arange(a, b)] = np.inf
        k = min(3, n - 1)
        part = np.partition(d, k - 1, axis=1)[:, :k] if k > 0 else np.zeros((b - a, 0))
        # fewer than 3 other points: the reference leaves FLT_MAX in the unused slots
        fill = np.full((b - a, 3 - k), np.finfo(np.float32).max, dtype=np.float64)
        out[a:b] = np.concatenate([part, fill], axis=1).sum(1) / 3.0
    return out


def dist2_mean3_tree(points: np.ndarray) -> np.ndarray:
    """Same result through scipy's cKDTree (exact), for sizes where brute force is too slow."""
    from scipy.spatial import cKDTree
    p = np.asarray(points, dtype=np.float32).astype(np.float64)
    d, _ = cKDTree(p).query(p, k=4)
    # column 0 is the point itself (distance 0); with duplicates the self match may sit in another column, but the
    # multiset of the remaining three distances is the same
    return (d[:, 1:4] ** 2).sum(1) / 3.0
