"""The bucket depth sort (csrc/depthsort.hip) restated in numpy: bucket mapping from the frame's key range, segment plan (windows of
2048 elements over the bucket-ordered array, binary searches over the scanned bucket totals), per-segment stable sort on the rebased
keys, inclusive tile scan from the segment bases, and the emission block table.  The restatement is checked against a plain stable
sort -- the result the LSD radix sort and the reference's stable 64-bit-key sort produce -- on the depth
distributions that select the kernel's code paths.  It documents the algorithm's invariants on the CPU; the kernels themselves are
compared bit for bit with the oracle on the GPU (tests/test_gpu_bins_sweep.py).

Caught while this model was written: the segment that ends the listed Gaussians is identified by its element count, not by its end
bucket (empty buckets behind the last non-empty one start at the same element) -- the GPU kernel had that bug in its first version."""
import numpy as np
import pytest

NB, BITS, SEG, CAP, IT = 2048, 11, 2048, 4096, 4096
CULL_BUCKET = NB - 1
KEY_CULLED = (1 << 27) - 1
KEY_BASE = 0x3E4CCCCD          # bits(0.2f), gsr_internal.h GSR_DEPTH_KEY_BASE


def ds_shift(kmin, kmax):
    if kmax <= kmin:
        return 0
    rng = kmax - kmin
    s = max(0, rng.bit_length() - BITS)
    if (rng >> s) > NB - 2:
        s += 1
    return s


def robust_range(wg_min, wg_max, threads=256):
    """ds_hist's robust key range from the per-workgroup (min, max) table of the key-producing kernel (None entries: the workgroup listed
    nothing): thread t of 256 groups the workgroups t, t + 256, ...; the group's smallest maximum / largest minimum ignores an outlier unless
    every workgroup of the group has one; the robust range is the widest of the groups' ranges.  Falls back to the true range."""
    have = [(a, b) for a, b in zip(wg_min, wg_max) if a is not None]
    if not have:
        return 0xFFFFFFFF, 0
    tmin, tmax = min(a for a, _ in have), max(b for _, b in have)
    gmin, gmax = [], []
    for t in range(threads):
        grp = [(wg_min[i], wg_max[i]) for i in range(t, len(wg_min), threads) if wg_min[i] is not None]
        if grp:
            gmin.append(max(a for a, _ in grp))
            gmax.append(min(b for _, b in grp))
    kmin, kmax = min(gmin), max(gmax)
    if kmax <= kmin or kmin < tmin or kmax > tmax:
        return tmin, tmax
    return kmin, kmax


def bucket_depth_sort(keys, tiles, key_range=None):
    """-> (order, inclusive tile scan in depth order, block_first dict, segment sizes).  key_range: the ROBUST (kmin, kmax) the buckets
    span (robust_range); default: the true extremes."""
    P = len(keys)
    listed = keys != KEY_CULLED
    assert ((tiles > 0) == listed).all()
    tmin, tmax = (int(keys[listed].min()), int(keys[listed].max())) if listed.any() else (0xFFFFFFFF, 0)
    kmin, kmax = (tmin, tmax) if key_range is None else key_range
    assert tmin <= kmin and kmax <= tmax
    sh = ds_shift(kmin, kmax)
    d = np.where(listed, np.minimum((np.maximum(keys.astype(np.int64), kmin) - kmin) >> sh, NB - 2), CULL_BUCKET)
    assert d[listed].max(initial=0) <= NB - 2, "bucket 2047 is reserved for the tile-less Gaussians"
    cnt = np.bincount(d, minlength=NB)
    tsum = np.bincount(d, weights=tiles, minlength=NB).astype(np.int64)
    cnt_excl = np.concatenate([[0], np.cumsum(cnt)])[:NB]
    tile_excl = np.concatenate([[0], np.cumsum(tsum)])[:NB]
    n_listed = int(cnt_excl[CULL_BUCKET])
    by_bucket = np.argsort(d, kind="stable")            # ds_scatter: stable, bucket-major

    def lower_bound(x):                                 # first bucket in [0, 2047] whose first element is >= x
        lo, hi = 0, CULL_BUCKET
        while lo < hi:
            mid = (lo + hi) >> 1
            if cnt_excl[mid] >= x:
                hi = mid
            else:
                lo = mid + 1
        return lo
    order = np.empty(P, np.int64)
    scan = np.empty(P, np.int64)
    block_first, sizes, covered = {}, [], 0
    R = int(tiles.sum())
    for s in range((P + SEG - 1) // SEG + 1):
        x0 = s * SEG
        if x0 >= n_listed:
            continue
        d0, x1 = lower_bound(x0), x0 + SEG
        d1 = CULL_BUCKET if x1 >= n_listed else lower_bound(x1)
        b, e = int(cnt_excl[d0]), int(cnt_excl[d1])
        if e <= b:
            continue
        assert b == covered, "segments tile the listed Gaussians without gaps"
        covered = e
        sizes.append(e - b)
        ids = by_bucket[b:e]
        base_key = tmin if d0 == 0 else kmin + (d0 << sh)
        span = (tmax + 1 if d1 > NB - 2 else kmin + (d1 << sh)) - base_key
        rem = keys[ids].astype(np.int64) - base_key
        assert rem.min() >= 0 and rem.max() < max(span, 1), "rebased keys fit the segment's span"
        nbits = 0 if span <= 1 else int(span - 1).bit_length()
        ids = ids[np.argsort(rem & ((1 << nbits) - 1), kind="stable")] if nbits else ids
        order[b:e] = ids
        incl = tile_excl[d0] + np.cumsum(tiles[ids])
        scan[b:e] = incl
        last_listed = e - 1 if e == n_listed else -1      # (NOT "d1 == 2047")
        for j, g in enumerate(range(b, e)):
            ex, inc = int(incl[j] - tiles[ids[j]]), int(incl[j])
            for blk in range((ex + IT - 1) // IT, (inc - 1) // IT + 1):
                block_first[blk] = (g, ex)
            if g == last_listed:
                block_first[(inc + IT - 1) // IT] = (g, ex)
    assert covered == n_listed
    order[n_listed:] = by_bucket[n_listed:]
    scan[n_listed:] = R
    return order, scan, block_first, sizes


def keys_from_depths(z, culled):
    k = (np.asarray(z, np.float32).view(np.uint32).astype(np.int64) - KEY_BASE).astype(np.uint32)
    k[culled] = KEY_CULLED
    return k


CASES = ["uniform", "ties", "crowd", "gap", "one_key", "all_culled", "single", "last_bucket_straddles_a_window", "outliers"]


@pytest.mark.parametrize("case", CASES)
def test_bucket_depth_sort_model_equals_a_stable_sort(case):
    rng = np.random.default_rng(hash(case) % 2**32)
    P = {"single": 1, "all_culled": 3000}.get(case, 60_000)
    z = rng.uniform(2, 12, P)
    if case == "ties":
        z = 2 + rng.integers(0, 64, P) * 0.125
    elif case == "crowd":
        z[: 3 * P // 4] = np.float32(5.0) + rng.integers(0, 48, 3 * P // 4) * np.float32(4.76837158203125e-07)
    elif case == "gap":
        z = np.where(rng.random(P) < 0.5, 0.25 + 0.05 * rng.random(P), 3000 + 6000 * rng.random(P))
    elif case == "one_key":
        z[:] = 4.0
    elif case == "last_bucket_straddles_a_window":
        z = np.concatenate([rng.uniform(2, 3, P - 2500), np.full(2500, 12.0)])      # the farthest bucket alone holds 2500 > 2048 elements
    elif case == "outliers":      # a trained scene -- the bulk within a fraction of an octave, a handful of floaters 100x farther / nearer
        z = rng.uniform(4.0, 4.2, P)
        z[rng.integers(0, P, 12)] = rng.uniform(300, 3000, 12)
        z[rng.integers(0, P, 5)] = rng.uniform(0.21, 0.3, 5)
    culled = rng.random(P) < (1.0 if case == "all_culled" else 0.12)
    keys = keys_from_depths(z, culled)
    tiles = np.where(culled, 0, rng.integers(1, 40, P)).astype(np.int64)
    key_range = None
    if case == "outliers":
        # the key-producing kernel's workgroups sample the array with a grid stride: workgroup w of 1024 holds the Gaussians w, w + 1024, ...
        nwg = 1024
        wmin = [int(keys[w::nwg][~culled[w::nwg]].min()) if (~culled[w::nwg]).any() else None for w in range(nwg)]
        wmax = [int(keys[w::nwg][~culled[w::nwg]].max()) if (~culled[w::nwg]).any() else None for w in range(nwg)]
        key_range = robust_range(wmin, wmax)
        _, _, _, sizes_true = bucket_depth_sort(keys, tiles)
        assert max(sizes_true) > CAP, "with the true extremes the outliers push the bulk into oversized segments"
    order, scan, block_first, sizes = bucket_depth_sort(keys, tiles, key_range)
    if case == "outliers":
        assert max(sizes) <= CAP, "the robust range keeps every segment inside the LDS capacity"
    ref = np.argsort(keys, kind="stable")               # (depth key, index): what the LSD sort and the reference's sort leave
    assert (order == ref).all()
    assert (scan == np.cumsum(tiles[ref])).all()
    R = int(tiles.sum())
    if R:
        incl = np.cumsum(tiles[ref])
        excl = incl - tiles[ref]
        nblk = (R + IT - 1) // IT
        for blk in range(nblk):                          # the Gaussian that owns instance blk * IT
            g = int(np.searchsorted(incl, blk * IT, side="right"))
            assert block_first[blk] == (g, int(excl[g])), blk
        last = int(np.nonzero(tiles[ref])[0][-1])
        assert block_first[nblk] == (last, int(excl[last])), "the sentinel entry closes the table in EVERY case"
    if case == "crowd":
        assert max(sizes) > CAP, "this case must exercise the oversized-segment path"
    if case == "uniform":
        assert max(sizes) <= CAP
