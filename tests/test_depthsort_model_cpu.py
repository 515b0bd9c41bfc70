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


EQ_SHIFT, EQ_BINS, EQ_SHIFT2 = 17, 1024, 7      # csrc/gsr_internal.h


def sample_keys(keys):
    """The 4096 keys every ds_hist workgroup histograms (csrc/depthsort.hip): 256 windows of 16 consecutive keys, window q at ((2 q + 1) P / 512)
    rounded down to a multiple of 16; an index past P reads the last key again; tile-less keys do not count."""
    P = len(keys)
    q = np.arange(256, dtype=np.int64)
    start = (((2 * q + 1) * P) >> 9) & ~np.int64(15)
    idx = np.minimum((start[:, None] + np.arange(16)[None, :]).reshape(-1), P - 1)
    k = keys[idx].astype(np.int64)
    return k[k != KEY_CULLED]


class EqTable:
    """ds_hist's bucket tables.  Level 1: (first bucket, buckets) per coarse bin -- every coarse bin inside the frame's true key range gets one
    bucket, the remaining ones of the 2046 are handed out in proportion to the sampled mass.  Level 2, only when one coarse bin holds an eighth
    of the sample or more: that bin's buckets spread over its 1024 sub-bins in proportion to their sampled mass.  Integer arithmetic as in the kernel."""

    def __init__(self, keys):
        listed = keys[keys != KEY_CULLED]
        nb = np.zeros(EQ_BINS, np.int64)
        any_ = listed.size > 0
        if any_:
            tmin, tmax = int(listed.min()), int(listed.max())
            b_lo, b_hi = tmin >> EQ_SHIFT, min(tmax >> EQ_SHIFT, EQ_BINS - 1)
        else:
            b_lo = b_hi = 0
        k = sample_keys(keys) if len(keys) else np.zeros(0, np.int64)
        c = np.bincount(k >> EQ_SHIFT, minlength=EQ_BINS).astype(np.int64)
        fold = np.bincount((k >> EQ_SHIFT2) & (EQ_BINS - 1), minlength=EQ_BINS).astype(np.int64)
        C = int(c.sum())
        nbins = b_hi - b_lo + 1
        spare = NB - 2 - nbins
        assert spare >= 1022
        for b in range(b_lo, b_hi + 1):
            nb[b] = 1 + ((spare * int(c[b])) // C if C else spare // nbins)
        self.start = np.concatenate([[0], np.cumsum(nb)])[:EQ_BINS]
        self.nb = nb
        self.used = int(nb.sum())
        assert self.used <= NB - 2
        H = int(np.argmax(c))
        self.hot = H if (any_ and C >= 256 and int(c[H]) * 8 >= C) else None
        if self.hot is not None:
            bg = (C - int(c[H]) + EQ_BINS - 1) // EQ_BINS      # the other coarse bins' keys, spread flat over the folded sub-bins
            f = np.maximum(fold - bg, 0) + max(1, int(c[H]) >> 10)      # excess over the background + the floor of an evenly filled bin
            F = int(f.sum())
            if F == 0:
                self.hot = None
            else:
                S, NBH = int(self.start[H]), int(nb[H])
                cum = np.concatenate([[0], np.cumsum(f)])
                self.start2 = S + (NBH * cum[:-1]) // F
                self.nb2 = S + (NBH * cum[1:]) // F - self.start2
                assert int(self.start2[-1] + self.nb2[-1]) == S + NBH

    def bucket_of(self, keys):
        k = keys.astype(np.int64)
        b = k >> EQ_SHIFT
        d = self.start[b] + (((k & ((1 << EQ_SHIFT) - 1)) * self.nb[b]) >> EQ_SHIFT)
        if self.hot is not None:
            h = b == self.hot
            j = (k[h] >> EQ_SHIFT2) & (EQ_BINS - 1)
            last = int(self.start[self.hot] + self.nb[self.hot]) - 1
            d2 = np.where(self.nb2[j] > 0, self.start2[j] + (((k[h] & ((1 << EQ_SHIFT2) - 1)) * self.nb2[j]) >> EQ_SHIFT2), np.minimum(self.start2[j], last))
            d = d.copy()
            d[h] = d2
        return d

    def first_key_of_bucket(self, d):
        """smallest key that maps to a bucket >= d (d < buckets in use): the kernel's binary searches + the inverse of the in-bin mapping"""
        lo, hi = 0, EQ_BINS
        while lo < hi:
            mid = (lo + hi) >> 1
            if self.start[mid] > d:
                hi = mid
            else:
                lo = mid + 1
        b = lo - 1
        st, nb = int(self.start[b]), int(self.nb[b])
        if self.hot is not None and b == self.hot:
            last = st + nb - 1
            lo, hi = 0, EQ_BINS - 1
            top = lambda j: int(self.start2[j] + self.nb2[j] - 1) if self.nb2[j] else min(int(self.start2[j]), last)      # noqa: E731
            while lo < hi:
                mid = (lo + hi) >> 1
                if top(mid) >= d:
                    hi = mid
                else:
                    lo = mid + 1
            st2, nb2 = int(self.start2[lo]), int(self.nb2[lo])
            x = (((d - st2) << EQ_SHIFT2) + nb2 - 1) // nb2 if (nb2 and st2 < d) else 0
            return (b << EQ_SHIFT) + (lo << EQ_SHIFT2) + x
        x = (((d - st) << EQ_SHIFT) + nb - 1) // nb if nb else 0
        return (b << EQ_SHIFT) + x


def bucket_depth_sort(keys, tiles):
    """-> (order, inclusive tile scan in depth order, block_first dict, segment sizes)."""
    P = len(keys)
    listed = keys != KEY_CULLED
    assert ((tiles > 0) == listed).all()
    tmin, tmax = (int(keys[listed].min()), int(keys[listed].max())) if listed.any() else (0xFFFFFFFF, 0)
    eq = EqTable(keys)
    used = eq.used
    d = np.where(listed, eq.bucket_of(np.where(listed, keys, 0)), CULL_BUCKET)
    if listed.any():
        dl, kl = d[listed], keys[listed].astype(np.int64)
        o = np.argsort(kl, kind="stable")
        assert (np.diff(dl[o]) >= 0).all(), "the bucket mapping is monotone in the key"
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
    block_first, sizes, covered, spans = {}, [], 0, []
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
        spans.append(None)
        base_key = max(tmin, eq.first_key_of_bucket(d0))
        span = (tmax + 1 if d1 >= used else min(tmax + 1, eq.first_key_of_bucket(d1))) - base_key
        rem = keys[ids].astype(np.int64) - base_key
        assert rem.min() >= 0 and rem.max() < max(span, 1), "rebased keys fit the segment's span"
        nbits = 0 if span <= 1 else int(span - 1).bit_length()
        spans[-1] = (e - b, int(span))
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
    bucket_depth_sort.last_table = eq
    bucket_depth_sort.last_spans = spans
    return order, scan, block_first, sizes


def keys_from_depths(z, culled):
    k = (np.asarray(z, np.float32).view(np.uint32).astype(np.int64) - KEY_BASE).astype(np.uint32)
    k[culled] = KEY_CULLED
    return k


CASES = ["uniform", "ties", "crowd", "gap", "one_key", "all_culled", "single", "last_bucket_straddles_a_window", "outliers", "heavy_tails", "wall", "wall_thin", "full_bin"]


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
    elif case == "heavy_tails":      # round 6: 3 % of the Gaussians far behind / in front of a narrow bulk -- every workgroup of the key-producing kernel
        z = rng.uniform(4.0, 4.4, P)  # has some, so a range estimate from per-workgroup extremes (round 5) spans them all and the bulk crowds a few buckets
        far = rng.random(P) < 0.03
        z[far] = np.exp(rng.uniform(np.log(0.25), np.log(9000.0), int(far.sum())))
    elif case == "wall":             # half of the scene within 0.3 % of one depth (a wall seen head-on), the rest spread over the frustum
        z = np.where(rng.random(P) < 0.5, rng.normal(6.0, 0.006, P), rng.uniform(1.0, 40.0, P))
    elif case == "wall_thin":        # 60 % of the scene within 2e-4 of one depth (~1700 consecutive keys): narrower than one coarse bin's bucket
        z = np.where(rng.random(P) < 0.6, 5.0 * (1.0 + 2e-4 * rng.random(P)), rng.uniform(1.0, 40.0, P))
    elif case == "full_bin":         # a third of the scene EVENLY spread over one coarse bin (a well-filled bin, not a wall): the second level must not concentrate its buckets
        P = 600_000
        z = np.where(rng.random(P) < 0.35, rng.uniform(4.0, 4.0 * (1 + 1.0 / 64), P), rng.uniform(2.0, 40.0, P))
    culled = rng.random(P) < (1.0 if case == "all_culled" else 0.12)
    keys = keys_from_depths(z, culled)
    tiles = np.where(culled, 0, rng.integers(1, 40, P)).astype(np.int64)
    order, scan, block_first, sizes = bucket_depth_sort(keys, tiles)
    if case in ("outliers", "gap", "heavy_tails", "wall", "wall_thin", "full_bin"):
        assert max(sizes) <= CAP, "the equalised buckets keep every segment inside the LDS capacity"
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
        # 3/4 of the keys on 48 consecutive values: the second-level table gives every value buckets of its own; the oversized segments that
        # remain hold EQUAL keys only (span 1: nothing to sort, the stable bucket order is the answer)
        assert bucket_depth_sort.last_table.hot is not None
        assert all(sp <= 1 for n, sp in bucket_depth_sort.last_spans if n > CAP)
        assert max(sizes) <= CAP, "940 ties per value at this size: every value's bucket fits a segment (round 5: one 45 000-key segment)"
    if case == "wall_thin":
        assert bucket_depth_sort.last_table.hot is not None and max(sizes) <= CAP
    if case == "uniform":
        assert max(sizes) <= CAP
