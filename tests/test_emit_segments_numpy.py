"""The level-1 scatter that ranks ROW PIECES instead of instances (csrc/ab/emit_scatter_segments.inc, measurement build only, not yet run on
a GPU), restated step by step in numpy -- same piece enumeration (rows cut at bucket boundaries and into chunks of <= 31), same exclusive
scan, same per-wave rounds with the WEIGHTED rank taken bit plane by bit plane from ballots, same (wave, bucket) starts, same slot marks +
max-scan, same expansion -- and compared with a plain stable sort of the block's instances by bucket, which is what the shipped
instance-wise kernel (csrc/tilesort.hip emit_scatter) produces.  This pins the ALGORITHM; the HIP code still has to be brought up against
tests/test_gpu_bins_sweep.py on a GPU."""
import zlib

import numpy as np
import pytest

TS_ITEMS, TS_NGCAP, SEG_PCAP, SEG_MAXLEN, SEG_LEN_BITS, WAVES = 4096, 1024, 3072, 31, 5, 4


def pieces_of(excl, incl, minx, wd, miny, gx, lb, b0, b1):
    """seg_part + seg_for_each_piece: [(first tile id, length)] of the Gaussian's instances inside the block, in emission order."""
    lo, hi = max(excl, b0), min(incl, b1)
    if hi <= lo or wd == 0:
        return []
    ra, xa = divmod(lo - excl, wd)
    rb, xl = divmod(hi - 1 - excl, wd)
    out = []
    for r in range(ra, rb + 1):
        x0 = xa if r == ra else 0
        x1 = xl + 1 if r == rb else wd
        t0 = (miny + r) * gx + minx + x0
        t1 = t0 + (x1 - x0)
        while t0 < t1:
            bend = ((t0 >> lb) + 1) << lb
            tend = min(t1, bend, t0 + SEG_MAXLEN)
            out.append((t0, tend - t0))
            t0 = tend
    return out


def ballot(flags):
    m = 0
    for l, f in enumerate(flags):
        if f:
            m |= 1 << l
    return m


def seg_block(b, R, gx, lb, hb, rects, excl_all, incl_all, ids, j_lo, j_hi, global_base):
    """One workgroup of emit_scatter_seg.  rects[j] = (minx, maxx, miny); global_base[d] = bbase + my_hist of the block.  Returns
    {global position: packed word} of the block's instances, or None when the block takes the instance-wise path."""
    b0 = b * TS_ITEMS
    b1 = min(R, b0 + TS_ITEMS)
    nvalid = b1 - b0
    nG = j_hi - j_lo + 1
    lomask = (1 << lb) - 1
    if nG > TS_NGCAP:
        return None
    # 2: pieces per Gaussian, exclusive scan
    per = []
    for i in range(nG):
        j = j_lo + i
        minx, maxx, miny = rects[j]
        per.append(pieces_of(excl_all[j], incl_all[j], minx, maxx - minx, miny, gx, lb, b0, b1))
    cnt = [len(p) for p in per]
    pstart = np.concatenate([[0], np.cumsum(cnt)]).astype(int)
    NP = int(pstart[-1])
    if NP > SEG_PCAP:
        return None
    # 3: piece records in emission order
    pw, pd, plen = [0] * NP, [0] * NP, [0] * NP
    for i in range(nG):
        q = pstart[i]
        for t0, ln in per[i]:
            assert 1 <= ln <= SEG_MAXLEN
            pw[q] = (ids[j_lo + i] << lb) | (t0 & lomask)
            pd[q] = t0 >> lb
            plen[q] = ln
            q += 1
    # 4: weighted stable rank per wave and round
    nb1 = 1 << hb
    wave_cnt = np.zeros((WAVES, nb1), dtype=np.int64)
    rank = [0] * NP
    npw = ((NP + WAVES * 64 - 1) // (WAVES * 64)) * 64
    rounds = npw // 64
    assert WAVES * npw >= NP
    for w in range(WAVES):
        for r in range(rounds):
            qs = [w * npw + r * 64 + lane for lane in range(64)]
            valid = [q < NP for q in qs]
            d = [pd[q] if v else 0 for q, v in zip(qs, valid)]
            ln = [plen[q] if v else 0 for q, v in zip(qs, valid)]
            vmask = ballot(valid)
            planes = [ballot([v and ((x >> bit) & 1) for v, x in zip(valid, ln)]) for bit in range(SEG_LEN_BITS)]
            updates = []
            for lane in range(64):
                if not valid[lane]:
                    continue
                mask = vmask                                   # match_digit
                for bit in range(hb):
                    bal = ballot([(x >> bit) & 1 for x in d])
                    mask &= bal if (d[lane] >> bit) & 1 else ~bal
                lt = (1 << lane) - 1
                below = sum(bin(p & mask & lt).count("1") << bit for bit, p in enumerate(planes))
                total = sum(bin(p & mask).count("1") << bit for bit, p in enumerate(planes))
                prior = int(wave_cnt[w][d[lane]])
                rank[qs[lane]] = prior + below
                if mask & lt == 0:
                    updates.append((d[lane], prior + total))
            for dd, v in updates:                              # (one leader per bucket and round)
                wave_cnt[w][dd] = v
    # per-bucket totals over the waves -> local start of every (wave, bucket) run
    tot = wave_cnt.sum(axis=0)
    lbase = np.concatenate([[0], np.cumsum(tot)])[:-1]
    start = np.zeros_like(wave_cnt)
    for dd in range(nb1):
        run = lbase[dd]
        for k in range(WAVES):
            start[k][dd] = run
            run += wave_cnt[k][dd]
    digit_base = [int(global_base[dd]) - int(lbase[dd]) for dd in range(nb1)]
    # 5: slot marks + max-scan
    mark = np.zeros(TS_ITEMS, dtype=np.int64)
    qat = np.zeros(TS_ITEMS, dtype=np.int64)
    for q in range(NP):
        w = q // npw
        lp = int(start[w][pd[q]]) + rank[q]
        assert mark[lp] == 0, "two pieces claim one slot"
        mark[lp] = lp + 1
        qat[lp] = q
    mark = np.maximum.accumulate(mark)
    # 6: expansion
    out = {}
    for i in range(nvalid):
        first = int(mark[i]) - 1
        assert first >= 0
        q = int(qat[first])
        assert i - first < plen[q]
        out[digit_base[pd[q]] + i] = pw[q] + (i - first)
    return out


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


@pytest.mark.parametrize("kind,P,gx,gy", [("mixed", 3000, 120, 68), ("small", 6000, 120, 68), ("wide", 900, 120, 68), ("huge", 700, 120, 68),
                                          ("ones", 9000, 37, 21), ("columns", 4000, 120, 68), ("mixed", 2500, 250, 131), ("small", 5, 8, 8)])
def test_piece_ranking_restatement_equals_a_stable_sort_by_bucket(kind, P, gx, gy):
    rng = np.random.default_rng(zlib.crc32(f"{kind}-{P}-{gx}-{gy}".encode()))
    rects4 = make_frame(rng, P, gx, gy, kind)
    n_tiles = gx * gy
    nbits = 1
    while (1 << nbits) < n_tiles:
        nbits += 1
    lb = (nbits + 1) // 2
    hb = nbits - lb
    tiles = np.array([(r[1] - r[0]) * (r[3] - r[2]) for r in rects4], dtype=np.int64)
    incl_all = np.cumsum(tiles)
    excl_all = incl_all - tiles
    R = int(incl_all[-1])
    ids = rng.permutation(P).astype(np.int64)                  # `order`: the Gaussian id at depth rank j
    rects = [(r[0], r[1], r[2]) for r in rects4]
    # reference: every instance in emission order, stable sort by bucket
    inst_tile, inst_id = [], []
    for j, (minx, maxx, miny, maxy) in enumerate(rects4):
        for y in range(miny, maxy):
            for x in range(minx, maxx):
                inst_tile.append(y * gx + x)
                inst_id.append(ids[j])
    inst_tile, inst_id = np.array(inst_tile, dtype=np.int64), np.array(inst_id, dtype=np.int64)
    assert len(inst_tile) == R
    bucket = inst_tile >> lb
    order = np.argsort(bucket, kind="stable")
    ref_words = ((inst_id << lb) | (inst_tile & ((1 << lb) - 1)))[order]
    # what the kernels get: per-block bucket histograms, bucket totals
    nblk = (R + TS_ITEMS - 1) // TS_ITEMS
    nb1 = 1 << hb
    hist = np.zeros((nb1, nblk), dtype=np.int64)
    np.add.at(hist, (bucket, np.arange(R) // TS_ITEMS), 1)
    total = hist.sum(axis=1)
    bbase = np.concatenate([[0], np.cumsum(total)])[:-1]
    before = np.cumsum(hist, axis=1) - hist                     # instances of earlier blocks, per bucket
    got = np.full(R, -1, dtype=np.int64)
    n_piece_blocks = 0
    for b in range(nblk):
        b0 = b * TS_ITEMS
        j_lo = int(np.searchsorted(incl_all, b0, side="right"))                       # the Gaussian that holds instance b0
        nxt = (b + 1) * TS_ITEMS
        j_hi = int(np.searchsorted(incl_all, nxt, side="right")) if nxt < R else int(np.nonzero(tiles)[0][-1])
        res = seg_block(b, R, gx, lb, hb, rects, excl_all, incl_all, ids, j_lo, j_hi, bbase + before[:, b])
        if res is None:                                                                # the instance-wise path: reference behaviour by definition
            sel = np.arange(b0, min(R, b0 + TS_ITEMS))
            o = np.argsort(bucket[sel], kind="stable")
            d_sorted = bucket[sel][o]
            lstart = {}
            for i, dd in enumerate(d_sorted):
                lstart.setdefault(int(dd), i)
            for i, (k, dd) in enumerate(zip(sel[o], d_sorted)):
                got[bbase[dd] + before[dd, b] + (i - lstart[int(dd)])] = (inst_id[k] << lb) | (inst_tile[k] & ((1 << lb) - 1))
        else:
            n_piece_blocks += 1
            for pos, word in res.items():
                assert got[pos] == -1, "two instances at one position"
                got[pos] = word
    assert np.array_equal(got, ref_words)
    if kind not in ("ones", "columns"):
        assert n_piece_blocks > 0, "the piece path was never taken"
    else:
        assert n_piece_blocks < nblk, "a block of > 1024 Gaussians / > 3072 pieces must take the instance-wise path"
