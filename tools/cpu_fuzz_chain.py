"""Randomised run of the product's binning chain SOURCE on the CPU (SIMT shim, tests/simt/chain_harness.cpp; build it first:
python -m pytest tests/test_simt_chain_cpu.py) against stable numpy sorts: degenerate grids (1 x 1, 3 x 200), Gaussians that cover the whole frame,
P = 1, 90 % tile-less, tied / narrow / full-range depth keys.

    python tools/cpu_fuzz_chain.py [seed] [seconds]      (round 4: 218 frames in 240 s, no failure)

Test infrastructure, not product code."""
import ctypes as C, numpy as np, os, sys, time
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_build", "libsimt_chain.so")); lib.simt_chain_last_error.restype = C.c_char_p; lib.simt_bin.restype = C.c_int64
ptr = lambda a: a.ctypes.data_as(C.c_void_p)
CULLED = (1 << 27) - 1
def run(rng, P, gx, gy, mode):
    if mode == 'full':   # a few Gaussians covering everything
        w = np.full(P, gx); h = np.full(P, gy); minx = np.zeros(P, int); miny = np.zeros(P, int)
    else:
        mw = {'small': 4, 'mid': 12, 'wide': gx}[mode if mode in ('small','mid','wide') else 'small']
        w = rng.integers(1, min(mw, gx) + 1, P); h = rng.integers(1, min(6, gy) + 1, P)
        minx = rng.integers(0, gx - w + 1); miny = rng.integers(0, gy - h + 1)
    dead = rng.random(P) < rng.choice([0.0, 0.1, 0.9])
    if dead.all(): dead[rng.integers(0, P)] = False
    w = np.where(dead, 0, w)
    tiles = (w * h).astype(np.uint32)
    if mode == 'exact':  # make R an exact multiple of 4096 when possible
        pass
    rect = np.stack([minx | ((minx + w) << 16), miny | ((miny + h) << 16)], axis=1).astype(np.uint32)
    rect[tiles == 0] = 0
    kk = rng.choice(['uniform', 'ties', 'narrow', 'wide'])
    base = 0x00400000
    if kk == 'uniform': keys = base + rng.integers(0, 1 << 20, P)
    elif kk == 'ties': keys = base + rng.integers(0, 5, P) * 1000
    elif kk == 'narrow': keys = base + rng.integers(0, 3, P)
    else: keys = rng.integers(1, CULLED - 1, P)
    keys = np.where(tiles > 0, keys, CULLED).astype(np.uint32)
    R = int(tiles.sum())
    listed = tiles > 0
    wg = np.array([[(~np.uint32(keys[listed].min())) & np.uint32(0xFFFFFFFF), keys[listed].max()]], dtype=np.uint32)
    order = np.zeros(P, dtype=np.uint32); pl = np.full(R, 0xFFFFFFFF, dtype=np.uint32); rg = np.full((gx * gy, 2), 0xFFFFFFFF, dtype=np.uint32)
    got = lib.simt_bin(P, gx, gy, ptr(keys), ptr(tiles), ptr(rect), ptr(wg), 1, R, ptr(order), ptr(pl), ptr(rg))
    if got != R: return f"error {lib.simt_chain_last_error()}"
    # reference
    o = np.argsort(keys.astype(np.int64), kind='stable')
    it, ii = [], []
    for j in o:
        if tiles[j] == 0: continue
        ys, xs = np.meshgrid(np.arange(miny[j], miny[j] + h[j]), np.arange(minx[j], minx[j] + w[j]), indexing='ij')
        t = (ys * gx + xs).reshape(-1); it.append(t); ii.append(np.full(t.size, j))
    it = np.concatenate(it); ii = np.concatenate(ii)
    o2 = np.argsort(it, kind='stable')
    if not np.array_equal(pl, ii[o2].astype(np.uint32)): return "point list differs"
    cnt = np.bincount(it, minlength=gx * gy); st = np.concatenate([[0], np.cumsum(cnt)])[:-1]
    ref = np.stack([np.where(cnt > 0, st, 0), np.where(cnt > 0, st + cnt, 0)], axis=1).astype(np.uint32)
    if not np.array_equal(rg, ref): return "ranges differ"
    return None
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); n = 0; fails = []
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
while time.time() - t0 < budget:
    gx, gy = [(120, 68), (8, 8), (37, 21), (250, 131), (1, 1), (256, 256), (3, 200), (200, 3)][rng.integers(0, 8)]
    mode = rng.choice(['small', 'mid', 'wide', 'full'])
    P = int(rng.choice([1, 2, 3, 17, 300, 1500, 5000]))
    if mode == 'full': P = min(P, 6)
    if mode == 'wide' and P > 1500: P = 1500
    seed_state = rng.bit_generator.state
    r = run(rng, P, gx, gy, mode)
    n += 1
    if r: fails.append((P, gx, gy, mode, r)); print("FAIL", P, gx, gy, mode, r, flush=True)
    if len(fails) > 5: break
print("frames", n, "fails", len(fails), "seconds", round(time.time() - t0, 1))
