"""Bins of the PRODUCT'S KERNEL SOURCE against the oracle, in the build container: the default forward's binning chain after the per-Gaussian
kernel -- bucket depth sort (csrc/depthsort.hip) + fused emission / two-level tile sort (csrc/tilesort.hip), ten launches -- compiled with g++
against the SIMT shim of tests/simt/ and run lane by lane through the product's own launchers, fed with what the oracle's preprocess computes
(rectangles, tile counts, depths -> 27-bit depth keys).  The sorted point list and the tile ranges must equal `O.bin_and_sort` -- the same
contract the GPU tests hold the library to (tests/test_gpu_bins_sweep.py), here without a GPU.  Test infrastructure: a checker of the kernel
source, not a CPU path of the product (tests/_build/libsimt_chain.so is never shipped)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from helpers import O, make_camera, make_edge_scene, make_scene, oracle_settings, reference_tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libsimt_chain.so")
KEY_BASE, CULLED = 0x3E4CCCCD, (1 << 27) - 1

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


def _build(defines=(), tag=""):
    from simt_build import build
    h = build("chain", defines=defines, tag=tag)
    h.simt_chain_last_error.restype = C.c_char_p
    h.simt_bin.restype = C.c_int64
    return h


@pytest.fixture(scope="module")
def lib():
    return _build()


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


CASES = {
    # name: (scene, P, W, H, s_med, seed, reference rectangles)
    "cloud": ("cloud", 6000, 320, 240, 0.02, 1, False),
    "cloud_reference_rectangles": ("cloud", 4000, 320, 240, 0.02, 2, True),
    "tiny_splats": ("cloud", 30000, 320, 240, 0.002, 3, False),
    "huge_splats": ("cloud", 400, 400, 304, 0.5, 4, False),
    "odd_frame": ("cloud", 5000, 250, 131, 0.03, 5, False),
    "edge_scene": ("edge", 3000, 250, 131, 0.0, 6, False),
    "depth_ties": ("ties", 8000, 320, 240, 0.01, 7, False),
    # round 5: a narrow bulk + floaters far behind / in front of it, with the per-workgroup key ranges of a 274-workgroup projection launch: ds_hist's
    # ROBUST key range is narrower than the true one (the floaters sit in workgroups whose group partner has none), the end buckets hold the floaters
    "depth_outliers": ("outliers", 70000, 320, 240, 0.002, 8, False),
}


@pytest.mark.parametrize("name", list(CASES))
def test_binning_chain_source_on_the_cpu_equals_the_oracle_bins(lib, name):
    import contextlib
    kind, P, W, H, s_med, seed, ref_rects = CASES[name]
    cam = make_camera(W, H)
    sc = make_edge_scene(P, cam, seed=seed) if kind == "edge" else make_scene(P, cam, seed=seed, s_med=s_med)
    if kind == "ties":      # depths quantised to a few values: tie order = Gaussian index
        z = sc.means3D[:, 2].clone()
        zq = (torch.round(z * 4.0) / 4.0).clamp_min(0.5)
        sc.means3D[:, 0] *= zq / z
        sc.means3D[:, 1] *= zq / z
        sc.means3D[:, 2] = zq
    if kind == "outliers":
        g = torch.Generator().manual_seed(seed)
        z = sc.means3D[:, 2].clone()
        zn = 4.0 + 0.2 * torch.rand(P, generator=g)
        far = torch.tensor([3 * 256 + 5, 3 * 256 + 77, 7 * 256 + 1, 260 * 256 + 9, 260 * 256 + 200])      # chunks 3, 7, 260 of 256 Gaussians
        near = torch.tensor([7 * 256 + 100, 260 * 256 + 30])
        zn[far] = torch.tensor([900.0, 350.0, 2000.0, 1200.0, 500.0])
        zn[near] = torch.tensor([0.3, 0.45])
        f = (zn / z).unsqueeze(1)
        sc.means3D.mul_(f)
        sc.scales.mul_(f)
        sc.opacities[far] = 0.9
        sc.opacities[near] = 0.9
    s = oracle_settings(cam)
    with (reference_tiles() if ref_rects else contextlib.nullcontext()), torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        bins = O.bin_and_sort(pre)
    gx, gy = pre["grid"]
    tiles = pre["tiles_touched"].numpy().astype(np.uint32)
    R = int(tiles.sum())
    assert R == int(bins["R"]) and R > 0
    r = pre["rect"].numpy().astype(np.uint32)                       # minx, miny, maxx, maxy
    rect = np.stack([r[:, 0] | (r[:, 2] << 16), r[:, 1] | (r[:, 3] << 16)], axis=1).astype(np.uint32)
    rect[tiles == 0] = 0
    depth_bits = pre["depths"].detach().to(torch.float32).contiguous().numpy().view(np.uint32).astype(np.int64)
    keys = np.where(tiles > 0, depth_bits - KEY_BASE, CULLED)
    assert keys[tiles > 0].min() >= 0 and keys[tiles > 0].max() < CULLED - 1, "a listed depth outside the 27-bit key range (the host then re-keys: not this test)"
    keys = keys.astype(np.uint32)
    wg = np.array([[(~np.uint32(keys[tiles > 0].min())) & np.uint32(0xFFFFFFFF), keys[tiles > 0].max()]], dtype=np.uint32)
    if kind == "outliers":      # what the projection kernel leaves: workgroup w of ceil(P / 256) holds the Gaussians [256 w, 256 w + 256)
        nwg = (P + 255) // 256
        wg = np.zeros((nwg, 2), dtype=np.uint32)
        for w in range(nwg):
            kk = keys[256 * w:256 * w + 256][tiles[256 * w:256 * w + 256] > 0]
            if kk.size:
                wg[w] = ((~np.uint32(kk.min())) & np.uint32(0xFFFFFFFF), kk.max())
        listed = np.sort(keys[tiles > 0].astype(np.int64))
        bulk = listed[len(listed) // 100: -len(listed) // 100]
        assert int(listed.max()) - int(listed.min()) > 8 * (int(bulk.max()) - int(bulk.min())), "the case must have floaters far outside the bulk of the keys"
    order = np.zeros(P, dtype=np.uint32)
    point_list = np.full(R, 0xFFFFFFFF, dtype=np.uint32)
    ranges = np.full((gx * gy, 2), 0xFFFFFFFF, dtype=np.uint32)
    got = lib.simt_bin(P, gx, gy, ptr(keys), ptr(tiles), ptr(rect), ptr(wg), int(wg.shape[0]), R, ptr(order), ptr(point_list), ptr(ranges))
    assert got == R, lib.simt_chain_last_error()
    ref_list = bins["point_list"].numpy().astype(np.uint32)
    if not np.array_equal(point_list, ref_list):
        bad = np.nonzero(point_list != ref_list)[0]
        raise AssertionError(f"sorted point list differs at {bad.size} of {R} positions, first {bad[:8].tolist()}")
    assert np.array_equal(ranges, bins["ranges"].numpy().astype(np.uint32)), "tile ranges differ"
    V = int((tiles > 0).sum())
    assert np.array_equal(order[:V].astype(np.int64), np.argsort(np.where(tiles > 0, depth_bits, 1 << 40), kind="stable")[:V]), "depth order differs"


def _run_rects(lib, w, h, minx, miny, keys, gx, gy):
    P = len(w)
    tiles = (w * h).astype(np.uint32)
    R = int(tiles.sum())
    rect = np.stack([minx | ((minx + w) << 16), miny | ((miny + h) << 16)], axis=1).astype(np.uint32)
    rect[tiles == 0] = 0
    keys = np.where(tiles > 0, keys, CULLED).astype(np.uint32)
    listed = tiles > 0
    wg = np.array([[(~np.uint32(keys[listed].min())) & np.uint32(0xFFFFFFFF), keys[listed].max()]], dtype=np.uint32)
    order = np.zeros(P, dtype=np.uint32)
    pl = np.full(R, 0xFFFFFFFF, dtype=np.uint32)
    rg = np.full((gx * gy, 2), 0xFFFFFFFF, dtype=np.uint32)
    assert lib.simt_bin(P, gx, gy, ptr(keys), ptr(tiles), ptr(rect), ptr(wg), 1, R, ptr(order), ptr(pl), ptr(rg)) == R, lib.simt_chain_last_error()
    it, ii = [], []
    for j in np.argsort(keys.astype(np.int64), kind="stable"):
        if tiles[j]:
            ys, xs = np.meshgrid(np.arange(miny[j], miny[j] + h[j]), np.arange(minx[j], minx[j] + w[j]), indexing="ij")
            it.append((ys * gx + xs).reshape(-1))
            ii.append(np.full(int(tiles[j]), j))
    it, ii = np.concatenate(it), np.concatenate(ii)
    assert np.array_equal(pl, ii[np.argsort(it, kind="stable")].astype(np.uint32)), "sorted point list differs"
    cnt = np.bincount(it, minlength=gx * gy)
    st = np.concatenate([[0], np.cumsum(cnt)])[:-1]
    assert np.array_equal(rg, np.stack([np.where(cnt > 0, st, 0), np.where(cnt > 0, st + cnt, 0)], axis=1).astype(np.uint32)), "tile ranges differ"
    return R


def test_binning_chain_source_at_emission_block_boundaries(lib):
    """R an exact multiple of the 4096-instance emission block (the table entry one past the last block), with and without a tile-less tail,
    one Gaussian that is exactly one block, block boundaries that coincide with Gaussian boundaries."""
    rng = np.random.default_rng(5)
    gx, gy = 120, 68
    for k in (1, 2, 5):
        P = 256 * k                                        # 16 tiles each
        assert _run_rects(lib, np.full(P, 4), np.full(P, 4), rng.integers(0, gx - 4, P), rng.integers(0, gy - 4, P), 0x00400000 + rng.integers(0, 1 << 16, P), gx, gy) == 4096 * k
    P = 512 + 100
    w = np.concatenate([np.full(512, 4), np.zeros(100, dtype=np.int64)])
    assert _run_rects(lib, w, np.full(P, 4), rng.integers(0, gx - 4, P), rng.integers(0, gy - 4, P), 0x00400000 + rng.integers(0, 1 << 16, P), gx, gy) == 8192
    assert _run_rects(lib, np.array([64]), np.array([64]), np.array([3]), np.array([2]), np.array([0x00400000]), gx, gy) == 4096
    assert _run_rects(lib, np.array([64, 1, 64, 2]), np.array([64, 1, 64, 3]), np.array([0, 5, 10, 7]), np.array([0, 1, 2, 3]), 0x00400000 + np.array([5, 6, 7, 8]), gx, gy) == 8199
    # degenerate grids
    assert _run_rects(lib, np.array([1, 1, 1]), np.array([1, 1, 1]), np.zeros(3, dtype=np.int64), np.zeros(3, dtype=np.int64), 0x00400000 + np.array([9, 3, 3]), 1, 1) == 3
    assert _run_rects(lib, np.array([3, 2]), np.array([200, 7]), np.array([0, 1]), np.array([0, 50]), 0x00400000 + np.array([2, 1]), 3, 200) == 614
