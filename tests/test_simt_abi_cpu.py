"""THE WHOLE LIBRARY on the CPU: every translation unit of gaussian-splatting_amd/build.py -- the kernels AND the C-ABI host code of csrc/gsr_api.cpp
(argument checks, scratch carving through the resize callbacks, the R read-back through the leased host word, the speculative binning buffer, the
choice of depth sort / tile sort, the 32-bit re-key of deep frames, the backward's launch plan) -- compiled with g++ against the SIMT shim of
tests/simt/ into tests/_build/libgsr_simt.so, which exports the same C ABI as libgsr_hip.so (include/gsr.h, ABI 4).  The tests call
gsr_rasterize_forward / gsr_rasterize_backward / gsr_forward_views / gsr_set_option exactly as the Python package does on a GPU (numpy buffers stand
in for device memory) and hold the results to the GPU parity suite's scenes and bars against the oracle.

Test infrastructure: a checker of the SOURCE in the build container.  The product has no CPU path -- libgsr_simt.so is built under tests/_build/,
never shipped, and the Python package refuses tensors that are not on a HIP device."""
import contextlib
import ctypes as C
import os
import shutil
import sys

import numpy as np
import pytest
import torch

from helpers import O, make_camera, make_scene, oracle_settings, reference_tiles
import test_gpu_parity as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")

RESIZE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class Views(C.Structure):
    _fields_ = [("splats", C.c_void_p), ("tiles_touched", C.c_void_p), ("depth_order", C.c_void_p), ("point_list", C.c_void_p), ("ranges", C.c_void_p),
                ("final_T", C.c_void_p), ("n_contrib", C.c_void_p), ("tile_scan", C.c_void_p)]


@pytest.fixture(scope="module")
def lib():
    from simt_build import build_library
    h = C.CDLL(build_library())
    h.gsr_last_error.restype = C.c_char_p
    h.gsr_backward_scratch_bytes.restype = C.c_size_t
    h.gsr_backward_scratch_bytes.argtypes = [C.c_int, C.c_int64]
    assert h.gsr_abi_version() == 4
    return h


def f32(t):
    return None if t is None else np.ascontiguousarray(t.detach().to(torch.float32).numpy())


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Buffer:
    """A caller-owned scratch buffer behind a resize callback (diff_gaussian_rasterization._Buffer with numpy instead of device memory)."""

    def __init__(self):
        self.a = np.zeros(16, dtype=np.uint8)
        self.calls = 0

        def resize(_user, nbytes):
            self.calls += 1
            if self.a.size < nbytes:
                self.a = np.zeros(nbytes + 64, dtype=np.uint8)
            return self.a.ctypes.data
        self.cb = RESIZE(resize)

    def view(self, address, nbytes, dtype):
        off = address - self.a.ctypes.data
        assert 0 <= off and off + nbytes <= self.a.size, "view outside its buffer"
        return self.a[off:off + nbytes].view(dtype)


def settings_of(s, keep, tile_rows=None, no_backward=False):
    from diff_gaussian_rasterization._lib import GsrRasterSettings
    keep += [f32(s.bg), f32(s.viewmatrix), f32(s.projmatrix), f32(s.campos)]
    bg, view, proj, campos = keep[-4:]
    y0, y1 = (0, 0) if tile_rows is None else tile_rows
    return GsrRasterSettings(int(s.image_height), int(s.image_width), float(s.tanfovx), float(s.tanfovy), ptr(bg).value, float(s.scale_modifier), ptr(view).value,
                             ptr(proj).value, int(s.sh_degree), ptr(campos).value, 0, 0, 1 if s.antialiasing else 0, int(y0), int(y1), 1 if no_backward else 0, None, None)


def forward(lib, s, sc, colors=None, cov=None, tile_rows=None, no_backward=False, allow_fail=False):
    keep = []
    rs = settings_of(s, keep, tile_rows, no_backward)
    H, W, P = int(s.image_height), int(s.image_width), sc.P
    arr = dict(means3D=f32(sc.means3D), shs=None if colors is not None else f32(sc.shs), colors=f32(colors), opacities=f32(sc.opacities),
               scales=None if cov is not None else f32(sc.scales), rotations=None if cov is not None else f32(sc.rotations), cov=f32(cov))
    M = 0 if colors is not None else sc.shs.shape[1]
    color, invd, radii = np.zeros((3, H, W), np.float32), np.zeros((1, H, W), np.float32), np.zeros(max(P, 1), np.int32)
    geom, binning, img = Buffer(), Buffer(), Buffer()
    nr = C.c_int32(0)
    rc = lib.gsr_rasterize_forward(C.byref(rs), P, M, ptr(arr["means3D"]), ptr(arr["shs"]), ptr(arr["colors"]), ptr(arr["opacities"]), ptr(arr["scales"]),
                                   ptr(arr["rotations"]), ptr(arr["cov"]), geom.cb, None, binning.cb, None, img.cb, None, ptr(color), ptr(invd), ptr(radii),
                                   C.byref(nr), None)
    if allow_fail and rc != 0:
        return {"rc": rc}
    assert rc == 0, lib.gsr_last_error()
    R = int(nr.value)
    out = {"rc": 0, "color": torch.from_numpy(color), "invdepth": torch.from_numpy(invd), "radii": torch.from_numpy(radii[:P]), "R": R,
           "state": (rs, keep, arr, M, geom, binning, img, radii)}
    if P == 0:
        return out
    v = Views()
    assert lib.gsr_forward_views(P, C.c_int64(R), W, H, ptr(geom.a), ptr(binning.a), ptr(img.a), C.byref(v)) == 0, lib.gsr_last_error()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    out["tiles_touched"] = torch.from_numpy(geom.view(v.tiles_touched, P * 4, np.int32).astype(np.int64))
    out["point_list"] = torch.from_numpy(binning.view(v.point_list, R * 4, np.int32).astype(np.int64)) if R else torch.zeros(0, dtype=torch.int64)
    out["ranges"] = torch.from_numpy(img.view(v.ranges, gx * gy * 8, np.int32).reshape(gx * gy, 2).astype(np.int64))
    if not no_backward:
        out["final_T"] = torch.from_numpy(img.view(v.final_T, H * W * 4, np.float32).reshape(H, W).copy())
        out["n_contrib"] = torch.from_numpy(img.view(v.n_contrib, H * W * 4, np.int32).reshape(H, W).astype(np.int64))
    return out


def backward(lib, s, sc, out, dL_dcolor, dL_dinvdepth=None):
    rs, keep, arr, M, geom, binning, img, radii = out["state"]
    P, R = sc.P, out["R"]
    g = dict(means2D=np.zeros((P, 3), np.float32), colors=np.zeros((P, 3), np.float32), opacities=np.zeros((P, 1), np.float32), means3D=np.zeros((P, 3), np.float32),
             cov=np.zeros((P, 6), np.float32), shs=np.zeros((P, max(M, 1), 3), np.float32), scales=np.zeros((P, 3), np.float32), rotations=np.zeros((P, 4), np.float32))
    scratch = np.zeros(lib.gsr_backward_scratch_bytes(P, R) + 64, dtype=np.uint8)
    dcol, dinv = f32(dL_dcolor), f32(dL_dinvdepth)
    rc = lib.gsr_rasterize_backward(C.byref(rs), P, M, R, ptr(arr["means3D"]), ptr(arr["shs"]), ptr(arr["colors"]), ptr(arr["opacities"]), ptr(arr["scales"]),
                                    ptr(arr["rotations"]), ptr(arr["cov"]), ptr(radii), ptr(geom.a), ptr(binning.a), ptr(img.a), ptr(dcol), ptr(dinv),
                                    ptr(g["means2D"]), ptr(g["colors"]) if arr["colors"] is not None else None, ptr(g["opacities"]), ptr(g["means3D"]),
                                    ptr(g["cov"]) if arr["cov"] is not None else None, ptr(g["shs"]) if arr["shs"] is not None else None,
                                    ptr(g["scales"]) if arr["scales"] is not None else None, ptr(g["rotations"]) if arr["rotations"] is not None else None,
                                    ptr(scratch), None, None)
    assert rc == 0, lib.gsr_last_error()
    return g


@contextlib.contextmanager
def option(lib, name, value, default):
    assert lib.gsr_set_option(name, value) == 0, lib.gsr_last_error()
    try:
        yield
    finally:
        lib.gsr_set_option(name, default)


@pytest.mark.parametrize("no_backward", [False, True], ids=["track", "inference"])
@pytest.mark.parametrize("name", ["c1", "odd_aa", "edge_lookat", "deg0_dense"])
def test_c_abi_forward_on_the_cpu_against_the_oracle(lib, name, no_backward):
    cam, sc, opts = G.mk(name)
    s, col, radii, invd, aux = G.run_oracle(cam, sc, opts)
    out = forward(lib, s, sc, no_backward=no_backward)
    G.check_forward(s, col, radii, invd, aux, out)




def test_c_abi_forward_call_forms_band_and_empty_scene(lib):
    cam, sc, opts = G.mk("edge_lookat")
    colors = torch.rand(sc.P, 3, generator=torch.Generator().manual_seed(1))
    cov = O.compute_cov3d(sc.scales, sc.rotations, 1.0, torch.float32)
    s, col, radii, invd, aux = G.run_oracle(cam, sc, opts, colors=colors, cov=cov)
    G.check_forward(s, col, radii, invd, aux, forward(lib, s, sc, colors=colors, cov=cov))
    # a band of tile rows (the multi-GPU extension): rows outside stay untouched
    cam, sc, opts = G.mk("edge_aa_scale")
    band = (5, 11)
    s, col, radii, invd, aux = G.run_oracle(cam, sc, opts, tile_rows=band)
    G.check_forward(s, col, radii, invd, aux, forward(lib, s, sc, tile_rows=band), band=band)
    # P = 0: the reference's zero image, no callback, R = 0
    import copy
    empty = copy.copy(sc)
    empty.means3D, empty.scales, empty.rotations, empty.opacities, empty.shs = sc.means3D[:0], sc.scales[:0], sc.rotations[:0], sc.opacities[:0], sc.shs[:0]
    out = forward(lib, s, empty)
    assert out["R"] == 0 and float(out["color"].abs().max()) == 0.0 and all(b.calls == 0 for b in out["state"][4:7])


@pytest.mark.parametrize("name,n,seed,use_depth", [("c1", 1000, 0, True), ("edge_aa_scale", 300, 3, False)])
def test_c_abi_backward_on_the_cpu_against_the_oracles_autograd(lib, name, n, seed, use_depth):
    import copy
    cam, sc, opts = G.mk(name)
    idx = torch.arange(0, sc.P, max(1, sc.P // n))[:n]                     # (a strided sample keeps every kind of splat the scene has)
    sc = copy.copy(sc)
    sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs = sc.means3D[idx], sc.scales[idx], sc.rotations[idx], sc.opacities[idx], sc.shs[idx]
    s = oracle_settings(cam, bg=opts.get("bg"), sh_degree=opts.get("sh_degree", 3), scale_modifier=opts.get("scale_modifier", 1.0), antialiasing=opts.get("antialiasing", False))
    wc, wd = G._loss_weights(cam.image_height, cam.image_width, seed)
    L = {k: v.detach().clone().requires_grad_(True) for k, v in dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations).items()}
    L["means2D"] = torch.zeros(sc.P, 3, requires_grad=True)
    col, radii, invd = O.rasterize(L["means3D"], L["means2D"], L["opacities"], s, shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
    ((col * wc).sum() + ((invd * wd).sum() if use_depth else 0.0)).backward()
    out = forward(lib, s, sc)
    assert torch.equal(out["radii"], radii.to(torch.int32))
    g = backward(lib, s, sc, out, wc, wd if use_depth else None)
    for k in L:
        a, b = torch.from_numpy(g[k]).double(), L[k].grad.double()
        d = (a - b).abs() / (b.abs().max().item() + 1e-30)
        assert b.abs().max().item() > 0 and d.max().item() < 1e-4, f"{k}: max err {d.max().item():.3e}"
        assert torch.quantile(d.flatten()[:4_000_000], 0.999).item() < 1e-5, f"{k}: 99.9th pct err too large"
    # the launch order of the blend backward is scheduling only: the same bits in all four orders
    for order in ((0, 1, 3) if name == "c1" else ()):      # (huge splats cost minutes per backward on the CPU)
        with option(lib, b"bwd_heavy_first", order, 2):
            g2 = backward(lib, s, sc, out, wc, wd if use_depth else None)
        for k in g:
            assert np.array_equal(g[k], g2[k]), f"launch order {order} changed {k}"


def test_c_abi_host_choices_leave_the_bins_alone(lib):
    """What gsr_api.cpp decides per frame: bucket depth sort or LSD passes, fused two-level tile sort or emission + LSD passes, the level-2 scan as its
    own launch or folded in, snug rectangles or the reference's squares -- every combination against the oracle's bins (the reference's own
    rectangles in reference mode)."""
    cam = make_camera(320, 240)
    sc = make_scene(6000, cam, seed=21, s_med=0.02)
    s = oracle_settings(cam)
    with torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        bins = O.bin_and_sort(pre)
        with reference_tiles():
            pre_r = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
            bins_r = O.bin_and_sort(pre_r)
    assert int(bins_r["R"]) > int(bins["R"])
    base = forward(lib, s, sc, no_backward=True)
    for opts in ({b"depth_sort_mode": (1, 0)}, {b"depth_sort_mode": (2, 0), b"level2_scan_mode": (1, 0)}, {b"tile_sort_mode": (1, 0)}, {b"level2_scan_mode": (2, 0)},
                 {b"snug_tiles": (0, 1)}, {b"snug_tiles": (0, 1), b"depth_sort_mode": (1, 0), b"tile_sort_mode": (1, 0)}):
        with contextlib.ExitStack() as st:
            for k, (v, d) in opts.items():
                st.enter_context(option(lib, k, v, d))
            out = forward(lib, s, sc, no_backward=True)
        want_pre, want = (pre_r, bins_r) if b"snug_tiles" in opts else (pre, bins)
        assert out["R"] == int(want["R"]) and torch.equal(out["tiles_touched"], want_pre["tiles_touched"]), opts
        assert torch.equal(out["point_list"], want["point_list"]) and torch.equal(out["ranges"], want["ranges"]), opts
        assert torch.equal(out["color"], base["color"]), f"{opts}: the image changed"


def test_c_abi_a_frame_counter_that_is_not_zero_costs_one_frame(lib):
    """csrc/gsr_frame.h: R is summed in a device counter that must be zero when a frame begins.  Round 6 met one that was not (a slot created by a second
    concurrent caller; gsr_api.cpp lease_host_word): such a counter publishes a partial R early and is left non-zero by the workgroups behind -- every later
    frame on that slot was wrong.  The test hook preloads tickets into the NEXT frame's counter:
    (a) one stray ticket: that frame may be wrong (never a crash), every frame after it is right -- the slot alternates between two counters and the
        last workgroup clears both;
    (b) more tickets than workgroups: nobody draws the last ticket, the host waits its 2 s, finds the stream idle and the frame words unpublished,
        clears the slot and REFUSES the frame; the next call is right."""
    cam = make_camera(320, 240)
    sc = make_scene(6000, cam, seed=21, s_med=0.02)
    s = oracle_settings(cam)
    base = forward(lib, s, sc, no_backward=True)
    assert base["R"] > 0

    def same(out):
        return out["rc"] == 0 and out["R"] == base["R"] and torch.equal(out["color"], base["color"]) and torch.equal(out["point_list"], base["point_list"])
    other = make_scene(9000, cam, seed=22, s_med=0.03)      # (the frame that meets the stray ticket is of ANOTHER scene: with the same scene every time the
    assert lib.gsr_set_option(b"debug_dirty_control_block", 1) == 0      # leftover of one frame is exactly what the next one is missing)
    forward(lib, s, other, no_backward=True, allow_fail=True)
    for i in range(3):
        assert same(forward(lib, s, sc, no_backward=True)), f"frame {i + 1} after a stray ticket"
    assert lib.gsr_set_option(b"debug_dirty_control_block", 1500) == 0
    out = forward(lib, s, sc, no_backward=True, allow_fail=True)
    assert out["rc"] != 0 and b"never published" in lib.gsr_last_error(), (out["rc"], lib.gsr_last_error())
    for i in range(2):
        assert same(forward(lib, s, sc, no_backward=True)), f"frame {i + 1} after a refused frame"


def test_c_abi_deep_frame_takes_the_32_bit_rekey(lib):
    """Depths beyond 0.2 * 2^16: the 27-bit depth key of a listed Gaussian overflows, the per-Gaussian kernel reports it through the host word and
    gsr_api.cpp repeats the depth sort with full 32-bit keys (four LSD passes) and the tile scan -- bins against the oracle."""
    cam = make_camera(320, 240)
    sc = make_scene(3000, cam, seed=33, s_med=0.02)
    k = 2500.0                                         # z in [5 000, 30 000]: beyond 13 107
    sc.means3D = sc.means3D * k
    sc.scales = sc.scales * k
    s = oracle_settings(cam)
    with torch.no_grad():
        pre = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        bins = O.bin_and_sort(pre)
    assert float(pre["depths"][pre["tiles_touched"] > 0].max()) > 13107.2 and int(bins["R"]) > 0
    out = forward(lib, s, sc)
    assert out["R"] == int(bins["R"]) and torch.equal(out["point_list"], bins["point_list"]) and torch.equal(out["ranges"], bins["ranges"])


def test_c_abi_argument_checks_on_the_cpu(lib):
    cam, sc, opts = G.mk("c1")
    s = oracle_settings(cam)
    keep = []
    rs = settings_of(s, keep)
    color = np.zeros((3, 256, 256), np.float32)
    nr = C.c_int32(0)
    b = Buffer()
    m, sh, op, sl, rt = f32(sc.means3D), f32(sc.shs), f32(sc.opacities), f32(sc.scales), f32(sc.rotations)
    # neither shs nor colors_precomp: refused
    rc = lib.gsr_rasterize_forward(C.byref(rs), sc.P, 16, ptr(m), None, None, ptr(op), ptr(sl), ptr(rt), None, b.cb, None, b.cb, None, b.cb, None, ptr(color), None,
                                   ptr(np.zeros(sc.P, np.int32)), C.byref(nr), None)
    assert rc == -1, lib.gsr_last_error()
    rc = lib.gsr_rasterize_forward(C.byref(rs), sc.P, 16, ptr(m), ptr(sh), None, ptr(op), ptr(sl), ptr(rt), None, None, None, None, None, None, None, ptr(color), None,
                                   ptr(np.zeros(sc.P, np.int32)), C.byref(nr), None)
    assert rc == -1 and b"resize callbacks" in lib.gsr_last_error()
    assert lib.gsr_set_option(b"depth_sort_mode", 3) == -1 and lib.gsr_set_option(b"no_such_option", 1) == -1
