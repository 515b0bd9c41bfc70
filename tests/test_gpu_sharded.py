"""GPU tests of round 3's hot-path changes that the oracle-parity suites do not single out:
  * mode C of the multi-GPU design (Gaussian-sharded forward with the destination-targeted exchange): the route kernels
    against their torch restatement, the single-rank renderer against the fused operator, and the per-rank pieces driven
    by hand for 3 shards x 3 bands on one GPU (collectives replaced by concatenations);
  * the 27-bit depth keys: a scene deeper than 0.2 * 2^16 must take the 32-bit fallback sort and still give the oracle's bins.
All calls go through the C ABI."""
import ctypes as C

import pytest
import torch

from helpers import O, make_camera, make_scene, make_edge_scene, oracle_settings
from test_gpu_parity import gpu_settings
from test_gpu_next_rows import _fused_reference

pytestmark = pytest.mark.gpu


def _unpack_rect(bits_x, bits_y):
    bx, by = bits_x.view(torch.int32), bits_y.view(torch.int32)
    return bx & 0xFFFF, (bx >> 16) & 0xFFFF, by & 0xFFFF, (by >> 16) & 0xFFFF


@pytest.mark.parametrize("P,bounds", [(5001, [0, 5, 13]), (20000, [0, 2, 2, 9, 13]), (700, [0, 13]), (4097, [0, 1, 2, 3, 4, 5, 6, 7, 13])])
def test_route_kernels_match_torch_restatement(P, bounds):
    """gsr_route_count / gsr_route_pack == parallel.route_plan_torch: same counts, same order (stable, ascending Gaussian index
    inside a band), packed rows carry the record's first ten floats and the full-frame rectangle bit for bit."""
    from diff_gaussian_rasterization.parallel import (hip_preprocess_shard, hip_route_count, hip_route_pack, route_plan_torch)
    dev = torch.device("cuda:0")
    cam = make_camera(336, 200)
    sc = make_scene(P, cam, seed=5, s_med=0.04).to(dev)
    rs = gpu_settings(oracle_settings(cam), dev)
    records, radii, M, _ = hip_preprocess_shard(rs, sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)
    counts, scratch = hip_route_count(records, bounds)
    minx, maxx, miny, maxy = _unpack_rect(records[:, 12].contiguous(), records[:, 13].contiguous())
    tiles = records[:, 15].contiguous().view(torch.int32)
    want_idx, want_counts = route_plan_torch(miny.cpu().long(), maxy.cpu().long(), tiles.cpu().long(), bounds)
    assert counts.cpu().tolist() == want_counts
    packed, send_ids, offsets = hip_route_pack(records, bounds, want_counts, scratch)
    torch.cuda.synchronize()
    assert offsets[-1] == want_idx.numel()
    assert torch.equal(send_ids.cpu().long(), want_idx)
    src = records[want_idx.to(dev)]
    assert torch.equal(packed[:, :10].view(torch.int32), src[:, :10].contiguous().view(torch.int32))
    assert torch.equal(packed[:, 10:12].contiguous().view(torch.int32), src[:, 12:14].contiguous().view(torch.int32))
    if len(bounds) > 2:
        assert sum(want_counts) > int((tiles > 0).sum())          # some splats straddle a band boundary and travel twice


def test_gaussian_sharded_single_rank_equals_fused_operator():
    """render_gaussian_sharded with one rank (no collective): preprocess shard -> route (one band) -> packed records ->
    gsr_rasterize_from_packed -> blend backward -> gsr_route_return -> per-Gaussian backward must reproduce the fused
    operator: image / radii / inverse depth bit for bit (tau and 1/depth are recomputed from the packed record), gradients to
    rounding."""
    from diff_gaussian_rasterization.parallel import BandPlan, render_gaussian_sharded
    dev = torch.device("cuda:0")
    cam = make_camera(320, 208)
    sc = make_scene(6000, cam, seed=31, s_med=0.03).to(dev)
    rs = gpu_settings(oracle_settings(cam, bg=torch.tensor([0.1, 0.3, 0.2])), dev)
    g = torch.Generator().manual_seed(3)
    wgt, wd = torch.randn(3, 208, 320, generator=g).to(dev), torch.randn(1, 208, 320, generator=g).to(dev)
    col, radii, invd, ref = _fused_reference(rs, sc, wgt, wd, dev)
    L = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
    c2, r2, d2 = render_gaussian_sharded(rs, L[0], L[1], L[2], L[3], L[4], BandPlan.uniform(13, 1), means2D=m2)
    assert torch.equal(c2, col) and torch.equal(r2, radii) and torch.equal(d2, invd)
    ((c2 * wgt).sum() + (d2 * wd).sum()).backward()
    for got, want in zip([t.grad for t in L] + [m2.grad], ref):
        assert (got - want).abs().max().item() <= 5e-5 * want.abs().max().item()
    # colour-only loss: the blend backward runs its build without the 1/depth terms (no gradient arrives for output 3)
    L2 = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    c3, _, _ = render_gaussian_sharded(rs, L2[0], L2[1], L2[2], L2[3], L2[4], BandPlan.uniform(13, 1), gather_invdepth=False)
    (c3 * wgt).sum().backward()
    from diff_gaussian_rasterization import rasterize_gaussians
    L3 = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    c4, _, _ = rasterize_gaussians(L3[0], None, L3[1], None, L3[2], L3[3], L3[4], None, rs)
    (c4 * wgt).sum().backward()
    for a, b in zip(L2, L3):
        assert (a.grad - b.grad).abs().max().item() <= 5e-5 * b.grad.abs().max().item()


def test_gaussian_sharded_pieces_three_shards_three_bands_on_one_gpu():
    """The per-rank pieces of mode C driven by hand for 3 Gaussian shards x 3 pixel bands on one GPU (the all-to-all replaced
    by slicing + concatenation in rank order): every band renders bit-identically to the fused operator, the bins of a band
    are the fused operator's bins of its tiles, and the gradient rows routed back give each shard its parameter gradients."""
    from diff_gaussian_rasterization import _lib, _make_settings, _ptr, _stream_ptr
    from diff_gaussian_rasterization.parallel import (hip_preprocess_shard, hip_route_count, hip_route_pack, hip_render_packed,
                                                     _i64_array)
    lib = _lib.load()
    dev = torch.device("cuda:0")
    W, H = 336, 200
    cam = make_camera(W, H)
    sc = make_edge_scene(6001, cam, seed=17).to(dev)
    rs = gpu_settings(oracle_settings(cam, bg=torch.tensor([0.3, 0.1, 0.2])), dev)
    g = torch.Generator().manual_seed(9)
    wgt, wd = torch.randn(3, H, W, generator=g).to(dev), torch.randn(1, H, W, generator=g).to(dev)
    col, radii, invd, ref = _fused_reference(rs, sc, wgt, wd, dev)
    cuts, bounds = [0, 1900, 1900 + 2500, 6001], [0, 4, 9, 13]
    G = 3
    st = _stream_ptr(dev)
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        t = [x[a:b].contiguous() for x in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
        records, rad, M, _ = hip_preprocess_shard(rs, *t)
        assert torch.equal(rad, radii[a:b])
        counts, scratch = hip_route_count(records, bounds)
        send_counts = counts.cpu().tolist()
        packed, send_ids, offsets = hip_route_pack(records, bounds, send_counts, scratch)
        shards.append(dict(t=t, rad=rad, packed=packed, send_ids=send_ids, offsets=offsets, counts=send_counts, a=a, b=b))
    image = torch.zeros(3, H, W, device=dev)
    depth = torch.zeros(1, H, W, device=dev)
    returned = [[None] * G for _ in range(G)]           # returned[src][band] = gradient rows of the segment src sent to band
    for band in range(G):
        segs = [sh["packed"][sh["offsets"][band]:sh["offsets"][band + 1]] for sh in shards]      # = the all-to-all, rank order
        recv = torch.cat(segs, dim=0).contiguous()
        color, invdp, (geom, binning, img, nr) = hip_render_packed(rs, (bounds[band], bounds[band + 1]), recv, False)
        rows = slice(bounds[band] * 16, min(bounds[band + 1] * 16, H))
        image[:, rows], depth[:, rows] = color[:, rows], invdp[:, rows]
        gc, gd = torch.zeros_like(wgt), torch.zeros_like(wd)
        gc[:, rows], gd[:, rows] = wgt[:, rows], wd[:, rows]
        keep = []
        s = _make_settings(rs, keep, (bounds[band], bounds[band + 1]))
        P_recv = int(recv.shape[0])
        scr = torch.empty(int(lib.gsr_backward_scratch_bytes(P_recv, nr)), dtype=torch.uint8, device=dev)
        rp = C.c_void_p(0)
        _lib.check(lib.gsr_backward_blend(C.byref(s), P_recv, nr, _ptr(geom), _ptr(binning), _ptr(img), _ptr(gc), _ptr(gd),
                                          _ptr(scr), C.byref(rp), st), "gsr_backward_blend")
        off = int(rp.value) - scr.data_ptr()
        full = scr[off:off + P_recv * 48].view(torch.float32).view(P_recv, 12).clone()
        pos = 0
        for src in range(G):
            n = shards[src]["counts"][band]
            returned[src][band] = full[pos:pos + n]
            pos += n
    torch.cuda.synchronize()
    assert torch.equal(image, col) and torch.equal(depth, invd)
    f = dict(dtype=torch.float32, device=dev)
    for src, sh in enumerate(shards):
        P = sh["b"] - sh["a"]
        back = torch.cat(returned[src], dim=0).contiguous()
        mine = torch.empty(P, 12, **f)
        _lib.check(lib.gsr_route_return(P, G, _i64_array(sh["offsets"]), _ptr(sh["send_ids"]), _ptr(back), _ptr(mine), st),
                   "gsr_route_return")
        want = torch.zeros(P, 12, **f)
        for band in range(G):                                   # torch restatement: index_add per segment
            ids = sh["send_ids"][sh["offsets"][band]:sh["offsets"][band + 1]].long()
            want.index_add_(0, ids, returned[src][band])
        assert torch.equal(mine, want)                          # at most one row per (Gaussian, band): same association order
        outs = [torch.empty(P, 3, **f), torch.empty(P, 3, **f), torch.empty(P, 1, **f), torch.empty(P, 3, **f), torch.empty(P, 6, **f),
                torch.empty(P, 16, 3, **f), torch.empty(P, 3, **f), torch.empty(P, 4, **f)]
        keep = []
        s = _make_settings(rs, keep, None)
        t = sh["t"]
        _lib.check(lib.gsr_backward_preprocess(C.byref(s), P, 16, _ptr(t[0]), _ptr(t[1]), None, _ptr(t[2]), _ptr(t[3]), _ptr(t[4]), None,
                                               _ptr(sh["rad"]), None, _ptr(mine), *[_ptr(o) for o in outs], st), "gsr_backward_preprocess")
        torch.cuda.synchronize()
        d_m2, _, d_op, d_m3, _, d_sh, d_sc, d_rot = outs
        a, b = sh["a"], sh["b"]
        for got, w in zip([d_m3, d_sh, d_op, d_sc, d_rot, d_m2], ref):
            assert (got.view(-1) - w[a:b].reshape(-1)).abs().max().item() <= 5e-5 * w.abs().max().item()


def test_depth_keys_beyond_27_bits_take_the_32_bit_fallback():
    """A scene scaled so that listed Gaussians lie deeper than 0.2 * 2^16 = 13 107: the 27-bit depth key overflows, the library
    must notice (mapped host word), repeat the depth sort on the full 32-bit keys and still deliver the oracle's bins bit for
    bit -- and the backward, which reads the depth order from a fixed buffer, must agree with the oracle's gradients."""
    from diff_gaussian_rasterization.debug import forward_with_views
    from diff_gaussian_rasterization import rasterize_gaussians
    dev = torch.device("cuda:0")
    cam = make_camera(256, 192)
    sc = make_scene(3000, cam, seed=11, s_med=0.03)
    k = 3000.0                                           # depths 6 000 .. 36 000: about half of them beyond the 27-bit range
    means = sc.means3D * k
    scales = sc.scales * k
    s = oracle_settings(cam)
    pre = O.preprocess(means, sc.opacities, s, shs=sc.shs, scales=scales, rotations=sc.rotations)
    assert float(pre["depths"][pre["tiles_touched"] > 0].max()) > 13107.2 > float(pre["depths"][pre["tiles_touched"] > 0].min())
    bins = O.bin_and_sort(pre)
    rs = gpu_settings(s, dev)
    for no_backward in (True, False):
        v = forward_with_views(rs, means.to(dev), sc.opacities.to(dev), shs=sc.shs.to(dev), scales=scales.to(dev),
                               rotations=sc.rotations.to(dev), no_backward=no_backward)
        assert v["R"] == bins["R"]
        assert torch.equal(v["radii"].cpu(), pre["radii"])
        assert torch.equal(v["ranges"].cpu().long(), bins["ranges"])
        assert torch.equal(v["point_list"].cpu().long(), bins["point_list"])
    # and a scene inside the range right afterwards takes the 3-pass sort again (the flag is per call)
    pre2 = O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    bins2 = O.bin_and_sort(pre2)
    v2 = forward_with_views(rs, sc.means3D.to(dev), sc.opacities.to(dev), shs=sc.shs.to(dev), scales=sc.scales.to(dev),
                            rotations=sc.rotations.to(dev))
    assert torch.equal(v2["point_list"].cpu().long(), bins2["point_list"])
    # gradients through the fallback path
    wc = torch.randn(3, 192, 256, generator=torch.Generator().manual_seed(2))
    Lc = [t.clone().requires_grad_(True) for t in (means, sc.shs, sc.opacities, scales, sc.rotations)]
    col, _, _ = O.rasterize(Lc[0], None, Lc[2], s, shs=Lc[1], scales=Lc[3], rotations=Lc[4])
    (col * wc).sum().backward()
    Lg = [t.clone().to(dev).requires_grad_(True) for t in (means, sc.shs, sc.opacities, scales, sc.rotations)]
    gcol, _, _ = rasterize_gaussians(Lg[0], None, Lg[1], None, Lg[2], Lg[3], Lg[4], None, rs)
    (gcol * wc.to(dev)).sum().backward()
    for a, b in zip(Lg, Lc):
        assert (a.grad.cpu() - b.grad).abs().max().item() <= 2e-3 * b.grad.abs().max().item()


@pytest.mark.parametrize("no_backward", [True, False])
def test_band_that_receives_no_record_shows_the_background(no_backward):
    """a rank whose band receives no record while the scene is not empty must render the BACKGROUND in its
    band -- what the single-GPU blend writes where no splat lands (T = 1) -- not the zero image of the reference's P == 0
    early-out; rows outside the band stay zero."""
    from diff_gaussian_rasterization.parallel import hip_render_packed
    dev = torch.device("cuda:0")
    cam = make_camera(200, 120)                      # 13 x 8 tiles, last tile row partial
    bg = torch.tensor([0.25, 0.5, 0.75])
    rs = gpu_settings(oracle_settings(cam, bg=bg), dev)
    recv = torch.empty(0, 12, dtype=torch.float32, device=dev)
    color, invdepth, (geom, binning, img, nr) = hip_render_packed(rs, (2, 5), recv, no_backward)
    torch.cuda.synchronize()
    assert nr == 0
    want = torch.zeros(3, 120, 200)
    want[:, 32:80, :] = bg.view(3, 1, 1)
    assert torch.equal(color.cpu(), want)
    assert torch.equal(invdepth.cpu(), torch.zeros(1, 120, 200))
    # the last band reaches the partial tile row at the bottom of the frame
    color, _, _ = hip_render_packed(rs, (5, 8), recv, no_backward)
    want = torch.zeros(3, 120, 200)
    want[:, 80:, :] = bg.view(3, 1, 1)
    assert torch.equal(color.cpu(), want)


# ------------------------------------------------------------------------------------------------------------------
# fixed-capacity form of the exchange (round 4): segments with the count in a header row
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("P,bounds,capacity", [(5001, [0, 5, 13], 4096), (5001, [0, 5, 13], 700), (20000, [0, 2, 2, 9, 13], 8192),
                                               (4097, [0, 1, 2, 3, 4, 5, 6, 7, 13], 300)])
def test_fixed_capacity_pack_matches_the_exact_pack(P, bounds, capacity):
    """gsr_route_pack_fixed lays the records of band b into rows b*(capacity+1)+1.. of the segment buffer: the header row carries
    (count, capacity) as integer bits, the first min(count, capacity) records are those of gsr_route_pack bit for bit in the same
    order, the send ids of header rows and unused rows are -1 (also when a band overflows the capacity)."""
    from diff_gaussian_rasterization.parallel import hip_preprocess_shard, hip_route_count, hip_route_pack, hip_route_pack_fixed
    dev = torch.device("cuda:0")
    cam = make_camera(336, 200)
    sc = make_scene(P, cam, seed=5, s_med=0.04).to(dev)
    rs = gpu_settings(oracle_settings(cam), dev)
    records, radii, M, _ = hip_preprocess_shard(rs, sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)
    counts, scratch = hip_route_count(records, bounds)
    want_counts = counts.cpu().tolist()
    packed, send_ids, offsets = hip_route_pack(records, bounds, want_counts, scratch)
    segs, seg_ids = hip_route_pack_fixed(records, bounds, capacity, scratch, counts)
    torch.cuda.synchronize()
    G = len(bounds) - 1
    assert segs.shape == (G * (capacity + 1), 12) and seg_ids.shape == (G * (capacity + 1),)
    overflowed = False
    for b in range(G):
        base = b * (capacity + 1)
        hdr = segs[base].view(torch.int32).cpu().tolist()
        assert hdr[0] == want_counts[b] and hdr[1] == capacity and all(v == 0 for v in hdr[2:])
        n = min(want_counts[b], capacity)
        overflowed |= want_counts[b] > capacity
        assert int(seg_ids[base]) == -1
        assert torch.equal(seg_ids[base + 1:base + 1 + n], send_ids[offsets[b]:offsets[b] + n])
        assert bool((seg_ids[base + 1 + n:base + capacity + 1] == -1).all())
        assert torch.equal(segs[base + 1:base + 1 + n].view(torch.int32), packed[offsets[b]:offsets[b] + n].view(torch.int32))
    assert overflowed == (capacity < max(want_counts))


def test_fixed_capacity_pieces_three_shards_three_bands_on_one_gpu():
    """The fixed-capacity pieces driven by hand for 3 Gaussian shards x 3 bands on one GPU (the equal-split all-to-all replaced by
    slicing + concatenation in rank order): header rows and unused rows enter a band's frame as Gaussians without tiles, so every
    band still renders bit-identically to the fused operator; the gradient rows go back in the same layout and
    gsr_route_return (which skips the rows with id -1) gives every shard its [P,12] record -> parameter gradients."""
    from diff_gaussian_rasterization import _lib, _make_settings, _ptr, _stream_ptr
    from diff_gaussian_rasterization.parallel import (hip_preprocess_shard, hip_route_count, hip_route_pack_fixed, hip_render_segments,
                                                     _i64_array)
    lib = _lib.load()
    dev = torch.device("cuda:0")
    W, H = 336, 200
    cam = make_camera(W, H)
    sc = make_edge_scene(6001, cam, seed=17).to(dev)
    rs = gpu_settings(oracle_settings(cam, bg=torch.tensor([0.3, 0.1, 0.2])), dev)
    g = torch.Generator().manual_seed(9)
    wgt, wd = torch.randn(3, H, W, generator=g).to(dev), torch.randn(1, H, W, generator=g).to(dev)
    col, radii, invd, ref = _fused_reference(rs, sc, wgt, wd, dev)
    cuts, bounds = [0, 1900, 1900 + 2500, 6001], [0, 4, 9, 13]
    G, cap = 3, 2560
    seg = cap + 1
    st = _stream_ptr(dev)
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        t = [x[a:b].contiguous() for x in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
        records, rad, M, _ = hip_preprocess_shard(rs, *t)
        counts, scratch = hip_route_count(records, bounds)
        assert int(counts.max()) <= cap
        segs, seg_ids = hip_route_pack_fixed(records, bounds, cap, scratch, counts)
        shards.append(dict(t=t, rad=rad, segs=segs, ids=seg_ids, a=a, b=b))
    image = torch.zeros(3, H, W, device=dev)
    depth = torch.zeros(1, H, W, device=dev)
    returned = [[None] * G for _ in range(G)]
    for band in range(G):
        recv = torch.cat([sh["segs"][band * seg:(band + 1) * seg] for sh in shards], dim=0).contiguous()      # = the all-to-all
        color, invdp, (geom, binning, img, nr) = hip_render_segments(rs, (bounds[band], bounds[band + 1]), recv, G, cap, False)
        rows = slice(bounds[band] * 16, min(bounds[band + 1] * 16, H))
        image[:, rows], depth[:, rows] = color[:, rows], invdp[:, rows]
        gc, gd = torch.zeros_like(wgt), torch.zeros_like(wd)
        gc[:, rows], gd[:, rows] = wgt[:, rows], wd[:, rows]
        keep = []
        s = _make_settings(rs, keep, (bounds[band], bounds[band + 1]))
        P_recv = G * seg
        scr = torch.empty(int(lib.gsr_backward_scratch_bytes(P_recv, nr)), dtype=torch.uint8, device=dev)
        rp = C.c_void_p(0)
        _lib.check(lib.gsr_backward_blend(C.byref(s), P_recv, nr, _ptr(geom), _ptr(binning), _ptr(img), _ptr(gc), _ptr(gd),
                                          _ptr(scr), C.byref(rp), st), "gsr_backward_blend")
        off = int(rp.value) - scr.data_ptr()
        full = scr[off:off + P_recv * 48].view(torch.float32).view(P_recv, 12).clone()
        for src in range(G):
            returned[src][band] = full[src * seg:(src + 1) * seg]
    torch.cuda.synchronize()
    assert torch.equal(image, col) and torch.equal(depth, invd)
    f = dict(dtype=torch.float32, device=dev)
    for src, sh in enumerate(shards):
        P = sh["b"] - sh["a"]
        back = torch.cat(returned[src], dim=0).contiguous()
        mine = torch.empty(P, 12, **f)
        _lib.check(lib.gsr_route_return(P, G, _i64_array([bnd * seg for bnd in range(G + 1)]), _ptr(sh["ids"]), _ptr(back), _ptr(mine), st),
                   "gsr_route_return")
        want = torch.zeros(P, 12, **f)
        for band in range(G):
            ids = sh["ids"][band * seg:(band + 1) * seg].long()
            ok = ids >= 0
            want.index_add_(0, ids[ok], returned[src][band][ok])
        assert torch.equal(mine, want)
        outs = [torch.empty(P, 3, **f), torch.empty(P, 3, **f), torch.empty(P, 1, **f), torch.empty(P, 3, **f), torch.empty(P, 6, **f),
                torch.empty(P, 16, 3, **f), torch.empty(P, 3, **f), torch.empty(P, 4, **f)]
        keep = []
        s = _make_settings(rs, keep, None)
        t = sh["t"]
        _lib.check(lib.gsr_backward_preprocess(C.byref(s), P, 16, _ptr(t[0]), _ptr(t[1]), None, _ptr(t[2]), _ptr(t[3]), _ptr(t[4]), None,
                                               _ptr(sh["rad"]), None, _ptr(mine), *[_ptr(o) for o in outs], st), "gsr_backward_preprocess")
        torch.cuda.synchronize()
        d_m2, _, d_op, d_m3, _, d_sh, d_sc, d_rot = outs
        a, b = sh["a"], sh["b"]
        for got, w in zip([d_m3, d_sh, d_op, d_sc, d_rot, d_m2], ref):
            assert (got.view(-1) - w[a:b].reshape(-1)).abs().max().item() <= 5e-5 * w.abs().max().item()


def test_gaussian_sharded_fixed_mode_learns_the_capacity_and_falls_back_on_overflow():
    """render_gaussian_sharded with set_exchange_mode("fixed") on one rank: the first frame runs the exact form and teaches the
    policy its capacity, the following frames use fixed-capacity segments (no count read-back) and reproduce the fused operator --
    image / radii / inverse depth bit for bit, gradients to rounding; a capacity that has become too small (another camera) is
    noticed through the overflow flag, the frame is repeated in the exact form and the capacity grows."""
    from diff_gaussian_rasterization.parallel import BandPlan, render_gaussian_sharded, set_exchange_mode
    dev = torch.device("cuda:0")
    cam = make_camera(320, 208)
    sc = make_scene(6000, cam, seed=31, s_med=0.03).to(dev)
    rs = gpu_settings(oracle_settings(cam, bg=torch.tensor([0.1, 0.3, 0.2])), dev)
    g = torch.Generator().manual_seed(3)
    wgt, wd = torch.randn(3, 208, 320, generator=g).to(dev), torch.randn(1, 208, 320, generator=g).to(dev)
    col, radii, invd, ref = _fused_reference(rs, sc, wgt, wd, dev)
    plan = BandPlan.uniform(13, 1)
    policy = set_exchange_mode("fixed")
    try:
        for frame in range(3):
            L = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
            m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
            c2, r2, d2 = render_gaussian_sharded(rs, L[0], L[1], L[2], L[3], L[4], plan, means2D=m2)
            assert torch.equal(c2, col) and torch.equal(r2, radii) and torch.equal(d2, invd), f"frame {frame}"
            ((c2 * wgt).sum() + (d2 * wd).sum()).backward()
            for got, want in zip([t.grad for t in L] + [m2.grad], ref):
                assert (got - want).abs().max().item() <= 5e-5 * want.abs().max().item(), f"frame {frame}"
        assert (policy.frames_exact, policy.frames_fixed, policy.overflows) == (1, 2, 0)
        learned = policy.capacity
        policy.capacity = 256                                   # far below the ~5 000 records of the band
        with torch.no_grad():
            c3, r3, d3 = render_gaussian_sharded(rs, sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations, plan)
        assert torch.equal(c3, col) and torch.equal(d3, invd)
        assert policy.overflows == 1 and policy.frames_exact == 2 and policy.capacity == learned
        with torch.no_grad():
            c4, _, _ = render_gaussian_sharded(rs, sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations, plan)
        assert torch.equal(c4, col) and policy.frames_fixed == 3
    finally:
        set_exchange_mode("exact")
