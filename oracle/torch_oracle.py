"""CPU ORACLE (test infrastructure -- NOT a product path).

Pure-PyTorch, differentiable restatement of the differentiable tile rasterizer that sits behind
`GaussianRasterizer` / `GaussianRasterizationSettings` (gaussian_renderer/__init__.py:14,36-52,91-110).

PARITY UNPINNED: the rasterizer's own source is an un-vendored, empty git submodule in /root/reference
(.gitmodules:4-7) and the reference ships no tests (F2), so no golden vector of
the reference pins this file.  What *is* pinned by in-tree reference code, and checked in tests/:
  * SH -> RGB          against utils/sh_utils.py:57-112 + gaussian_renderer/__init__.py:76-80
  * Sigma3D / packing  against utils/general_utils.py:64-110 + scene/gaussian_model.py:33-37
  * matrix layouts     against scene/cameras.py:80-89 + utils/graphics_utils.py:51-71
Everything else follows SURVEY.md Appendix A (published algorithm of diff-gaussian-rasterization,
branch dr_aa) and is frozen here as this repo's specification.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

SNUG TILE RECTANGLES (`SNUG_TILES`, `preprocess(..., snug=)`): the reference bins every Gaussian into the tiles of the square
of radius 3 sqrt(lambda_max) around its centre.  The product bins it only into the tiles the ellipse q <= 2 ln(255 opacity)
+ 0.01 can reach -- everywhere else alpha < 1/255 and the reference's blend skips the pair -- which changes tiles_touched, the
instance lists and the contributor positions, and NO output (tests/test_oracle.py::test_snug_tiles_change_no_output renders both
ways; tests/test_gpu_parity.py::test_snug_tiles_change_no_bit compares the product's two ways bit for bit).  The default here is the reference's square (`SNUG_TILES = False`); the parity tests of the product
switch it on, and the restatement below follows csrc/gsr_math.h operation by operation (fp64, frexp + atanh series for the
logarithm: no libm call whose last bit differs between implementations).

Arithmetic contract (what "bit-exact bin counts" means): every quantity that decides an integer
output (radii, tile rectangle, tiles_touched, sort keys) is computed in fp32 by the EXACT expression
trees below -- one IEEE-754 rounding per written operation, left-to-right, no fused multiply-add --
and the HIP preprocess kernel is compiled with -ffp-contract=off and IEEE divide/sqrt to evaluate the
same trees.  torch CPU elementwise ops are single IEEE operations, so the two agree bit-for-bit.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import numpy as np
import torch

TILE = 16
NEAR_Z = 0.2
ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.99
T_EPS = 1e-4
LOWPASS = 0.3
AA_FLOOR = 0.000025
FRUSTUM_CLAMP = 1.3

# utils/sh_utils.py:26-54
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


SNUG_TILES = False      # module default of preprocess(snug=None): False = the reference's tile square, True = the product's snug rectangle


def det_log(v: torch.Tensor) -> torch.Tensor:
    """ln(v) for finite v > 0 (float64) from IEEE basic operations only -- csrc/gsr_math.h gsr_log_det, operation by operation."""
    m, e = torch.frexp(v)
    low = m < 0.70710678118654752
    m = torch.where(low, m * 2.0, m)
    e = torch.where(low, e - 1, e)
    sv = (m - 1.0) / (m + 1.0)
    s2 = sv * sv
    p = torch.full_like(v, 1.0 / 19.0)
    for k in (17.0, 15.0, 13.0, 11.0, 9.0, 7.0, 5.0, 3.0):
        p = p * s2 + 1.0 / k
    p = p * s2 + 1.0
    t1 = e.to(torch.float64) * 0.6931471805599453
    t3 = (2.0 * sv) * p
    return t1 + t3


def tau_of_opacity(opacity: torch.Tensor) -> torch.Tensor:
    """tau = 2 ln(255 opacity) + 0.01 as float32 (csrc/gsr_math.h gsr_tau): a splat reaches alpha >= 1/255 only where q <= tau."""
    vf = torch.tensor(255.0, dtype=torch.float32) * opacity.to(torch.float32)
    ok = (vf > 0) & (vf < 3.0e38)
    t = (2.0 * det_log(torch.where(ok, vf, torch.ones_like(vf)).to(torch.float64)) + 0.01).to(torch.float32)
    t = torch.where(ok, t, torch.where(vf == 0, torch.full_like(t, -math.inf),
                                       torch.where(vf >= 3.0e38, torch.full_like(t, math.inf), torch.full_like(t, math.nan))))
    return t


class Settings(NamedTuple):
    """Same 13 fields, same order, as the reference's GaussianRasterizationSettings
    (gaussian_renderer/__init__.py:36-50)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool


def settings_from_camera(cam, bg, sh_degree=3, scale_modifier=1.0, antialiasing=False, debug=False) -> Settings:
    return Settings(int(cam.image_height), int(cam.image_width), cam.tanfovx, cam.tanfovy, bg, scale_modifier,
                    cam.world_view_transform, cam.full_proj_transform, sh_degree, cam.camera_center,
                    False, debug, antialiasing)


# ----------------------------------------------------------------------------------------------
# custom autograd pieces encoding the reference's (recalled) backward conventions, SURVEY 8(c)
# ----------------------------------------------------------------------------------------------
class _ConicFromCov(torch.autograd.Function):
    """conic = inverse of [[a,b],[b,c]].  Forward exact; backward uses 1/(det^2 + 1e-7) like the
    reference's computeCov2D backward instead of 1/det^2."""

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        det_inv = 1.0 / det
        ctx.save_for_backward(a, b, c, det)
        return c * det_inv, -b * det_inv, a * det_inv

    @staticmethod
    def backward(ctx, gA, gB, gC):
        # A = c/det, B = -b/det, C = a/det, det = ac - b^2
        a, b, c, det = ctx.saved_tensors
        d2 = 1.0 / (det * det + 1e-7)
        ga = d2 * (-c * c * gA + b * c * gB + (det - a * c) * gC)
        gc = d2 * (-a * a * gC + a * b * gB + (det - a * c) * gA)
        gb = d2 * (2 * b * c * gA - (det + 2 * b * b) * gB + 2 * a * b * gC)
        return ga, gb, gc


def _st_min(x, cap):
    """min(cap, x) in the forward, identity in the backward (reference's alpha cap convention)."""
    return x + (torch.clamp(x, max=cap) - x).detach()


# ----------------------------------------------------------------------------------------------
# A.2  preprocess
# ----------------------------------------------------------------------------------------------
def _cols(t):
    return [t[:, i] for i in range(t.shape[1])]


def _sqrt(x):
    """Correctly rounded sqrt.  torch.sqrt on CPU fp32 tensors is NOT IEEE-exact (its vectorised kernel is
    off by one ulp for ~0.7 % of inputs); the HIP kernel's sqrtf is.  sqrt in fp64 followed by rounding to fp32
    is correctly rounded (53 >= 2*24+2), differentiable, and what the bit-exactness contract needs."""
    if x.dtype == torch.float32:
        return torch.sqrt(x.double()).float()
    return torch.sqrt(x)


def compute_cov3d(scales, rotations, scale_modifier, dtype):
    """Sigma = (R S)(R S)^T, packed [xx,xy,xz,yy,yz,zz]; R as utils/general_utils.py:90-98 but WITHOUT
    re-normalising q (Python normalises before the call, scene/gaussian_model.py:107-108)."""
    mod = torch.tensor(scale_modifier, dtype=dtype)
    s0, s1, s2 = [mod * s for s in _cols(scales)]
    r, x, y, z = _cols(rotations)
    two = torch.tensor(2.0, dtype=dtype)
    one = torch.tensor(1.0, dtype=dtype)
    R = [[one - two * (y * y + z * z), two * (x * y - r * z), two * (x * z + r * y)],
         [two * (x * y + r * z), one - two * (x * x + z * z), two * (y * z - r * x)],
         [two * (x * z - r * y), two * (y * z + r * x), one - two * (x * x + y * y)]]
    s = [s0, s1, s2]
    M = [[R[i][j] * s[j] for j in range(3)] for i in range(3)]

    def dot(i, j):
        return M[i][0] * M[j][0] + M[i][1] * M[j][1] + M[i][2] * M[j][2]

    return torch.stack([dot(0, 0), dot(0, 1), dot(0, 2), dot(1, 1), dot(1, 2), dot(2, 2)], dim=1)


def eval_sh_colors(deg, shs, means3D, campos, dtype):
    """SH -> RGB, expression order of utils/sh_utils.py:78-104, then +0.5 and clamp >= 0
    (gaussian_renderer/__init__.py:80).  Returns (rgb [P,3], clamped [P,3] bool)."""
    d = means3D - campos[None, :]
    dx, dy, dz = _cols(d)
    n = _sqrt(dx * dx + dy * dy + dz * dz)
    x, y, z = (dx / n)[:, None], (dy / n)[:, None], (dz / n)[:, None]
    c = lambda v: torch.tensor(v, dtype=dtype)
    sh = shs  # [P,M,3]
    result = c(SH_C0) * sh[:, 0]
    if deg > 0:
        result = result - c(SH_C1) * y * sh[:, 1] + c(SH_C1) * z * sh[:, 2] - c(SH_C1) * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + c(SH_C2[0]) * xy * sh[:, 4] + c(SH_C2[1]) * yz * sh[:, 5]
                      + c(SH_C2[2]) * (c(2.0) * zz - xx - yy) * sh[:, 6]
                      + c(SH_C2[3]) * xz * sh[:, 7] + c(SH_C2[4]) * (xx - yy) * sh[:, 8])
            if deg > 2:
                result = (result + c(SH_C3[0]) * y * (c(3.0) * xx - yy) * sh[:, 9]
                          + c(SH_C3[1]) * xy * z * sh[:, 10]
                          + c(SH_C3[2]) * y * (c(4.0) * zz - xx - yy) * sh[:, 11]
                          + c(SH_C3[3]) * z * (c(2.0) * zz - c(3.0) * xx - c(3.0) * yy) * sh[:, 12]
                          + c(SH_C3[4]) * x * (c(4.0) * zz - xx - yy) * sh[:, 13]
                          + c(SH_C3[5]) * z * (xx - yy) * sh[:, 14]
                          + c(SH_C3[6]) * x * (xx - c(3.0) * yy) * sh[:, 15])
    result = result + c(0.5)
    clamped = (result < 0).detach()
    return torch.clamp_min(result, 0.0), clamped


def preprocess(means3D, opacities, s: Settings, shs=None, colors_precomp=None, scales=None, rotations=None,
               cov3D_precomp=None, means2D=None, tile_y0: int = 0, tile_y1: Optional[int] = None, snug: Optional[bool] = None):
    """SURVEY Appendix A.2.  All tensors on CPU, dtype = means3D.dtype (fp32 for parity, fp64 for
    finite-difference references).  tile_y0/tile_y1 restrict binning to a band of tile rows
    (multi-GPU screen sharding); radii are NOT affected by the band."""
    dtype = means3D.dtype
    P = means3D.shape[0]
    W, H = int(s.image_width), int(s.image_height)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    if tile_y1 is None:
        tile_y1 = gy
    c = lambda v: torch.tensor(v, dtype=dtype)
    f32 = np.float32
    if dtype == torch.float32:
        tanx, tany = f32(s.tanfovx), f32(s.tanfovy)
        focal_x = float(f32(W) / (f32(2.0) * tanx))
        focal_y = float(f32(H) / (f32(2.0) * tany))
        limx = float(f32(FRUSTUM_CLAMP) * tanx)
        limy = float(f32(FRUSTUM_CLAMP) * tany)
    else:
        focal_x = W / (2.0 * s.tanfovx)
        focal_y = H / (2.0 * s.tanfovy)
        limx = FRUSTUM_CLAMP * s.tanfovx
        limy = FRUSTUM_CLAMP * s.tanfovy
    vm = s.viewmatrix.detach().to("cpu", dtype).reshape(16)
    pm = s.projmatrix.detach().to("cpu", dtype).reshape(16)
    campos = s.campos.detach().to("cpu", dtype).reshape(3)
    x, y, z = _cols(means3D)

    # 1. view space (transformPoint4x3; flat index = column-major math matrix)
    pvx = vm[0] * x + vm[4] * y + vm[8] * z + vm[12]
    pvy = vm[1] * x + vm[5] * y + vm[9] * z + vm[13]
    pvz = vm[2] * x + vm[6] * y + vm[10] * z + vm[14]
    in_front = (pvz > NEAR_Z).detach()

    # 2. clip space (transformPoint4x4) + homogeneous divide 1/(w + 1e-7)
    hx = pm[0] * x + pm[4] * y + pm[8] * z + pm[12]
    hy = pm[1] * x + pm[5] * y + pm[9] * z + pm[13]
    hw = pm[3] * x + pm[7] * y + pm[11] * z + pm[15]
    pw = c(1.0) / (hw + c(1e-7))
    projx, projy = hx * pw, hy * pw

    # 3. Sigma3D
    if cov3D_precomp is not None:
        cov3D = cov3D_precomp
    else:
        cov3D = compute_cov3d(scales, rotations, s.scale_modifier, dtype)
    S00, S01, S02, S11, S12, S22 = _cols(cov3D)

    # 4. EWA Sigma2D = J W Sigma W^T J^T, frustum clamp with stop-gradient outside
    txtz, tytz = pvx / pvz, pvy / pvz
    cx = torch.clamp(txtz, -limx, limx)
    cy = torch.clamp(tytz, -limy, limy)
    inx = ((txtz >= -limx) & (txtz <= limx)).detach()
    iny = ((tytz >= -limy) & (tytz <= limy)).detach()
    tx_c = cx * pvz
    ty_c = cy * pvz
    tx = torch.where(inx, tx_c, tx_c.detach())
    ty = torch.where(iny, ty_c, ty_c.detach())
    tz = pvz
    fx, fy = c(focal_x), c(focal_y)
    tz2 = tz * tz
    J00 = fx / tz
    J02 = -(fx * tx) / tz2
    J11 = fy / tz
    J12 = -(fy * ty) / tz2
    # W[i][j] = vm[i + 4j]
    T00 = J00 * vm[0] + J02 * vm[2]
    T01 = J00 * vm[4] + J02 * vm[6]
    T02 = J00 * vm[8] + J02 * vm[10]
    T10 = J11 * vm[1] + J12 * vm[2]
    T11 = J11 * vm[5] + J12 * vm[6]
    T12 = J11 * vm[9] + J12 * vm[10]
    u0 = S00 * T00 + S01 * T01 + S02 * T02
    u1 = S01 * T00 + S11 * T01 + S12 * T02
    u2 = S02 * T00 + S12 * T01 + S22 * T02
    v0 = S00 * T10 + S01 * T11 + S02 * T12
    v1 = S01 * T10 + S11 * T11 + S12 * T12
    v2 = S02 * T10 + S12 * T11 + S22 * T12
    a0 = T00 * u0 + T01 * u1 + T02 * u2
    b = T10 * u0 + T11 * u1 + T12 * u2
    c0 = T10 * v0 + T11 * v1 + T12 * v2

    # 5. low-pass, anti-aliasing scale, conic
    det0 = a0 * c0 - b * b
    a = a0 + c(LOWPASS)
    cc = c0 + c(LOWPASS)
    det = a * cc - b * b
    if s.antialiasing:
        aa = _sqrt(torch.clamp_min(det0 / det, AA_FLOOR))
    else:
        aa = torch.ones_like(det)
    det_ok = (det != 0).detach()
    conA, conB, conC = _ConicFromCov.apply(a, b, cc)

    # 6. radius
    with torch.no_grad():
        mid = c(0.5) * (a + cc)
        disc = _sqrt(torch.clamp_min(mid * mid - det, 0.1))
        lam = torch.maximum(mid + disc, mid - disc)
        radius_f = torch.ceil(c(3.0) * _sqrt(lam))
    # 7. pixel centre; means2D is the zero "grad sink" (gaussian_renderer/__init__.py:26-30): its
    #    gradient must equal dL/d(NDC xy) = dL/dpix * (0.5 W, 0.5 H)  (SURVEY A.5 units trap)
    pixx = ((projx + c(1.0)) * c(float(W)) - c(1.0)) * c(0.5)
    pixy = ((projy + c(1.0)) * c(float(H)) - c(1.0)) * c(0.5)
    if means2D is not None:
        pixx = pixx + _zero_with_grad(means2D[:, 0], 0.5 * W)
        pixy = pixy + _zero_with_grad(means2D[:, 1], 0.5 * H)

    # 8. tile rectangle
    with torch.no_grad():
        def tile_lo(p, g):
            v = torch.nan_to_num(torch.trunc((p - radius_f) / c(float(TILE))), nan=0.0)
            return torch.clamp(v, 0.0, float(g)).to(torch.int64)

        def tile_hi(p, g):
            v = torch.nan_to_num(torch.trunc((p + radius_f + c(float(TILE - 1))) / c(float(TILE))), nan=0.0)
            return torch.clamp(v, 0.0, float(g)).to(torch.int64)

        rminx, rmaxx = tile_lo(pixx, gx), tile_hi(pixx, gx)
        rminy, rmaxy = tile_lo(pixy, gy), tile_hi(pixy, gy)
        area_full = (rmaxx - rminx) * (rmaxy - rminy)
        visible = in_front & det_ok & (area_full > 0) & (radius_f < 2.0e9)
        radii = torch.where(visible, torch.nan_to_num(radius_f, nan=0.0, posinf=0.0).to(torch.int64),
                            torch.zeros_like(area_full)).to(torch.int32)
        if SNUG_TILES if snug is None else snug:
            # the product's snug rectangle (csrc/gsr_math.h gsr_project, same operations in the same order, fp64 from the fp32
            # conic / centre / opacity); `radii` keeps the reference's value
            td = tau_of_opacity((opacities.reshape(P) * aa).detach()).to(torch.float64)
            Ad, Bd, Cd = conA.detach().to(torch.float64), conB.detach().to(torch.float64), conC.detach().to(torch.float64)
            detc = Ad * Cd - Bd * Bd
            dead = td <= 0.0
            shrink = (~dead) & (detc > 0.0) & (td < 1.0e30)
            safe = torch.where(shrink, detc, torch.ones_like(detc))
            tds = torch.where(shrink, td, torch.ones_like(td))
            sqrt64 = lambda t: torch.from_numpy(np.sqrt(t.numpy()))      # (torch.sqrt is not correctly rounded on CPU, see _sqrt)
            ex = sqrt64(tds * Cd / safe) * 1.01 + 0.5
            ey = sqrt64(tds * Ad / safe) * 1.01 + 0.5
            shrink = shrink & (ex < 1.0e9) & (ey < 1.0e9)
            cxd, cyd = pixx.detach().to(torch.float64), pixy.detach().to(torch.float64)

            def snug_axis(cd, e, lo, hi):
                lo_d, hi_d = lo.to(torch.float64), hi.to(torch.float64)
                l = torch.floor((cd - e) / 16.0)
                h = torch.floor((cd + e) / 16.0) + 1.0
                l = torch.nan_to_num(l, nan=0.0, posinf=0.0, neginf=0.0)
                h = torch.nan_to_num(h, nan=0.0, posinf=0.0, neginf=0.0)
                lo2 = torch.where(shrink & (l > lo_d), torch.where(l < hi_d, l, hi_d), lo_d)
                hi2 = torch.where(shrink & (h < hi_d), torch.where(h > lo2, h, lo2), hi_d)
                return lo2.to(torch.int64), hi2.to(torch.int64)

            sminx, smaxx = snug_axis(cxd, ex, rminx, rmaxx)
            sminy, smaxy = snug_axis(cyd, ey, rminy, rmaxy)
            smaxx = torch.where(dead, sminx, smaxx)
            smaxy = torch.where(dead, sminy, smaxy)
            rminx, rmaxx, rminy, rmaxy = sminx, smaxx, sminy, smaxy
        # band restriction (multi-GPU): only the rows [tile_y0, tile_y1) are binned
        bminy = torch.clamp(rminy, tile_y0, tile_y1)
        bmaxy = torch.clamp(rmaxy, tile_y0, tile_y1)
        tiles_touched = torch.where(visible, (rmaxx - rminx) * (bmaxy - bminy), torch.zeros_like(area_full))

    # 9. colour
    if colors_precomp is not None:
        rgb = colors_precomp
        clamped = torch.zeros(P, 3, dtype=torch.bool)
    else:
        rgb, clamped = eval_sh_colors(int(s.sh_degree), shs, means3D, campos, dtype)

    opac = opacities.reshape(P) * aa
    return {
        "depths": pvz, "radii": radii, "visible": visible, "means2D": torch.stack([pixx, pixy], dim=1),
        "cov3D": cov3D, "conic": torch.stack([conA, conB, conC], dim=1), "opacity": opac, "rgb": rgb,
        "clamped": clamped, "tiles_touched": tiles_touched.to(torch.int64),
        "rect": torch.stack([rminx, bminy, rmaxx, bmaxy], dim=1), "grid": (gx, gy),
        "band": (tile_y0, tile_y1),
    }


def _zero_with_grad(v, k):
    """A tensor that is exactly zero in the forward pass and has d/dv = k."""
    kv = v * k
    return kv - kv.detach()


# ----------------------------------------------------------------------------------------------
# A.3  binning
# ----------------------------------------------------------------------------------------------
def depth_bits(depths32: torch.Tensor) -> torch.Tensor:
    """fp32 bit pattern as non-negative int64 (positive floats order like their bit patterns)."""
    return depths32.detach().to(torch.float32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF


def bin_and_sort(pre):
    """Emit (tile|depth, idx) for every overlapped tile (y-major, then x), stable sort ascending,
    per-tile [start,end).  Returns dict(point_list [R], tile_of [R], ranges [T,2], R)."""
    gx, gy = pre["grid"]
    tt = pre["tiles_touched"]
    P = tt.shape[0]
    offsets = torch.cumsum(tt, 0)
    R = int(offsets[-1]) if P > 0 else 0
    idx = torch.repeat_interleave(torch.arange(P, dtype=torch.int64), tt)
    start = (offsets - tt)[idx]
    local = torch.arange(R, dtype=torch.int64) - start
    rect = pre["rect"]
    w = (rect[:, 2] - rect[:, 0])[idx].clamp_min(1)
    ty = rect[:, 1][idx] + torch.div(local, w, rounding_mode="floor")
    tx = rect[:, 0][idx] + local % w
    tile = ty * gx + tx
    key = (tile << 32) | depth_bits(pre["depths"])[idx]
    skey, order = torch.sort(key, stable=True)
    point_list = idx[order]
    tile_sorted = skey >> 32
    counts = torch.bincount(tile_sorted, minlength=gx * gy)
    ends = torch.cumsum(counts, 0)
    starts = ends - counts
    ranges = torch.stack([starts, ends], dim=1)
    ranges[counts == 0] = 0
    return {"point_list": point_list, "keys": skey, "ranges": ranges, "R": R, "tile_counts": counts}


# ----------------------------------------------------------------------------------------------
# A.4  forward blend (one tile, all its pixels at once; differentiable)
# ----------------------------------------------------------------------------------------------
def _blend_tile(px, py, xy, conic, opac, rgb, invd, want_fragile=False, frag_eps=3e-5):
    """px,py [n] pixel coords (float), per-Gaussian arrays in list order [N,...].
    Returns C [n,3], D [n], final_T [n], n_contrib [n] (int64), fragile [n] bool."""
    dtype = xy.dtype
    n = px.shape[0]
    N = xy.shape[0]
    if N == 0:
        z = torch.zeros(n, dtype=dtype)
        return (torch.zeros(n, 3, dtype=dtype), z, torch.ones(n, dtype=dtype),
                torch.zeros(n, dtype=torch.int64), torch.zeros(n, dtype=torch.bool))
    dx = xy[None, :, 0] - px[:, None]
    dy = xy[None, :, 1] - py[:, None]
    A, B, Cc = conic[None, :, 0], conic[None, :, 1], conic[None, :, 2]
    power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
    G = torch.exp(power)
    a_raw = opac[None, :] * G
    alpha = _st_min(a_raw, ALPHA_MAX)
    with torch.no_grad():
        keep = (power <= 0) & (alpha >= ALPHA_MIN)
    alpha_eff = torch.where(keep, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - alpha_eff
    Tincl = torch.cumprod(one_minus, dim=1)
    Texcl = torch.cat([torch.ones(n, 1, dtype=dtype), Tincl[:, :-1]], dim=1)
    with torch.no_grad():
        term = keep & (Tincl < T_EPS)
        dead = torch.cumsum(term.to(torch.int32), dim=1) > 0
        contrib = keep & ~dead
        any_dead = dead[:, -1]
        first_dead = torch.argmax(dead.to(torch.int8), dim=1)
        pos = torch.arange(1, N + 1, dtype=torch.int64)[None, :]
        n_contrib = (contrib.to(torch.int64) * pos).max(dim=1).values
    w = torch.where(contrib, alpha_eff * Texcl, torch.zeros_like(alpha_eff))
    C = (w[:, :, None] * rgb[None, :, :]).sum(dim=1)
    D = (w * invd[None, :]).sum(dim=1)
    T_at = torch.gather(Texcl, 1, first_dead[:, None])[:, 0]
    final_T = torch.where(any_dead, T_at, Tincl[:, -1])
    fragile = torch.zeros(n, dtype=torch.bool)
    if want_fragile:
        with torch.no_grad():
            # pairs the sequential loop actually evaluates: everything up to and including the
            # terminating one.  A pixel is "fragile" when one of the loop's three hard thresholds is
            # within rounding-noise distance, i.e. another correct fp32 evaluation order (FMA
            # contraction, a different exp) may legitimately take the other branch there.
            live = ~dead | term
            near_a = (torch.abs(alpha - ALPHA_MIN) < frag_eps * ALPHA_MIN) & (power <= 1e-6)
            near_p = (torch.abs(power) < 1e-6) & (power != 0) & (a_raw >= ALPHA_MIN * 0.5)
            near_t = keep & (torch.abs(Tincl - T_EPS) < 10 * frag_eps * T_EPS)
            fragile = ((near_a | near_p | near_t) & live).any(dim=1)
    return C, D, final_T, n_contrib, fragile


def render_tiles(pre, bins, s: Settings, want_fragile=False, tiles=None):
    """Blend every tile (or the listed tile ids).  Returns color [3,H,W], invdepth [1,H,W],
    final_T [H,W], n_contrib [H,W], fragile [H,W]."""
    dtype = pre["means2D"].dtype
    W, H = int(s.image_width), int(s.image_height)
    gx, gy = pre["grid"]
    bg = s.bg.detach().to("cpu", dtype).reshape(3)
    color = torch.zeros(3, H, W, dtype=dtype)
    invdepth = torch.zeros(1, H, W, dtype=dtype)
    final_T = torch.ones(H, W, dtype=dtype)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    fragile = torch.zeros(H, W, dtype=torch.bool)
    invd_all = 1.0 / pre["depths"]
    ranges = bins["ranges"]
    y0, y1 = pre["band"]
    tile_ids = range(y0 * gx, y1 * gx) if tiles is None else tiles
    col_parts = []
    for t in tile_ids:
        tyi, txi = divmod(int(t), gx)
        x0, yy0 = txi * TILE, tyi * TILE
        x1, yy1 = min(x0 + TILE, W), min(yy0 + TILE, H)
        ys, xs = torch.meshgrid(torch.arange(yy0, yy1), torch.arange(x0, x1), indexing="ij")
        px = xs.reshape(-1).to(dtype)
        py = ys.reshape(-1).to(dtype)
        a, b = int(ranges[t, 0]), int(ranges[t, 1])
        ids = bins["point_list"][a:b]
        C, D, fT, nc, fr = _blend_tile(px, py, pre["means2D"][ids], pre["conic"][ids], pre["opacity"][ids],
                                       pre["rgb"][ids], invd_all[ids], want_fragile)
        out = C + fT[:, None] * bg[None, :]
        col_parts.append((yy0, yy1, x0, x1, out, D))
        final_T[yy0:yy1, x0:x1] = fT.detach().reshape(yy1 - yy0, x1 - x0)
        n_contrib[yy0:yy1, x0:x1] = nc.reshape(yy1 - yy0, x1 - x0)
        fragile[yy0:yy1, x0:x1] = fr.reshape(yy1 - yy0, x1 - x0)
    # assemble differentiably (index_put on a fresh tensor keeps autograd history per tile)
    for (yy0, yy1, x0, x1, out, D) in col_parts:
        hh, ww = yy1 - yy0, x1 - x0
        color[:, yy0:yy1, x0:x1] = out.transpose(0, 1).reshape(3, hh, ww)
        invdepth[0, yy0:yy1, x0:x1] = D.reshape(hh, ww)
    return color, invdepth, final_T, n_contrib, fragile


# ----------------------------------------------------------------------------------------------
# top level: mirrors GaussianRasterizer.forward (gaussian_renderer/__init__.py:102-110)
# ----------------------------------------------------------------------------------------------
def rasterize(means3D, means2D, opacities, s: Settings, shs=None, colors_precomp=None, scales=None,
              rotations=None, cov3D_precomp=None, want_fragile=False, tile_y0=0, tile_y1=None, return_aux=False):
    """Returns (color[3,H,W], radii[P] int32, invdepth[1,H,W]); differentiable w.r.t. every float input.
    With return_aux also returns a dict of intermediates (tiles_touched, ranges, point_list, R, ...)."""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    P = means3D.shape[0]
    H, W = int(s.image_height), int(s.image_width)
    if P == 0:
        z = torch.zeros(3, H, W, dtype=means3D.dtype)
        out = (z, torch.zeros(0, dtype=torch.int32), torch.zeros(1, H, W, dtype=means3D.dtype))
        return out + ({"R": 0},) if return_aux else out
    pre = preprocess(means3D, opacities, s, shs, colors_precomp, scales, rotations, cov3D_precomp, means2D,
                     tile_y0, tile_y1)
    bins = bin_and_sort(pre)
    color, invdepth, final_T, n_contrib, fragile = render_tiles(pre, bins, s, want_fragile)
    if return_aux:
        aux = dict(pre)
        aux.update(bins)
        aux.update({"final_T": final_T, "n_contrib": n_contrib, "fragile": fragile})
        return color, pre["radii"], invdepth, aux
    return color, pre["radii"], invdepth


def mark_visible(means3D, viewmatrix):
    """checkFrustum / markVisible: in front of the 0.2 near plane."""
    vm = viewmatrix.detach().to("cpu", means3D.dtype).reshape(16)
    x, y, z = _cols(means3D)
    pvz = vm[2] * x + vm[6] * y + vm[10] * z + vm[14]
    return pvz > NEAR_Z
