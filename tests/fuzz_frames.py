"""The random frames of tools/gpu_fuzz_render.py as a replayable stream: `FrameStream(seed)` yields the fuzzer's frame `it` with the
same draws in the same order, WITHOUT rendering anything, so that a frame a fuzz run reported (seed, it) can be rebuilt anywhere --
by the fuzzer itself, by tests/golden/make_golden_hard_frames.py (which freezes the reported hard frames as fixtures) and by a
debugging session in the build container.

Test infrastructure (imports oracle/ through tests/helpers.py), not product code."""
import random

import torch

from helpers import O, make_camera, look_at_camera, make_scene, make_edge_scene, oracle_settings

SIZES = [(64, 48), (17, 9), (1, 40), (300, 2), (250, 131), (333, 200), (16, 16), (129, 65), (480, 270), (31, 257)]
KINDS = ["cloud", "cloud", "edge", "huge", "needles", "extreme_needles", "single"]


class Frame:
    """One fuzz frame: camera, scene, the oracle's settings record, call form and the loss weights."""

    def leaves(self, where, dtype=torch.float32):
        """The differentiable inputs of the operator call for this frame's call form (fresh leaf tensors)."""
        sc = self.sc
        L = {"means3D": sc.means3D, "opacities": sc.opacities}
        if self.colors_form:
            L["colors_precomp"] = self.colors
        elif self.split_form:
            L["dc"], L["shs"] = sc.shs[:, :1].contiguous(), sc.shs[:, 1:].contiguous()
        else:
            L["shs"] = sc.shs
        if self.cov_form:
            L["cov3D_precomp"] = self.cov
        else:
            L["scales"], L["rotations"] = sc.scales, sc.rotations
        L = {k: v.detach().clone().to(device=where, dtype=dtype).requires_grad_(True) for k, v in L.items()}
        L["means2D"] = torch.zeros(sc.P, 3, device=where, dtype=dtype, requires_grad=True)
        return L

    def oracle_backward(self, dtype=torch.float32):
        """Gradients of the frame's loss through the oracle's autograd, in `dtype`.  -> (leaves, color, radii, invdepth)"""
        L = self.leaves("cpu", dtype)
        kw = {k: v for k, v in L.items() if k not in ("means3D", "means2D", "opacities", "dc")}
        if self.split_form:
            kw["shs"] = torch.cat([L["dc"], L["shs"]], dim=1)
        col, radii, invd = O.rasterize(L["means3D"], L["means2D"], L["opacities"], self.s, **kw)
        loss = (col * self.wc.to(dtype)).sum() + ((invd * self.wd.to(dtype)).sum() if self.use_depth else 0.0)
        if loss.requires_grad:      # (nothing visible: the oracle's image is a constant, every gradient is zero)
            loss.backward()
        return L, col.detach(), radii, invd.detach()


class FrameStream:
    """Iterating yields Frame 0, 1, 2, ... of `python tools/gpu_fuzz_render.py N seed`."""

    def __init__(self, seed):
        self.seed = seed
        self.rng = random.Random(seed)
        self.it = 0

    def _build(self, it):
        rng = self.rng
        W, H = rng.choice(SIZES)
        fov = rng.choice([25.0, 45.0, 60.0, 90.0, 110.0])
        if rng.random() < 0.5:
            cam = make_camera(W, H, fovx_deg=fov)
        else:
            eye = (rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), rng.uniform(-1.5, 0.5))
            cam = look_at_camera(W, H, eye, (rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 3.0), fovx_deg=fov)
        kind = rng.choice(KINDS)
        P = {"single": 1, "huge": rng.randint(2, 40)}.get(kind, int(10 ** rng.uniform(0.3, 3.3)))
        max_deg = rng.choice([3, 3, 3, 0, 1, 2])
        if kind == "edge":
            sc = make_edge_scene(max(P, 8), cam, seed=100 + it)
            max_deg = 3
        else:
            sc = make_scene(P, cam, seed=100 + it, s_med=10 ** rng.uniform(-2.2, -0.9), max_sh_degree=max_deg)
        g = torch.Generator().manual_seed(it)
        if kind == "huge":          # splats much larger than the frame (radius clamps, every tile touched)
            sc.scales.mul_(rng.choice([30.0, 100.0]))
        elif kind == "needles":     # one axis 10-40x the others: rectangles mostly empty
            sc.scales[:, 0].mul_(rng.choice([5.0, 20.0]))
            sc.scales[:, 1:].mul_(0.5)
        elif kind == "extreme_needles":     # 250-1500x: conics with condition numbers of 1e5 and more -- bins bit-exact, image / gradients only loosely
            sc.scales[:, 0].mul_(rng.choice([50.0, 300.0]))
            sc.scales[:, 1:].mul_(0.2)
        return cam, sc, kind, max_deg, g

    def __iter__(self):
        return self

    def __next__(self):
        """Frame `self.it`; a frame whose construction raises is returned as the exception (the draws it made stay made, like in the fuzzer)."""
        rng = self.rng
        it = self.it
        self.it += 1
        f = Frame()
        f.it, f.seed = it, self.seed
        f.cam, f.sc, f.kind, f.max_deg, f.g = self._build(it)
        f.P = f.sc.P
        f.H, f.W = f.cam.image_height, f.cam.image_width
        f.deg = rng.randint(0, f.max_deg)
        f.opts = dict(bg=torch.rand(3, generator=f.g) if rng.random() < 0.7 else None, sh_degree=f.deg,
                      scale_modifier=rng.choice([1.0, 1.0, 0.5, 1.7]), antialiasing=rng.random() < 0.4)
        f.s = oracle_settings(f.cam, **f.opts)
        f.colors_form = rng.random() < 0.25
        f.cov_form = rng.random() < 0.25
        f.split_form = (not f.colors_form) and rng.random() < 0.4
        f.use_depth = rng.random() < 0.5
        f.form = ("colors" if f.colors_form else ("split_sh" if f.split_form else "shs")) + ("+cov" if f.cov_form else "")
        f.colors = torch.rand(f.P, 3, generator=f.g) if f.colors_form else None
        f.cov = O.compute_cov3d(f.sc.scales, f.sc.rotations, f.s.scale_modifier, torch.float32) if f.cov_form else None
        f.wc = f.wd = None
        return f

    @staticmethod
    def loss_weights(f):
        """The loss weights are drawn from the frame's torch generator AFTER the forward checks (the fuzzer's order)."""
        if f.wc is None:
            f.wc = torch.randn(3, f.H, f.W, generator=f.g)
            f.wd = torch.randn(1, f.H, f.W, generator=f.g) * 0.3 if f.use_depth else None
        return f.wc, f.wd


def frame(seed, it):
    """Frame `it` of the stream `seed`, loss weights drawn."""
    st = FrameStream(seed)
    for _ in range(it):
        next(st)
    f = next(st)
    FrameStream.loss_weights(f)
    return f


GAMMA = 2.0 ** -22      # two fp32 ulps of the quadratic form's TERMS (measured: the worst pixel of 345 frames sits at 0.1 of 2^-21)


def conditioning(aux, s_):
    """What fp32 can resolve at every pixel.  The exponent of a pair is -(A dx^2 + C dy^2)/2 - B dx dy: for an anisotropic splat its terms
    are orders of magnitude larger than their sum, so two correct fp32 evaluations (other association, FMA contraction -- the reference's
    nvcc build contracts too) differ by ~GAMMA * M with M = (|A| dx^2 + |C| dy^2)/2 + |B dx dy|.  First-order propagation through the
    compositing sum: |d image| <= 2 cmax GAMMA sum_i w_i M_i (w_i = alpha_i T_i), |d T| <= T sum_i alpha_i/(1-alpha_i) GAMMA M_i.
    Returns per pixel: the image bound's sum_i w_i M_i, the relative T bound, and a flag "a hard threshold of an evaluated pair is within
    that noise" (alpha vs 1/255, power vs 0, T vs 1e-4) -- there either branch is a correct fp32 result."""
    H_, W_ = int(s_.image_height), int(s_.image_width)
    gx_, gy_ = aux["grid"]
    B = torch.zeros(H_, W_, dtype=torch.float64)
    E = torch.zeros(H_, W_, dtype=torch.float64)
    J = torch.zeros(H_, W_, dtype=torch.float64)
    flag = torch.zeros(H_, W_, dtype=torch.bool)
    xy, con, opa = aux["means2D"].double(), aux["conic"].double(), aux["opacity"].double().reshape(-1)
    pl, rng_ = aux["point_list"], aux["ranges"]
    for t in range(gx_ * gy_):
        a, b = int(rng_[t, 0]), int(rng_[t, 1])
        if b <= a:
            continue
        ty, tx = divmod(t, gx_)
        x0, y0 = tx * 16, ty * 16
        x1, y1 = min(x0 + 16, W_), min(y0 + 16, H_)
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        px, py = xs.reshape(-1).double(), ys.reshape(-1).double()
        ids = pl[a:b]
        dx = xy[ids, 0][None, :] - px[:, None]
        dy = xy[ids, 1][None, :] - py[:, None]
        A_, B_, C_ = con[ids, 0][None, :], con[ids, 1][None, :], con[ids, 2][None, :]
        power = -0.5 * (A_ * dx * dx + C_ * dy * dy) - B_ * dx * dy
        M = 0.5 * (A_.abs() * dx * dx + C_.abs() * dy * dy) + (B_ * dx * dy).abs()
        alpha = torch.clamp(opa[ids][None, :] * torch.exp(power), max=0.99)
        keep = (power <= 0) & (alpha >= 1.0 / 255.0)
        ae = torch.where(keep, alpha, torch.zeros_like(alpha))
        Tincl = torch.cumprod(1.0 - ae, dim=1)
        Texcl = torch.cat([torch.ones(len(px), 1, dtype=torch.float64), Tincl[:, :-1]], dim=1)
        dead = torch.cumsum((keep & (Tincl < 1e-4)).to(torch.int32), dim=1) > 0
        evaluated = ~torch.cat([torch.zeros(len(px), 1, dtype=torch.bool), dead[:, :-1]], dim=1)      # pairs the loop reaches
        w = torch.where(keep & ~dead, ae * Texcl, torch.zeros_like(ae))
        noise = GAMMA * M
        relT = torch.cumsum(torch.where(keep, ae / (1.0 - ae) * noise, torch.zeros_like(ae)), dim=1)
        near = evaluated & (((alpha - 1.0 / 255.0).abs() <= 2.0 * alpha * noise + 1e-12) & (power <= noise)
                            | ((power.abs() <= noise) & (alpha >= 0.5 / 255.0))
                            | (keep & ((Tincl - 1e-4).abs() <= Tincl * (relT + 1e-6) + 1e-12)))
        # Round 5: a flipped SIGN TEST of the exponent (power > 0: the entry is skipped) is not an alpha-quantum event -- the entry sits at the splat's
        # centre line, alpha = opacity * exp(~0) can be anything up to 0.99.  Only needles reach it (|power| within the cancellation noise of its terms);
        # the jump such a flip may cause at the pixel is bounded by sum alpha_i T_i over the pairs whose sign is within noise (seed 71, frame 78).
        near_sign = evaluated & (power.abs() <= noise) & (alpha >= 0.5 / 255.0)
        J[y0:y1, x0:x1] = torch.where(near_sign, alpha * Texcl, torch.zeros_like(alpha)).sum(dim=1).reshape(y1 - y0, x1 - x0)
        B[y0:y1, x0:x1] = (w * M).sum(dim=1).reshape(y1 - y0, x1 - x0)
        E[y0:y1, x0:x1] = torch.where(evaluated, relT, torch.zeros_like(relT)).amax(dim=1).reshape(y1 - y0, x1 - x0)
        flag[y0:y1, x0:x1] = near.any(dim=1).reshape(y1 - y0, x1 - x0)
    return B, E, flag, J


def threshold_pixels(f):
    """[H, W] bool: the pixels of frame `f` where one of the blend's hard thresholds (alpha vs 1/255, power vs 0, T vs 1e-4) lies inside fp32's rounding
    noise -- the oracle's own `fragile` flag or the conditioning flag above.  There either branch is a correct fp32 result, for the image AND for the
    gradients the pixel feeds (tools/gpu_fuzz_render.py checks a failing frame again with the loss weights zeroed at these pixels)."""
    sc, s = f.sc, f.s
    with torch.no_grad():
        out = O.rasterize(sc.means3D, None, sc.opacities, s, shs=None if f.colors_form else sc.shs, colors_precomp=f.colors,
                          scales=None if f.cov_form else sc.scales, rotations=None if f.cov_form else sc.rotations, cov3D_precomp=f.cov,
                          want_fragile=True, return_aux=True)
    aux = out[-1]
    return aux["fragile"] | conditioning(aux, s)[2]
