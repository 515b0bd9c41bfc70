"""Seeded synthetic scenes + camera helpers for tests and bench.py.

No dataset or trained model ships with this repo (there is no network), so every
config in BASELINE.json gets a statistically similar synthetic stand-in
(SURVEY.md section 8(d)).  The generator is frozen: (seed, P, W, H, s_med) fully
determine a scene.  Everything is produced on the CPU with torch's default
generator and moved to the requested device afterwards, so the CPU oracle and
the GPU path see bit-identical inputs.

Camera conventions follow the reference exactly:
  * world_view_transform = W2C transposed            (scene/cameras.py:86)
  * full_proj_transform  = W2C^T @ P^T               (scene/cameras.py:87-88)
  * camera_center        = inverse(W2C^T)[3, :3]     (scene/cameras.py:89)
  * P from getProjectionMatrix(znear=0.01, zfar=100) (utils/graphics_utils.py:51-71)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import torch


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """OpenGL-style perspective matrix with z_sign=+1, same entries as
    utils/graphics_utils.py:51-71 (getProjectionMatrix)."""
    tan_y = math.tan(fovy / 2)
    tan_x = math.tan(fovx / 2)
    top = tan_y * znear
    bottom = -top
    right = tan_x * znear
    left = -right
    P = torch.zeros(4, 4, dtype=torch.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class Camera:
    """The per-view quantities gaussian_renderer.render() hands to the rasterizer
    (gaussian_renderer/__init__.py:33-50)."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # [4,4] W2C^T
    full_proj_transform: torch.Tensor    # [4,4]
    camera_center: torch.Tensor          # [3]
    znear: float = 0.01
    zfar: float = 100.0

    @property
    def tanfovx(self) -> float:
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self) -> float:
        return math.tan(self.FoVy * 0.5)

    def to(self, device) -> "Camera":
        return Camera(self.image_width, self.image_height, self.FoVx, self.FoVy,
                      self.world_view_transform.to(device), self.full_proj_transform.to(device),
                      self.camera_center.to(device), self.znear, self.zfar)


def make_camera(width: int, height: int, fovx_deg: float = 60.0,
                R: Optional[torch.Tensor] = None, t: Optional[torch.Tensor] = None) -> Camera:
    """Camera with world->camera rotation R (3x3, x_cam = R x_world + t).  Default: identity
    at the origin looking down +z.  FoVy follows from square pixels: tanfovy = tanfovx*H/W."""
    fovx = math.radians(fovx_deg)
    tanx = math.tan(fovx / 2)
    tany = tanx * height / width
    fovy = 2.0 * math.atan(tany)
    w2c = torch.eye(4, dtype=torch.float32)
    if R is not None:
        w2c[:3, :3] = R.to(torch.float32)
    if t is not None:
        w2c[:3, 3] = t.to(torch.float32)
    wvt = w2c.transpose(0, 1).contiguous()
    proj = projection_matrix(0.01, 100.0, fovx, fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return Camera(width, height, fovx, fovy, wvt, full, center)


def look_at_camera(width: int, height: int, eye, target, up=(0.0, -1.0, 0.0), fovx_deg: float = 60.0) -> Camera:
    """Non-trivial camera (rotated + translated) for tests that must exercise the
    full view/projection matrix layout, not just identity."""
    eye = torch.tensor(eye, dtype=torch.float64)
    target = torch.tensor(target, dtype=torch.float64)
    upv = torch.tensor(up, dtype=torch.float64)
    zc = target - eye
    zc = zc / zc.norm()
    xc = torch.linalg.cross(upv, zc)
    xc = xc / xc.norm()
    yc = torch.linalg.cross(zc, xc)
    R = torch.stack([xc, yc, zc], dim=0)          # rows = camera axes in world coords
    t = -(R @ eye)
    return make_camera(width, height, fovx_deg, R.to(torch.float32), t.to(torch.float32))


@dataclass
class Scene:
    means3D: torch.Tensor      # [P,3]
    scales: torch.Tensor       # [P,3]  post-exp   (scene/gaussian_model.py:103-104)
    rotations: torch.Tensor    # [P,4]  normalised, w first (gaussian_model.py:107-108)
    opacities: torch.Tensor    # [P,1]  post-sigmoid (gaussian_model.py:129-130)
    shs: torch.Tensor          # [P,M,3] DC || rest along dim 1 (gaussian_model.py:115-118)
    sh_degree: int
    meta: dict = field(default_factory=dict)

    @property
    def P(self) -> int:
        return self.means3D.shape[0]

    def to(self, device) -> "Scene":
        return Scene(self.means3D.to(device), self.scales.to(device), self.rotations.to(device),
                     self.opacities.to(device), self.shs.to(device), self.sh_degree, dict(self.meta))


def make_scene(P: int, cam: Camera, seed: int = 0, s_med: float = 0.012, overscan: float = 1.1,
               z_min: float = 2.0, z_max: float = 12.0, sigma_log_scale: float = 0.6,
               sh_degree: int = 3, max_sh_degree: int = 3) -> Scene:
    """SURVEY.md 8(d) recipe.  Points are generated in CAMERA space of `cam` and mapped back to
    world space, so the recipe gives the same screen statistics for any camera pose."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    f32 = torch.float32
    z = torch.rand(P, generator=g, dtype=f32) * (z_max - z_min) + z_min
    u = torch.rand(P, generator=g, dtype=f32) * 2 - 1
    v = torch.rand(P, generator=g, dtype=f32) * 2 - 1
    xc = u * z * (cam.tanfovx * overscan)
    yc = v * z * (cam.tanfovy * overscan)
    pc = torch.stack([xc, yc, z], dim=1)
    # camera -> world: x_w = R^T (x_c - t); W2C^T stored row-major => W2C = wvt^T
    w2c = cam.world_view_transform.transpose(0, 1).to(f32)
    Rm, t = w2c[:3, :3], w2c[:3, 3]
    means = (pc - t[None, :]) @ Rm            # (R^T (x_c - t))^T = (x_c - t)^T R
    log_s = torch.randn(P, 3, generator=g, dtype=f32) * sigma_log_scale + math.log(s_med)
    scales = torch.exp(log_s)
    q = torch.randn(P, 4, generator=g, dtype=f32)
    q = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g, dtype=f32) * 1.5)
    M = (max_sh_degree + 1) ** 2
    shs = torch.randn(P, M, 3, generator=g, dtype=f32) * 0.1
    shs[:, 0, :] = torch.randn(P, 3, generator=g, dtype=f32) * 0.5
    return Scene(means.contiguous(), scales.contiguous(), q.contiguous(), opac.contiguous(), shs.contiguous(),
                 sh_degree, {"seed": seed, "P": P, "s_med": s_med, "overscan": overscan,
                             "W": cam.image_width, "H": cam.image_height})


def make_edge_scene(P: int, cam: Camera, seed: int = 1) -> Scene:
    """SURVEY.md 8(c) edge-case scene: 1.6 overscan (reaches the 1.3*tanfov clamp), x4 scales with
    sigma 1.0 (huge / anisotropic splats), z straddling the 0.2 near cull, opacities below 1/255,
    near-opaque stacks (T < 1e-4 termination) and duplicated depths (tie order)."""
    sc = make_scene(P, cam, seed=seed, s_med=0.048, overscan=1.6, z_min=0.1, z_max=8.0, sigma_log_scale=1.0)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 1000)
    n = P // 8
    if n > 0:
        # (a) opacities below the 1/255 skip threshold
        sc.opacities[:n] = torch.rand(n, 1, generator=g) * (1.0 / 255.0)
        # (b) near-opaque stack in front of the camera centre to trigger early termination
        w2c = cam.world_view_transform.transpose(0, 1)
        Rm, t = w2c[:3, :3], w2c[:3, 3]
        pc = torch.stack([torch.randn(n, generator=g) * 0.05, torch.randn(n, generator=g) * 0.05,
                          1.0 + torch.rand(n, generator=g) * 0.5], dim=1)
        sc.means3D[n:2 * n] = (pc - t[None, :]) @ Rm
        sc.opacities[n:2 * n] = 0.97 + 0.03 * torch.rand(n, 1, generator=g)
        sc.scales[n:2 * n] = 0.05 + 0.05 * torch.rand(n, 3, generator=g)
        # (c) duplicated positions => bit-identical depths => tie order must follow Gaussian index
        sc.means3D[2 * n:3 * n] = sc.means3D[3 * n:4 * n]
    sc.meta["kind"] = "edge"
    return sc


def make_clustered_scene(P: int, cam: Camera, seed: int = 0, n_clusters: int = 200, s_med: float = 0.012, overscan: float = 1.6,
                         z_min: float = 1.5, z_max: float = 14.0, floaters: float = 0.05, sh_degree: int = 3,
                         max_sh_degree: int = 3) -> Scene:
    """A stand-in that looks more like a TRAINED scene than the i.i.d.-uniform cloud of make_scene (VERDICT r02 missing #6:
    configs name garden / bicycle / truck, every timing was on uniform Gaussians):
      * a mixture of `n_clusters` anisotropic clusters ("objects / surfaces"), log-normal cluster populations (sigma 1.0: the
        largest clusters hold tens of times the median), cluster extents log-normal around 0.35 (sigma 0.7) per axis with one
        axis flattened x0.15 (surface-like sheets), random orientation;
      * cluster centres spread over 1.6x the frustum of `cam` laterally, so only ~40 % of the Gaussians are visible from any
        one view (a trained scene is much larger than one camera's frustum -- this is what SparseGaussianAdam exploits,
        README.md:496);
      * per-Gaussian scales log-normal around s_med x a per-cluster factor (sigma 0.5), flattened x0.2 along one axis;
      * bimodal opacities (sigmoid of N(0, 2.5^2): many nearly transparent, many nearly opaque);
      * `floaters` (5 %): isolated Gaussians anywhere in the volume with 4-12x the scale and low opacity -- the large
        semi-transparent splats a real optimisation leaves behind, which touch dozens to hundreds of tiles each.
    Same frozen-generator contract as make_scene: (seed, P, camera, arguments) determine the scene."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    f32 = torch.float32
    n_float = int(P * floaters)
    n_clu = P - n_float
    # cluster populations
    wgt = torch.exp(torch.randn(n_clusters, generator=g, dtype=torch.float64) * 1.0)
    owner = torch.multinomial(wgt / wgt.sum(), n_clu, replacement=True, generator=g)
    # cluster frames in the camera space of `cam`
    cz = torch.rand(n_clusters, generator=g, dtype=f32) * (z_max - z_min) + z_min
    cu = torch.rand(n_clusters, generator=g, dtype=f32) * 2 - 1
    cv = torch.rand(n_clusters, generator=g, dtype=f32) * 2 - 1
    centre = torch.stack([cu * cz * cam.tanfovx * overscan, cv * cz * cam.tanfovy * overscan, cz], dim=1)
    ext = torch.exp(torch.randn(n_clusters, 3, generator=g, dtype=f32) * 0.7 + math.log(0.35))
    flat_axis = torch.randint(0, 3, (n_clusters,), generator=g)
    ext[torch.arange(n_clusters), flat_axis] *= 0.15
    cq = torch.randn(n_clusters, 4, generator=g, dtype=f32)
    cq = cq / cq.norm(dim=1, keepdim=True)
    r, x, y, z = cq[:, 0], cq[:, 1], cq[:, 2], cq[:, 3]
    Rc = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                      2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                      2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).view(n_clusters, 3, 3)
    local = torch.randn(n_clu, 3, generator=g, dtype=f32) * ext[owner]
    pc = centre[owner] + torch.bmm(Rc[owner], local.unsqueeze(2)).squeeze(2)
    cscale = torch.exp(torch.randn(n_clusters, generator=g, dtype=f32) * 0.5)
    log_s = torch.randn(n_clu, 3, generator=g, dtype=f32) * 0.6 + math.log(s_med) + torch.log(cscale[owner])[:, None]
    thin = torch.randint(0, 3, (n_clu,), generator=g)
    log_s[torch.arange(n_clu), thin] += math.log(0.2)
    opac_logit = torch.randn(n_clu, 1, generator=g, dtype=f32) * 2.5
    # floaters
    fz = torch.rand(n_float, generator=g, dtype=f32) * (z_max - z_min) + z_min
    fu = torch.rand(n_float, generator=g, dtype=f32) * 2 - 1
    fv = torch.rand(n_float, generator=g, dtype=f32) * 2 - 1
    fpc = torch.stack([fu * fz * cam.tanfovx * overscan, fv * fz * cam.tanfovy * overscan, fz], dim=1)
    flog_s = torch.randn(n_float, 3, generator=g, dtype=f32) * 0.5 + math.log(s_med) + \
        torch.log(4.0 + 8.0 * torch.rand(n_float, 1, generator=g, dtype=f32))
    fop_logit = torch.randn(n_float, 1, generator=g, dtype=f32) * 1.0 - 2.0
    pc = torch.cat([pc, fpc], dim=0)
    log_s = torch.cat([log_s, flog_s], dim=0)
    opac = torch.sigmoid(torch.cat([opac_logit, fop_logit], dim=0))
    # interleave floaters and cluster members (a trained model's order carries no structure)
    perm = torch.randperm(P, generator=g)
    pc, log_s, opac = pc[perm], log_s[perm], opac[perm]
    w2c = cam.world_view_transform.transpose(0, 1).to(f32)
    Rm, t = w2c[:3, :3], w2c[:3, 3]
    means = (pc - t[None, :]) @ Rm
    q = torch.randn(P, 4, generator=g, dtype=f32)
    q = q / q.norm(dim=1, keepdim=True)
    M = (max_sh_degree + 1) ** 2
    shs = torch.randn(P, M, 3, generator=g, dtype=f32) * 0.1
    shs[:, 0, :] = torch.randn(P, 3, generator=g, dtype=f32) * 0.5
    return Scene(means.contiguous(), torch.exp(log_s).contiguous(), q.contiguous(), opac.contiguous(), shs.contiguous(), sh_degree,
                 {"seed": seed, "P": P, "s_med": s_med, "overscan": overscan, "kind": "clustered", "n_clusters": n_clusters,
                  "floaters": floaters, "W": cam.image_width, "H": cam.image_height})
