"""Adaptive density control next to the operator: the clone / split / prune step the reference runs
every `densification_interval` iterations (train.py:164-174 -> scene/gaussian_model.py:409-469) and the optimizer-state
surgery that goes with it (gaussian_model.py:316-405), as ONE repack.

The reference applies the step as a chain of boolean-mask selections and concatenations -- clone, then split on the
grown set, then two prunes -- each of which re-creates all six parameter tensors and both Adam moments (several dozen
masked copies per call).  Every decision depends only on per-Gaussian quantities of the ORIGINAL set, so the whole step
collapses into one source-index list

    out = [ originals that are neither split nor pruned | clones | split children (N per parent) ]   (pruned ones dropped)

and every tensor (6 parameters, 12 moment tensors) is produced by a single `index_select` with that list; only the split
children's positions and scales are patched afterwards.  Result and ordering are those of the reference
(tests/test_densify_cpu.py runs the reference's own `densify_and_prune` beside it in the build container).

Works on any optimizer with torch.optim.Adam's layout -- one tensor per param group, groups named
xyz / f_dc / f_rest / opacity / scaling / rotation, state keys exp_avg / exp_avg_sq -- i.e. torch.optim.Adam,
gsr_optim.FusedAdam and diff_gaussian_rasterization.SparseGaussianAdam."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn as nn

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


@dataclass
class DensifyStats:
    """Running statistics between two densification steps (GaussianModel.xyz_gradient_accum / denom / max_radii2D)."""
    xyz_gradient_accum: torch.Tensor      # [P,1]
    denom: torch.Tensor                   # [P,1]
    max_radii2D: torch.Tensor             # [P]

    @staticmethod
    def zeros(P: int, device) -> "DensifyStats":
        return DensifyStats(torch.zeros(P, 1, device=device), torch.zeros(P, 1, device=device), torch.zeros(P, device=device))

    def add(self, viewspace_grad: torch.Tensor, visible: torch.Tensor, radii: Optional[torch.Tensor] = None) -> None:
        """add_densification_stats (gaussian_model.py:471-473) + the max_radii2D update of train.py:166:
        accumulate the norm of the screen-space position gradient (the operator's dL/dmeans2D, x and y) of the visible
        Gaussians."""
        g = viewspace_grad
        P = int(g.shape[0]) if g.dim() == 2 else -1
        acc = (self.xyz_gradient_accum, self.denom, self.max_radii2D)
        if g.is_cuda and visible.dtype == torch.int64 and P > 0:
            # train.py:129,166 hands over `(radii > 0).nonzero()`, a [V, 1] list of DISTINCT row indices: the same rows as a mask
            # (two small kernels), so that the unchanged caller's argument reaches the one-pass form below as well
            mask = torch.zeros(P, dtype=torch.bool, device=g.device)
            mask[visible.reshape(-1)] = True
            visible = mask
        # the HIP pass reads P mask bytes and P rows of every array: a MASK of exactly P elements (bool / uint8) on the gradient's
        # device, accumulators of P rows -- an index tensor, a shorter mask or a tensor on another device takes the torch expression
        if (g.is_cuda and g.dtype == torch.float32 and g.dim() == 2 and g.shape[1] == 3 and g.is_contiguous()
                and visible.dtype in (torch.bool, torch.uint8) and visible.numel() == P and visible.device == g.device
                and (radii is None or (radii.dtype == torch.int32 and radii.is_contiguous() and radii.numel() == P and radii.device == g.device))
                and all(t.is_contiguous() and t.dtype == torch.float32 and t.device == g.device and t.shape[0] == P for t in acc)
                and self.xyz_gradient_accum.numel() == P and self.denom.numel() == P and self.max_radii2D.numel() == P):
            # one HIP pass (gsr_density_stats): the boolean-mask form below is ~25 kernels and three host synchronisations per
            # training iteration (0.9 ms at 1 M Gaussians)
            import ctypes as C
            from diff_gaussian_rasterization import _lib
            lib = _lib.load()
            vis = visible.contiguous()
            vis = vis.view(torch.uint8) if vis.dtype == torch.bool else vis
            p = lambda t: None if t is None else C.c_void_p(t.data_ptr())      # noqa: E731
            with torch.cuda.device(g.device):
                _lib.check(lib.gsr_density_stats(int(g.shape[0]), p(g), p(vis), p(radii), p(self.xyz_gradient_accum), p(self.denom),
                                                 p(self.max_radii2D) if radii is not None else None,
                                                 C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)), "gsr_density_stats")
            return
        # CPU tensors / other layouts: the reference's own expression (tests/test_densify_cpu.py runs it beside the reference)
        self.xyz_gradient_accum[visible] += torch.norm(viewspace_grad[visible, :2], dim=-1, keepdim=True)
        self.denom[visible] += 1
        if radii is not None:
            self.max_radii2D[visible] = torch.max(self.max_radii2D[visible], radii[visible].to(self.max_radii2D.dtype))


def quaternion_to_rotation(q: torch.Tensor) -> torch.Tensor:
    """[N,4] (w,x,y,z; normalised here) -> [N,3,3], the convention of utils/general_utils.py:78-100."""
    # explicit left-to-right sum of squares: bit-compatible with the reference's normalisation (a fused vector norm rounds differently)
    n = torch.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])
    q = q / n[:, None]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)


def _groups(optimizer) -> Dict[str, dict]:
    out = {}
    for g in optimizer.param_groups:
        assert len(g["params"]) == 1, "one tensor per param group (scene/gaussian_model.py:183-190)"
        out[g["name"]] = g
    missing = [n for n in GROUPS if n not in out]
    assert not missing, f"param groups {missing} not found"
    return out


def _repack(optimizer, src: torch.Tensor, n_kept: int, patches: Dict[str, torch.Tensor]) -> Dict[str, nn.Parameter]:
    """Rebuild every parameter (and its Adam moments) as tensor[src]; rows [n_kept:] are new Gaussians: their moments
    are zero (cat_tensors_to_optimizer) and `patches[name]` overwrites their values where given."""
    new_params = {}
    for name, g in _groups(optimizer).items():
        old = g["params"][0]
        data = old.detach().index_select(0, src)
        if name in patches:
            data[n_kept:][patches[name + "_rows"]] = patches[name]
        state = optimizer.state.get(old, None)
        p = nn.Parameter(data.requires_grad_(True))
        if state is not None:
            del optimizer.state[old]
            for k in ("exp_avg", "exp_avg_sq"):
                if k in state:
                    m = state[k].index_select(0, src)
                    m[n_kept:] = 0
                    state[k] = m
            optimizer.state[p] = state
        g["params"][0] = p
        new_params[name] = p
    return new_params


@torch.no_grad()
def densify_and_prune(optimizer, stats: DensifyStats, max_grad: float, min_opacity: float, extent: float,
                      max_screen_size: Optional[float] = None, percent_dense: float = 0.01, n_split: int = 2,
                      scaling_activation=torch.exp, scaling_inverse_activation=torch.log,
                      opacity_activation=torch.sigmoid, radii: Optional[torch.Tensor] = None):
    """The reference's densify_and_prune(max_grad, min_opacity, extent, max_screen_size, radii) as one repack.
    Returns (params: name -> nn.Parameter (also installed in the optimizer), new DensifyStats (zeros), tmp_radii or None).

    Decisions (all on the original set, gaussian_model.py:409-469):
      grad_i   = xyz_gradient_accum_i / denom_i (NaN -> 0);  big_i = max(scale_i) > percent_dense * extent
      clone_i  = |grad_i| >= max_grad and not big_i          (copy appended, parent kept)
      split_i  = grad_i  >= max_grad and big_i               (n_split children sampled from the Gaussian, parent dropped)
      pruned   : opacity < min_opacity, or max(scale) > 0.1 * extent when max_screen_size is given -- evaluated on the
                 FINAL set, i.e. with the children's reduced scales (the reference's screen-size test reads max_radii2D
                 after densification_postfix has zeroed it, so it never fires; kept that way)."""
    grp = _groups(optimizer)
    xyz, opacity_raw = grp["xyz"]["params"][0].detach(), grp["opacity"]["params"][0].detach()
    scaling_raw, rotation = grp["scaling"]["params"][0].detach(), grp["rotation"]["params"][0].detach()
    P, dev = xyz.shape[0], xyz.device
    grads = stats.xyz_gradient_accum / stats.denom
    grads[grads.isnan()] = 0.0
    gnorm = torch.norm(grads, dim=-1)
    scale = scaling_activation(scaling_raw)
    smax = scale.max(dim=1).values
    big = smax > percent_dense * extent
    hot = gnorm >= max_grad
    clone = hot & ~big
    split = (grads.squeeze(-1) >= max_grad) & big

    # children first (the only random part; same call shape as the reference so that a shared seed gives the same draw)
    idx_split = torch.nonzero(split).squeeze(-1)
    par = idx_split.repeat(n_split)
    stds = scale[idx_split].repeat(n_split, 1)
    samples = torch.normal(mean=torch.zeros_like(stds), std=stds)
    rots = quaternion_to_rotation(rotation[idx_split]).repeat(n_split, 1, 1)
    child_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + xyz[par]
    child_scale = scale[par] / (0.8 * n_split)
    child_scaling_raw = scaling_inverse_activation(child_scale)

    # prune masks on the final set, evaluated per source row
    low = (opacity_activation(opacity_raw) < min_opacity).squeeze(-1)
    use_ws = bool(max_screen_size)
    drop_orig = low | (smax > 0.1 * extent if use_ws else torch.zeros_like(low))
    keep_orig = ~split & ~drop_orig
    keep_clone = clone & ~drop_orig                      # a clone is an exact copy of its parent
    keep_child = ~low[par]
    if use_ws:
        # the reference tests exp(log(child_scale)): round-trip the activation so that borderline children decide the same way
        keep_child &= ~(scaling_activation(child_scaling_raw).max(dim=1).values > 0.1 * extent)

    idx_orig = torch.nonzero(keep_orig).squeeze(-1)
    idx_clone = torch.nonzero(keep_clone).squeeze(-1)
    child_rows = torch.nonzero(keep_child).squeeze(-1)
    src = torch.cat([idx_orig, idx_clone, par[child_rows]])
    n_kept, n_clone = idx_orig.numel(), idx_clone.numel()
    rows = torch.arange(n_clone, n_clone + child_rows.numel(), device=dev)       # children inside the "new" block
    patches = {"xyz": child_xyz[child_rows], "xyz_rows": rows, "scaling": child_scaling_raw[child_rows], "scaling_rows": rows}
    params = _repack(optimizer, src, n_kept, patches)
    tmp = radii.index_select(0, src) if radii is not None else None
    return params, DensifyStats.zeros(src.numel(), dev), tmp


@torch.no_grad()
def reset_opacity(optimizer, cap: float = 0.01, opacity_activation=torch.sigmoid,
                  inverse_opacity_activation=lambda p: torch.log(p / (1 - p))) -> nn.Parameter:
    """reset_opacity (gaussian_model.py:258-261) + replace_tensor_to_optimizer: opacity <- min(opacity, cap), moments zeroed."""
    g = _groups(optimizer)["opacity"]
    old = g["params"][0]
    new = inverse_opacity_activation(torch.min(opacity_activation(old.detach()), torch.full_like(old, cap)))
    p = nn.Parameter(new.requires_grad_(True))
    state = optimizer.state.get(old, None)
    if state is not None:
        del optimizer.state[old]
        state["exp_avg"] = torch.zeros_like(new)
        state["exp_avg_sq"] = torch.zeros_like(new)
        optimizer.state[p] = state
    g["params"][0] = p
    return p


def attach(gaussians, fused_adam: bool = True):
    """Binds this module's density control to an instance of the reference's `scene.gaussian_model.GaussianModel` (duck-typed: any object with
    its attributes), so that the UNCHANGED caller reaches it through the method calls it already makes (train.py:164-174):

        gaussians = GaussianModel(dataset.sh_degree, opt.optimizer_type)      # train.py:36
        ...
        gaussians.training_setup(opt)                                          # train.py:38
        gsr_scene.densify.attach(gaussians)                                    # <- the one added line

    Replaced on the instance (the class is not touched):
      add_densification_stats(viewspace_point_tensor, update_filter)   gaussian_model.py:471-473 -> one HIP pass (gsr_density_stats) on a HIP device
      densify_and_prune(max_grad, min_opacity, extent, max_screen_size, radii)   gaussian_model.py:448-469 -> the single repack above; the model's
          parameter attributes, its three statistics arrays and `tmp_radii` are left exactly as the reference leaves them
      reset_opacity()                                                  gaussian_model.py:258-261
    fused_adam (default): when `gaussians.optimizer` is a plain `torch.optim.Adam` over HIP tensors -- train.py's default `optimizer_type`
    (scene/gaussian_model.py:193-199) -- it is replaced by `gsr_optim.FusedAdam` with the SAME param groups and state (one HIP kernel for the
    six tensors instead of torch's foreach chain; same state-dict layout, so `capture()` / `restore()` and the density control keep working).
    Returns the instance."""
    import types
    opt = getattr(gaussians, "optimizer", None)
    if fused_adam and type(opt) is torch.optim.Adam and all(p.is_cuda and p.dtype == torch.float32 for g in opt.param_groups for p in g["params"]) \
            and not any(g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False) for g in opt.param_groups):
        from gsr_optim import FusedAdam
        new_opt = FusedAdam(opt.param_groups, lr=opt.defaults["lr"], betas=opt.defaults["betas"], eps=opt.defaults["eps"])
        for p_, st_ in opt.state.items():      # (the same tensors: exp_avg / exp_avg_sq / step, torch.optim.Adam's layout)
            new_opt.state[p_] = {"step": int(st_["step"]) if "step" in st_ else 0, "exp_avg": st_["exp_avg"], "exp_avg_sq": st_["exp_avg_sq"]}
        gaussians.optimizer = new_opt

    def _stats(self):
        return DensifyStats(self.xyz_gradient_accum, self.denom, self.max_radii2D)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        _stats(self).add(viewspace_point_tensor.grad, update_filter, None)      # (max_radii2D is updated by the caller itself, train.py:166)

    def densify_and_prune_(self, max_grad, min_opacity, extent, max_screen_size, radii):
        params, new_stats, _tmp = densify_and_prune(self.optimizer, _stats(self), max_grad, min_opacity, extent, max_screen_size,
                                                    percent_dense=self.percent_dense, scaling_activation=self.scaling_activation,
                                                    scaling_inverse_activation=self.scaling_inverse_activation,
                                                    opacity_activation=self.opacity_activation, radii=radii)
        self._xyz, self._features_dc, self._features_rest = params["xyz"], params["f_dc"], params["f_rest"]
        self._opacity, self._scaling, self._rotation = params["opacity"], params["scaling"], params["rotation"]
        self.xyz_gradient_accum, self.denom, self.max_radii2D = new_stats.xyz_gradient_accum, new_stats.denom, new_stats.max_radii2D
        self.tmp_radii = None
        if self._xyz.is_cuda:
            torch.cuda.empty_cache()

    def reset_opacity_(self):
        self._opacity = reset_opacity(self.optimizer, 0.01, self.opacity_activation, self.inverse_opacity_activation)

    gaussians.add_densification_stats = types.MethodType(add_densification_stats, gaussians)
    gaussians.densify_and_prune = types.MethodType(densify_and_prune_, gaussians)
    gaussians.reset_opacity = types.MethodType(reset_opacity_, gaussians)
    return gaussians
