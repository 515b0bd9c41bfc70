"""Screen-tile sharding of the rasterizer across the GPUs of one node (net-new: the reference
has no multi-GPU code).  One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" in the CPU tests).

Partitioning
  * pixel axis: rank g owns a contiguous band of 16-pixel tile ROWS [y0_g, y1_g).  Bands are balanced by the
    per-row instance counts of the previous frame (uniform rows are imbalanced on real scenes), see BandPlan.
  * Gaussian axis: parameters are replicated for the forward (every rank runs the HBM-streaming preprocess on
    all P -- it costs less than all-gathering 48-byte splat records over xGMI at these sizes), and the
    per-Gaussian 2-D gradient record is summed across ranks in the backward.
Exchange steps
  * forward : all_gather of the rendered strips (3*H*W*4 bytes in total, +H*W*4 for the inverse-depth strip);
  * backward: every rank back-propagates only its own band's dL/dpixel, producing dense partial per-Gaussian
    gradients; they are combined with all_reduce(sum) (replicated parameters) or reduce_scatter (sharded
    optimizer state, `owner_shard=True`).
The band renderer is injected (`render_band`) so the index math and collectives are testable on CPU with gloo;
the product passes the HIP rasterizer.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

TILE = 16


@dataclass
class BandPlan:
    """Contiguous tile-row bands, one per rank."""
    bounds: List[int]            # len = world+1, bounds[g]..bounds[g+1]

    @staticmethod
    def uniform(n_rows: int, world: int) -> "BandPlan":
        return BandPlan([(n_rows * g) // world for g in range(world + 1)])

    @staticmethod
    def balanced(row_cost: Sequence[float], world: int) -> "BandPlan":
        """Split rows so that every band carries ~1/world of the total cost (prefix-sum cut points; every band
        keeps at least one row while rows remain)."""
        n = len(row_cost)
        total = float(sum(row_cost))
        if total <= 0 or world == 1:
            return BandPlan.uniform(n, world)
        cum = [0.0]
        for c in row_cost:
            cum.append(cum[-1] + float(c))
        bounds = [0]
        for g in range(1, world):
            target = total * g / world
            r = bounds[-1]
            while r < n and cum[r + 1] <= target:
                r += 1
            # r = last cut with cum[r] <= target; take the nearer of r / r+1
            if r < n and (cum[r + 1] - target) < (target - cum[r]):
                r += 1
            bounds.append(max(r, bounds[-1]))
        bounds.append(n)
        return BandPlan(bounds)

    def band(self, rank: int) -> Tuple[int, int]:
        return self.bounds[rank], self.bounds[rank + 1]

    def pixel_rows(self, rank: int, H: int) -> Tuple[int, int]:
        y0, y1 = self.band(rank)
        return min(y0 * TILE, H), min(y1 * TILE, H)


def gather_strips(local: torch.Tensor, plan: BandPlan, H: int, group=None) -> torch.Tensor:
    """local: [C,H,W] with only this rank's rows valid.  Returns the full [C,H,W] image on every rank.
    Strips have different heights, so each rank contributes a max-height padded strip to one all_gather."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    C, _, W = local.shape
    rows = [plan.pixel_rows(g, H) for g in range(world)]
    hmax = max(b - a for a, b in rows)
    a, b = rows[rank]
    send = local.new_zeros(C, hmax, W)
    send[:, : b - a] = local[:, a:b]
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    full = torch.empty_like(local)
    for g, (ga, gb) in enumerate(rows):
        full[:, ga:gb] = recv[g][:, : gb - ga]
    return full


class _ShardedRaster(torch.autograd.Function):
    """forward: render own band + all_gather strips; backward: mask dL/dpixel to the own band, run the band's
    backward, sum the per-Gaussian gradients across ranks."""

    @staticmethod
    def forward(ctx, render_band, plan, group, n_inputs, *inputs):
        rank = dist.get_rank(group)
        y0, y1 = plan.band(rank)
        detached = [t.detach().requires_grad_(t.requires_grad) if isinstance(t, torch.Tensor) else t for t in inputs]
        with torch.enable_grad():
            color, radii, invdepth = render_band(detached, (y0, y1))
        H = color.shape[1]
        full_color = gather_strips(color.detach(), plan, H, group)
        full_inv = gather_strips(invdepth.detach(), plan, H, group)
        # radii are band-independent (every rank preprocesses all Gaussians)
        ctx.saved = (detached, color, invdepth)
        ctx.plan, ctx.group, ctx.H = plan, group, H
        ctx.mark_non_differentiable(radii)
        return full_color, radii, full_inv

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_inv):
        detached, color, invdepth = ctx.saved
        rank = dist.get_rank(ctx.group)
        a, b = ctx.plan.pixel_rows(rank, ctx.H)
        gc = torch.zeros_like(color)
        gi = torch.zeros_like(invdepth)
        gc[:, a:b] = g_color[:, a:b]
        if g_inv is not None:
            gi[:, a:b] = g_inv[:, a:b]
        diff = [t for t in detached if isinstance(t, torch.Tensor) and t.requires_grad]
        grads = torch.autograd.grad([color, invdepth], diff, [gc, gi], allow_unused=True)
        out = []
        it = iter(grads)
        for t in detached:
            if isinstance(t, torch.Tensor) and t.requires_grad:
                g = next(it)
                if g is None:
                    g = torch.zeros_like(t)
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
                out.append(g)
            else:
                out.append(None)
        return (None, None, None, None, *out)


def render_sharded(render_band: Callable, inputs: Sequence, plan: BandPlan, group=None):
    """render_band(inputs, (y0, y1)) -> (color[3,H,W], radii[P], invdepth[1,H,W]) with only the band's rows valid.
    Returns the full image / radii / inverse depth on every rank; differentiable w.r.t. the tensor inputs."""
    return _ShardedRaster.apply(render_band, plan, group, len(inputs), *inputs)


def row_costs_from_ranges(ranges: torch.Tensor, gx: int, gy: int, group=None) -> List[float]:
    """Per tile-row instance counts for re-balancing.  `ranges` [gx*gy,2] holds valid entries only for this
    rank's band (zeros elsewhere), so a SUM all_reduce yields the global per-row histogram."""
    cnt = (ranges[:, 1] - ranges[:, 0]).to(torch.float32).view(gy, gx).sum(dim=1)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    # every row also costs a fixed amount (pixels to write)
    return (cnt + 256.0 * gx * 0.05).tolist()
