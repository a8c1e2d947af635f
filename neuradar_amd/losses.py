"""The two per-ray regularisers that consume the path's weight lists (reference:
model_components/losses.py:107-112,137-156,626-705).  The inter-level loss is
the ONLY thing that trains the proposal fields.

`distortion_loss` / `zipnerf_interlevel_loss` run the HIP kernels (nr_distortion_loss,
nr_interlevel_loss: value and gradient in one pass each) behind torch.autograd; the `*_torch` versions
are the op-by-op restatement of the reference they are tested against (~200 launches per step).

Flat inputs: `c` = s-space bin edges [B,S+1], `w` = weights [B,S].
"""
from typing import List

import torch
from torch import Tensor

from . import _lib, ops

PULSE_WIDTHS = (0.03, 0.003)  # losses.py:660


class _Distortion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, w):
        slots = torch.zeros(_lib.NR_LOSS_SLOTS, device=w.device, dtype=torch.float32)
        g_w = ops.distortion_loss(c.contiguous(), w.contiguous(), w.shape[1], 1.0, slots)
        ctx.save_for_backward(g_w)
        return slots.sum()

    @staticmethod
    def backward(ctx, g):
        (g_w,) = ctx.saved_tensors
        return None, g_w * g


class _Interlevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, w, cp, wp, pulse):
        slots = torch.zeros(_lib.NR_LOSS_SLOTS, device=wp.device, dtype=torch.float32)
        g_wp = ops.interlevel_loss(c.contiguous(), w.contiguous(), w.shape[1], cp.contiguous(), wp.contiguous(), pulse, 1.0, slots)
        ctx.save_for_backward(g_wp)
        return slots.sum()

    @staticmethod
    def backward(ctx, g):
        (g_wp,) = ctx.saved_tensors
        return None, None, None, g_wp * g, None


def distortion_loss(c: Tensor, w: Tensor) -> Tensor:
    """mip-NeRF 360 distortion of the final level, mean over rays (losses.py:137-157): one launch."""
    return _Distortion.apply(c.detach(), w)


def zipnerf_interlevel_loss(c_list: List[Tensor], w_list: List[Tensor]) -> Tensor:
    """Anti-aliased inter-level loss (losses.py:654-705); proposal levels first, final level last: one
    launch per proposal level.  The final level is detached in the reference (:661-662), and so is every
    edge position (the samplers detach them, ray_samplers.py:364)."""
    c, w = c_list[-1].detach(), w_list[-1].detach()
    loss = 0
    for i, (cp, wp) in enumerate(zip(c_list[:-1], w_list[:-1])):
        loss = loss + _Interlevel.apply(c, w, cp.detach(), wp, PULSE_WIDTHS[i])
    return loss


def distortion_loss_torch(c: Tensor, w: Tensor) -> Tensor:
    """mip-NeRF 360 distortion of the final level, mean over rays (losses.py:137-157)."""
    mid = (c[..., 1:] + c[..., :-1]) / 2
    pair = torch.abs(mid[..., :, None] - mid[..., None, :])
    inter = torch.sum(w * torch.sum(w[..., None, :] * pair, dim=-1), dim=-1)
    intra = torch.sum(w**2 * (c[..., 1:] - c[..., :-1]), dim=-1) / 3
    return torch.mean(inter + intra)


@torch.no_grad()
def _resampled_target(c: Tensor, w: Tensor, cp: Tensor, pulse: float) -> Tensor:
    """Blur the final-level histogram (c, w) with a box of half-width `pulse`, integrate it to a
    piecewise-quadratic CDF and read that CDF at the proposal edges `cp` (losses.py:626-699).
    Entirely gradient-free: the final level is detached in the reference (:661-662)."""
    acc = torch.sum(w, dim=-1, keepdim=True)
    w = torch.cat([w[..., :-1], w[..., -1:] + (1 - acc)], dim=-1)
    dens = w / (c[..., 1:] - c[..., :-1])
    # box blur of a step function = piecewise-linear function with knots at c -+ pulse
    knots, order = torch.sort(torch.cat([c - pulse, c + pulse], dim=-1))
    zero = torch.zeros_like(dens[..., :1])
    jump = (torch.cat([dens, zero], dim=-1) - torch.cat([zero, dens], dim=-1)) / (2 * pulse)
    slope = torch.cat([jump, -jump], dim=-1).take_along_dim(order[..., :-1], dim=-1)
    vals = torch.cumsum((knots[..., 1:] - knots[..., :-1]) * torch.cumsum(slope, dim=-1), dim=-1).clamp_min(0)
    vals = torch.cat([torch.zeros_like(vals[..., :1]), vals], dim=-1)
    area = 0.5 * (vals[..., 1:] + vals[..., :-1]) * (knots[..., 1:] - knots[..., :-1])
    cdf = torch.cat([torch.zeros_like(area[..., :1]), torch.cumsum(area, dim=-1)], dim=-1)
    z, o = torch.zeros_like(knots[..., :1]), torch.ones_like(knots[..., :1])
    knots, vals, cdf = torch.cat([z, knots, o], -1), torch.cat([z, vals, z], -1), torch.cat([z, cdf, o], -1)
    right = torch.searchsorted(knots, cp.contiguous())
    left = (right - 1).clamp_min(0)
    right = right.clamp_max(knots.shape[-1] - 1)
    x0, x1 = knots.take_along_dim(left, -1), knots.take_along_dim(right, -1)
    v0, v1 = vals.take_along_dim(left, -1), vals.take_along_dim(right, -1)
    f0 = cdf.take_along_dim(left, -1)
    t = torch.clip(torch.nan_to_num((cp - x0) / (x1 - x0), 0), 0, 1)
    cdf_at = f0 + (cp - x0) * (v0 + v1 * t + v0 * (1 - t)) * 0.5
    return torch.diff(cdf_at, dim=-1)


def zipnerf_interlevel_loss_torch(c_list: List[Tensor], w_list: List[Tensor]) -> Tensor:
    """Anti-aliased inter-level loss (losses.py:654-705); proposal levels first, final level last."""
    c, w = c_list[-1].detach(), w_list[-1].detach()
    loss = 0
    for i, (cp, wp) in enumerate(zip(c_list[:-1], w_list[:-1])):
        target = _resampled_target(c, w, cp.detach(), PULSE_WIDTHS[i])
        loss = loss + ((target - wp).clamp_min(0) ** 2 / (wp + 1e-5)).sum(dim=-1).mean()
    return loss
