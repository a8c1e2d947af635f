"""Oracle: the two per-ray regularisers that consume the path's weight lists.  Test infrastructure only.

Restates model_components/losses.py:107-112 (ray_samples_to_sdist), :137-156 (distortion),
:626-651 (_blur_stepfun, _sorted_interp_quad) and :654-705 (zipnerf_interlevel_loss).
These are SURVEY section 8 row f-3 ("next"); they are restated because a training step of the hot
path needs *some* gradient source for the proposal fields and the reference's is this loss.
Inputs are flat: c [B,S+1] s-space bin edges, w [B,S] weights.
"""
import torch

PULSE_WIDTHS = (0.03, 0.003)  # losses.py:660


def lossfun_distortion(t, w):
    """losses.py:137-149."""
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = torch.abs(ut[..., :, None] - ut[..., None, :])
    inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
    intra = torch.sum(w**2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return inter + intra


def distortion_loss(c, w):
    """losses.py:152-157: mean over rays of the mip-NeRF-360 distortion of the final level."""
    return torch.mean(lossfun_distortion(c, w))


def blur_stepfun(x, y, r):
    """losses.py:626-635: convolve a step function with a box of half-width r."""
    xr, idx = torch.sort(torch.cat([x - r, x + r], dim=-1))
    y1 = (torch.cat([y, torch.zeros_like(y[..., :1])], dim=-1)
          - torch.cat([torch.zeros_like(y[..., :1]), y], dim=-1)) / (2 * r)
    y2 = torch.cat([y1, -y1], dim=-1).take_along_dim(idx[..., :-1], dim=-1)
    yr = torch.cumsum((xr[..., 1:] - xr[..., :-1]) * torch.cumsum(y2, dim=-1), dim=-1).clamp_min(0)
    yr = torch.cat([torch.zeros_like(yr[..., :1]), yr], dim=-1)
    return xr, yr


def sorted_interp_quad(x, xp, fpdf, fcdf):
    """losses.py:638-651: piecewise-quadratic CDF interpolation at sorted queries."""
    right = torch.searchsorted(xp, x)
    left = (right - 1).clamp_min(0)
    right = right.clamp_max(xp.shape[-1] - 1)
    xp0, xp1 = xp.take_along_dim(left, dim=-1), xp.take_along_dim(right, dim=-1)
    fpdf0, fpdf1 = fpdf.take_along_dim(left, dim=-1), fpdf.take_along_dim(right, dim=-1)
    fcdf0 = fcdf.take_along_dim(left, dim=-1)
    offset = torch.clip(torch.nan_to_num((x - xp0) / (xp1 - xp0), 0), 0, 1)
    return fcdf0 + (x - xp0) * (fpdf0 + fpdf1 * offset + fpdf0 * (1 - offset)) * 0.5


def interlevel_targets(c, w, pulse_width):
    """The detached part of zipnerf_interlevel_loss for one proposal level (losses.py:661-694):
    returns (c_, w_, cdf) describing the blurred, normalised final-level histogram."""
    c = c.detach()
    w = w.detach()
    accum = torch.sum(w, dim=-1, keepdim=True)
    w = torch.cat([w[..., :-1], w[..., -1:] + (1 - accum)], dim=-1)
    w_norm = w / (c[..., 1:] - c[..., :-1])
    c_, w_ = blur_stepfun(c, w_norm, pulse_width)
    area = 0.5 * (w_[..., 1:] + w_[..., :-1]) * (c_[..., 1:] - c_[..., :-1])
    cdf = torch.cat([torch.zeros_like(area[..., :1]), torch.cumsum(area, dim=-1)], dim=-1)
    c_ = torch.cat([torch.zeros_like(c_[..., :1]), c_, torch.ones_like(c_[..., :1])], dim=-1)
    w_ = torch.cat([torch.zeros_like(w_[..., :1]), w_, torch.zeros_like(w_[..., :1])], dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf, torch.ones_like(cdf[..., :1])], dim=-1)
    return c_, w_, cdf


def zipnerf_interlevel_loss(c_list, w_list):
    """losses.py:654-705.  c_list/w_list: proposal levels first, final level last.  Gradient only
    reaches the proposal weights (the final level is detached, :661-662)."""
    loss = 0
    for i, (cp, wp) in enumerate(zip(c_list[:-1], w_list[:-1])):
        c_, w_, cdf = interlevel_targets(c_list[-1], w_list[-1], PULSE_WIDTHS[i])
        w_s = torch.diff(sorted_interp_quad(cp, c_, w_, cdf), dim=-1)
        loss = loss + ((w_s - wp).clamp_min(0) ** 2 / (wp + 1e-5)).sum(dim=-1).mean()
    return loss
