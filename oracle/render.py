"""Oracle: alpha compositing.  Test infrastructure only.

`render_weight_from_alpha` / `accumulate_along_rays` restate the *batched* (packed_info=None,
ray_indices=None) branches of the third-party `nerfacc==0.5.2` (pyproject.toml:36), which is absent
from /root/reference -> PARITY UNPINNED for these two helpers (see oracle/__init__.py); call sites
models/neuradar.py:1016, models/neurad.py:727-728.  The rest restates
model_components/renderers.py:59-90,322-350 and models/neuradar.py:504-517,527-528.
"""
import torch


def render_weight_from_alpha(alphas):
    """nerfacc batched branch: T_i = prod_{j<i}(1 - alpha_j) (exclusive cumprod), w_i = alpha_i T_i.

    alphas [B,S] -> (weights [B,S], transmittance [B,S]).
    """
    one_minus = 1.0 - alphas
    trans = torch.cumprod(torch.cat([torch.ones_like(one_minus[:, :1]), one_minus[:, :-1]], dim=-1), dim=-1)
    return alphas * trans, trans


def render_weight_from_density(t_starts, t_ends, sigmas):
    """nerfacc batched branch (call site models/neuradar.py:1018-1022, use_sdf=False): alpha_i = 1 - exp(-sigma_i *
    (t_end_i - t_start_i)), then render_weight_from_alpha.  Returns (weights, transmittance, alphas)."""
    alphas = 1.0 - torch.exp(-sigmas * (t_ends - t_starts))
    w, trans = render_weight_from_alpha(alphas)
    return w, trans, alphas


def accumulate_along_rays(weights, values=None):
    """nerfacc batched branch: sum_s w[...,s,None] * values[...,s,:]."""
    src = weights[..., None] if values is None else weights[..., None] * values
    return torch.sum(src, dim=-2)


def composite(alpha, feature, starts, ends):
    """models/neuradar.py:504-517: weights, accumulation, sky fix-up, features, depth.

    alpha [B,S,1], feature [B,S,C], starts/ends [B,S]  ->  dict(weights [B,S] BEFORE the sky fix-up
    is dropped i.e. incl. the fixed-up last sample, accumulation [B,1], features [B,C], depth [B,1]).
    """
    weights, _ = render_weight_from_alpha(alpha[..., 0])  # :504,1016
    accumulation = torch.sum(weights[..., None], dim=-2)  # AccumulationRenderer, renderers.py:349
    weights = torch.cat((weights[..., :-1], weights[..., -1:] + 1 - accumulation), dim=-1)  # :508
    features = torch.sum(feature * weights[..., None], dim=-2)  # FeatureRenderer, renderers.py:85
    depth = depth_simple(weights[:, :-1], starts[:, :-1], ends[:, :-1])  # sky sample dropped, :515-517
    return {"weights": weights, "accumulation": accumulation, "features": features, "depth": depth}


def depth_simple(weights, starts, ends):
    """render_depth_simple: sum w * (start+end)/2, NOT normalised.  models/neurad.py:721-728."""
    steps = (starts + ends) / 2
    return accumulate_along_rays(weights, steps[..., None])
