"""Oracle: proposal sampling.  Test infrastructure only.

Restates utils/math.py:541-579 (power transform), model_components/ray_samplers.py:80-132
(SpacedSampler), :838-852 (PowerSampler), :280-376 (PDFSampler), :623-666
(ProposalNetworkSampler) and cameras/rays.py:188-210 (RaySamples.get_weights).

Randomness is INJECTED (jitter tensors are arguments) so parity never depends on RNG streams.
"""
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch

HISTOGRAM_PADDING = 0.01  # PDFSampler default, ray_samplers.py:272
POWER_LAMBDA = -1.0  # models/neuradar.py:133
POWER_SCALING = 0.1  # models/neuradar.py:135
SKY_DISTANCE = 20000.0  # models/neuradar.py:137


def power_fn(x, lam: float = POWER_LAMBDA):
    """ZipNeRF eq.4 power transform, finite-lambda branch.  utils/math.py:541-557."""
    lam_1 = abs(lam - 1)
    return (lam_1 / lam) * ((x / lam_1 + 1) ** lam - 1)


def inv_power_fn(x, lam: float = POWER_LAMBDA, eps: float = 1e-10):
    """Inverse of power_fn, finite-lambda branch.  utils/math.py:560-579."""
    lam_1 = abs(lam - 1)
    return ((x * lam / lam_1 + 1).clamp_min(eps) ** (1 / lam) - 1) * lam_1


@dataclass
class Samples:
    """The fields of `RaySamples` the path uses (cameras/rays.py:142-184), flat tensors.

    spacing [B,S+1] are bin edges in normalised s-space, euclid [B,S+1] the same edges in metres;
    starts = euclid[:, :-1], ends = euclid[:, 1:], deltas = ends - starts (rays.py:330).
    """

    spacing: torch.Tensor
    euclid: torch.Tensor
    s_near: torch.Tensor
    s_far: torch.Tensor

    @property
    def starts(self):
        return self.euclid[:, :-1]

    @property
    def ends(self):
        return self.euclid[:, 1:]

    @property
    def deltas(self):
        return self.euclid[:, 1:] - self.euclid[:, :-1]


def spacing_to_euclidean(s, s_near, s_far, lam=POWER_LAMBDA, scaling=POWER_SCALING):
    """ray_samplers.py:119-120 with PowerSampler's fns (:846-851)."""
    return inv_power_fn(s * s_far + (1 - s) * s_near, lam) / scaling


def power_bins(nears, fars, num_samples: int, t_rand: Optional[torch.Tensor] = None,
               lam=POWER_LAMBDA, scaling=POWER_SCALING) -> Samples:
    """PowerSampler / SpacedSampler.generate_ray_samples.  ray_samplers.py:98-132,838-852.

    nears/fars [B,1].  t_rand is the training jitter: [B,S+1] (PowerSampler never forwards
    `single_jitter`) or None for eval (bin edges at linspace).
    """
    B = nears.shape[0]
    bins = torch.linspace(0.0, 1.0, num_samples + 1, device=nears.device)[None, :]
    if t_rand is not None:
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * t_rand
    else:
        bins = bins.expand(B, -1)
    s_near = power_fn(nears * scaling, lam)
    s_far = power_fn(fars * scaling, lam)
    return Samples(bins, spacing_to_euclidean(bins, s_near, s_far, lam, scaling), s_near, s_far)


def weights_from_density(deltas, density):
    """RaySamples.get_weights.  cameras/rays.py:188-210.  deltas/density [B,S] -> weights [B,S]."""
    dd = deltas * density
    alphas = 1 - torch.exp(-dd)
    acc = torch.cumsum(dd[:, :-1], dim=-1)
    acc = torch.cat([torch.zeros_like(acc[:, :1]), acc], dim=-1)
    return torch.nan_to_num(alphas * torch.exp(-acc))


def pdf_resample(prev: Samples, weights, num_samples: int, jitter: Optional[torch.Tensor] = None,
                 eps: float = 1e-5, lam=POWER_LAMBDA, scaling=POWER_SCALING) -> Samples:
    """PDFSampler.generate_ray_samples (include_original=False).  ray_samplers.py:305-376.

    weights [B,S] over prev's S bins -> num_samples+1 new bin edges by inverse-CDF sampling.
    jitter: [B,1] uniform(0,1) (single_jitter=True, :325-326) for training, None for eval (:332-334).
    The new edges are detached (:364).
    """
    num_bins = num_samples + 1
    w = weights + HISTOGRAM_PADDING
    w_sum = torch.sum(w, dim=-1, keepdim=True)
    padding = torch.relu(eps - w_sum)
    w = w + padding / w.shape[-1]
    w_sum = w_sum + padding
    pdf = w / w_sum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)  # [B,S+1]

    u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins, device=cdf.device)
    if jitter is not None:
        u = u.expand(cdf.shape[0], num_bins) + jitter / num_bins
    else:
        u = (u + 1.0 / (2 * num_bins)).expand(cdf.shape[0], num_bins)
    u = u.contiguous()

    existing = prev.spacing
    inds = torch.searchsorted(cdf, u, side="right")
    below = torch.clamp(inds - 1, 0, existing.shape[-1] - 1)
    above = torch.clamp(inds, 0, existing.shape[-1] - 1)
    cdf0, cdf1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(existing, -1, below), torch.gather(existing, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf0) / (cdf1 - cdf0), 0), 0, 1)
    bins = (b0 + t * (b1 - b0)).detach()
    return Samples(bins, spacing_to_euclidean(bins, prev.s_near, prev.s_far, lam, scaling), prev.s_near, prev.s_far)


def proposal_sample(origins, directions, pixel_area, nears, fars, density_fns: List[Callable],
                    num_proposal_samples=(128, 64), num_nerf_samples=32,
                    t_rand=None, jitters=(None, None)):
    """ProposalNetworkSampler.generate_ray_samples.  ray_samplers.py:623-666.

    density_fns[i](starts, ends) -> density [B,S]; anneal = 1 (no-op, :650).  NOTE the reference
    builds its density_fns with a late-binding lambda so BOTH rounds evaluate proposal_fields[1]
    (models/neuradar.py:302) -- the caller decides what to pass here.
    Returns (final Samples, [weights per round], [Samples per round]).
    """
    n = len(num_proposal_samples)
    weights_list, samples_list = [], []
    samples, weights = None, None
    for level in range(n + 1):
        is_prop = level < n
        count = num_proposal_samples[level] if is_prop else num_nerf_samples
        if level == 0:
            samples = power_bins(nears, fars, count, t_rand)
        else:
            samples = pdf_resample(samples, weights, count, jitters[level - 1])
        if is_prop:
            density = density_fns[level](samples.starts, samples.ends)
            weights = weights_from_density(samples.deltas, density)
            weights_list.append(weights)
            samples_list.append(samples)
    return samples, weights_list, samples_list


def stretch_last_sample_to_sky(samples: Samples, sky_distance: float = SKY_DISTANCE) -> Samples:
    """The "sky field" trick.  models/neuradar.py:578-582: the last sample's end is moved to
    sky_distance (ends += d, deltas += d) and its spacing end is set to 1 - 1e-7."""
    euclid = samples.euclid.clone()
    euclid[:, -1] = euclid[:, -1] + (sky_distance - euclid[:, -1])
    spacing = samples.spacing.clone()
    spacing[:, -1] = 1 - 1e-7
    return Samples(spacing, euclid, samples.s_near, samples.s_far)
