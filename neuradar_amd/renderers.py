"""Renderers of the path (reference: model_components/renderers.py:59-90,322-350; models/neurad.py:
721-728; nerfacc.render_weight_from_alpha as used at models/neuradar.py:1016).

`composite()` is the fused kernel (weights, accumulation, sky fix-up, features, depth in one launch);
the reference-named pieces below are thin views over it for callers that use them one by one.
"""
from typing import Optional, Tuple

from torch import Tensor, nn

from . import ops
from .rays import RaySamples


def composite(alpha: Tensor, feature: Tensor, ray_samples: RaySamples):
    """alpha [B,S,1], feature [B,S,C] -> weights [B,S,1] (sky fix-up applied), accumulation [B,1],
    features [B,C], depth [B,1]; the sequence of models/neuradar.py:504-517."""
    w, acc, feats, depth = ops.composite(alpha[..., 0], feature, ray_samples.euclid)
    return w[..., None], acc[:, None], feats, depth[:, None]


def render_weight_from_alpha(alphas: Tensor, packed_info=None, ray_indices=None, n_rays=None, prefix_trans=None
                             ) -> Tuple[Tensor, Tensor]:
    """nerfacc.render_weight_from_alpha, batched branch (call site models/neuradar.py:1016): alphas [B,S] ->
    (weights, transmittance) with T_i = prod_{j<i}(1 - alpha_j) -- also where alpha_i = 0 -- and w_i = alpha_i T_i."""
    if packed_info is not None or ray_indices is not None or prefix_trans is not None:
        raise NotImplementedError("only nerfacc's batched branch (no packed_info / ray_indices) is on NeuRadar's path")
    return ops.render_weights(alphas)


def render_weight_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info=None, ray_indices=None,
                               n_rays=None, prefix_trans=None) -> Tuple[Tensor, Tensor, Tensor]:
    """nerfacc.render_weight_from_density, batched branch (models/neuradar.py:1018-1022, use_sdf=False):
    (weights, transmittance, alphas), alpha = 1 - exp(-sigma (t_end - t_start))."""
    if packed_info is not None or ray_indices is not None or prefix_trans is not None:
        raise NotImplementedError("only nerfacc's batched branch (no packed_info / ray_indices) is on NeuRadar's path")
    return ops.render_weights_from_density(t_starts, t_ends, sigmas)


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor] = None, ray_indices=None, n_rays=None) -> Tensor:
    """nerfacc.accumulate_along_rays, batched branch (models/neurad.py:727-728, renderers.py:88,345): weights [B,S],
    values [B,S,C] -> [B,C]; values None -> [B,1]."""
    if ray_indices is not None:
        raise NotImplementedError("only nerfacc's batched branch (ray_indices=None) is on NeuRadar's path")
    return ops.accumulate_along_rays(weights, values)


class FeatureRenderer(nn.Module):
    def forward(self, features: Tensor, weights: Tensor, ray_indices=None, num_rays=None) -> Tensor:
        """renderers.py:59-90: features [B,S,C], weights [B,S,1] -> [B,C]."""
        return accumulate_along_rays(weights[..., 0], features, ray_indices, num_rays)


class AccumulationRenderer(nn.Module):
    def forward(self, weights: Tensor, ray_indices=None, num_rays=None) -> Tensor:
        """renderers.py:322-350: weights [B,S,1] -> [B,1]."""
        return accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays)


def render_depth_simple(weights: Tensor, ray_samples: RaySamples, ray_indices=None, num_rays: Optional[int] = None
                        ) -> Tensor:
    """models/neurad.py:721-728: sum w * (start+end)/2, not normalised.  weights [B,S,1] -> [B,1]."""
    return ops.depth_from_weights(weights[..., 0], ray_samples.euclid)[:, None]
