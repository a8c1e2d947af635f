"""Renderers of the path (reference: model_components/renderers.py:59-90,322-350; models/neurad.py:
721-728; nerfacc.render_weight_from_alpha as used at models/neuradar.py:1016).

`composite()` is the fused kernel (weights, accumulation, sky fix-up, features, depth in one launch);
the reference-named pieces below are thin views over it for callers that use them one by one.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from . import ops
from .rays import RaySamples


def composite(alpha: Tensor, feature: Tensor, ray_samples: RaySamples):
    """alpha [B,S,1], feature [B,S,C] -> weights [B,S,1] (sky fix-up applied), accumulation [B,1],
    features [B,C], depth [B,1]; the sequence of models/neuradar.py:504-517."""
    w, acc, feats, depth = ops.composite(alpha[..., 0], feature, ray_samples.euclid)
    return w[..., None], acc[:, None], feats, depth[:, None]


def render_weight_from_alpha(alphas: Tensor) -> Tuple[Tensor, Tensor]:
    """nerfacc batched branch: alphas [B,S] -> (weights, transmittance).  Implemented with the fused
    kernel on a dummy feature; the sky fix-up of the last sample is undone."""
    B, S = alphas.shape
    dummy = torch.zeros((B, S, 1), device=alphas.device)
    edges = torch.zeros((B, S + 1), device=alphas.device)
    w, acc, _, _ = ops.composite(alphas, dummy, edges)
    w = torch.cat([w[:, :-1], w[:, -1:] - (1 - acc[:, None])], dim=-1)
    trans = torch.where(alphas > 0, w / alphas.clamp_min(1e-30), torch.ones_like(w))
    return w, trans


class FeatureRenderer(nn.Module):
    def forward(self, features: Tensor, weights: Tensor, ray_indices=None, num_rays=None) -> Tensor:
        return torch.sum(features * weights, dim=-2)  # renderers.py:85 (unpacked branch)


class AccumulationRenderer(nn.Module):
    def forward(self, weights: Tensor, ray_indices=None, num_rays=None) -> Tensor:
        return torch.sum(weights, dim=-2)  # renderers.py:349


def render_depth_simple(weights: Tensor, ray_samples: RaySamples, ray_indices=None, num_rays: Optional[int] = None
                        ) -> Tensor:
    """models/neurad.py:721-728: sum w * (start+end)/2, not normalised.  weights [B,S,1] -> [B,1]."""
    return ops.depth_from_weights(weights[..., 0], ray_samples.euclid)[:, None]
