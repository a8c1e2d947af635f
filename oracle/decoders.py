"""CPU restatement of the reference's per-ray lidar decoder and lidar losses (test infrastructure only:
imported by tests/, never by the product path).

Reference: nerfstudio/models/neuradar.py
  * decoder construction            :241-248  MLP(in = nff_out_dim + appearance_dim, width 32, 3 layers, out 2)
  * decode_features, lidar branch   :432-452  rows of `features` with is_lidar -> (intensity, ray_drop_logit);
                                              intensity = sigmoid(.)
  * lidar losses (training)         :612-636  depth L1 with non-return handling and quantile mask, intensity MSE on
                                              returned rays inside the quantile, ray-drop BCE with logits
  * multipliers                     :80-110, :690-700
The MLP itself (field_components/mlp.py:159-178, torch path) is pinned by tests/golden/sh_mlp.npz, generated from
the reference; the loss formulas have no vectors in the reference's tests (parity of those lines: restated)."""
from typing import Dict, List

import torch
from torch import Tensor

from . import field


def mlp(x: Tensor, weights: List[Tensor], biases: List[Tensor]) -> Tensor:
    """MLP.pytorch_fwd (mlp.py:159-178) -- the restatement in oracle/field.py, pinned by the sh_mlp golden."""
    return field.mlp(x, list(zip(weights, biases)))


def lidar_decode(features: Tensor, is_lidar: Tensor, weights: List[Tensor], biases: List[Tensor]):
    """neuradar.py:432-452.  features [B,C], is_lidar [B,1] bool -> intensity [n_lidar,1], ray_drop_logit [n_lidar,1]."""
    lidar_features = features[is_lidar[..., 0]]
    if lidar_features.numel() == 0:
        return None, None
    intensity, ray_drop_logit = mlp(lidar_features, weights, biases).split(1, dim=-1)
    return intensity.sigmoid(), ray_drop_logit


def lidar_losses(pred_depth: Tensor, intensity: Tensor, ray_drop_logits: Tensor, termination_depth: Tensor,
                 did_return: Tensor, points_intensities: Tensor, non_return_lidar_distance: float = 150.0,
                 non_return_loss_mult: float = 0.1, quantile_threshold: float = 0.95) -> Dict[str, Tensor]:
    """neuradar.py:612-636.  All per-lidar-ray tensors are [n,1] except did_return [n] bool."""
    target_depth = termination_depth.clone()
    nonret = torch.tensor(non_return_lidar_distance, device=pred_depth.device)
    target_depth[~did_return] = pred_depth.detach()[~did_return].maximum(nonret)
    unreduced = (target_depth - pred_depth).abs()
    unreduced[~did_return] = unreduced[~did_return] * non_return_loss_mult
    quantile = torch.quantile(unreduced, quantile_threshold)
    quantile_mask = (unreduced < quantile).squeeze(-1)
    out = {"depth_loss": torch.mean(unreduced[quantile_mask])}
    qr = quantile_mask & did_return
    out["intensity_loss"] = ((points_intensities[qr] - intensity[qr]) ** 2).mean()
    out["ray_drop_loss"] = torch.nn.functional.binary_cross_entropy_with_logits(
        ray_drop_logits, (~did_return).unsqueeze(-1).to(ray_drop_logits))
    return out
