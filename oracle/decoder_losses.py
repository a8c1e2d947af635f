"""Oracle: the training losses behind the rendered features of a mixed batch -- decode_features
(models/neuradar.py:410-493) in TRAINING mode followed by the terms of get_metrics_dict / get_loss_dict (:588-704) that
depend on the decoders and on the rendered depths.  Test infrastructure only (imported by tests/ and bench.py's checker
legs, never by the product path).

Pinned by tests/golden/model_train.npz, which the reference NeuRadarModel's own methods generate
(tests/golden/make_golden.py::golden_model_train; dropout 0 on the instance, batch norm in training mode).

Restates
  * :432-452  lidar rows -> MLP -> (sigmoid intensity, ray-drop logit)
  * :455-461  camera rows -> [patches, C, h, w] -> CNN (training-mode batch norm) -> rgb [patches, 3h, 3w, 3]
  * :463-491  radar rows -> transformer + heads -> radar_output  (oracle/radar.py)
  * :589-592,672-673          rgb_loss   = rgb_mult * MSE(image, rgb)
  * :612-636,690-700          depth_loss (non-returns pulled beyond 150 m, x non_return_loss_mult, 95 % quantile mask),
                              intensity_loss (returning rays inside the quantile), ray_drop_loss (BCE with logits)
  * :641-650,679-688          depth_loss_i on the proposal depths (no quantile), x prop_lidar_loss_mult * depth_mult
  * :652-662,702-703          radar_loss = radar_mult * calculate_radar_loss(...)
"""
from dataclasses import dataclass
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import Tensor

from . import radar as orad
from .decoders import mlp


@dataclass
class LossSettings:
    """models/neuradar.py:80-115."""

    rgb_mult: float = 5.0
    depth_mult: float = 0.01
    intensity_mult: float = 0.1
    quantile_threshold: float = 0.95
    non_return_lidar_distance: float = 150.0
    non_return_loss_mult: float = 0.1
    ray_drop_loss_mult: float = 0.01
    prop_lidar_loss_mult: float = 0.1
    radar_mult: float = 0.02
    radar_loss_type: str = "nll"


def lidar_depth_unreduced(pred_depth: Tensor, distance: Tensor, did_return: Tensor, c: LossSettings) -> Tensor:
    """:614-622 (also :643-648): L1 against the measured range; a non-returning ray's target is max(its own detached
    prediction, 150 m) and its loss is scaled by non_return_loss_mult.  All [n,1], did_return [n] bool."""
    target = distance.clone()
    target[~did_return] = pred_depth.detach()[~did_return].clamp_min(c.non_return_lidar_distance)
    un = (target - pred_depth).abs()
    scale = torch.where(did_return, 1.0, c.non_return_loss_mult)[:, None]
    return un * scale


def decoder_losses(features: Tensor, depth: Tensor, prop_depths: List[Tensor], spher: Tensor, is_lidar: Tensor, is_radar: Tensor,
                   batch: Dict[str, Tensor], p: Dict[str, Tensor], patch: int, n_scans: int, c: LossSettings = LossSettings()
                   ) -> Dict[str, Tensor]:
    """features [B,48] (rendered features | appearance), depth [B,1], prop_depths 2 x [B,1], spher [B,2], masks [B,1] bool;
    batch: image [P,3h,3w,3], distance [n_l,1], did_return [B,1], lidar [n_l,>=4] (intensity in column 3), radar [m,>=3],
    radar_indices [m,2].  p: decoder parameters by their state_dict names.  Returns the weighted loss terms (+ "rgb",
    "intensity", "ray_drop_logits", "radar_output", "assoc")."""
    lid, rad = is_lidar[:, 0], is_radar[:, 0]
    cam = ~(lid | rad)
    out: Dict[str, Tensor] = {}
    if bool(cam.any()):
        rgb = orad.rgb_decode(features[cam], (patch, patch), {k: v for k, v in p.items() if k.startswith("rgb_decoder")}, training=True)
        out["rgb"] = rgb
        out["rgb_loss"] = c.rgb_mult * F.mse_loss(rgb, batch["image"])
    if bool(lid.any()):
        ws = [p[f"lidar_decoder.layers.{i}.weight"] for i in range(3)]
        bs = [p[f"lidar_decoder.layers.{i}.bias"] for i in range(3)]
        y = mlp(features[lid], ws, bs)
        intensity, drop = y[:, :1].sigmoid(), y[:, 1:2]
        did = batch["did_return"][lid][:, 0]
        un = lidar_depth_unreduced(depth[lid], batch["distance"], did, c)
        q = torch.quantile(un, c.quantile_threshold)
        mask = (un < q)[:, 0]
        qr = mask & did
        out.update(intensity=intensity, ray_drop_logits=drop)
        out["depth_loss"] = c.depth_mult * un[mask].mean()
        out["intensity_loss"] = c.intensity_mult * ((batch["lidar"][:, 3:4][qr] - intensity[qr]) ** 2).mean()
        out["ray_drop_loss"] = c.ray_drop_loss_mult * F.binary_cross_entropy_with_logits(drop, (~did)[:, None].to(drop))
        for i, pd in enumerate(prop_depths):
            out[f"depth_loss_{i}"] = c.prop_lidar_loss_mult * c.depth_mult * lidar_depth_unreduced(pd[lid], batch["distance"], did, c).mean()
    if bool(rad.any()):
        ro = orad.decode_radar(features[rad], depth[rad], spher[rad], n_scans, p)
        loss, assocs = orad.radar_loss(batch["radar"], ro, batch["radar_indices"], c.radar_loss_type, training=True)
        out.update(radar_output=ro, assoc=assocs)
        out["radar_loss"] = c.radar_mult * loss
    return out


LOSS_KEYS = ("rgb_loss", "depth_loss", "intensity_loss", "ray_drop_loss", "radar_loss", "depth_loss_0", "depth_loss_1")


def total(losses: Dict[str, Tensor]) -> Tensor:
    return sum(losses[k] for k in LOSS_KEYS if k in losses)
