"""Oracle: the reference's per-ray decoders behind the rendered features -- radar transformer + heads, RGB CNN -- and
the radar point-set evaluation (MultiBernoulli, Hungarian-matched loss, sampled detections, Chamfer distance).
Test infrastructure only.

Restates
  * detr/models/position_encoding_3d.py:56-103  PositionEmbeddingCoordsSine.get_sine_embeddings (scale 2*pi, T=1e4)
  * detr/models/transformer.py:32-70,143-205    one pre-norm encoder layer (single head, d=48, ff=64) + final LayerNorm
  * models/neuradar.py:251-278,463-491          the four heads and decode_features' radar branch
  * models/neuradar.py:225-240,455-461, model_components/cnns.py:21-47   the RGB CNN decoder (eval-mode batch norm)
  * model_components/radar_utils.py:35-51       MultiBernoulli
  * model_components/radar_utils.py:54-93,95-118,156-168   calculate_radar_loss / get_cost_matrix / get_radar_loss,
                                                "euclidean" (deterministic head) branch, scipy's Hungarian matching
  * model_components/radar_utils.py:105-118,130-154  the "nll" (probabilistic head, the reference's default neuradar.py:114) cost
                                                matrix and loss: Laplace log-likelihoods of the detections
  * model_components/radar_utils.py:170-229     sample_radar_points, "euclidean" and "nll" branches
  * model_components/radar_utils.py:380-420     chamfer_distance (bidirectional mean nearest-neighbour distance; the
                                                reference asks sklearn's exact kd-tree, here brute force: same value)
Parameters are passed as a dict with the reference's state_dict names (prefix removed).
"""
import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

EPS, MIN_VAR = 1e-6, 1e-3  # radar_utils.py:30-31


def sine_position_embedding(xyz: torch.Tensor, num_channels: int, temperature: float = 10000.0) -> torch.Tensor:
    """xyz [N, n, 3] -> [N, num_channels, n].  Every coordinate gets ndim = (num_channels // 3 rounded down to even)
    channels (+2 for the first ones while a remainder is left): sin on even, cos on odd channel indices."""
    xyz = xyz.clone()
    d_in = xyz.shape[2]
    ndim = num_channels // d_in
    if ndim % 2 != 0:
        ndim -= 1
    rems = num_channels - ndim * d_in
    embeds = []
    for d in range(d_in):
        cdim = ndim
        if rems > 0:
            cdim += 2
            rems -= 2
        dim_t = torch.arange(cdim, dtype=torch.float32)
        dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / cdim)
        raw = xyz[:, :, d] * (2 * math.pi)
        pos = raw[:, :, None] / dim_t
        embeds.append(torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=3).flatten(2))
    return torch.cat(embeds, dim=2).permute(0, 2, 1)


def encoder(src: torch.Tensor, pos: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str = "radar_decoder.") -> torch.Tensor:
    """Transformer.forward in eval mode (dropout off).  src, pos [N, C, n] -> [N, n, C]."""
    L = prefix + "encoder.layers.0."
    x = src.permute(0, 2, 1)  # [N, n, C]; attention is per scan, so the batch axis can stay in front
    pe = pos.permute(0, 2, 1)
    C = x.shape[-1]
    x2 = F.layer_norm(x, (C,), p[L + "norm1.weight"], p[L + "norm1.bias"])
    qk = x2 + pe
    w, b = p[L + "self_attn.in_proj_weight"], p[L + "self_attn.in_proj_bias"]
    q = F.linear(qk, w[:C], b[:C])
    k = F.linear(qk, w[C:2 * C], b[C:2 * C])
    v = F.linear(x2, w[2 * C:], b[2 * C:])
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(C), dim=-1)  # one head of width C
    x = x + F.linear(att @ v, p[L + "self_attn.out_proj.weight"], p[L + "self_attn.out_proj.bias"])
    x2 = F.layer_norm(x, (C,), p[L + "norm2.weight"], p[L + "norm2.bias"])
    ff = F.linear(torch.relu(F.linear(x2, p[L + "linear1.weight"], p[L + "linear1.bias"])), p[L + "linear2.weight"], p[L + "linear2.bias"])
    x = x + ff
    return F.layer_norm(x, (C,), p[prefix + "encoder.norm.weight"], p[prefix + "encoder.norm.bias"])


def head(x: torch.Tensor, p: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
    """MLP(in, width 16, 3 layers) without its output activation (field_components/mlp.py:159-178)."""
    for i in range(3):
        x = F.linear(x, p[f"{name}.layers.{i}.weight"], p[f"{name}.layers.{i}.bias"])
        if i < 2:
            x = torch.relu(x)
    return x


def decode_radar(features: torch.Tensor, depth: torch.Tensor, directions_spher: torch.Tensor, num_scans: int,
                 p: Dict[str, torch.Tensor]) -> torch.Tensor:
    """neuradar.py:463-491.  features [n_radar, C], depth [n_radar, 1], directions_spher [n_radar, 2] (azimuth,
    elevation) of the radar rays, scan after scan -> radar_output [num_scans, n, 7] = (existence probability,
    x, y, z, three Laplace scales)."""
    C = features.shape[-1]
    depth = depth.view(num_scans, -1, 1)
    sph = directions_spher.view(num_scans, -1, 2)
    theta, phi = sph[..., 1:2], sph[..., 0:1]  # _get_cartesian_coords(depth, theta = elevation, phi = azimuth), :1025-1029
    xyz = torch.cat((depth * torch.cos(phi) * torch.cos(theta), depth * torch.sin(phi) * torch.cos(theta), depth * torch.sin(theta)), dim=2)
    with torch.no_grad():  # PositionEmbeddingCoordsSine.forward (position_encoding_3d.py:133-135): no gradient to the points
        pos = sine_position_embedding(xyz, C)
    out = encoder(features.view(num_scans, -1, C).permute(0, 2, 1), pos, p)
    offset = 1.5 * torch.tanh(head(out, p, "offset_head"))
    ep = torch.sigmoid(head(out, p, "existence_probability_head"))
    unc = F.softplus(head(out, p, "radar_uncertainty_head"))
    return torch.cat((ep, xyz + offset, unc), dim=-1)


def rgb_decode(cam_features: torch.Tensor, patch_size: Tuple[int, int], p: Dict[str, torch.Tensor], upsample: int = 3,
               training: bool = False) -> torch.Tensor:
    """neuradar.py:225-240,455-461.  cam_features [n_patches * h * w, C] -> rgb [n_patches, h*up, w*up, 3].  training=True:
    batch norm normalises with the batch's own (biased) statistics, as nn.BatchNorm2d does in train mode."""
    x = cam_features.view(-1, *patch_size, cam_features.shape[-1]).permute(0, 3, 1, 2)
    pre = "rgb_decoder."
    x = torch.relu(F.conv2d(x, p[pre + "0.weight"], p[pre + "0.bias"]))

    def bn(y, k):
        if training:
            return F.batch_norm(y, None, None, p[k + "weight"], p[k + "bias"], True)
        return F.batch_norm(y, p[k + "running_mean"], p[k + "running_var"], p[k + "weight"], p[k + "bias"], False)

    def block(x, i):  # BasicBlock(kernel 7, padding 3, batch norm), cnns.py:21-47
        b = f"{pre}{i}.main_branch."
        y = F.conv2d(x, p[b + "0.weight"], p[b + "0.bias"], padding=3)
        y = bn(y, b + "1.")
        y = F.conv2d(torch.relu(y), p[b + "3.weight"], p[b + "3.bias"], padding=3)
        y = bn(y, b + "4.")
        return torch.relu(x + y)

    x = block(block(x, 2), 3)
    x = F.conv_transpose2d(x, p[pre + "4.weight"], p[pre + "4.bias"], stride=upsample)
    x = block(block(x, 5), 6)
    return torch.sigmoid(F.conv2d(x, p[pre + "7.weight"], p[pre + "7.bias"])).permute(0, 2, 3, 1)


# ---- point-set evaluation ------------------------------------------------------------------------------------------
def multi_bernoulli(pred: torch.Tensor) -> Dict[str, torch.Tensor]:
    """radar_utils.py:35-45: clamps of the existence probability and the scales."""
    return {"ep": pred[..., 0].clamp(min=EPS, max=1 - EPS), "xyz": pred[..., 1:4], "scale": pred[..., 4:7].clamp(min=MIN_VAR)}


def laplace_log_prob(x: torch.Tensor, loc: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """torch.distributions.Laplace.log_prob: -log(2 b) - |x - mu| / b."""
    return -torch.log(2 * scale) - torch.abs(x - loc) / scale


def cost_matrix(gt: torch.Tensor, mb: Dict[str, torch.Tensor], method: str) -> torch.Tensor:
    """get_cost_matrix (radar_utils.py:95-127): [n predictions, m detections].  "euclidean": distance - log r;
    "nll": log(1 - r) - log r - sum of the three Laplace log-likelihoods.  Infinite entries -> MAX_COST = 1e9."""
    if method == "euclidean":
        cost = torch.cdist(mb["xyz"], gt[:, :3]) - mb["ep"].log()[:, None]
    else:
        ll = sum(laplace_log_prob(gt[None, :, a], mb["xyz"][:, None, a], mb["scale"][:, None, a]) for a in range(3))
        cost = ((1 - mb["ep"]).log() - mb["ep"].log())[:, None] - ll
    return torch.where(cost.isinf(), torch.full_like(cost, 1e9), cost)


def hungarian(cost: torch.Tensor) -> torch.Tensor:
    """scipy.optimize.linear_sum_assignment on the [n, m] cost -> association [n]: detection index or -1."""
    from scipy.optimize import linear_sum_assignment

    row, col = linear_sum_assignment(cost.detach().numpy())
    assoc = -torch.ones(cost.shape[0], dtype=torch.long)
    assoc[torch.as_tensor(row, dtype=torch.long)] = torch.as_tensor(col, dtype=torch.long)
    return assoc


def scan_loss(gt: torch.Tensor, mb: Dict[str, torch.Tensor], assoc: torch.Tensor, loss_type: str) -> torch.Tensor:
    """get_radar_loss (radar_utils.py:130-168) of one scan: unmatched predictions pay -log(1 - r); a matched one pays
    -log r + |xyz - detection| ("euclidean") or -log r - Laplace log-likelihood of its detection ("nll"); sum / n."""
    matched = assoc > -1
    per = -(1 - mb["ep"]).log()
    if bool(matched.any()):
        tgt = gt[assoc[matched], :3]
        if loss_type == "euclidean":
            pay = torch.norm(mb["xyz"][matched] - tgt, dim=-1)
        else:
            pay = -sum(laplace_log_prob(tgt[:, a], mb["xyz"][matched, a], mb["scale"][matched, a]) for a in range(3))
        per = per.clone()
        per[matched] = pay - mb["ep"].log()[matched]
    return per.sum() / mb["ep"].shape[-1]


def radar_loss(radar_batch: torch.Tensor, prediction: torch.Tensor, indices: torch.Tensor, loss_type: str = "nll",
               training: bool = True):
    """calculate_radar_loss (radar_utils.py:54-93): scans are the runs of `indices` starting where indices[:, 1] == 0; per
    scan the Hungarian matching on the EUCLIDEAN cost while training (:77-78; the loss type's own cost otherwise), then
    scan_loss; mean over scans.  Returns (loss, [association [n] per scan])."""
    seg = (indices[:, 1] == 0).nonzero(as_tuple=True)[0]
    seg = torch.cat((seg, torch.tensor([indices.shape[0]])))
    losses, assocs = [], []
    for i in range(seg.numel() - 1):
        gt = radar_batch[seg[i]:seg[i + 1], :3]
        mb = multi_bernoulli(prediction[i])
        assoc = hungarian(cost_matrix(gt, mb, "euclidean" if training else loss_type))
        losses.append(scan_loss(gt, mb, assoc, loss_type))
        assocs.append(assoc)
    return torch.stack(losses).mean(), assocs


def radar_loss_euclidean(radar_batch: torch.Tensor, prediction: torch.Tensor, indices: torch.Tensor):
    """The deterministic head of BASELINE configs[2]: (loss, the last scan's association [n, 2] as the reference returns it)."""
    loss, assocs = radar_loss(radar_batch, prediction, indices, "euclidean", True)
    n = assocs[-1].numel()
    return loss, torch.stack([torch.arange(n).float(), assocs[-1].float()], dim=1)


def sample_radar_points_nll(radar_output: torch.Tensor, max_detections: int = 1000):
    """radar_utils.py:181-213, "nll": walk the LAST scan's predictions in index order; each draws u ~ U(0,1) (torch.rand) and
    is kept when u < r and it is among the max_detections most probable; a kept one draws x, y, z from its Laplace
    distributions (rsample) right away -- the global torch generator is consumed in exactly that order.
    Returns (points [m,3], indices [m])."""
    from torch.distributions.laplace import Laplace

    mb = multi_bernoulli(radar_output[-1])
    ep = mb["ep"].flatten()
    top = set(torch.argsort(ep, descending=True)[:max_detections].tolist())
    pts, idx = [], []
    for i in range(ep.numel()):
        u = torch.rand(1)
        if bool(u < ep[i]) and i in top:
            idx.append(i)
            pts.append(torch.stack([Laplace(mb["xyz"][i, a], mb["scale"][i, a]).rsample(torch.Size([1])).view(()) for a in range(3)]))
    if not pts:
        return torch.empty(0, 3), torch.empty(0, dtype=torch.long)
    return torch.stack(pts), torch.tensor(idx, dtype=torch.long)


def sample_radar_points(radar_output: torch.Tensor, threshold: float = 0.5, max_detections: int = 1000):
    """radar_utils.py:170-229, "euclidean": the LAST scan's predictions, sorted by existence probability (descending,
    at most max_detections), kept where the probability exceeds the threshold.  Returns (points [m,3], indices [m])."""
    mb = multi_bernoulli(radar_output[-1])
    order = torch.argsort(mb["ep"].flatten(), descending=True)[:max_detections]
    keep = mb["ep"].flatten()[order] > threshold
    return mb["xyz"].reshape(-1, 3)[order][keep], order[keep]


def chamfer_distance(x: np.ndarray, y: np.ndarray) -> float:
    """radar_utils.py:380-420, direction "bi", metric l2: mean_y min_x |x - y| + mean_x min_y |x - y|."""
    d = np.linalg.norm(x[:, None, :].astype(np.float64) - y[None, :, :].astype(np.float64), axis=-1)
    return float(d.min(axis=0).mean() + d.min(axis=1).mean())
