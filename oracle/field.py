"""Oracle: NeuRAD field and proposal field, static scene.  Test infrastructure only.

Restates cameras/rays.py:109-124, field_components/spatial_distortions.py:103-113,126-136,
field_components/neurad_encoding.py:152-189,277-280,309-316, field_components/mlp.py:159-178,
utils/math.py:31-94, fields/base_field.py:135-141, fields/neurad_field.py:128-152,208-213,
model_components/utils.py:21-46, field_components/activations.py:28-54.
"""
from dataclasses import dataclass, field as dc_field
from typing import List, Optional, Tuple

import torch

from . import hashgrid

BETA_MIN = 1e-4  # model_components/utils.py:24


@dataclass
class GridParams:
    """One `HashEncoding` (encodings.py:326-352): table [L*T,F] + per-level scalings."""

    table: torch.Tensor
    scalings: torch.Tensor
    log2_hashmap_size: int

    @property
    def table_size(self) -> int:
        return 2**self.log2_hashmap_size

    @property
    def num_levels(self) -> int:
        return self.scalings.numel()

    @property
    def features_per_level(self) -> int:
        return self.table.shape[-1]


@dataclass
class FieldParams:
    """`NeuRADField` parameters (neurad_field.py:92-119).  Linear layers are (weight[out,in], bias[out])."""

    grid: GridParams
    geo: List[Tuple[torch.Tensor, torch.Tensor]]
    feat: List[Tuple[torch.Tensor, torch.Tensor]]
    beta: torch.Tensor  # SigmoidDensity.beta, shape [1]
    static_scale: float = 100.0
    actor_grids: List[GridParams] = dc_field(default_factory=list)

    def tensors(self):
        out = [self.grid.table]
        for w, b in self.geo + self.feat:
            out += [w, b]
        out.append(self.beta)
        return out


@dataclass
class ProposalParams:
    """`NeuRADProposalField` parameters (neurad_field.py:198-201): grid + Linear(L*F, 1, bias=False)."""

    grid: GridParams
    decoder: torch.Tensor  # [1, L*F]
    static_scale: float = 100.0
    actor_grids: List[GridParams] = dc_field(default_factory=list)

    def tensors(self):
        return [self.grid.table, self.decoder]


def isotropic_gaussian(origins, directions, starts, ends, pixel_area):
    """Frustums.get_fast_isotropic_gaussian with num_multisamples=1.  cameras/rays.py:109-124.

    origins/directions [B,3], starts/ends [B,S], pixel_area [B,1] -> mean [B,S,3], std [B,S,1].
    """
    half = (ends - starts) / 2  # multisample_dist with one multisample
    t = starts + 1.0 * half
    mean = origins[:, None, :] + directions[:, None, :] * t[..., None]
    cross_section = pixel_area[:, None, :] * t[..., None].pow(2)
    std = (cross_section * half[..., None]).pow(1 / 3)
    return mean, std


def scaled_contraction(mean, std, scale):
    """ScaledSceneContraction(order=inf) on a GaussiansStd.  spatial_distortions.py:103-113,126-136.

    L-inf MipNeRF-360 contraction of mean/scale, ZipNeRF-style linearised std scaling, then the
    [-2,2] cube is mapped to [0,1].
    """
    x = mean / scale
    s = std / scale
    mag = x.abs().amax(dim=-1, keepdim=True)  # linalg.norm(ord=inf)
    inside = mag < 1
    m = mag.clamp_min(1.0)
    x = torch.where(inside, x, (2 - (1 / m)) * (x / m))
    s = torch.where(inside, s, s * ((2 * m - 1).pow(1 / 3) / m) ** 2)
    return (x + 2.0) / 4.0, s / 4.0


def rescale_grid_features(feats, std01, grid: GridParams):
    """ZipNeRF-style per-level down-weighting.  neurad_encoding.py:309-316.

    feats [..., L*F], std01 [..., 1] -> [..., L*F]; weight_l = 1 / max(1, 2 * scaling_l * std).
    """
    L, F = grid.num_levels, grid.features_per_level
    w = 1 / (grid.scalings.to(std01) * 2 * std01).clamp_min(1.0)  # [..., L]
    return (feats.unflatten(-1, (L, F)) * w[..., None]).flatten(-2, -1)


def static_grid_features(mean, std, grid: GridParams, scale: float):
    """NeuRADHashEncoding.forward, static branch.  neurad_encoding.py:168-173,277-280."""
    x01, s01 = scaled_contraction(mean, std, scale)
    raw = hashgrid.encode(x01.reshape(-1, 3), grid.table, grid.scalings, grid.table_size)
    return rescale_grid_features(raw.view(*mean.shape[:-1], -1), s01, grid)


def mlp(x, layers):
    """MLP.pytorch_fwd: Linear + ReLU on hidden layers, no output activation.  mlp.py:159-178."""
    for i, (w, b) in enumerate(layers):
        x = torch.nn.functional.linear(x, w, b)
        if i < len(layers) - 1:
            x = torch.relu(x)
    return x


def sh4(d):
    """Degree-4 real spherical harmonics (16 components).  utils/math.py:31-78."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x * x, y * y, z * z
    comps = [
        torch.full_like(x, 0.28209479177387814),
        0.4886025119029199 * y,
        0.4886025119029199 * z,
        0.4886025119029199 * x,
        1.0925484305920792 * x * y,
        1.0925484305920792 * y * z,
        0.9461746957575601 * zz - 0.31539156525251999,
        1.0925484305920792 * x * z,
        0.5462742152960396 * (xx - yy),
        0.5900435899266435 * y * (3 * xx - yy),
        2.890611442640554 * x * y * z,
        0.4570457994644658 * y * (5 * zz - 1),
        0.3731763325901154 * z * (5 * zz - 3),
        0.4570457994644658 * x * (5 * zz - 1),
        1.445305721320277 * z * (xx - yy),
        0.5900435899266435 * x * (xx - 3 * yy),
    ]
    return torch.stack(comps, dim=-1)


def direction_encoding(directions):
    """SH of (d+1)/2, evaluated without grad.  neurad_field.py:140, base_field.py:135-141,
    encodings.py:797-800.  (The torch path feeds the [0,1]-mapped direction straight into the SH
    polynomial; tcnn would undo the mapping -- the torch behaviour is the oracle.)"""
    with torch.no_grad():
        return sh4((directions + 1.0) / 2.0)


def sigmoid_density(sdf, beta):
    """SigmoidDensity: alpha = sigmoid(-sdf * (|beta| + beta_min)).  model_components/utils.py:30-46."""
    return torch.sigmoid(-sdf * (beta.abs() + BETA_MIN))


class _TruncExp(torch.autograd.Function):
    """exp with the backward clamped to exp(clamp(x,-15,15)).  activations.py:28-41."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def field_forward(p: FieldParams, origins, directions, starts, ends, pixel_area,
                  actor_ctx: Optional[dict] = None, use_sdf: bool = True):
    """NeuRADField.forward.  neurad_field.py:128-152.

    Returns feature [B,S,C], sdf [B,S,1], alpha [B,S,1]; with use_sdf=False (:149-150) feature and
    density = trunc_exp(geo_out) [B,S,1].
    """
    B, S = starts.shape
    mean, std = isotropic_gaussian(origins, directions, starts, ends, pixel_area)
    feats = static_grid_features(mean, std, p.grid, p.static_scale)  # [B,S,L*F]
    dirs = directions[:, None, :].expand(B, S, 3)
    if actor_ctx is not None:
        from . import actors  # local import: actors are the second phase

        feats, dirs = actors.overwrite_actor_features(feats, dirs, mean, std, p.actor_grids, actor_ctx)
    feats = feats.reshape(B * S, -1)
    geo_dim = p.geo[-1][0].shape[0] - 1
    h = mlp(feats, p.geo)
    sdf, geo_embedding = torch.split(h, [1, geo_dim], dim=-1)
    sh = direction_encoding(dirs.reshape(-1, 3))
    feature = geo_embedding + mlp(torch.cat([geo_embedding, sh], dim=-1), p.feat)
    sdf = sdf.view(B, S, 1)
    if not use_sdf:
        return feature.view(B, S, -1), trunc_exp(sdf)
    return feature.view(B, S, -1), sdf, sigmoid_density(sdf, p.beta)


def proposal_density(p: ProposalParams, origins, directions, starts, ends, pixel_area,
                     actor_ctx: Optional[dict] = None):
    """NeuRADProposalField.get_density.  neurad_field.py:208-213.  Returns density [B,S,1]."""
    B, S = starts.shape
    mean, std = isotropic_gaussian(origins, directions, starts, ends, pixel_area)
    feats = static_grid_features(mean, std, p.grid, p.static_scale)
    if actor_ctx is not None:
        from . import actors

        feats, _ = actors.overwrite_actor_features(feats, None, mean, std, p.actor_grids, actor_ctx)
    logit = torch.nn.functional.linear(feats.reshape(B * S, -1), p.decoder)
    return trunc_exp(logit).view(B, S, 1)
