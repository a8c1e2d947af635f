"""Oracle: the hot loop `NeuRadarModel.get_nff_outputs` composed from the parts, plus the scalar
training loss the bench step differentiates.  Test infrastructure only.

Restates models/neuradar.py:495-548 (get_nff_outputs), :550-568 (_get_appearance_embedding),
:570-586 (_get_ray_samples), :971-994 (_compute_is_close_to_lidar).  The model itself cannot be
used for CPU goldens (its CPU `_render_weights` returns a constant 0.5, neuradar.py:1012-1014), so
the sequence is composed from components exactly as SURVEY section 8c prescribes.
"""
from typing import Dict, List, Optional

import torch

from . import field, losses, render, sampler

RGB_MULT = 5.0  # LossSettings.rgb_mult, neuradar.py:86
DEPTH_MULT = 0.01  # neuradar.py:88
INTERLEVEL_MULT = 0.001  # neuradar.py:98
DISTORTION_MULT = 0.002  # neuradar.py:100
CARVING_EPSILON = 0.1  # neuradar.py:94
NON_RETURN_LIDAR_DISTANCE = 150.0  # neuradar.py:102


def appearance_embedding(table, times, sensor_idx, duration: float, embeds_per_sensor: int):
    """Time-interpolated per-sensor embedding.  neuradar.py:556-565.

    table [n_sensors*E, D]; times [B,1]; sensor_idx [B,1] int64 -> [B,D].
    """
    time_idx = times / duration * embeds_per_sensor
    before = time_idx.floor().clamp(0, embeds_per_sensor - 1)
    after = (before + 1).clamp(0, embeds_per_sensor - 1)
    ratio = time_idx - before
    before = before + sensor_idx * embeds_per_sensor
    after = after + sensor_idx * embeds_per_sensor
    e0 = table[before.squeeze(-1).long()]
    e1 = table[after.squeeze(-1).long()]
    return e0 * (1 - ratio) + e1 * ratio


def is_close_to_lidar(starts, ends, is_lidar, directions_norm, did_return):
    """neuradar.py:971-994 for one RaySamples: per-sample mask [B,S] (False on non-lidar rays).

    starts/ends [B,S]; is_lidar, did_return [B,1] bool; directions_norm [B,1] (= lidar range).
    """
    mid = (starts + ends) * 0.5
    close_to_hit = (directions_norm - mid).abs() < CARVING_EPSILON
    in_range = mid < NON_RETURN_LIDAR_DISTANCE
    lidar_mask = (did_return & close_to_hit) | ((~did_return) & in_range)
    return is_lidar & lidar_mask


def nff_outputs(field_p: field.FieldParams, prop_ps: List[field.ProposalParams], bundle: Dict[str, torch.Tensor],
                t_rand: Optional[torch.Tensor] = None, jitters=(None, None),
                num_proposal_samples=(128, 64), num_nerf_samples=32,
                appearance: Optional[dict] = None, actor_ctx: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """get_nff_outputs in training mode.  neuradar.py:495-548.

    `bundle` holds origins, directions, pixel_area (ALREADY scaled by _scale_pixel_area), fars (and
    optionally nears, times, sensor_idx, is_lidar, directions_norm, did_return).  `prop_ps[i]` is
    the field evaluated in proposal round i; to mirror the reference's late-binding quirk pass
    [p1, p1] (neuradar.py:302).
    """
    o, d, area = bundle["origins"], bundle["directions"], bundle["pixel_area"]
    fars = bundle["fars"].clamp_max(sampler.SKY_DISTANCE)  # :573
    nears = bundle.get("nears", torch.zeros_like(fars))  # :576

    def density_fn(pp):
        return lambda s, e: field.proposal_density(pp, o, d, s, e, area, actor_ctx)[..., 0]

    final, prop_w, prop_s = sampler.proposal_sample(
        o, d, area, nears, fars, [density_fn(pp) for pp in prop_ps],
        num_proposal_samples, num_nerf_samples, t_rand, jitters)
    final = sampler.stretch_last_sample_to_sky(final)  # :578-582

    feature, sdf, alpha = field.field_forward(field_p, o, d, final.starts, final.ends, area, actor_ctx)
    comp = render.composite(alpha, feature, final.starts, final.ends)
    out = {
        "features": comp["features"], "depth": comp["depth"], "accumulation": comp["accumulation"],
        "weights": comp["weights"], "sdf": sdf, "alpha": alpha, "feature_samples": feature,
        "final_spacing": final.spacing, "final_euclid": final.euclid,
    }
    if appearance is not None:  # :510-512
        emb = appearance_embedding(appearance["table"], bundle["times"], bundle["sensor_idx"],
                                   appearance["duration"], appearance["embeds_per_sensor"])
        out["features"] = torch.cat([out["features"], emb], dim=-1)
    for i, (w, s) in enumerate(zip(prop_w, prop_s)):  # :527-528
        out[f"prop_depth_{i}"] = render.depth_simple(w, s.starts, s.ends)
        out[f"prop_weights_{i}"] = w
        out[f"prop_spacing_{i}"] = s.spacing
        out[f"prop_euclid_{i}"] = s.euclid
        if "is_lidar" in bundle:  # :529-531
            close = is_close_to_lidar(s.starts, s.ends, bundle["is_lidar"], bundle["directions_norm"],
                                      bundle["did_return"])
            mask = (~close) & bundle["is_lidar"]
            out[f"prop_weights_loss_{i}"] = ((w * mask) ** 2).sum()
    # weights_list / ray_samples_list for the regularisers: the sky sample is dropped (:515,534-535)
    out["c_list"] = [s.spacing for s in prop_s] + [final.spacing[:, :-1]]
    out["w_list"] = list(prop_w) + [comp["weights"][:, :-1]]
    return out


def train_loss(out: Dict[str, torch.Tensor], target_features: torch.Tensor, target_depth: torch.Tensor,
               depth_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Scalar loss of the bench training step.

    The reference's image/lidar/radar decoders and Hungarian radar loss are out of scope
    (K12, K13, radar_utils); they are replaced by direct supervision of the
    path's own outputs with the reference's multipliers, keeping the two regularisers that feed the
    proposal fields: rgb_mult*MSE(features) + depth_mult*L1(depth) + interlevel + distortion
    (neuradar.py:672-704).
    """
    loss = RGB_MULT * torch.mean((out["features"][:, : target_features.shape[1]] - target_features) ** 2)
    dl = (out["depth"] - target_depth).abs()
    loss = loss + DEPTH_MULT * (dl[depth_mask].mean() if depth_mask is not None else dl.mean())
    loss = loss + INTERLEVEL_MULT * losses.zipnerf_interlevel_loss(out["c_list"], out["w_list"])
    loss = loss + DISTORTION_MULT * losses.distortion_loss(out["c_list"][-1], out["w_list"][-1])
    return loss
