"""Oracle: dynamic actors.  Test infrastructure only.

Restates model_components/dynamic_actors.py:98-197 (learnable per-timestep actor poses, 6-D rotation),
utils/poses.py:35-49,90-149 (pose inverse, trajectory interpolation), cameras/camera_utils.py:422-443
(rotation_6d_to_matrix) and field_components/neurad_encoding.py:175-275,295-307 (ray/sample culling
against actor boxes, world->box transform, per-ray flip, one 3-D hash grid per actor, overwrite of the
static features).
"""
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn.functional as F

from . import field as ofield
from . import hashgrid

EPS = 1.0e-7  # neurad_encoding.py:33


@dataclass
class ActorState:
    """Buffers/parameters of `DynamicActors` (dynamic_actors.py:98-147)."""

    positions: torch.Tensor  # [T,A,3]   actor_positions
    rotations_6d: torch.Tensor  # [T,A,6] actor_rotations_6d
    timestamps: torch.Tensor  # [T]      unique_timestamps
    present: torch.Tensor  # [T,A] bool  actor_present_at_time
    sizes: torch.Tensor  # [A,3] wlh
    padding: torch.Tensor  # [3]

    def bounds(self):  # dynamic_actors.py:95-96
        return self.sizes / 2 + self.padding


def rotation_6d_to_matrix(d6):
    """camera_utils.py:422-443: Gram-Schmidt, the three vectors become the matrix ROWS."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def pose_inverse(pose):
    """utils/poses.py:35-49 -> [...,3,4]."""
    R, t = pose[..., :3, :3], pose[..., :3, 3:]
    Rt = R.transpose(-2, -1)
    return torch.cat([Rt, -Rt.matmul(t)], dim=-1)


def boxes2world(state: ActorState, query_times):
    """DynamicActors.get_boxes2world(flatten=False): dynamic_actors.py:183-197 over
    interpolate_trajectories_6d (poses.py:90-149).  query_times [B] -> ([B,A,4,4], valid [B,A])."""
    poses = torch.cat([state.rotations_6d, state.positions], dim=-1)  # [T,A,9]
    a1 = F.normalize(poses[..., :3], dim=-1)
    a2 = poses[..., 3:6]
    a2 = F.normalize(a2 - (a1 * a2).sum(-1, keepdim=True) * a1, dim=-1)
    poses = torch.cat([a1, a2, poses[..., 6:9]], dim=-1)
    right = torch.searchsorted(state.timestamps, query_times)
    left = (right - 1).clamp(min=0)
    right = right.clamp(max=len(state.timestamps) - 1)
    t_r, t_l = state.timestamps[right], state.timestamps[left]
    frac = ((query_times - t_l) / (t_r - t_l + 1e-6)).clamp(0.0, 1.0)
    valid = state.present[left] | state.present[right]
    interp = poses[left] + (poses[right] - poses[left]) * frac[:, None, None]
    b2w = torch.cat([rotation_6d_to_matrix(interp[..., :6]), interp[..., 6:].unsqueeze(-1)], dim=-1)  # [B,A,3,4]
    bottom = torch.zeros_like(b2w[..., :1, :])
    bottom[..., 3] = 1
    return torch.cat([b2w, bottom], dim=-2), valid


def transform_points_pairwise(points, transforms, with_translation=True):
    """cameras/lidars.py:506-519."""
    out = (points.unsqueeze(-2) @ transforms[..., :3, :3].swapaxes(-2, -1)).squeeze(-2)
    return out + transforms[..., :3, 3] if with_translation else out


@torch.no_grad()
def actor_indices(sample_pos, b2w, valid, w2b, bounds):
    """neurad_encoding.py:231-275: (ray, sample, actor) triples of samples inside an actor's box.

    Cull rays by distance of the box centre to the ray's first->last sample line (< box radius),
    then samples by centre distance, then the exact inside-box test in the box frame."""
    radii = bounds.norm(dim=-1)
    p0 = sample_pos[:, 0, :]
    line = sample_pos[:, -1, :] - p0
    line = (line / (torch.linalg.norm(line, dim=-1, keepdim=True) + EPS)).unsqueeze(-2)
    from_line = b2w[..., :3, 3] - p0.unsqueeze(-2)
    dist = torch.linalg.norm(torch.cross(from_line, line.expand_as(from_line), dim=-1), dim=-1)
    ray_idx, actor_idx = ((dist < radii) & valid).nonzero(as_tuple=False).T
    empty = torch.empty(0, dtype=torch.int64)
    if ray_idx.shape[0] == 0:
        return empty, empty, empty
    centre = b2w[ray_idx, actor_idx, :3, 3].unsqueeze(-2)
    within = (torch.linalg.norm(sample_pos[ray_idx] - centre, dim=-1) < radii[actor_idx].unsqueeze(-1)).nonzero(as_tuple=False)
    idx = torch.stack([ray_idx[within[:, 0]], within[:, 1], actor_idx[within[:, 0]]], dim=-1)
    in_box = transform_points_pairwise(sample_pos[idx[:, 0], idx[:, 1]], w2b[idx[:, 0], idx[:, 2]])
    idx = idx[(in_box.abs() < bounds[idx[:, 2]]).all(dim=-1)]
    return idx[:, 0], idx[:, 1], idx[:, 2]


def overwrite_actor_features(feats, dirs, mean, std, actor_grids: List[ofield.GridParams], ctx: dict):
    """NeuRADHashEncoding.forward, actor branch (neurad_encoding.py:175-187,191-229,295-307).

    feats [B,S,D] static features (returned tensor has the actor samples overwritten, zero padded),
    dirs [B,S,3] or None, mean [B,S,3], std [B,S,1].  ctx: state (ActorState), times [B],
    actor_scale (10.0), flip (None or [B] of +-1, the per-ray random x-flip), require_grad (bool).
    """
    state: ActorState = ctx["state"]
    if state.sizes.shape[0] == 0:
        return feats, dirs
    grad_ctx = torch.enable_grad() if ctx.get("require_grad", True) else torch.no_grad()
    with grad_ctx:
        b2w, valid = boxes2world(state, ctx["times"])
        w2b = pose_inverse(b2w)
        ray_idx, sample_idx, actor_idx = actor_indices(mean, b2w, valid, w2b, state.bounds())
        if ray_idx.shape[0] == 0:
            return feats, dirs
        sel = w2b[ray_idx, actor_idx]
        pos = transform_points_pairwise(mean[ray_idx, sample_idx][:, None, :], sel.unsqueeze(-3))  # [Na,1,3]
        if dirs is not None:
            dirs = dirs.clone()
            d = transform_points_pairwise(dirs[ray_idx, sample_idx], sel, with_translation=False)
            dirs[ray_idx, sample_idx] = d / (torch.linalg.norm(d, dim=-1, keepdim=True) + EPS)
        flip = ctx.get("flip")
        if flip is not None:
            f = torch.ones_like(pos[..., 0:1, :])
            f[..., 0] = flip[ray_idx].unsqueeze(-1)
            pos = pos * f
            if dirs is not None:
                dirs[ray_idx, sample_idx, 0] = dirs[ray_idx, sample_idx, 0] * f[..., 0].squeeze(-1)
    x01, s01 = ofield.scaled_contraction(pos[:, 0, :], std[ray_idx, sample_idx], ctx.get("actor_scale", 10.0))
    out = None
    for a in actor_idx.unique():
        grid = actor_grids[int(a)]
        m = actor_idx == a
        raw = hashgrid.encode(x01[m], grid.table, grid.scalings, grid.table_size)
        af = ofield.rescale_grid_features(raw, s01[m], grid)
        if out is None:
            out = torch.zeros((actor_idx.shape[0], af.shape[-1]), dtype=af.dtype)
        out[m] = af
    feats = feats.clone()
    feats[ray_idx, sample_idx] = F.pad(out, (0, feats.shape[-1] - out.shape[-1]))
    return feats, dirs
