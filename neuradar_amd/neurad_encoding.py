"""`NeuRADHashEncoding` (reference: field_components/neurad_encoding.py:36-316), static branch.

forward = contraction kernel (isotropic Gaussian already folded in by the caller) + hash gather with
the per-level rescale fused.  Dynamic actors (neurad_encoding.py:191-307) are the second phase of the
build and are rejected loudly here rather than silently ignored.
"""
from dataclasses import dataclass, field
from typing import Dict, Literal, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import ops
from .dynamic_actors import pose_inverse
from .encodings import HashEncoding

EPS = 1.0e-7


@dataclass
class StaticSettings:  # neurad_encoding.py:36-47
    hashgrid_dim: int = 4
    num_levels: int = 8
    base_res: int = 32
    max_res: int = 8192
    log2_hashmap_size: int = 22


@dataclass
class ActorSettings:  # neurad_encoding.py:50-68
    flip_prob: float = 0.5
    actor_scale: float = 10.0
    hashgrid_dim: int = 4
    num_levels: int = 4
    base_res: int = 64
    max_res: int = 1024
    log2_hashmap_size: int = 17
    use_4d_hashgrid: bool = True


@dataclass
class NeuRADHashEncodingConfig:  # neurad_encoding.py:71-84
    static: StaticSettings = field(default_factory=StaticSettings)
    actor: ActorSettings = field(default_factory=ActorSettings)
    disable_actors: bool = False
    require_actor_grad: bool = True

    def setup(self, **kwargs) -> "NeuRADHashEncoding":
        return NeuRADHashEncoding(self, **kwargs)


class NeuRADHashEncoding(nn.Module):
    def __init__(self, config: NeuRADHashEncodingConfig, dynamic_actors=None, static_scale: float = 1.0,
                 implementation: Literal["hip"] = "hip") -> None:
        super().__init__()
        self.config = config
        self.implementation = implementation
        self.actors = dynamic_actors
        n_actors = 0 if dynamic_actors is None or config.disable_actors else getattr(dynamic_actors, "n_actors", 0)
        self.n_actors = n_actors
        self.static_scale = float(static_scale)
        self.static_grid = HashEncoding(
            implementation=implementation, features_per_level=config.static.hashgrid_dim,
            num_levels=config.static.num_levels, min_res=config.static.base_res, max_res=config.static.max_res,
            log2_hashmap_size=config.static.log2_hashmap_size)
        # the torch path of the reference: one 3-D grid per actor (neurad_encoding.py:117-133); the 4-D
        # (xyz + actor id) grid exists only inside tcnn
        self.actor_grids = nn.ModuleList([
            HashEncoding(implementation=implementation, features_per_level=config.actor.hashgrid_dim,
                         num_levels=config.actor.num_levels, min_res=config.actor.base_res, max_res=config.actor.max_res,
                         log2_hashmap_size=config.actor.log2_hashmap_size) for _ in range(n_actors)])
        self.scene_repr_dim = self.static_grid.get_out_dim()

    def get_out_dim(self) -> int:
        return self.scene_repr_dim

    def get_param_groups(self, param_groups: Dict):
        param_groups["hashgrids"] += list(self.static_grid.parameters()) + list(self.actor_grids.parameters())

    def encode_samples(self, ray_samples, level_major: bool = True, sample_major: bool = True,
                       directions: bool = False, flip: Optional[Tensor] = None, rows_sample_major: bool = False):
        """Fast path used by the fields: frustum samples -> raw feature buffer + (stride_n, stride_l)
        (+ per-sample directions [B*S,3] when actors changed them, else None) + whether the buffer's rows
        are sample-major (row s*B+b; granted on request when no actor grid has to be written over it).

        = get_fast_isotropic_gaussian(1) (cameras/rays.py:109-124) -> static_contraction
        (neurad_encoding.py:169) -> static_grid -> _rescale_grid_features (:277-280,309-316) -> actor
        features written over the static ones (:175-187)."""
        g = self.static_grid
        B, S = ray_samples.shape
        rows_sm = rows_sample_major and self.n_actors == 0
        x01, std01 = ops.contract_gaussians(ray_samples.origins, ray_samples.directions, ray_samples.pixel_area,
                                            ray_samples.euclid, self.static_scale, sample_major_rows=rows_sm)
        buf = ops.hash_encode(x01, g.hash_table, g.scalings, g.log2_hashmap_size, std=std01,
                              level_major=level_major, sample_major=S if (sample_major and not rows_sm) else 0)
        F_, n = g.features_per_level, B * S
        strides = (F_, n * F_) if level_major else (g.get_out_dim(), F_)
        dirs = None
        if self.n_actors > 0:
            dirs = self._overwrite_actor_features(buf, level_major, ray_samples, directions, flip)
        return buf, strides, dirs, rows_sm

    def forward(self, ray_samples, times: Optional[Tensor] = None, directions: Optional[Tensor] = None
                ) -> Tuple[Tensor, Optional[Tensor]]:
        """Reference-shaped result: features [B*S, L*F] (torch layout) and the directions."""
        buf, _, dirs, _ = self.encode_samples(ray_samples, level_major=False, directions=directions is not None)
        return buf, (dirs if dirs is not None else directions)

    # ------------------------------------------------------------------------------------------ actors
    @torch.no_grad()
    def _get_actor_indices(self, pos: Tensor, b2w: Tensor, valid: Tensor, w2b: Tensor):
        """neurad_encoding.py:231-275: rays are culled by the distance of the box centre to the
        first->last-sample line, samples by their distance to the centre, then the exact box test."""
        bounds = self.actors.actor_bounds()
        radii = bounds.norm(dim=-1)
        p0 = pos[:, 0, :]
        line = pos[:, -1, :] - p0
        line = (line / (torch.linalg.norm(line, dim=-1, keepdim=True) + EPS)).unsqueeze(-2)
        from_line = b2w[..., :3, 3] - p0.unsqueeze(-2)
        dist = torch.linalg.norm(torch.cross(from_line, line.expand_as(from_line), dim=-1), dim=-1)
        ray_idx, actor_idx = ((dist < radii) & valid).nonzero(as_tuple=False).T
        if ray_idx.shape[0] == 0:
            return None
        centre = b2w[ray_idx, actor_idx, :3, 3].unsqueeze(-2)
        within = (torch.linalg.norm(pos[ray_idx] - centre, dim=-1) < radii[actor_idx].unsqueeze(-1)).nonzero(as_tuple=False)
        idx = torch.stack([ray_idx[within[:, 0]], within[:, 1], actor_idx[within[:, 0]]], dim=-1)
        sel = w2b[idx[:, 0], idx[:, 2]]
        in_box = (pos[idx[:, 0], idx[:, 1]].unsqueeze(-2) @ sel[..., :3, :3].swapaxes(-2, -1)).squeeze(-2) + sel[..., :3, 3]
        idx = idx[(in_box.abs() < bounds[idx[:, 2]]).all(dim=-1)]
        return (idx[:, 0], idx[:, 1], idx[:, 2]) if idx.shape[0] else None

    def _overwrite_actor_features(self, buf: Tensor, level_major: bool, rs, want_dirs: bool, flip: Optional[Tensor]):
        """neurad_encoding.py:175-229,295-307.  Returns per-sample directions [B*S,3] (or None)."""
        B, S = rs.shape
        # sample centres and isotropic std (cameras/rays.py:109-124); cheap torch ops, only the culling
        # needs all of them
        e0, e1 = rs.euclid[:, :-1], rs.euclid[:, 1:]
        half = (e1 - e0) / 2
        t = e0 + half
        mean = rs.origins[:, None, :] + rs.directions[:, None, :] * t[..., None]
        std = (rs.pixel_area[:, None, :] * t[..., None] ** 2 * half[..., None]).pow(1 / 3)
        cfg = self.config
        with torch.enable_grad() if cfg.require_actor_grad else torch.no_grad():
            b2w, valid = self.actors.get_boxes2world(rs.times[:, 0])
            w2b = pose_inverse(b2w)
            found = self._get_actor_indices(mean, b2w, valid, w2b)
            dirs_full = rs.directions[:, None, :].expand(B, S, 3).reshape(B * S, 3) if want_dirs else None
            if found is None:
                return dirs_full
            ray_idx, sample_idx, actor_idx = found
            sel = w2b[ray_idx, actor_idx]
            rot, trans = sel[..., :3, :3], sel[..., :3, 3]
            pos = (mean[ray_idx, sample_idx].unsqueeze(-2) @ rot.swapaxes(-2, -1)).squeeze(-2) + trans
            flat = ray_idx * S + sample_idx
            sign = None
            if self.training and cfg.actor.flip_prob > EPS:  # per-ray random x-flip (:218-225)
                if flip is None:
                    flip = torch.bernoulli(torch.full((B,), cfg.actor.flip_prob, device=pos.device)) * -2 + 1
                sign = flip[ray_idx]
                pos = torch.cat([pos[:, :1] * sign[:, None], pos[:, 1:]], dim=-1)
            if want_dirs:
                d = (rs.directions[ray_idx].unsqueeze(-2) @ rot.swapaxes(-2, -1)).squeeze(-2)
                d = d / (torch.linalg.norm(d, dim=-1, keepdim=True) + EPS)
                if sign is not None:
                    d = torch.cat([d[:, :1] * sign[:, None], d[:, 1:]], dim=-1)
                dirs_full = dirs_full.clone()
                dirs_full[flat] = d
        # actor_contraction (ScaledSceneContraction, scale = actor_scale) in torch: the positions carry
        # the trajectory gradient (spatial_distortions.py:103-136)
        x = pos / cfg.actor.actor_scale
        s = std[ray_idx, sample_idx] / cfg.actor.actor_scale
        mag = x.abs().amax(dim=-1, keepdim=True)
        m = mag.clamp_min(1.0)
        x = torch.where(mag < 1, x, (2 - (1 / m)) * (x / m))
        s = torch.where(mag < 1, s, s * ((2 * m - 1).pow(1 / 3) / m) ** 2)
        x01, s01 = (x + 2.0) / 4.0, (s / 4.0)[:, 0]
        out = None
        for a in actor_idx.unique().tolist():  # one 3-D grid per actor (_get_actor_features_slow)
            grid: HashEncoding = self.actor_grids[int(self.actors.actor_to_id[a])]
            mask = actor_idx == a
            feats = ops.hash_encode(x01[mask].contiguous(), grid.hash_table, grid.scalings, grid.log2_hashmap_size,
                                    std=s01[mask].contiguous())
            if out is None:
                out = torch.zeros((actor_idx.shape[0], feats.shape[-1]), device=feats.device)
            out = out.index_put((mask.nonzero(as_tuple=True)[0],), feats)
        padded = F.pad(out, (0, self.scene_repr_dim - out.shape[-1]))
        g = self.static_grid
        if level_major:  # buf is [L, N, F]: write through its [N, L, F] view
            buf.permute(1, 0, 2)[flat] = padded.view(-1, g.num_levels, g.features_per_level)
        else:
            buf[flat] = padded
        return dirs_full
