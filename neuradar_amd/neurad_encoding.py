"""`NeuRADHashEncoding` (reference: field_components/neurad_encoding.py:36-316).

forward = contraction kernel (isotropic Gaussian already folded in by the caller) + hash gather with
the per-level rescale fused; with dynamic actors (neurad_encoding.py:191-307) the samples inside actor boxes are then
found and re-encoded on the device by fixed-shape launches (csrc/actors.hip: no `nonzero`, no host read, no Python loop
over actors).
"""
from dataclasses import dataclass, field
from typing import Dict, Literal, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import ops
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
    layout: str = "torch"
    """"torch": the reference's torch tables ([L*T, F], all levels hashed; one 3-D grid per actor) -- the layout of the hot
    path.  "tcnn": tiny-cuda-nn's function and parameter layout (tcnn_compat.TcnnHashEncoding), with the single 4-D
    xyz + actor-id grid when actor.use_4d_hashgrid -- for evaluating / fine-tuning tables trained through the reference's
    tcnn path (SURVEY 8f-4); modular path only, trajectories are not optimised through it."""

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
        if config.layout == "tcnn":
            self._build_tcnn_grids(n_actors)
            return
        if config.layout != "torch":
            raise ValueError(f"unknown table layout {config.layout!r}")
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

    # ------------------------------------------------------------------------------------------ tcnn layout (f-4)
    def _build_tcnn_grids(self, n_actors: int) -> None:
        """neurad_encoding.py:104-133 with implementation="tcnn": static 3-D grid; ONE 4-D (xyz, actor id / n_actors) grid for
        all actors when actor.use_4d_hashgrid (:112-116), else one 3-D grid per actor."""
        from .tcnn_compat import TcnnHashEncoding

        c = self.config
        self.static_grid = TcnnHashEncoding(features_per_level=c.static.hashgrid_dim, num_levels=c.static.num_levels,
                                            min_res=c.static.base_res, max_res=c.static.max_res,
                                            log2_hashmap_size=c.static.log2_hashmap_size)
        self.actor_4d = bool(c.actor.use_4d_hashgrid)
        n_grids, dims = ((1, 4) if self.actor_4d else (n_actors, 3)) if n_actors > 0 else (0, 3)
        self.actor_grids = nn.ModuleList([
            TcnnHashEncoding(features_per_level=c.actor.hashgrid_dim, num_levels=c.actor.num_levels, min_res=c.actor.base_res,
                             max_res=c.actor.max_res, log2_hashmap_size=c.actor.log2_hashmap_size, n_input_dims=dims)
            for _ in range(n_grids)])
        self.scene_repr_dim = self.static_grid.get_out_dim()

    @staticmethod
    def _rescale(feats: Tensor, std: Tensor, grid) -> Tensor:
        """_rescale_grid_features (neurad_encoding.py:309-316): per level 1 / max(1, 2 * scaling_l * std)."""
        n, L = feats.shape[0], grid.num_levels
        w = 1.0 / torch.clamp(2.0 * grid.scalings[None, :] * std[:, None], min=1.0)
        return (feats.view(n, L, -1) * w[:, :, None]).view(n, -1)

    def _encode_samples_tcnn(self, rs, level_major: bool, want_dirs: bool, flip: Optional[Tensor]):
        """encode_samples for layout="tcnn": the same chain (:152-189) composed from the tcnn-layout grid op and torch
        elementwise ops; rows are ray-major (row b*S+s)."""
        g = self.static_grid
        B, S = rs.shape
        n, L, F_ = B * S, g.num_levels, g.features_per_level
        x01, std01 = ops.contract_gaussians(rs.origins, rs.directions, rs.pixel_area, rs.euclid, self.static_scale,
                                            sample_major_rows=False)
        feats = self._rescale(g(x01), std01, g)
        dirs = None
        if self.n_actors > 0:
            geom = self.actor_geometry(rs, flip)
            dev = feats.device
            slot = torch.empty(n, device=dev, dtype=torch.int32)
            x01a, std01a = torch.empty(n, 3, device=dev), torch.empty(n, device=dev)
            dirs = torch.empty(n, 3, device=dev) if want_dirs else None
            lib, p = ops._lib.lib(), ops._p
            K = self.MAX_CANDIDATES
            ops.check(lib.nr_actor_assign(p(rs.origins.contiguous()), p(rs.directions.contiguous()), p(rs.pixel_area.reshape(-1).contiguous()),
                                          p(rs.euclid.contiguous()), B, S, 0, p(geom["cand"]), K, p(geom["w2b"].detach()), p(geom["centres"]),
                                          p(geom["bounds"]), self.config.actor.actor_scale, p(geom["flip"]), p(slot), p(x01a), p(std01a),
                                          p(dirs), ops._stream()), "nr_actor_assign")
            inside = slot >= 0
            # rows outside every box keep whatever the buffers held: give them a harmless position (their features are
            # masked out below, their gradient is exactly zero)
            x01a = torch.where(inside[:, None], x01a, torch.zeros_like(x01a))
            std01a = torch.where(inside, std01a, torch.ones_like(std01a))
            ray = torch.arange(n, device=dev) // S
            actor = geom["cand"][ray, slot.clamp(min=0).long()].long()  # actor id of the box the sample sits in
            if getattr(self.actors, "actor_to_id", None) is not None:  # (:183: hash-grid id of the actor)
                actor = self.actors.actor_to_id.to(dev)[actor].long()
            ag = self.actor_grids[0]
            if self.actor_4d:  # _get_actor_features_fast (:282-293): 4th coordinate = actor index / n_actors
                pos4 = torch.cat([x01a, (actor.float() / self.n_actors)[:, None]], dim=-1)
                fa = self._rescale(ag(pos4), std01a, ag)
            else:  # _get_actor_features_slow (:295-307)
                fa = torch.zeros(n, ag.get_out_dim(), device=dev)
                for a, grid in enumerate(self.actor_grids):
                    fa = torch.where((inside & (actor == a))[:, None], self._rescale(grid(x01a), std01a, grid), fa)
            fa = torch.nn.functional.pad(fa, (0, feats.shape[1] - fa.shape[1]))  # zero-padded to scene_repr_dim (:186)
            feats = torch.where(inside[:, None], fa, feats)
        if level_major:
            return feats.view(n, L, F_).permute(1, 0, 2).contiguous(), (F_, n * F_), dirs, False
        return feats, (g.get_out_dim(), F_), dirs, False

    def get_out_dim(self) -> int:
        return self.scene_repr_dim

    def _apply(self, fn, *args, **kwargs):
        """`.to()` / `.cuda()` give every parameter storage of its own again: re-home the actor tables (and their
        gradients) into their shared buffers right away, so that optimizers built afterwards see the final storage."""
        out = super()._apply(fn, *args, **kwargs)
        self._actor_flat = None
        if self.n_actors > 0 and self.config.layout == "torch":
            self._actor_tables()
        return out

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
        if self.config.layout == "tcnn":
            return self._encode_samples_tcnn(ray_samples, level_major, directions, flip)
        g = self.static_grid
        B, S = ray_samples.shape
        rows_sm = bool(rows_sample_major)
        x01, std01 = ops.contract_gaussians(ray_samples.origins, ray_samples.directions, ray_samples.pixel_area,
                                            ray_samples.euclid, self.static_scale, sample_major_rows=rows_sm)
        buf = ops.hash_encode(x01, g.hash_table, g.scalings, g.log2_hashmap_size, std=std01,
                              level_major=level_major, sample_major=S if (sample_major and not rows_sm) else 0)
        F_, n = g.features_per_level, B * S
        strides = (F_, n * F_) if level_major else (g.get_out_dim(), F_)
        dirs = None
        if self.n_actors > 0:
            dirs = self._overwrite_actor_features(buf, level_major, ray_samples, directions, flip,
                                                  sample_major_rows=B if rows_sm else 0)
        return buf, strides, dirs, rows_sm

    def forward(self, ray_samples, times: Optional[Tensor] = None, directions: Optional[Tensor] = None
                ) -> Tuple[Tensor, Optional[Tensor]]:
        """Reference-shaped result: features [B*S, L*F] (torch layout) and the directions."""
        buf, _, dirs, _ = self.encode_samples(ray_samples, level_major=False, directions=directions is not None)
        return buf, (dirs if dirs is not None else directions)

    # ------------------------------------------------------------------------------------------ actors
    MAX_CANDIDATES = 8  # actors whose bounding sphere one ray may cross; further ones are dropped (actor_overflow records it)

    def _actor_tables(self) -> Tensor:
        """The actor grids' tables as ONE buffer [A, L*T, F] the kernels index by actor id; the per-actor
        `hash_table` parameters (state_dict keys unchanged) are views of it.  Re-homed lazily: `.to(device)` and
        `load_state_dict(assign=True)` give every parameter storage of its own again."""
        grids = list(self.actor_grids)
        rows, flat = grids[0].hash_table.shape[0], getattr(self, "_actor_flat", None)
        ok = flat is not None and flat.device == grids[0].hash_table.device and all(
            g.hash_table.data_ptr() == flat.data_ptr() + a * rows * flat.shape[-1] * 4 for a, g in enumerate(grids))
        if not ok:
            flat = torch.stack([g.hash_table.data for g in grids]).contiguous()
            flat_grad = torch.zeros_like(flat)
            for a, g in enumerate(grids):
                if g.hash_table.grad is not None:
                    flat_grad[a].copy_(g.hash_table.grad)
                g.hash_table.data = flat[a]
                g.hash_table.grad = flat_grad[a]
            self._actor_flat, self._actor_flat_grad = flat, flat_grad
        return flat

    def actor_table_grads(self) -> Tensor:
        """Gradient buffer [A, L*T, F] behind the per-actor `hash_table.grad` views (see _actor_tables)."""
        self._actor_tables()
        grids = list(self.actor_grids)
        fg = self._actor_flat_grad
        if any(g.hash_table.grad is None or g.hash_table.grad.data_ptr() != fg[a].data_ptr() for a, g in enumerate(grids)):
            raise RuntimeError("actor table gradients were re-homed (zero_grad(set_to_none=True)?): the fused step needs the "
                               "flat buffer set up by NeuRADHashEncoding._actor_tables()")
        return fg

    def actor_geometry(self, rs, flip: Optional[Tensor] = None, draw_flip: bool = True):
        """Per-RAY part of the actor lookup, shared by every sampling level of a step (fixed shapes, no host read):
        keyframe interval of each ray's time, candidate actors (nr_actor_candidates) and -- in torch, so that autograd
        reaches the trajectories -- the world->box transforms of the (ray, candidate) pairs
        (dynamic_actors.py:183-197, utils/poses.py:90-149)."""
        act, K = self.actors, self.MAX_CANDIDATES
        B = rs.origins.shape[0]
        dev = rs.origins.device
        times = rs.times[:, 0].contiguous()
        ts = act.unique_timestamps
        lib, p = ops._lib.lib(), ops._p
        left, right = torch.empty(B, device=dev, dtype=torch.int64), torch.empty(B, device=dev, dtype=torch.int64)
        frac = torch.empty(B, device=dev, dtype=torch.float32)
        ops.check(lib.nr_actor_keyframes(p(times), B, p(ts.contiguous()), ts.numel(), p(left), p(right), p(frac), ops._stream()),
                  "nr_actor_keyframes")
        bounds = act.actor_bounds().contiguous()
        cand = torch.empty((B, K), device=dev, dtype=torch.int32)
        if not hasattr(self, "actor_overflow") or self.actor_overflow.device != dev:
            self.actor_overflow = torch.zeros(1, device=dev, dtype=torch.int32)
        euclid = rs.euclid.contiguous()
        ops.check(lib.nr_actor_candidates(p(rs.origins.contiguous()), p(rs.directions.contiguous()), p(euclid), B, euclid.shape[1] - 1,
                                          p(left), p(right), p(frac), p(act.actor_positions.detach().contiguous()),
                                          p(act.actor_present_at_time.to(torch.uint8).contiguous()), p(bounds), act.n_actors, K,
                                          p(cand), p(self.actor_overflow), ops._stream()), "nr_actor_candidates")
        with torch.enable_grad() if self.config.require_actor_grad else torch.no_grad():
            w2b, centres = _ActorPoses.apply(act.actor_rotations_6d, act.actor_positions, cand, left, right, frac)
        if flip is None and draw_flip and self.training and self.config.actor.flip_prob > EPS:  # per-ray random x-flip (:218-225)
            flip = torch.bernoulli(torch.full((B,), self.config.actor.flip_prob, device=dev)) * -2 + 1
        if not self.training:
            flip = None
        return dict(cand=cand, w2b=w2b, centres=centres, bounds=bounds, flip=flip)

    def actor_table_ids(self) -> Optional[Tensor]:
        """actors.actor_to_id as int32 [A] (neurad_encoding.py:183: `actor_hashgrid_idx = self.actors.actor_to_id[actor_idx]`;
        the closed-loop server rewrites the buffer, so it is re-read on every call) -- the table_of_actor argument of
        nr_actor_encode_fwd/bwd; None when the actors module carries no such buffer (identity)."""
        ids = getattr(self.actors, "actor_to_id", None)
        return None if ids is None else ids.to(device=self._actor_tables().device, dtype=torch.int32).contiguous()

    def _overwrite_actor_features(self, buf: Tensor, level_major: bool, rs, want_dirs: bool, flip: Optional[Tensor],
                                  sample_major_rows: int = 0):
        """neurad_encoding.py:175-229,295-307 on the device: nr_actor_assign + nr_actor_encode_fwd (backward:
        nr_actor_encode_bwd).  Returns per-sample directions [B*S,3] (or None)."""
        B, S = rs.shape
        geom = self.actor_geometry(rs, flip)
        g = self.static_grid
        ag = self.actor_grids[0]
        n = B * S
        dev = buf.device
        slot = torch.empty(n, device=dev, dtype=torch.int32)
        x01a, std01a = torch.empty(n, 3, device=dev), torch.empty(n, device=dev)
        dirs = torch.empty(n, 3, device=dev) if want_dirs else None
        lib, p = ops._lib.lib(), ops._p
        euclid, area = rs.euclid.contiguous(), rs.pixel_area.reshape(-1).contiguous()
        o, d = rs.origins.contiguous(), rs.directions.contiguous()
        K = self.MAX_CANDIDATES
        ops.check(lib.nr_actor_assign(p(o), p(d), p(area), p(euclid), B, S, sample_major_rows, p(geom["cand"]), K,
                                      p(geom["w2b"].detach()), p(geom["centres"]), p(geom["bounds"]), self.config.actor.actor_scale,
                                      p(geom["flip"]), p(slot), p(x01a), p(std01a), p(dirs), ops._stream()), "nr_actor_assign")
        F_, L_s = g.features_per_level, g.num_levels
        strides = (F_, n * F_) if level_major else (g.get_out_dim(), F_)
        meta = dict(slot=slot, x01a=x01a, std01a=std01a, cand=geom["cand"], K=K, B=B, S=S, sm=sample_major_rows, strides=strides,
                    static_levels=L_s, L=ag.num_levels, F=ag.features_per_level, log2t=ag.log2_hashmap_size, scalings=ag.scalings,
                    table_ids=self.actor_table_ids(), o=o, d=d, area=area, euclid=euclid, flip=geom["flip"], actor_scale=self.config.actor.actor_scale)
        assert ag.features_per_level == F_, "actor grids must have the static grid's features per level"
        _ActorEncode.apply(buf, geom["w2b"], self._actor_tables(), meta, *[gr.hash_table for gr in self.actor_grids])
        return dirs


class _ActorPoses(torch.autograd.Function):
    """nr_actor_w2b_fwd / nr_actor_w2b_bwd: trajectories -> world->box transforms of the (ray, candidate) pairs.  One
    launch each way (the same chain in torch ops -- fancy-index gathers, batched 3x3 matmuls -- cost 12 ms per step in
    its backward alone at 16 384 rays)."""

    @staticmethod
    def forward(ctx, rot6, pos, cand, left, right, frac):
        B, K = cand.shape
        rot6c, posc = rot6.detach().contiguous(), pos.detach().contiguous()
        w2b = torch.empty((B, K, 3, 4), device=cand.device, dtype=torch.float32)
        centres = torch.empty((B, K, 3), device=cand.device, dtype=torch.float32)
        lib, p = ops._lib.lib(), ops._p
        ops.check(lib.nr_actor_w2b_fwd(p(cand), B, K, rot6.shape[1], p(left), p(right), p(frac), p(rot6c), p(posc), p(w2b), p(centres),
                                       ops._stream()), "nr_actor_w2b_fwd")
        ctx.save_for_backward(rot6c, posc, cand, left, right, frac)
        ctx.mark_non_differentiable(centres)
        return w2b, centres

    @staticmethod
    def backward(ctx, g_w2b, _g_centres):
        rot6, pos, cand, left, right, frac = ctx.saved_tensors
        B, K = cand.shape
        g_rot6, g_pos = torch.zeros_like(rot6), torch.zeros_like(pos)
        lib, p = ops._lib.lib(), ops._p
        ops.check(lib.nr_actor_w2b_bwd(p(cand), B, K, rot6.shape[1], p(left), p(right), p(frac), p(rot6), p(pos),
                                       p(g_w2b.contiguous()), p(g_rot6), p(g_pos), ops._stream()), "nr_actor_w2b_bwd")
        return g_rot6, g_pos, None, None, None, None


class _ActorEncode(torch.autograd.Function):
    """nr_actor_encode_fwd / nr_actor_encode_bwd: writes the actor features over rows of `buf` IN PLACE."""

    @staticmethod
    def forward(ctx, buf, w2b, flat, meta, *tables):
        m = meta
        lib, p = ops._lib.lib(), ops._p
        ops.check(lib.nr_actor_encode_fwd(p(m["x01a"]), p(m["std01a"]), p(m["slot"]), p(m["cand"]), m["K"], m["B"], m["S"], m["sm"],
                                          p(flat), p(m["table_ids"]), p(m["scalings"]), m["L"], m["F"], m["log2t"], p(buf), m["strides"][0],
                                          m["strides"][1],
                                          m["static_levels"], ops._stream()), "nr_actor_encode_fwd")
        ctx.mark_dirty(buf)
        ctx.meta, ctx.flat, ctx.w2b = m, flat, w2b.detach()
        ctx.want_pose = w2b.requires_grad
        return buf

    @staticmethod
    def backward(ctx, g_buf):
        m, flat = ctx.meta, ctx.flat
        lib, p = ops._lib.lib(), ops._p
        g_buf = g_buf.contiguous().clone()  # rows of actor samples are zeroed for the static grid's scatter
        g_flat = torch.zeros_like(flat)
        g_w2b = torch.zeros_like(ctx.w2b) if ctx.want_pose else None
        ops.check(lib.nr_actor_encode_bwd(p(m["x01a"]), p(m["std01a"]), p(m["slot"]), p(m["cand"]), m["K"], m["B"], m["S"], m["sm"],
                                          p(flat), p(m["table_ids"]), p(m["scalings"]), m["L"], m["F"], m["log2t"], p(g_buf), m["strides"][0],
                                          m["strides"][1], m["static_levels"], p(g_flat), p(m["o"]), p(m["d"]), p(m["area"]),
                                          p(m["euclid"]), p(ctx.w2b), m["actor_scale"], p(m["flip"]), p(g_w2b), ops._stream()),
                  "nr_actor_encode_bwd")
        return (g_buf, g_w2b, None, None, *[g_flat[a] for a in range(g_flat.shape[0])])
