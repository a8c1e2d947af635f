"""`NeuRADHashEncoding` (reference: field_components/neurad_encoding.py:36-316), static branch.

forward = contraction kernel (isotropic Gaussian already folded in by the caller) + hash gather with
the per-level rescale fused.  Dynamic actors (neurad_encoding.py:191-307) are the second phase of the
build and are rejected loudly here rather than silently ignored.
"""
from dataclasses import dataclass, field
from typing import Dict, Literal, Optional, Tuple

from torch import Tensor, nn

from . import ops
from .encodings import HashEncoding


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
        n_actors = 0 if dynamic_actors is None else getattr(dynamic_actors, "n_actors", 0)
        if n_actors > 0 and not config.disable_actors:
            raise NotImplementedError("dynamic actors are not built yet (SURVEY 8a a10); pass disable_actors=True")
        self.static_scale = float(static_scale)
        self.static_grid = HashEncoding(
            implementation=implementation, features_per_level=config.static.hashgrid_dim,
            num_levels=config.static.num_levels, min_res=config.static.base_res, max_res=config.static.max_res,
            log2_hashmap_size=config.static.log2_hashmap_size)
        self.actor_grids = nn.ModuleList([])
        self.scene_repr_dim = self.static_grid.get_out_dim()

    def get_out_dim(self) -> int:
        return self.scene_repr_dim

    def get_param_groups(self, param_groups: Dict):
        param_groups["hashgrids"] += list(self.static_grid.parameters()) + list(self.actor_grids.parameters())

    def encode_samples(self, ray_samples, level_major: bool = True, sample_major: bool = True
                       ) -> Tuple[Tensor, Tuple[int, int]]:
        """Fast path used by the fields: frustum samples -> raw feature buffer + (stride_n, stride_l).

        = get_fast_isotropic_gaussian(1) (cameras/rays.py:109-124) -> static_contraction
        (neurad_encoding.py:169) -> static_grid -> _rescale_grid_features (:277-280,309-316)."""
        g = self.static_grid
        B, S = ray_samples.shape
        x01, std01 = ops.contract_gaussians(ray_samples.origins, ray_samples.directions, ray_samples.pixel_area,
                                            ray_samples.euclid, self.static_scale)
        buf = ops.hash_encode(x01, g.hash_table, g.scalings, g.log2_hashmap_size, std=std01,
                              level_major=level_major, sample_major=S if sample_major else 0)
        F, n = g.features_per_level, B * S
        return buf, ((F, n * F) if level_major else (g.get_out_dim(), F))

    def forward(self, ray_samples, times: Optional[Tensor] = None, directions: Optional[Tensor] = None
                ) -> Tuple[Tensor, Optional[Tensor]]:
        """Reference-shaped result: features [B*S, L*F] (torch layout) and the (unchanged) directions."""
        buf, _ = self.encode_samples(ray_samples, level_major=False)
        return buf, directions
