"""Ray data structures of the path (reference: cameras/rays.py:33-357).

Flat, device-resident tensors instead of broadcast [B,S,...] dataclasses: a `RaySamples` keeps the
per-ray quantities once ([B,3] origins/directions, [B,1] pixel_area) plus the S+1 bin EDGES in
normalised s-space (`spacing`) and in metres (`euclid`).  The reference-shaped views the callers and
losses read (`frustums.starts/ends`, `deltas`, `spacing_starts/ends`, `[..., :-1]` slicing,
`spacing_to_euclidean_fn`) are provided as properties over the same storage.
"""
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from . import ops


@dataclass
class RayBundle:
    """cameras/rays.py:251-273 (fields used on the NeuRadar path)."""

    origins: Tensor  # [B,3]
    directions: Tensor  # [B,3]
    pixel_area: Tensor  # [B,1]
    camera_indices: Optional[Tensor] = None
    nears: Optional[Tensor] = None  # [B,1]
    fars: Optional[Tensor] = None  # [B,1]
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None  # [B,1]

    def __len__(self) -> int:
        return self.origins.shape[0]


class _FrustumsView:
    """`RaySamples.frustums` with the reference's shapes ([B,S,1] starts/ends, [B,1,3] origins)."""

    def __init__(self, rs: "RaySamples"):
        self._rs = rs

    @property
    def starts(self) -> Tensor:
        return self._rs.euclid[:, :-1, None]

    @property
    def ends(self) -> Tensor:
        return self._rs.euclid[:, 1:, None]

    @property
    def origins(self) -> Tensor:
        return self._rs.origins[:, None, :]

    @property
    def directions(self) -> Tensor:
        return self._rs.directions[:, None, :]

    @property
    def pixel_area(self) -> Tensor:
        return self._rs.pixel_area[:, None, :]

    def get_positions(self) -> Tensor:
        return self.origins + self.directions * (self.starts + self.ends) / 2


@dataclass
class RaySamples:
    """cameras/rays.py:142-184.  `spacing`/`euclid` are [B,S+1] bin edges."""

    origins: Tensor
    directions: Tensor
    pixel_area: Tensor
    spacing: Tensor
    euclid: Tensor
    nears: Tensor
    fars: Tensor
    times: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)  # [B,S,C] broadcast views of the bundle's [B,C] entries
    camera_indices: Optional[Tensor] = None

    @property
    def num_samples(self) -> int:
        return self.euclid.shape[1] - 1

    @property
    def shape(self):
        return (self.euclid.shape[0], self.num_samples)

    @property
    def frustums(self) -> _FrustumsView:
        return _FrustumsView(self)

    @property
    def deltas(self) -> Tensor:
        return (self.euclid[:, 1:] - self.euclid[:, :-1])[..., None]

    @property
    def spacing_starts(self) -> Tensor:
        return self.spacing[:, :-1, None]

    @property
    def spacing_ends(self) -> Tensor:
        return self.spacing[:, 1:, None]

    @property
    def spacing_to_euclidean_fn(self) -> Callable[[Tensor], Tensor]:
        """ray_samplers.py:119-120 for the PowerSampler (utils/math.py:541-579), as a torch closure."""
        lam, scaling = ops.POWER_LAMBDA, ops.POWER_SCALING
        lam_1 = abs(lam - 1)

        def power(x):
            return (lam_1 / lam) * ((x / lam_1 + 1) ** lam - 1)

        s_near, s_far = power(self.nears * scaling), power(self.fars * scaling)

        def fn(s):
            x = s * s_far + (1 - s) * s_near
            return ((x * lam / lam_1 + 1).clamp_min(1e-10) ** (1 / lam) - 1) * lam_1 / scaling

        return fn

    def drop_last(self) -> "RaySamples":
        """`ray_samples[..., :-1]` (models/neuradar.py:515): discard the last (sky) sample."""
        meta = {k: v[:, :-1] for k, v in self.metadata.items()}
        return RaySamples(self.origins, self.directions, self.pixel_area, self.spacing[:, :-1], self.euclid[:, :-1],
                          self.nears, self.fars, self.times, meta, self.camera_indices)

    def __getitem__(self, idx):
        if idx == (Ellipsis, slice(None, -1, None)):
            return self.drop_last()
        raise NotImplementedError("only ray_samples[..., :-1] is supported")

    def get_weights(self, densities: Tensor) -> Tensor:
        """cameras/rays.py:188-210: densities [B,S,1] -> weights [B,S,1] (wavefront-scan kernel)."""
        return ops.weights_from_density(densities[..., 0], self.euclid)[..., None]
