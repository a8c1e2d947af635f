"""Drop-in `HashEncoding` / `SHEncoding` (reference: field_components/encodings.py:311-471,760-805).

Same constructor arguments and attributes (`num_levels`, `features_per_level`, `scalings`,
`hash_table` as an nn.Parameter in the torch layout [L*T, F]); `implementation="hip"` runs the
gfx950 kernels.  Semantics follow the reference's *torch* path (every level hashed, ceil/floor
corners, float32 `scalings`), which is what the oracle pins.
"""
from typing import Literal, Optional

import numpy as np
import torch
from torch import Tensor, nn

from . import ops


class HashEncoding(nn.Module):
    def __init__(self, num_levels: int = 16, min_res: int = 16, max_res: int = 1024, log2_hashmap_size: int = 19,
                 features_per_level: int = 2, hash_init_scale: float = 0.001,
                 implementation: Literal["hip"] = "hip", interpolation: Optional[str] = None,
                 n_input_dims: int = 3) -> None:
        super().__init__()
        if implementation != "hip":
            raise ValueError("neuradar_amd only provides implementation='hip'")
        if n_input_dims != 3:
            raise NotImplementedError("4-D (actor-id) hash grids are the tcnn-only path; use one 3-D grid per actor")
        assert interpolation is None or interpolation == "Linear"
        self.in_dim = 3
        self.num_levels = num_levels
        self.min_res = min_res
        self.features_per_level = features_per_level
        self.hash_init_scale = hash_init_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.hash_table_size = 2**log2_hashmap_size
        levels = torch.arange(num_levels)
        self.growth_factor = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1.0
        # same expression as the reference (encodings.py:350): evaluates in float32 -> e.g. 8191 at the top level
        self.register_buffer("scalings", torch.floor(min_res * self.growth_factor**levels))
        table = torch.rand(size=(self.hash_table_size * num_levels, features_per_level)) * 2 - 1
        self.hash_table = nn.Parameter(table * hash_init_scale)

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def forward(self, in_tensor: Tensor, std: Optional[Tensor] = None) -> Tensor:
        """in_tensor [*bs,3] in [0,1] -> [*bs, L*F].  `std` ([*bs] or [*bs,1]) optionally fuses the
        NeuRAD per-level rescale (neurad_encoding.py:309-316) into the gather."""
        x = in_tensor.reshape(-1, 3)
        s = None if std is None else std.reshape(-1).contiguous()
        out = ops.hash_encode(x, self.hash_table, self.scalings, self.log2_hashmap_size, std=s)
        return out.view(*in_tensor.shape[:-1], self.get_out_dim())


class SHEncoding(nn.Module):
    def __init__(self, levels: int = 4, implementation: Literal["hip"] = "hip") -> None:
        super().__init__()
        if levels != 4:
            raise NotImplementedError("the NeuRadar path uses SH degree 4 (fields/neurad_field.py:107)")
        self.levels = levels

    def get_out_dim(self) -> int:
        return self.levels**2

    @torch.no_grad()
    def forward(self, in_tensor: Tensor) -> Tensor:
        return ops.sh4(in_tensor)
