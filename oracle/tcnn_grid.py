"""TEST INFRASTRUCTURE (imported only by tests/): CPU restatement of tiny-cuda-nn's multiresolution hash grid
(`GridEncoding`, grid type "Hash", interpolation "Linear") for 3-D and 4-D inputs, and of the parameter layout of its
`FullyFusedMLP`.

The reference reaches this algorithm through `tcnn.Encoding(n_input_dims, encoding_config)` /
`tcnn.Network(...)` (field_components/encodings.py:361-373,468-471 with the config of :386-401; the 4-D grid of the actors:
field_components/neurad_encoding.py:112-133,282-293; field_components/mlp.py:102-113).  tiny-cuda-nn is an un-vendored,
UNPINNED dependency (`pip install git+https://github.com/NVlabs/tiny-cuda-nn.git#subdirectory=bindings/torch`,
reference Dockerfile:112) and is not in this image, so this file restates its published algorithm
(include/tiny-cuda-nn/encodings/grid.h, common_device.h of the master branch) and **parity is unpinned**: there is no
golden vector of the real library to check it against.  What anchors it: the reference's call sites (which
configuration values reach the library and how the 4-th coordinate is formed) and the self-consistency tests in
tests/test_tcnn_grid.py (dense levels reproduce an explicit n-linear interpolation of a dense array; partition of unity).

Algorithm, per level l:
    scale_l      = exp2f(l * log2f(per_level_scale)) * base_resolution - 1            (float32)
    resolution_l = ceilf(scale_l) + 1
    size_l       = min(next_multiple(resolution_l ** D, 8), 2 ** log2_hashmap_size)    (entries; levels are packed)
    pos          = x * scale_l + 0.5;  cell = floor(pos);  w = pos - cell              (x in [0, 1])
    index(c)     = sum_d c_d * resolution_l ** d, summed while the stride stays <= size_l; if the last stride exceeds
                   size_l the index is the hash  xor_d c_d * prime_d  (uint32) instead; finally  % size_l
    out_l        = sum over the 2 ** D corners of  prod_d (w_d or 1 - w_d) * params[offset_l + index(corner)]
"""
import math
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737)


@dataclass
class GridGeometry:
    n_dims: int
    scales: List[float]        # float32 values
    resolutions: List[int]
    offsets: List[int]         # entries, len L + 1
    features: int

    @property
    def n_params(self) -> int:
        return self.offsets[-1] * self.features


def geometry(n_dims: int, num_levels: int, features: int, log2_hashmap_size: int, base_resolution: int, per_level_scale: float) -> GridGeometry:
    f32 = np.float32
    log2_pls = np.log2(f32(per_level_scale))  # log2f / exp2f in float32, as tiny-cuda-nn's host and device code
    scales, res, offsets, off = [], [], [0], 0
    max_params = (2**32 - 1) // 2
    for lvl in range(num_levels):
        scale = f32(f32(np.exp2(f32(lvl) * log2_pls)) * f32(base_resolution) - f32(1.0))
        r = int(math.ceil(float(scale))) + 1
        n = max_params if float(r) ** n_dims > float(max_params) else r**n_dims
        n = (n + 7) // 8 * 8
        n = min(n, 1 << log2_hashmap_size)
        scales.append(float(scale))
        res.append(r)
        off += n
        offsets.append(off)
    return GridGeometry(n_dims, scales, res, offsets, features)


def _index(cell: torch.Tensor, resolution: int, size: int) -> torch.Tensor:
    """cell [n, D] int64 (non-negative) -> entry index [n] with uint32 semantics."""
    m = 0xFFFFFFFF
    D = cell.shape[1]
    stride, index = 1, torch.zeros(cell.shape[0], dtype=torch.int64)
    d = 0
    while d < D and stride <= size:
        index = (index + cell[:, d] * stride) & m
        stride = (stride * resolution) & m
        d += 1
    if size < stride:
        index = torch.zeros(cell.shape[0], dtype=torch.int64)
        for k in range(D):
            index = index ^ ((cell[:, k] * PRIMES[k]) & m)
    return index % size


def encode(x: torch.Tensor, params: torch.Tensor, g: GridGeometry) -> torch.Tensor:
    """x [n, D] in [0, 1], params [n_params] (entry-major, F contiguous) -> [n, L * F]; differentiable in params."""
    n, D, F = x.shape[0], g.n_dims, g.features
    table = params.view(-1, F)
    outs = []
    for lvl, (scale, r) in enumerate(zip(g.scales, g.resolutions)):
        size = g.offsets[lvl + 1] - g.offsets[lvl]
        pos = (x.double() * scale + 0.5).float()  # fmaf(scale, x, 0.5f): the float64 product is exact, one rounding
        cell = torch.floor(pos)
        w = pos - cell
        cell = cell.long()
        acc = torch.zeros(n, F, dtype=params.dtype)
        for corner in range(1 << D):
            weight = torch.ones(n, dtype=torch.float32)
            c = cell.clone()
            for d in range(D):
                if corner & (1 << d):
                    weight = weight * w[:, d]
                    c[:, d] += 1
                else:
                    weight = weight * (1 - w[:, d])
            idx = _index(c, r, size) + g.offsets[lvl]
            acc = acc + weight[:, None] * table[idx]
        outs.append(acc)
    return torch.cat(outs, dim=-1)


def mlp_weights(params: torch.Tensor, in_dim: int, width: int, n_hidden_layers: int, out_dim: int) -> List[torch.Tensor]:
    """FullyFusedMLP parameter vector -> per-layer weight matrices [out, in] (no biases exist).  Layout of
    tiny-cuda-nn's fully_fused_mlp.cu: first layer [width, pad16(in)], (n_hidden_layers - 1) matrices [width, width],
    last layer [pad16(out), width], each row-major, concatenated; padded rows / columns are present in the vector."""
    pad = lambda v: (v + 15) // 16 * 16  # noqa: E731
    sizes = [(width, pad(in_dim))] + [(width, width)] * (n_hidden_layers - 1) + [(pad(out_dim), width)]
    assert params.numel() == sum(a * b for a, b in sizes), (params.numel(), sizes)
    out, off = [], 0
    for i, (a, b) in enumerate(sizes):
        w = params[off:off + a * b].view(a, b)
        off += a * b
        if i == 0:
            w = w[:, :in_dim]
        if i == len(sizes) - 1:
            w = w[:out_dim]
        out.append(w)
    return out
