"""tiny-cuda-nn compatibility: the reference's `implementation="tcnn"` modules keep their parameters in
tiny-cuda-nn's own layouts (`tcnn_encoding.params`), which are neither the torch path's `[L*T, F]` hash table nor its
`nn.Linear` weights -- and for the grid not even the same function (dense indexing on coarse levels, a half-cell
offset, scale = res - 1; field_components/encodings.py:361-401).  This module makes such checkpoints usable:

  * `TcnnHashEncoding` -- the grid with tcnn's function and layout for 3-D and 4-D inputs on HIP kernels
    (nr_tcnn_grid_fwd/bwd), state-dict compatible with the reference's tcnn-backed `HashEncoding`
    (key `tcnn_encoding.params`); the 4-D grid is the actors' xyz + actor-id grid (neurad_encoding.py:112-133,282-293);
  * `fully_fused_mlp_weights` / `load_fully_fused_mlp` -- `tcnn.Network{FullyFusedMLP}` parameter vectors
    (field_components/mlp.py:102-113) -> the `[out, in]` matrices of this package's `MLP` (biases zero: tcnn has none);
  * `load_tcnn_state_dict` -- a reference state dict of tcnn-backed fields into `NeuRADField` / `NeuRADProposalField`
    modules built with `NeuRADHashEncodingConfig(layout="tcnn")`.

tiny-cuda-nn is an unpinned dependency that is not in the image; the restatement is pinned by oracle/tcnn_grid.py, its
self-consistency tests and the published parameter count of the instant-ngp default grid.
"""
from typing import Dict, List, Literal, Optional

import numpy as np
import torch
from torch import Tensor, nn

from . import ops


class _TcnnParams(nn.Module):
    """Holder named like the reference's `tcnn.Encoding` / `tcnn.Network` attribute: state-dict key `<...>.tcnn_encoding.params`."""

    def __init__(self, n: int, init_scale: float) -> None:
        super().__init__()
        self.params = nn.Parameter((torch.rand(n) * 2 - 1) * init_scale)


class _TcnnGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, cfg):
        D, L, F, log2t, base, pls = cfg
        x = x.detach().contiguous().float()
        n = x.shape[0]
        out = torch.empty((n, L * F), device=x.device, dtype=torch.float32)
        lib, p = ops._lib.lib(), ops._p
        ops.check(lib.nr_tcnn_grid_fwd(p(x), p(params.detach()), D, L, F, log2t, base, pls, p(out), n, ops._stream()), "nr_tcnn_grid_fwd")
        ctx.save_for_backward(x)
        ctx.cfg, ctx.n_params = cfg, params.numel()
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        D, L, F, log2t, base, pls = ctx.cfg
        gp = torch.zeros(ctx.n_params, device=g.device, dtype=torch.float32)
        lib, p = ops._lib.lib(), ops._p
        ops.check(lib.nr_tcnn_grid_bwd(p(x), D, L, F, log2t, base, pls, p(g.contiguous()), p(gp), x.shape[0], ops._stream()), "nr_tcnn_grid_bwd")
        return None, gp, None


class TcnnHashEncoding(nn.Module):
    """`HashEncoding(implementation="tcnn", n_input_dims in {3, 4})` of the reference (encodings.py:326-404): same
    constructor arguments and buffers, parameters in tcnn's layout."""

    def __init__(self, num_levels: int = 16, min_res: int = 16, max_res: int = 1024, log2_hashmap_size: int = 19,
                 features_per_level: int = 2, hash_init_scale: float = 0.001, implementation: Literal["hip"] = "hip",
                 interpolation: Optional[str] = None, n_input_dims: int = 3) -> None:
        super().__init__()
        if implementation != "hip":
            raise ValueError("neuradar_amd only provides implementation='hip'")
        if n_input_dims not in (3, 4):
            raise NotImplementedError("tcnn-layout grids: 3-D and 4-D inputs")
        assert interpolation is None or interpolation == "Linear", "only tcnn's default (Linear) interpolation"
        self.in_dim = n_input_dims
        self.num_levels, self.min_res, self.features_per_level = num_levels, min_res, features_per_level
        self.log2_hashmap_size, self.hash_table_size = log2_hashmap_size, 2**log2_hashmap_size
        levels = torch.arange(num_levels)
        self.growth_factor = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1.0
        self.register_buffer("scalings", torch.floor(min_res * self.growth_factor**levels))  # (used by NeuRAD's rescale)
        self._cfg = (n_input_dims, num_levels, features_per_level, log2_hashmap_size, int(min_res), float(self.growth_factor))
        n = ops._lib.lib().nr_tcnn_grid_param_count(*self._cfg)
        if n < 0:
            raise ValueError(f"unsupported tcnn grid configuration {self._cfg}")
        self.tcnn_encoding = _TcnnParams(int(n), 1e-4)  # tcnn initialises grids uniformly in [-1e-4, 1e-4]

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def forward(self, in_tensor: Tensor) -> Tensor:
        x = in_tensor.reshape(-1, self.in_dim)
        return _TcnnGrid.apply(x, self.tcnn_encoding.params, self._cfg).view(*in_tensor.shape[:-1], self.get_out_dim())


def fully_fused_mlp_weights(params: Tensor, in_dim: int, width: int, n_hidden_layers: int, out_dim: int) -> List[Tensor]:
    """`tcnn.Network{FullyFusedMLP}.params` -> per-layer `[out, in]` matrices (no biases exist in tcnn).  Layout: first
    layer [width, pad16(in)], n_hidden_layers - 1 matrices [width, width], last layer [pad16(out), width], row-major,
    concatenated, padding included (num_layers of the reference's `MLP` = n_hidden_layers + 1, mlp.py:116-131)."""
    pad = lambda v: (v + 15) // 16 * 16  # noqa: E731
    sizes = [(width, pad(in_dim))] + [(width, width)] * (n_hidden_layers - 1) + [(pad(out_dim), width)]
    if params.numel() != sum(a * b for a, b in sizes):
        raise ValueError(f"FullyFusedMLP({in_dim}->{width}x{n_hidden_layers}->{out_dim}) has {sum(a * b for a, b in sizes)} "
                         f"parameters, got {params.numel()}")
    out, off = [], 0
    for i, (a, b) in enumerate(sizes):
        w = params[off:off + a * b].view(a, b).float()
        off += a * b
        w = w[:, :in_dim] if i == 0 else w
        w = w[:out_dim] if i == len(sizes) - 1 else w
        out.append(w)
    return out


def load_fully_fused_mlp(mlp: nn.Module, params: Tensor) -> None:
    """Copy a FullyFusedMLP parameter vector into an `MLP` of this package (its `layers` of nn.Linear)."""
    layers = [m for m in mlp.layers if isinstance(m, nn.Linear)]
    ws = fully_fused_mlp_weights(params, layers[0].in_features, layers[0].out_features, len(layers) - 1, layers[-1].out_features)
    with torch.no_grad():
        for lin, w in zip(layers, ws):
            lin.weight.copy_(w.to(lin.weight))
            if lin.bias is not None:
                lin.bias.zero_()


def load_tcnn_state_dict(module: nn.Module, state_dict: Dict[str, Tensor], prefix: str = "") -> List[str]:
    """Load a reference state dict whose hash grids and MLPs are tcnn-backed into `module` (a field of this package built
    with `layout="tcnn"`).  Returns the keys that were consumed.  Grid vectors go in as they are; `*.tcnn_encoding.params`
    of MLPs are unpacked into the Linear layers; everything else (`sdf_to_density.beta`, `density_decoder.weight`, ...) is a
    plain copy."""
    used = []
    own = dict(module.named_parameters())
    own.update(dict(module.named_buffers()))
    mods = dict(module.named_modules())
    for key, val in state_dict.items():
        if not key.startswith(prefix):
            continue
        k = key[len(prefix):]
        if k.endswith(".tcnn_encoding.params"):
            owner = mods.get(k[: -len(".tcnn_encoding.params")])
            if owner is None:
                continue
            if isinstance(owner, TcnnHashEncoding):
                with torch.no_grad():
                    owner.tcnn_encoding.params.copy_(val.float().reshape(-1))
            else:
                load_fully_fused_mlp(owner, val.reshape(-1))
            used.append(key)
        elif k in own and own[k].shape == val.shape:
            with torch.no_grad():
                own[k].copy_(val.to(own[k]))
            used.append(key)
    return used
