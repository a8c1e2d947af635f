"""Drop-in `MLP` (reference: field_components/mlp.py:60-183) on the MFMA kernels.

Parameters are ordinary `nn.Linear`-layout tensors (`layers[i].weight [out,in]`, `.bias [out]`) so
optimizer groups and state_dicts are interchangeable with the reference's torch implementation.
"""
from typing import Literal, Optional

from torch import Tensor, nn

from . import ops


class MLP(nn.Module):
    def __init__(self, in_dim: int, num_layers: int, layer_width: int, out_dim: Optional[int] = None,
                 skip_connections=None, activation: Optional[nn.Module] = nn.ReLU(),
                 out_activation: Optional[nn.Module] = None, implementation: Literal["hip"] = "hip") -> None:
        super().__init__()
        if skip_connections:
            raise NotImplementedError("skip connections are unused on the NeuRadar path")
        if not isinstance(activation, nn.ReLU):
            raise NotImplementedError("the fused kernels implement ReLU hidden activations (NeuRadar's only choice)")
        self.in_dim, self.num_layers, self.layer_width = in_dim, num_layers, layer_width
        self.out_dim = out_dim if out_dim is not None else layer_width
        self.out_activation = out_activation
        layers = []
        if num_layers == 1:
            layers.append(nn.Linear(in_dim, self.out_dim))
        else:
            layers.append(nn.Linear(in_dim, layer_width))
            layers += [nn.Linear(layer_width, layer_width) for _ in range(num_layers - 2)]
            layers.append(nn.Linear(layer_width, self.out_dim))
        self.layers = nn.ModuleList(layers)

    def weights(self):
        return [l.weight for l in self.layers], [l.bias for l in self.layers]

    def forward(self, in_tensor: Tensor) -> Tensor:
        ws, bs = self.weights()
        y = ops.mlp(in_tensor.reshape(-1, self.in_dim), ws, bs).view(*in_tensor.shape[:-1], self.out_dim)
        return self.out_activation(y) if self.out_activation is not None else y
