"""`NeuRADField` / `NeuRADProposalField` (reference: fields/neurad_field.py:44-216) on the HIP kernels.

Same config dataclasses, parameter names (`hashgrid.static_grid.hash_table`, `mlp_geo.layers.N.*`,
`mlp_feature.layers.N.*`, `sdf_to_density.beta`, `density_decoder.weight`) and outputs
(`{FEATURE [B,S,C], SDF [B,S,1], ALPHA [B,S,1]}` / `(density [B,S,1], None)`).
"""
from dataclasses import dataclass, field
from typing import Dict, Literal, Optional, Tuple

import torch
from torch import Tensor, nn

from . import ops
from .field_heads import FieldHeadNames
from .mlp import MLP
from .neurad_encoding import ActorSettings, NeuRADHashEncoding, NeuRADHashEncodingConfig, StaticSettings
from .rays import RaySamples


class SigmoidDensity(nn.Module):
    """model_components/utils.py:21-46 (parameter container; the sigmoid is fused into nr_field_fwd)."""

    def __init__(self, init_val: float, beta_min: float = 0.0001, learnable_beta: bool = False):
        super().__init__()
        self.register_buffer("beta_min", torch.tensor(beta_min))
        self.beta = nn.Parameter(init_val * torch.ones(1), requires_grad=learnable_beta)

    def get_beta(self) -> Tensor:
        return self.beta.abs() + self.beta_min


@dataclass
class NeuRADFieldConfig:  # neurad_field.py:44-75
    grid: NeuRADHashEncodingConfig = field(
        default_factory=lambda: NeuRADHashEncodingConfig(require_actor_grad=True, actor=ActorSettings(flip_prob=0.25)))
    geo_hidden_dim: int = 32
    geo_num_layers: int = 2
    nff_hidden_dim: int = 32
    nff_num_layers: int = 3
    nff_out_dim: int = 32
    num_multisamples: int = 1
    use_sdf: bool = True
    sdf_beta: float = 20.0
    learnable_beta: bool = True
    # precision of the MLP stack's MFMA operands (this build; the reference reaches the same thing through
    # torch.autocast + tcnn's fp16 FullyFusedMLP, engine/trainer.py:189-200,564): "float32" is the 1e-4 parity path,
    # "bfloat16" / "float16" run on v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation.  mlp_grad_scale: static loss
    # scale applied where gradients enter the 16-bit domain (the role of trainer.py's GradScaler; fp16 needs one).
    mlp_dtype: str = "float32"
    mlp_grad_scale: float = 1.0

    def setup(self, **kwargs) -> "NeuRADField":
        return NeuRADField(self, **kwargs)


class NeuRADField(nn.Module):
    def __init__(self, config: NeuRADFieldConfig, actors=None, static_scale: float = 1.0,
                 implementation: Literal["hip"] = "hip") -> None:
        super().__init__()
        if config.num_multisamples != 1:
            raise NotImplementedError("the HIP path implements NeuRadar's default num_multisamples=1")
        self.config = config
        self.implementation = implementation
        self.hashgrid: NeuRADHashEncoding = config.grid.setup(dynamic_actors=actors, static_scale=static_scale,
                                                              implementation=implementation)
        self.geo_feat_dim = config.nff_out_dim
        self.mlp_geo = MLP(in_dim=self.hashgrid.get_out_dim(), num_layers=config.geo_num_layers,
                           layer_width=config.geo_hidden_dim, out_dim=self.geo_feat_dim + 1, implementation=implementation)
        self.mlp_feature = MLP(in_dim=16 + self.geo_feat_dim, num_layers=config.nff_num_layers,
                               layer_width=config.nff_hidden_dim, out_dim=config.nff_out_dim, implementation=implementation)
        if config.use_sdf:
            self.sdf_to_density = SigmoidDensity(config.sdf_beta, learnable_beta=config.learnable_beta)
        else:  # neurad_field.py:149-150: density = trunc_exp(geo_out).  The fused kernels still want a beta; it takes
            # no part in the density outputs and is not a parameter of the module (the reference builds none either)
            self.register_buffer("_unused_beta", torch.ones(1), persistent=False)

    def get_param_groups(self, param_groups: Dict):
        self.hashgrid.get_param_groups(param_groups)
        param_groups["fields"] += list(self.mlp_geo.parameters()) + list(self.mlp_feature.parameters())
        if self.config.use_sdf:
            param_groups["fields"] += list(self.sdf_to_density.parameters())

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False, flip: Optional[Tensor] = None
                ) -> Dict[FieldHeadNames, Tensor]:
        B, S = ray_samples.shape
        buf, strides, dirs, rows_sm = self.hashgrid.encode_samples(ray_samples, directions=True, flip=flip,
                                                                   rows_sample_major=True)
        # per-ray directions, or per-sample ones (nr_field_t.sample_dirs) when dynamic actors rotated some of them
        feature, sdf, alpha = ops.field_mlp(buf, strides, self.hashgrid.static_grid.features_per_level,
                                            ray_samples.directions, S, B * S, self.mlp_geo.weights(), self.mlp_feature.weights(),
                                            self.sdf_to_density.beta if self.config.use_sdf else self._unused_beta,
                                            rows_sample_major=rows_sm, dtype=self.config.mlp_dtype,
                                            grad_scale=self.config.mlp_grad_scale, sample_dirs=dirs)
        if not self.config.use_sdf:
            # geo_out is the kernels' `sdf` output (row 0 of mlp_geo); density = trunc_exp(geo_out) on the proposal head's
            # kernel (trunc_exp of a 1-feature "grid" times the weight 1: nr_prop_density_fwd/bwd, activations.py:28-54)
            one = torch.ones(1, 1, device=sdf.device)
            density = ops.prop_density(sdf.view(-1, 1), (1, 1), 1, one, B * S)
            return {FieldHeadNames.FEATURE: feature.view(B, S, -1), FieldHeadNames.DENSITY: density.view(B, S, 1)}
        return {FieldHeadNames.FEATURE: feature.view(B, S, -1), FieldHeadNames.SDF: sdf.view(B, S, 1),
                FieldHeadNames.ALPHA: alpha.view(B, S, 1)}


@dataclass
class NeuRADProposalFieldConfig:  # neurad_field.py:155-182
    grid: NeuRADHashEncodingConfig = field(
        default_factory=lambda: NeuRADHashEncodingConfig(
            static=StaticSettings(log2_hashmap_size=20, num_levels=6, max_res=4096, base_res=128, hashgrid_dim=1),
            actor=ActorSettings(log2_hashmap_size=15, num_levels=4, base_res=64, max_res=1024, hashgrid_dim=1),
            require_actor_grad=False))
    hidden_dim: int = 16

    def setup(self, **kwargs) -> "NeuRADProposalField":
        return NeuRADProposalField(self, **kwargs)


class NeuRADProposalField(nn.Module):
    def __init__(self, config: NeuRADProposalFieldConfig, actors=None, static_scale: float = 1.0,
                 implementation: Literal["hip"] = "hip") -> None:
        super().__init__()
        self.config = config
        self.implementation = implementation
        self.hashgrid: NeuRADHashEncoding = config.grid.setup(dynamic_actors=actors, static_scale=static_scale,
                                                              implementation=implementation)
        self.density_decoder = nn.Linear(self.hashgrid.get_out_dim(), 1, bias=False)

    def get_param_groups(self, param_groups: Dict):
        self.hashgrid.get_param_groups(param_groups)
        param_groups["fields"] += list(self.density_decoder.parameters())

    def get_density(self, ray_samples: RaySamples, flip: Optional[Tensor] = None) -> Tuple[Tensor, None]:
        """flip [B] of +-1: the per-ray x-flip of samples inside actor boxes (drawn at random in training when None)."""
        B, S = ray_samples.shape
        buf, strides, _, rows_sm = self.hashgrid.encode_samples(ray_samples, rows_sample_major=True, flip=flip)
        density = ops.prop_density(buf, strides, self.hashgrid.static_grid.features_per_level,
                                   self.density_decoder.weight, B * S, n_samples=S, rows_sample_major=rows_sm)
        return density.view(B, S, 1), None

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None) -> dict:
        return {}
