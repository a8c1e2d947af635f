"""Samplers of the path (reference: model_components/ray_samplers.py:55-132,255-376,569-666,838-852).

`nn.Module`s with the reference's names and call signatures; `self.training` selects stratified
jitter exactly like the reference.  Jitter tensors can be injected (`t_rand=`, `jitter=`) so parity
tests do not depend on RNG streams; otherwise they are drawn with torch.rand on the device.
"""
from typing import Callable, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import ops
from .rays import RayBundle, RaySamples


def _samples(bundle: RayBundle, spacing: Tensor, euclid: Tensor) -> RaySamples:
    # cameras/rays.py:336,353: the bundle's metadata "mimics the shape of the rays" -- [B,C] becomes a broadcast [B,S,C]
    # view (no copy), which is what the reference's caller indexes (models/neuradar.py:978-993)
    S = euclid.shape[1] - 1
    meta = {k: v[:, None, :].expand(v.shape[0], S, v.shape[-1]) for k, v in bundle.metadata.items()}
    return RaySamples(bundle.origins, bundle.directions, bundle.pixel_area, spacing, euclid, bundle.nears,
                      bundle.fars, bundle.times, meta, bundle.camera_indices)


class PowerSampler(nn.Module):
    """ZipNeRF power-law initial sampler (ray_samplers.py:838-852 over SpacedSampler :55-132)."""

    def __init__(self, num_samples: Optional[int] = None, lambda_: float = -1.5, scaling: float = 2.0,
                 train_stratified: bool = True, single_jitter: bool = False) -> None:
        super().__init__()
        self.num_samples, self.lambda_, self.scaling = num_samples, lambda_, scaling
        self.train_stratified, self.single_jitter = train_stratified, single_jitter

    def forward(self, ray_bundle: RayBundle, num_samples: Optional[int] = None, t_rand: Optional[Tensor] = None
                ) -> RaySamples:
        n = num_samples or self.num_samples
        B = len(ray_bundle)
        if t_rand is None and self.train_stratified and self.training:
            shape = (B, 1) if self.single_jitter else (B, n + 1)
            t_rand = torch.rand(shape, device=ray_bundle.origins.device).expand(B, n + 1)
        sp, eu = ops.power_bins(ray_bundle.nears, ray_bundle.fars, n, t_rand, self.lambda_, self.scaling)
        return _samples(ray_bundle, sp, eu)

    generate_ray_samples = forward


class PDFSampler(nn.Module):
    """Inverse-CDF resampling (ray_samplers.py:255-376), include_original=False."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False,
                 include_original: bool = False, histogram_padding: float = 0.01, lambda_: float = -1.0,
                 scaling: float = 0.1) -> None:
        super().__init__()
        if include_original or not single_jitter or histogram_padding != 0.01:
            raise NotImplementedError("HIP PDFSampler implements NeuRadar's configuration: include_original=False, "
                                      "single_jitter=True, histogram_padding=0.01 (ray_samplers.py:606)")
        self.num_samples, self.train_stratified = num_samples, train_stratified
        self.lambda_, self.scaling = lambda_, scaling

    def forward(self, ray_bundle: RayBundle, ray_samples: RaySamples, weights: Tensor,
                num_samples: Optional[int] = None, jitter: Optional[Tensor] = None) -> RaySamples:
        n = num_samples or self.num_samples
        if jitter is None and self.train_stratified and self.training:
            jitter = torch.rand((len(ray_bundle), 1), device=weights.device)
        sp, eu = ops.pdf_resample(weights[..., 0], ray_samples.spacing, ray_bundle.nears, ray_bundle.fars, n, jitter,
                                  self.lambda_, self.scaling)
        return _samples(ray_bundle, sp, eu)

    generate_ray_samples = forward


class ProposalNetworkSampler(nn.Module):
    """ray_samplers.py:569-666: power bins -> (density -> weights -> resample) x rounds."""

    def __init__(self, num_proposal_samples_per_ray: Tuple[int, ...] = (64,), num_nerf_samples_per_ray: int = 32,
                 num_proposal_network_iterations: int = 2, single_jitter: bool = False,
                 update_sched: Callable = lambda x: 1, initial_sampler: Optional[nn.Module] = None,
                 pdf_sampler: Optional[PDFSampler] = None) -> None:
        super().__init__()
        self.num_proposal_samples_per_ray = num_proposal_samples_per_ray
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        if initial_sampler is None:
            raise NotImplementedError("NeuRadar always passes a PowerSampler (models/neuradar.py:286-289)")
        self.initial_sampler = initial_sampler
        self.pdf_sampler = pdf_sampler or PDFSampler(include_original=False, single_jitter=single_jitter,
                                                     lambda_=initial_sampler.lambda_, scaling=initial_sampler.scaling)
        self._anneal, self._steps_since_update, self._step = 1.0, 0, 0

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step) -> None:
        self._step = step
        self._steps_since_update += 1

    def forward(self, ray_bundle: RayBundle, density_fns: List[Callable], pass_ray_samples: bool = True,
                t_rand: Optional[Tensor] = None, jitters: Tuple[Optional[Tensor], ...] = (None, None)
                ) -> Tuple[RaySamples, List[Tensor], List[RaySamples]]:
        assert pass_ray_samples, "the NeuRadar path always passes RaySamples to the density fns (neuradar.py:577)"
        weights_list, samples_list = [], []
        n = self.num_proposal_network_iterations
        weights, ray_samples = None, None
        updated = self._steps_since_update > self.update_sched(self._step) or self._step < 10
        for i_level in range(n + 1):
            is_prop = i_level < n
            num = self.num_proposal_samples_per_ray[i_level] if is_prop else self.num_nerf_samples_per_ray
            if i_level == 0:
                ray_samples = self.initial_sampler(ray_bundle, num_samples=num, t_rand=t_rand)
            else:
                annealed = weights if self._anneal == 1.0 else torch.pow(weights, self._anneal)
                ray_samples = self.pdf_sampler(ray_bundle, ray_samples, annealed, num_samples=num,
                                               jitter=jitters[i_level - 1])
            if is_prop:
                if updated:
                    density = density_fns[i_level](ray_samples)
                else:
                    with torch.no_grad():
                        density = density_fns[i_level](ray_samples)
                weights = ray_samples.get_weights(density)
                weights_list.append(weights)
                samples_list.append(ray_samples)
        if updated:
            self._steps_since_update = 0
        return ray_samples, weights_list, samples_list

    generate_ray_samples = forward
