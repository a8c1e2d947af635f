"""The hot loop as one module: `NeuRadarHotPath.get_nff_outputs` mirrors
`NeuRadarModel.get_nff_outputs` (reference models/neuradar.py:495-548 with _get_ray_samples :570-586),
and `TrainStep` is the step the benchmark times: forward + bench loss + backward + Adam, optionally
captured into one hipGraph (the path is launch-bound at 4k-16k rays).

What replaces the out-of-scope tail of the reference model (CNN / lidar / radar decoders, Hungarian
radar loss): direct supervision of the path's own outputs with the reference's loss multipliers,
plus the reference's two regularisers -- see DESIGN.md.
"""
import math
import os
import dataclasses
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import losses, ops
from .field_heads import FieldHeadNames
from .neurad_field import NeuRADField, NeuRADFieldConfig, NeuRADProposalField, NeuRADProposalFieldConfig
from .ray_samplers import PowerSampler, ProposalNetworkSampler
from .rays import RayBundle, RaySamples
from .renderers import composite, render_depth_simple

SKY_DISTANCE = 20000.0  # SamplingSettings.sky_distance, neuradar.py:137
EPS = 1e-7


@dataclass
class HotPathConfig:
    """The slice of NeuRadarModelConfig (neuradar.py:118-187) that parameterises the path."""

    proposal_field_1: NeuRADProposalFieldConfig = dataclasses.field(default_factory=NeuRADProposalFieldConfig)
    proposal_field_2: NeuRADProposalFieldConfig = dataclasses.field(default_factory=NeuRADProposalFieldConfig)
    field: NeuRADFieldConfig = dataclasses.field(default_factory=NeuRADFieldConfig)
    num_proposal_samples: Tuple[int, ...] = (128, 64)
    num_nerf_samples: int = 32
    power_lambda: float = -1.0
    power_scaling: float = 0.1
    single_jitter: bool = True
    static_scale: float = 100.0
    appearance_dim: int = 0  # 16 in the reference; the embedding lookup is a torch gather (neuradar.py:550-568)
    duration: float = 20.0
    num_sensors: int = 1
    temporal_appearance_freq: float = 1.0
    # loss multipliers (LossSettings, neuradar.py:80-115)
    rgb_mult: float = 5.0
    depth_mult: float = 0.01
    interlevel_loss_mult: float = 0.001
    distortion_loss_mult: float = 0.002
    # lidar decoder (K7) and its losses (neuradar.py:89-110)
    lidar_decoder: bool = False
    intensity_mult: float = 0.1
    ray_drop_loss_mult: float = 0.01
    non_return_lidar_distance: float = 150.0
    carving_mult: float = 0.01  # LossSettings.carving_mult, neuradar.py:92
    carving_epsilon: float = 0.1  # :94
    prop_lidar_loss_mult: float = 0.1  # :96
    non_return_loss_mult: float = 0.1
    quantile_threshold: float = 0.95
    # the modality decoders of neuradar.py:225-278 (RGB CNN, lidar MLP, radar transformer + heads) as attributes of the model,
    # under the reference's own names; with them the fused step can supervise through the decoders (FusedTrainStep.set_decoders)
    decoders: bool = False
    radar_mult: float = 0.02
    radar_loss_type: str = "nll"  # LossSettings.radar_loss_type (:114); "euclidean" = the deterministic head
    # rendering (get_outputs_for_camera_ray_bundle): method_configs.py:380, neurad.py:141,149
    eval_num_rays_per_chunk: int = 1 << 15
    rgb_upsample_factor: int = 3
    compensate_upsampling_when_rendering: bool = True
    # rendering: run the RGB CNN on the operand type the model trains with (torch.autocast) when the field's MLPs are 16-bit.
    # The reference renders its torch CNN in fp32 (autocast wraps train_iteration only, engine/trainer.py:564) -- 17 ms for the
    # 1080p image here, 6 ms in bf16; a CNN trained on 16-bit operands is evaluated consistently on them.  False: fp32.
    render_decoders_in_training_dtype: bool = True


def _wait_table_syncs_hook(module, prefix, keep_vars) -> None:
    from .parallel import wait_all_table_syncs

    wait_all_table_syncs()


class NeuRadarHotPath(nn.Module):
    def __init__(self, config: HotPathConfig, actors=None) -> None:
        """actors: a `DynamicActors` module shared by the main and the proposal fields (neuradar.py:196-208,293-300)."""
        super().__init__()
        self.config = config
        c = config
        self.dynamic_actors = actors
        self.field = c.field.setup(actors=actors, static_scale=c.static_scale, implementation="hip")
        self.proposal_fields = nn.ModuleList(
            [pc.setup(actors=actors, static_scale=c.static_scale, implementation="hip")
             for pc in (c.proposal_field_1, c.proposal_field_2)])
        # Reference quirk (neuradar.py:302): the density_fns list is built with a
        # late-binding lambda, so BOTH proposal rounds evaluate proposal_fields[1]; proposal_fields[0]
        # exists (state_dict, optimizer, all-reduce) but never runs.  Reproduced on purpose.
        last = self.proposal_fields[-1]
        # state_dict() of a model whose main table a sharded data-parallel step is still all-gathering: wait (stream side) first
        self.register_state_dict_pre_hook(_wait_table_syncs_hook)
        self._flips = (None, None, None)  # injected per-ray actor flips of (round 0, round 1, main field); None = draw
        self.density_fns = [lambda rs, i=i: last.get_density(rs, flip=self._flips[i])[0] for i in range(len(self.proposal_fields))]
        self.sampler = ProposalNetworkSampler(
            num_proposal_samples_per_ray=c.num_proposal_samples, num_nerf_samples_per_ray=c.num_nerf_samples,
            num_proposal_network_iterations=len(c.num_proposal_samples), single_jitter=c.single_jitter,
            initial_sampler=PowerSampler(lambda_=c.power_lambda, scaling=c.power_scaling), update_sched=lambda x: 0)
        if c.appearance_dim > 0:
            self._num_embeds_per_sensor = math.ceil(c.duration * c.temporal_appearance_freq)
            self.appearance_embedding = nn.Embedding(c.num_sensors * self._num_embeds_per_sensor, c.appearance_dim)
        if c.decoders:  # neuradar.py:225-278: registered under the reference's attribute names (state dicts interchange)
            from .decoders import Decoders

            dec = Decoders(n_features=c.field.nff_out_dim + c.appearance_dim)
            for name, child in dec.named_children():
                self.add_module(name, child)
            object.__setattr__(self, "_decoders", dec)  # the methods (decode_radar, forward) -- not a second registration
        elif c.lidar_decoder:  # neuradar.py:241-248: per-ray MLP on the rendered features (+ appearance embedding)
            from .mlp import MLP

            self.lidar_decoder = MLP(in_dim=c.field.nff_out_dim + c.appearance_dim, layer_width=32, out_dim=2, num_layers=3,
                                     implementation="hip", out_activation=None)

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        groups: Dict[str, List[nn.Parameter]] = {"hashgrids": [], "fields": []}
        for f in (self.field, *self.proposal_fields):
            f.get_param_groups(groups)
        if self.dynamic_actors is not None:
            self.dynamic_actors.get_param_groups(groups)  # "trajectory_opt" (dynamic_actors.py:203-205)
        if self.config.appearance_dim > 0:
            groups["fields"] += list(self.appearance_embedding.parameters())
        if self.config.lidar_decoder or self.config.decoders:
            groups["fields"] += list(self.lidar_decoder.parameters())  # neuradar.py:343
        if self.config.decoders:  # neuradar.py:344-353
            groups["cnn"] = list(self.rgb_decoder.parameters())
            groups["transformer"] = [p for m in (self.radar_decoder, self.offset_head, self.radar_angle_head,
                                                 self.radar_uncertainty_head, self.existence_probability_head)
                                     for p in m.parameters()]
        return groups

    def appearance_of(self, times: Tensor, sensor_idx: Tensor) -> Tensor:
        """_get_appearance_embedding (neuradar.py:550-568) from per-ray arrays: times [B], sensor_idx [B] int64 -> [B, A]."""
        from types import SimpleNamespace

        return self._get_appearance_embedding(SimpleNamespace(times=times.reshape(-1, 1), metadata={"sensor_idxs": sensor_idx.reshape(-1, 1)}))

    def decode_radar(self, radar_features: Tensor, depth: Tensor, directions_spher: Tensor, num_radar_scans: int, seed_epoch=None):
        """decode_features, radar branch (neuradar.py:463-491) -> radar_output [scans, n, 7]."""
        return self._decoders.decode_radar(radar_features, depth, directions_spher, num_radar_scans, seed_epoch=seed_epoch)

    def decode_features(self, features: Tensor, patch_size, depth: Tensor, directions_spher: Tensor, is_lidar=None, is_radar=None,
                        num_radar_scans=None):
        """neuradar.py:410-493: (rgb, intensity, ray_drop_logit, radar_output)."""
        return self._decoders(features, patch_size, depth, directions_spher, is_lidar, is_radar, num_radar_scans)

    def _get_ray_samples(self, bundle: RayBundle, t_rand=None, jitters=(None, None)):
        """neuradar.py:570-586."""
        if bundle.fars is not None:
            bundle.fars = bundle.fars.clamp_max(SKY_DISTANCE)
        else:
            bundle.fars = torch.full_like(bundle.pixel_area, SKY_DISTANCE)
        if bundle.nears is None:
            bundle.nears = torch.zeros_like(bundle.fars)
        rs, prop_w, prop_rs = self.sampler(bundle, self.density_fns, pass_ray_samples=True, t_rand=t_rand,
                                           jitters=jitters)
        # "sky field": the last sample is stretched to sky_distance (:578-582)
        dist_to_sky = SKY_DISTANCE - rs.euclid[:, -1]
        euclid = torch.cat([rs.euclid[:, :-1], (rs.euclid[:, -1] + dist_to_sky)[:, None]], dim=-1)
        spacing = torch.cat([rs.spacing[:, :-1], torch.full_like(rs.spacing[:, -1:], 1 - EPS)], dim=-1)
        rs = RaySamples(rs.origins, rs.directions, rs.pixel_area, spacing, euclid, rs.nears, rs.fars, rs.times,
                        rs.metadata, rs.camera_indices)
        return rs, prop_rs, prop_w

    def _compute_is_close_to_lidar(self, rs: RaySamples, carving_epsilon: float = 0.1,
                                   non_return_lidar_distance: float = 150.0) -> Tensor:
        """neuradar.py:971-994: per-sample mask [B,S,1], False on non-lidar rays.  Branch-free
        (no nonzero / host sync): lidar rays with a return are "close" within carving_epsilon of the
        measured range, non-returns everywhere below non_return_lidar_distance."""
        md = rs.metadata
        mid = (rs.euclid[:, :-1] + rs.euclid[:, 1:])[..., None] * 0.5
        close_to_hit = (md["directions_norm"] - mid).abs() < carving_epsilon
        did_return = md["did_return"] if "did_return" in md else torch.ones_like(close_to_hit)
        in_range = mid < non_return_lidar_distance
        return md["is_lidar"] & ((did_return & close_to_hit) | ((~did_return) & in_range))

    def _get_appearance_embedding(self, bundle: RayBundle) -> Tensor:
        """neuradar.py:550-568 (temporal appearance)."""
        E = self._num_embeds_per_sensor
        sensor_idx = bundle.metadata["sensor_idxs"]
        time_idx = bundle.times / self.config.duration * E
        before = time_idx.floor().clamp(0, E - 1)
        after = (before + 1).clamp(0, E - 1)
        ratio = time_idx - before
        before, after = before + sensor_idx * E, after + sensor_idx * E
        e0 = self.appearance_embedding(before.squeeze(-1).long())
        e1 = self.appearance_embedding(after.squeeze(-1).long())
        return e0 * (1 - ratio) + e1 * ratio

    def get_nff_outputs(self, bundle: RayBundle, t_rand=None, jitters=(None, None), flips=(None, None, None)) -> Dict[str, Tensor]:
        """neuradar.py:495-548.  `bundle.pixel_area` is expected to be already scaled
        (sensors.scale_pixel_area = _scale_pixel_area).  flips: per-ray actor x-flips (+-1 [B]) of proposal round 0,
        round 1 and the main field, for callers that inject the random draws (None: drawn inside, training only)."""
        self._flips = tuple(flips)
        rs, prop_rs, prop_w = self._get_ray_samples(bundle, t_rand, jitters)
        outputs = self.field(rs, flip=self._flips[2])
        self._flips = (None, None, None)
        if FieldHeadNames.ALPHA in outputs:
            alpha = outputs[FieldHeadNames.ALPHA]
        else:  # use_sdf=False (_render_weights, neuradar.py:1018-1022): nerfacc.render_weight_from_density is alpha
            # compositing of alpha_i = 1 - exp(-sigma_i * (t_end_i - t_start_i)) -- one elementwise op in front of the same kernel
            alpha = 1.0 - torch.exp(-outputs[FieldHeadNames.DENSITY] * rs.deltas)
        weights, accumulation, features, depth = composite(alpha, outputs[FieldHeadNames.FEATURE], rs)
        if self.config.appearance_dim > 0:
            features = torch.cat([features, self._get_appearance_embedding(bundle)], dim=-1)
        out = {"features": features, "depth": depth, "accumulation": accumulation}
        lidar = self.training and "is_lidar" in bundle.metadata
        if lidar:  # neuradar.py:584-585
            for s in (rs, *prop_rs):
                s.metadata = dict(s.metadata, is_close_to_lidar=self._compute_is_close_to_lidar(s))
        for i, (w, s) in enumerate(zip(prop_w, prop_rs)):
            out[f"prop_depth_{i}"] = render_depth_simple(w, s)
            if lidar:  # :529-531
                mask = (~s.metadata["is_close_to_lidar"]) & s.metadata["is_lidar"]
                out[f"prop_weights_loss_{i}"] = ((w * mask) ** 2).sum()
        if lidar:  # :537-541: weights of lidar samples away from the measured return (carving loss input)
            md = rs.metadata
            m = ((~md["is_close_to_lidar"][:, :-1]) & md["is_lidar"][:, :-1]).squeeze(-1)
            w_ = weights[:, :-1]  # [B, S - 1, 1]; = w_[m] ([k, 1]) of the reference
            out["non_nearby_weights"] = ops.rows_where(w_.reshape(-1, w_.shape[-1]) if w_.dim() == 3 else w_.reshape(-1), m.reshape(-1))
        # the sky sample is dropped from the lists the regularisers see (:515,534-535)
        out["weights_list"] = prop_w + [weights[:, :-1]]
        out["ray_samples_list"] = prop_rs + [rs.drop_last()]
        out["weights"] = weights
        out["field_outputs"] = outputs
        out["ray_samples"] = rs
        return out

    forward = get_nff_outputs

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, bundle: RayBundle, image_shape: Optional[Tuple[int, int]] = None,
                                          num_radar_scans: int = 1) -> Dict[str, Tensor]:
        """models/neuradar.py:905-969 (evaluation / rendering entry).  `bundle` holds the rays of ONE sensor reading, row
        major: a camera image or patch of `image_shape` = (H, W) rays, or -- image_shape None -- a lidar / radar scan marked
        by `is_lidar` / `is_radar` in its metadata.  Camera rays are shot at 1 / rgb_upsample_factor of the resolution
        (every `step`-th ray from `step // 2` on, both axes) because the RGB decoder upsamples; the field is evaluated in
        chunks of eval_num_rays_per_chunk rays (a radar scan in one piece: its decoder attends over the whole scan), the
        decoders once over the whole reading.  Outputs: features / depth / accumulation / prop_depth_i shaped [*size, -1],
        rgb [H, W, 3] (camera), intensity and ray_drop_logits / ray_drop_prob (every ray, "intensity_for_cam"),
        radar_output [scans, n, 7] (radar)."""
        assert not self.training, "rendering runs in eval mode (deterministic samplers, no carving masks)"
        from .parallel import wait_all_table_syncs

        wait_all_table_syncs()  # (a sharded data-parallel step may still be all-gathering the table it updated)
        n_all = len(bundle)
        take = None
        if image_shape is None:  # lidar or radar
            md = bundle.metadata
            is_lidar = torch.ones_like(bundle.pixel_area, dtype=torch.bool) if "is_lidar" in md else None
            is_radar = torch.ones_like(bundle.pixel_area, dtype=torch.bool) if "is_radar" in md else None
            if is_lidar is not None and is_radar is not None:
                raise ValueError("a reading is a lidar scan or a radar scan, not both")
            if is_radar is not None:
                is_lidar = torch.zeros_like(is_radar)
            elif is_lidar is not None:
                is_radar = torch.zeros_like(is_lidar)
            output_size, patch_size = (n_all,), (1, 1)
        else:
            H, W = image_shape
            assert H * W == n_all, "image_shape does not match the bundle"
            if self.config.compensate_upsampling_when_rendering:
                step = self.config.rgb_upsample_factor
                rows = torch.arange(step // 2, H, step, device=bundle.origins.device)
                cols = torch.arange(step // 2, W, step, device=bundle.origins.device)
                take = (rows[:, None] * W + cols[None, :]).reshape(-1)
                H, W = rows.numel(), cols.numel()
            output_size = patch_size = (H, W)
            is_lidar = is_radar = None
        pick = (lambda t: t if (t is None or take is None) else t[take])  # noqa: E731
        rays = RayBundle(pick(bundle.origins), pick(bundle.directions), pick(bundle.pixel_area), pick(bundle.camera_indices),
                         pick(bundle.nears), pick(bundle.fars), {k: pick(v) for k, v in bundle.metadata.items()}, pick(bundle.times))
        n = len(rays)
        radar = image_shape is None and "is_radar" in bundle.metadata
        chunk = n if radar else self.config.eval_num_rays_per_chunk
        keep = ("features", "depth", "accumulation", "prop_depth_0", "prop_depth_1")
        fused = self._fused_renderer(min(chunk, n)) if (rays.origins.is_cuda and rays.nears is None and n > 0) else None
        if fused is not None:
            # forward-only launch chain over preallocated buffers (fused_render.FusedRenderer: ten launches per chunk, straight
            # into the output arrays) instead of the modular modules' ~60 launches and their temporaries per chunk
            dev = rays.origins.device
            C = self.config.field.nff_out_dim
            flat = {"features": torch.empty(n, C, device=dev), **{k: torch.empty(n, 1, device=dev) for k in keep[1:]}}
            fused.refresh()  # this reading's MLP weight image (parameters change through raw pointers between renders)
            for lo in range(0, n, chunk):
                hi = min(lo + chunk, n)
                fused.render(rays.origins[lo:hi], rays.directions[lo:hi], rays.pixel_area[lo:hi],
                             None if rays.fars is None else rays.fars[lo:hi], flat, lo)
            if self.config.appearance_dim > 0:  # neuradar.py:510-512
                flat["features"] = torch.cat([flat["features"], self._get_appearance_embedding(rays)], dim=-1)
            outputs = {k: v.view(*output_size, -1) for k, v in flat.items()}
        else:
            lists: Dict[str, List[Tensor]] = {k: [] for k in keep}
            for lo in range(0, n, chunk):
                sl = slice(lo, min(lo + chunk, n))
                cut = (lambda t: None if t is None else t[sl])  # noqa: E731
                part = RayBundle(rays.origins[sl], rays.directions[sl], rays.pixel_area[sl], cut(rays.camera_indices), cut(rays.nears),
                                 cut(rays.fars), {k: v[sl] for k, v in rays.metadata.items()}, cut(rays.times))
                out = self.get_nff_outputs(part)
                for k in keep:
                    if k in out:
                        lists[k].append(out[k])
            outputs = {k: torch.cat(v).view(*output_size, -1) for k, v in lists.items() if v}
        if getattr(self, "_decoders", None) is None:
            return outputs
        features = outputs["features"].view(-1, outputs["features"].shape[-1])
        dec = self._decoders
        # intensity_for_cam (neuradar.py:446-447): the lidar decoder over every ray of the reading
        intensity, ray_drop_logit = dec.lidar_decoder(features).float().split(1, dim=-1)
        outputs["intensity"] = intensity.sigmoid().view(*output_size, -1)
        outputs["ray_drop_logits"] = ray_drop_logit.view(*output_size, -1)
        outputs["ray_drop_prob"] = outputs["ray_drop_logits"].sigmoid()
        if image_shape is not None:  # the CNN over the whole image as one patch
            patch = features.view(1, *patch_size, features.shape[-1]).permute(0, 3, 1, 2)
            if any(p.dim() == 4 and not p.is_contiguous(memory_format=torch.channels_last) for p in dec.rgb_decoder.parameters()):
                patch = patch.contiguous()
            low = {"bfloat16": torch.bfloat16, "float16": torch.float16}.get(self.field.config.mlp_dtype)
            if low is not None and self.config.render_decoders_in_training_dtype and patch.is_cuda:
                # (the BasicBlocks' 7 x 7 convolutions on the matrix-core kernel with their batch norms folded in: conv7.hip)
                dec.prepare_conv7_eval(low if os.environ.get("NR_CONV7", "1") != "0" else None)
                # the whole CNN on the hand-written kernels where the layouts fit (Decoders.render_rgb16); else torch.autocast
                rgb16 = (dec.render_rgb16(features.contiguous(), patch_size[0], patch_size[1], low)
                         if os.environ.get("NR_RENDER_PW", "1") != "0" else None)
                if rgb16 is not None:
                    outputs["rgb"] = rgb16
                    return outputs
                with torch.autocast("cuda", dtype=low):
                    rgb = dec.rgb_decoder(patch).float()
            else:
                dec.prepare_conv7_eval(None)
                rgb = dec.rgb_decoder(patch)
            outputs["rgb"] = rgb.permute(0, 2, 3, 1).squeeze(0)
        elif radar:
            outputs["radar_output"] = dec.decode_radar(features, outputs["depth"].reshape(-1, 1), rays.metadata["directions_spher"],
                                                        num_radar_scans)
        return outputs

    def _fused_renderer(self, max_rays: int):
        """The forward-only launch chain of the rendering entry (fused_render.FusedRenderer); None where it does not apply
        (dynamic actors, density branch, tcnn-layout tables, NR_FUSED_RENDER=0): the modular path.  ONE renderer is kept, sized to
        the largest request so far rounded up to a power of two (render() takes any n <= its size; about 20 KB of buffers per
        ray): a 16-ray lidar reading allocates a 16-ray renderer, a full image one chunk (eval_num_rays_per_chunk = 32 768 rays:
        650 MB) that then also serves the small readings, a radar scan longer than a chunk (rendered in one piece: its decoder
        attends over the whole scan) a larger one.  It stays on the model across training steps; release_render_buffers() drops
        it."""
        if os.environ.get("NR_FUSED_RENDER", "1") == "0":
            return None
        cache = self.__dict__.setdefault("_fused_render_cache", {})
        have = cache.get("renderer")
        if cache.get("unsupported"):
            return None
        if have is not None and have.B >= max_rays:
            return have
        from .fused_render import FusedRenderer

        size = max(16, 1 << (int(max_rays) - 1).bit_length())
        cache["renderer"] = None  # (the smaller one goes first: never two alive)
        del have
        try:
            cache["renderer"] = FusedRenderer(self, size)
        except NotImplementedError:
            cache["unsupported"] = True
        return cache["renderer"]

    def release_render_buffers(self) -> None:
        """Free the rendering entry's persistent device buffers (the FusedRenderer; rebuilt by the next rendered reading)."""
        self.__dict__.pop("_fused_render_cache", None)

    def decode_lidar(self, features: Tensor, is_lidar: Tensor):
        """decode_features, lidar branch (neuradar.py:432-452): the lidar rays' rendered features through the
        decoder MLP (MFMA kernels) -> (intensity in (0,1) [n_lidar,1], ray_drop_logit [n_lidar,1]); (None, None)
        when the batch holds no lidar ray."""
        lidar_features = ops.rows_where(features, is_lidar[..., 0])
        if lidar_features.numel() == 0:
            return None, None
        intensity, ray_drop_logit = self.lidar_decoder(lidar_features).split(1, dim=-1)
        return intensity.sigmoid(), ray_drop_logit

    def lidar_losses(self, pred_depth: Tensor, intensity: Tensor, ray_drop_logits: Tensor, termination_depth: Tensor,
                     did_return: Tensor, points_intensities: Tensor) -> Dict[str, Tensor]:
        """Training losses of the lidar rays (neuradar.py:612-636) with the multipliers of :690-700 applied:
        quantile-masked depth L1 (non-returning rays pulled beyond 150 m), intensity MSE, ray-drop BCE."""
        c = self.config
        target = termination_depth.clone()
        far = torch.tensor(c.non_return_lidar_distance, device=pred_depth.device)
        target[~did_return] = pred_depth.detach()[~did_return].maximum(far)
        unreduced = (target - pred_depth).abs()
        # (`unreduced[~did_return] *= mult` of the reference as a select: same values, no masked write in the autograd graph)
        unreduced = torch.where(did_return.reshape(-1, *([1] * (unreduced.dim() - 1))), unreduced, unreduced * c.non_return_loss_mult)
        mask = (unreduced < torch.quantile(unreduced, c.quantile_threshold)).squeeze(-1)
        qr = mask & did_return
        return {"depth_loss": c.depth_mult * ops.rows_where(unreduced, mask).mean(),
                "intensity_loss": c.intensity_mult * ((ops.rows_where(points_intensities, qr) - ops.rows_where(intensity, qr)) ** 2).mean(),
                "ray_drop_loss": c.ray_drop_loss_mult * torch.nn.functional.binary_cross_entropy_with_logits(
                    ray_drop_logits, (~did_return).unsqueeze(-1).to(ray_drop_logits))}

    def bench_loss(self, out: Dict[str, Tensor], target_features: Tensor, target_depth: Tensor) -> Tensor:
        """rgb_mult*MSE(features) + depth_mult*L1(depth) + inter-level + distortion (neuradar.py:672-704
        with the decoders replaced by direct supervision)."""
        c = self.config
        cs = [s.spacing for s in out["ray_samples_list"]]
        ws = [w[..., 0] for w in out["weights_list"]]
        loss = c.rgb_mult * torch.mean((out["features"][:, : target_features.shape[1]] - target_features) ** 2)
        loss = loss + c.depth_mult * (out["depth"] - target_depth).abs().mean()
        loss = loss + c.interlevel_loss_mult * losses.zipnerf_interlevel_loss(cs, ws)
        loss = loss + c.distortion_loss_mult * losses.distortion_loss(cs[-1], ws[-1])
        return loss


class GradScalerState:
    """torch.cuda.amp.GradScaler (engine/trainer.py:200,572-594; engine/optimizers.py:154-166) as device-resident state
    (nr_amp, include/neuradar_hip.h): the loss scale, its growth tracker and one found-inf flag per optimizer.  Nothing is
    read on the host, so a step that uses it is captured into a hipGraph like any other:
      * the 16-bit field backward takes its scale from the state and raises the flags of the optimizers it feeds when a
        gradient it writes is inf / NaN (FusedTrainStep.set_grad_scaler);
      * FlatAdam.check_buffer flags a small gradient buffer directly, the RGB CNN's 16-bit gradients are flagged while they
        are unscaled into the fp32 buffer (DecoderLossHead);
      * every Adam launch of a flagged optimizer leaves parameters and moments alone and clears the gradient
        (GradScaler.step); `update()` -- once per step, after the optimizers -- backs the scale off or grows it
        (GradScaler.update: x0.5 on inf / NaN, x2 after 2 000 clean steps) and the next step's schedule kernel does not
        count a skipped step (trainer.py:590-594).
    attach(optimizers): optimizer i becomes group i."""

    def __init__(self, device, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000) -> None:
        from . import _lib

        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.buf = torch.zeros(_lib.NR_AMP_FLOATS, device=device, dtype=torch.float32)
        ops.check(_lib.lib().nr_amp_init(ops._p(self.buf), float(init_scale), ops._stream()), "nr_amp_init")
        self.n_groups = 0
        self._F = _lib.NR_AMP_FOUND

    def attach(self, optimizers) -> "GradScalerState":
        from . import _lib

        assert len(optimizers) <= _lib.NR_AMP_MAX_GROUPS
        for g, o in enumerate(optimizers):
            o.amp, o.amp_group = self, g
        self.n_groups = len(optimizers)
        self.optimizers = list(optimizers)
        return self

    def group_of(self, param: nn.Parameter) -> Optional[int]:
        """The group (attached optimizer) that steps `param`; None if no attached optimizer manages it."""
        for g, o in enumerate(getattr(self, "optimizers", [])):
            try:
                o.buffer_of(param)
                return g
            except KeyError:
                continue
        return None

    def found(self, group: int) -> Tensor:
        """The found-inf flag of optimizer `group` (a one-element view of the state)."""
        return self.buf[self._F + group:self._F + group + 1]

    @property
    def scale(self) -> Tensor:
        return self.buf[0:1]

    @property
    def inv_scale(self) -> Tensor:
        return self.buf[2:3]

    def update(self) -> None:
        from . import _lib

        ops.check(_lib.lib().nr_amp_update(ops._p(self.buf), max(self.n_groups, 1), self.growth_factor, self.backoff_factor,
                                           self.growth_interval, ops._stream()), "nr_amp_update")

    def get_scale(self) -> float:
        """Host read (diagnostics / checkpoints only -- never inside a step)."""
        return float(self.buf[0])

    def skipped_steps(self) -> int:
        return int(self.buf[4])

    def state_dict(self) -> Dict[str, object]:
        return {"amp": self.buf.clone(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval}

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        self.buf.copy_(sd["amp"])
        self.growth_factor, self.backoff_factor, self.growth_interval = sd["growth_factor"], sd["backoff_factor"], sd["growth_interval"]


class FlatAdam:
    """Adam/AdamW with the fused HIP kernel.  Small parameters are re-homed into ONE flat buffer
    (views keep names/shapes), so a step is one launch for all of them plus one per hash table.

    Hyper-parameters of configs/method_configs.py:384-409; the (lr, bias-correction) triple lives in a
    device tensor updated by a few scalar ops so that a captured graph can be replayed.  Parameters in
    `skip` never receive a gradient (proposal_fields[0], see NeuRadarHotPath) and are left untouched,
    like torch.optim skips parameters whose .grad is None."""

    BIG = 1 << 16

    def __init__(self, params: List[nn.Parameter], lr: float, eps: float = 1e-15, weight_decay: float = 0.0,
                 adamw: bool = False, betas=(0.9, 0.999), lr_final: Optional[float] = None, max_steps: int = 20001,
                 warmup_steps: int = 500, skip: Optional[List[nn.Parameter]] = None, flatten: bool = True) -> None:
        skip_ids = {id(p) for p in (skip or [])}
        self.params = [p for p in params if p.requires_grad and id(p) not in skip_ids]
        self.lr, self.eps, self.wd, self.adamw, self.betas = lr, eps, weight_decay, adamw, betas
        self.lr_final, self.max_steps, self.warmup = lr_final, max_steps, warmup_steps
        dev = self.params[0].device
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        small = [p for p in self.params if p.numel() <= self.BIG] if flatten else []
        big = [p for p in self.params if p.numel() > self.BIG or not flatten]
        self.buffers = self._coalesce(big)  # (param, grad) flat views stepped by one launch each
        if small:
            from .fused_step import flatten_parameters

            flat = flatten_parameters(small)
            self.buffers.append((flat["param"], flat["grad"]))
        # Adam's moments.  A hash table's live 16-byte groups are scattered (15 % of them in a fresh step, a cache line each in
        # every array touched), so its two moments live in ONE array of [exp_avg x 4 | exp_avg_sq x 4] records: a live group costs
        # three lines (parameter, gradient, moments) instead of four.  `state` exposes the halves as strided views.
        self.state = [self._new_state(b.numel(), b.device, b.numel() > self.BIG and b.numel() % 4 == 0
                                      and os.environ.get("NR_ADAM_INTERLEAVED", "0") == "1") for b, _ in self.buffers]
        # one byte per four parameters of the hash tables: "has had a gradient" (zero together with the moments), lets
        # the kernel leave never-touched rows alone after reading 4 instead of 12 bytes per parameter (no weight decay)
        self.seen = [torch.zeros(b.numel() // 4, device=dev, dtype=torch.uint8) if (b.numel() > self.BIG and weight_decay == 0.0)
                     else None for b, _ in self.buffers]
        self.step_t = torch.zeros(2, device=dev, dtype=torch.float32)  # [scheduler steps, optimizer updates]: nr_adam_hyper
        self.hyper = torch.zeros(3, device=dev, dtype=torch.float32)
        self.amp, self.amp_group = None, 0  # GradScalerState.attach

    @staticmethod
    def _new_state(n: int, device, interleaved: bool):
        if not interleaved:
            return torch.zeros(n, device=device), torch.zeros(n, device=device)
        mv = torch.zeros(n // 4, 2, 4, device=device)
        return mv[:, 0, :], mv[:, 1, :]

    @staticmethod
    def _coalesce(big: List[nn.Parameter]):
        """One (param, grad) buffer per RUN of parameters that sit back to back in one storage with their gradients back to
        back in another -- the per-actor hash tables, views of one [A, L*T, F] buffer per field (neurad_encoding.py): a step
        is then one launch per field instead of one per actor (12 vehicles x 3 fields: 36 launches of 5 us at the end of
        every step)."""
        out, run = [], []

        def flush():
            if not run:
                return
            if len(run) == 1:
                out.append((run[0].data, run[0].grad))
            else:
                n = sum(q.numel() for q in run)
                pd, gd = run[0].data, run[0].grad
                out.append((torch.empty(0, device=pd.device, dtype=pd.dtype).set_(pd.untyped_storage(), pd.storage_offset(), (n,)),
                            torch.empty(0, device=gd.device, dtype=gd.dtype).set_(gd.untyped_storage(), gd.storage_offset(), (n,))))
            run.clear()

        for p in big:
            last = run[-1] if run else None
            if last is not None and not (
                    p.is_contiguous() and p.grad.is_contiguous() and last.is_contiguous() and last.grad.is_contiguous()
                    and p.dtype == last.dtype and p.data_ptr() == last.data_ptr() + last.numel() * last.element_size()
                    and p.grad.data_ptr() == last.grad.data_ptr() + last.numel() * last.element_size()
                    and p.untyped_storage().data_ptr() == last.untyped_storage().data_ptr()
                    and p.grad.untyped_storage().data_ptr() == last.grad.untyped_storage().data_ptr()):
                flush()
            run.append(p)
        flush()
        return out

    def grad_buffers(self) -> List[Tensor]:
        return [g for _, g in self.buffers]

    def shard_buffer(self, i: int, rank: int, world: int, force: bool = False) -> Optional[Tuple[int, int]]:
        """Data-parallel optimizer-state sharding of buffer i (a hash table): this rank keeps Adam's moments for -- and
        steps -- only elements [lo, hi) = its 1/world of the buffer (parallel.GradAllReducer.shard_step reduce-scatters the
        gradient onto the owners and all-gathers the updated rows).  Moments and `seen` flags shrink to the shard (2 x 537 MB
        -> 2 x 67 MB per GPU for NeuRadar's main table on 8 GPUs).  Returns (lo, hi), or None when the buffer does not
        divide into world pieces of whole 16-byte groups (it then stays replicated)."""
        p, g = self.buffers[i]
        n = p.numel()
        if (world <= 1 and not force) or n % (world * 4) != 0:  # (force: a one-rank group that runs the collectives anyway -- tests)
            return None
        self.buffers[i] = (p.view(-1), g.view(-1))  # (a table's buffer is its [L*T, F] parameter: shards are element ranges)
        per = n // world
        lo, hi = rank * per, (rank + 1) * per
        if not hasattr(self, "shards"):
            self.shards = {}
        self.shards[i] = (lo, hi)
        m, v = self.state[i]
        ms, vs = self._new_state(hi - lo, m.device, not m.is_contiguous())
        ms.reshape(-1).copy_(m.reshape(-1)[lo:hi]) if ms.is_contiguous() else ms.copy_(m[lo // 4:hi // 4])
        vs.reshape(-1).copy_(v.reshape(-1)[lo:hi]) if vs.is_contiguous() else vs.copy_(v[lo // 4:hi // 4])
        self.state[i] = (ms, vs)
        if self.seen[i] is not None:
            self.seen[i] = self.seen[i][lo // 4:hi // 4].clone()
        return lo, hi

    def _schedule(self, step: Tensor) -> Tensor:
        """ExponentialDecayScheduler (engine/schedulers.py:112-143): cosine-ramp warm-up from
        lr_pre_warmup = 1e-8, then log-linear decay to lr_final.  `step` is the 0-based scheduler step."""
        lr_final = self.lr if self.lr_final is None else self.lr_final
        t = torch.clamp((step - self.warmup) / max(self.max_steps - self.warmup, 1), 0, 1)
        decayed = torch.exp(math.log(self.lr) * (1 - t) + math.log(lr_final) * t)
        if self.warmup > 0:
            pre = 1e-8
            ramp = pre + (self.lr - pre) * torch.sin(0.5 * math.pi * torch.clamp(step / self.warmup, 0, 1))
            return torch.where(step < self.warmup, ramp, decayed)
        return decayed

    @torch.no_grad()
    def advance(self) -> None:
        """One tiny kernel: lr(step) [LambdaLR: step k uses func(k-1)], bias corrections, step += 1."""
        ops.check(ops._lib.lib().nr_adam_hyper(ops._p(self.step_t), ops._p(self.hyper), self.lr,
                                               self.lr if self.lr_final is None else self.lr_final, self.warmup,
                                               self.max_steps, self.betas[0], self.betas[1],
                                               ops._p(self.amp.buf) if self.amp is not None else None, self.amp_group,
                                               ops._stream()), "nr_adam_hyper")

    @torch.no_grad()
    def step_buffer(self, i: int, grad_scale: float = 1.0, delta16: Optional[Tensor] = None, skip_extra: Optional[Tensor] = None,
                    part: Optional[Tuple[int, int]] = None) -> None:
        """Adam on buffer i (after `advance()`), on the current stream.  grad_scale = 1/world turns the
        SUM all-reduce of the data-parallel ranks into DDP's mean without a separate pass.  delta16 (bf16, one per element of
        this rank's shard): the update also leaves as a rounded delta for the other replicas (GradAllReducer.shard_step).
        skip_extra: a second skip flag (device float; GradAllReducer.reduce_sparse's overflow flag, value 2 = skip and keep the
        gradient): the loss scaler's found-inf flag, when one is attached and raised, takes precedence."""
        (p, g), (m, v) = self.buffers[i], self.state[i]
        lo_hi = getattr(self, "shards", {}).get(i)
        if lo_hi is not None:  # this rank's rows only (the rest of the gradient buffer is cleared by the exchange)
            p, g = p[lo_hi[0]:lo_hi[1]], g[lo_hi[0]:lo_hi[1]]
        # marked[i]: this step's scatter set the buffer's `seen` bytes itself (FusedTrainStep, single GPU): groups that never had
        # a gradient are skipped on their byte alone
        marked = bool(getattr(self, "marked", {}).get(i)) and lo_hi is None and self.seen[i] is not None
        skip = self.amp.found(self.amp_group) if self.amp is not None else None
        if skip_extra is not None:
            skip = skip_extra if skip is None else torch.where(skip != 0, skip, skip_extra)
        seen = self.seen[i]
        if part is not None:
            # elements [lo, hi) of the buffer only (multiples of 4; one LEVEL of a hash table: FusedTrainStep steps a level as
            # soon as its scatter is done, beside the next level's scatter) -- element-wise the same update as the whole launch
            lo, hi = part
            assert lo_hi is None and delta16 is None and lo % 4 == 0 and hi % 4 == 0 and m.is_contiguous() and v.is_contiguous()
            p, g, m, v = p.view(-1)[lo:hi], g.view(-1)[lo:hi], m.view(-1)[lo:hi], v.view(-1)[lo:hi]
            seen = None if seen is None else seen[lo // 4:hi // 4]
        ops.adam_step(p, g, m, v, self.lr, 1, self.betas, self.eps, self.wd, self.adamw, grad_scale=grad_scale,
                      zero_grad=True, dev_hyper=self.hyper, seen_grad=seen, marked=marked, skip=skip, delta16=delta16)

    @torch.no_grad()
    def step_buffer_split(self, i: int, phase: int, stamp: Tensor) -> None:
        """The marked update of buffer i (a hash table whose `seen` bytes this step's scatter sets) in two launches around the
        scatter -- ops.adam_step_split: phase 1 (groups with a history that `stamp` says this step does not touch: zero gradient)
        any time after `advance()`, e.g. beside the forward; phase 2 (the stamped groups) after the scatter.  No loss scaler, no
        shards, no weight decay: FusedTrainStep falls back to step_buffer otherwise."""
        (p, g), (m, v) = self.buffers[i], self.state[i]
        assert self.amp is None and i not in getattr(self, "shards", {}) and self.seen[i] is not None and self.wd == 0.0
        ops.adam_step_split(p.view(-1), g.view(-1), m, v, self.betas, self.eps, 1.0, self.hyper, self.seen[i], stamp, self.step_t[1:], phase)

    @torch.no_grad()
    def check_buffer(self, i: int) -> None:
        """With a loss scaler attached: raise this optimizer's found-inf flag if gradient buffer i holds an inf / NaN (one
        small launch; meant for the flat buffer of the small parameters -- the tables' gradients are flagged by their
        producer, the 16-bit field backward)."""
        if self.amp is None:
            return
        g = self.buffers[i][1]
        ops.check(ops._lib.lib().nr_nonfinite_check(ops._p(g), g.numel(), ops._p(self.amp.found(self.amp_group)), ops._stream()),
                  "nr_nonfinite_check")

    def state_dict(self) -> Dict[str, object]:
        """Moments, the device-side step counters / schedule triple (what the reference's trainer checkpoints as optimizer +
        scheduler state, engine/trainer.py:514-548).  A buffer sharded by `shard_buffer` holds the moments of this rank's
        elements [lo, hi) only: `shards` records (lo, hi, numel) per such buffer, so a checkpoint written by one rank is not
        mistaken for the whole table -- `gather_state_dict()` assembles the full moments on every rank for a checkpoint that
        any world size can load."""
        from .parallel import wait_all_table_syncs

        wait_all_table_syncs()
        sh = getattr(self, "shards", {})
        return {"exp_avg": [m.reshape(-1).clone() for m, _ in self.state], "exp_avg_sq": [v.reshape(-1).clone() for _, v in self.state],
                "step_t": self.step_t.clone(), "hyper": self.hyper.clone(),
                "shards": {i: (lo, hi, self.buffers[i][0].numel()) for i, (lo, hi) in sh.items()}}

    def gather_state_dict(self, group=None) -> Dict[str, object]:
        """state_dict() with every sharded buffer's moments all-gathered into the full table (collective: every rank calls it;
        equal shard sizes by construction of shard_buffer).  The result carries no `shards` and loads into any world size."""
        import torch.distributed as dist

        sd = self.state_dict()
        for i, (lo, hi, n) in sd["shards"].items():
            for key in ("exp_avg", "exp_avg_sq"):
                mine = sd[key][i].contiguous()
                world = n // (hi - lo)
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine, group=group)
                sd[key][i] = torch.cat(parts)
        sd["shards"] = {}
        return sd

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        mine = getattr(self, "shards", {})
        theirs = {int(k): tuple(v) for k, v in (sd.get("shards") or {}).items()}
        for i, ((m, v), m_, v_) in enumerate(zip(self.state, sd["exp_avg"], sd["exp_avg_sq"])):
            n = self.buffers[i][0].numel()
            if i in theirs:  # the checkpoint holds ONE rank's shard of this buffer
                lo, hi, n_ = theirs[i]
                if i not in mine or mine[i] != (lo, hi) or n_ != n:
                    raise RuntimeError(
                        f"FlatAdam.load_state_dict: buffer {i} of the checkpoint holds the moments of elements [{lo}, {hi}) of {n_} "
                        f"only (a per-rank shard), this optimizer " + (f"owns [{mine[i][0]}, {mine[i][1]})" if i in mine else "is not sharded")
                        + "; save with gather_state_dict() or load each rank's own file")
                m_, v_ = m_.reshape(-1), v_.reshape(-1)
            elif i in mine:  # a full-table checkpoint into a sharded optimizer: this rank's slice
                if m_.numel() != n:
                    raise RuntimeError(f"FlatAdam.load_state_dict: buffer {i} has {n} elements, the checkpoint {m_.numel()}")
                lo, hi = mine[i]
                m_, v_ = m_.reshape(-1)[lo:hi], v_.reshape(-1)[lo:hi]
            if m_.numel() != m.numel():
                raise RuntimeError(f"FlatAdam.load_state_dict: buffer {i}: {m_.numel()} checkpointed moments for {m.numel()} elements")
            m.copy_(m_.reshape(m.shape))
            v.copy_(v_.reshape(v.shape))
        st = sd["step_t"].reshape(-1)
        self.step_t.copy_(st if st.numel() == 2 else st[:1].expand(2))  # (checkpoints of ABI < 20 hold one counter)
        self.hyper.copy_(sd["hyper"])
        for s, (m, v) in zip(self.seen, self.state):  # "has had a gradient" = any moment non-zero, per group of four
            if s is not None:
                nz = ((m != 0) | (v != 0)).reshape(-1)[: s.numel() * 4].view(-1, 4).any(dim=1)
                s.copy_(nz.to(torch.uint8))

    def check_views(self) -> None:
        """The flat buffers are only the parameters' storage while nobody re-homes them (zero_grad(set_to_none=True),
        module.to(), load_state_dict(assign=True) do): raise instead of silently stepping detached buffers."""
        for p in self.params:
            i = self.buffer_of(p)
            g = self.buffers[i][1]
            if p.grad is None or not (g.data_ptr() <= p.grad.data_ptr() < g.data_ptr() + g.numel() * 4):
                raise RuntimeError("FlatAdam: a parameter's .grad no longer lives in the optimizer's flat buffer "
                                   "(re-homed by zero_grad(set_to_none=True) / .to() / assign=True?)")

    def buffer_of(self, param: nn.Parameter) -> int:
        """Index of the buffer that holds `param` (its own for tables, the flat one for small parameters)."""
        ptr = param.data_ptr()
        for i, (p, _) in enumerate(self.buffers):
            if p.data_ptr() <= ptr < p.data_ptr() + p.numel() * 4:
                return i
        raise KeyError("parameter is not managed by this optimizer")

    @torch.no_grad()
    def step(self) -> None:
        self.advance()
        if hasattr(self, "marked"):  # whoever calls step() wrote the gradients without marking `seen` (modular backward, a
            self.marked.clear()      # collective): only FusedTrainStep pairs its marking scatter with step_buffer()
        for i in range(len(self.buffers)):
            self.step_buffer(i)
