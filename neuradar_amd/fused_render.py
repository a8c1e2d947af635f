"""Forward-only launch chain of the hot path for rendering / evaluation: what `NeuRadarModel.get_nff_outputs` does in eval mode
(models/neuradar.py:495-548 with _get_ray_samples :570-586; deterministic samplers: PowerSampler without jitter,
ray_samplers.py:98-132, PDFSampler with centred u, :332-334), as ten library launches per chunk over buffers allocated once --
no autograd graph, no per-op allocations, the fused per-ray launches of the training step (nr_power_bins_contract,
nr_proposal_round, nr_prop_field_fwd, nr_field_fwd_gather) instead of the modular path's ~60 launches.  The rendering entry
(`NeuRadarHotPath.get_outputs_for_camera_ray_bundle`, models/neuradar.py:905-969) walks its chunks through it.

Same numbers as the modular eval path (tests/test_gpu_render_entry.py compares both with the CPU oracle's eval pipeline).
"""
from ctypes import byref
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib, ops
from ._lib import NrField, check
from .step import SKY_DISTANCE, NeuRadarHotPath


class FusedRenderer:
    def __init__(self, model: NeuRadarHotPath, max_rays: int) -> None:
        c = model.config
        if not c.field.use_sdf or model.dynamic_actors is not None or len(c.num_proposal_samples) != 2:
            raise NotImplementedError("FusedRenderer: static scene, sigmoid-SDF alphas, two proposal rounds (the modular path covers the rest)")
        if model.field.hashgrid.config.layout != "torch" or model.proposal_fields[-1].hashgrid.config.layout != "torch":
            raise NotImplementedError("FusedRenderer runs on the torch table layout")
        self.model, self.cfg, self.B = model, c, int(max_rays)
        self.lib = _lib.lib()
        dev = next(model.parameters()).device
        self.dev = dev
        self.prop = model.proposal_fields[-1]  # both rounds evaluate proposal_fields[1] (neuradar.py:302 quirk)
        self.pgrid, self.mgrid = self.prop.hashgrid.static_grid, model.field.hashgrid.static_grid
        self.S = (*c.num_proposal_samples, c.num_nerf_samples)
        B = self.B
        f32 = dict(device=dev, dtype=torch.float32)
        self.nears, self.fars = torch.zeros(B, **f32), torch.empty(B, **f32)
        self.sp = [torch.empty(B, S + 1, **f32) for S in self.S]
        self.eu = [torch.empty(B, S + 1, **f32) for S in self.S]
        self.x01 = [torch.empty(B * S, 3, **f32) for S in self.S]
        self.std = [torch.empty(B * S, **f32) for S in self.S]
        self.feats = [torch.empty((self.pgrid if l < 2 else self.mgrid).num_levels, B * S, (self.pgrid if l < 2 else self.mgrid).features_per_level, **f32)
                      for l, S in enumerate(self.S)]
        self.dens = [torch.empty(B, S, **f32) for S in self.S[:2]]
        self.w = [torch.empty(B, S, **f32) for S in self.S]
        self.C = c.field.nff_out_dim
        Sm = self.S[2]
        self.feature, self.sdf, self.alpha = torch.empty(B * Sm, self.C, **f32), torch.empty(B * Sm, **f32), torch.empty(B * Sm, **f32)
        self.field_struct = self.field_image = None

    def refresh(self) -> NrField:
        """The field's parameter struct and packed weight image, REBUILT (one small launch).  Call it once per rendered reading,
        before the chunks' render() calls: the optimizers of this package write parameters through raw pointers (nr_adam_step,
        nr_apply_delta16, graph replays), which torch's `_version` counters never see -- a cache keyed on them served a stale
        MLP image beside current hash tables after render -> train -> render."""
        fld = self.model.field
        gw, gb = fld.mlp_geo.weights()
        fw, fb = fld.mlp_feature.weights()
        fs = NrField()
        fs.geo, fs.feat = ops._mlp_struct(gw, gb), ops._mlp_struct(fw, fb)
        fs.beta = fld.sdf_to_density.beta.data_ptr()
        fs.dtype = _lib.NR_DTYPES[fld.config.mlp_dtype]
        fs.grad_scale = 1.0
        n_img = self.lib.nr_field_image_floats(byref(fs))
        if getattr(self, "field_image", None) is None or self.field_image.numel() != n_img:
            self.field_image = torch.empty(n_img, device=self.dev)
        fs.packed = self.field_image.data_ptr()
        check(self.lib.nr_field_pack(byref(fs), ops._p(self.field_image), ops._stream()), "field_pack")
        self.field_struct = fs
        return fs

    def _field(self) -> NrField:
        return self.field_struct if getattr(self, "field_struct", None) is not None else self.refresh()

    @torch.no_grad()
    def render(self, origins: Tensor, directions: Tensor, pixel_area: Tensor, fars: Optional[Tensor], out: Dict[str, Tensor], lo: int) -> None:
        """n <= max_rays rays -> out[...][lo:lo+n]: features [*, C], depth, accumulation, prop_depth_0 / _1 [*, 1].  pixel_area
        already scaled (sensors.scale_pixel_area); fars None = the camera's 1e6 (clamped to the sky distance)."""
        lib, p, c = self.lib, ops._p, self.cfg
        n = origins.shape[0]
        assert 0 < n <= self.B
        o, d, area = ops._f32(origins, "origins"), ops._f32(directions, "directions"), ops._f32(pixel_area.reshape(-1), "pixel_area")
        st = ops._stream()
        if fars is None:
            self.fars[:n].fill_(SKY_DISTANCE)
        else:
            torch.clamp(fars.reshape(-1), max=SKY_DISTANCE, out=self.fars[:n])
        fs = self._field()
        scale = self.model.field.hashgrid.static_scale
        lam, scal = c.power_lambda, c.power_scaling
        sm = n  # rows sample-major: a wave's 64 rows are 64 neighbouring rays (an image row) at one sample slot
        check(lib.nr_power_bins_contract(p(self.nears), p(self.fars), None, p(o), p(d), p(area), n, self.S[0], lam, scal, scale, sm,
                                         p(self.sp[0]), p(self.eu[0]), p(self.x01[0]), p(self.std[0]), st), "power_bins")
        pg, w_dec = self.pgrid, self.prop.density_decoder.weight
        prop_depth = [out["prop_depth_0"], out["prop_depth_1"]]
        for lvl in range(2):
            S, ns = self.S[lvl], n * self.S[lvl]
            check(lib.nr_prop_field_fwd(p(self.x01[lvl]), p(self.std[lvl]), p(pg.hash_table), p(pg.scalings), pg.num_levels,
                                        pg.features_per_level, pg.log2_hashmap_size, p(w_dec), p(self.feats[lvl]), pg.features_per_level,
                                        ns * pg.features_per_level, ns, S, sm, p(self.dens[lvl]), st), "prop_field_fwd")
            check(lib.nr_proposal_round(p(self.dens[lvl]), p(self.eu[lvl]), p(self.sp[lvl]), None, p(self.nears), p(self.fars), p(o), p(d),
                                        p(area), n, S, self.S[lvl + 1], lam, scal, SKY_DISTANCE if lvl == 1 else 0.0, scale, sm,
                                        p(self.w[lvl]), prop_depth[lvl].data_ptr() + 4 * lo, p(self.sp[lvl + 1]), p(self.eu[lvl + 1]),
                                        p(self.x01[lvl + 1]), p(self.std[lvl + 1]), st), "proposal_round")
        mg, Sm = self.mgrid, self.S[2]
        nm, F = n * Sm, mg.features_per_level
        fused_gather = mg.num_levels == 8 and F == 4 and fs.dtype != 0 and self.model.field.config.geo_hidden_dim == 32
        if fused_gather:
            check(lib.nr_field_fwd_gather(byref(fs), p(self.x01[2]), p(self.std[2]), p(mg.hash_table), p(mg.scalings), mg.num_levels, F,
                                          mg.log2_hashmap_size, None, nm * F, p(d), Sm, sm, nm, p(self.feature), p(self.sdf), p(self.alpha),
                                          st), "field_fwd_gather")
        else:
            check(lib.nr_hash_encode_fwd(p(self.x01[2]), p(self.std[2]), p(mg.hash_table), p(mg.scalings), mg.num_levels, F,
                                         mg.log2_hashmap_size, p(self.feats[2]), F, nm * F, nm, 0, st), "hash_fwd")
            check(lib.nr_field_fwd(byref(fs), p(self.feats[2]), F, nm * F, F, p(d), Sm, sm, nm, p(self.feature), p(self.sdf), p(self.alpha),
                                   st), "field_fwd")
        C = self.C
        check(lib.nr_composite_fwd(p(self.alpha), p(self.feature), p(self.eu[2]), n, Sm, C, p(self.w[2]), out["accumulation"].data_ptr() + 4 * lo,
                                   out["features"].data_ptr() + 4 * lo * C, out["depth"].data_ptr() + 4 * lo, st), "composite_fwd")
