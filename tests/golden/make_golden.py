#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REFERENCE's pure-PyTorch path on CPU.

Build-container tool (needs /root/reference; see ref_shim.py).  Run:  python tests/golden/make_golden.py
Outputs small .npz files next to this script.  Each file holds inputs, parameters and the
reference's outputs for one function group of SURVEY section 8a.  Tables are shrunk (T = 2^12..2^14) so
the files stay small; all seeds are fixed; random draws the reference makes internally are
reproduced by re-seeding and re-drawing in the same order, and stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

from nerfstudio.cameras.cameras import Cameras, CameraType  # noqa: E402
from nerfstudio.cameras.lidars import Lidars  # noqa: E402
from nerfstudio.cameras.radars import Radars  # noqa: E402
from nerfstudio.cameras.rays import Frustums, RayBundle, RaySamples  # noqa: E402
from nerfstudio.field_components.encodings import HashEncoding, SHEncoding  # noqa: E402
from nerfstudio.field_components.field_heads import FieldHeadNames  # noqa: E402
from nerfstudio.field_components.mlp import MLP  # noqa: E402
from nerfstudio.field_components.neurad_encoding import (  # noqa: E402
    ActorSettings, NeuRADHashEncodingConfig, StaticSettings)
from nerfstudio.field_components.spatial_distortions import ScaledSceneContraction  # noqa: E402
from nerfstudio.fields.neurad_field import NeuRADFieldConfig, NeuRADProposalFieldConfig  # noqa: E402
from nerfstudio.model_components import losses as ref_losses  # noqa: E402
from nerfstudio.model_components.dynamic_actors import DynamicActorsConfig  # noqa: E402
from nerfstudio.model_components.ray_generators import LidarRayGenerator, RayGenerator  # noqa: E402
from nerfstudio.model_components.ray_samplers import PDFSampler, PowerSampler, ProposalNetworkSampler  # noqa: E402
from nerfstudio.model_components.renderers import AccumulationRenderer, FeatureRenderer  # noqa: E402
from nerfstudio.utils.math import GaussiansStd  # noqa: E402

torch.set_default_dtype(torch.float32)
STATIC_SCALE = 100.0


def npy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def synth_rays(n, gen, near_camera=True):
    """Ray batch around an ego vehicle inside the +-100 m scene box."""
    origins = (torch.rand(n, 3, generator=gen) - 0.5) * torch.tensor([120.0, 120.0, 4.0])
    d = torch.randn(n, 3, generator=gen)
    d[:, 2] *= 0.3
    d = d / d.norm(dim=-1, keepdim=True)
    pixel_area = (2.5e-7 + 1e-6 * torch.rand(n, 1, generator=gen)) * 9.0
    times = torch.rand(n, 1, generator=gen) * 20.0
    return origins, d, pixel_area, times


# ----------------------------------------------------------------------------------------------
def golden_hash():
    """a8: HashEncoding.pytorch_fwd + hash_fn + scalings, three (L,F) shapes, fwd and table grad."""
    out = {}
    for tag, (L, F, lo, hi, log2t) in {
        "l8f4": (8, 4, 32, 8192, 11), "l6f1": (6, 1, 128, 4096, 11), "l16f2": (16, 2, 16, 1024, 10),
        "l4f4": (4, 4, 64, 1024, 10),
    }.items():
        torch.manual_seed(11)
        enc = HashEncoding(num_levels=L, min_res=lo, max_res=hi, log2_hashmap_size=log2t,
                           features_per_level=F, implementation="torch")
        with torch.no_grad():
            enc.hash_table.mul_(1000.0)  # O(1) features: makes relative checks meaningful
        g = torch.Generator().manual_seed(5)
        x = torch.rand(300, 3, generator=g)
        x[:8] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [0.25, 0.75, 1.0],
                              [1.0, 0.0, 0.5], [0.125, 0.0625, 0.03125], [0.999999, 0.000001, 0.5],
                              [0.3333333, 0.6666667, 0.1]])
        y = enc(x)
        gout = torch.randn(y.shape, generator=g)
        (gtab,) = torch.autograd.grad(y, enc.hash_table, gout)
        out.update({f"{tag}_x": x, f"{tag}_table": enc.hash_table, f"{tag}_scalings": enc.scalings,
                    f"{tag}_log2t": log2t, f"{tag}_out": y, f"{tag}_gout": gout, f"{tag}_gtable": gtab,
                    f"{tag}_cfg": np.array([L, F, lo, hi, log2t])})
    # the real NeuRadar level scalings (float32 floor quirk)
    out["scalings_static_main"] = HashEncoding(8, 32, 8192, 4, 4, implementation="torch").scalings
    out["scalings_static_prop"] = HashEncoding(6, 128, 4096, 4, 1, implementation="torch").scalings
    out["scalings_actor"] = HashEncoding(4, 64, 1024, 4, 4, implementation="torch").scalings
    out["scalings_class_default"] = HashEncoding(16, 16, 1024, 4, 2, implementation="torch").scalings
    out["scalings_nerfacto"] = HashEncoding(16, 16, 2048, 4, 2, implementation="torch").scalings
    # raw hash slots for a few integer corners incl. negatives
    enc = HashEncoding(num_levels=2, min_res=16, max_res=32, log2_hashmap_size=14, features_per_level=1,
                       implementation="torch")
    corners = torch.tensor([[[0, 0, 0], [0, 0, 0]], [[1, 2, 3], [3, 2, 1]], [[8191, 8191, 8191], [17, 0, 4095]],
                            [[-1, 5, 7], [5, -7, 2]], [[123456, 654321, 999], [2, 4, 8]]], dtype=torch.int32)
    out["hashfn_corners"] = corners
    out["hashfn_slots"] = enc.hash_fn(corners)
    save("hash_encode", **out)


def golden_gaussian_contraction():
    """a6 + a7 + a9: fast isotropic gaussian, scaled L-inf contraction, per-level rescale."""
    g = torch.Generator().manual_seed(21)
    B, S = 48, 16
    o, d, area, _ = synth_rays(B, g)
    edges = torch.sort(torch.rand(B, S + 1, generator=g) ** 3 * 900.0, dim=-1).values + 0.01
    edges[:, -1] = torch.where(torch.arange(B) % 3 == 0, torch.tensor(20000.0), edges[:, -1])
    bundle = RayBundle(origins=o, directions=d, pixel_area=area, nears=torch.zeros(B, 1),
                       fars=torch.full((B, 1), 1e6), metadata={})
    rs = bundle.get_ray_samples(bin_starts=edges[:, :-1, None], bin_ends=edges[:, 1:, None])
    gs = rs.frustums.get_fast_isotropic_gaussian(1)
    con = ScaledSceneContraction(order=float("inf"), scale=STATIC_SCALE)(GaussiansStd(gs.mean.clone(), gs.std.clone()))
    con_actor = ScaledSceneContraction(order=float("inf"), scale=10.0)(GaussiansStd(gs.mean.clone(), gs.std.clone()))
    # rescale via a NeuRADHashEncoding instance (static branch only)
    actors = DynamicActorsConfig().setup(trajectories=[])
    torch.manual_seed(3)
    hg = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=12)).setup(
        dynamic_actors=actors, static_scale=STATIC_SCALE, implementation="torch")
    with torch.no_grad():
        hg.static_grid.hash_table.mul_(1000.0)
    feats, _ = hg(gs, torch.zeros(B, S, 1), None)
    save("gaussian_contraction", origins=o, directions=d, pixel_area=area, edges=edges,
         mean=gs.mean[:, :, 0, :], std=gs.std[:, :, 0, :], mean01=con.mean[:, :, 0, :], std01=con.std[:, :, 0, :],
         mean01_actor=con_actor.mean[:, :, 0, :], std01_actor=con_actor.std[:, :, 0, :],
         table=hg.static_grid.hash_table, scalings=hg.static_grid.scalings, log2t=12,
         grid_features=feats.view(B, S, -1))


def _field_param_dict(prefix, fld):
    out = {f"{prefix}table": fld.hashgrid.static_grid.hash_table, f"{prefix}scalings": fld.hashgrid.static_grid.scalings}
    if hasattr(fld, "mlp_geo"):
        for i, lyr in enumerate(fld.mlp_geo.layers):
            out[f"{prefix}geo_w{i}"], out[f"{prefix}geo_b{i}"] = lyr.weight, lyr.bias
        for i, lyr in enumerate(fld.mlp_feature.layers):
            out[f"{prefix}feat_w{i}"], out[f"{prefix}feat_b{i}"] = lyr.weight, lyr.bias
        out[f"{prefix}beta"] = fld.sdf_to_density.beta
    else:
        out[f"{prefix}decoder"] = fld.density_decoder.weight
    return out


def _make_fields(log2t_main=13, log2t_prop=12, static=None, hidden=32, seed=7):
    actors = DynamicActorsConfig().setup(trajectories=[])
    torch.manual_seed(seed)
    static = static or StaticSettings(log2_hashmap_size=log2t_main)
    fld = NeuRADFieldConfig(
        grid=NeuRADHashEncodingConfig(static=static, actor=ActorSettings(flip_prob=0.25)),
        geo_hidden_dim=hidden, nff_hidden_dim=hidden,
    ).setup(actors=actors, static_scale=STATIC_SCALE, implementation="torch")
    props = []
    for _ in range(2):
        pc = NeuRADProposalFieldConfig()
        pc.grid.static.log2_hashmap_size = log2t_prop
        props.append(pc.setup(actors=actors, static_scale=STATIC_SCALE, implementation="torch"))
    with torch.no_grad():  # O(0.1-1) features so the MLP/density outputs are not degenerate
        fld.hashgrid.static_grid.hash_table.mul_(300.0)
        for p in props:
            p.hashgrid.static_grid.hash_table.mul_(2000.0)
            p.density_decoder.weight.mul_(2.0)
    return fld, props


def golden_field():
    """a15 (+a8,a9): NeuRADField forward/backward at NeuRadar dims and at the L16/F2/64-wide variant;
    a11: NeuRADProposalField.get_density fwd/bwd."""
    for tag, kw in {"neurad": {}, "l16f2w64": dict(
            static=StaticSettings(hashgrid_dim=2, num_levels=16, base_res=16, max_res=1024, log2_hashmap_size=12),
            hidden=64)}.items():
        fld, props = _make_fields(**kw)
        g = torch.Generator().manual_seed(31)
        B, S = 40, 12
        o, d, area, times = synth_rays(B, g)
        edges = torch.sort(torch.rand(B, S + 1, generator=g) ** 2 * 300.0, dim=-1).values + 0.05
        bundle = RayBundle(origins=o, directions=d, pixel_area=area, nears=torch.zeros(B, 1),
                           fars=torch.full((B, 1), 1e6), times=times, metadata={})
        rs = bundle.get_ray_samples(bin_starts=edges[:, :-1, None], bin_ends=edges[:, 1:, None])
        out = fld(rs)
        feat, sdf, alpha = out[FieldHeadNames.FEATURE], out[FieldHeadNames.SDF], out[FieldHeadNames.ALPHA]
        g_feat, g_alpha = torch.randn(feat.shape, generator=g), torch.randn(alpha.shape, generator=g)
        params = [p for p in fld.parameters()]
        names = [n for n, _ in fld.named_parameters()]
        grads = torch.autograd.grad((feat * g_feat).sum() + (alpha * g_alpha).sum(), params, allow_unused=True)
        arrays = dict(origins=o, directions=d, pixel_area=area, edges=edges, feature=feat, sdf=sdf, alpha=alpha,
                      g_feature=g_feat, g_alpha=g_alpha, log2t=int(np.log2(fld.hashgrid.static_grid.hash_table_size)))
        arrays.update(_field_param_dict("", fld))
        gmap = dict(zip(names, grads))
        arrays["grad_table"] = gmap["hashgrid.static_grid.hash_table"]
        for i in range(len(fld.mlp_geo.layers)):
            arrays[f"grad_geo_w{i}"], arrays[f"grad_geo_b{i}"] = gmap[f"mlp_geo.layers.{i}.weight"], gmap[f"mlp_geo.layers.{i}.bias"]
        for i in range(len(fld.mlp_feature.layers)):
            arrays[f"grad_feat_w{i}"], arrays[f"grad_feat_b{i}"] = gmap[f"mlp_feature.layers.{i}.weight"], gmap[f"mlp_feature.layers.{i}.bias"]
        arrays["grad_beta"] = gmap["sdf_to_density.beta"]
        if tag == "neurad":
            pf = props[1]
            dens, _ = pf.get_density(rs)
            g_d = torch.randn(dens.shape, generator=g)
            gt, gw = torch.autograd.grad((dens * g_d).sum(), [pf.hashgrid.static_grid.hash_table, pf.density_decoder.weight])
            arrays.update(_field_param_dict("prop_", pf))
            arrays.update(prop_density=dens, prop_g_density=g_d, prop_grad_table=gt, prop_grad_decoder=gw,
                          prop_log2t=12)
        save(f"field_{tag}", **arrays)


def golden_field_autocast():
    """N2: the reference's own NeuRADField under torch.autocast (engine/trainer.py:564 wraps the whole forward in it) on
    the inputs and parameters of field_{neurad,l16f2w64}.npz -- only the outputs and gradients are stored here, the
    inputs live in those files.  CPU autocast runs every nn.Linear with 16-bit inputs, weights AND outputs."""
    arrays = {}
    for tag, kw in {"neurad": {}, "l16f2w64": dict(
            static=StaticSettings(hashgrid_dim=2, num_levels=16, base_res=16, max_res=1024, log2_hashmap_size=12),
            hidden=64)}.items():
        fld, _ = _make_fields(**kw)
        g = torch.Generator().manual_seed(31)
        B, S = 40, 12
        o, d, area, times = synth_rays(B, g)
        edges = torch.sort(torch.rand(B, S + 1, generator=g) ** 2 * 300.0, dim=-1).values + 0.05
        bundle = RayBundle(origins=o, directions=d, pixel_area=area, nears=torch.zeros(B, 1),
                           fars=torch.full((B, 1), 1e6), times=times, metadata={})
        rs = bundle.get_ray_samples(bin_starts=edges[:, :-1, None], bin_ends=edges[:, 1:, None])
        g_feat, g_alpha = torch.randn(B, S, 32, generator=g), torch.randn(B, S, 1, generator=g)  # same draws as golden_field
        for dtype in (torch.bfloat16, torch.float16):
            with torch.autocast("cpu", dtype=dtype):
                out = fld(rs)
            feat, sdf, alpha = (out[k].float() for k in (FieldHeadNames.FEATURE, FieldHeadNames.SDF, FieldHeadNames.ALPHA))
            names = [n for n, _ in fld.named_parameters()]
            grads = torch.autograd.grad((feat * g_feat).sum() + (alpha * g_alpha).sum(), list(fld.parameters()), allow_unused=True)
            gmap = dict(zip(names, grads))
            pre = f"{tag}_{str(dtype).split('.')[-1]}_"
            arrays.update({pre + "feature": feat, pre + "sdf": sdf, pre + "alpha": alpha,
                           pre + "grad_table": gmap["hashgrid.static_grid.hash_table"]})
            for i in range(2):
                arrays[pre + f"grad_geo_w{i}"] = gmap[f"mlp_geo.layers.{i}.weight"]
            for i in range(3):
                arrays[pre + f"grad_feat_w{i}"] = gmap[f"mlp_feature.layers.{i}.weight"]
    save("field_autocast", **arrays)


def golden_field_density():
    """a16, use_sdf=False branch: NeuRADField emits DENSITY = trunc_exp(geo_out) (fields/neurad_field.py:149-150)."""
    actors = DynamicActorsConfig().setup(trajectories=[])
    torch.manual_seed(17)
    fld = NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=10), actor=ActorSettings(flip_prob=0.25)),
                            use_sdf=False).setup(actors=actors, static_scale=STATIC_SCALE, implementation="torch")
    with torch.no_grad():
        fld.hashgrid.static_grid.hash_table.mul_(300.0)
    g = torch.Generator().manual_seed(33)
    B, S = 24, 10
    o, d, area, times = synth_rays(B, g)
    edges = torch.sort(torch.rand(B, S + 1, generator=g) ** 2 * 300.0, dim=-1).values + 0.05
    bundle = RayBundle(origins=o, directions=d, pixel_area=area, nears=torch.zeros(B, 1), fars=torch.full((B, 1), 1e6), times=times, metadata={})
    rs = bundle.get_ray_samples(bin_starts=edges[:, :-1, None], bin_ends=edges[:, 1:, None])
    out = fld(rs)
    assert FieldHeadNames.SDF not in out and FieldHeadNames.ALPHA not in out
    feat, dens = out[FieldHeadNames.FEATURE], out[FieldHeadNames.DENSITY]
    g_feat, g_dens = torch.randn(feat.shape, generator=g), torch.randn(dens.shape, generator=g)
    names = [n for n, _ in fld.named_parameters()]
    grads = torch.autograd.grad((feat * g_feat).sum() + (dens * g_dens).sum(), list(fld.parameters()), allow_unused=True)
    gmap = dict(zip(names, grads))
    arrays = dict(origins=o, directions=d, pixel_area=area, edges=edges, feature=feat, density=dens, g_feature=g_feat, g_density=g_dens,
                  log2t=10)
    arrays.update({"table": fld.hashgrid.static_grid.hash_table, "scalings": fld.hashgrid.static_grid.scalings})
    for i, lyr in enumerate(fld.mlp_geo.layers):
        arrays[f"geo_w{i}"], arrays[f"geo_b{i}"] = lyr.weight, lyr.bias
        arrays[f"grad_geo_w{i}"], arrays[f"grad_geo_b{i}"] = gmap[f"mlp_geo.layers.{i}.weight"], gmap[f"mlp_geo.layers.{i}.bias"]
    for i, lyr in enumerate(fld.mlp_feature.layers):
        arrays[f"feat_w{i}"], arrays[f"feat_b{i}"] = lyr.weight, lyr.bias
        arrays[f"grad_feat_w{i}"] = gmap[f"mlp_feature.layers.{i}.weight"]
    arrays["grad_table"] = gmap["hashgrid.static_grid.hash_table"]
    save("field_density", **arrays)


def golden_sh_mlp():
    """SHEncoding torch path (a15) and a bare MLP."""
    g = torch.Generator().manual_seed(41)
    d = torch.randn(64, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    sh_raw = SHEncoding(levels=4, implementation="torch")(d)
    sh_01 = SHEncoding(levels=4, implementation="torch")((d + 1.0) / 2.0)
    torch.manual_seed(2)
    m = MLP(in_dim=48, num_layers=3, layer_width=32, out_dim=2, implementation="torch")  # lidar decoder shape (K7)
    x = torch.randn(64, 48, generator=g)
    save("sh_mlp", dirs=d, sh_raw=sh_raw, sh_01=sh_01, mlp_x=x, mlp_y=m(x),
         **{f"mlp_w{i}": l.weight for i, l in enumerate(m.layers)}, **{f"mlp_b{i}": l.bias for i, l in enumerate(m.layers)})


def golden_sampler():
    """a5, a12, a13, a14: PowerSampler bins, get_weights, PDFSampler, ProposalNetworkSampler."""
    g = torch.Generator().manual_seed(51)
    B = 32
    o, d, area, times = synth_rays(B, g)
    fars = torch.full((B, 1), 20000.0)
    fars[::4] = 150.0 + 100 * torch.rand(B // 4, 1, generator=g)
    nears = torch.zeros(B, 1)
    bundle = RayBundle(origins=o, directions=d, pixel_area=area, nears=nears, fars=fars, times=times, metadata={})
    out = dict(origins=o, directions=d, pixel_area=area, nears=nears, fars=fars)

    ps = PowerSampler(lambda_=-1.0, scaling=0.1)
    ps.eval()
    rs = ps(bundle, num_samples=128)
    out.update(power_eval_spacing=torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], -1),
               power_eval_euclid=torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[:, -1:, 0]], -1))
    ps.train()
    torch.manual_seed(99)
    rs_t = ps(bundle, num_samples=128)
    torch.manual_seed(99)
    t_rand = torch.rand((B, 129))
    out.update(power_train_t_rand=t_rand,
               power_train_spacing=torch.cat([rs_t.spacing_starts[..., 0], rs_t.spacing_ends[:, -1:, 0]], -1),
               power_train_euclid=torch.cat([rs_t.frustums.starts[..., 0], rs_t.frustums.ends[:, -1:, 0]], -1))

    # get_weights on a peaky density incl. zeros and huge values
    dens = torch.exp(torch.randn(B, 128, 1, generator=g) * 3.0 - 4.0)
    dens[0] = 0.0
    dens[1, 40:] = 1e6
    w = rs_t.get_weights(dens)
    out.update(gw_density=dens[..., 0], gw_weights=w[..., 0], gw_deltas=rs_t.deltas[..., 0])

    pdf = PDFSampler(include_original=False, single_jitter=True)
    pdf.eval()
    rs2 = pdf(bundle, rs_t, w, num_samples=64)
    out.update(pdf_eval_spacing=torch.cat([rs2.spacing_starts[..., 0], rs2.spacing_ends[:, -1:, 0]], -1),
               pdf_eval_euclid=torch.cat([rs2.frustums.starts[..., 0], rs2.frustums.ends[:, -1:, 0]], -1))
    pdf.train()
    torch.manual_seed(123)
    rs3 = pdf(bundle, rs_t, w, num_samples=64)
    torch.manual_seed(123)
    jit = torch.rand((B, 1))
    out.update(pdf_train_jitter=jit,
               pdf_train_spacing=torch.cat([rs3.spacing_starts[..., 0], rs3.spacing_ends[:, -1:, 0]], -1),
               pdf_train_euclid=torch.cat([rs3.frustums.starts[..., 0], rs3.frustums.ends[:, -1:, 0]], -1))
    save("sampler", **out)


def _edges(rs):
    return (torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], -1),
            torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[:, -1:, 0]], -1))


def golden_pipeline():
    """a4 + a14..a18 end to end: reference sampler (with the late-binding quirk) -> sky trick ->
    reference field -> restated nerfacc weights -> reference renderers; plus the two regularisers
    and the parameter gradients of the bench loss."""
    fld, props = _make_fields(log2t_main=13, log2t_prop=12)
    g = torch.Generator().manual_seed(61)
    B = 24
    o, d, area, times = synth_rays(B, g)
    fars = torch.full((B, 1), 1e6)
    bundle = RayBundle(origins=o, directions=d, pixel_area=area, fars=fars.clone(), times=times, metadata={})
    sampler = ProposalNetworkSampler(
        num_proposal_samples_per_ray=(128, 64), num_nerf_samples_per_ray=32, num_proposal_network_iterations=2,
        single_jitter=True, initial_sampler=PowerSampler(lambda_=-1.0, scaling=0.1), update_sched=lambda x: 0)
    sampler.train()
    density_fns = [lambda x: prop_field.get_density(x)[0] for prop_field in props]  # same quirk as neuradar.py:302
    # _get_ray_samples (neuradar.py:570-586)
    bundle.fars.clamp_max_(20000.0)
    bundle.nears = torch.zeros_like(bundle.fars)
    torch.manual_seed(777)
    rs, prop_w, prop_rs = sampler(bundle, density_fns, pass_ray_samples=True)
    torch.manual_seed(777)
    t_rand, j1, j2 = torch.rand((B, 129)), torch.rand((B, 1)), torch.rand((B, 1))
    dist_to_sky = 20000.0 - rs.frustums.ends[..., -1, 0]
    rs.frustums.ends[..., -1, 0] += dist_to_sky
    rs.deltas[..., -1, 0] += dist_to_sky
    rs.spacing_ends[..., -1, 0] = 1 - 1e-7
    outs = fld(rs)
    alpha = outs[FieldHeadNames.ALPHA]
    a = alpha[..., 0]
    trans = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), (1 - a)[:, :-1]], -1), -1)  # nerfacc 0.5.2 batched branch
    weights = a * trans
    acc = AccumulationRenderer()(weights=weights[..., None])
    weights = torch.cat((weights[..., :-1], weights[..., -1:] + 1 - acc), dim=-1).unsqueeze(-1)
    features = FeatureRenderer()(features=outs[FieldHeadNames.FEATURE], weights=weights)
    w31, rs31 = weights[..., :-1, :], rs[..., :-1]
    steps = (rs31.frustums.starts + rs31.frustums.ends) / 2
    depth = torch.sum(w31[..., 0][..., None] * steps, dim=-2)
    # in-repo cross-check of the unpinned nerfacc formula (differs by +1e-7 per factor)
    w_alt = RaySamples.get_weights_and_transmittance_from_alphas(alpha, weights_only=True)
    weights_list = prop_w + [w31]
    rs_list = prop_rs + [rs31]
    inter = ref_losses.zipnerf_interlevel_loss(weights_list, rs_list)
    dist = ref_losses.distortion_loss(weights_list, rs_list)
    tgt_f = torch.randn(B, 32, generator=g) * 0.1
    tgt_d = torch.rand(B, 1, generator=g) * 50.0
    loss = 5.0 * torch.mean((features - tgt_f) ** 2) + 0.01 * (depth - tgt_d).abs().mean() + 1e-3 * inter + 2e-3 * dist
    plist = [("main_" + n, p) for n, p in fld.named_parameters()] + [("prop1_" + n, p) for n, p in props[1].named_parameters()]
    grads = torch.autograd.grad(loss, [p for _, p in plist], allow_unused=True)
    plist, grads = zip(*[(pl, gr) for pl, gr in zip(plist, grads) if gr is not None])
    sp0, eu0 = _edges(prop_rs[0])
    sp1, eu1 = _edges(prop_rs[1])
    sp2, eu2 = _edges(rs)
    arrays = dict(origins=o, directions=d, pixel_area=area, fars=fars, t_rand=t_rand, jitter1=j1, jitter2=j2,
                  prop_spacing_0=sp0, prop_euclid_0=eu0, prop_spacing_1=sp1, prop_euclid_1=eu1,
                  final_spacing=sp2, final_euclid=eu2, prop_weights_0=prop_w[0][..., 0], prop_weights_1=prop_w[1][..., 0],
                  alpha=alpha, sdf=outs[FieldHeadNames.SDF], feature_samples=outs[FieldHeadNames.FEATURE],
                  weights=weights[..., 0], weights_alt_inrepo=w_alt[..., 0], accumulation=acc, features=features,
                  depth=depth, prop_depth_0=torch.sum(prop_w[0] * (prop_rs[0].frustums.starts + prop_rs[0].frustums.ends) / 2, dim=-2),
                  prop_depth_1=torch.sum(prop_w[1] * (prop_rs[1].frustums.starts + prop_rs[1].frustums.ends) / 2, dim=-2),
                  interlevel=inter, distortion=dist, target_features=tgt_f, target_depth=tgt_d, loss=loss,
                  main_log2t=13, prop_log2t=12)
    arrays.update(_field_param_dict("main_", fld))
    arrays.update(_field_param_dict("prop1_", props[1]))
    for (n, _), gr in zip(plist, grads):
        arrays["grad_" + n.replace(".", "_")] = gr
    save("pipeline", **arrays)


def golden_losses():
    """f-3: zipnerf_interlevel_loss / distortion_loss on hand-made histograms (reference functions)."""
    g = torch.Generator().manual_seed(71)
    B = 16

    def mk(S):
        c = torch.sort(torch.rand(B, S + 1, generator=g), dim=-1).values
        c[:, 0], c[:, -1] = 0.0, 1.0 - 1e-7
        w = torch.rand(B, S, generator=g)
        w = w / w.sum(-1, keepdim=True) * torch.rand(B, 1, generator=g)
        return c, w.requires_grad_(True)

    cs, ws = zip(*[mk(S) for S in (128, 64, 31)])

    def rs_of(c):
        fr = Frustums(origins=torch.zeros(B, c.shape[1] - 1, 3), directions=torch.ones(B, c.shape[1] - 1, 3),
                      starts=c[:, :-1, None], ends=c[:, 1:, None], pixel_area=torch.ones(B, c.shape[1] - 1, 1))
        return RaySamples(frustums=fr, spacing_starts=c[:, :-1, None], spacing_ends=c[:, 1:, None])

    wl = [w[..., None] for w in ws]
    rl = [rs_of(c) for c in cs]
    inter = ref_losses.zipnerf_interlevel_loss(wl, rl)
    dist = ref_losses.distortion_loss(wl, rl)
    gi = torch.autograd.grad(inter, [ws[0], ws[1]])
    (gd,) = torch.autograd.grad(dist, [ws[2]])
    save("losses", c0=cs[0], c1=cs[1], c2=cs[2], w0=ws[0], w1=ws[1], w2=ws[2], interlevel=inter, distortion=dist,
         g_inter_w0=gi[0], g_inter_w1=gi[1], g_dist_w2=gd)


def golden_raygen():
    """a1, a2, a3: camera (pinhole + rolling shutter), lidar, radar (ZOD + VoD FOV)."""
    g = torch.Generator().manual_seed(81)

    def rand_pose(n):
        q = torch.randn(n, 4, generator=g)
        q = q / q.norm(dim=-1, keepdim=True)
        w, x, y, z = q.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).view(n, 3, 3)
        t = (torch.rand(n, 3, 1, generator=g) - 0.5) * 150.0
        return torch.cat([R, t], dim=-1)

    # --- camera
    C, H, W = 5, 1080, 1920
    c2w = rand_pose(C)
    fx = 1900.0 + 200 * torch.rand(C, 1, generator=g)
    fy = 1900.0 + 200 * torch.rand(C, 1, generator=g)
    cx = W / 2 + 10 * torch.randn(C, 1, generator=g)
    cy = H / 2 + 10 * torch.randn(C, 1, generator=g)
    cam_times = torch.rand(C, generator=g) * 20
    vel = torch.randn(C, 3, generator=g) * 10
    rs_off = torch.tensor([[-0.015, 0.017]]).repeat(C, 1)
    cams = Cameras(camera_to_worlds=c2w, fx=fx, fy=fy, cx=cx, cy=cy, width=W, height=H,
                   camera_type=CameraType.PERSPECTIVE, times=cam_times,
                   metadata={"velocities": vel, "rolling_shutter_offsets": rs_off})
    n = 200
    ridx = torch.stack([torch.randint(0, C, (n,), generator=g), torch.randint(0, H, (n,), generator=g),
                        torch.randint(0, W, (n,), generator=g)], -1)
    ridx[:4] = torch.tensor([[0, 0, 0], [1, H - 1, W - 1], [2, 0, W - 1], [3, H - 1, 0]])
    cb = RayGenerator(cams)(ridx)
    out = dict(cam_ray_indices=ridx, cam_c2w=c2w, cam_fx=fx[:, 0], cam_fy=fy[:, 0], cam_cx=cx[:, 0], cam_cy=cy[:, 0],
               cam_times_in=cam_times, cam_vel=vel, cam_rs_offsets=rs_off, cam_heights=torch.full((C,), float(H)),
               cam_origins=cb.origins, cam_directions=cb.directions, cam_pixel_area=cb.pixel_area, cam_times=cb.times,
               cam_fars=cb.fars, cam_directions_norm=cb.metadata["directions_norm"])
    cams_nors = Cameras(camera_to_worlds=c2w, fx=fx, fy=fy, cx=cx, cy=cy, width=W, height=H,
                        camera_type=CameraType.PERSPECTIVE, times=cam_times)
    cb2 = RayGenerator(cams_nors)(ridx)
    out.update(cam_nors_origins=cb2.origins, cam_nors_times=cb2.times)
    # ZOD-style cameras: FISHEYE model with radial distortion [k1..k4, 0, 0] (zod_dataparser.py:236-261)
    dist = torch.cat([torch.tensor([[0.08, -0.02, 0.004, -0.0005]]).repeat(C, 1) * (1 + 0.1 * torch.randn(C, 4, generator=g)),
                      torch.zeros(C, 2)], -1)
    dist[1, 4:] = torch.tensor([1e-3, -2e-3])  # one camera with tangential terms as well
    cams_fe = Cameras(camera_to_worlds=c2w, fx=fx, fy=fy, cx=cx, cy=cy, width=W, height=H, distortion_params=dist,
                      camera_type=CameraType.FISHEYE, times=cam_times)
    cb3 = RayGenerator(cams_fe)(ridx)
    cams_pd = Cameras(camera_to_worlds=c2w, fx=fx, fy=fy, cx=cx, cy=cy, width=W, height=H, distortion_params=dist,
                      camera_type=CameraType.PERSPECTIVE, times=cam_times)
    cb4 = RayGenerator(cams_pd)(ridx)
    out.update(cam_dist=dist, cam_fe_directions=cb3.directions, cam_fe_pixel_area=cb3.pixel_area,
               cam_pd_directions=cb4.directions, cam_pd_pixel_area=cb4.pixel_area)

    # --- lidar
    NL = 4
    l2w = rand_pose(NL)
    l_times = torch.rand(NL, generator=g) * 20
    l_vel = torch.randn(NL, 3, generator=g) * 10
    lid = Lidars(lidar_to_worlds=l2w, times=l_times, metadata={"velocities": l_vel}, assume_ego_compensated=True)
    npts = 150
    pts = torch.cat([torch.randn(npts, 3, generator=g) * 40, torch.rand(npts, 1, generator=g),
                     (torch.rand(npts, 1, generator=g) - 0.5) * 0.1], -1)
    pts[:5, :3] = pts[:5, :3] / pts[:5, :3].norm(dim=-1, keepdim=True) * 2000.0  # non-returns (>1e3 m)
    lidx = torch.randint(0, NL, (npts, 1), generator=g)
    lb = LidarRayGenerator(lid)(torch.cat([lidx, torch.arange(npts)[:, None]], -1), points=pts)
    out.update(lid_indices=lidx[:, 0], lid_points=pts, lid_l2w=l2w, lid_times_in=l_times, lid_vel=l_vel,
               lid_origins=lb.origins, lid_directions=lb.directions, lid_pixel_area=lb.pixel_area, lid_times=lb.times,
               lid_fars=lb.fars, lid_directions_norm=lb.metadata["directions_norm"],
               lid_did_return=lb.metadata["did_return"], lid_is_lidar=lb.metadata["is_lidar"])

    # --- radar: ZOD FOV (zod_dataparser.py:138-140) and VoD FOV (vod_dataparser.py:46-48)
    NR = 3
    r2w = rand_pose(NR)
    r_times = torch.rand(NR, generator=g) * 20
    zod = Radars(radar_to_worlds=r2w, times=r_times, radar_azimuth_ray_divergence=0.015,
                 radar_elevation_ray_divergence=0.015, min_azimuth=-0.80, max_azimuth=0.80,
                 min_elevation=-0.08, max_elevation=0.4)
    scans = torch.tensor([2, 0])
    rb = zod.generate_rays(scan_indices=scans)
    out.update(rad_r2w=r2w, rad_times_in=r_times, rad_scans=scans, rad_zod_fov=np.array([-0.80, 0.80, 0.015, -0.08, 0.4, 0.015]),
               rad_origins=rb.origins, rad_directions=rb.directions, rad_pixel_area=rb.pixel_area, rad_times=rb.times,
               rad_fars=rb.fars, rad_directions_spher=rb.metadata["directions_spher"],
               rad_directions_norm=rb.metadata["directions_norm"], rad_scan_of_ray=rb.camera_indices[:, 0])
    dflt = Radars(radar_to_worlds=r2w, times=r_times)
    rb2 = dflt.generate_rays(scan_indices=torch.tensor([1]))
    out.update(rad_default_fov=np.array([-0.5, 0.5, 0.0625, -0.5, 0.5, 0.0625]), rad_default_directions=rb2.directions,
               rad_default_pixel_area=rb2.pixel_area, rad_default_directions_spher=rb2.metadata["directions_spher"])
    vod = Radars(radar_to_worlds=r2w, times=r_times, radar_azimuth_ray_divergence=0.02,
                 radar_elevation_ray_divergence=0.02, min_azimuth=-1.0, max_azimuth=1.0,
                 min_elevation=-0.39, max_elevation=0.49)
    rb3 = vod.generate_rays(scan_indices=torch.tensor([1]))
    out.update(rad_vod_fov=np.array([-1.0, 1.0, 0.02, -0.39, 0.49, 0.02]), rad_vod_directions=rb3.directions,
               rad_vod_pixel_area=rb3.pixel_area, rad_vod_directions_spher=rb3.metadata["directions_spher"])
    save("raygen", **out)


def golden_actors():
    """a10: dynamic actors -- 2 boxes on a toy trajectory, rays that cross them; NeuRADField (one 3-D grid
    per actor, the torch path) in eval mode and in train mode with the per-ray flip reproduced;
    gradients incl. the trajectory parameters; NeuRADProposalField density."""
    from nerfstudio.cameras.camera_utils import matrix_to_rotation_6d  # noqa: F401

    def pose(x, y, yaw):
        c, s = np.cos(yaw), np.sin(yaw)
        m = torch.eye(4)
        m[:3, :3] = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        m[:3, 3] = torch.tensor([x, y, 0.8])
        return m

    ts = torch.tensor([0.0, 1.0, 2.0])
    trajs = [
        {"poses": torch.stack([pose(8.0 + 3 * t, 1.0, 0.1 * t) for t in ts.tolist()]), "timestamps": ts.clone(),
         "dims": torch.tensor([2.0, 4.6, 1.6]), "symmetric": True, "deformable": False},
        {"poses": torch.stack([pose(14.0, -3.0, 1.2) for _ in ts.tolist()[:2]]), "timestamps": ts[:2].clone(),
         "dims": torch.tensor([1.9, 4.2, 1.5]), "symmetric": True, "deformable": False},
    ]
    torch.manual_seed(17)
    actors = DynamicActorsConfig().setup(trajectories=trajs)
    grid_cfg = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=12),
                                        actor=ActorSettings(flip_prob=0.25, log2_hashmap_size=10))
    fld = NeuRADFieldConfig(grid=grid_cfg).setup(actors=actors, static_scale=STATIC_SCALE, implementation="torch")
    pcfg = NeuRADProposalFieldConfig()
    pcfg.grid.static.log2_hashmap_size = 12
    pcfg.grid.actor.log2_hashmap_size = 10
    prop = pcfg.setup(actors=actors, static_scale=STATIC_SCALE, implementation="torch")
    with torch.no_grad():
        fld.hashgrid.static_grid.hash_table.mul_(300.0)
        for gr in fld.hashgrid.actor_grids:
            gr.hash_table.mul_(400.0)
        prop.hashgrid.static_grid.hash_table.mul_(2000.0)
        for gr in prop.hashgrid.actor_grids:
            gr.hash_table.mul_(2500.0)
        prop.density_decoder.weight.mul_(2.0)
    g = torch.Generator().manual_seed(91)
    B, S = 48, 24
    o = torch.cat([torch.randn(B, 2, generator=g) * 0.5, torch.full((B, 1), 1.2)], -1)
    tgt = torch.stack([8.0 + 8 * torch.rand(B, generator=g), -4.0 + 6 * torch.rand(B, generator=g), 0.3 + torch.rand(B, generator=g)], -1)
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    area = torch.full((B, 1), 2e-6)
    times = torch.rand(B, 1, generator=g) * 2.2
    edges = torch.linspace(0.5, 26.0, S + 1)[None, :].repeat(B, 1) + 0.3 * torch.rand(B, 1, generator=g)
    bundle = RayBundle(origins=o, directions=d, pixel_area=area, nears=torch.zeros(B, 1), fars=torch.full((B, 1), 1e6),
                       times=times, metadata={})
    rs = bundle.get_ray_samples(bin_starts=edges[:, :-1, None], bin_ends=edges[:, 1:, None])
    out = {}
    fld.eval(); prop.eval()
    ev = fld(rs)
    out.update(eval_feature=ev[FieldHeadNames.FEATURE], eval_sdf=ev[FieldHeadNames.SDF], eval_alpha=ev[FieldHeadNames.ALPHA],
               eval_prop_density=prop.get_density(rs)[0])
    # how many samples fell inside actors (sanity of the fixture)
    with torch.no_grad():
        gs = rs.frustums.get_fast_isotropic_gaussian(1)
        (ri, si, ai), _, _ = fld.hashgrid._split_static_vs_actors(gs, rs.times, rs.frustums.directions)
    out.update(actor_ray_idx=ri, actor_sample_idx=si, actor_actor_idx=ai)
    fld.train(); prop.train()
    torch.manual_seed(5)
    tr = fld(rs)
    torch.manual_seed(5)
    flip = torch.bernoulli(torch.full((B,), 0.25)) * -2 + 1
    feat, alpha = tr[FieldHeadNames.FEATURE], tr[FieldHeadNames.ALPHA]
    g_feat, g_alpha = torch.randn(feat.shape, generator=g), torch.randn(alpha.shape, generator=g)
    named = dict(fld.named_parameters())
    keys = ["hashgrid.static_grid.hash_table", "hashgrid.actor_grids.0.hash_table", "hashgrid.actor_grids.1.hash_table",
            "hashgrid.actors.actor_positions", "hashgrid.actors.actor_rotations_6d", "mlp_geo.layers.0.weight"]
    grads = torch.autograd.grad((feat * g_feat).sum() + (alpha * g_alpha).sum(), [named[k] for k in keys])
    out.update(train_feature=feat, train_sdf=tr[FieldHeadNames.SDF], train_alpha=alpha, flip=flip, g_feature=g_feat, g_alpha=g_alpha)
    for k, gr in zip(keys, grads):
        out["grad_" + k.replace(".", "_")] = gr
    a = fld.hashgrid.actors
    out.update(origins=o, directions=d, pixel_area=area, times=times, edges=edges, actor_positions=a.actor_positions,
               actor_rotations_6d=a.actor_rotations_6d, actor_timestamps=a.unique_timestamps, actor_present=a.actor_present_at_time,
               actor_sizes=a.actor_sizes, actor_padding=a.actor_padding, log2t=12, actor_log2t=10, prop_log2t=12)
    out.update(_field_param_dict("", fld))
    out.update({f"actor{i}_table": gr.hash_table for i, gr in enumerate(fld.hashgrid.actor_grids)})
    out["actor_scalings"] = fld.hashgrid.actor_grids[0].scalings
    out.update(_field_param_dict("prop_", prop))
    out.update({f"prop_actor{i}_table": gr.hash_table for i, gr in enumerate(prop.hashgrid.actor_grids)})
    out["prop_actor_scalings"] = prop.hashgrid.actor_grids[0].scalings
    save("actors", **out)


def _build_reference_model(seed=11):
    """The reference NeuRadarModel itself, small tables, VGG loss patched out (needs torchvision weights)."""
    import nerfstudio.models.neuradar as nm
    from nerfstudio.data.scene_box import SceneBox

    class _NoVGG(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            return torch.zeros(())

    nm.VGGPerceptualLossPix2Pix = _NoVGG
    cfg = nm.NeuRadarModelConfig()
    cfg.implementation = "torch"
    cfg.loss.radar_loss_type = "euclidean"  # configs[2]'s "deterministic" head
    cfg.field.grid.static.log2_hashmap_size = 12
    cfg.field.grid.actor.log2_hashmap_size = 10
    for s in (cfg.sampling.proposal_field_1, cfg.sampling.proposal_field_2):
        s.grid.static.log2_hashmap_size = 12
        s.grid.actor.log2_hashmap_size = 10
    torch.manual_seed(seed)
    model = cfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-100.0, -100, -10], [100, 100, 30]])), num_train_data=10,
                      metadata={"duration": 20.0, "sensor_idx_to_name": {0: "cam", 1: "lidar", 2: "radar"}, "trajectories": []})
    return nm, model


def golden_model():
    """Rows a19, a20, f-2 and the lidar losses, pinned by calling the reference NeuRadarModel's OWN methods:
    _get_appearance_embedding (neuradar.py:550-568), _compute_is_close_to_lidar (:971-994), decode_features (:410-493:
    lidar MLP, RGB CNN, radar transformer + heads), get_metrics_dict / get_loss_dict (:588-704: lidar losses, radar
    loss incl. the Hungarian association), sample_radar_points + chamfer_distance (radar_utils.py:170-229,380-420)."""
    from nerfstudio.model_components import radar_utils as ru

    nm, model = _build_reference_model()
    g = torch.Generator().manual_seed(5)
    out = {}
    # ---- a19: temporal appearance embedding
    B = 64
    times = torch.rand(B, 1, generator=g) * 20.0
    times[0], times[1] = 0.0, 20.0  # clamp edges
    sensor = torch.randint(0, 3, (B, 1), generator=g)
    bundle = RayBundle(origins=torch.zeros(B, 3), directions=torch.zeros(B, 3), pixel_area=torch.ones(B, 1), times=times,
                       metadata={"sensor_idxs": sensor})
    out.update(app_times=times, app_sensor=sensor, app_table=model.appearance_embedding.weight,
               app_embed=model._get_appearance_embedding(bundle, torch.zeros(B, 32)))
    # ---- a20: is_close_to_lidar on ray samples of a mixed batch
    S = 12
    edges = torch.sort(torch.rand(B, S + 1, generator=g) * 200.0, dim=-1).values
    is_lidar = torch.rand(B, 1, generator=g) < 0.5
    did_return = torch.rand(B, 1, generator=g) < 0.8
    dist = torch.rand(B, 1, generator=g) * 180.0
    mid = (edges[:, :-1] + edges[:, 1:]) / 2
    dist[is_lidar[:, 0]][:5] = mid[is_lidar[:, 0]][:5, 3:4]  # some samples exactly on the measured range
    rb = RayBundle(origins=torch.zeros(B, 3), directions=torch.ones(B, 3), pixel_area=torch.ones(B, 1), times=times,
                   metadata={"is_lidar": is_lidar, "did_return": did_return, "directions_norm": dist})
    rs = rb.get_ray_samples(bin_starts=edges[:, :-1, None], bin_ends=edges[:, 1:, None])
    model._compute_is_close_to_lidar(rs)
    out.update(close_edges=edges, close_is_lidar=is_lidar, close_did_return=did_return, close_dist=dist,
               close_mask=rs.metadata["is_close_to_lidar"])
    # ---- f-2: decode_features of a mixed batch (1 camera patch of 8x8, 40 lidar rays, 2 radar scans of 90 rays), eval mode
    model.eval()
    n_cam, n_lid, n_scan, nr = 64, 40, 2, 90
    n = n_cam + n_lid + n_scan * nr
    feats = torch.randn(n, 48, generator=g) * 0.5
    is_l = torch.zeros(n, 1, dtype=torch.bool)
    is_l[n_cam:n_cam + n_lid] = True
    is_r = torch.zeros(n, 1, dtype=torch.bool)
    is_r[n_cam + n_lid:] = True
    depth = torch.rand(n, 1, generator=g) * 80.0 + 1.0
    spher = torch.stack([torch.rand(n, generator=g) * 1.6 - 0.8, torch.rand(n, generator=g) * 0.48 - 0.08], dim=-1)
    params = {k: v for k, v in model.named_parameters() if k.split(".")[0] in (
        "rgb_decoder", "lidar_decoder", "radar_decoder", "offset_head", "radar_angle_head", "radar_uncertainty_head",
        "existence_probability_head")}
    buffers = {k: v for k, v in model.named_buffers() if k.startswith("rgb_decoder")}
    rgb, intensity, drop, radar_output = model.decode_features(feats, (8, 8), depth, spher, is_lidar=is_l, is_radar=is_r,
                                                               num_radar_scans=n_scan)
    out.update(dec_features=feats, dec_is_lidar=is_l, dec_is_radar=is_r, dec_depth=depth, dec_spher=spher, dec_rgb=rgb,
               dec_intensity=intensity, dec_ray_drop_logit=drop, dec_radar_output=radar_output)
    g_ro = torch.randn(radar_output.shape, generator=g)
    keys = ["radar_decoder.encoder.layers.0.self_attn.in_proj_weight", "radar_decoder.encoder.layers.0.linear1.weight",
            "radar_decoder.encoder.norm.weight", "offset_head.layers.0.weight", "existence_probability_head.layers.2.bias"]
    named = dict(model.named_parameters())
    grads = torch.autograd.grad((radar_output * g_ro).sum(), [named[k] for k in keys])
    out["dec_g_radar_output"] = g_ro
    for k, gr in zip(keys, grads):
        out["dec_grad." + k] = gr
    for k, v in {**params, **buffers}.items():
        out["param." + k] = v
    # ---- radar loss (Hungarian association, euclidean) + sampled points + Chamfer against synthetic detections
    n_det = 25
    gt_pts = torch.cat([torch.randn(n_det, 3, generator=g) * 20.0 + torch.tensor([30.0, 0.0, 0.0]), torch.rand(n_det, 6, generator=g)], dim=1)
    gt2 = torch.cat([torch.randn(n_det + 7, 3, generator=g) * 20.0 + torch.tensor([30.0, 0.0, 0.0]), torch.rand(n_det + 7, 6, generator=g)], dim=1)
    radar_batch = torch.cat([gt_pts, gt2])
    indices = torch.cat([torch.stack([torch.zeros(n_det), torch.arange(n_det)], 1), torch.stack([torch.ones(n_det + 7), torch.arange(n_det + 7)], 1)]).long()
    loss, assoc, _ = ru.calculate_radar_loss(radar_batch, radar_output.detach(), indices, loss_type="euclidean", training=True)
    ro = radar_output.detach().clone()
    ro[..., 0] = torch.rand(ro.shape[:-1], generator=g)  # spread existence probabilities around the 0.5 threshold
    pts, ber = ru.sample_radar_points(ro, loss_type="euclidean", threshold=0.5)
    cd = ru.chamfer_distance(pts[:, :3].numpy(), gt2[:, :3].numpy())
    out.update(radar_batch=radar_batch, radar_indices=indices, radar_loss=loss, radar_assoc_last=assoc, cd_radar_output=ro,
               cd_points=pts, cd_ber=ber, cd_gt=gt2[:, :3], cd_value=float(cd))
    # ---- lidar losses through get_metrics_dict / get_loss_dict (training branch)
    model.train()
    nl = 200
    is_lb = torch.zeros(nl + 30, 1, dtype=torch.bool)
    is_lb[:nl] = True
    did = torch.rand(nl + 30, 1, generator=g) < 0.85
    lidar_pts = torch.cat([torch.randn(nl, 3, generator=g), torch.rand(nl, 1, generator=g), torch.rand(nl, 1, generator=g) * 0.1], dim=1)
    dist_l = torch.rand(nl, 1, generator=g) * 100.0 + 2.0
    outputs = {"depth": torch.rand(nl + 30, 1, generator=g) * 100.0, "ray_drop_logits": torch.randn(nl, 1, generator=g),
               "intensity": torch.rand(nl, 1, generator=g), "non_nearby_weights": torch.rand(500, generator=g),
               "prop_depth_0": torch.rand(nl + 30, 1, generator=g) * 100.0, "prop_depth_1": torch.rand(nl + 30, 1, generator=g) * 100.0,
               "prop_weights_loss_0": torch.tensor(3.0), "prop_weights_loss_1": torch.tensor(1.5)}
    batch = {"lidar": lidar_pts, "is_lidar": is_lb, "did_return": did, "distance": dist_l}
    metrics, _ = model.get_metrics_dict(dict(outputs), dict(batch))
    losses = model.get_loss_dict(dict(outputs), dict(batch), metrics)
    out.update(ll_depth=outputs["depth"], ll_ray_drop_logits=outputs["ray_drop_logits"], ll_intensity=outputs["intensity"],
               ll_non_nearby=outputs["non_nearby_weights"], ll_prop_depth_0=outputs["prop_depth_0"], ll_prop_depth_1=outputs["prop_depth_1"],
               ll_is_lidar=is_lb, ll_did_return=did, ll_points=lidar_pts, ll_distance=dist_l)
    for k in ("depth_loss", "intensity_loss", "ray_drop_loss", "carving_loss", "depth_loss_0", "depth_loss_1", "carving_loss_0", "carving_loss_1"):
        out["ll_metric." + k] = metrics[k]
        if k in losses:  # the proposal-level entries need "weights_list" in outputs (:679-688)
            out["ll_loss." + k] = losses[k]
    save("model", **out)


def golden_model_train():
    """The TRAINING branch behind the rendered features, from the reference NeuRadarModel's own methods (dropout set to 0
    on the instance -- the masks are torch-RNG draws -- batch norm in training mode): decode_features (neuradar.py:
    410-493) on a mixed batch (2 camera patches 8x8, 60 lidar rays, 2 radar scans of 90 rays) -> get_metrics_dict /
    get_loss_dict (:588-704) with radar_loss_type "nll" (the reference's default, :114) and "euclidean": every loss term,
    the Hungarian associations, d(total loss)/d(features, depth) and parameter gradients; the evaluation-side "nll" cost
    matrix (radar_utils.py:105-118) and the nll branch of sample_radar_points (:181-213) under a fixed torch seed.
    Parameters are those of model.npz (same construction seed); only the inputs and results are stored here."""
    from nerfstudio.model_components import radar_utils as ru

    nm, model = _build_reference_model()
    g = torch.Generator().manual_seed(17)
    model.train()
    for m in model.radar_decoder.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    n_patch, ps, n_lid, n_scan, nr = 2, 8, 60, 2, 90
    n_cam = n_patch * ps * ps
    n = n_cam + n_lid + n_scan * nr
    feats = (torch.randn(n, 48, generator=g) * 0.5).requires_grad_(True)
    is_l = torch.zeros(n, 1, dtype=torch.bool)
    is_l[n_cam:n_cam + n_lid] = True
    is_r = torch.zeros(n, 1, dtype=torch.bool)
    is_r[n_cam + n_lid:] = True
    depth = (torch.rand(n, 1, generator=g) * 80.0 + 1.0).requires_grad_(True)
    spher = torch.stack([torch.rand(n, generator=g) * 1.6 - 0.8, torch.rand(n, generator=g) * 0.48 - 0.08], dim=-1)
    prop_depth = [(torch.rand(n, 1, generator=g) * 90.0).requires_grad_(True) for _ in range(2)]
    image = torch.rand(n_patch, ps * 3, ps * 3, 3, generator=g)
    did = torch.rand(n, 1, generator=g) < 0.85
    lidar_pts = torch.cat([torch.randn(n_lid, 3, generator=g), torch.rand(n_lid, 1, generator=g), torch.rand(n_lid, 1, generator=g) * 0.1], dim=1)
    dist_l = torch.rand(n_lid, 1, generator=g) * 100.0 + 2.0
    dets = [27, 1]  # detections per scan (one scan with a single detection: the squeeze(0) quirk of radar_utils.py:142-144)
    radar_batch = torch.cat([torch.cat([torch.randn(m, 3, generator=g) * 15.0 + torch.tensor([30.0, 0.0, 0.0]), torch.rand(m, 6, generator=g)], 1)
                             for m in dets])
    indices = torch.cat([torch.stack([torch.full((m,), float(i)), torch.arange(m).float()], 1) for i, m in enumerate(dets)]).long()
    out = dict(features=feats, depth=depth, spher=spher, is_lidar=is_l, is_radar=is_r, prop_depth_0=prop_depth[0],
               prop_depth_1=prop_depth[1], image=image, did_return=did, lidar=lidar_pts, distance=dist_l, radar=radar_batch,
               radar_indices=indices, n_patch=n_patch, patch=ps, n_scan=n_scan)
    named = dict(model.named_parameters())
    dec = [k for k in named if k.split(".")[0] in ("rgb_decoder", "lidar_decoder", "radar_decoder", "offset_head",
                                                    "radar_uncertainty_head", "existence_probability_head")]
    for k in dec:  # fingerprint: the parameters are model.npz's
        out["param_sum." + k] = named[k].double().sum()
    for loss_type in ("nll", "euclidean"):
        model.config.loss.radar_loss_type = loss_type
        rgb, intensity, drop, radar_output = model.decode_features(feats, (ps, ps), depth, spher, is_lidar=is_l, is_radar=is_r,
                                                                   num_radar_scans=n_scan)
        outputs = {"rgb": rgb, "intensity": intensity.float(), "ray_drop_logits": drop.float(), "radar_output": radar_output,
                   "depth": depth, "prop_depth_0": prop_depth[0], "prop_depth_1": prop_depth[1],
                   "non_nearby_weights": torch.zeros(1), "prop_weights_loss_0": torch.tensor(0.0), "prop_weights_loss_1": torch.tensor(0.0)}
        batch = {"image": image, "lidar": lidar_pts, "is_lidar": is_l, "did_return": did, "distance": dist_l,
                 "radar": radar_batch.clone(), "radar_indices": indices}
        metrics, _ = model.get_metrics_dict(dict(outputs), dict(batch))
        losses = model.get_loss_dict(dict(outputs), dict(batch), metrics)
        conf = model.config.loss
        # the proposal-level lidar depth terms enter get_loss_dict only next to "weights_list" (:679-688): added here with the
        # same multipliers so that their gradient is part of the vector
        for i in range(2):
            losses[f"depth_loss_{i}"] = conf.prop_lidar_loss_mult * conf.depth_mult * metrics[f"depth_loss_{i}"]
        keep = ["rgb_loss", "depth_loss", "intensity_loss", "ray_drop_loss", "radar_loss", "depth_loss_0", "depth_loss_1"]
        total = sum(losses[k] for k in keep)
        wrt = [feats, depth, prop_depth[0], prop_depth[1]] + [named[k] for k in dec]
        grads = torch.autograd.grad(total, wrt, allow_unused=True)
        t = loss_type + "."
        out.update({t + "rgb": rgb, t + "intensity": intensity, t + "ray_drop_logits": drop, t + "radar_output": radar_output,
                    t + "total": total, t + "g_features": grads[0], t + "g_depth": grads[1], t + "g_prop_depth_0": grads[2],
                    t + "g_prop_depth_1": grads[3]})
        for k in keep:
            out[t + "loss." + k] = losses[k]
        for k, gr in zip(dec, grads[4:]):
            if gr is None:
                continue
            out[t + "gsum." + k] = gr.double().sum()
            out[t + "gabs." + k] = gr.double().abs().sum()
            if gr.numel() <= 4096:
                out[t + "grad." + k] = gr
        # per-scan associations (calculate_radar_loss returns the last one only): the same calls, scan by scan
        seg = [0, dets[0], dets[0] + dets[1]]
        for i in range(n_scan):
            gt = radar_batch[seg[i]:seg[i + 1], :3]
            mb = ru.MultiBernoulli(prediction=radar_output[i].detach())
            c = ru.get_cost_matrix(gt, mb, "euclidean")  # training: always the euclidean cost (radar_utils.py:77-78)
            row, col = ru.linear_sum_assignment(c.numpy())
            assoc = -torch.ones(mb.n_mb, dtype=torch.long)
            assoc[torch.as_tensor(row)] = torch.as_tensor(col)
            out[t + f"cost_{i}"] = c
            out[t + f"assoc_{i}"] = assoc
            out[t + f"scan_loss_{i}"] = ru.get_radar_loss(gt, mb, torch.stack([torch.arange(mb.n_mb).float(), assoc.float()], 1), loss_type)
            if loss_type == "nll" and gt.shape[0] > 1:  # (a single detection breaks the reference's own broadcast at :118)
                out[f"nll.evalcost_{i}"] = ru.get_cost_matrix(gt, mb, "nll")
    # ---- sample_radar_points, "nll" branch: Bernoulli draw per prediction, Laplace rsample per kept coordinate
    ro = radar_output.detach().clone()
    ro[..., 0] = torch.rand(ro.shape[:-1], generator=g)
    torch.manual_seed(2024)
    pts, ber = ru.sample_radar_points(ro, loss_type="nll", max_detections=50)
    out.update(sample_radar_output=ro, sample_seed=2024, sample_max_detections=50, sample_points=pts, sample_ber=ber)
    save("model_train", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["hash", "gaussian_contraction", "field", "field_autocast", "field_density", "sh_mlp", "sampler", "pipeline", "losses", "raygen", "actors", "model", "model_train"]
    for w in which:
        globals()["golden_" + w]()
