"""Worker of tests/test_boundary_contract.py (own process: it puts the reference and its import shim on sys.path).

Boundary contract against the reference's OWN caller (round-1 verdict, item 9; INTEGRATION.md Option A).

In the build container the reference imports (tests/golden/ref_shim.py).  Its `NeuRadarModel` is built twice: once as it
is (implementation="torch"), once with the neuradar_amd drop-in classes substituted for the sampler, the field and the
proposal fields -- and `get_nff_outputs` (models/neuradar.py:495-548, incl. `_get_ray_samples` :570-586 with its in-place
edits of the returned samples and `_compute_is_close_to_lidar` :971-994) is run through both on the same bundle and the same
random numbers.  Every attribute the reference touches on the sampler, the fields and `RaySamples` must resolve, and the
tensors it produces must agree.  No GPU here: `neuradar_amd.ops` is replaced by a tests-only shim that maps each op to the
CPU oracle (same signatures and buffer layouts), so what is exercised is the HOST surface of the drop-in.
Skipped where /root/reference is absent (the GPU box)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _install_oracle_ops():
    """neuradar_amd.ops -> oracle functions with the ops' signatures (tests only)."""
    from neuradar_amd import ops
    from oracle import field as of, hashgrid, sampler as osamp

    def rows(t, B, S, sm):  # [B,S,...] -> row order of the ABI: the first sm rays sample-major
        if not sm:
            return t.reshape(B * S, *t.shape[2:])
        assert sm == B
        return t.transpose(0, 1).reshape(B * S, *t.shape[2:])

    def unrows(t, B, S, sm):  # inverse, -> [B,S,...]
        return t.view(B, S, *t.shape[1:]) if not sm else t.view(S, B, *t.shape[1:]).transpose(0, 1)

    def contract_gaussians(origins, directions, pixel_area, euclid, scale, sample_major_rows=False):
        B, S = euclid.shape[0], euclid.shape[1] - 1
        mean, std = of.isotropic_gaussian(origins, directions, euclid[:, :-1], euclid[:, 1:], pixel_area.reshape(B, 1))
        x01, s01 = of.scaled_contraction(mean, std, scale)
        sm = B if sample_major_rows is True else int(sample_major_rows)
        return rows(x01, B, S, sm).contiguous(), rows(s01[..., 0], B, S, sm).contiguous()

    def hash_encode(x, table, scalings, log2_hashmap_size, std=None, level_major=False, sample_major=0):
        L, F = scalings.numel(), table.shape[1]
        raw = hashgrid.encode(x, table, scalings, 2**log2_hashmap_size)  # [n, L*F]
        if std is not None:
            w = 1 / (scalings[None, :] * 2 * std[:, None]).clamp_min(1.0)
            raw = (raw.view(-1, L, F) * w[..., None]).reshape(-1, L * F)
        return raw.view(-1, L, F).permute(1, 0, 2).contiguous() if level_major else raw

    def feats_of(buf, strides, F, n):
        return buf.permute(1, 0, 2).reshape(n, -1) if strides[0] == F else buf  # level-major [L,n,F] or [n, L*F]

    def field_mlp(feats, strides, feat_f, directions, n_samples, n, geo, feat, beta, rows_sample_major=False, dtype="float32",
                  grad_scale=1.0, sample_dirs=None):
        x = feats_of(feats, strides, feat_f, n)
        B = n // n_samples
        sm = B if rows_sample_major else 0
        x = unrows(x, B, n_samples, sm).reshape(n, -1)
        d = sample_dirs if sample_dirs is not None else directions[:, None, :].expand(B, n_samples, 3).reshape(n, 3)
        h = of.mlp(x, list(zip(*geo)))
        sdf, e = h[:, 0], h[:, 1:]
        feature = e + of.mlp(torch.cat([e, of.direction_encoding(d)], -1), list(zip(*feat)))
        return feature, sdf, of.sigmoid_density(sdf, beta)

    def prop_density(feats, strides, feat_f, w, n, n_samples=0, rows_sample_major=False):
        x = feats_of(feats, strides, feat_f, n)
        if n_samples:
            B = n // n_samples
            x = unrows(x, B, n_samples, B if rows_sample_major else 0).reshape(n, -1)
        return of.trunc_exp(torch.nn.functional.linear(x, w))[:, 0]

    def power_bins(nears, fars, n_samples, t_rand=None, lam=ops.POWER_LAMBDA, scaling=ops.POWER_SCALING):
        s = osamp.power_bins(nears.reshape(-1, 1), fars.reshape(-1, 1), n_samples, t_rand, lam, scaling)
        return s.spacing, s.euclid

    def pdf_resample(weights, spacing_in, nears, fars, n_out, jitter=None, lam=ops.POWER_LAMBDA, scaling=ops.POWER_SCALING,
                     sky_distance=0.0):
        s_near, s_far = osamp.power_fn(nears.reshape(-1, 1) * scaling, lam), osamp.power_fn(fars.reshape(-1, 1) * scaling, lam)
        prev = osamp.Samples(spacing_in, None, s_near, s_far)
        s = osamp.pdf_resample(prev, weights, n_out, None if jitter is None else jitter.reshape(-1, 1), lam=lam, scaling=scaling)
        return s.spacing, s.euclid

    def weights_from_density(density, euclid):
        return osamp.weights_from_density(euclid[:, 1:] - euclid[:, :-1], density)

    from oracle import render as orender

    for name, fn in dict(contract_gaussians=contract_gaussians, hash_encode=hash_encode, field_mlp=field_mlp, prop_density=prop_density,
                         power_bins=power_bins, pdf_resample=pdf_resample, weights_from_density=weights_from_density,
                         render_weights=orender.render_weight_from_alpha, render_weights_from_density=orender.render_weight_from_density,
                         accumulate_along_rays=orender.accumulate_along_rays).items():
        setattr(ops, name, fn)


def main():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import ref_shim

    ref_shim.install()
    import copy

    import nerfstudio.models.neuradar as nm
    from nerfstudio.cameras.rays import RayBundle as RefBundle
    from nerfstudio.data.scene_box import SceneBox
    from nerfstudio.field_components.field_heads import FieldHeadNames as RefHeads

    import neuradar_amd.field_heads as fh
    from neuradar_amd.neurad_encoding import ActorSettings, NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig, NeuRADProposalFieldConfig
    from neuradar_amd.ray_samplers import PowerSampler, ProposalNetworkSampler

    assert fh.FieldHeadNames is RefHeads, "inside a nerfstudio installation the drop-in must use the reference's enum"
    _install_oracle_ops()

    class _NoVGG(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    nm.VGGPerceptualLossPix2Pix = _NoVGG
    cfg = nm.NeuRadarModelConfig()
    cfg.implementation = "torch"
    cfg.field.grid.static.log2_hashmap_size = 12
    for s in (cfg.sampling.proposal_field_1, cfg.sampling.proposal_field_2):
        s.grid.static.log2_hashmap_size = 12
    torch.manual_seed(0)
    ref = cfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-100.0, -100, -10], [100, 100, 30]])), num_train_data=10,
                    metadata={"duration": 20.0, "sensor_idx_to_name": {0: "cam", 1: "lidar"}, "trajectories": []})
    with torch.no_grad():
        ref.field.hashgrid.static_grid.hash_table.mul_(300.0)
        for pf in ref.proposal_fields:
            pf.hashgrid.static_grid.hash_table.mul_(1500.0)
    # ---- the drop-in classes with the reference's parameters (same state_dict keys)
    mine = copy.copy(ref)
    mine._modules = dict(ref._modules)
    field = NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=12), actor=ActorSettings(flip_prob=0.25))
                              ).setup(actors=None, static_scale=float(ref.scene_box.aabb.max()), implementation="hip")
    missing = field.load_state_dict({k: v for k, v in ref.field.state_dict().items() if not k.startswith("hashgrid.actors")}, strict=False)
    assert not missing.missing_keys, missing
    props = []
    for pf in ref.proposal_fields:
        pc = NeuRADProposalFieldConfig()
        pc.grid.static.log2_hashmap_size = 12
        p = pc.setup(actors=None, static_scale=float(ref.scene_box.aabb.max()), implementation="hip")
        assert not p.load_state_dict({k: v for k, v in pf.state_dict().items() if not k.startswith("hashgrid.actors")}, strict=False).missing_keys
        props.append(p)
    mine._modules["field"] = field
    mine._modules["proposal_fields"] = torch.nn.ModuleList(props)
    mine.density_fns = [lambda x: prop_field.get_density(x)[0] for prop_field in props]  # the reference's own construction (:302)
    mine._modules["sampler"] = ProposalNetworkSampler(
        num_proposal_samples_per_ray=cfg.sampling.num_proposal_samples, num_nerf_samples_per_ray=cfg.sampling.num_nerf_samples,
        num_proposal_network_iterations=cfg.num_proposal_rounds, single_jitter=cfg.sampling.single_jitter,
        update_sched=lambda x: 0, initial_sampler=PowerSampler(lambda_=cfg.sampling.power_lambda, scaling=cfg.sampling.power_scaling))

    def bundle():
        g = torch.Generator().manual_seed(4)
        B = 48
        o = torch.cat([torch.randn(B, 2, generator=g) * 3, torch.full((B, 1), 1.7)], 1)
        d = torch.nn.functional.normalize(torch.cat([torch.ones(B, 1), 0.5 * torch.randn(B, 2, generator=g)], 1), dim=-1)
        is_l = torch.zeros(B, 1, dtype=torch.bool)
        is_l[24:] = True
        return RefBundle(origins=o, directions=d, pixel_area=torch.full((B, 1), 2.5e-7), fars=torch.full((B, 1), 1e6),
                         times=torch.rand(B, 1, generator=g) * 20, camera_indices=torch.zeros(B, 1, dtype=torch.long),
                         metadata={"is_lidar": is_l, "did_return": torch.rand(B, 1, generator=g) < 0.8,
                                   "directions_norm": 2 + 60 * torch.rand(B, 1, generator=g),
                                   "sensor_idxs": is_l.long()})

    def depth_simple(weights, ray_samples):  # nerfstudio.models.neurad is not part of the checkout: NeuRAD's
        return (weights * (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2).sum(-2)  # expected-depth renderer, restated

    outs = []
    for m in (ref, mine):
        m.renderer_depth = depth_simple
        m.train()
        torch.manual_seed(123)  # the samplers draw their jitter inside: same generator state for both models
        outs.append(m.get_nff_outputs(bundle(), calc_lidar_losses=True))
    a, b = outs
    # the CPU branch of the reference's _render_weights is a constant 0.5 (:1012-1014) for BOTH models; what differs between
    # the two runs is everything upstream of it: samples, field outputs, masks
    for k in ("features", "accumulation", "depth", "prop_depth_0", "prop_depth_1", "prop_weights_loss_0", "prop_weights_loss_1", "non_nearby_weights"):
        torch.testing.assert_close(b[k], a[k], rtol=1e-4, atol=1e-5, msg=lambda s, k=k: f"{k}: {s}")
    for i in range(3):
        torch.testing.assert_close(b["weights_list"][i], a["weights_list"][i], rtol=1e-4, atol=1e-6)
        ra, rb = a["ray_samples_list"][i], b["ray_samples_list"][i]
        torch.testing.assert_close(rb.frustums.starts, ra.frustums.starts, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(rb.frustums.ends, ra.frustums.ends, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(rb.spacing_starts, ra.spacing_starts, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rb.deltas, ra.deltas, rtol=1e-4, atol=1e-4)
        assert torch.equal(rb.metadata["is_close_to_lidar"], ra.metadata["is_close_to_lidar"])
    # the losses the reference computes from these outputs accept the drop-in's RaySamples (losses.py:107-112,137-157,645-705;
    # get_metrics_dict :588-668, get_loss_dict :670-705) -- a lidar batch like the datamanager's, decoder outputs as in
    # decode_features :432-452
    g = torch.Generator().manual_seed(9)
    bm = bundle().metadata
    n_l = int(bm["is_lidar"].sum())
    batch = {"lidar": torch.rand(n_l, 4, generator=g), "is_lidar": bm["is_lidar"], "did_return": bm["did_return"],
             "distance": bm["directions_norm"][bm["is_lidar"][:, 0]]}
    losses = []
    for m, o in ((ref, dict(a)), (mine, dict(b))):
        inten, drop = m.lidar_decoder(o["features"][bm["is_lidar"][:, 0]]).split(1, dim=-1)
        o["intensity"], o["ray_drop_logits"] = inten.sigmoid(), drop
        md, _ = m.get_metrics_dict(o, dict(batch))
        losses.append(m.get_loss_dict(o, dict(batch), md))
    la, lb = losses
    assert set(la) == set(lb) and {"interlevel_loss", "distortion_loss", "carving_loss", "carving_loss_0", "depth_loss_1"} <= set(la)
    for k in la:
        torch.testing.assert_close(lb[k], la[k], rtol=2e-4, atol=1e-8, msg=lambda s, k=k: f"{k}: {s}")
    # the nerfacc call forms of _render_weights (:1016,1018-1022), render_depth_simple (neurad.py:727-728) and the renderer
    # modules (renderers.py:59-90,322-350) resolve on neuradar_amd.renderers (host surface; values are the GPU tests')
    from neuradar_amd import renderers as rr

    rs = a["ray_samples_list"][2]
    alpha = torch.rand(rs.frustums.starts.shape[:2], generator=g)
    w, T = rr.render_weight_from_alpha(alpha)
    assert w.shape == alpha.shape and float(T[:, 0].min()) == 1.0
    w2, T2, al = rr.render_weight_from_density(t_ends=rs.frustums.ends.squeeze(-1), t_starts=rs.frustums.starts.squeeze(-1), sigmas=alpha)
    assert w2.shape == T2.shape == al.shape == alpha.shape
    steps = (rs.frustums.starts + rs.frustums.ends) / 2
    d1 = rr.accumulate_along_rays(w, steps, None, None)
    torch.testing.assert_close(d1, depth_simple(w[..., None], rs))
    feats = torch.randn(*alpha.shape, 5, generator=g)
    torch.testing.assert_close(rr.FeatureRenderer()(feats, w[..., None]), nm.FeatureRenderer()(feats, w[..., None]))
    torch.testing.assert_close(rr.AccumulationRenderer()(w[..., None]), nm.AccumulationRenderer()(w[..., None]))
    print("boundary contract: OK --", len(la), "loss terms,", len(a), "outputs compared")


if __name__ == "__main__":
    main()
