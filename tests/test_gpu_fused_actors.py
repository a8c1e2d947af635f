"""The fused training step with dynamic actors, the appearance embedding and the lidar carving masks inside it
(round-1 verdict, item 6) against the modular autograd path, which tests/test_gpu_parity.py pins to the reference."""
import pytest
import torch

from helpers import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model_with_actors(g):
    from neuradar_amd.dynamic_actors import DynamicActors
    from neuradar_amd.neurad_encoding import ActorSettings, NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.step import HotPathConfig, NeuRadarHotPath

    actors = DynamicActors.from_state(g["actor_positions"], g["actor_rotations_6d"], g["actor_timestamps"], g["actor_present"],
                                      g["actor_sizes"])
    cfg = HotPathConfig(field=NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(
        static=StaticSettings(log2_hashmap_size=14), actor=ActorSettings(flip_prob=0.25, log2_hashmap_size=10))))
    for pc in (cfg.proposal_field_1, cfg.proposal_field_2):
        pc.grid.static.log2_hashmap_size = 14
        pc.grid.actor.log2_hashmap_size = 10
    torch.manual_seed(3)
    model = NeuRadarHotPath(cfg, actors=actors).to(DEV).train()
    with torch.no_grad():  # features of O(0.1..1): actor and static contributions both matter
        for f in (model.field, model.proposal_fields[1]):
            f.hashgrid.static_grid.hash_table.mul_(300.0)
            for gr in f.hashgrid.actor_grids:
                gr.hash_table.mul_(600.0)
    return model, actors


@pytest.mark.parametrize("coherent,permuted", [(None, False), (0.5, False), (0.5, True)])
def test_fused_step_with_two_actors_matches_autograd_path(coherent, permuted):
    from neuradar_amd.fused_step import FusedTrainStep
    from neuradar_amd.rays import RayBundle

    g = load_golden("actors")
    model, actors = _model_with_actors(g)
    if permuted:  # actor a's hash grid is actor_grids[actor_to_id[a]] (neurad_encoding.py:183) in the fused launches too; the
        with torch.no_grad():  # modular path with a permuted map is pinned to the reference golden in test_gpu_parity.py
            actors.actor_to_id.copy_(torch.tensor([1, 0]))
    gen = torch.Generator().manual_seed(11)
    B = 96
    # rays from around the origin towards the actors' trajectories, times inside the trajectories' span
    pos = g["actor_positions"].reshape(-1, 3)
    tgt = pos[torch.randint(0, pos.shape[0], (B,), generator=gen)] + 0.8 * torch.randn(B, 3, generator=gen)
    o = torch.cat([torch.randn(B, 2, generator=gen) * 2.0, torch.full((B, 1), 1.5)], dim=1)
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    ts = g["actor_timestamps"]
    times = ts.min() + (ts.max() - ts.min()) * torch.rand(B, generator=gen)
    area = torch.full((B,), 2.25e-6)
    fars = torch.full((B,), 1e6)
    t_rand, j1, j2 = torch.rand(B, 129, generator=gen), torch.rand(B, generator=gen), torch.rand(B, generator=gen)
    tf, td = 0.1 * torch.randn(B, 32, generator=gen), 5.0 + 20.0 * torch.rand(B, generator=gen)
    flips = [(torch.rand(B, generator=gen) < pr).float() * -2 + 1 for pr in (0.5, 0.5, 0.25)]
    dv = lambda x: x.to(DEV)  # noqa: E731
    bundle = RayBundle(dv(o), dv(d), dv(area)[:, None], fars=dv(fars)[:, None].clone(), times=dv(times)[:, None])
    out = model.get_nff_outputs(bundle, t_rand=dv(t_rand), jitters=(dv(j1)[:, None], dv(j2)[:, None]), flips=[dv(f) for f in flips])
    loss = model.bench_loss(out, dv(tf), dv(td)[:, None])
    params = {n: p for n, p in model.named_parameters() if p.requires_grad}
    ref = dict(zip(params, torch.autograd.grad(loss, list(params.values()), allow_unused=True)))
    inside = sum(int((s >= 0).sum()) for s in ())  # (filled below from the fused step's own assignment buffers)
    for p in params.values():
        if p.grad is not None:
            p.grad.zero_()
    fused = FusedTrainStep(model, B, coherent_rays=None if coherent is None else int(coherent * B))
    floss = fused.forward_backward(dv(o), dv(d), dv(area), dv(fars), dv(tf), dv(td), dv(t_rand), dv(j1), dv(j2), times=dv(times),
                                   flips=[dv(f) for f in flips])
    inside = [int((s >= 0).sum()) for s in fused.a_slot]
    assert min(inside) > 20, f"the test batch must put samples inside actor boxes at every level: {inside}"
    assert int(model.field.hashgrid.actor_overflow) == 0
    fo = fused.outputs()
    assert_close(fo["features"].cpu(), out["features"].detach().cpu(), rtol=1e-4, atol_scale=1e-5, what="features")
    assert_close(fo["depth"].cpu(), out["depth"].detach().cpu(), rtol=1e-4, atol_scale=1e-5, what="depth")
    assert_close(fo["prop_weights_1"].cpu(), out["weights_list"][1][..., 0].detach().cpu(), rtol=1e-4, atol_scale=1e-5, what="prop weights")
    assert_close(floss.sum().cpu(), loss.detach().cpu(), rtol=1e-4, atol_scale=1e-6, what="loss")
    checked = 0
    for n, p in params.items():
        if ref[n] is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        assert_close(p.grad.cpu(), ref[n].cpu(), rtol=1e-3, atol_scale=1e-4, what="fused grad " + n)
        checked += 1
    for must in ("dynamic_actors.actor_positions", "dynamic_actors.actor_rotations_6d", "field.hashgrid.actor_grids.0.hash_table",
                 "proposal_fields.1.hashgrid.actor_grids.1.hash_table"):
        assert ref[must] is not None and float(ref[must].abs().max()) > 0, must


def test_fused_step_with_lidar_masks_appearance_and_lidar_decoder_matches_autograd_path():
    """a19 + a20 inside the fused step: carving masks on all three levels, appearance embedding + lidar decoder + its two
    losses on the lidar rows -- against the same loss assembled from the modular path's outputs (step.py:
    get_nff_outputs' is_close_to_lidar side outputs, decode_lidar), every parameter gradient included."""
    from neuradar_amd.fused_step import FusedTrainStep
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.rays import RayBundle
    from neuradar_amd.step import HotPathConfig, NeuRadarHotPath

    cfg = HotPathConfig(field=NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=14))),
                        appearance_dim=16, num_sensors=3, lidar_decoder=True)
    for pc in (cfg.proposal_field_1, cfg.proposal_field_2):
        pc.grid.static.log2_hashmap_size = 14
    torch.manual_seed(5)
    model = NeuRadarHotPath(cfg).to(DEV).train()
    with torch.no_grad():
        model.field.hashgrid.static_grid.hash_table.mul_(300.0)
        model.proposal_fields[1].hashgrid.static_grid.hash_table.mul_(1500.0)
    gen = torch.Generator().manual_seed(2)
    n_cam, n_lid = 64, 80
    B = n_cam + n_lid
    o = torch.cat([torch.randn(B, 2, generator=gen) * 3.0, torch.full((B, 1), 1.7)], dim=1)
    d = torch.nn.functional.normalize(torch.cat([torch.ones(B, 1), 0.5 * torch.randn(B, 2, generator=gen)], dim=1), dim=-1)
    area = torch.cat([torch.full((n_cam,), 2.25e-6), torch.full((n_lid,), 4.5e-6)])
    times = 20.0 * torch.rand(B, generator=gen)
    is_lidar = torch.zeros(B, dtype=torch.bool)
    is_lidar[n_cam:] = True
    did_return = torch.ones(B, dtype=torch.bool)
    did_return[n_cam:] = torch.rand(n_lid, generator=gen) < 0.8
    rng = torch.ones(B)
    rng[n_cam:] = 2.0 + 60.0 * torch.rand(n_lid, generator=gen)
    sensor = torch.cat([torch.zeros(n_cam), torch.ones(n_lid)]).long()
    target_i = torch.rand(B, generator=gen)
    t_rand, j1, j2 = torch.rand(B, 129, generator=gen), torch.rand(B, generator=gen), torch.rand(B, generator=gen)
    tf, td = 0.1 * torch.randn(B, 32, generator=gen), 5.0 + 50.0 * torch.rand(B, generator=gen)
    dv = lambda x: x.to(DEV)  # noqa: E731
    bundle = RayBundle(dv(o), dv(d), dv(area)[:, None], fars=torch.full((B, 1), 1e6, device=DEV), times=dv(times)[:, None],
                       metadata={"is_lidar": dv(is_lidar)[:, None], "did_return": dv(did_return)[:, None],
                                 "directions_norm": dv(rng)[:, None], "sensor_idxs": dv(sensor)[:, None]})
    out = model.get_nff_outputs(bundle, t_rand=dv(t_rand), jitters=(dv(j1)[:, None], dv(j2)[:, None]))
    c = model.config
    loss = model.bench_loss(out, dv(tf), dv(td)[:, None])
    loss = loss + c.carving_mult * (out["non_nearby_weights"] ** 2).sum() / n_lid
    for i in (0, 1):
        loss = loss + c.prop_lidar_loss_mult * c.carving_mult * out[f"prop_weights_loss_{i}"] / n_lid
    intensity, drop = model.decode_lidar(out["features"], dv(is_lidar)[:, None])
    ret = dv(did_return)[n_cam:]
    loss = loss + c.intensity_mult * ((intensity[ret, 0] - dv(target_i)[n_cam:][ret]) ** 2).mean()
    loss = loss + c.ray_drop_loss_mult * torch.nn.functional.binary_cross_entropy_with_logits(drop[:, 0], (~ret).float())
    params = {n: p for n, p in model.named_parameters() if p.requires_grad}
    ref = dict(zip(params, torch.autograd.grad(loss, list(params.values()), allow_unused=True)))
    for p in params.values():
        if p.grad is not None:
            p.grad.zero_()
    fused = FusedTrainStep(model, B, coherent_rays=n_cam)
    fused.set_lidar(dv(is_lidar).to(torch.uint8), dv(did_return).to(torch.uint8), dv(rng), n_cam, n_lid,
                    target_intensity=dv(target_i), sensor_idx=dv(sensor))
    floss = fused.forward_backward(dv(o), dv(d), dv(area), torch.full((B,), 1e6, device=DEV), dv(tf), dv(td), dv(t_rand), dv(j1),
                                   dv(j2), times=dv(times))
    assert_close(floss.sum().cpu(), loss.detach().cpu(), rtol=1e-4, atol_scale=1e-6, what="loss")
    assert_close(fused.outputs()["features"].cpu(), out["features"][:, :32].detach().cpu(), rtol=1e-4, atol_scale=1e-5, what="features")
    for n, p in params.items():
        if ref[n] is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        assert_close(p.grad.cpu(), ref[n].cpu(), rtol=1e-3, atol_scale=1e-4, what="fused grad " + n)
    for must in ("appearance_embedding.weight", "lidar_decoder.layers.0.weight", "lidar_decoder.layers.2.bias"):
        assert ref[must] is not None and float(ref[must].abs().max()) > 0, must
