"""The decoders and their losses inside the training step (BASELINE configs[2] "full", configs[3] "nll",
configs[4] transformer in the step):
  * DecoderLossHead (RGB CNN + MSE, lidar MLP + quantile-masked lidar losses, radar transformer + heads + Hungarian-matched
    radar loss, all on the device) against the vectors the reference NeuRadarModel's own training-branch methods produce
    (tests/golden/model_train.npz);
  * the lidar loss kernels against the oracle on ragged / degenerate inputs;
  * FusedTrainStep.set_decoders: loss and EVERY parameter gradient of the fused step against the modular HIP path for the
    field part + the CPU oracle (oracle/decoder_losses.py) for everything behind the rendered features."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _decoder_model(n_features):
    """The decoder attributes DecoderLossHead reads, without a field around them (the golden's features are 48 wide)."""
    from neuradar_amd.decoders import Decoders

    dec = Decoders(n_features=n_features).to(DEV).train()
    dec.radar_decoder.encoder.layers[0].p_drop = 0.0  # the golden was generated with dropout 0 (its masks are RNG draws)
    m = SimpleNamespace(config=SimpleNamespace(appearance_dim=0), rgb_decoder=dec.rgb_decoder, lidar_decoder=dec.lidar_decoder,
                        decode_radar=dec.decode_radar, parameters=dec.parameters, named_parameters=dec.named_parameters)
    return dec, m


def _load_reference_parameters(dec, names=None):
    g = load_golden("model")
    sd = {k[len("param."):]: v for k, v in g.items() if k.startswith("param.") and torch.is_tensor(v)}
    missing = dec.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if "num_batches_tracked" not in k], missing
    return sd


@pytest.mark.parametrize("loss_type", ["nll", "euclidean"])
def test_decoder_loss_head_vs_reference_training_branch(loss_type):
    from neuradar_amd.decoder_losses import DecoderLossHead, DecoderLossSettings

    g = load_golden("model_train")
    dec, m = _decoder_model(48)
    _load_reference_parameters(dec)
    n_cam = int(g["n_patch"]) * int(g["patch"]) ** 2
    n_lid = int(g["is_lidar"].sum())
    n = g["features"].shape[0]
    n_rad = n - n_cam - n_lid
    layout = {"camera": (0, n_cam), "lidar": (n_cam, n_lid), "radar": (n_cam + n_lid, n_rad)}
    head = DecoderLossHead(m, layout, int(g["patch"]), int(g["n_scan"]), 32, DecoderLossSettings(radar_loss_type=loss_type))
    d = lambda x: x.to(DEV)  # noqa: E731
    rng, ti = torch.ones(n), torch.zeros(n)
    rng[n_cam:n_cam + n_lid] = g["distance"][:, 0]
    ti[n_cam:n_cam + n_lid] = g["lidar"][:, 3]
    batch = {"image": d(g["image"]), "did_return": d(g["did_return"][:, 0].to(torch.uint8)), "range": d(rng), "target_intensity": d(ti),
             "directions_spher": d(g["spher"]), "radar": d(g["radar"]), "radar_seg": torch.tensor([0, 27, 28], dtype=torch.int32, device=DEV)}
    feats = d(g["features"]).requires_grad_(True)
    depth = d(g["depth"][:, 0]).requires_grad_(True)
    terms = head.losses(feats, depth, None, None, batch)
    t = loss_type + "."
    assert_close(head.last["rgb"].detach().cpu(), g[t + "rgb"], rtol=1e-4, atol_scale=1e-5, what="rgb")
    assert_close(head.last["radar_output"].detach().cpu(), g[t + "radar_output"], rtol=1e-4, atol_scale=1e-5, what="radar_output")
    for i in range(2):
        assert torch.equal(head.last["assoc"][i].cpu().long(), g[t + f"assoc_{i}"]), f"association of scan {i}"
    assert_close(terms["rgb_loss"].detach().cpu(), g[t + "loss.rgb_loss"], rtol=1e-4, what="rgb loss")
    assert_close(terms["radar_loss"].detach().cpu(), g[t + "loss.radar_loss"], rtol=1e-4, what="radar loss")
    lidar_ref = g[t + "loss.depth_loss"] + g[t + "loss.intensity_loss"] + g[t + "loss.ray_drop_loss"]
    assert_close(terms["lidar_losses"].detach().cpu(), lidar_ref, rtol=1e-4, what="lidar depth + intensity + ray-drop losses")
    names = [k for k, _ in dec.named_parameters() if (t + "gsum." + k) in g]
    params = dict(dec.named_parameters())
    grads = torch.autograd.grad(sum(terms.values()), [feats, depth] + [params[k] for k in names], allow_unused=True)
    # the golden's total also holds the two proposal-level depth terms, which do not depend on features / depth / decoders
    assert_close(grads[0].cpu(), g[t + "g_features"], rtol=1e-3, atol_scale=1e-4, what="d loss / d features")
    assert_close(grads[1].cpu(), g[t + "g_depth"][:, 0], rtol=1e-3, atol_scale=1e-5, what="d loss / d depth")
    checked = 0
    # (a convolution bias in front of a training-mode batch norm has a mathematically zero gradient: rounding noise ~1e-8 on
    # both sides, hence the absolute floor)
    for k, gr in zip(names, grads[2:]):
        scale = float(g[t + "gabs." + k])
        if gr is None:
            assert scale == 0.0, k
            continue
        assert abs(float(gr.double().sum()) - float(g[t + "gsum." + k])) <= 1e-3 * scale + 1e-6, k
        assert abs(float(gr.double().abs().sum()) - scale) <= 1e-3 * scale + 1e-6, k
        if (t + "grad." + k) in g and scale > 1e-5:
            assert_close(gr.cpu(), g[t + "grad." + k], rtol=2e-3, atol_scale=2e-4, what="grad " + k)
        checked += 1
    assert checked > 40


@pytest.mark.parametrize("n,frac_ret", [(4661, 0.9), (200, 0.85), (1, 1.0), (2, 0.0), (4097, 0.5), (300, 1.0)])
def test_lidar_loss_kernels_vs_oracle(n, frac_ret):
    """nr_lidar_depth_quantile / nr_lidar_losses: order statistics without a sort, torch.quantile's interpolation, the two
    masks, values and gradients -- at the bench's lidar count, around the rank tile size, and on tiny segments."""
    from neuradar_amd.decoder_losses import DecoderLossSettings, lidar_losses
    from oracle import decoder_losses as odl

    gen = torch.Generator().manual_seed(n)
    row0, B = 37, n + 90
    depth = torch.rand(B, generator=gen) * 200.0
    rng = torch.rand(B, generator=gen) * 120.0 + 1.0
    did = torch.rand(B, generator=gen) < frac_ret
    if n > 10:
        depth[row0 + 3] = rng[row0 + 3]  # an exact hit: loss 0, gradient sign(0) = 0
        depth[row0 + 5:row0 + 9] = depth[row0 + 4]  # ties
        rng[row0 + 5:row0 + 9] = rng[row0 + 4]
        did[row0 + 4:row0 + 9] = True
    ti = torch.rand(B, generator=gen)
    y = torch.randn(n, 2, generator=gen)
    c = DecoderLossSettings()
    d_h, y_h = depth.to(DEV).requires_grad_(True), y.to(DEV).requires_grad_(True)
    loss, stats = lidar_losses(d_h, y_h, did.to(torch.uint8).to(DEV), rng.to(DEV), ti.to(DEV), row0, n, c)
    loss.backward()
    # oracle on the lidar segment
    seg = slice(row0, row0 + n)
    d_r, y_r = depth[seg, None].clone().requires_grad_(True), y.clone().requires_grad_(True)
    oc = odl.LossSettings()
    un = odl.lidar_depth_unreduced(d_r, rng[seg, None], did[seg], oc)
    q = torch.quantile(un.detach(), oc.quantile_threshold)
    mask = (un.detach() < q)[:, 0]
    assert abs(float(stats[2]) - float(q)) <= 1e-6 * max(1.0, abs(float(q))), "quantile value"
    assert int(stats[5]) == int(mask.sum()) and int(stats[6]) == int((mask & did[seg]).sum())
    ref = torch.zeros(())
    if bool(mask.any()):
        ref = ref + oc.depth_mult * un[mask].mean()
    qr = mask & did[seg]
    if bool(qr.any()):
        ref = ref + oc.intensity_mult * ((ti[seg][qr] - y_r[qr, 0].sigmoid()) ** 2).mean()
    ref = ref + oc.ray_drop_loss_mult * torch.nn.functional.binary_cross_entropy_with_logits(y_r[:, 1], (~did[seg]).float())
    ref.backward()
    assert_close(loss.detach().cpu(), ref.detach(), rtol=1e-5, what="lidar losses")
    g_d = d_h.grad.cpu()
    assert float(g_d[:row0].abs().max()) == 0.0 and float(g_d[row0 + n:].abs().max()) == 0.0
    ref_gd = d_r.grad[:, 0] if d_r.grad is not None else torch.zeros(n)
    assert_close(g_d[seg], ref_gd, rtol=1e-5, atol_scale=1e-7, what="d / d depth")
    assert_close(y_h.grad.cpu(), y_r.grad, rtol=1e-4, atol_scale=1e-6, what="d / d decoder outputs")


def _full_model(loss_type):
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.step import HotPathConfig, NeuRadarHotPath

    cfg = HotPathConfig(field=NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=14))),
                        appearance_dim=16, num_sensors=3, decoders=True, radar_loss_type=loss_type)
    for pc in (cfg.proposal_field_1, cfg.proposal_field_2):
        pc.grid.static.log2_hashmap_size = 14
    torch.manual_seed(5)
    model = NeuRadarHotPath(cfg).to(DEV).train()
    model.radar_decoder.encoder.layers[0].p_drop = 0.0  # parity runs: dropout masks are RNG draws
    with torch.no_grad():
        model.field.hashgrid.static_grid.hash_table.mul_(300.0)
        model.proposal_fields[1].hashgrid.static_grid.hash_table.mul_(1500.0)
        model.appearance_embedding.weight.mul_(0.3)
    return model


@pytest.mark.parametrize("loss_type", ["nll", "euclidean"])
def test_fused_step_with_decoders_matches_modular_path_and_oracle(loss_type):
    from neuradar_amd import losses
    from neuradar_amd.decoder_losses import DecoderLossHead, DecoderLossSettings
    from neuradar_amd.fused_step import FusedTrainStep
    from neuradar_amd.rays import RayBundle
    from oracle import decoder_losses as odl

    model = _full_model(loss_type)
    c = model.config
    gen = torch.Generator().manual_seed(3)
    patch, n_patch, n_scan, per_scan, n_lid = 8, 2, 1, 45, 60
    n_cam, n_rad = n_patch * patch * patch, n_scan * per_scan
    B = n_cam + n_rad + n_lid
    r0_rad, r0_lid = n_cam, n_cam + n_rad  # batch order camera, radar, lidar (the bench's: coherent segments first)
    o = torch.cat([torch.randn(B, 2, generator=gen) * 3.0, torch.full((B, 1), 1.7)], dim=1)
    d = torch.nn.functional.normalize(torch.cat([torch.ones(B, 1), 0.5 * torch.randn(B, 2, generator=gen)], dim=1), dim=-1)
    area = torch.cat([torch.full((n_cam,), 2.25e-6), torch.full((n_rad,), 9e-6), torch.full((n_lid,), 4.5e-6)])
    times = 20.0 * torch.rand(B, generator=gen)
    is_lidar = torch.zeros(B, dtype=torch.bool)
    is_lidar[r0_lid:] = True
    is_radar = torch.zeros(B, dtype=torch.bool)
    is_radar[r0_rad:r0_lid] = True
    did_return = torch.ones(B, dtype=torch.bool)
    did_return[r0_lid:] = torch.rand(n_lid, generator=gen) < 0.8
    rng = torch.ones(B)
    rng[r0_lid:] = 2.0 + 60.0 * torch.rand(n_lid, generator=gen)
    sensor = torch.cat([torch.zeros(n_cam), 2 * torch.ones(n_rad), torch.ones(n_lid)]).long()
    target_i = torch.rand(B, generator=gen)
    spher = torch.zeros(B, 2)
    spher[r0_rad:r0_lid] = torch.stack([torch.rand(n_rad, generator=gen) * 1.6 - 0.8, torch.rand(n_rad, generator=gen) * 0.48 - 0.08], dim=-1)
    image = torch.rand(n_patch, patch * 3, patch * 3, 3, generator=gen)
    n_det = 11
    radar = torch.cat([torch.randn(n_det, 3, generator=gen) * 8.0 + torch.tensor([15.0, 0.0, 0.0]), torch.rand(n_det, 2, generator=gen)], 1)
    t_rand, j1, j2 = torch.rand(B, 129, generator=gen), torch.rand(B, generator=gen), torch.rand(B, generator=gen)
    dv = lambda x: x.to(DEV)  # noqa: E731

    # ---- reference value: modular HIP path up to the rendered outputs, the CPU oracle behind them
    bundle = RayBundle(dv(o), dv(d), dv(area)[:, None], fars=torch.full((B, 1), 1e6, device=DEV), times=dv(times)[:, None],
                       metadata={"is_lidar": dv(is_lidar)[:, None], "did_return": dv(did_return)[:, None],
                                 "directions_norm": dv(rng)[:, None], "sensor_idxs": dv(sensor)[:, None]})
    out = model.get_nff_outputs(bundle, t_rand=dv(t_rand), jitters=(dv(j1)[:, None], dv(j2)[:, None]))
    cs = [s.spacing for s in out["ray_samples_list"]]
    ws = [w[..., 0] for w in out["weights_list"]]
    loss = c.interlevel_loss_mult * losses.zipnerf_interlevel_loss(cs, ws) + c.distortion_loss_mult * losses.distortion_loss(cs[-1], ws[-1])
    loss = loss + c.carving_mult * (out["non_nearby_weights"] ** 2).sum() / n_lid
    for i in (0, 1):
        loss = loss + c.prop_lidar_loss_mult * c.carving_mult * out[f"prop_weights_loss_{i}"] / n_lid
    dec_names = [k for k, _ in model.named_parameters() if k.split(".")[0] in (
        "rgb_decoder", "lidar_decoder", "radar_decoder", "offset_head", "radar_uncertainty_head", "existence_probability_head")]
    named = dict(model.named_parameters())
    p_cpu = {k: named[k].detach().cpu().clone().requires_grad_(True) for k in dec_names}
    batch_cpu = {"image": image, "distance": rng[r0_lid:, None], "did_return": did_return[:, None],
                 "lidar": torch.cat([torch.zeros(n_lid, 3), target_i[r0_lid:, None]], 1), "radar": radar,
                 "radar_indices": torch.stack([torch.zeros(n_det), torch.arange(n_det).float()], 1).long()}
    oc = odl.LossSettings(radar_loss_type=loss_type)
    terms = odl.decoder_losses(out["features"].cpu(), out["depth"].cpu(), [out["prop_depth_0"].cpu(), out["prop_depth_1"].cpu()], spher,
                               is_lidar[:, None], is_radar[:, None], batch_cpu, p_cpu, patch, n_scan, oc)
    loss_cpu = odl.total(terms)
    total = loss.cpu() + loss_cpu
    nff = {n: p for n, p in model.named_parameters() if p.requires_grad and n not in p_cpu}
    grads = torch.autograd.grad(total, list(nff.values()) + list(p_cpu.values()), allow_unused=True)
    ref = dict(zip(list(nff) + list(p_cpu), grads))
    for p in model.parameters():
        if p.grad is not None:
            p.grad.zero_()

    # ---- the fused step with the decoder head
    fused = FusedTrainStep(model, B, coherent_rays=n_cam + n_rad)
    fused.set_lidar(dv(is_lidar).to(torch.uint8), dv(did_return).to(torch.uint8), dv(rng), r0_lid, n_lid, prop_depth_loss=True)
    layout = {"camera": (0, n_cam), "radar": (r0_rad, n_rad), "lidar": (r0_lid, n_lid)}
    head = DecoderLossHead(model, layout, patch, n_scan, 16, DecoderLossSettings(radar_loss_type=loss_type))
    batch = {"image": dv(image), "did_return": dv(did_return).to(torch.uint8), "range": dv(rng), "target_intensity": dv(target_i),
             "directions_spher": dv(spher), "radar": dv(radar), "radar_seg": torch.tensor([0, n_det], dtype=torch.int32, device=DEV)}
    fused.set_decoders(head, [batch, batch], dv(sensor))
    floss = fused.forward_backward(dv(o), dv(d), dv(area), torch.full((B,), 1e6, device=DEV), None, None, dv(t_rand), dv(j1), dv(j2),
                                   times=dv(times))
    assert torch.equal(head.last["assoc"][0].cpu().long(), terms["assoc"][0]), "Hungarian association"
    assert_close(fused.outputs()["features"].cpu(), out["features"][:, :32].detach().cpu(), rtol=1e-4, atol_scale=1e-5, what="features")
    assert_close(floss.sum().cpu(), total.detach(), rtol=1e-4, atol_scale=1e-6, what="loss")
    checked = 0
    for n_, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if ref[n_] is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n_
            continue
        if float(ref[n_].abs().sum()) < 1e-5 and "main_branch" in n_ and n_.endswith("bias"):
            assert float(p.grad.abs().sum()) < 1e-5, n_  # convolution bias in front of a training-mode batch norm: zero up to rounding
            continue
        assert_close(p.grad.cpu(), ref[n_].cpu(), rtol=2e-3, atol_scale=2e-4, what="fused grad " + n_)
        checked += 1
    for must in ("appearance_embedding.weight", "lidar_decoder.layers.0.weight", "rgb_decoder.2.main_branch.0.weight",
                 "radar_decoder.encoder.layers.0.self_attn.in_proj_weight", "offset_head.layers.0.weight",
                 "field.hashgrid.static_grid.hash_table", "proposal_fields.1.hashgrid.static_grid.hash_table"):
        assert ref[must] is not None and float(ref[must].abs().max()) > 0, must
    assert checked > 50


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_cnn_16_bit_working_copies_equal_autocast(dtype, monkeypatch):
    """DecoderLossHead runs the RGB CNN on 16-bit working copies of its convolution parameters (one copy before the forward,
    one mixed-precision add of the gradients after the backward) instead of torch.autocast's per-parameter casts: the same
    output, loss and batch-norm statistics (to the operand type's rounding); gradients as close to the fp32 CNN's as autocast's are (the backward runs the same
    operations on the same 16-bit operands, but MIOpen's benchmark mode picks the algorithms of every head independently and
    the gradients sit at 1e-4 ... 1e-8 -- fp16's subnormal range -- so two 16-bit runs differ by ~2 % in relative L2)."""
    from neuradar_amd.decoder_losses import DecoderLossHead
    from neuradar_amd.fused_step import flatten_parameters

    g = load_golden("model_train")
    n_cam = int(g["n_patch"]) * int(g["patch"]) ** 2
    layout = {"camera": (0, n_cam), "lidar": (n_cam, 0), "radar": (n_cam, 0)}
    feats = g["features"][:n_cam].to(DEV)
    image = g["image"].to(DEV)
    res = {}
    for mode in ("copies", "autocast", "fp32"):
        monkeypatch.setenv("NR_CNN_SHADOW", "1" if mode == "copies" else "0")
        torch.manual_seed(0)
        dec, m = _decoder_model(48)
        _load_reference_parameters(dec)
        flatten_parameters(list(dec.rgb_decoder.parameters()))
        head = DecoderLossHead(m, layout, int(g["patch"]), 0, 0, cnn_autocast=None if mode == "fp32" else dtype)
        slots = torch.zeros(1025, device=DEV)
        for _ in range(2):  # twice: the working copies' gradient buffer is cleared and reused
            for p_ in dec.rgb_decoder.parameters():
                p_.grad.zero_()
            g_f, _ = head.backward_into(feats, torch.ones(n_cam, device=DEV), None, None, {"image": image}, slots)
        assert bool(head._shadow) == (mode == "copies")
        res[mode] = (head.last["rgb"].detach().clone(), float(slots.sum()), g_f.clone(),
                     {k: v.grad.clone() for k, v in dec.rgb_decoder.named_parameters()},
                     {k: v.clone() for k, v in dec.rgb_decoder.named_buffers()})
    a, b, f = res["copies"], res["autocast"], res["fp32"]
    # (usually bit-identical; MIOpen's benchmark mode may pick another forward algorithm for the second head: 1 run in ~8)
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert_close(a[0].cpu(), b[0].cpu(), rtol=tol, atol_scale=tol, what="rgb")
    assert abs(a[1] - b[1]) <= tol * abs(b[1]), ("loss", a[1], b[1])
    for k in a[4]:
        assert_close(a[4][k].float().cpu(), b[4][k].float().cpu(), rtol=tol, atol_scale=tol, what="batch-norm statistics " + k)
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-30))  # noqa: E731
    rows = {"d loss / d features": (a[2], b[2], f[2])}
    rows.update({k: (a[3][k], b[3][k], f[3][k]) for k in a[3] if not k.endswith(("main_branch.0.bias", "main_branch.3.bias"))})
    for k, (x, y, z) in rows.items():  # (a convolution bias in front of a batch norm has no gradient: rounding noise only)
        e_copies, e_autocast = rel(x, z), rel(y, z)
        assert e_copies <= 1.5 * e_autocast + 2e-3, (k, e_copies, e_autocast)
        assert rel(x, y) < 0.05, (k, rel(x, y))
