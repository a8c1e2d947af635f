"""The decoders and their losses inside the training step (BASELINE configs[2] "full", configs[3] "nll",
configs[4] transformer in the step):
  * DecoderLossHead (RGB CNN + MSE, lidar MLP + quantile-masked lidar losses, radar transformer + heads + Hungarian-matched
    radar loss, all on the device) against the vectors the reference NeuRadarModel's own training-branch methods produce
    (tests/golden/model_train.npz);
  * the lidar loss kernels against the oracle on ragged / degenerate inputs;
  * FusedTrainStep.set_decoders: loss and EVERY parameter gradient of the fused step against the modular HIP path for the
    field part + the CPU oracle (oracle/decoder_losses.py) for everything behind the rendered features."""
import os
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


EPS = {"bfloat16": 2.0 ** -8, "float16": 2.0 ** -11}  # unit roundoff of the 16-bit operand types (tests/test_gpu_lp.py)


@pytest.mark.parametrize("mlp_dtype", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("loss_type", ["nll", "euclidean"])
def test_fused_step_with_decoders_matches_modular_path_and_oracle(loss_type, mlp_dtype):
    """fp32: element-wise parity (rtol 1e-4 outputs, 2e-3 gradients).  bfloat16 / float16 = the step bench.py's `full_model`
    numbers are timed on (BASELINE configs[2] full / configs[4]): field MLPs on 16-bit MFMA operands, the RGB CNN on 16-bit
    working copies of its parameters inside the optimizer's flat buffers, 32-bit tile sums in the binned scatter -- against the
    SAME fp32 reference (modular fp32 HIP path + CPU oracle).  Bounds as tests/test_gpu_lp.py states and justifies them:
    outputs within 5u of their scale (five chained 16-bit layers), every parameter gradient within 2*sqrt(u) in relative L2
    (a u-fraction of ReLU masks flips; x sqrt(16) for the parameters behind the 11-convolution CNN's 16-bit backward, see
    below), Hungarian associations exact."""
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
    lp = mlp_dtype != "float32"
    if lp:
        # what bench.py builds for the 16-bit workloads: 16-bit MFMA operands in the field (static loss scale for fp16), every
        # parameter inside the optimizers' flat buffers (channels-last convolution weights), the CNN on 16-bit working copies
        import bench

        model.field.config.mlp_dtype = mlp_dtype
        model.field.config.mlp_grad_scale = 8192.0 if mlp_dtype == "float16" else 1.0
        bench.build_optimizers(model)  # re-homes parameters and gradients (values unchanged); the optimizers are not stepped
        for p in model.parameters():
            if p.grad is not None:
                p.grad.zero_()
    fused = FusedTrainStep(model, B, coherent_rays=n_cam + n_rad)
    fused.set_lidar(dv(is_lidar).to(torch.uint8), dv(did_return).to(torch.uint8), dv(rng), r0_lid, n_lid, prop_depth_loss=True)
    layout = {"camera": (0, n_cam), "radar": (r0_rad, n_rad), "lidar": (r0_lid, n_lid)}
    head = DecoderLossHead(model, layout, patch, n_scan, 16, DecoderLossSettings(radar_loss_type=loss_type),
                           cnn_autocast={"float32": None, "bfloat16": torch.bfloat16, "float16": torch.float16}[mlp_dtype])
    batch = {"image": dv(image), "did_return": dv(did_return).to(torch.uint8), "range": dv(rng), "target_intensity": dv(target_i),
             "directions_spher": dv(spher), "radar": dv(radar), "radar_seg": torch.tensor([0, n_det], dtype=torch.int32, device=DEV)}
    fused.set_decoders(head, [batch, batch], dv(sensor))
    floss = fused.forward_backward(dv(o), dv(d), dv(area), torch.full((B,), 1e6, device=DEV), None, None, dv(t_rand), dv(j1), dv(j2),
                                   times=dv(times))
    assert torch.equal(head.last["assoc"][0].cpu().long(), terms["assoc"][0]), "Hungarian association"
    if lp:
        assert bool(head._shadow), "the CNN must run on its 16-bit working copies (the path bench.py times)"
        assert fused.bin_sum_bits == 32 and fused.field_struct.dtype == {"bfloat16": 1, "float16": 2}[mlp_dtype]
    u = EPS.get(mlp_dtype)
    out_tol = dict(rtol=1e-4, atol_scale=1e-5) if not lp else dict(rtol=5 * u, atol_scale=5 * u)
    assert_close(fused.outputs()["features"].cpu(), out["features"][:, :32].detach().cpu(), what="features", **out_tol)
    assert_close(fused.outputs()["depth"].cpu(), out["depth"].detach().cpu(), what="depth", **out_tol)
    assert_close(floss.sum().cpu(), total.detach(), rtol=1e-4 if not lp else 5 * u, atol_scale=1e-6, what="loss")
    checked, rows, bad = 0, [], []
    for n_, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if ref[n_] is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n_
            continue
        if float(ref[n_].abs().sum()) < 1e-5 and "main_branch" in n_ and n_.endswith("bias"):
            # convolution bias in front of a training-mode batch norm: zero up to rounding (16-bit: rounding noise of the
            # activation gradients summed over the patch, far below the weight gradient of the same convolution)
            w_grad = dict(model.named_parameters())[n_[:-4] + "weight"].grad
            assert float(p.grad.abs().sum()) < (1e-5 if not lp else 0.05 * float(w_grad.abs().sum())), n_
            continue
        if not lp:
            assert_close(p.grad.cpu(), ref[n_].cpu(), rtol=2e-3, atol_scale=2e-4, what="fused grad " + n_)
        else:
            want = ref[n_].double().cpu()
            err = float((p.grad.double().cpu() - want).norm() / want.norm().clamp_min(1e-30))
            # parameters whose gradient passes through the RGB CNN's 16-bit backward sit behind up to 11 convolutions (each
            # followed by a batch norm / ReLU) plus the field's 5 layers: independent sqrt(u)-sized perturbations per layer
            # add up like a random walk -- 2 sqrt(u) sqrt(16).  torch.autocast of the same CNN sits at the same distance from
            # its fp32 self (test_cnn_16_bit_working_copies_equal_autocast compares the two directly).  Measured: bf16 0.27
            # (appearance embedding) / 0.23 (first convolutions), fp16 0.13 / 0.10; 1e-3 ... 7e-2 everywhere else.
            behind_cnn = n_.startswith(("rgb_decoder.", "appearance_embedding", "field.mlp_", "field.hashgrid"))
            bound = 2 * u ** 0.5 * (4.0 if behind_cnn else 1.0)
            rows.append((n_, err, float(want.norm()), bound))
            if not err < bound:
                bad.append((n_, err, bound))
        checked += 1
    for n_, err, nrm, bound in rows:
        print(f"{mlp_dtype} {loss_type} grad {n_:70s} rel L2 {err:.3e}   |ref| {nrm:.3e}   bound {bound:.3e}")
    assert not bad, f"{mlp_dtype}: gradients outside their relative-L2 bound (2*sqrt(u) = {2 * u ** 0.5:.3e}, x4 behind the CNN): {bad}"
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
        if os.environ.get("NR_TEST_TRACE"):  # tools/repro_abort.sh: which leg, and where the allocator's segments end
            torch.cuda.synchronize()
            segs = sorted((s_["address"], s_["total_size"]) for s_ in torch.cuda.memory_snapshot())
            print(f"[leg] {dtype} {mode} segments: " + " ".join(f"{a:x}+{n:x}" for a, n in segs), flush=True)
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
        print(f"{dtype} {k:50s} copies vs fp32 {e_copies:.3e}   autocast vs fp32 {e_autocast:.3e}   copies vs autocast {rel(x, y):.3e}")
        assert e_copies <= 1.5 * e_autocast + 2e-3, (k, e_copies, e_autocast)
        # (the two 16-bit runs are not compared with each other: the working copies' 7 x 7 convolutions run on conv7.hip, autocast's
        # on the library's kernels -- two valid roundings of every layer's output, eleven layers deep: 15 % apart in bf16 while both
        # sit equally far from fp32 -- and in fp16 only the copies' backward runs under the loss scale)


@pytest.mark.parametrize("workload", ["mixed16384_neuradar_full", "mixed16384_neuradar_full_fp16", "mixed8192_vod_nll"])
def test_full_model_workloads_train_at_full_size(workload):
    """BASELINE configs[2] "full" (bf16), configs[4] (fp16 MFMA, transformer in the step) and configs[3] (VoD scan, nll) per-GPU
    shapes, exactly as bench.py builds and times them (the optimizers incl. cnn / transformer, graph replay of the pipelined
    step): 50 training steps at FULL size, then size-independent properties -- loss finite at every step, the decoders' loss terms
    lower at the end than at the start; every parameter, Adam moment and batch-norm running statistic finite; the Hungarian association of the
    last step a partial matching: every detection matched exactly once, matched count = min(detections, predictions) per
    scan, indices in range."""
    import bench
    from neuradar_amd.parallel import GradAllReducer

    dev = torch.device(DEV)
    wl = bench.WORKLOADS[workload]
    n_rays = wl["rays"]
    mlp_dtype = wl.get("mlp_dtype", "bfloat16")
    model = bench.build_model(wl, dev, mlp_dtype, 8192.0 if mlp_dtype == "float16" else 1.0)
    opts = bench.build_optimizers(model)
    reducer = GradAllReducer(None, buffers=[g for o in opts for g in o.grad_buffers()])
    scene = bench.SyntheticScene(dev, seed=1000, radar=wl.get("radar", "zod"))
    torch.manual_seed(1234)
    targets = (0.1 * torch.randn(n_rays, 32, device=dev), 5.0 + 50.0 * torch.rand(n_rays, 1, device=dev))
    fwd_bwd, _, stepper = bench.make_step(model, scene, opts, reducer, targets, n_rays, fused=True, fuse_optimizer=True, mixed=wl)
    head = stepper.dec["head"]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    losses_, dec_ = [], []

    def record():
        losses_.append(float(stepper.loss.sum()))
        dec_.append(float(sum(head.last["terms"].values())))

    with torch.cuda.stream(side):
        for _ in range(4):  # eager: MIOpen's algorithm search, lazy allocations
            fwd_bwd()
            record()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fwd_bwd()
            fwd_bwd()
        for _ in range(23):
            g.replay()
            record()  # the second step of the pair
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ls, ds = torch.tensor(losses_), torch.tensor(dec_)
    print(f"{workload}: loss {losses_[0]:.4f} -> {losses_[-1]:.4f}; decoder-side terms (rgb + lidar + radar) {dec_[0]:.4f} -> {dec_[-1]:.4f}"
          f"  (mean of the first 4: {float(ds[:4].mean()):.4f}, of the last 4: {float(ds[-4:].mean()):.4f})")
    assert bool(torch.isfinite(ls).all()) and bool(torch.isfinite(ds).all()), (losses_, dec_)
    # What must go down is what the decoders are supervised with (rgb, lidar depth / intensity / ray drop, radar).  The TOTAL of
    # this synthetic batch does not, and that is the reference's loss, not the kernels: its proposal-level lidar depth terms
    # (neuradar.py:641-648) use the UN-normalised depth sum(w * t) over samples that reach the 20-km sky distance, so while the
    # density field is still spread out (accumulation 0.6-0.8 around step 50-150; random lidar ranges never pull it to 1) a
    # few per cent of proposal weight on the far bins is kilometres of depth error (tools/diag_full_losses.py: 0.14 -> 23 -> 7).
    assert float(ds[-4:].mean()) < float(ds[:4].mean()), "the decoders' losses did not decrease over 50 steps"
    for n_, p in model.named_parameters():
        assert bool(torch.isfinite(p).all()), f"parameter {n_}"
    for n_, b in model.named_buffers():
        if b.is_floating_point():
            assert bool(torch.isfinite(b).all()), f"buffer {n_}"
    for o_ in opts:
        for m, v in o_.state:
            assert bool(torch.isfinite(m).all()) and bool(torch.isfinite(v).all()) and float(v.min()) >= 0.0
    assoc = head.last["assoc"].cpu().long()  # [scans, n]: matched detection of every prediction, -1 = none
    n_det, n_pred = scene.radar_detections_per_scan, assoc.shape[1]
    for s_ in range(assoc.shape[0]):
        a = assoc[s_]
        m = a[a >= 0]
        assert int(m.numel()) == min(n_det, n_pred), f"scan {s_}: {m.numel()} matches for {n_det} detections / {n_pred} predictions"
        assert int(m.max()) < n_det and torch.unique(m).numel() == m.numel(), f"scan {s_}: a detection is matched twice"
