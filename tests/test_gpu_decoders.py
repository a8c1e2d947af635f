"""GPU parity of the per-ray decoders (f-2) and the radar point-set bar of the north star (N1: Chamfer within 1e-3)
against vectors from the reference NeuRadarModel's own decode_features / sample_radar_points / chamfer_distance."""
import pytest
import torch

from helpers import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _decoders(g):
    from neuradar_amd.decoders import Decoders

    dec = Decoders(48).to(DEV).eval()
    sd = {k[len("param."):]: v for k, v in g.items() if k.startswith("param.") and isinstance(v, torch.Tensor)}
    sd.update({k[len("param."):]: torch.tensor(v) for k, v in g.items() if k.endswith("num_batches_tracked")})
    missing, unexpected = dec.load_state_dict(sd, strict=True), None  # the reference's names load unchanged
    return dec


def test_decode_features_vs_reference_model():
    g = load_golden("model")
    dec = _decoders(g)
    d = lambda k: g[k].to(DEV)  # noqa: E731
    rgb, intensity, drop, ro = dec(d("dec_features"), (8, 8), d("dec_depth"), d("dec_spher"), is_lidar=d("dec_is_lidar"),
                                   is_radar=d("dec_is_radar"), num_radar_scans=2)
    assert_close(intensity.cpu(), g["dec_intensity"], rtol=1e-4, atol_scale=1e-5, what="intensity")
    assert_close(drop.cpu(), g["dec_ray_drop_logit"], rtol=1e-4, atol_scale=1e-5, what="ray_drop_logit")
    assert_close(ro.detach().cpu(), g["dec_radar_output"], rtol=1e-4, atol_scale=1e-5, what="radar_output")
    assert_close(rgb.detach().cpu(), g["dec_rgb"], rtol=1e-4, atol_scale=1e-4, what="rgb")
    named = dict(dec.named_parameters())
    keys = [k[len("dec_grad."):] for k in g if k.startswith("dec_grad.")]
    grads = torch.autograd.grad((ro * d("dec_g_radar_output")).sum(), [named[k] for k in keys])
    for k, gr in zip(keys, grads):
        assert_close(gr.cpu(), g["dec_grad." + k], rtol=1e-3, atol_scale=1e-4, what="grad " + k)


def test_radar_point_set_chamfer_within_1e3_of_reference():
    """North star: "radar-set Chamfer within 1e-3".  The HIP path's radar_output -> sampled detections -> Chamfer distance
    to the detections of the scan, against the reference's value for ITS radar_output on the same inputs."""
    from neuradar_amd.decoders import sample_radar_points
    from oracle import radar as orad

    g = load_golden("model")
    dec = _decoders(g)
    d = lambda k: g[k].to(DEV)  # noqa: E731
    is_r = d("dec_is_radar")[:, 0]
    ro = dec.decode_radar(d("dec_features")[is_r], d("dec_depth")[is_r], d("dec_spher")[is_r], 2).detach()
    # the golden spreads the existence probabilities around the 0.5 threshold (cd_radar_output[..., 0]); positions are ours
    ro[..., 0] = d("cd_radar_output")[..., 0]
    pts, ber = sample_radar_points(ro, 0.5)
    assert torch.equal(ber.cpu(), g["cd_ber"])
    cd = orad.chamfer_distance(pts.cpu().numpy(), g["cd_gt"].numpy())
    assert abs(cd - g["cd_value"]) <= 1e-3 * g["cd_value"], (cd, g["cd_value"])
    # and the Hungarian-matched radar loss of the deterministic head on our output
    loss, assoc = orad.radar_loss_euclidean(g["radar_batch"], dec.decode_radar(d("dec_features")[is_r], d("dec_depth")[is_r],
                                                                               d("dec_spher")[is_r], 2).detach().cpu(), g["radar_indices"])
    assert abs(float(loss) - g["radar_loss"]) <= 1e-3 * g["radar_loss"] and torch.equal(assoc, g["radar_assoc_last"])


def test_field_density_branch_vs_reference_golden():
    """a16, use_sdf=False (fields/neurad_field.py:149-150, models/neuradar.py:1018-1022): DENSITY head on the HIP kernels
    against the reference's own field, then density -> weights against the oracle's render_weight_from_density."""
    from neuradar_amd.field_heads import FieldHeadNames
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.rays import RaySamples
    from oracle import render as orender

    g = load_golden("field_density")
    fld = NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=int(g["log2t"]))), use_sdf=False
                            ).setup(actors=None, static_scale=100.0).to(DEV)
    assert "sdf_to_density.beta" not in dict(fld.named_parameters())  # the reference builds none in this branch either
    with torch.no_grad():
        fld.hashgrid.static_grid.hash_table.copy_(g["table"])
        fld.hashgrid.static_grid.scalings.copy_(g["scalings"])
        for i, l in enumerate(fld.mlp_geo.layers):
            l.weight.copy_(g[f"geo_w{i}"]); l.bias.copy_(g[f"geo_b{i}"])
        for i, l in enumerate(fld.mlp_feature.layers):
            l.weight.copy_(g[f"feat_w{i}"]); l.bias.copy_(g[f"feat_b{i}"])
    e = g["edges"].to(DEV)
    B = e.shape[0]
    rs = RaySamples(g["origins"].to(DEV), g["directions"].to(DEV), g["pixel_area"].to(DEV), torch.zeros_like(e), e,
                    torch.zeros(B, 1, device=DEV), torch.full((B, 1), 1e6, device=DEV))
    out = fld(rs)
    assert FieldHeadNames.SDF not in out and FieldHeadNames.ALPHA not in out
    assert_close(out[FieldHeadNames.FEATURE].detach().cpu(), g["feature"], rtol=1e-4, atol_scale=1e-5, what="feature")
    assert_close(out[FieldHeadNames.DENSITY].detach().cpu(), g["density"], rtol=1e-4, atol_scale=1e-5, what="density")
    loss = (out[FieldHeadNames.FEATURE] * g["g_feature"].to(DEV)).sum() + (out[FieldHeadNames.DENSITY] * g["g_density"].to(DEV)).sum()
    named = dict(fld.named_parameters())
    keys = {"hashgrid.static_grid.hash_table": "grad_table", "mlp_geo.layers.0.weight": "grad_geo_w0", "mlp_geo.layers.1.weight": "grad_geo_w1",
            "mlp_geo.layers.1.bias": "grad_geo_b1", "mlp_feature.layers.0.weight": "grad_feat_w0"}
    grads = torch.autograd.grad(loss, [named[k] for k in keys])
    for (k, gk), gr in zip(keys.items(), grads):
        assert_close(gr.cpu(), g[gk], rtol=2e-4, atol_scale=2e-5, what="grad " + k)
    # density -> alpha -> weights: the elementwise step in front of the compositing kernel (step.py) == nerfacc's formula
    dens = out[FieldHeadNames.DENSITY].detach()
    alpha = 1.0 - torch.exp(-dens * rs.deltas)
    from neuradar_amd.renderers import render_weight_from_alpha
    w, _ = render_weight_from_alpha(alpha[..., 0])
    want, _, _ = orender.render_weight_from_density(g["edges"][:, :-1], g["edges"][:, 1:], g["density"][..., 0])
    assert_close(w.cpu(), want, rtol=1e-4, atol_scale=1e-5, what="weights from density")


def _attention_reference(q, k, v, keep, p):
    """torch fp64: softmax(q k^T / sqrt(D)), dropout with the given keep mask, times v."""
    s = (q.double() @ k.double().transpose(1, 2)) / (q.shape[-1] ** 0.5)
    pr = torch.softmax(s, dim=-1)
    if keep is not None:
        pr = pr * keep.double() / (1.0 - p)
    return pr @ v.double()


@pytest.mark.parametrize("shape", [(1, 3531, 48), (2, 257, 48), (1, 64, 32), (3, 65, 64), (1, 1, 48)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_kernels_match_torch(shape, p):
    """nr_attention_fwd / nr_attention_bwd against softmax attention in float64 (rtol 1e-4 of each tensor's scale), with an
    explicit keep mask for p > 0: the radar scan's size (107 x 33 tokens, d_model 48), several scans, token counts around
    the 64-row tiles, a single token."""
    from neuradar_amd import ops

    N, n, D = shape
    gen = torch.Generator().manual_seed(N * 1000 + n)
    q, k, v = (torch.randn(N, n, D, generator=gen) for _ in range(3))
    go = torch.randn(N, n, D, generator=gen)
    keep = (torch.rand(N, n, n, generator=gen) >= p).float() if p > 0 else None
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = _attention_reference(qr, kr, vr, keep, p)
    ref.backward(go.double())
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    out = ops.attention(qd, kd, vd, p, seed=5, keep_mask=None if keep is None else keep.to(DEV))
    out.backward(go.to(DEV))
    assert_close(out.detach().cpu(), ref.detach().float(), rtol=1e-4, atol_scale=1e-5, what="attention output")
    for got, want, what in ((qd.grad, qr.grad, "dq"), (kd.grad, kr.grad, "dk"), (vd.grad, vr.grad, "dv")):
        if n == 1 and what != "dv":  # one token: the softmax is the constant 1, dq = dk = 0 up to rounding (dP and delta are
            # the same dot product of ~7-sized terms summed in two orders: a few 1e-7, times k or q)
            assert float(got.abs().max()) < 2e-5, what
            continue
        assert_close(got.cpu(), want.float(), rtol=1e-4, atol_scale=2e-5, what=what)


def test_attention_with_peaked_softmax():
    """Logits of several hundred (q, k eight times the unit scale: every query attends to one or two keys, all other
    probabilities underflow): the online softmax and the recomputation from the log-sum-exp stay finite and match float64."""
    from neuradar_amd import ops

    gen = torch.Generator().manual_seed(31)
    q, k, v, go = (torch.randn(2, 333, 48, generator=gen) * s for s in (8.0, 8.0, 1.0, 1.0))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = _attention_reference(qr, kr, vr, None, 0.0)
    ref.backward(go.double())
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    out = ops.attention(qd, kd, vd)
    out.backward(go.to(DEV))
    assert torch.isfinite(out).all() and all(torch.isfinite(t.grad).all() for t in (qd, kd, vd))
    assert_close(out.detach().cpu(), ref.detach().float(), rtol=1e-4, atol_scale=1e-4, what="attention output")
    for got, want, what in ((qd.grad, qr.grad, "dq"), (kd.grad, kr.grad, "dk"), (vd.grad, vr.grad, "dv")):
        assert_close(got.cpu(), want.float(), rtol=1e-3, atol_scale=1e-3, what=what)


def test_attention_dropout_by_hash_is_consistent_between_forward_and_backward():
    """Without an explicit mask the keep decisions come from the hash of (seed, scan, query, key).  (i) about 1 - p of the
    probabilities survive: with v = 1 the output is sum_j P_ij keep_ij / (1 - p), mean 1; (ii) the backward uses the SAME
    decisions: the gradient of sum(out * go) with respect to v equals finite differences of the forward at that seed;
    (iii) another seed gives another mask."""
    from neuradar_amd import ops

    torch.manual_seed(2)
    N, n, D, p = 1, 512, 48, 0.25
    q, k = torch.randn(N, n, D, device=DEV), torch.randn(N, n, D, device=DEV)
    ones = torch.ones(N, n, D, device=DEV)
    o = ops.attention(q, k, ones, p, seed=11)
    assert abs(float(o.mean()) - 1.0) < 0.02 and float(o.std()) > 1e-3
    assert not torch.equal(o, ops.attention(q, k, ones, p, seed=12))
    assert torch.equal(o, ops.attention(q, k, ones, p, seed=11))
    v = torch.randn(N, n, D, device=DEV, requires_grad=True)
    go = torch.randn(N, n, D, device=DEV)
    (ops.attention(q, k, v, p, seed=11) * go).sum().backward()
    d = torch.zeros_like(v)
    d[0, 37, 5] = 1.0
    fd = ((ops.attention(q, k, v.detach() + 0.5 * d, p, seed=11) - ops.attention(q, k, v.detach() - 0.5 * d, p, seed=11)) * go).sum()
    assert abs(float(fd) - float(v.grad[0, 37, 5])) <= 1e-3 * max(1.0, abs(float(fd)))  # (linear in v: the difference is exact up to rounding)


def test_transformer_layer_with_hip_attention(monkeypatch):
    """decoders.Transformer(attention="hip") in eval mode at the radar scan's size against the same layer with its attention
    evaluated in float64 (same parameters): output rtol 1e-4, gradient of the input rtol 1e-3 of its scale; the default
    (torch's fused attention) is checked against the same reference at the looser 2e-2 its reduced internal precision
    allows."""
    from neuradar_amd import ops
    from neuradar_amd.decoders import Transformer

    torch.manual_seed(0)
    hip = Transformer(d_model=48, attention="hip").to(DEV).eval()
    tor = Transformer(d_model=48, attention="torch").to(DEV).eval()
    tor.load_state_dict(hip.state_dict())
    src, pos = torch.randn(1, 3531, 48, device=DEV), torch.randn(1, 3531, 48, device=DEV)
    wgt = torch.randn(1, 3531, 48, device=DEV)  # (sum(y^2) is constant behind the final LayerNorm: its gradient is pure rounding)

    def run(m):
        x = src.clone().requires_grad_(True)
        y = m(x, pos)
        (y * wgt).sum().backward()
        return y.detach().cpu(), x.grad.cpu()

    y_hip, g_hip = run(hip)  # (the fused layer: nr_encoder_* around nr_attention_*)
    y_tor, g_tor = run(tor)
    monkeypatch.setenv("NR_FUSED_ENCODER", "0")  # the reference: the modular layer (torch ops) with its attention in float64
    monkeypatch.setattr(ops, "attention", lambda q, k, v, p, seed=0, seed_epoch=None: _attention_reference(q, k, v, None, 0.0).float())
    y_ref, g_ref = run(hip)
    assert_close(y_hip, y_ref, rtol=1e-4, atol_scale=1e-5, what="encoder output, hip attention")
    assert_close(g_hip, g_ref, rtol=1e-3, atol_scale=1e-4, what="input gradient, hip attention")
    assert_close(y_tor, y_ref, rtol=2e-2, atol_scale=2e-3, what="encoder output, torch attention")
    assert_close(g_tor, g_ref, rtol=2e-2, atol_scale=2e-2, what="input gradient, torch attention")


@pytest.mark.parametrize("C", [48, 52, 32])
def test_radar_points_kernel_equals_the_torch_expression(C):
    """nr_radar_points_fwd/bwd against the expression it replaces (neuradar.py:470-476 + sine_position_embedding, the port of
    position_encoding_3d.py:56-103): points, embedding (channel split 16/16/16, 18/18/16, 12/10/10) and d points / d depth."""
    import math

    from neuradar_amd import ops
    from neuradar_amd.decoders import sine_position_embedding

    gen = torch.Generator().manual_seed(C)
    n = 2 * 1177
    depth = (torch.rand(n, generator=gen) * 120.0 + 0.5).to(DEV).requires_grad_(True)
    sph = torch.stack([(torch.rand(n, generator=gen) - 0.5) * 2.0, (torch.rand(n, generator=gen) - 0.5) * 0.5], 1).to(DEV)
    xyz, pos = ops.radar_points(depth, sph, C)
    d2 = depth.detach().clone().requires_grad_(True)
    theta, phi = sph[:, 1:2], sph[:, 0:1]
    ref = torch.cat((d2[:, None] * torch.cos(phi) * torch.cos(theta), d2[:, None] * torch.sin(phi) * torch.cos(theta),
                     d2[:, None] * torch.sin(theta)), dim=1)
    ref_pos = sine_position_embedding(ref.detach()[None], C)[0]
    assert torch.equal(xyz.detach(), ref.detach())
    # the embedding's argument reaches ~750 rad: sin / cos of identical float arguments, up to the libraries' last-place differences
    assert float((pos - ref_pos).abs().max()) <= 2e-6, float((pos - ref_pos).abs().max())
    assert not pos.requires_grad
    w = torch.randn(n, 3, generator=gen).to(DEV)
    (xyz * w).sum().backward()
    (ref * w).sum().backward()
    assert_close(depth.grad.cpu(), d2.grad.cpu(), rtol=1e-6, atol_scale=1e-6, what="d points / d depth")
    assert math.isfinite(float(pos.sum()))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("with_residual,relu", [(True, True), (False, True), (False, False)])
def test_bn_act_kernels_vs_torch_batch_norm(dtype, tol, with_residual, relu):
    """nr_bn_act_fwd/bwd (batch norm in training mode + residual + ReLU, channels-last) against torch.nn.functional.batch_norm
    + add + relu on the same tensors: output, running statistics, and the gradients w.r.t. input, residual, weight, bias.
    Odd pixel count (ragged last block), C = 32 like the decoder and C = 8."""
    import torch.nn.functional as F

    from neuradar_amd import ops

    for C, shape in ((32, (3, 32, 17, 19)), (8, (2, 8, 5, 7))):
        gen = torch.Generator().manual_seed(C)
        x = (torch.randn(shape, generator=gen) * 1.7 + 0.3).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
        res = torch.randn(shape, generator=gen).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last) if with_residual else None
        w = (torch.rand(C, generator=gen) + 0.5).to(DEV)
        b = torch.randn(C, generator=gen).to(DEV)
        gy = torch.randn(shape, generator=gen).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
        outs = []
        for which in ("hip", "torch"):
            xi = x.clone().requires_grad_(True)
            ri = res.clone().requires_grad_(True) if res is not None else None
            wi, bi = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            rm, rv = torch.full((C,), 0.25, device=DEV), torch.full((C,), 2.0, device=DEV)
            if which == "hip":
                y = ops.bn_act(xi, wi, bi, rm, rv, ri, 0.1, 1e-5, relu)
            else:  # the reference in fp32 on the same (16-bit representable) inputs
                y = F.batch_norm(xi.float(), rm, rv, wi, bi, True, 0.1, 1e-5)
                if ri is not None:
                    y = y + ri.float()
                y = torch.relu(y) if relu else y
            y.backward(gy.to(y.dtype))
            outs.append((y.detach().float(), rm, rv, xi.grad.float(), None if ri is None else ri.grad.float(), wi.grad, bi.grad))
        h, t = outs
        assert h[0].is_contiguous(memory_format=torch.channels_last)
        assert_close(h[0].cpu(), t[0].cpu(), rtol=tol, atol_scale=tol, what="output")
        assert_close(h[1].cpu(), t[1].cpu(), rtol=1e-5, atol_scale=1e-6, what="running mean")
        assert_close(h[2].cpu(), t[2].cpu(), rtol=1e-4, atol_scale=1e-5, what="running var")
        # (16-bit: the mask y > 0 is taken on the rounded output, and gradients are rounded to the activation type)
        assert_close(h[3].cpu(), t[3].cpu(), rtol=tol, atol_scale=4 * tol, what="d input")
        if with_residual:
            assert_close(h[4].cpu(), t[4].cpu(), rtol=tol, atol_scale=4 * tol, what="d residual")
        assert_close(h[5].cpu(), t[5].cpu(), rtol=max(tol, 1e-4), atol_scale=4 * tol, what="d weight")
        assert_close(h[6].cpu(), t[6].cpu(), rtol=max(tol, 1e-4), atol_scale=4 * tol, what="d bias")


@pytest.mark.parametrize("C,n", [(48, 3531), (64, 130), (48, 1)])
def test_radar_heads_kernels_vs_the_three_mlp_heads(C, n, monkeypatch):
    """nr_radar_heads_fwd/bwd against the expression it replaces (three MLP heads on nr_mlp_fwd/bwd + tanh / sigmoid / softplus +
    concatenation, neuradar.py:480-491): radar_output and the gradients w.r.t. the transformer output, the points and all 18
    parameter tensors; ray counts that are not multiples of the block's 64."""
    from neuradar_amd.decoders import Decoders

    torch.manual_seed(C + n)
    dec = Decoders(C).to(DEV).train()
    with torch.no_grad():  # away from the small default initialisation: saturating activations included
        for head in (dec.offset_head, dec.existence_probability_head, dec.radar_uncertainty_head):
            for p in head.parameters():
                p.mul_(4.0)
    x0 = torch.randn(n, C, device=DEV)
    xyz0 = torch.randn(n, 3, device=DEV) * 30.0
    gy = torch.randn(n, 7, device=DEV)
    heads = (dec.offset_head, dec.existence_probability_head, dec.radar_uncertainty_head)
    params = [p for h in heads for p in h.parameters()]
    res = []
    for fused in (True, False):
        x, xyz = x0.clone().requires_grad_(True), xyz0.clone().requires_grad_(True)
        if fused:
            from neuradar_amd import ops

            y = ops.radar_heads(x, xyz, *heads)
        else:
            y = torch.cat((dec.existence_probability_head(x), xyz + 1.5 * dec.offset_head(x), dec.radar_uncertainty_head(x)), dim=-1)
        grads = torch.autograd.grad((y * gy).sum(), [x, xyz] + params)
        res.append((y.detach(), grads))
    (ya, ga), (yb, gb) = res
    assert_close(ya.cpu(), yb.cpu(), rtol=1e-5, atol_scale=1e-6, what="radar_output")
    for i, (a, b) in enumerate(zip(ga, gb)):
        assert_close(a.cpu(), b.cpu(), rtol=1e-4, atol_scale=1e-5, what=f"gradient {i}")
