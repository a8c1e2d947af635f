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
