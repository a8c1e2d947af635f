"""The oracle pieces that round 1 left unpinned, against vectors produced by the reference NeuRadarModel's OWN methods
(tests/golden/make_golden.py::golden_model): appearance embedding (a19), is_close_to_lidar (a20), lidar decoder and
lidar losses, radar transformer + heads and RGB CNN (f-2), Hungarian-matched radar loss, sampled detections, Chamfer."""
import numpy as np
import torch

from helpers import assert_close, load_golden
from oracle import decoders, pipeline, radar


def _params(g):
    return {k[len("param."):]: v for k, v in g.items() if k.startswith("param.") and isinstance(v, torch.Tensor)}


def test_appearance_embedding_and_lidar_masks_vs_reference_model():
    g = load_golden("model")
    E = g["app_table"].shape[0] // 3
    got = pipeline.appearance_embedding(g["app_table"], g["app_times"], g["app_sensor"], 20.0, E)
    assert_close(got, g["app_embed"], rtol=1e-6, atol_scale=1e-7, what="appearance embedding")
    e = g["close_edges"]
    mask = pipeline.is_close_to_lidar(e[:, :-1], e[:, 1:], g["close_is_lidar"], g["close_dist"], g["close_did_return"])
    assert torch.equal(mask, g["close_mask"][..., 0]) and bool(mask.any()) and not bool(mask.all())


def test_lidar_decoder_and_losses_vs_reference_model():
    g = load_golden("model")
    p = _params(g)
    ws = [p[f"lidar_decoder.layers.{i}.weight"] for i in range(3)]
    bs = [p[f"lidar_decoder.layers.{i}.bias"] for i in range(3)]
    intensity, drop = decoders.lidar_decode(g["dec_features"], g["dec_is_lidar"], ws, bs)
    assert_close(intensity, g["dec_intensity"], rtol=1e-5, atol_scale=1e-6, what="intensity")
    assert_close(drop, g["dec_ray_drop_logit"], rtol=1e-5, atol_scale=1e-6, what="ray drop logit")
    is_l = g["ll_is_lidar"][:, 0]
    did = g["ll_did_return"][is_l][:, 0]
    out = decoders.lidar_losses(g["ll_depth"][is_l], g["ll_intensity"], g["ll_ray_drop_logits"], g["ll_distance"], did,
                                g["ll_points"][:, 3:4])
    for k in ("depth_loss", "intensity_loss", "ray_drop_loss"):
        assert abs(float(out[k]) - g["ll_metric." + k]) <= 1e-6 * abs(g["ll_metric." + k]) + 1e-7, k
    assert abs(float((g["ll_non_nearby"] ** 2).sum() / is_l.sum()) - g["ll_metric.carving_loss"]) < 1e-6  # neuradar.py:637-638
    for k, mult in (("depth_loss", 0.01), ("intensity_loss", 0.1), ("ray_drop_loss", 0.01), ("carving_loss", 0.01)):  # :80-110,690-700
        assert abs(g["ll_loss." + k] - mult * g["ll_metric." + k]) < 1e-7 * max(1.0, abs(g["ll_loss." + k]))


def test_radar_decoder_and_cnn_vs_reference_model():
    g = load_golden("model")
    p = {k: v.clone().requires_grad_(v.dtype == torch.float32 and "running" not in k) for k, v in _params(g).items()}
    is_r = g["dec_is_radar"][:, 0]
    ro = radar.decode_radar(g["dec_features"][is_r], g["dec_depth"][is_r], g["dec_spher"][is_r], 2, p)
    assert_close(ro.detach(), g["dec_radar_output"], rtol=1e-4, atol_scale=1e-5, what="radar_output")
    keys = [k[len("dec_grad."):] for k in g if k.startswith("dec_grad.")]
    grads = torch.autograd.grad((ro * g["dec_g_radar_output"]).sum(), [p[k] for k in keys])
    for k, gr in zip(keys, grads):
        assert_close(gr, g["dec_grad." + k], rtol=1e-3, atol_scale=1e-4, what="grad " + k)
    cam = ~(g["dec_is_lidar"][:, 0] | is_r)
    rgb = radar.rgb_decode(g["dec_features"][cam], (8, 8), {k: v.detach() for k, v in p.items()})
    assert_close(rgb, g["dec_rgb"], rtol=1e-4, atol_scale=1e-5, what="rgb")


def test_radar_loss_points_and_chamfer_vs_reference_model():
    g = load_golden("model")
    loss, assoc = radar.radar_loss_euclidean(g["radar_batch"], g["dec_radar_output"], g["radar_indices"])
    assert abs(float(loss) - g["radar_loss"]) < 1e-5 * g["radar_loss"]
    assert torch.equal(assoc, g["radar_assoc_last"])
    pts, ber = radar.sample_radar_points(g["cd_radar_output"], 0.5)
    assert torch.equal(ber, g["cd_ber"]) and torch.equal(pts, g["cd_points"])
    cd = radar.chamfer_distance(pts.numpy(), g["cd_gt"].numpy())
    assert abs(cd - g["cd_value"]) < 1e-6 * g["cd_value"]
    assert radar.chamfer_distance(np.zeros((3, 3)), np.zeros((5, 3))) == 0.0
