"""oracle/decoder_losses.py and the "nll" additions of oracle/radar.py against vectors produced by the reference
NeuRadarModel's own training-branch methods (tests/golden/model_train.npz, make_golden.py::golden_model_train)."""
import pytest
import torch

from helpers import assert_close, load_golden


def _params(need_fingerprint=None):
    g = load_golden("model")
    p = {k[len("param."):]: v for k, v in g.items() if k.startswith("param.")}
    if need_fingerprint is not None:  # model_train.npz stores only sums of the parameters it was generated with
        for k, v in need_fingerprint.items():
            if k.startswith("param_sum."):
                assert abs(float(p[k[len("param_sum."):]].double().sum()) - float(v)) <= 1e-9 * max(1.0, abs(float(v))), k
    return p


def _inputs(g):
    lid = g["is_lidar"][:, 0]
    batch = {"image": g["image"], "distance": g["distance"], "did_return": g["did_return"], "lidar": g["lidar"],
             "radar": g["radar"], "radar_indices": g["radar_indices"]}
    return lid, batch


@pytest.mark.parametrize("loss_type", ["nll", "euclidean"])
def test_decoder_losses_and_gradients_match_the_reference_training_branch(loss_type):
    from oracle import decoder_losses as odl

    g = load_golden("model_train")
    p = {k: v.clone().requires_grad_(True) for k, v in _params(g).items() if torch.is_tensor(v) and v.dtype == torch.float32 and "running" not in k}
    feats, depth = g["features"].clone().requires_grad_(True), g["depth"].clone().requires_grad_(True)
    pds = [g[f"prop_depth_{i}"].clone().requires_grad_(True) for i in range(2)]
    _, batch = _inputs(g)
    out = odl.decoder_losses(feats, depth, pds, g["spher"], g["is_lidar"], g["is_radar"], batch, p, int(g["patch"]), int(g["n_scan"]),
                             odl.LossSettings(radar_loss_type=loss_type))
    t = loss_type + "."
    assert_close(out["rgb"], g[t + "rgb"], rtol=1e-4, atol_scale=1e-5, what="rgb (training-mode batch norm)")
    assert_close(out["intensity"], g[t + "intensity"], what="intensity")
    assert_close(out["ray_drop_logits"], g[t + "ray_drop_logits"], what="ray drop logits")
    assert_close(out["radar_output"], g[t + "radar_output"], rtol=1e-4, atol_scale=1e-5, what="radar output")
    for k in odl.LOSS_KEYS:
        assert_close(out[k], g[t + "loss." + k], rtol=1e-4, atol_scale=1e-6, what=k)
    for i in range(int(g["n_scan"])):
        assert torch.equal(out["assoc"][i], g[t + f"assoc_{i}"]), f"association of scan {i}"
    tot = odl.total(out)
    assert_close(tot, g[t + "total"], rtol=1e-4, what="total")
    names = [k for k in p if (t + "gsum." + k) in g]
    grads = torch.autograd.grad(tot, [feats, depth, *pds] + [p[k] for k in names])
    assert_close(grads[0], g[t + "g_features"], rtol=1e-3, atol_scale=1e-4, what="d total / d features")
    assert_close(grads[1], g[t + "g_depth"], rtol=1e-3, atol_scale=1e-5, what="d total / d depth")
    assert_close(grads[2], g[t + "g_prop_depth_0"], what="d total / d prop depth 0")
    assert_close(grads[3], g[t + "g_prop_depth_1"], what="d total / d prop depth 1")
    assert len(names) > 40
    for k, gr in zip(names, grads[4:]):
        scale = float(g[t + "gabs." + k])  # (0 for the uncertainty head under the euclidean loss)
        assert abs(float(gr.double().sum()) - float(g[t + "gsum." + k])) <= 2e-4 * scale, k
        assert abs(float(gr.double().abs().sum()) - scale) <= 2e-4 * scale, k
        if (t + "grad." + k) in g:
            assert_close(gr, g[t + "grad." + k], rtol=2e-3, atol_scale=2e-4, what="grad " + k)


def test_nll_cost_matrix_scan_losses_and_sampling_match_the_reference():
    from oracle import radar as orad

    g = load_golden("model_train")
    seg = [0, 27, 28]
    for lt in ("nll", "euclidean"):
        for i in range(2):
            gt = g["radar"][seg[i]:seg[i + 1], :3]
            mb = orad.multi_bernoulli(g[lt + ".radar_output"][i])
            cost = orad.cost_matrix(gt, mb, "euclidean")
            assert_close(cost, g[lt + f".cost_{i}"], rtol=1e-5, atol_scale=1e-6, what="training cost (euclidean)")
            assoc = orad.hungarian(cost)
            assert torch.equal(assoc, g[lt + f".assoc_{i}"])
            assert_close(orad.scan_loss(gt, mb, assoc, lt), g[lt + f".scan_loss_{i}"], rtol=1e-5, what=f"{lt} scan loss {i}")
    gt = g["radar"][:27, :3]
    assert_close(orad.cost_matrix(gt, orad.multi_bernoulli(g["nll.radar_output"][0]), "nll"), g["nll.evalcost_0"], rtol=1e-5,
                 atol_scale=1e-6, what="evaluation cost (nll)")
    torch.manual_seed(int(g["sample_seed"]))
    pts, idx = orad.sample_radar_points_nll(g["sample_radar_output"], int(g["sample_max_detections"]))
    assert torch.equal(idx, g["sample_ber"]) and idx.numel() > 3
    assert_close(pts, g["sample_points"], rtol=1e-6, what="sampled detections (nll)")
