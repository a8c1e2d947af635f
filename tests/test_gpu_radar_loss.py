"""Radar point-set loss on the device (nr_radar_assign / nr_radar_loss, radar.hip) against the reference's vectors
(tests/golden/model_train.npz: radar_utils.calculate_radar_loss per scan), the oracle and scipy's Hungarian solver."""
import numpy as np
import pytest
import torch

from helpers import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _seg(counts):
    return torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=DEV)


@pytest.mark.parametrize("loss_type", ["nll", "euclidean"])
def test_radar_loss_and_association_vs_reference_golden(loss_type):
    from neuradar_amd import ops

    g = load_golden("model_train")
    t = loss_type + "."
    pred = g[t + "radar_output"].to(DEV).requires_grad_(True)
    det = g["radar"].to(DEV)
    loss, assoc = ops.radar_loss(pred, det, _seg([27, 1]), 32, loss_type, mult=0.02)
    for i in range(2):
        assert torch.equal(assoc[i].cpu().long(), g[t + f"assoc_{i}"]), f"association of scan {i}"
    assert_close(loss.detach().cpu(), g[t + "loss.radar_loss"], rtol=1e-5, what="radar loss")
    # gradient against the oracle's autograd on the same association
    from oracle import radar as orad

    ref = g[t + "radar_output"].clone().requires_grad_(True)
    l_ref, _ = orad.radar_loss(g["radar"], ref, g["radar_indices"], loss_type, True)
    (0.02 * l_ref).backward()
    loss.backward()
    assert_close(pred.grad.cpu(), ref.grad, rtol=1e-4, atol_scale=1e-6, what="d loss / d radar_output")


@pytest.mark.parametrize("n,counts", [(3531, [180]), (4545, [260, 1, 0, 97]), (64, [64, 63]), (40, [70]), (1, [1]), (300, [2, 299])])
@pytest.mark.parametrize("cost_type", ["euclidean", "nll"])
def test_assignment_is_scipys_optimum(n, counts, cost_type):
    """Full-size scans (ZOD 107 x 33, VoD 101 x 45 rays), more detections than predictions (transposed search), empty and
    single-detection scans: the same assignment as scipy.optimize.linear_sum_assignment on the oracle's cost matrix."""
    from neuradar_amd import ops
    from oracle import radar as orad

    gen = torch.Generator().manual_seed(n + sum(counts))
    N = len(counts)
    pred = torch.cat([torch.rand(N, n, 1, generator=gen), torch.randn(N, n, 3, generator=gen) * 20.0 + torch.tensor([30.0, 0.0, 0.0]),
                      torch.rand(N, n, 3, generator=gen) * 2.0 + 1e-4], dim=-1)
    pred[:, : max(1, n // 50), 0] = 0.0  # clamped existence probabilities
    det = torch.cat([torch.randn(sum(counts), 3, generator=gen) * 20.0 + torch.tensor([30.0, 0.0, 0.0]), torch.rand(sum(counts), 2, generator=gen)], 1)
    seg = np.concatenate([[0], np.cumsum(counts)])
    assoc = ops.radar_assign(pred.to(DEV), det.to(DEV), _seg(counts), max(counts), cost_type).cpu().long()
    for i, m in enumerate(counts):
        gt = det[seg[i]:seg[i + 1], :3]
        if m == 0:
            assert bool((assoc[i] == -1).all())
            continue
        cost = orad.cost_matrix(gt, orad.multi_bernoulli(pred[i]), cost_type).double()
        want = orad.hungarian(cost)
        got = assoc[i]
        assert int((got >= 0).sum()) == min(m, n) and len(set(got[got >= 0].tolist())) == min(m, n)
        c_got = float(cost[got >= 0, got[got >= 0]].sum())
        c_want = float(cost[want >= 0, want[want >= 0]].sum())
        assert c_got <= c_want * (1 + 1e-6) + 1e-6, (c_got, c_want)  # optimal (the kernel's float32 cost entries differ in the last bits)
        assert float((got != want).float().mean()) <= 2e-3, "assignment differs from scipy's beyond near-ties"


def test_assignment_status_words_report_an_unassigned_scan():
    """The per-scan status words nr_radar_assign leaves in its workspace (include/neuradar_hip.h): 0 for assigned scans, 2 for a
    scan with more detections than max_detections -- it is left unassigned (assoc = -1), which the callers must be able to see
    (scipy has no such limit); ops.validate_radar_segments rejects such a batch before the step."""
    from neuradar_amd import ops

    gen = torch.Generator().manual_seed(11)
    n, counts = 300, [40, 90, 7]
    pred = torch.cat([torch.rand(3, n, 1, generator=gen), torch.randn(3, n, 3, generator=gen) * 20.0, torch.rand(3, n, 3, generator=gen) + 1e-4], dim=-1)
    det = torch.randn(sum(counts), 5, generator=gen) * 20.0
    assoc = ops.radar_assign(pred.to(DEV), det.to(DEV), _seg(counts), 90, "euclidean")
    assert assoc.status.cpu().tolist() == [0, 0, 0] and int((assoc.cpu() >= 0).sum()) == sum(counts)
    assoc = ops.radar_assign(pred.to(DEV), det.to(DEV), _seg(counts), 64, "euclidean")  # scan 1 has 90 > 64 detections
    assert assoc.status.cpu().tolist() == [0, 2, 0]
    a = assoc.cpu()
    assert bool((a[1] == -1).all()) and int((a[0] >= 0).sum()) == 40 and int((a[2] >= 0).sum()) == 7
    ops.validate_radar_segments(_seg(counts), 90)
    with pytest.raises(ValueError):
        ops.validate_radar_segments(_seg(counts), 64)
