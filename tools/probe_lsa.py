"""nr_radar_assign alone (cost matrix + linear sum assignment) on a spread and on a degenerate (all predictions in one small
cluster: the freshly initialised model) scan, against scipy on the host (development tool)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuradar_amd import ops  # noqa: E402
from oracle import radar as orad  # noqa: E402

dev = "cuda"
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 3531, int(sys.argv[2]) if len(sys.argv) > 2 else 200
gen = torch.Generator().manual_seed(0)
det = torch.cat([torch.randn(m, 3, generator=gen) * 25.0 + torch.tensor([40.0, 0.0, 0.0]), torch.rand(m, 2, generator=gen)], 1)
seg = torch.tensor([0, m], dtype=torch.int32, device=dev)
for tag, spread in (("spread", 25.0), ("cluster 1 m", 0.5), ("cluster 1 cm", 0.005)):
    pred = torch.cat([torch.sigmoid(0.1 * torch.randn(1, n, 1, generator=gen)), torch.randn(1, n, 3, generator=gen) * spread + torch.tensor([40.0 if spread > 1 else 1.0, 0.0, 0.0]),
                      torch.rand(1, n, 3, generator=gen) + 0.1], dim=-1)
    p_d, d_d = pred.to(dev), det.to(dev)
    ws = torch.empty(ops._lib.lib().nr_radar_assign_workspace_bytes(1, n, m), device=dev, dtype=torch.uint8)
    for _ in range(2):
        assoc = ops.radar_assign(p_d, d_d, seg, m, "euclidean", ws)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        assoc = ops.radar_assign(p_d, d_d, seg, m, "euclidean", ws)
    b.record()
    torch.cuda.synchronize()
    cost = orad.cost_matrix(det[:, :3], orad.multi_bernoulli(pred[0]), "euclidean").double()
    t0 = time.perf_counter()
    want = orad.hungarian(cost)
    t_cpu = time.perf_counter() - t0
    got = assoc[0].cpu().long()
    c_got, c_want = float(cost[got >= 0, got[got >= 0]].sum()), float(cost[want >= 0, want[want >= 0]].sum())
    print(f"{tag:14s} n={n} m={m}: device {a.elapsed_time(b) / 5 * 1e3:9.1f} us   scipy {t_cpu * 1e3:7.2f} ms   cost {c_got:.6f} vs {c_want:.6f}   "
          f"mismatches {int((got != want).sum())}", flush=True)
