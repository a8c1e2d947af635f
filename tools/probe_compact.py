"""GPU probe: nr_grad_compact / nr_grad_apply on a main-table-sized gradient with ~1 % non-zero rows (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd import ops  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
for rows, F, nnz in ((16 << 19, 2, 75_000), (8 << 22, 4, 150_000), (16 << 19, 2, 600_000)):
    grad = torch.zeros(rows, F, device=dev)
    hit = torch.randperm(rows, device=dev)[:nnz]
    vals = torch.randn(nnz, F, device=dev)
    cap = rows // 16
    idx = torch.zeros(cap, dtype=torch.int32, device=dev)
    val = torch.zeros(cap * F, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)

    def compact():
        grad[hit] = vals  # refill (timed separately below)
        count.zero_()
        ops.grad_compact(grad.view(-1), F, idx, val, count)

    def refill():
        grad[hit] = vals
        count.zero_()

    t_all = bench.time_kernel(compact, 10)
    t_fill = bench.time_kernel(refill, 10)
    compact()
    t_apply = bench.time_kernel(lambda: ops.grad_apply(idx, val, count, F, grad.view(-1)), 10)
    print(f"rows {rows} F {F} nnz {nnz}: compact {(t_all - t_fill) * 1e6:8.1f} us   apply {t_apply * 1e6:7.1f} us   (count {int(count)})")
