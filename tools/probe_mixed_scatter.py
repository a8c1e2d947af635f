"""GPU probe: where the proposal-grid scatter of the mixed (camera + radar + lidar) batch spends its time -- the merging
kernel on the coherent rows, on the lidar rows, and the binned kernels on the lidar rows, on the bench's own positions
and gradients (after a few steps of the fused step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd import ops  # noqa: E402
from neuradar_amd.parallel import GradAllReducer  # noqa: E402
from neuradar_amd.step import FlatAdam  # noqa: E402

wl = bench.WORKLOADS["mixed16384_neuradar"]
dev = torch.device("cuda")
model = bench.build_model(wl, dev)
groups = model.get_param_groups()
unused = list(model.proposal_fields[0].parameters())
opts = [FlatAdam(groups["hashgrids"], lr=1e-2, eps=1e-15, lr_final=1e-3, skip=unused),
        FlatAdam(groups["fields"], lr=1e-2, eps=1e-15, weight_decay=1e-7, adamw=True, lr_final=1e-3, skip=unused)]
red = GradAllReducer(None, buffers=[g for o in opts for g in o.grad_buffers()])
scene = bench.SyntheticScene(dev, seed=1000)
n_rays = wl["rays"]
targets = (0.1 * torch.randn(n_rays, 32, device=dev), 5.0 + 50.0 * torch.rand(n_rays, 1, device=dev))
fwd_bwd, optim, st = bench.make_step(model, scene, opts, red, targets, n_rays, fused=True, fuse_optimizer=True, mixed=wl)
for _ in range(int(os.environ.get("PROBE_STEPS", "5"))):
    fwd_bwd()
torch.cuda.synchronize()
lib, p, s = st.lib, ops._p, ops._stream
B = n_rays
for lvl, grid, tag in ((0, st.pgrid, "prop_s128"), (1, st.pgrid, "prop_s64"), (2, st.mgrid, "main_s32")):
    S, F, L, T = st.S[lvl], grid.features_per_level, grid.num_levels, grid.log2_hashmap_size
    nl, n_coh = B * S, st.sm * S
    x, sd, g = st.x01[lvl], st.std[lvl], st.g_feats[lvl]
    gt = torch.zeros_like(grid.hash_table)

    def merging(r0, n):
        return lambda: lib.nr_hash_encode_bwd(p(x[r0:]), p(sd[r0:]), p(grid.scalings), L, F, T, p(g[:, r0:, :]), F, nl * F, p(gt), n, 0, s())

    t_coh = bench.time_kernel(merging(0, n_coh), 10)
    t_lid = bench.time_kernel(merging(n_coh, nl - n_coh), 10)
    t_all = bench.time_kernel(merging(0, nl), 10)
    need = lib.nr_hash_encode_bwd_binned_workspace_bytes(L, F, T, nl)
    t_bin = t_bin_all = t_bin_coh = float("nan")
    if need > 0:
        ws = torch.empty(need, device=dev, dtype=torch.uint8)

        def binned(r0, n):
            return lambda: lib.nr_hash_encode_bwd_binned(p(x[r0:]), p(sd[r0:]), p(grid.scalings), L, F, T, p(g[:, r0:, :]), F, nl * F,
                                                         p(gt), n, p(ws), s())

        t_bin = bench.time_kernel(binned(n_coh, nl - n_coh), 10)
        t_bin_coh = bench.time_kernel(binned(0, n_coh), 10)
        t_bin_all = bench.time_kernel(binned(0, nl), 10)
    zero = float((g[:, n_coh:, :] == 0).float().mean())
    print(f"{tag}: merging all {t_all * 1e6:7.1f} us = coherent rows {t_coh * 1e6:7.1f} + lidar rows {t_lid * 1e6:7.1f}; "
          f"binned: all {t_bin_all * 1e6:7.1f} = coherent {t_bin_coh * 1e6:7.1f} + lidar {t_bin * 1e6:7.1f} us; zero gradients among lidar rows {zero:.2f}")
