"""GPU probe: the main grid's scatter (wide merge tables) ALONE on the mixed step's own positions / gradients, per level and
over all levels, for PMC passes over the L2's atomic path (tools/pmc_main_atomics.sh): where the float atomics execute
(L2 vs forwarded to the memory side), hit / miss, stalls."""
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
model = bench.build_model(wl, dev, "bfloat16")
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
grid = st.mgrid
S, F, L, T = st.S[2], grid.features_per_level, grid.num_levels, grid.log2_hashmap_size
nl = n_rays * S
x, sd, g = st.x01[2], st.std[2], st.g_feats[2].clone()
gt = torch.zeros_like(grid.hash_table)
iters = int(os.environ.get("PROBE_ITERS", "10"))


def launch(l0, nlev):
    return lambda: lib.nr_hash_encode_bwd_tuned(p(x), p(sd), grid.scalings.data_ptr() + 4 * l0, nlev, F, T, g.data_ptr() + 4 * l0 * nl * F, F,
                                                nl * F, gt.data_ptr() + 4 * l0 * (1 << T) * F, nl, 0, 256, s())


print(f"all {L} levels: {bench.time_kernel(launch(0, L), iters) * 1e6:7.1f} us")
for l0 in range(L):
    print(f"level {l0} (scale {float(grid.scalings[l0]):6.0f}): {bench.time_kernel(launch(l0, 1), iters) * 1e6:7.1f} us")
