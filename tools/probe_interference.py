"""GPU probe: how much the proposal scatter (bin + apply) of the mixed step slows down next to the kernels it shares the
chip with inside the step -- the other proposal scatter, the main grid's merging scatter, an HBM-bound stream (Adam's
shape) -- each on a stream of its own.  Positions and gradients are the step's own after PROBE_STEPS training steps."""
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
for _ in range(int(os.environ.get("PROBE_STEPS", "300"))):
    fwd_bwd()
torch.cuda.synchronize()
lib, p = st.lib, ops._p
B = n_rays
streams = [torch.cuda.Stream() for _ in range(4)]


def scatter_fn(lvl, grid, stream, binned):
    S, F, L, T = st.S[lvl], grid.features_per_level, grid.num_levels, grid.log2_hashmap_size
    nl = B * S
    x, sd, g = st.x01[lvl], st.std[lvl], st.g_feats[lvl]
    gt = torch.zeros_like(grid.hash_table)
    if binned:
        ws = torch.empty(lib.nr_hash_encode_bwd_binned_workspace_bytes(L, F, T, nl), device=dev, dtype=torch.uint8)
        return lambda: lib.nr_hash_encode_bwd_binned(p(x), p(sd), p(grid.scalings), L, F, T, p(g), F, nl * F, p(gt), nl, p(ws), stream.cuda_stream)
    return lambda: lib.nr_hash_encode_bwd(p(x), p(sd), p(grid.scalings), L, F, T, p(g), F, nl * F, p(gt), nl, 0, stream.cuda_stream)


s128 = scatter_fn(0, st.pgrid, streams[0], True)
s64 = scatter_fn(1, st.pgrid, streams[1], True)
main = scatter_fn(2, st.mgrid, streams[2], False)
big = [torch.empty(1 << 28, device=dev) for _ in range(3)]  # 1 GiB each


def hbm():
    with torch.cuda.stream(streams[3]):
        torch.add(big[0], big[1], out=big[2])  # 3 GiB of traffic, ~0.5 ms


def timed(subject, stream, others, iters=5):
    """elapsed of `subject` (on `stream`) while `others` are started just before it on their own streams"""
    tot = 0.0
    for _ in range(iters + 1):
        torch.cuda.synchronize()
        for o in others:
            o()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        subject()
        b.record(stream)
        torch.cuda.synchronize()
        tot += a.elapsed_time(b) if _ else 0.0
    return tot / iters * 1e3


for name, others in (("alone", []), ("+ prop_s64 scatter", [s64]), ("+ main scatter", [main]), ("+ HBM stream", [hbm]),
                     ("+ HBM stream x2", [hbm, hbm]), ("+ prop_s64 + main", [s64, main]), ("+ prop_s64 + main + HBM", [s64, main, hbm])):
    print(f"prop_s128 binned scatter {name:28s}: {timed(s128, streams[0], others):8.1f} us")
print(f"main scatter alone: {timed(main, streams[2], []):8.1f} us;  + HBM stream: {timed(main, streams[2], [hbm]):8.1f} us;  "
      f"+ both proposal scatters: {timed(main, streams[2], [s128, s64]):8.1f} us")
print(f"HBM stream alone: {timed(hbm, streams[3], []):8.1f} us")
