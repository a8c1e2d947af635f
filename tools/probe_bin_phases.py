"""GPU probe: wave-cycles per phase of the bin pass, in the step and alone.  Needs the instrumented build:
`make -C neuradar_amd/csrc EXTRA=-DNR_BIN_CLOCKS && cp libneuradar_hip.so lib_clk.so`, then NR_LIB_PATH=.../lib_clk.so."""
import ctypes
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
lib = st.lib
buf = (ctypes.c_ulonglong * 8)()
names = ["tile top (copy prefetch, issue next)", "rows -> pairs, scan, max", "barrier A", "insert (LDS atomics)", "barrier B", "flush", "-", "-"]


def report(tag, launches):
    lib.nr_debug_bin_clocks(buf, 1)
    v = list(buf)
    tot = sum(v)
    print(f"{tag}: {tot / 1e6:.1f} M wave-cycles over {launches} launches")
    for n, x in zip(names[:6], v[:6]):
        print(f"    {n:40s} {x / tot * 100:5.1f} %   {x / launches / 4096:9.0f} cycles per wave and launch")


for _ in range(5):
    fwd_bwd()
lib.nr_debug_bin_clocks(buf, 1)
for _ in range(10):
    fwd_bwd()
report("in the step", 20)
# alone: the S=128 chain's scatter with the density head, on the step's own buffers
p, s = ops._p, ops._stream
pg, w_dec = st.pgrid, st.prop.density_decoder.weight
lvl = 0
S, nl, Fg = st.S[lvl], n_rays * st.S[lvl], pg.features_per_level
gt, gw = torch.zeros_like(pg.hash_table), torch.zeros_like(w_dec)
torch.cuda.synchronize()
for _ in range(10):
    lib.nr_prop_density_scatter_binned(p(st.x01[lvl]), p(st.std[lvl]), p(pg.scalings), pg.num_levels, Fg, pg.log2_hashmap_size,
                                       p(st.feats[lvl]), Fg, nl * Fg, p(w_dec), p(st.g_dens[lvl]), S, st.sm, p(gt), p(gw), nl,
                                       p(st.binned_ws[lvl]), s())
report("S=128 chain alone", 10)
