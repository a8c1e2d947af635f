"""GPU probe: per LEVEL, what the three table scatters of the mixed step have to do and how long the kernels take for it.
Positions and gradients are the fused step's own after PROBE_STEPS training steps.  Per grid, level and row segment
(coherent = camera + radar rows, lidar rows): samples with a non-zero gradient, distinct cells / table entries / 64-B lines
they touch (the lower bounds of any scatter), and the time of the merging kernel (and of the binned kernels) on that level
alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd import ops  # noqa: E402
from neuradar_amd.parallel import GradAllReducer  # noqa: E402
from neuradar_amd.step import FlatAdam  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mixed16384_neuradar"
wl = bench.WORKLOADS[name]
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
mixed = wl if "cam_rays" in wl else None
fwd_bwd, optim, st = bench.make_step(model, scene, opts, red, targets, n_rays, fused=True, fuse_optimizer=True, mixed=mixed)
for _ in range(int(os.environ.get("PROBE_STEPS", "5"))):
    fwd_bwd()
torch.cuda.synchronize()
lib, p, s = st.lib, ops._p, ops._stream
B = n_rays
P1, P2 = 2654435761, 805459861


def distinct(x, scale, T, F, keep):
    pos = x[keep].double() * scale
    lo = torch.floor(pos).long()
    cells = torch.unique(lo, dim=0).shape[0]
    ent = []
    for c in range(8):
        ix, iy, iz = lo[:, 0] + (c & 1), lo[:, 1] + ((c >> 1) & 1), lo[:, 2] + ((c >> 2) & 1)
        ent.append(((ix & 0xFFFFFFFF) ^ ((iy * P1) & 0xFFFFFFFF) ^ ((iz * P2) & 0xFFFFFFFF)) & (T - 1))
    ent = torch.unique(torch.cat(ent))
    lines = torch.unique(ent * F * 4 // 64).numel()
    return cells, ent.numel(), lines


for lvl, grid, tag in ((0, st.pgrid, "prop_s128"), (1, st.pgrid, "prop_s64"), (2, st.mgrid, "main_s32")):
    S, F, L, log2T = st.S[lvl], grid.features_per_level, grid.num_levels, grid.log2_hashmap_size
    T = 1 << log2T
    nl, n_coh = B * S, st.sm * S
    x, sd, g = st.x01[lvl], st.std[lvl], st.g_feats[lvl]
    gt = torch.zeros_like(grid.hash_table)
    print(f"== {tag}: {nl} rows ({n_coh} coherent), L={L} F={F} T=2^{log2T}")
    for l in range(L):
        sc = grid.scalings[l:l + 1].contiguous()
        gl = g[l:l + 1]
        gtl = gt[l * T:(l + 1) * T]
        for seg, r0, n in (("coh", 0, n_coh), ("lid", n_coh, nl - n_coh), ("all", 0, nl)):
            if n <= 0:
                continue
            nz = (gl[0, r0:r0 + n] != 0).any(-1)
            if int(nz.sum()) == 0:
                print(f"  level {l} {seg}: all gradients zero")
                continue
            cells, ents, lines = distinct(x[r0:r0 + n], float(sc[0]), T, F, nz)
            t = bench.time_kernel(lambda: lib.nr_hash_encode_bwd(p(x[r0:]), p(sd[r0:]), p(sc), 1, F, log2T, p(gl[:, r0:, :]), F, nl * F,
                                                                 p(gtl), n, 0, s()), 10)
            tb = float("nan")
            need = lib.nr_hash_encode_bwd_binned_workspace_bytes(1, F, log2T, n)
            if need > 0:
                ws = torch.zeros(need, device=dev, dtype=torch.uint8)
                if os.environ.get("PROBE_STATS"):
                    lib.nr_hash_encode_bwd_binned(p(x[r0:]), p(sd[r0:]), p(sc), 1, F, log2T, p(gl[:, r0:, :]), F, nl * F, p(gtl), n, p(ws), s())
                    torch.cuda.synchronize()
                    st_ = ws[-256:].view(torch.int32)
                    print(f"      binned stats: fallback pairs {int(st_[0])}, records written {int(st_[32])} of {4 * int(nz.sum())} pairs")
                    for w0 in (40, 48):
                        it = max(int(st_[w0 + 7]), 1)
                        print("      cycles/iteration (block 7, wave %d): compute %d | barrier A %d | CAS %d | adds %d | barrier B %d | flush %d  (%d iterations)"
                              % ((w0 - 40) // 8 * 7, *[int(st_[w0 + k]) // it for k in range(6)], it))
                tb = bench.time_kernel(lambda: lib.nr_hash_encode_bwd_binned(p(x[r0:]), p(sd[r0:]), p(sc), 1, F, log2T, p(gl[:, r0:, :]), F,
                                                                             nl * F, p(gtl), n, p(ws), s()), 10)
            print(f"  level {l} scale {float(sc[0]):6.0f} {seg}: rows {n:8d} nonzero {int(nz.sum()):8d} cells {cells:8d} entries {ents:8d} "
                  f"lines {lines:8d} | merging {t * 1e6:7.1f} us ({lines / t / 1e9:5.1f} G lines/s)  binned {tb * 1e6:7.1f} us")
