"""GPU probe: how the sample distribution and the scatter's merge opportunities change as the bench model
trains (development tool).  usage: probe_trained.py [--scene] [steps ...]
--scene: supervise with an analytic street canyon (ground plane, two walls) instead of bench.py's per-slot random
targets, so that training converges to surfaces the way it does on real data."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd.parallel import GradAllReducer  # noqa: E402
from neuradar_amd.step import FlatAdam  # noqa: E402

wl = bench.WORKLOADS["cam4096_l16f2_w64"]
dev = torch.device("cuda")
n_rays = wl["rays"]
model = bench.build_model(wl, dev)
groups = model.get_param_groups()
unused = list(model.proposal_fields[0].parameters())
opts = [FlatAdam(groups["hashgrids"], lr=1e-2, eps=1e-15, lr_final=1e-3, max_steps=20001, warmup_steps=500, skip=unused),
        FlatAdam(groups["fields"], lr=1e-2, eps=1e-15, weight_decay=1e-7, adamw=True, lr_final=1e-3, max_steps=20001,
                 warmup_steps=500, skip=unused)]
reducer = GradAllReducer(None, buffers=[g for o in opts for g in o.grad_buffers()])
scene = bench.SyntheticScene(dev, seed=1000)
torch.manual_seed(1234)
targets = (0.1 * torch.randn(n_rays, 32, device=dev), 5.0 + 50.0 * torch.rand(n_rays, 1, device=dev))
fwd_bwd, optim, st = bench.make_step(model, scene, opts, reducer, targets, n_rays, fused=True, fuse_optimizer=True)


def uniq_rows(keys, group):
    n = keys.shape[0] // group * group
    s, _ = torch.sort(keys[:n].reshape(n // group, -1), dim=1)
    return ((s[:, 1:] != s[:, :-1]).sum(1) + 1).float().mean().item()


def report(tag):
    torch.cuda.synchronize()
    print(f"==== after {tag} steps: loss slots sum {float(st.loss.sum()):.4f}")
    for lvl, (name, grid) in enumerate([("prop_s128", st.pgrid), ("prop_s64", st.pgrid), ("main_s32", st.mgrid)]):
        S = st.S[lvl]
        eu = st.eu[lvl]  # [B,S+1] euclidean edges
        span = (eu[:, -1] - eu[:, 0])
        w = st.w[lvl]
        # effective number of samples carrying weight, and the depth interval that holds 90 % of the samples
        mid = 0.5 * (eu[:, 1:] + eu[:, :-1])
        q = torch.quantile(mid, torch.tensor([0.05, 0.5, 0.95], device=dev), dim=1)
        neff = (w.sum(1) ** 2 / (w * w).sum(1).clamp_min(1e-30))
        print(f"{name}: S={S} median sample depth {float(q[1].median()):8.2f} m; 5-95% interval per ray (median over rays) "
              f"{float((q[2] - q[0]).median()):9.2f} m of span {float(span.median()):9.1f}; weight n_eff {float(neff.mean()):.1f}")
        x = st.x01[lvl]  # rows in the step's order
        line = []
        for l in range(grid.num_levels):
            p = torch.floor(x * grid.scalings[l]).long()
            cell = (p[:, 0] * 2000003 + p[:, 1]) * 2000003 + p[:, 2]
            c64, c512 = uniq_rows(cell[:, None], 64), uniq_rows(cell[:, None], 512)
            # requests of a vertex-keyed table: distinct (x-pair of the cell, y corner, z corner) per chunk
            yz = torch.stack([(p[:, 0] * 2000003 + p[:, 1] + dy) * 2000003 + p[:, 2] + dz for dy in (0, 1) for dz in (0, 1)], 1)
            v512, v2048 = uniq_rows(yz, 512), uniq_rows(yz, 2048)
            c2048 = uniq_rows(cell[:, None], 2048)
            line.append(f"L{l}:{c64:.0f}/{c512:.0f}/{torch.unique(cell).numel()} req {4 * c512:.0f}->{v512:.0f} (2048: {4 * c2048:.0f}->{v2048:.0f})")
        print("   distinct cells per 64 rows / per 512 rows / global; requests per 512-row chunk cell-keyed -> vertex-keyed: " + " | ".join(line))


args = [a for a in sys.argv[1:] if a != "--scene"]
if "--scene" in sys.argv:
    S0 = model.config.num_proposal_samples[0]
    n_p = n_rays // (scene.PATCH * scene.PATCH)
    Wf = 0.3 * torch.randn(3, 32, device=dev)

    def fwd_bwd():
        bundle, _ = scene.cameras.generate_patch_rays(torch.rand(n_p, 3, device=dev), scene.PATCH, scene.STRIDE, scene.H, scene.W,
                                                      area_scale=9.0)
        o, d = bundle.origins, bundle.directions
        inf = torch.full_like(d[:, 0], 200.0)
        tg = torch.where(d[:, 2] < -1e-6, -o[:, 2] / d[:, 2].clamp(max=-1e-6), inf)
        tw = torch.where(d[:, 1].abs() > 1e-6, (12.0 * torch.sign(d[:, 1]) - o[:, 1]) / torch.where(d[:, 1].abs() > 1e-6, d[:, 1], inf), inf)
        depth = torch.minimum(torch.minimum(tg, tw), inf).contiguous()
        tf = (0.5 * torch.sin((o + depth[:, None] * d) @ Wf)).contiguous()
        return st.forward_backward(o, d, bundle.pixel_area[:, 0], None, tf, depth, torch.rand(n_rays, S0 + 1, device=dev),
                                   torch.rand(n_rays, device=dev), torch.rand(n_rays, device=dev), optimizers=opts)


def kernel_report():
    st.timers = {}
    for _ in range(20):
        fwd_bwd()
        optim()
    torch.cuda.synchronize()
    t = st.kernel_times()
    st.timers = None
    print("   in-step kernel us: " + " ".join(f"{k}={v * 1e6:.0f}" for k, v in t.items()))


done = 0
for target in [int(a) for a in args] or [1, 1000, 3000]:
    while done < target:
        fwd_bwd()
        optim()
        done += 1
    report(done)
    kernel_report()
    done += 20
