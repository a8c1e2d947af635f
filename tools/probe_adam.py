"""GPU probe: the main table's Adam ALONE on the mixed step's own gradients / seen bytes -- two moment arrays against the one
interleaved array of [exp_avg x 4 | exp_avg_sq x 4] records, marked and unmarked launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd import ops  # noqa: E402
from neuradar_amd.parallel import GradAllReducer  # noqa: E402

wl = bench.WORKLOADS["mixed16384_neuradar"]
dev = torch.device("cuda")
model = bench.build_model(wl, dev, "bfloat16")
opts = bench.build_optimizers(model)
red = GradAllReducer(None, buffers=[g for o in opts for g in o.grad_buffers()])
scene = bench.SyntheticScene(dev, seed=1000)
n_rays = wl["rays"]
targets = (0.1 * torch.randn(n_rays, 32, device=dev), 5.0 + 50.0 * torch.rand(n_rays, 1, device=dev))
fwd_bwd, optim, st = bench.make_step(model, scene, opts, red, targets, n_rays, fused=True, fuse_optimizer=True, mixed=wl)
done = 0
for upto in [int(v) for v in os.environ.get("PROBE_STEPS", "24,600").split(",")]:
    for _ in range(upto - done):
        fwd_bwd()
    done = upto
    torch.cuda.synchronize()
    t = opts[0]
    i = t.buffer_of(model.field.hashgrid.static_grid.hash_table)
    p, g = t.buffers[i]
    m, v = t.state[i]
    seen = t.seen[i]
    n = p.numel()
    # a gradient like the step's: run the scatter part once more without the optimizer -> grads in g
    fwd_bwd_grad = g.clone()
    live = float((seen != 0).float().mean())
    print(f"--- after {upto} steps: {100 * live:.1f} % of the 16-byte groups live")
    for tag, inter in (("two arrays", False), ("interleaved", True)):
        if inter:
            mv = torch.zeros(n // 4, 2, 4, device=dev)
            mm, vv = mv[:, 0, :], mv[:, 1, :]
        else:
            mm, vv = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        mm.reshape(-1).copy_(m.reshape(-1)) if not inter else mm.copy_(m.reshape(-1, 4))
        vv.reshape(-1).copy_(v.reshape(-1)) if not inter else vv.copy_(v.reshape(-1, 4))
        pp = p.clone().view(-1)
        for marked in (True, False):
            gg = torch.zeros(n, device=dev)

            def launch():
                ops.adam_step(pp, gg, mm, vv, 1e-3, 5, eps=1e-15, seen_grad=seen, marked=marked, zero_grad=False)

            print(f"{tag:12s} marked={marked}: {bench.time_kernel(launch, 10) * 1e6:7.1f} us")
