"""Per-term losses of a decoder workload over the first training steps (eager steps, one host read per step)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from neuradar_amd.parallel import GradAllReducer  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "mixed16384_neuradar_full"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda")
wl = bench.WORKLOADS[workload]
n_rays = wl["rays"]
mlp_dtype = wl.get("mlp_dtype", "bfloat16")
model = bench.build_model(wl, dev, mlp_dtype, 8192.0 if mlp_dtype == "float16" else 1.0)
opts = bench.build_optimizers(model)
reducer = GradAllReducer(None, buffers=[g for o in opts for g in o.grad_buffers()])
scene = bench.SyntheticScene(dev, seed=1000, radar=wl.get("radar", "zod"))
torch.manual_seed(1234)
targets = (0.1 * torch.randn(n_rays, 32, device=dev), 5.0 + 50.0 * torch.rand(n_rays, 1, device=dev))
fwd_bwd, _, stepper = bench.make_step(model, scene, opts, reducer, targets, n_rays, fused=True, fuse_optimizer=True,
                                      mixed=wl if "cam_rays" in wl else None, scene_targets=os.environ.get("SCENE") == "1")
head = stepper.dec["head"] if stepper.dec is not None else None
for k in range(steps):
    fwd_bwd()
    if k % every == 0 or k == steps - 1:
        torch.cuda.synchronize()
        tot = float(stepper.loss.sum())
        terms = {n: float(v) for n, v in head.last["terms"].items()} if head is not None else {}
        out = stepper.outputs()
        d = out["depth"]
        print(f"step {k:4d} total {tot:9.4f} field-side {tot - sum(terms.values()):9.4f} " + " ".join(f"{n} {v:9.4f}" for n, v in terms.items())
              + f" | depth mean {float(d.mean()):8.2f} max {float(d.max()):9.1f} acc {float(out['accumulation'].mean()):.3f}"
              + (f" lidar stats {[round(float(x), 3) for x in head.last['lidar_stats']]}" if head is not None and 'lidar_stats' in head.last else ""),
              flush=True)
