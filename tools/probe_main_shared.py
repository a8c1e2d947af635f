"""GPU probe: the main grid's scatter ALONE on the mixed step's own positions / gradients -- the merging kernel (wave-private
cell tables, wide configuration) against the block-shared vertex-keyed LDS table (grid_shared.hip): time over all levels and
per level, and the two results against each other (relative L2, largest entry-wise difference, write sets)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd import ops  # noqa: E402
from neuradar_amd.parallel import GradAllReducer  # noqa: E402

wl = bench.WORKLOADS[os.environ.get("PROBE_WORKLOAD", "mixed16384_neuradar")]
dev = torch.device("cuda")
model = bench.build_model(wl, dev, "bfloat16")
opts = bench.build_optimizers(model)
red = GradAllReducer(None, buffers=[g for o in opts for g in o.grad_buffers()])
scene = bench.SyntheticScene(dev, seed=1000)
n_rays = wl["rays"]
targets = (0.1 * torch.randn(n_rays, 32, device=dev), 5.0 + 50.0 * torch.rand(n_rays, 1, device=dev))
fwd_bwd, optim, st = bench.make_step(model, scene, opts, red, targets, n_rays, fused=True, fuse_optimizer=True,
                                     mixed=wl if "cam_rays" in wl else None, scene_targets=os.environ.get("PROBE_SCENE", "0") == "1")
lib, p, s = st.lib, ops._p, ops._stream
grid = st.mgrid
S, F, L, T = st.S[2], grid.features_per_level, grid.num_levels, grid.log2_hashmap_size
nl = n_rays * S
iters = int(os.environ.get("PROBE_ITERS", "10"))
done = 0
for upto in [int(v) for v in os.environ.get("PROBE_STEPS", "6,600").split(",")]:
    for _ in range(upto - done):
        fwd_bwd()
    done = upto
    torch.cuda.synchronize()
    x, sd, g = st.x01[2].clone(), st.std[2].clone(), st.g_feats[2].clone()
    gt_a, gt_b = torch.zeros_like(grid.hash_table), torch.zeros_like(grid.hash_table)

    def merging(l0, nlev, out):
        return lambda: lib.nr_hash_encode_bwd_tuned(p(x), p(sd), grid.scalings.data_ptr() + 4 * l0, nlev, F, T, g.data_ptr() + 4 * l0 * nl * F,
                                                    F, nl * F, out.data_ptr() + 4 * l0 * (1 << T) * F, nl, 0, 256, s())

    def shared(l0, nlev, out):
        return lambda: lib.nr_hash_encode_bwd_shared(p(x), p(sd), grid.scalings.data_ptr() + 4 * l0, nlev, F, T, g.data_ptr() + 4 * l0 * nl * F,
                                                     F, nl * F, out.data_ptr() + 4 * l0 * (1 << T) * F, nl, None, s())

    assert merging(0, L, gt_a)() == 0 and shared(0, L, gt_b)() == 0
    torch.cuda.synchronize()
    a, b = gt_a.double(), gt_b.double()
    rel = float((a - b).norm() / a.norm())
    print(f"--- after {upto} steps: |g| max {float(g.abs().max()):.3e}; shared vs merging: rel L2 {rel:.3e}, max |d| {float((a - b).abs().max()):.3e} "
          f"(max |a| {float(a.abs().max()):.3e}); written by shared only: {int(((b != 0) & (a == 0)).sum())}, by merging only: "
          f"{int(((a != 0) & (b == 0)).sum())} of {int((a != 0).sum())} entries")
    scr_a, scr_b = torch.zeros_like(grid.hash_table), torch.zeros_like(grid.hash_table)
    print(f"all {L} levels: merging {bench.time_kernel(merging(0, L, scr_a), iters) * 1e6:7.1f} us   shared {bench.time_kernel(shared(0, L, scr_b), iters) * 1e6:7.1f} us")
    if hasattr(lib, "nr_debug_shared_clocks"):  # a -DNR_SHARED_CLOCKS build (NR_LIB_PATH): wave-cycles per phase of one launch
        import ctypes

        buf = (ctypes.c_ulonglong * 8)()
        lib.nr_debug_shared_clocks(buf, 1)
        shared(0, L, scr_b)()
        torch.cuda.synchronize()
        lib.nr_debug_shared_clocks(buf, 1)
        c = [int(v) for v in buf]
        tot = max(sum(c), 1)
        names = ["tile load + setup", "corners, hash, wave max", "barrier A", "append to the occupied list", "barrier B", "flush",
                 "block max, fixed point, segmented scan", "cmpst round trip, probing, ds_add"]
        print("phase share of wave-cycles: " + "; ".join(f"{n} {100.0 * v / tot:.1f} %" for n, v in zip(names, c)) + f"  [total {tot / 1e6:.1f} M wave-cycles]")
    if os.environ.get("PROBE_LEVELS", "1") == "1":
        for l0 in range(L):
            print(f"level {l0} (scale {float(grid.scalings[l0]):6.0f}): merging {bench.time_kernel(merging(l0, 1, scr_a), iters) * 1e6:7.1f} us"
                  f"   shared {bench.time_kernel(shared(l0, 1, scr_b), iters) * 1e6:7.1f} us")
