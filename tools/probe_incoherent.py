"""GPU probe: scatter cost for incoherent (lidar-like) rays, rows stored sample-major vs ray-major (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes  # noqa: E402

import bench  # noqa: E402
from neuradar_amd import ops  # noqa: E402

lab = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scatter_lab.so"))
P, I, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lab.lab_scatter.argtypes = [I, P, P, P, I, I, I, P, L64, L64, P, L64, I, I, P]
lab.lab_scatter.restype = I

dev = torch.device("cuda")
wl = bench.WORKLOADS["mixed16384_neuradar"]
model = bench.build_model(wl, dev)
scene = bench.SyntheticScene(dev, 1000)
lib, p, st = ops._lib.lib(), ops._p, ops._stream
torch.manual_seed(0)
for tag, n_rays in (("lidar", 4661),):
    pick = torch.randint(0, scene.lidar_points.shape[0], (n_rays,), device=dev)
    b = scene.lidars.generate_rays(scene.lidar_owner[pick], scene.lidar_points[pick])
    model2 = bench.build_model(bench.WORKLOADS["cam4096_l16f2_w64"], dev)
    for S, grid_owner, name in ((128, model.proposal_fields[1], "prop_s128"), (32, model.field, "main_s32 F4"), (32, model2.field, "main_s32 F2")):
        g = grid_owner.hashgrid.static_grid
        L, F = g.num_levels, g.features_per_level
        sp, eu = ops.power_bins(torch.zeros(n_rays, device=dev), torch.full((n_rays,), 20000.0, device=dev), S)
        n = n_rays * S
        for sm in (True, False):
            x01, std = ops.contract_gaussians(b.origins, b.directions, b.pixel_area[:, 0], eu, 100.0, sample_major_rows=sm)
            gb = torch.randn(L, n, F, device=dev)
            gt = torch.zeros_like(g.hash_table)
            fn = lambda: lib.nr_hash_encode_bwd(p(x01), p(std), p(g.scalings), L, F, g.log2_hashmap_size, p(gb), F, n * F, p(gt), n, 0, st())  # noqa: E731
            t = bench.time_kernel(fn, 10)
            ob = torch.empty(L, n, F, device=dev)
            fnf = lambda: lib.nr_hash_encode_fwd(p(x01), p(std), p(g.hash_table), p(g.scalings), L, F, g.log2_hashmap_size, p(ob), F, n * F, n, 0, st())  # noqa: E731
            tf = bench.time_kernel(fnf, 10)
            extra = ""
            for variant in (1, 2, 3):
                fnv = lambda: lab.lab_scatter(variant, p(x01), p(std), p(g.scalings), L, F, g.log2_hashmap_size, p(gb), F, n * F, p(gt), n, 0, 0, st())  # noqa: E731
                if fnv() == 0:
                    extra += f" lab[{variant}] {bench.time_kernel(fnv, 10) * 1e6:7.1f}"
            print(extra)
            print(f"{tag} {n_rays} rays {name}: rows {'sample' if sm else 'ray   '}-major: scatter {t * 1e6:8.1f} us   gather {tf * 1e6:7.1f} us "
                  f"(atomic floor at 20 G req/s: {n * L * 8 * max(1, F * 4 // 16) / 20e9 * 1e6:7.1f} us)")
