"""GPU probe: per-level time of nr_hash_encode_bwd on the bench workload (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd import ops  # noqa: E402
from neuradar_amd.sensors import scale_pixel_area  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cam4096_l16f2_w64"]
dev = torch.device("cuda")
model = bench.build_model(wl, dev)
scene = bench.SyntheticScene(dev, 1000)
torch.manual_seed(1)
with torch.no_grad():
    bundle = scene.cameras.generate_rays(scene.sample_ray_indices(wl["rays"]))
    scale_pixel_area(bundle)
    out = model.get_nff_outputs(bundle)
lib, p, st = ops._lib.lib(), ops._p, ops._stream
for tag, fld, rs in [("prop_s128", model.proposal_fields[1], out["ray_samples_list"][0]),
                     ("main_s32", model.field, out["ray_samples"])]:
    g = fld.hashgrid.static_grid
    B, S = rs.shape
    n, L, F = B * S, g.num_levels, g.features_per_level
    x01, std01 = ops.contract_gaussians(rs.origins, rs.directions, rs.pixel_area, rs.euclid, fld.hashgrid.static_scale)
    T = 2 ** g.log2_hashmap_size
    gtab = torch.zeros_like(g.hash_table)
    tot = 0.0
    for l in range(L):
        gb = torch.randn((1, n, F), device=dev)
        sc = g.scalings[l:l + 1].contiguous()
        gt = gtab[l * T:(l + 1) * T]
        fn = lambda: lib.nr_hash_encode_bwd(p(x01), p(std01), p(sc), 1, F, g.log2_hashmap_size, p(gb), F, n * F, p(gt), n, S, st())  # noqa: E731
        t = bench.time_kernel(fn, 10)
        tot += t
        print(f"{tag} level {l:2d} scale {float(sc[0]):7.0f}: {t * 1e6:8.1f} us")
    print(f"{tag} sum of single-level launches {tot * 1e6:.1f} us")
