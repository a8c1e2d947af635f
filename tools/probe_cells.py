"""GPU probe: how many distinct grid cells does the bench batch touch per level, per wave, per block?
Guides the on-chip pre-reduction design of the scatter kernel.  Development tool."""
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


def uniq_per_group(cell, group):
    n = cell.shape[0] // group * group
    c = cell[:n].view(-1, group)
    s, _ = torch.sort(c, dim=1)
    return ((s[:, 1:] != s[:, :-1]).sum(1) + 1).float()


for tag, fld, rs in [("prop_s128", model.proposal_fields[1], out["ray_samples_list"][0]),
                     ("main_s32", model.field, out["ray_samples"])]:
    g = fld.hashgrid.static_grid
    B, S = rs.shape
    x01, _ = ops.contract_gaussians(rs.origins, rs.directions, rs.pixel_area, rs.euclid, fld.hashgrid.static_scale)
    x = x01.view(B, S, 3).permute(1, 0, 2).reshape(-1, 3)  # sample-major order, as the kernel walks it
    for l in range(g.num_levels):
        p = torch.floor(x * g.scalings[l]).long()
        cell = (p[:, 0] * 2000003 + p[:, 1]) * 2000003 + p[:, 2]
        total = torch.unique(cell).numel()
        runs = ((cell[1:] != cell[:-1]).view(-1)).sum().item() + 1
        print(f"{tag} L{l:2d} scale {float(g.scalings[l]):6.0f}: samples {cell.numel():7d} distinct cells {total:7d} "
              f"runs {runs:7d} | distinct/wave64 {uniq_per_group(cell, 64).mean():5.1f} /block256 {uniq_per_group(cell, 256).mean():6.1f} "
              f"/1024 {uniq_per_group(cell, 1024).mean():6.1f} /4096 {uniq_per_group(cell, 4096).mean():7.1f}")
