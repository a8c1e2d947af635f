"""GPU probe: distinct cells / vertices / 64-B table lines per scatter chunk and globally (development tool)."""
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


def uniq_rows(keys, group):
    """mean number of distinct values per consecutive group of rows; keys [n, k] -> groups of `group` rows"""
    n = keys.shape[0] // group * group
    c = keys[:n].reshape(n // group, -1)
    s, _ = torch.sort(c, dim=1)
    return ((s[:, 1:] != s[:, :-1]).sum(1) + 1).float().mean().item()


P2, P3 = 2654435761, 805459861
for tag, fld, rs in [("prop_s128", model.proposal_fields[1], out["ray_samples_list"][0]),
                     ("prop_s64", model.proposal_fields[1], out["ray_samples_list"][1]),
                     ("main_s32", model.field, out["ray_samples"])]:
    g = fld.hashgrid.static_grid
    B, S = rs.shape
    F = g.features_per_level
    T = 2 ** g.log2_hashmap_size
    x01, _ = ops.contract_gaussians(rs.origins, rs.directions, rs.pixel_area, rs.euclid, fld.hashgrid.static_scale)
    x = x01.view(B, S, 3).permute(1, 0, 2).reshape(-1, 3)
    tot = dict(cells=0, verts=0, lines=0, gverts=0, glines=0, cells4=0, verts4=0, lines4=0)
    for l in range(g.num_levels):
        p = torch.floor(x * g.scalings[l]).long()
        cell = (p[:, 0] * 2000003 + p[:, 1]) * 2000003 + p[:, 2]
        offs = torch.tensor([[i & 1, (i >> 1) & 1, (i >> 2) & 1] for i in range(8)], device=dev)
        v = p[:, None, :] + offs[None]  # [n,8,3]
        slot = ((v[..., 0] * 1) ^ (v[..., 1] * P2) ^ (v[..., 2] * P3)) & (T - 1)  # [n,8]
        line = (slot * F * 4) >> 6
        r = dict(cells=uniq_rows(cell[:, None], 1024), verts=uniq_rows(slot, 1024), lines=uniq_rows(line, 1024),
                 cells4=uniq_rows(cell[:, None], 4096), verts4=uniq_rows(slot, 4096), lines4=uniq_rows(line, 4096),
                 gverts=torch.unique(slot).numel(), glines=torch.unique(line).numel())
        nch = x.shape[0] / 1024
        print(f"{tag} L{l:2d} s={float(g.scalings[l]):6.0f}: per 1024-chunk cells {r['cells']:6.1f} verts {r['verts']:7.1f} lines {r['lines']:7.1f}"
              f" | per 4096: cells {r['cells4']:7.1f} verts {r['verts4']:7.1f} lines {r['lines4']:7.1f} | global verts {r['gverts']:7d} lines {r['glines']:7d}")
        for k in ("cells", "verts", "lines"):
            tot[k] += r[k] * nch
            tot[k + "4"] += r[k + "4"] * nch / 4
        tot["gverts"] += r["gverts"]
        tot["glines"] += r["glines"]
    print(f"{tag} TOTAL per-1024-chunk sums: cells {tot['cells']:.0f} (x8 = {tot['cells'] * 8:.0f}) verts {tot['verts']:.0f} lines {tot['lines']:.0f};"
          f" per-4096: cells {tot['cells4']:.0f} verts {tot['verts4']:.0f} lines {tot['lines4']:.0f}; global verts {tot['gverts']} lines {tot['glines']}")
