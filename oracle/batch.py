"""Oracle: the reference's batch assembly -- point / scan samplers and the camera+lidar+radar merge.
Test infrastructure only.

Restates data/pixel_samplers.py:538-577 (LidarPointSampler.collate_image_dataset_batch), :640-649 (RadarPointSampler's
scan choice) and data/datamanagers/image_lidar_radar_datamanager.py:335-409 (_merge_img_lidar_radar).  The reference
draws its random numbers inside these functions; here they are arguments (the drawn tensors), so that a device
implementation fed the same numbers can be compared index for index.
"""
import math
from typing import Dict, Optional, Sequence

import torch


def lidar_point_sample(u: torch.Tensor, shuffle: torch.Tensor, points_per_lidar: torch.Tensor, num_rays: int):
    """pixel_samplers.py:550-565.  u [num_lidars, rays_per_lidar] float64 in [0,1) (the `torch.rand` of :553), shuffle =
    the `torch.randperm(num_lidars)` of :550.  Returns indices [num_rays, 2] = (lidar, point) and the flat row of each
    point in the concatenated point table."""
    num_lidars = points_per_lidar.numel()
    rays_per_lidar = math.ceil(num_rays / num_lidars)
    assert u.shape == (num_lidars, rays_per_lidar)
    n_points = points_per_lidar.to(torch.int64)
    cum = torch.zeros(num_lidars, dtype=torch.int64)
    cum[1:] = torch.cumsum(n_points, 0)[:-1]
    point = torch.floor(u * n_points.view(num_lidars, 1)).long()
    lidar = torch.arange(num_lidars).unsqueeze(1).repeat(1, rays_per_lidar)
    lidar, point, cum_s = lidar[shuffle], point[shuffle], cum.view(num_lidars, 1)[shuffle]
    indices = torch.stack((lidar.flatten(), point.flatten()), dim=-1)[:num_rays]
    flat = (point + cum_s).flatten()[:num_rays]
    return indices, flat


def radar_scan_choice(u: Optional[torch.Tensor], n_scans: int, num_radars: int) -> torch.Tensor:
    """pixel_samplers.py:640-649; u [n_scans] stands for randint(0, num_radars - 1) = floor(u * (num_radars - 1))."""
    if num_radars <= n_scans:
        idx = torch.zeros(n_scans, dtype=torch.int64)
        idx[:num_radars] = torch.arange(num_radars)
        return idx
    return torch.floor(u.double() * (num_radars - 1)).long().clamp_(max=num_radars - 2)


def merge_img_lidar_radar(bundles: Dict[str, Dict[str, torch.Tensor]], order: Sequence[str] = ("camera", "lidar", "radar")
                          ) -> Dict[str, torch.Tensor]:
    """image_lidar_radar_datamanager.py:335-409: per-sensor ray bundles (dicts with origins, directions, pixel_area,
    times and whatever metadata the sensor's generator sets) -> one batch with is_lidar / is_radar / did_return /
    directions_spher filled in for every ray (:350-385).  `order` is the concatenation order (the reference's is camera,
    lidar, radar)."""
    out: Dict[str, list] = {}
    for name in order:
        b = bundles[name]
        n = b["origins"].shape[0]
        full = dict(b)
        full["is_lidar"] = torch.full((n, 1), name == "lidar", dtype=torch.bool)
        full["is_radar"] = torch.full((n, 1), name == "radar", dtype=torch.bool)
        full.setdefault("did_return", torch.ones(n, 1, dtype=torch.bool))
        full.setdefault("directions_spher", torch.zeros(n, 2))
        full.setdefault("directions_norm", torch.ones(n, 1))
        for k in ("origins", "directions", "pixel_area", "times", "is_lidar", "is_radar", "did_return", "directions_spher",
                  "directions_norm"):
            out.setdefault(k, []).append(full[k].reshape(n, -1))
    return {k: torch.cat(v, dim=0) for k, v in out.items()}
