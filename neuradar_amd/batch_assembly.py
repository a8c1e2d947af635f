"""Device-side batch assembly.

The reference builds a training batch on CPU workers: `ScaledPatchSampler` / `LidarPointSampler` / `RadarPointSampler`
(data/pixel_samplers.py:789-841, 538-577, 629-682) pick pixels, points and scans, the ray generators turn them into
three `RayBundle`s, and `_merge_img_lidar_radar` (data/datamanagers/image_lidar_radar_datamanager.py:335-409)
concatenates them and fills the `is_lidar` / `is_radar` / `did_return` / `directions_spher` metadata and the
dataset-offset camera indices.  Here the sensor tables live in HBM and one step's batch is FOUR launches that write
straight into the merged buffers (no per-sensor bundles, no concatenation): camera patches, lidar points (sampling +
ray generation fused), radar scan choice, radar FOV grids.  The per-ray constants of the merge are built once.
"""
import math
from typing import Dict, Optional, Sequence

import torch
from torch import Tensor

from . import _lib, ops
from ._lib import check
from .rays import RayBundle
from .sensors import FAR, Cameras, Lidars, Radars


class SensorBatchAssembler:
    def __init__(self, cameras: Cameras, height: int, width: int, patch: int, stride: int, n_patches: int,
                 lidars: Optional[Lidars] = None, lidar_points: Optional[Tensor] = None,
                 points_per_lidar: Optional[Tensor] = None, n_lidar_rays: int = 0,
                 radars: Optional[Radars] = None, n_radar_scans: int = 0,
                 order: Sequence[str] = ("camera", "lidar", "radar"), rgb_upsample_factor: int = 3, n_slots: int = 2) -> None:
        """order: segment order inside the merged batch.  The reference's is (camera, lidar, radar); rays are independent,
        so a caller may put the spatially coherent segments first (camera, radar, lidar: FusedTrainStep's coherent_rays).
        n_slots: buffer sets (a pipelined step fills one while the other is in use)."""
        self.cameras, self.lidars, self.radars = cameras, lidars, radars
        self.H, self.W, self.patch, self.stride, self.n_patches = height, width, patch, stride, n_patches
        self.area_scale = float(rgb_upsample_factor**2)  # _scale_pixel_area, models/neuradar.py:996-1008
        dev = cameras.fx.device
        self.dev = dev
        self.n_cam = n_patches * patch * patch
        self.n_lidar = n_lidar_rays if lidars is not None else 0
        self.n_scans = n_radar_scans if radars is not None else 0
        self.per_scan = 0
        if self.n_scans:
            n_az, n_el = radars.grid_shape()
            self.per_scan, self.n_az, self.n_el = n_az * n_el, n_az, n_el
        self.n_radar = self.n_scans * self.per_scan
        self.n = self.n_cam + self.n_lidar + self.n_radar
        sizes = {"camera": self.n_cam, "lidar": self.n_lidar, "radar": self.n_radar}
        assert sorted(order) == ["camera", "lidar", "radar"]
        self.offset, off = {}, 0
        for name in order:
            self.offset[name] = off
            off += sizes[name]
        self.order = tuple(order)
        if self.n_lidar:
            self.lidar_points = lidar_points.contiguous()
            self.points_per_lidar = points_per_lidar.to(dev).to(torch.int64).contiguous()
            self.num_lidars = int(self.points_per_lidar.numel())
            cum = torch.zeros(self.num_lidars, device=dev, dtype=torch.int64)
            cum[1:] = torch.cumsum(self.points_per_lidar, 0)[:-1]
            self.cum_points = cum
            self.rays_per_lidar = math.ceil(self.n_lidar / self.num_lidars)  # pixel_samplers.py:552
            self.lidar_order = torch.arange(self.num_lidars, device=dev, dtype=torch.int64)
        else:
            self.num_lidars = 0
        self.num_radars = int(radars.times.numel()) if self.n_scans else 0
        f32 = dict(device=dev, dtype=torch.float32)
        n = self.n
        self.slots = []
        for _ in range(n_slots):
            s = dict(origins=torch.empty(n, 3, **f32), directions=torch.empty(n, 3, **f32), pixel_area=torch.empty(n, **f32),
                     times=torch.empty(n, **f32), directions_norm=torch.ones(n, **f32),
                     did_return=torch.ones(n, device=dev, dtype=torch.uint8), directions_spher=torch.zeros(n, 2, **f32),
                     lidar_indices=torch.zeros(max(self.n_lidar, 1), 2, device=dev, dtype=torch.int64),
                     scan_indices=torch.zeros(max(self.n_scans, 1), device=dev, dtype=torch.int64))
            self.slots.append(s)
        # constants of _merge_img_lidar_radar (:350-385)
        seg = lambda name: slice(self.offset[name], self.offset[name] + sizes[name])  # noqa: E731
        self.seg = seg
        self.is_lidar = torch.zeros(n, 1, device=dev, dtype=torch.bool)
        self.is_lidar[seg("lidar")] = True
        self.is_radar = torch.zeros(n, 1, device=dev, dtype=torch.bool)
        self.is_radar[seg("radar")] = True
        self.fars = torch.full((n, 1), FAR, **f32)  # cameras.py:948, lidars.py / radars.py likewise

    def uniform_count(self) -> int:
        """Uniform numbers one batch consumes: 3 per patch, 1 per lidar ray, 1 per lidar (shuffle), 1 per radar scan."""
        return 3 * self.n_patches + self.n_lidar + self.num_lidars + self.n_scans

    def assemble(self, u: Tensor, slot: int = 0) -> Dict[str, Tensor]:
        """u [uniform_count()] in [0,1) -> the merged batch in buffer set `slot` (views of it are returned)."""
        lib, p, st = _lib.lib(), ops._p, ops._stream()
        s = self.slots[slot]
        c = self.cameras
        rs = c.velocities is not None and c.rolling_shutter_offsets is not None
        o0 = self.offset["camera"]
        n_u = 3 * self.n_patches
        check(lib.nr_gen_rays_camera_patches(
            p(u[:n_u]), self.n_patches, c.fx.shape[0], self.H, self.W, self.patch, self.stride, self.area_scale,
            p(c.camera_to_worlds), p(c.fx), p(c.fy), p(c.cx), p(c.cy), p(c.times), p(c.velocities) if rs else None,
            p(c.rolling_shutter_offsets) if rs else None, p(c.height) if rs else None, p(c.distortion_params), p(c.camera_type),
            p(s["origins"][o0:]), p(s["directions"][o0:]), p(s["pixel_area"][o0:]), p(s["times"][o0:]), None, None, st),
            "nr_gen_rays_camera_patches")
        if self.n_lidar:
            l, o1 = self.lidars, self.offset["lidar"]
            u_perm = u[n_u + self.n_lidar:n_u + self.n_lidar + self.num_lidars]
            check(lib.nr_permutation_from_uniform(p(u_perm), self.num_lidars, p(self.lidar_order), st), "nr_permutation_from_uniform")
            check(lib.nr_gen_rays_lidar_sampled(
                p(u[n_u:]), self.n_lidar, self.rays_per_lidar, p(self.lidar_order), p(self.points_per_lidar), p(self.cum_points),
                p(self.lidar_points), self.lidar_points.shape[1], p(l.lidar_to_worlds), p(l.times), p(l.velocities),
                p(s["origins"][o1:]), p(s["directions"][o1:]), p(s["pixel_area"][o1:]), p(s["times"][o1:]),
                p(s["directions_norm"][o1:]), p(s["did_return"][o1:]), p(s["lidar_indices"]), st), "nr_gen_rays_lidar_sampled")
        if self.n_scans:
            r, o2 = self.radars, self.offset["radar"]
            u_scan = u[n_u + self.n_lidar + self.num_lidars:]
            check(lib.nr_sample_radar_scans(p(u_scan), self.n_scans, self.num_radars, p(s["scan_indices"]), st), "nr_sample_radar_scans")
            check(lib.nr_gen_rays_radar(p(s["scan_indices"]), self.n_scans, p(r.radar_to_worlds), p(r.times), r.min_azimuth,
                                        r.radar_azimuth_ray_divergence, self.n_az, r.min_elevation,
                                        r.radar_elevation_ray_divergence, self.n_el, p(s["origins"][o2:]),
                                        p(s["directions"][o2:]), p(s["pixel_area"][o2:]), p(s["times"][o2:]),
                                        p(s["directions_spher"][o2:]), st), "nr_gen_rays_radar")
        return s

    def bundle(self, slot: int = 0) -> RayBundle:
        """The merged batch as a RayBundle (pixel_area already scaled for camera rays, neuradar.py:996-1008)."""
        s = self.slots[slot]
        meta = {"is_lidar": self.is_lidar, "is_radar": self.is_radar, "did_return": s["did_return"].bool()[:, None],
                "directions_norm": s["directions_norm"][:, None], "directions_spher": s["directions_spher"]}
        return RayBundle(s["origins"], s["directions"], s["pixel_area"][:, None], fars=self.fars, times=s["times"][:, None],
                         metadata=meta)
