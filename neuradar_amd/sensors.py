"""Device-side sensor ray generation (reference: cameras/cameras.py:505-949 perspective branch,
cameras/lidars.py:356-417, cameras/radars.py:268-358, model_components/ray_generators.py:32-117,
models/neuradar.py:996-1008).

The reference generates rays in CPU worker processes; here the sensor tables live on the GPU and a
batch of ray *indices* becomes a `RayBundle` in one kernel per sensor type.
"""
from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib, ops
from ._lib import check
from .rays import RayBundle

FAR = 1_000_000.0  # cameras.py:948


def _new(n, *shape, device, dtype=torch.float32):
    return torch.empty((n, *shape), device=device, dtype=dtype)


@dataclass
class Cameras:
    """Perspective / fisheye cameras with optional lens distortion and rolling shutter (cameras/cameras.py)."""

    camera_to_worlds: Tensor  # [C,3,4]
    fx: Tensor  # [C]
    fy: Tensor
    cx: Tensor
    cy: Tensor
    height: Tensor  # [C] float
    times: Tensor  # [C]
    velocities: Optional[Tensor] = None  # [C,3]  (metadata["velocities"], ad_dataparser.py:398-403)
    rolling_shutter_offsets: Optional[Tensor] = None  # [C,2]
    distortion_params: Optional[Tensor] = None  # [C,6] k1,k2,k3,k4,p1,p2 (cameras.py:100)
    camera_type: Optional[Tensor] = None  # [C] int32: 0 PERSPECTIVE, 1 FISHEYE (ZOD: zod_dataparser.py:261)
    _far_cache: Optional[Tensor] = None

    def generate_rays(self, ray_indices: Tensor) -> RayBundle:
        """RayGenerator.forward: ray_indices [n,3] int64 (camera,row,col) -> RayBundle."""
        n, dev = ray_indices.shape[0], ray_indices.device
        o, d = _new(n, 3, device=dev), _new(n, 3, device=dev)
        area, t, norm = _new(n, device=dev), _new(n, device=dev), _new(n, device=dev)
        rs = self.velocities is not None and self.rolling_shutter_offsets is not None
        p = ops._p
        check(_lib.lib().nr_gen_rays_camera(
            p(ray_indices.contiguous()), p(self.camera_to_worlds.contiguous()), p(self.fx), p(self.fy), p(self.cx),
            p(self.cy), p(self.times), p(self.velocities) if rs else None,
            p(self.rolling_shutter_offsets) if rs else None, p(self.height) if rs else None, p(self.distortion_params),
            p(self.camera_type), n, p(o), p(d), p(area), p(t), p(norm), ops._stream()), "nr_gen_rays_camera")
        return RayBundle(o, d, area[:, None], camera_indices=ray_indices[:, :1], fars=torch.full((n, 1), FAR, device=dev),
                         times=t[:, None], metadata={"directions_norm": norm[:, None]})


    def generate_patch_rays(self, u: Tensor, patch: int, stride: int, height: int, width: int,
                            area_scale: float = 1.0, return_indices: bool = False, out: Optional[RayBundle] = None):
        """On-device batch assembly: u [n_patches,3] uniform -> rays of random
        patch x patch blocks at pixel stride `stride`, one kernel.  Returns (RayBundle, ray_indices|None).
        out: a bundle returned by an earlier call of the same size, overwritten in place (callers that prefetch
        the next batch into fixed buffers)."""
        n_p, dev = u.shape[0], u.device
        n = n_p * patch * patch
        if out is not None:
            assert out.origins.shape[0] == n
            o, d, area, t = out.origins, out.directions, out.pixel_area.view(-1), out.times.view(-1)
        else:
            o, d = _new(n, 3, device=dev), _new(n, 3, device=dev)
            area, t = _new(n, device=dev), _new(n, device=dev)
        idx = _new(n, 3, device=dev, dtype=torch.int64) if return_indices else None
        rs = self.velocities is not None and self.rolling_shutter_offsets is not None
        p = ops._p
        check(_lib.lib().nr_gen_rays_camera_patches(
            p(u), n_p, self.fx.shape[0], height, width, patch,
            stride, area_scale, p(self.camera_to_worlds), p(self.fx), p(self.fy), p(self.cx), p(self.cy), p(self.times),
            p(self.velocities) if rs else None, p(self.rolling_shutter_offsets) if rs else None,
            p(self.height) if rs else None, p(self.distortion_params), p(self.camera_type), p(o), p(d), p(area), p(t), None,
            p(idx), ops._stream()),
            "nr_gen_rays_camera_patches")
        if self._far_cache is None or self._far_cache.shape[0] != n or self._far_cache.device != o.device:
            self._far_cache = torch.full((n, 1), FAR, device=dev)  # constant (cameras.py:948): one fill, not one per step
        if out is not None:
            return out, idx
        return RayBundle(o, d, area[:, None], fars=self._far_cache, times=t[:, None]), idx


@dataclass
class Lidars:
    lidar_to_worlds: Tensor  # [N,3,4]
    times: Tensor  # [N]
    velocities: Optional[Tensor] = None  # [N,3]

    def generate_rays(self, lidar_indices: Tensor, points: Tensor) -> RayBundle:
        """LidarRayGenerator.forward: lidar_indices [n] int64, points [n,>=5] (x,y,z,intensity,dt)."""
        n, dev = lidar_indices.shape[0], lidar_indices.device
        o, d = _new(n, 3, device=dev), _new(n, 3, device=dev)
        area, t, dist = _new(n, device=dev), _new(n, device=dev), _new(n, device=dev)
        ret = _new(n, device=dev, dtype=torch.uint8)
        points = points.contiguous()
        p = ops._p
        check(_lib.lib().nr_gen_rays_lidar(p(lidar_indices.contiguous()), p(points), points.shape[1],
                                           p(self.lidar_to_worlds.contiguous()), p(self.times), p(self.velocities), n,
                                           p(o), p(d), p(area), p(t), p(dist), p(ret), ops._stream()),
              "nr_gen_rays_lidar")
        meta = {"directions_norm": dist[:, None], "is_lidar": torch.ones((n, 1), dtype=torch.bool, device=dev),
                "did_return": ret.bool()[:, None]}
        return RayBundle(o, d, area[:, None], camera_indices=lidar_indices[:, None],
                         fars=torch.full((n, 1), FAR, device=dev), times=t[:, None], metadata=meta)


@dataclass
class Radars:
    """FOV-grid radars; defaults are the class constants of cameras/radars.py:38-44."""

    radar_to_worlds: Tensor  # [N,3,4]
    times: Tensor  # [N]
    radar_azimuth_ray_divergence: float = 0.0625
    radar_elevation_ray_divergence: float = 0.0625
    min_azimuth: float = -0.5
    max_azimuth: float = 0.5
    min_elevation: float = -0.5
    max_elevation: float = 0.5

    def grid_shape(self):
        """Lengths of torch.arange(min, max, step) with float32-rounded limits (radars.py:279-290)."""
        f = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731
        n_az = torch.arange(f(self.min_azimuth), f(self.max_azimuth), f(self.radar_azimuth_ray_divergence)).numel()
        n_el = torch.arange(f(self.min_elevation), f(self.max_elevation), f(self.radar_elevation_ray_divergence)).numel()
        return n_az, n_el

    def generate_rays(self, scan_indices: Tensor) -> RayBundle:
        n_az, n_el = self.grid_shape()
        dev = scan_indices.device
        n = scan_indices.shape[0] * n_az * n_el
        o, d = _new(n, 3, device=dev), _new(n, 3, device=dev)
        area, t, spher = _new(n, device=dev), _new(n, device=dev), _new(n, 2, device=dev)
        p = ops._p
        check(_lib.lib().nr_gen_rays_radar(p(scan_indices.contiguous()), scan_indices.shape[0],
                                           p(self.radar_to_worlds.contiguous()), p(self.times), self.min_azimuth,
                                           self.radar_azimuth_ray_divergence, n_az, self.min_elevation,
                                           self.radar_elevation_ray_divergence, n_el, p(o), p(d), p(area), p(t), p(spher),
                                           ops._stream()), "nr_gen_rays_radar")
        owner = scan_indices.repeat_interleave(n_az * n_el)
        meta = {"directions_spher": spher, "did_return": torch.ones((n, 1), dtype=torch.bool, device=dev),
                "directions_norm": torch.ones((n, 1), device=dev), "is_radar": torch.ones((n, 1), dtype=torch.bool, device=dev)}
        return RayBundle(o, d, area[:, None], camera_indices=owner[:, None], fars=torch.full((n, 1), FAR, device=dev),
                         times=t[:, None], metadata=meta)


def scale_pixel_area(bundle: RayBundle, rgb_upsample_factor: int = 3) -> None:
    """NeuRadarModel._scale_pixel_area (neuradar.py:996-1008): camera rays x upsample^2, in place."""
    is_lidar, is_radar = bundle.metadata.get("is_lidar"), bundle.metadata.get("is_radar")
    scaling = torch.ones_like(bundle.pixel_area)
    f = float(rgb_upsample_factor**2)
    if is_lidar is not None and is_radar is not None:
        scaling[~(is_lidar | is_radar)] = f
    elif is_lidar is not None:
        scaling[~is_lidar] = f
    elif is_radar is not None:
        scaling[~is_radar] = f
    else:
        scaling = f
    bundle.pixel_area = bundle.pixel_area * scaling


def merge_bundles(*bundles: RayBundle) -> RayBundle:
    """Concatenate camera / lidar / radar bundles (image_lidar_radar_datamanager.py:335-409 contract):
    boolean is_lidar / is_radar metadata are filled with False where a sensor does not set them."""
    n_tot = sum(len(b) for b in bundles)
    dev = bundles[0].origins.device
    cat = lambda name: torch.cat([getattr(b, name) for b in bundles], dim=0)  # noqa: E731
    meta: Dict[str, Tensor] = {}
    for key, fill, dtype, width in (("is_lidar", False, torch.bool, 1), ("is_radar", False, torch.bool, 1),
                                    ("did_return", True, torch.bool, 1), ("directions_norm", 1.0, torch.float32, 1)):
        parts = []
        for b in bundles:
            v = b.metadata.get(key)
            parts.append(v if v is not None else torch.full((len(b), width), fill, dtype=dtype, device=dev))
        meta[key] = torch.cat(parts, dim=0)
    assert meta["is_lidar"].shape[0] == n_tot
    cam_idx = torch.cat([b.camera_indices.reshape(len(b), -1)[:, :1] if b.camera_indices is not None
                         else torch.zeros((len(b), 1), dtype=torch.int64, device=dev) for b in bundles], dim=0)
    return RayBundle(cat("origins"), cat("directions"), cat("pixel_area"), camera_indices=cam_idx,
                     fars=cat("fars"), times=cat("times"), metadata=meta)
