"""Oracle: sensor ray generation (camera / lidar / radar).  Test infrastructure only.

Restates cameras/cameras.py:596-660,782-787,887-949 (perspective pinhole incl. rolling shutter),
cameras/camera_utils.py:596-610 (normalize_with_norm), cameras/lidars.py:356-417,506-519,
cameras/radars.py:268-358 and models/neuradar.py:996-1008 (_scale_pixel_area).

Each generator returns a dict of flat tensors mirroring `RayBundle` (cameras/rays.py:251-273):
origins [B,3], directions [B,3], pixel_area [B,1], times [B,1], fars [B,1] plus metadata entries.
"""
import torch

_EPS = 1e-7  # camera_utils._EPS
FAR = 1_000_000.0  # cameras.py:948, lidars.py:416, radars.py:356
LIDAR_H_DIVERGENCE = 3.0e-3  # lidars.py:41
LIDAR_V_DIVERGENCE = 1.5e-3  # lidars.py:42
LIDAR_VALID_DISTANCE = 1.0e3  # lidars.py valid_lidar_distance_threshold default


def normalize_with_norm(x):
    """camera_utils.py:596-610: x / max(|x|, eps), and the norm."""
    norm = torch.maximum(torch.linalg.vector_norm(x, dim=-1, keepdim=True),
                         torch.tensor([_EPS], dtype=x.dtype, device=x.device))
    return x / norm, norm


def undistort(coords, distortion_params, eps: float = 1e-3, max_iterations: int = 10):
    """radial_and_tangential_undistort: 10 Newton steps on the OpenCV model [k1,k2,k3,k4,p1,p2].
    camera_utils.py:655-758.  coords [...,2], distortion_params broadcastable [...,6]."""
    k1, k2, k3, k4, p1, p2 = (distortion_params[..., i] for i in range(6))
    xd, yd = coords[..., 0], coords[..., 1]
    x, y = xd, yd
    for _ in range(max_iterations):
        r = x * x + y * y
        d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
        fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
        fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
        d_r = k1 + r * (2.0 * k2 + r * (3.0 * k3 + r * 4.0 * k4))
        d_x, d_y = 2.0 * x * d_r, 2.0 * y * d_r
        fx_x = d + d_x * x + 2.0 * p1 * y + 6.0 * p2 * x
        fx_y = d_y * x + 2.0 * p1 * x + 2.0 * p2 * y
        fy_x = d_x * y + 2.0 * p2 * y + 2.0 * p1 * x
        fy_y = d + d_y * y + 2.0 * p2 * x + 6.0 * p1 * y
        den = fy_x * fx_y - fx_x * fy_y
        ok = torch.abs(den) > eps
        x = x + torch.where(ok, (fx * fy_y - fy * fx_y) / den, torch.zeros_like(den))
        y = y + torch.where(ok, (fy * fx_x - fx * fy_x) / den, torch.zeros_like(den))
    return torch.stack([x, y], dim=-1)


def camera_rays(ray_indices, c2w, fx, fy, cx, cy, cam_times, velocities=None, rs_offsets=None, heights=None,
                distortion=None, fisheye=None):
    """Pinhole / fisheye rays.  RayGenerator.forward (ray_generators.py:47-62) ->
    Cameras._generate_rays_from_coords (cameras.py:596-949).  distortion [C,6] (optional) applies the
    iterative undistortion (:636-653); fisheye [C] bool selects the FISHEYE branch (:789-804, ZOD's
    camera model) instead of PERSPECTIVE (:782-787).

    ray_indices [B,3] int64 (camera, row, col); c2w [C,3,4]; fx,fy,cx,cy [C]; cam_times [C];
    velocities [C,3] + rs_offsets [C,2] + heights [C] enable the top-to-bottom rolling shutter
    (cameras.py:922-939).  Pixel centres are at +0.5 (cameras.py:293-313).
    """
    c = ray_indices[:, 0]
    y = ray_indices[:, 1].to(c2w.dtype) + 0.5
    x = ray_indices[:, 2].to(c2w.dtype) + 0.5
    fx, fy, cx, cy = fx[c], fy[c], cx[c], cy[c]
    # image-plane coords of the pixel and of its +1 neighbours in x and in y (cameras.py:622-624)
    coord = torch.stack([(x - cx) / fx, (y - cy) / fy], -1)
    coord_dx = torch.stack([(x - cx + 1) / fx, (y - cy) / fy], -1)
    coord_dy = torch.stack([(x - cx) / fx, (y - cy + 1) / fy], -1)
    stack = torch.stack([coord, coord_dx, coord_dy], dim=0)  # [3,B,2]
    if distortion is not None:
        stack = undistort(stack, distortion[c][None])
    stack = stack * torch.tensor([1.0, -1.0], dtype=stack.dtype)  # OpenCV -> OpenGL (cameras.py:656)
    dirs = torch.cat([stack, -torch.ones_like(stack[..., :1])], dim=-1)  # perspective branch :782-787
    if fisheye is not None:
        theta = torch.clip(torch.sqrt(torch.sum(stack**2, dim=-1)), 0.0, torch.pi)
        fish = torch.stack([stack[..., 0] * torch.sin(theta) / theta, stack[..., 1] * torch.sin(theta) / theta,
                            -torch.cos(theta)], dim=-1)
        dirs = torch.where(fisheye[c][None, :, None], fish, dirs)
    rot = c2w[c][:, :3, :3]  # [B,3,3]
    dirs = torch.sum(dirs[..., None, :] * rot, dim=-1)  # cameras.py:892-894
    dirs, norms = normalize_with_norm(dirs)
    origins = c2w[c][:, :3, 3]
    d0 = dirs[0]
    dx = torch.sqrt(torch.sum((d0 - dirs[1]) ** 2, dim=-1))
    dy = torch.sqrt(torch.sum((d0 - dirs[2]) ** 2, dim=-1))
    pixel_area = (dx * dy)[:, None]
    times = cam_times[c][:, None]
    if velocities is not None and rs_offsets is not None:
        offs = rs_offsets[c]  # [B,2]
        duration = offs[:, 1:2] - offs[:, 0:1]  # offsets.diff()
        rows = (ray_indices[:, 1:2].to(c2w.dtype) + 0.5)  # coords[..., 0:1]
        time_offsets = rows / heights[c][:, None] * duration + offs[:, 0:1]
        origins = origins + velocities[c] * time_offsets
        times = times + time_offsets
    return {
        "origins": origins, "directions": d0, "pixel_area": pixel_area, "times": times,
        "fars": torch.full_like(pixel_area, FAR), "directions_norm": norms[0],
    }


def transform_points_pairwise(points, transforms, with_translation=True):
    """lidars.py:506-519: p @ R^T (+ t) with one transform per point."""
    rot = transforms[..., :3, :3]
    out = (points.unsqueeze(-2) @ rot.swapaxes(-2, -1)).squeeze(-2)
    return out + transforms[..., :3, 3] if with_translation else out


def lidar_rays(lidar_indices, points, l2w, scan_times, velocities=None, assume_ego_compensated=True):
    """Lidars._generate_rays_from_points.  lidars.py:356-417.

    lidar_indices [B] int64; points [B,>=5] (x,y,z,intensity,time offset) in the sensor frame;
    l2w [N,3,4]; scan_times [N]; velocities [N,3].
    """
    pose = l2w[lidar_indices]
    points_world = transform_points_pairwise(points[:, :3], pose)
    origins = pose[:, :3, 3]
    if points.shape[-1] >= 5 and velocities is not None:
        origins = origins + points[:, 4:5] * velocities[lidar_indices]
        if not assume_ego_compensated:
            points_world = points_world + points[:, 4:5] * velocities[lidar_indices]
    directions, distance = normalize_with_norm(points_world - origins)
    pixel_area = torch.full_like(distance, LIDAR_H_DIVERGENCE) * torch.full_like(distance, LIDAR_V_DIVERGENCE)
    times = scan_times[lidar_indices][:, None] + points[:, 4:5]
    return {
        "origins": origins, "directions": directions, "pixel_area": pixel_area, "times": times,
        "fars": torch.full_like(pixel_area, FAR), "directions_norm": distance,
        "is_lidar": torch.ones_like(distance, dtype=torch.bool),
        "did_return": distance < LIDAR_VALID_DISTANCE,
    }


def radar_fov_grid(min_az, max_az, d_az, min_el, max_el, d_el):
    """Azimuth-major FOV grid of one scan.  radars.py:279-295.  Returns [n_az*n_el, 2] (az, el)."""
    # the reference passes 0-dim float32 tensors (rows of [N,1] buffers), so limits and step are
    # float32-rounded BEFORE arange evaluates start + i*step in double
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731
    az = torch.arange(f32(min_az), f32(max_az), f32(d_az))
    el = torch.arange(f32(min_el), f32(max_el), f32(d_el))
    g_az, g_el = torch.meshgrid(az, el, indexing="ij")
    return torch.stack((g_az.flatten(), g_el.flatten()), dim=1)


def radar_rays(scan_indices, r2w, scan_times, min_az, max_az, d_az, min_el, max_el, d_el):
    """Radars._generate_rays_from_fov.  radars.py:268-358.  One FOV grid per requested scan.

    scan_indices [n] int64; r2w [N,3,4]; scan_times [N,1]; FOV limits/divergences are per-radar
    python floats here (the reference holds [N,1] tensors with identical rows per sensor).
    """
    spher, owner = [], []
    for index in scan_indices.tolist():
        grid = radar_fov_grid(min_az, max_az, d_az, min_el, max_el, d_el)
        spher.append(grid)
        owner.append(torch.full((grid.shape[0],), index, dtype=torch.int64))
    spher, owner = torch.cat(spher), torch.cat(owner)
    pose = r2w[owner]
    origins = pose[:, :3, 3]
    local = torch.zeros((spher.shape[0], 3))
    local[:, 0] = torch.cos(spher[:, 1]) * torch.cos(spher[:, 0])
    local[:, 1] = torch.cos(spher[:, 1]) * torch.sin(spher[:, 0])
    local[:, 2] = torch.sin(spher[:, 1])
    world = transform_points_pairwise(local, pose)  # WITH translation, then origins subtracted (:317-320)
    directions, distance = normalize_with_norm(world - origins)
    n = spher.shape[0]
    pixel_area = (torch.full((n, 1), d_az, dtype=torch.float32) / 5) * (torch.full((n, 1), d_el, dtype=torch.float32) / 5)
    return {
        "origins": origins, "directions": directions, "pixel_area": pixel_area,
        "times": scan_times[owner], "fars": torch.full_like(pixel_area, FAR),
        "directions_norm": distance, "directions_spher": spher,
        "did_return": torch.ones((spher.shape[0], 1), dtype=torch.bool),
        "scan_of_ray": owner,
    }


def scale_pixel_area(pixel_area, is_camera, rgb_upsample_factor: int = 3):
    """NeuRadarModel._scale_pixel_area: camera rays x upsample^2, lidar/radar x1.  neuradar.py:996-1008."""
    scaling = torch.ones_like(pixel_area)
    scaling[is_camera] = float(rgb_upsample_factor**2)
    return pixel_area * scaling
