"""`DynamicActors` (reference: model_components/dynamic_actors.py:30-222): learnable per-timestep actor
poses (6-D rotation + position), interpolated at each ray's time.  Host-side glue in torch (tiny
tensors, autograd carries the trajectory gradients); the per-actor hash grids behind it run on the HIP
kernels (see NeuRADHashEncoding)."""
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn


@dataclass
class DynamicActorsConfig:
    optimize_trajectories: bool = True
    actor_bbox_padding: Tuple[float, float, float] = (0.25, 0.25, 0.1)

    def setup(self, **kwargs) -> "DynamicActors":
        return DynamicActors(self, **kwargs)


def rotation_6d_to_matrix(d6: Tensor) -> Tensor:
    """cameras/camera_utils.py:422-443 (Gram-Schmidt; the vectors are the matrix rows)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def pose_inverse(pose: Tensor) -> Tensor:
    """utils/poses.py:35-49."""
    R, t = pose[..., :3, :3], pose[..., :3, 3:]
    Rt = R.transpose(-2, -1)
    return torch.cat([Rt, -Rt.matmul(t)], dim=-1)


class DynamicActors(nn.Module):
    def __init__(self, config: DynamicActorsConfig, trajectories: List[dict]):
        super().__init__()
        self.config = config
        times = sorted({float(t) for traj in trajectories for t in traj["timestamps"]})
        unique = torch.tensor(times, dtype=torch.float32)
        self.n_actors, self.n_times = len(trajectories), len(times)
        poses = torch.eye(4).view(1, 1, 4, 4).repeat(self.n_times, self.n_actors, 1, 1)
        present = torch.zeros((self.n_times, self.n_actors), dtype=torch.bool)
        sizes = torch.zeros((self.n_actors, 3))
        for a, traj in enumerate(trajectories):  # dynamic_actors.py:111-133
            sizes[a] = traj["dims"]
            for ti, t in enumerate(unique):
                diff = (traj["timestamps"] - t).abs()
                k = diff.argmin(dim=0)
                present[ti, a] = bool(diff[k] < 1e-4)
                poses[ti, a] = traj["poses"][k]  # absent timestamps duplicate the closest pose
        self.register_buffer("unique_timestamps", unique)
        self.register_buffer("actor_present_at_time", present)
        self.register_buffer("actor_sizes", sizes)
        self.register_buffer("actor_padding", torch.tensor(config.actor_bbox_padding))
        self.register_buffer("actor_to_id", torch.arange(self.n_actors, dtype=torch.int64))
        self.actor_positions = nn.Parameter(poses[..., :3, 3].clone(), requires_grad=config.optimize_trajectories)
        self.actor_rotations_6d = nn.Parameter(poses[..., :2, :3].clone().reshape(self.n_times, self.n_actors, 6),
                                               requires_grad=config.optimize_trajectories)

    @classmethod
    def from_state(cls, positions: Tensor, rotations_6d: Tensor, timestamps: Tensor, present: Tensor, sizes: Tensor,
                   config: DynamicActorsConfig = None) -> "DynamicActors":
        """Rebuild from checkpointed tensors (state_dict keys of the reference: actor_positions,
        actor_rotations_6d, unique_timestamps, actor_present_at_time, actor_sizes)."""
        config = config or DynamicActorsConfig()
        T, A = positions.shape[:2]
        eye = torch.eye(4).view(1, 4, 4).repeat(T, 1, 1)
        self = cls(config, [{"poses": eye, "timestamps": timestamps.clone(), "dims": sizes[a]} for a in range(A)])
        with torch.no_grad():
            self.actor_positions.copy_(positions)
            self.actor_rotations_6d.copy_(rotations_6d)
            self.actor_present_at_time.copy_(present)
        return self

    def actor_bounds(self) -> Tensor:
        return self.actor_sizes / 2 + self.actor_padding

    def get_param_groups(self, param_groups: Dict):
        if self.config.optimize_trajectories:
            param_groups["trajectory_opt"] = param_groups.get("trajectory_opt", []) + list(self.parameters())

    def get_boxes2world(self, query_times: Tensor):
        """query_times [B] -> (boxes2world [B,A,4,4], valid [B,A]); dynamic_actors.py:183-197 over
        interpolate_trajectories_6d(flatten=False) (utils/poses.py:90-149)."""
        poses = torch.cat([self.actor_rotations_6d, self.actor_positions], dim=-1)
        a1 = F.normalize(poses[..., :3], dim=-1)
        a2 = poses[..., 3:6]
        a2 = F.normalize(a2 - (a1 * a2).sum(-1, keepdim=True) * a1, dim=-1)
        poses = torch.cat([a1, a2, poses[..., 6:9]], dim=-1)
        ts = self.unique_timestamps
        right = torch.searchsorted(ts, query_times.contiguous())
        left = (right - 1).clamp(min=0)
        right = right.clamp(max=len(ts) - 1)
        frac = ((query_times - ts[left]) / (ts[right] - ts[left] + 1e-6)).clamp(0.0, 1.0)
        valid = self.actor_present_at_time[left] | self.actor_present_at_time[right]
        interp = poses[left] + (poses[right] - poses[left]) * frac[:, None, None]
        b2w = torch.cat([rotation_6d_to_matrix(interp[..., :6]), interp[..., 6:].unsqueeze(-1)], dim=-1)
        bottom = torch.zeros_like(b2w[..., :1, :])
        bottom[..., 3] = 1
        return torch.cat([b2w, bottom], dim=-2), valid
