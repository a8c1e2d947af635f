"""GPU probe: the merging scatter (nr_hash_encode_bwd) against the two-pass binned scatter (nr_hash_encode_bwd_binned)
on incoherent rows (uniformly random positions ~ lidar rays) and on the rows of a camera-patch batch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd import ops  # noqa: E402
from neuradar_amd.encodings import HashEncoding  # noqa: E402

dev = torch.device("cuda")
lib, p, st = ops._lib.lib(), ops._p, ops._stream
for tag, L, F, log2t, rmin, rmax, rays, S in (("prop_s128", 6, 1, 20, 128, 4096, 4661, 128), ("prop_s64", 6, 1, 20, 128, 4096, 4661, 64),
                                              ("l16f2_s32", 16, 2, 19, 16, 1024, 4661, 32), ("prop_s128_16k", 6, 1, 20, 128, 4096, 16384, 128)):
    n = rays * S
    sc = HashEncoding(num_levels=L, min_res=rmin, max_res=rmax, log2_hashmap_size=12, features_per_level=F).scalings.to(dev)
    torch.manual_seed(0)
    # lidar-like: rays from one origin in random directions, power-spaced samples, ray-major rows
    o = torch.tensor([0.5, 0.5, 0.5], device=dev)
    d = torch.nn.functional.normalize(torch.randn(rays, 3, device=dev), dim=-1)
    t = (torch.linspace(0, 1, S, device=dev) ** 3 * 0.45)[None, :, None]
    x = (o + d[:, None, :] * t).reshape(n, 3).clamp(0, 1).contiguous()
    std = 0.0005 * torch.rand(n, device=dev)
    g = torch.randn(L, n, F, device=dev)
    gt = torch.zeros(L << log2t, F, device=dev)
    ws = torch.empty(lib.nr_hash_encode_bwd_binned_workspace_bytes(L, F, log2t, n), device=dev, dtype=torch.uint8)
    a = bench.time_kernel(lambda: lib.nr_hash_encode_bwd(p(x), p(std), p(sc), L, F, log2t, p(g), F, n * F, p(gt), n, 0, st()), 10)
    b = bench.time_kernel(lambda: lib.nr_hash_encode_bwd_binned(p(x), p(std), p(sc), L, F, log2t, p(g), F, n * F, p(gt), n, p(ws), st()), 10)
    print(f"{tag:14s} n={n:8d}: merging {a * 1e6:8.1f} us   binned {b * 1e6:8.1f} us   workspace {ws.numel() / 1e6:7.1f} MB")
