"""DecoderLossHead segments alone at the bench shape: which part costs what (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd.decoder_losses import DecoderLossHead, DecoderLossSettings  # noqa: E402

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "mixed16384_neuradar_full"]
model = bench.build_model(wl, dev, "bfloat16")
n_rays, n_cam, n_lid = wl["rays"], wl["cam_rays"], wl["lidar_rays"]
n_rad = n_rays - n_cam - n_lid
P = n_cam // 1024
gen = torch.Generator(device=dev).manual_seed(0)
f32 = dict(device=dev, dtype=torch.float32)
feats = torch.randn(n_rays, 32, generator=gen, **f32) * 0.3
depth = torch.rand(n_rays, generator=gen, **f32) * (1.0 if os.environ.get("CLUSTER", "1") == "1" else 80.0) + 0.5
times = torch.rand(n_rays, generator=gen, **f32) * 20
sensor = torch.zeros(n_rays, dtype=torch.int64, device=dev)
batch = dict(image=torch.rand(P, 96, 96, 3, generator=gen, **f32), did_return=(torch.rand(n_rays, generator=gen, **f32) < 0.9).to(torch.uint8),
             range=torch.rand(n_rays, generator=gen, **f32) * 100 + 2, target_intensity=torch.rand(n_rays, generator=gen, **f32),
             directions_spher=torch.stack([torch.rand(n_rays, generator=gen, **f32) * 1.6 - 0.8, torch.rand(n_rays, generator=gen, **f32) * 0.48 - 0.08], -1),
             radar=torch.cat([torch.randn(200, 3, generator=gen, **f32) * 20 + torch.tensor([40.0, 0, 0], device=dev), torch.rand(200, 2, generator=gen, **f32)], 1),
             radar_seg=torch.tensor([0, 200], dtype=torch.int32, device=dev))
full = {"camera": (0, n_cam), "radar": (n_cam, n_rad), "lidar": (n_cam + n_rad, n_lid)}
for part in ("camera", "lidar", "radar", "all"):
    layout = {k: (v if part in (k, "all") else (v[0], 0)) for k, v in full.items()}
    head = DecoderLossHead(model, layout, 32, 1, 200, DecoderLossSettings(radar_loss_type=wl.get("radar_loss", "nll")), cnn_autocast=torch.bfloat16)
    loss = torch.zeros(1024, **f32)
    for _ in range(3):
        head.backward_into(feats, depth, times, sensor, batch, loss)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        head.backward_into(feats, depth, times, sensor, batch, loss)
    b.record()
    torch.cuda.synchronize()
    print(f"{part:8s} {a.elapsed_time(b) / 5:8.2f} ms   terms " + ", ".join(f"{k}={float(v):.4f}" for k, v in head.last["terms"].items()), flush=True)
