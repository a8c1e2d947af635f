"""GPU probe: what the rendering entry spends on REBUILDING weight images per reading: the field's packed
image (FusedRenderer.refresh), the BN-folded 7 x 7 images of the RGB decoder (Decoders.prepare_conv7_eval), the whole entry."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd.sensors import scale_pixel_area  # noqa: E402

dev = torch.device("cuda")
wl = bench.WORKLOADS["mixed16384_neuradar_full"]
model = bench.build_model(wl, dev, "bfloat16").eval()
scene = bench.SyntheticScene(dev, seed=1000)
H, W = scene.H, scene.W
ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
idx = torch.stack([torch.full_like(ys, 100), ys, xs], dim=-1).reshape(-1, 3)


def timed(fn, reps=9):
    fn()
    torch.cuda.synchronize()
    ts, hs = [], []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        fn()
        b.record()
        hs.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2], sorted(hs)[len(hs) // 2]


with torch.no_grad():
    b = scene.cameras.generate_rays(idx)
    scale_pixel_area(b)
    b.metadata["sensor_idxs"] = torch.zeros_like(b.pixel_area, dtype=torch.int64)
    entry = timed(lambda: model.get_outputs_for_camera_ray_bundle(b, image_shape=(H, W)))
    fr = model._fused_renderer(32768)
    refresh = timed(fr.refresh)
    fold = timed(lambda: model._decoders.prepare_conv7_eval(torch.bfloat16))
print(f"entry {entry[0]:.3f} ms (host {entry[1]:.3f}); field image refresh {refresh[0]:.3f} ms (host {refresh[1]:.3f}); "
      f"conv7 fold + pack {fold[0]:.3f} ms (host {fold[1]:.3f})")
