"""GPU probe: where the rendering entry's time goes on one 1920 x 1080 camera image (230 400 rays): ray generation of the full
image, the field in chunks (fused chain / modular modules), the lidar decoder over every ray, the RGB CNN over the feature image."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd.sensors import scale_pixel_area  # noqa: E402

dev = torch.device("cuda")
wl = bench.WORKLOADS["mixed16384_neuradar_full"]
model = bench.build_model(wl, dev, os.environ.get("PROBE_DTYPE", "bfloat16")).eval()
scene = bench.SyntheticScene(dev, seed=1000)
n_train = int(os.environ.get("PROBE_TRAIN_STEPS", "0"))
if n_train:  # the model as bench.py renders it: after some fused training steps (decoders in the step, optimizers through raw pointers)
    from neuradar_amd.parallel import GradAllReducer

    model.train()
    opts = bench.build_optimizers(model)
    red = GradAllReducer(None, buffers=[g for o in opts for g in o.grad_buffers()])
    targets = (0.1 * torch.randn(wl["rays"], 32, device=dev), 5.0 + 50.0 * torch.rand(wl["rays"], 1, device=dev))
    fwd_bwd, optim, st = bench.make_step(model, scene, opts, red, targets, wl["rays"], fused=True, fuse_optimizer=True, mixed=wl)
    for _ in range(n_train):
        fwd_bwd()
    torch.cuda.synchronize()
    model.eval()
H, W = scene.H, scene.W
ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
idx = torch.stack([torch.full_like(ys, 100), ys, xs], dim=-1).reshape(-1, 3)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2], r


with torch.no_grad():
    def gen():
        b = scene.cameras.generate_rays(idx)
        scale_pixel_area(b)
        b.metadata["sensor_idxs"] = torch.zeros_like(b.pixel_area, dtype=torch.int64)
        return b

    t_gen, bundle = timed(gen)
    t_all, out = timed(lambda: model.get_outputs_for_camera_ray_bundle(gen(), image_shape=(H, W)))
    feats = out["features"].reshape(-1, out["features"].shape[-1])
    n = feats.shape[0]
    take = (torch.arange(1, H, 3, device=dev)[:, None] * W + torch.arange(1, W, 3, device=dev)[None, :]).reshape(-1)
    o, d, a = bundle.origins[take], bundle.directions[take], bundle.pixel_area[take]
    fr = model._fused_renderer(32768)
    flat = {"features": torch.empty(n, 32, device=dev), **{k: torch.empty(n, 1, device=dev) for k in ("depth", "accumulation", "prop_depth_0", "prop_depth_1")}}

    def field():
        for lo in range(0, n, 32768):
            hi = min(lo + 32768, n)
            fr.render(o[lo:hi], d[lo:hi], a[lo:hi], None, flat, lo)

    t_field, _ = timed(field)
    dec = model._decoders
    t_lidar, _ = timed(lambda: dec.lidar_decoder(feats))
    patch = feats.view(1, H // 3, W // 3, feats.shape[-1]).permute(0, 3, 1, 2)
    t_cnn, _ = timed(lambda: dec.rgb_decoder(patch))
    t_cnn_cl, _ = timed(lambda: dec.rgb_decoder(patch.contiguous(memory_format=torch.channels_last)))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        t_cnn16, _ = timed(lambda: dec.rgb_decoder(patch))
    t_take, _ = timed(lambda: (bundle.origins[take], bundle.directions[take], bundle.pixel_area[take]))
print(f"rays {n}: whole entry {t_all:.2f} ms = ray generation of {H * W} rays {t_gen:.2f} + strided selection {t_take:.2f} + field (fused chain, "
      f"{(n + 32767) // 32768} chunks) {t_field:.2f} + lidar decoder {t_lidar:.2f} + RGB CNN {t_cnn:.2f} (channels-last copy first: {t_cnn_cl:.2f}; "
      f"autocast bf16: {t_cnn16:.2f}) ms")
if os.environ.get("PROBE_BENCH_RENDER", "1") == "1":  # the same entry through bench.py's own measurement, in this process
    r = bench.measure_render(model, scene, dev)
    print("bench.measure_render:", r["camera_image"]["ms"], "ms camera image,", r["radar_scan"]["ms"], "ms radar scan")

