"""RGB CNN decoder (models/neuradar.py:225-240) at the bench shape: torch-ROCm (MIOpen) forward + backward time under a few
settings -- the baseline a hand-written convolution has to beat (development tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuradar_amd.decoders import make_rgb_decoder  # noqa: E402

dev = "cuda"
n_p = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def run(tag, dtype, channels_last, benchmark, nhwc_loss=False, flat=False):
    torch.backends.cudnn.benchmark = benchmark
    torch.manual_seed(0)
    net = make_rgb_decoder(48).to(dev).train()
    x = torch.randn(n_p, 48, 32, 32, device=dev, requires_grad=True)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
        x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    tgt = torch.rand(n_p, 3, 96, 96, device=dev)
    if flat:
        from neuradar_amd.fused_step import flatten_parameters

        flatten_parameters(list(net.parameters()))
    tgt_nhwc = tgt.permute(0, 2, 3, 1).contiguous()

    def step():
        with torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
            y = net(x)
        loss = torch.nn.functional.mse_loss(y.float().permute(0, 2, 3, 1), tgt_nhwc) if nhwc_loss else torch.nn.functional.mse_loss(y.float(), tgt)
        loss.backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        step()
    b.record()
    torch.cuda.synchronize()
    print(f"{tag:40s} {a.elapsed_time(b) / 10:8.2f} ms / step (host {1e2 * (time.perf_counter() - t0):6.1f} ms)", flush=True)


run("bf16 + loss on the permuted (NHWC) output", torch.bfloat16, False, False, nhwc_loss=True)
run("bf16 + parameters as views of one flat buffer", torch.bfloat16, False, False, flat=True)
run("bf16 + both", torch.bfloat16, False, False, nhwc_loss=True, flat=True)
for dtype, dn in ((None, "fp32"), (torch.bfloat16, "bf16"), (torch.float16, "fp16"))[:int(os.environ.get("NR_PROBE_ALL", "0")) * 3]:
    for cl in (False, True):
        for bm in (False, True):
            try:
                run(f"{dn} channels_last={cl} benchmark={bm}", dtype, cl, bm)
            except Exception as e:  # noqa: BLE001
                print(f"{dn} channels_last={cl} benchmark={bm}: {type(e).__name__}: {e}")
