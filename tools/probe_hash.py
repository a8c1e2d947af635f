"""GPU probe: time the hash-grid kernels of one bench step (development tool)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cam4096_l16f2_w64"]
dev = torch.device("cuda")
model = bench.build_model(wl, dev)
scene = bench.SyntheticScene(dev, 1000)
torch.manual_seed(1)
for r in bench.roofline_probe(model, scene, wl["rays"]):
    print(f"{r['kernel']:32s} {r['seconds'] * 1e6:8.1f} us  {r['bytes'] / r['seconds'] / 1e9:8.1f} GB/s")
