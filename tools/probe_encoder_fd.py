"""Directional derivatives of the fused encoder layer (development probe): finite differences against the backward, with and
without dropout, for several step sizes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuradar_amd import ops
from neuradar_amd.decoders import Transformer

DEV = "cuda"
torch.manual_seed(0)
m = Transformer(d_model=48, dim_feedforward=64, dropout=0.1, attention="hip").to(DEV).train()
with torch.no_grad():
    for prm in m.parameters():
        prm.add_(0.2 * torch.randn_like(prm))
lyr = m.encoder.layers[0]
torch.manual_seed(3)
src, pos, wgt = (torch.randn(1, 256, 48, device=DEV) for _ in range(3))
epoch = torch.full((1,), 5.0, device=DEV)
for p in (0.0, 0.1):
    f = lambda x: ops.encoder_layer(x, pos, lyr, m.encoder.norm, p, seed=1234, seed_epoch=epoch if p > 0 else None)
    x = src.clone().requires_grad_(True)
    (f(x).double() * wgt.double()).sum().backward()
    for trial in range(3):
        d = torch.randn_like(src); d /= d.norm()
        ana = float((x.grad.double() * d.double()).sum())
        for eps in (1e-1, 3e-2, 1e-2, 3e-3):
            with torch.no_grad():
                fp = (f(src + eps * d).double() * wgt.double()).sum(); fm = (f(src - eps * d).double() * wgt.double()).sum()
            print(f"p={p} trial {trial} eps={eps:g}: numeric {float((fp - fm) / (2 * eps)):+.5f} analytic {ana:+.5f}")
# the attention alone with dropout
q, k, v = (torch.randn(1, 256, 48, device=DEV) for _ in range(3))
for p in (0.0, 0.1):
    g = lambda q_: ops.attention(q_, k, v, p, seed=77)
    x = q.clone().requires_grad_(True)
    (g(x).double() * wgt.double()).sum().backward()
    d = torch.randn_like(q); d /= d.norm()
    ana = float((x.grad.double() * d.double()).sum())
    for eps in (1e-1, 1e-2):
        with torch.no_grad():
            fp = (g(q + eps * d).double() * wgt.double()).sum(); fm = (g(q - eps * d).double() * wgt.double()).sum()
        print(f"attention p={p} eps={eps:g}: numeric {float((fp - fm) / (2 * eps)):+.5f} analytic {ana:+.5f}")
