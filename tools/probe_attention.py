"""Radar decoder attention at the scan's size: nr_attention_fwd / _bwd against torch's fused attention (times, max error
against float64).  NR_ATT_VALU=1 selects the vector-ALU kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuradar_amd import ops  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


SHAPES = ((3531, 48), (4545, 48), (3531, 64), (3531, 32))
if len(sys.argv) == 3:
    SHAPES = ((int(sys.argv[1]), int(sys.argv[2])),)
for n, D in SHAPES:
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, n, D, device=dev, requires_grad=True) for _ in range(3))
    go = torch.randn(1, n, D, device=dev)
    ref = torch.softmax(q.double() @ k.double().transpose(1, 2) / D ** 0.5, -1) @ v.double()
    gref = torch.autograd.grad((ref * go.double()).sum(), (q, k, v))
    out = ops.attention(q, k, v)
    g = torch.autograd.grad((out * go).sum(), (q, k, v))
    sd = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    gs = torch.autograd.grad((sd * go).sum(), (q, k, v))
    err = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())  # noqa: E731
    # the library calls themselves on preallocated buffers (the autograd wrapper's host time would hide the kernels)
    lib, p, st = ops._lib.lib(), ops._p, ops._stream
    qd, kd, vd = q.detach(), k.detach(), v.detach()
    o_, lse = torch.empty_like(qd), torch.empty(1, n, device=dev)
    ws = torch.empty(int(lib.nr_attention_workspace_floats(1, n, D)), device=dev)
    gq, gk, gv = (torch.zeros_like(qd) for _ in range(3))
    fwd = lambda: lib.nr_attention_fwd(p(qd), p(kd), p(vd), 1, n, D, 0.0, 0, None, None, p(o_), p(lse), p(ws), st())  # noqa: E731
    bwd = lambda: lib.nr_attention_bwd(p(qd), p(kd), p(vd), p(o_), p(lse), p(go), 1, n, D, 0.0, 0, None, None, p(gq), p(gk), p(gv),  # noqa: E731
                                       p(ws), st())
    t_f = timeit(fwd, 50)
    t_fb = t_f + timeit(bwd, 50)
    s_f = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q.detach(), k.detach(), v.detach()))
    s_fb = timeit(lambda: torch.autograd.grad((torch.nn.functional.scaled_dot_product_attention(q, k, v) * go).sum(), (q, k, v)))
    print(f"n={n} D={D}: hip fwd {t_f:7.1f} us  fwd+bwd {t_fb:7.1f} us  err out {err(out, ref):.1e} grads "
          f"{max(err(a, b) for a, b in zip(g, gref)):.1e} | torch fwd {s_f:7.1f} us  fwd+bwd {s_fb:7.1f} us  err out {err(sd, ref):.1e} "
          f"grads {max(err(a, b) for a, b in zip(gs, gref)):.1e}")
