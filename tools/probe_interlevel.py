"""GPU probe: nr_interlevel_loss_to_density time vs number of rays / proposal samples (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda")
lib, p, st = _lib.lib(), ops._p, ops._stream
f32 = dict(device=dev, dtype=torch.float32)
torch.manual_seed(0)
S = 32
for B in (256, 1024, 4096, 16384):
    for Sp in (8, 64, 128):
        sp = torch.sort(torch.rand(B, S + 1, **f32), dim=1).values
        wfin = torch.softmax(torch.randn(B, S, **f32), dim=1) * 0.9
        cp = torch.sort(torch.rand(B, Sp + 1, **f32), dim=1).values
        eup = cp * 70 + 0.1
        dens = torch.rand(B, Sp, **f32) * 0.3
        wp = torch.empty(B, Sp, **f32)
        lib.nr_weights_from_density_fwd(p(dens), p(eup), B, Sp, p(wp), st())
        gd, loss = torch.empty(B, Sp, **f32), torch.zeros(_lib.NR_LOSS_SLOTS, **f32)
        fn = lambda: lib.nr_interlevel_loss_to_density(p(sp), S + 1, p(wfin), S, S - 1, p(cp), p(wp), p(dens), p(eup), Sp, B, 0.03, 1.0,  # noqa: E731
                                                       p(gd), p(loss), None, st())
        gw = torch.empty(B, Sp, **f32)
        fn2 = lambda: lib.nr_interlevel_loss(p(sp), S + 1, p(wfin), S, S - 1, p(cp), p(wp), Sp, B, 0.03, 1.0, p(gw), p(loss), st())  # noqa: E731
        fn3 = lambda: lib.nr_weights_from_density_bwd(p(dens), p(eup), p(gw), B, Sp, p(gd), st())  # noqa: E731
        print(f"B={B:6d} Sp={Sp:4d}: fused {bench.time_kernel(fn, 30) * 1e6:7.1f} us   interlevel only {bench.time_kernel(fn2, 30) * 1e6:7.1f} us   "
              f"weights_bwd only {bench.time_kernel(fn3, 30) * 1e6:7.1f} us")
