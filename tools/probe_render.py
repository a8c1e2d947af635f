"""GPU probe: nr_render_train (and the proposal-chain kernels) alone at the bench shape (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda")
lib, p, st = _lib.lib(), ops._p, ops._stream
B, S, C = int(os.environ.get("PROBE_B", "4096")), 32, 32
f32 = dict(device=dev, dtype=torch.float32)
torch.manual_seed(0)
alpha = torch.rand(B, S, **f32) * 0.3
feature = torch.randn(B * S, C, **f32)
sp = torch.sort(torch.rand(B, S + 1, **f32), dim=1).values
eu = sp * 80.0 + 0.5
tf, td = torch.rand(B, C, **f32), torch.rand(B, **f32) * 60
w, acc, f, d = torch.empty(B, S, **f32), torch.empty(B, **f32), torch.empty(B, C, **f32), torch.empty(B, **f32)
ga, gf, loss = torch.empty(B, S, **f32), torch.empty(B * S, C, **f32), torch.zeros(_lib.NR_LOSS_SLOTS, **f32)
fn = lambda: lib.nr_render_train(p(alpha), p(feature), p(eu), p(sp), p(tf), p(td), B, S, C, 1.0, 0.1, 0.002, p(w), p(acc), p(f),  # noqa: E731
                                 p(d), p(ga), p(gf), p(loss), None, None, None, st())
print(f"render_train B={B} S={S} C={C}: {bench.time_kernel(fn, 50) * 1e6:7.1f} us")
# inter-level loss (+ weights backward) of the two proposal levels
for Sp in (64, 128):
    cp = torch.sort(torch.rand(B, Sp + 1, **f32), dim=1).values
    eup = cp * 70 + 0.1
    dens = torch.rand(B, Sp, **f32) * 0.3
    wp = torch.empty(B, Sp, **f32)
    lib.nr_weights_from_density_fwd(p(dens), p(eup), B, Sp, p(wp), st())
    gd = torch.empty(B, Sp, **f32)
    wfin = torch.softmax(torch.randn(B, S, **f32), dim=1) * 0.9
    fn = lambda: lib.nr_interlevel_loss_to_density(p(sp), S + 1, p(wfin), S, S - 1, p(cp), p(wp), p(dens), p(eup), Sp, B, 0.03, 1.0,  # noqa: E731
                                                   p(gd), p(loss), None, st())
    print(f"interlevel_loss_to_density Sp={Sp}: {bench.time_kernel(fn, 50) * 1e6:7.1f} us")
    n, L = B * Sp, 6
    feats = torch.randn(L, n, 1, **f32)
    gfe = torch.empty_like(feats)
    wd = torch.randn(L, **f32)
    gw = torch.zeros(L, **f32)
    fn = lambda: lib.nr_prop_density_bwd(p(feats), 1, n, 1, p(wd), L, n, Sp, 1, p(dens), p(gd), p(gfe), p(gw), st())  # noqa: E731
    print(f"prop_density_bwd Sp={Sp}: {bench.time_kernel(fn, 50) * 1e6:7.1f} us")
