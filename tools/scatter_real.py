"""GPU lab: scatter kernels on the REAL per-sample gradients of one fused step (development tool).
Needs tools/scatter_lab.so (python tools/scatter_lab.py --build)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from neuradar_amd import ops  # noqa: E402
from neuradar_amd.fused_step import FusedTrainStep  # noqa: E402

lab = ctypes.CDLL(os.path.join(ROOT, "tools", "scatter_lab.so"))
P, I, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lab.lab_scatter.argtypes = [I, P, P, P, I, I, I, P, L64, L64, P, L64, I, I, P]
lab.lab_scatter.restype = I

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cam4096_l16f2_w64"]
dev = torch.device("cuda")
model = bench.build_model(wl, dev)
scene = bench.SyntheticScene(dev, 1000)
B = wl["rays"]
torch.manual_seed(1)
tg = (0.1 * torch.randn(B, 32, device=dev), 5.0 + 50.0 * torch.rand(B, 1, device=dev))
stp = FusedTrainStep(model, B)
S0 = stp.S[0]
n_p = B // (scene.PATCH * scene.PATCH)
r = torch.rand(B * (S0 + 1) + 2 * B + 3 * n_p, device=dev)
n_t = B * (S0 + 1)
bundle, _ = scene.cameras.generate_patch_rays(r[n_t + 2 * B:].view(n_p, 3), scene.PATCH, scene.STRIDE, scene.H, scene.W, area_scale=9.0)
stp.forward_backward(bundle.origins, bundle.directions, bundle.pixel_area[:, 0], None, tg[0], tg[1][:, 0],
                     r[:n_t].view(B, S0 + 1), r[n_t:n_t + B], r[n_t + B:n_t + 2 * B])
torch.cuda.synchronize()
lib, p, st = ops._lib.lib(), ops._p, ops._stream
for lvl, tag in ((0, "prop_s128"), (1, "prop_s64"), (2, "main_s32")):
    g = stp.pgrid if lvl < 2 else stp.mgrid
    n, Lv, F = B * stp.S[lvl], g.num_levels, g.features_per_level
    x01, std, gbuf = stp.x01[lvl], stp.std[lvl], stp.g_feats[lvl]
    nzfrac = float((gbuf != 0).float().mean())
    scratch = torch.zeros_like(g.hash_table)
    prod = lambda: lib.nr_hash_encode_bwd(p(x01), p(std), p(g.scalings), Lv, F, g.log2_hashmap_size, p(gbuf), F, n * F, p(scratch), n, 0, st())  # noqa: E731
    t = bench.time_kernel(prod, 20)
    grand = torch.randn_like(gbuf)
    prod_r = lambda: lib.nr_hash_encode_bwd(p(x01), p(std), p(g.scalings), Lv, F, g.log2_hashmap_size, p(grand), F, n * F, p(scratch), n, 0, st())  # noqa: E731
    tr = bench.time_kernel(prod_r, 20)
    print(f"{tag}: real gradients ({nzfrac * 100:.1f} % non-zero) {t * 1e6:7.1f} us; random gradients {tr * 1e6:7.1f} us")
    for variant in (0, 1, 2, 3):
        for flags, what in ((0, "with atomics"), (1, "no global atomics")):
            fn = lambda: lab.lab_scatter(variant, p(x01), p(std), p(g.scalings), Lv, F, g.log2_hashmap_size, p(gbuf), F, n * F, p(scratch), n, 0, flags, st())  # noqa: E731
            if fn() != 0:
                continue
            print(f"   lab v8[{variant}] {what}: {bench.time_kernel(fn, 20) * 1e6:7.1f} us")
    ref = torch.zeros_like(g.hash_table)
    prod_into = lambda gt: lib.nr_hash_encode_bwd(p(x01), p(std), p(g.scalings), Lv, F, g.log2_hashmap_size, p(gbuf), F, n * F, p(gt), n, 0, st())  # noqa: E731
    prod_into(ref)
    for variant in (1, 2, 3):
        got = torch.zeros_like(g.hash_table)
        if lab.lab_scatter(variant, p(x01), p(std), p(g.scalings), Lv, F, g.log2_hashmap_size, p(gbuf), F, n * F, p(got), n, 0, 0, st()) == 0:
            torch.cuda.synchronize()
            print(f"   lab variant {variant} max rel err vs production: {float((got - ref).abs().max() / ref.abs().max()):.2e}")
    # per-level on real gradients
    T = 1 << g.log2_hashmap_size
    lv = []
    for l in range(Lv):
        sc = g.scalings[l:l + 1].contiguous()
        fn = lambda: lib.nr_hash_encode_bwd(p(x01), p(std), p(sc), 1, F, g.log2_hashmap_size, p(gbuf[l]), F, n * F, p(scratch[l * T:]), n, 0, st())  # noqa: E731
        lv.append(bench.time_kernel(fn, 10) * 1e6)
    print("   per level:", " ".join(f"{v:6.1f}" for v in lv))
