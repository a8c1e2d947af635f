"""GPU lab: time candidate scatter kernels (tools/scatter_lab.hip) against the production
nr_hash_encode_bwd on the bench workload's real sample positions, and check they agree.
Build here (no GPU needed):  python tools/scatter_lab.py --build
Run on the GPU box:          python tools/scatter_lab.py [workload]
Development tool (not part of tests/bench)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "scatter_lab.so")

if "--build" in sys.argv:
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-munsafe-fp-atomics",
                           "-ffp-contract=off", "-o", SO, os.path.join(ROOT, "tools", "scatter_lab.hip")])
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, ROOT)
import bench  # noqa: E402
from neuradar_amd import ops  # noqa: E402
from neuradar_amd.sensors import scale_pixel_area  # noqa: E402

lab = ctypes.CDLL(SO)
P, I, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lab.lab_scatter.argtypes = [I, P, P, P, I, I, I, P, L64, L64, P, L64, I, I, P]
lab.lab_scatter.restype = I

args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl_name = args[0] if args else "cam4096_l16f2_w64"
wl = bench.WORKLOADS[wl_name]
dev = torch.device("cuda")
model = bench.build_model(wl, dev)
scene = bench.SyntheticScene(dev, 1000)
torch.manual_seed(1)
with torch.no_grad():
    bundle = scene.cameras.generate_rays(scene.sample_ray_indices(wl["rays"]))
    scale_pixel_area(bundle)
    out = model.get_nff_outputs(bundle)
lib, p, st = ops._lib.lib(), ops._p, ops._stream
for tag, fld, rs in [("prop_s128", model.proposal_fields[1], out["ray_samples_list"][0]),
                     ("prop_s64", model.proposal_fields[1], out["ray_samples_list"][1]),
                     ("main_s32", model.field, out["ray_samples"])]:
    g = fld.hashgrid.static_grid
    B, S = rs.shape
    n, L, F = B * S, g.num_levels, g.features_per_level
    x01, std01 = ops.contract_gaussians(rs.origins, rs.directions, rs.pixel_area, rs.euclid, fld.hashgrid.static_scale)
    gbuf = torch.randn((L, n, F), device=dev)
    log2t = g.log2_hashmap_size

    def prod(gt, lv=None):
        sc, gb, nl = (g.scalings, gbuf, L) if lv is None else (g.scalings[lv:lv + 1].contiguous(), gbuf[lv:lv + 1], 1)
        gt_ = gt if lv is None else gt[lv << log2t:(lv + 1) << log2t]
        return lib.nr_hash_encode_bwd(p(x01), p(std01), p(sc), nl, F, log2t, p(gb), F, n * F, p(gt_), n, S, st())

    def cand(variant, gt, flags=0, lv=None):
        sc, gb, nl = (g.scalings, gbuf, L) if lv is None else (g.scalings[lv:lv + 1].contiguous(), gbuf[lv:lv + 1], 1)
        gt_ = gt if lv is None else gt[lv << log2t:(lv + 1) << log2t]
        return lab.lab_scatter(variant, p(x01), p(std01), p(sc), nl, F, log2t, p(gb), F, n * F, p(gt_), n, S, flags, st())

    ref = torch.zeros_like(g.hash_table)
    assert prod(ref) == 0
    if "--pmc" in sys.argv:  # a few launches only, for rocprofv3 --pmc passes
        for _ in range(3):
            prod(ref)
        torch.cuda.synchronize()
        continue
    scratch = torch.zeros_like(g.hash_table)
    t_prod = bench.time_kernel(lambda: prod(scratch), 20)
    print(f"== {tag}: n={n} L={L} F={F} T=2^{log2t}  production {t_prod * 1e6:7.1f} us")
    per_level = [bench.time_kernel(lambda lv=lv: prod(scratch, lv), 10) * 1e6 for lv in range(L)]
    print("   production per level:", " ".join(f"{t:6.1f}" for t in per_level))
    # the same kernels on inputs STORED sample-major ([S,B,...] order, identity walk): coalesced reads
    x_sm = x01.view(B, S, 3).permute(1, 0, 2).contiguous().view(n, 3)
    std_sm = std01.view(B, S).t().contiguous().view(n)
    g_sm = gbuf.view(L, B, S, F).permute(0, 2, 1, 3).contiguous().view(L, n, F)
    got = torch.zeros_like(g.hash_table)
    lib.nr_hash_encode_bwd(p(x_sm), p(std_sm), p(g.scalings), L, F, log2t, p(g_sm), F, n * F, p(got), n, 0, st())
    torch.cuda.synchronize()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    t_sm = bench.time_kernel(lambda: lib.nr_hash_encode_bwd(p(x_sm), p(std_sm), p(g.scalings), L, F, log2t, p(g_sm), F, n * F,
                                                            p(scratch), n, 0, st()), 20)
    lv_t = []
    for lv in range(L):
        sc = g.scalings[lv:lv + 1].contiguous()
        lv_t.append(bench.time_kernel(lambda: lib.nr_hash_encode_bwd(p(x_sm), p(std_sm), p(sc), 1, F, log2t, p(g_sm[lv]), F, n * F,
                                                                     p(scratch[lv << log2t:]), n, 0, st()), 10) * 1e6)
    print(f"   production, sample-major STORAGE: {t_sm * 1e6:7.1f} us rel.err {err:.2e} per level: " + " ".join(f"{t:6.1f}" for t in lv_t))
    obuf = torch.empty((L, n, F), device=dev)
    t_f = bench.time_kernel(lambda: lib.nr_hash_encode_fwd(p(x01), p(std01), p(g.hash_table), p(g.scalings), L, F, log2t, p(obuf),
                                                           F, n * F, n, S, st()), 20)
    t_fsm = bench.time_kernel(lambda: lib.nr_hash_encode_fwd(p(x_sm), p(std_sm), p(g.hash_table), p(g.scalings), L, F, log2t,
                                                             p(obuf), F, n * F, n, 0, st()), 20)
    print(f"   forward: ray-major storage {t_f * 1e6:7.1f} us, sample-major storage {t_fsm * 1e6:7.1f} us")
    if "--all" not in sys.argv:
        continue
    for variant in range(8):
        got = torch.zeros_like(g.hash_table)
        rc = cand(variant, got)
        if rc == -1:
            continue
        torch.cuda.synchronize()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        t = bench.time_kernel(lambda: cand(variant, scratch), 20)
        t_noat = bench.time_kernel(lambda: cand(variant, scratch, 1), 20)
        lv_t = [bench.time_kernel(lambda lv=lv: cand(variant, scratch, 0, lv), 10) * 1e6 for lv in range(L)]
        print(f"   v7[{variant}] rc={rc} {t * 1e6:7.1f} us  (no global atomics {t_noat * 1e6:7.1f} us)  rel.err {err:.2e}  per level: "
              + " ".join(f"{t:6.1f}" for t in lv_t))
