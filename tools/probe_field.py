"""GPU probe: time nr_field_fwd / nr_field_bwd alone on the bench workload (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuradar_amd import _lib, ops  # noqa: E402
from neuradar_amd.fused_step import FusedTrainStep  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cam4096_l16f2_w64"]
dev = torch.device("cuda")
model = bench.build_model(wl, dev, os.environ.get("PROBE_DTYPE", "float32"), float(os.environ.get("PROBE_GRAD_SCALE", "1")))
B = wl["rays"]
st = FusedTrainStep(model, B)
n = B * 32
F = st.mgrid.features_per_level
torch.manual_seed(0)
st.feats[2].normal_()
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
st.g_feature.normal_()
st.g_alpha.normal_()
from ctypes import byref  # noqa: E402
lib, p, s = st.lib, ops._p, ops._stream
lib.nr_field_pack(byref(st.field_struct), p(st.field_image), s())  # the weight image the kernels read
only = os.environ.get("PROBE_N")  # PROBE_N=<samples>: just that size, 50 calls (for a rocprofv3 --kernel-trace run)
if only:
    nn = int(only)
    for _ in range(50):
        lib.nr_field_fwd(byref(st.field_struct), p(st.feats[2]), F, n * F, F, p(d), 32, 0, nn, p(st.feature), p(st.sdf), p(st.alpha), s())
        lib.nr_field_bwd(byref(st.field_struct), p(st.feats[2]), F, n * F, F, p(d), 32, 0, nn, p(st.g_feature), p(st.g_alpha), None, p(st.g_feats[2]), byref(st.field_grads), p(st.field_ws), s())
    torch.cuda.synchronize()
    sys.exit(0)
for fb in os.environ.get("FWD_BLOCKS", "256,512").split(","):
    _lib.set_tuning("NR_FIELD_FWD_BLOCKS", int(fb))
    t = bench.time_kernel(lambda: lib.nr_field_fwd(byref(st.field_struct), p(st.feats[2]), F, n * F, F, p(d), 32, B, n, p(st.feature), p(st.sdf), p(st.alpha), s()), 20)
    print(f"field_fwd blocks={fb}: {t * 1e6:7.1f} us")
for bb in os.environ.get("BWD_BLOCKS", "128,256").split(","):
    _lib.set_tuning("NR_FIELD_BWD_BLOCKS", int(bb))
    t = bench.time_kernel(lambda: lib.nr_field_bwd(byref(st.field_struct), p(st.feats[2]), F, n * F, F, p(d), 32, B, n, p(st.g_feature), p(st.g_alpha), None, p(st.g_feats[2]), byref(st.field_grads), p(st.field_ws), s()), 20)
    print(f"field_bwd blocks={bb}: {t * 1e6:7.1f} us")
# fixed cost vs per-tile cost: shrink n at a fixed grid
for frac in (1.0, 0.5, 0.25, 0.125, 1.0 / 64):
    nn = int(n * frac) // 32 * 32
    _lib.set_tuning("NR_FIELD_FWD_BLOCKS", int("512"))
    _lib.set_tuning("NR_FIELD_BWD_BLOCKS", int("256"))
    tf = bench.time_kernel(lambda: lib.nr_field_fwd(byref(st.field_struct), p(st.feats[2]), F, n * F, F, p(d), 32, 0, nn, p(st.feature), p(st.sdf), p(st.alpha), s()), 20)
    tb = bench.time_kernel(lambda: lib.nr_field_bwd(byref(st.field_struct), p(st.feats[2]), F, n * F, F, p(d), 32, 0, nn, p(st.g_feature), p(st.g_alpha), None, p(st.g_feats[2]), byref(st.field_grads), p(st.field_ws), s()), 20)
    print(f"n={nn:7d} ({nn // 32} tiles): field_fwd {tf * 1e6:7.1f} us   field_bwd {tb * 1e6:7.1f} us")
