"""CPU oracle for the NeuRadar volumetric-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU (torch fp32) restatement of the reference's *pure-PyTorch* path
(`implementation="torch"`).  It exists to check the HIP kernels:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * the product package `neuradar_amd/` never imports it and has no CPU fallback;
  * every function cites the reference file:line it restates (paths relative to
    /root/reference/nerfstudio/).

Pinning: the reference's own tests hold no numeric vectors for this path, so the
oracle is pinned against golden vectors produced by importing the reference itself in the build
container (`tests/golden/make_golden.py`, fixtures in `tests/golden/*.npz`; checked by
`tests/test_oracle_golden.py`).  One piece is NOT pinned that way: the `nerfacc==0.5.2` batched
compositing helpers (`render_weight_from_alpha`, `accumulate_along_rays`) are an un-vendored
third-party dependency absent from /root/reference; `oracle.render` restates their published
algorithm and is cross-checked against the in-repo near-equivalent `RaySamples.
get_weights_and_transmittance_from_alphas` (cameras/rays.py:226-248) -> "parity unpinned" for
that boundary only.

Floating point throughout (fp32); the only integer work is the spatial hash.
"""
from . import hashgrid, field, sampler, render, raygen, losses, pipeline  # noqa: F401
