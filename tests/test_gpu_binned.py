"""GPU parity of the two-pass binned scatter-add (nr_hash_encode_bwd_binned, for incoherent rows) against the oracle at
small sizes and against the merging kernel (nr_hash_encode_bwd) at BASELINE sizes; bit-exact work: the set of rows
written."""
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _binned(x, std, sc, log2t, gout_rows, F, rows, sum_bits=64):
    """gout_rows [n, L*F] -> grad_table [rows, F] through the binned kernels."""
    from neuradar_amd import ops

    gt = torch.zeros(rows, F, device=DEV)
    L = sc.numel()
    ops.hash_encode_bwd_binned(x, std, sc, log2t, gout_rows.contiguous(), (L * F, F), F, gt, sum_bits=sum_bits)
    return gt


@pytest.mark.parametrize("F,L,log2t,n", [(1, 6, 20, 4661 * 128), (1, 3, 17, 30_000), (2, 2, 15, 5_003)])
def test_binned_scatter_with_32_bit_tile_sums_stays_within_its_stated_bound(F, L, log2t, n):
    """sum_bits = 32 (the companion of 16-bit MLP operands): every addend is rounded to a multiple of 2^(e - 21), 2^e the power
    of two above the largest contribution of its 512-row tile -- at most 2^-21 of the tile's largest per addend, half of that
    on average.  Checked against the 64-bit path on the same rows: the difference of an entry stays below
    (addends into it) x 2^-21 x (largest single contribution anywhere), 2^-21 being 4.8e-7; in relative L2 the two gradients
    agree to 1e-5.  No entry is written that no row touches."""
    from oracle import hashgrid

    torch.manual_seed(F + L + n)
    sc = hashgrid.level_scalings(L, 16, 2048 if L < 6 else 4096).to(DEV)
    x = torch.rand(n, 3, device=DEV)
    std = 0.002 * torch.rand(n, device=DEV)
    gout = torch.randn(n, L * F, device=DEV) * torch.exp(2.0 * torch.randn(n, 1, device=DEV))  # magnitudes over two decades
    rows = L << log2t
    exact = _binned(x, std, sc, log2t, gout, F, rows, 64)
    lp = _binned(x, std, sc, log2t, gout, F, rows, 32)
    counts = _binned(x, std, sc, log2t, torch.ones_like(gout), F, rows, 64)  # sum of the trilinear weights >= (addends) / 8 ... a proxy
    biggest = float(gout.abs().max())  # a contribution is g * w * rescale with w, rescale <= 1
    err = (lp - exact).abs()
    bound = (8.0 * counts.abs() + 8.0) * 2.0 ** -21 * biggest
    assert bool((err <= bound).all()), f"worst excess {float((err - bound).max()):.3e}"
    rel = float((lp - exact).norm() / exact.norm())
    assert rel < 1e-5, rel
    # No entry is written that no row touches.  (Not "that the 64-bit path leaves at zero": the two launches are not bit-reproducible
    # from run to run -- which contributions find a slot in a block's LDS table and which overflow to direct float atomics depends
    # on timing -- and a contribution of 1.6e-7 beside ones of 3e4 is below the fixed-point resolution of its tile in one run and
    # added directly in the next, in EITHER path: measured, 1 run in ~4 had such an entry.)
    assert not bool(((lp != 0) & (counts == 0)).any())


@pytest.mark.parametrize("F", [1, 2, 4])
@pytest.mark.parametrize("case", ["uniform", "one_cell", "ragged_tiny_table"])
def test_binned_scatter_vs_oracle(F, case):
    """uniform: every bucket of every level in use; one_cell: all rows in ONE cell (bins and queue regions overflow ->
    the direct-atomic fallback carries most of the sum); ragged_tiny_table: n not a multiple of 64 / 1024, table smaller
    than one slice."""
    from oracle import hashgrid

    torch.manual_seed(10 * F + len(case))
    L, log2t, n = (3, 17, 30_000) if case != "ragged_tiny_table" else (2, 9, 1025 + 37)
    sc = hashgrid.level_scalings(L, 16, 2048)
    x = torch.rand(n, 3)
    if case == "one_cell":
        x = (torch.tensor([0.3137, 0.7211, 0.5003]) + 1e-6 * torch.rand(n, 3)).clamp(0, 1)
    x[:5] = torch.tensor([0.0, 0.25, 0.5])  # exact grid planes
    std = 0.01 * torch.rand(n)
    gout = torch.randn(n, L * F)
    gout[::7] = 0.0  # zero gradients are skipped, not binned
    table = hashgrid.init_table(L, log2t, F, scale=1.0).requires_grad_(True)
    resc = 1.0 / torch.clamp(2.0 * sc[None, :] * std[:, None], min=1.0)  # neurad_encoding.py:309-316
    ref = hashgrid.encode(x, table, sc, 2**log2t).view(n, L, F) * resc[:, :, None]
    (gref,) = torch.autograd.grad(ref, table, gout.view(n, L, F))
    gt = _binned(x.to(DEV), std.to(DEV), sc.to(DEV), log2t, gout.to(DEV), F, table.shape[0])
    assert_close(gt.cpu(), gref, rtol=1e-4, atol_scale=2e-6, what=f"binned F={F} {case}")
    assert torch.equal(gt.cpu() != 0, gref != 0) or case == "one_cell", "rows written differ from the oracle's"


@pytest.mark.parametrize("cfg", [("prop", 6, 1, 20, 128, 4096, 4661 * 128), ("l16f2", 16, 2, 19, 16, 1024, 4661 * 32)])
def test_binned_scatter_equals_merging_kernel_at_baseline_sizes(cfg):
    """At BASELINE table sizes and the lidar share of the configs[2] batch (4 661 rays), on incoherent rows: the two
    scatter implementations agree (rtol 1e-4 of the gradient's scale) and write the same set of rows."""
    from neuradar_amd import ops
    from oracle import hashgrid

    tag, L, F, log2t, rmin, rmax, n = cfg
    torch.manual_seed(L + F)
    sc = hashgrid.level_scalings(L, rmin, rmax).to(DEV)
    x = torch.rand(n, 3, device=DEV)
    std = 0.002 * torch.rand(n, device=DEV)
    gout = torch.randn(n, L * F, device=DEV)
    rows = L << log2t
    want = torch.zeros(rows, F, device=DEV)
    lib, p, st = ops._lib.lib(), ops._p, ops._stream
    ops.check(lib.nr_hash_encode_bwd(p(x), p(std), p(sc), L, F, log2t, p(gout), L * F, F, p(want), n, 0, st()), "bwd")
    got = _binned(x, std, sc, log2t, gout, F, rows)
    assert_close(got.cpu(), want.cpu(), rtol=1e-4, atol_scale=1e-5, what=tag)
    assert torch.equal(got != 0, want != 0), f"{tag}: different rows written"


def test_binned_scatter_rejects_tables_with_too_many_slices():
    from neuradar_amd import _lib

    assert _lib.lib().nr_hash_encode_bwd_binned_workspace_bytes(8, 4, 22, 1000) == -1  # NeuRadar main grid: 512 slices per level
    assert _lib.lib().nr_hash_encode_bwd_binned_workspace_bytes(6, 1, 20, 1000) > 0


@pytest.mark.parametrize("F", [1, 2])
@pytest.mark.parametrize("run", [3, 17, 40, 64, 200])
def test_scatters_sum_long_runs_of_one_cell(F, run):
    """Rows come in runs of `run` consecutive rows inside one cell (samples of a ray in a coarse cell; camera pixels at a
    coarse level): both scatter kernels sum such runs across the wave's lanes before anything leaves the wave -- runs
    shorter than, equal to and longer than a row of 16 lanes, a half wave and a wave."""
    from neuradar_amd import ops
    from oracle import hashgrid

    torch.manual_seed(run + F)
    L, log2t = 2, 14
    n = run * 97 + 5
    sc = hashgrid.level_scalings(L, 16, 64)
    cells = torch.rand(n // run + 1, 3)
    x = (cells.repeat_interleave(run, dim=0)[:n] + 1e-5 * torch.rand(n, 3)).clamp(0, 1)
    gout = torch.randn(n, L * F)
    table = hashgrid.init_table(L, log2t, F, scale=1.0).requires_grad_(True)
    ref = hashgrid.encode(x, table, sc, 2**log2t)
    (gref,) = torch.autograd.grad(ref, table, gout)
    xd, scd, gd = x.to(DEV), sc.to(DEV), gout.to(DEV)
    got_binned = _binned(xd, None, scd, log2t, gd, F, table.shape[0])
    got_merge = torch.zeros_like(got_binned)
    lib, p, st = ops._lib.lib(), ops._p, ops._stream
    ops.check(lib.nr_hash_encode_bwd(p(xd), None, p(scd), L, F, log2t, p(gd), L * F, F, p(got_merge), n, 0, st()), "bwd")
    assert_close(got_merge.cpu(), gref, rtol=1e-4, atol_scale=1e-5, what=f"merging kernel, runs of {run}")
    assert_close(got_binned.cpu(), gref, rtol=1e-4, atol_scale=1e-5, what=f"binned kernel, runs of {run}")


@pytest.mark.parametrize("sample_major", [0, 1])
def test_density_head_folded_into_the_binned_scatter(sample_major):
    """nr_prop_density_scatter_binned == nr_prop_density_bwd followed by nr_hash_encode_bwd_binned (the proposal field's
    backward, neurad_field.py:208-213): same table gradient (rtol 1e-4 of its scale; the per-row g is computed by the same
    expressions, the sums differ in order only) and same weight gradient of the density head, with ray-major and
    sample-major rows; rows with g_density == 0 (masked) contribute nothing."""
    from neuradar_amd import ops
    from oracle import hashgrid

    torch.manual_seed(5 + sample_major)
    L, F, log2t, B, S = 6, 1, 20, 1500, 64
    n = B * S
    sm = B if sample_major else 0
    sc = hashgrid.level_scalings(L, 128, 4096).to(DEV)
    x = torch.rand(n, 3, device=DEV)
    std = 0.002 * torch.rand(n, device=DEV)
    feats = 0.3 * torch.randn(L, n, F, device=DEV)
    w = torch.randn(L * F, device=DEV)
    g_density = torch.randn(B, S, device=DEV)
    g_density[::5] = 0.0
    rows = L << log2t
    lib, p, st = ops._lib.lib(), ops._p, ops._stream
    need = lib.nr_hash_encode_bwd_binned_workspace_bytes(L, F, log2t, n)
    ws = torch.empty(need, device=DEV, dtype=torch.uint8)
    # two steps
    g_feats = torch.empty_like(feats)
    gw_ref, gt_ref = torch.zeros_like(w), torch.zeros(rows, F, device=DEV)
    ops.check(lib.nr_prop_density_bwd(p(feats), F, n * F, F, p(w), L * F, n, S, sm, None, p(g_density), p(g_feats), p(gw_ref), st()), "dens_bwd")
    ops.check(lib.nr_hash_encode_bwd_binned(p(x), p(std), p(sc), L, F, log2t, p(g_feats), F, n * F, p(gt_ref), n, p(ws), st()), "binned")
    # one pass
    gw, gt = torch.full_like(w, 0.5), torch.zeros(rows, F, device=DEV)
    ops.check(lib.nr_prop_density_scatter_binned(p(x), p(std), p(sc), L, F, log2t, p(feats), F, n * F, p(w), p(g_density), S, sm,
                                                 p(gt), p(gw), n, p(ws), st()), "fused")
    assert_close(gt.cpu(), gt_ref.cpu(), rtol=1e-4, atol_scale=1e-5, what="table gradient")
    assert torch.equal(gt != 0, gt_ref != 0)
    assert_close((gw - 0.5).cpu(), gw_ref.cpu(), rtol=1e-4, atol_scale=1e-5, what="head weight gradient (accumulated onto g_w)")


def test_binned_entry_points_reject_bad_arguments_and_accept_empty_input():
    """n = 0 is a no-op (0); NULL pointers, a misaligned workspace, more than 8 levels for the folded density head and a
    row order that does not fit n are NR_EINVAL, like every other entry of the ABI."""
    from neuradar_amd import ops

    lib, p, st = ops._lib.lib(), ops._p, ops._stream
    L, F, log2t, n, S = 6, 1, 12, 640, 64
    x, sd, sc = torch.rand(n, 3, device=DEV), torch.rand(n, device=DEV), torch.arange(1, L + 1, device=DEV).float() * 16
    feats, w, gd = torch.randn(L, n, F, device=DEV), torch.randn(L * F, device=DEV), torch.randn(n // S, S, device=DEV)
    gt, gw = torch.zeros(L << log2t, F, device=DEV), torch.zeros(L * F, device=DEV)
    ws = torch.empty(lib.nr_hash_encode_bwd_binned_workspace_bytes(L, F, log2t, n) + 16, device=DEV, dtype=torch.uint8)
    args = lambda n_=n, ws_=ws, L_=L, sm=0: (p(x), p(sd), p(sc), L_, F, log2t, p(feats), F, n * F, p(w), p(gd), S, sm, p(gt), p(gw), n_, p(ws_), st())  # noqa: E731
    assert lib.nr_prop_density_scatter_binned(*args(n_=0)) == 0 and float(gt.abs().sum()) == 0.0
    assert lib.nr_prop_density_scatter_binned(*args()) == 0 and float(gt.abs().sum()) > 0.0
    assert lib.nr_prop_density_scatter_binned(*args(ws_=ws[4:])) != 0          # workspace not 16-byte aligned
    assert lib.nr_prop_density_scatter_binned(*args(sm=n // S + 1)) != 0      # more sample-major rays than rays
    assert lib.nr_prop_density_scatter_binned(p(x), p(sd), p(sc), L, F, log2t, p(feats), F, n * F, None, p(gd), S, 0, p(gt), p(gw), n, p(ws), st()) != 0
    assert lib.nr_hash_encode_bwd_binned(p(x), p(sd), p(sc), L, F, log2t, None, F, n * F, p(gt), n, p(ws), st()) != 0
    assert lib.nr_hash_encode_bwd_binned_workspace_bytes(9, F, log2t, n) > 0   # the plain scatter takes any number of levels ...
    sc9, f9 = torch.arange(1, 10, device=DEV).float() * 16, torch.randn(9, n, F, device=DEV)
    ws9 = torch.empty(lib.nr_hash_encode_bwd_binned_workspace_bytes(9, F, log2t, n), device=DEV, dtype=torch.uint8)
    assert lib.nr_prop_density_scatter_binned(p(x), p(sd), p(sc9), 9, F, log2t, p(f9), F, n * F, p(torch.randn(9, device=DEV)), p(gd), S, 0,
                                              p(torch.zeros(9 << log2t, F, device=DEV)), p(torch.zeros(9, device=DEV)), n, p(ws9), st()) != 0  # ... the folded head at most 8


@pytest.mark.parametrize("rows", ["incoherent", "runs"])
def test_wide_merging_configuration_equals_the_default(rows):
    """nr_hash_encode_bwd_tuned(wave_cells=256) -- the F = 4 configuration the fused step uses for batches with lidar / radar
    rays -- sums the same gradient as the default configuration (rtol 1e-4 of its scale) and writes the same rows, on
    incoherent rows and on runs of rows in one cell, at the NeuRadar main grid's size; other widths ignore the argument."""
    from neuradar_amd import ops
    from oracle import hashgrid

    torch.manual_seed(4)
    L, F, log2t, n = 8, 4, 22, 4661 * 32 + 7
    sc = hashgrid.level_scalings(L, 32, 8192).to(DEV)
    x = torch.rand(n, 3, device=DEV)
    if rows == "runs":
        x = (torch.rand(n // 24 + 1, 3, device=DEV).repeat_interleave(24, dim=0)[:n] + 2e-5 * torch.rand(n, 3, device=DEV)).clamp(0, 1)
    std = 0.002 * torch.rand(n, device=DEV)
    gout = torch.randn(n, L * F, device=DEV)
    lib, p, st = ops._lib.lib(), ops._p, ops._stream
    want, got = torch.zeros(L << log2t, F, device=DEV), torch.zeros(L << log2t, F, device=DEV)
    ops.check(lib.nr_hash_encode_bwd(p(x), p(std), p(sc), L, F, log2t, p(gout), L * F, F, p(want), n, 0, st()), "default")
    ops.check(lib.nr_hash_encode_bwd_tuned(p(x), p(std), p(sc), L, F, log2t, p(gout), L * F, F, p(got), n, 0, 256, st()), "wide")
    assert_close(got.cpu(), want.cpu(), rtol=1e-4, atol_scale=1e-5, what=f"wide merge tables, {rows}")
    assert torch.equal(got != 0, want != 0)
    assert lib.nr_hash_encode_bwd_tuned(p(x), p(std), p(sc), L, F, log2t, p(gout), L * F, F, p(got), n, 0, 192, st()) != 0  # unknown size
