"""GPU parity of the block-shared scatter-add (nr_hash_encode_bwd_shared, grid_shared.hip: vertex-keyed LDS table on 32-bit
integer atomics; the main grid's scatter of steps on 16-bit MLP operands) -- against the oracle (autograd of the reference's
gather, oracle/hashgrid.py) on small cases incl. degenerate ones, against the merging kernel at NeuRadar's table size, its
stated rounding bound, the write set, the `seen` bytes it marks, and its argument checks."""
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _addends(x, sc, log2t, live):
    """Number of addends the scatter sums into every table entry (8 corners of every row with a non-zero gradient, per level;
    the reference's ceil / floor corners hashed as encodings.py:406-421) -> [L << log2t] int64, computed with torch on the device."""
    T = 1 << log2t
    out = torch.zeros(sc.numel() * T, device=x.device, dtype=torch.int64)
    for l in range(sc.numel()):
        p = x * sc[l]
        lo, hi = torch.floor(p).to(torch.int64), torch.ceil(p).to(torch.int64)
        for c in range(8):
            ix = (hi if c & 1 else lo)[:, 0]
            iy = (hi if c & 2 else lo)[:, 1]
            iz = (hi if c & 4 else lo)[:, 2]
            h = ((ix * 1) ^ (iy * 2654435761) ^ (iz * 805459861)) % T
            out.index_add_(0, h + l * T, live[:, l].to(torch.int64))
    return out


def _shared(x, std, sc, log2t, gout_rows, seen=None, threads=None):
    """gout_rows [n, L*4] -> grad_table [L << log2t, 4] through nr_hash_encode_bwd_shared (level-major gradient rows); threads:
    nr_hash_encode_bwd_shared_split with that many threads per row."""
    from neuradar_amd import ops

    n, L = x.shape[0], sc.numel()
    g = gout_rows.view(n, L, 4).permute(1, 0, 2).contiguous()  # [L, n, 4]
    gt = torch.zeros(L << log2t, 4, device=DEV)
    lib, p = ops._lib.lib(), ops._p
    if threads is None:
        ops.check(lib.nr_hash_encode_bwd_shared(p(x), p(std), p(sc), L, 4, log2t, p(g), 4, n * 4, p(gt), n, p(seen), ops._stream()), "shared")
    else:
        ops.check(lib.nr_hash_encode_bwd_shared_split(p(x), p(std), p(sc), L, 4, log2t, p(g), 4, n * 4, p(gt), n, p(seen), threads, ops._stream()),
                  "shared_split")
    return gt


@pytest.mark.parametrize("threads", [None, 1, 2, 4])
@pytest.mark.parametrize("case", ["uniform", "one_cell", "ragged_tiny_table", "grid_planes", "runs"])
def test_shared_scatter_vs_oracle(case, threads):
    """Against torch.autograd of the oracle's gather; threads: 1 / 2 / 4 threads sharing a row's 8 corners (ABI v28: blocks of 256 /
    512 / 1 024 threads on the same tile and table), None = the v27 entry point.  uniform: distinct cells everywhere (a full table at the fine level);
    one_cell: all rows in ONE cell (256 lanes on 8 LDS addresses); ragged_tiny_table: n not a multiple of 256, a table of
    512 entries (many vertices share a slot chain); grid_planes: positions on exact cell boundaries (ceil == floor: two
    "corners" are one vertex); runs: runs of rows in one cell.  Tolerance: the kernel's own resolution -- an addend is rounded
    to 2^-21 of the tile's largest gradient entry -- on top of the usual rtol 1e-4."""
    from oracle import hashgrid

    torch.manual_seed(len(case))
    L, log2t, n = (3, 17, 30_000) if case != "ragged_tiny_table" else (2, 9, 1025 + 37)
    sc = hashgrid.level_scalings(L, 16, 2048)
    x = torch.rand(n, 3)
    if case == "one_cell":
        x = (torch.tensor([0.3137, 0.7211, 0.5003]) + 1e-6 * torch.rand(n, 3)).clamp(0, 1)
    if case == "grid_planes":
        x = torch.randint(0, 17, (n, 3)).float() / 16.0
        x[::2, 1] = torch.rand(n // 2)
    if case == "runs":
        x = (torch.rand(n // 24 + 1, 3).repeat_interleave(24, dim=0)[:n] + 2e-5 * torch.rand(n, 3)).clamp(0, 1)
    x[:5] = torch.tensor([0.0, 0.25, 0.5])
    std = 0.01 * torch.rand(n)
    gout = torch.randn(n, L * 4)
    gout[::7] = 0.0  # zero gradients are skipped
    table = hashgrid.init_table(L, log2t, 4, scale=1.0).requires_grad_(True)
    resc = 1.0 / torch.clamp(2.0 * sc[None, :] * std[:, None], min=1.0)  # neurad_encoding.py:309-316
    ref = hashgrid.encode(x, table, sc, 2**log2t).view(n, L, 4) * resc[:, :, None]
    (gref,) = torch.autograd.grad(ref, table, gout.view(n, L, 4))
    gt = _shared(x.to(DEV), std.to(DEV), sc.to(DEV), log2t, gout.to(DEV), threads=threads).cpu()
    # per entry: every addend is rounded to at most 2^-21 of the largest gradient entry (half a unit of 2^(e-21), 2^e the power
    # of two above its tile's largest entry), and the fp32 sums of either side carry ~1e-7 of the addends' magnitude
    n_add = _addends(x.to(DEV), sc.to(DEV), log2t, (gout.view(n, L, 4) != 0).any(dim=2).to(DEV)).cpu().double()[:, None]
    bound = 1e-4 * gref.abs() + (n_add + 1.0) * (2.0 ** -21 + 2e-7) * float(gout.abs().max())
    err = (gt - gref).abs()
    assert bool((err <= bound).all()), f"{case}: worst excess {float((err - bound).max()):.3e} at scale {float(gref.abs().max()):.3e}"
    assert float((gt - gref).norm() / gref.norm()) < 5e-5
    assert not bool(((gt != 0) & (gref == 0)).any()), "an entry was written that no row touches"


@pytest.mark.parametrize("threads", [1, 2, 4])
@pytest.mark.parametrize("rows", ["incoherent", "runs", "patch"])
def test_shared_scatter_equals_merging_kernel_at_neuradar_size(rows, threads):
    """NeuRadar's main grid (8 levels x 2^22 entries x 4 floats), gradients with magnitudes over two decades: the two scatter
    implementations agree within the shared kernel's stated bound and to 1e-5 in relative L2; the shared kernel writes no
    entry the merging kernel leaves untouched."""
    from neuradar_amd import ops
    from oracle import hashgrid

    torch.manual_seed(4)
    L, log2t, n = 8, 22, 4661 * 32 + 7
    sc = hashgrid.level_scalings(L, 32, 8192).to(DEV)
    x = torch.rand(n, 3, device=DEV)
    if rows == "runs":
        x = (torch.rand(n // 24 + 1, 3, device=DEV).repeat_interleave(24, dim=0)[:n] + 2e-5 * torch.rand(n, 3, device=DEV)).clamp(0, 1)
    if rows == "patch":  # 256 neighbouring rays at one depth: a patch of the fine level's cell size per ray
        base = torch.rand(n // 256 + 1, 3, device=DEV).repeat_interleave(256, dim=0)[:n]
        ij = torch.arange(n, device=DEV) % 256
        x = (base * 0.9 + torch.stack([(ij % 16).float(), (ij // 16).float(), torch.zeros(n, device=DEV)], 1) * 1.2e-4).clamp(0, 1)
    std = 0.002 * torch.rand(n, device=DEV)
    gout = torch.randn(n, L * 4, device=DEV) * torch.exp(2.0 * torch.randn(n, 1, device=DEV))
    lib, p, st = ops._lib.lib(), ops._p, ops._stream
    want = torch.zeros(L << log2t, 4, device=DEV)
    ops.check(lib.nr_hash_encode_bwd(p(x), p(std), p(sc), L, 4, log2t, p(gout), L * 4, 4, p(want), n, 0, st()), "merging")
    n_add = _addends(x, sc, log2t, torch.ones(n, L, dtype=torch.bool, device=DEV)).double()[:, None]
    seen = torch.zeros(L << log2t, device=DEV, dtype=torch.uint8)
    got = _shared(x, std, sc, log2t, gout, seen=seen, threads=threads)
    err = (got - want).abs()
    bound = 1e-4 * want.abs() + (n_add + 1.0) * (2.0 ** -21 + 2e-7) * float(gout.abs().max())
    assert bool((err <= bound).all()), f"{rows}: worst excess {float((err - bound).max()):.3e}"
    # relative L2: the gradients' magnitudes span two decades (exp(2 N(0,1))) INSIDE a tile here, so the quantum -- 2^-21 of the
    # tile's largest entry -- is 2^-14 ... 2^-21 of a typical addend; measured 2.3e-5 (1e-6 on the step's own gradients,
    # tools/probe_main_shared.py)
    rel = float((got - want).norm() / want.norm())
    assert rel < 1e-4, rel
    assert not bool(((got != 0) & (want == 0)).any())
    # the optimizer's bytes (one per 4-float entry): set wherever a non-zero sum was added -- every entry that ended up non-zero
    # is marked, and nothing is marked that no row touches (a marked entry may be zero: quantised sums of +q and -q from two tiles
    # cancel exactly)
    written = (got != 0).any(dim=1)
    assert bool((seen[written] != 0).all()), "an entry was written without its seen byte"
    assert not bool(((seen != 0) & (n_add[:, 0] == 0)).any()), "a seen byte was set for an entry no row touches"
    assert int((seen != 0).sum()) <= int(written.sum()) * 1.01 + 16


def test_shared_scatter_is_linear_and_order_independent():
    """Integer sums: the result does not depend on the order of the rows inside a tile, and scaling every gradient by a power of
    two scales the result exactly."""
    from oracle import hashgrid

    torch.manual_seed(9)
    L, log2t, n = 4, 16, 256 * 40
    sc = hashgrid.level_scalings(L, 16, 512).to(DEV)
    x = torch.rand(n, 3, device=DEV)
    std = 0.01 * torch.rand(n, device=DEV)
    gout = torch.randn(n, L * 4, device=DEV)
    a = _shared(x, std, sc, log2t, gout)
    perm = (torch.arange(n, device=DEV).view(-1, 256)[:, torch.randperm(256, device=DEV)]).reshape(-1)  # shuffles inside each tile
    b = _shared(x[perm].contiguous(), std[perm].contiguous(), sc, log2t, gout[perm].contiguous())
    # (the float atomics of different tiles onto one entry still arrive in any order: equal to rounding of those few adds)
    assert_close(b.cpu(), a.cpu(), rtol=1e-5, atol_scale=1e-6, what="rows shuffled inside their tiles")
    c = _shared(x, std, sc, log2t, gout * 8.0)
    assert_close(c.cpu(), (8.0 * a).cpu(), rtol=1e-5, atol_scale=1e-6, what="gradients x 8")


def test_shared_scatter_argument_checks_empty_input_and_nonfinite_rows():
    from neuradar_amd import ops

    lib, p, st = ops._lib.lib(), ops._p, ops._stream
    L, log2t, n = 2, 12, 300
    x, sd, sc = torch.rand(n, 3, device=DEV), 0.01 * torch.rand(n, device=DEV), torch.tensor([16.0, 64.0], device=DEV)
    g = torch.randn(L, n, 4, device=DEV)
    gt = torch.zeros(L << log2t, 4, device=DEV)
    ok = lambda **kw: lib.nr_hash_encode_bwd_shared(p(x), p(sd), p(sc), kw.get("L", L), kw.get("F", 4), log2t, p(kw.get("g", g)),  # noqa: E731
                                                    kw.get("sn", 4), n * 4, p(gt), kw.get("n", n), None, st())
    assert ok(n=0) == 0 and float(gt.abs().sum()) == 0.0
    assert ok() == 0 and float(gt.abs().sum()) > 0.0
    assert ok(F=2) != 0 and ok(sn=8) != 0 and ok(L=9) != 0
    assert lib.nr_hash_encode_bwd_shared(p(x), p(sd), p(sc), L, 4, log2t, None, 4, n * 4, p(gt), n, None, st()) != 0
    # a row with inf / NaN gradients goes to the table directly (torch's index_put would write them too); the finite rows of the
    # same tile are not disturbed
    g2 = g.clone()
    g2[0, 17, 2] = float("inf")
    g2[1, 200, 0] = float("nan")
    gt2 = torch.zeros_like(gt)
    assert lib.nr_hash_encode_bwd_shared(p(x), p(sd), p(sc), L, 4, log2t, p(g2), 4, n * 4, p(gt2), n, None, st()) == 0
    bad = ~torch.isfinite(gt2)
    assert 0 < int(bad.sum()) <= 16
    g3 = g.clone()
    g3[0, 17], g3[1, 200] = 0.0, 0.0
    gt3 = torch.zeros_like(gt)
    assert lib.nr_hash_encode_bwd_shared(p(x), p(sd), p(sc), L, 4, log2t, p(g3), 4, n * 4, p(gt3), n, None, st()) == 0
    rows_bad = bad.any(dim=1)
    assert_close(gt2[~rows_bad].cpu(), gt3[~rows_bad].cpu(), rtol=1e-5, atol_scale=1e-6, what="finite rows beside non-finite ones")
    # threads per row (ABI v28): 0 = the library's default, 1 / 2 / 4; anything else is refused; the non-finite rows take the same
    # way whatever the shape
    split = lambda t, g_, out: lib.nr_hash_encode_bwd_shared_split(p(x), p(sd), p(sc), L, 4, log2t, p(g_), 4, n * 4, p(out), n, None, t, st())  # noqa: E731
    assert split(3, g, gt) != 0 and split(8, g, gt) != 0 and split(-1, g, gt) != 0
    for t in (0, 2, 4):
        gt4 = torch.zeros_like(gt)
        assert split(t, g2, gt4) == 0
        assert bool((~torch.isfinite(gt4) == bad).all())
        assert_close(gt4[~rows_bad].cpu(), gt3[~rows_bad].cpu(), rtol=1e-5, atol_scale=1e-6, what=f"{t} threads per row: finite rows")
