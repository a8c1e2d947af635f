"""The sharded data-parallel step's gradient half as row lists to the shard owners: the three kernels behind
GradAllReducer._lists_to_owners (nr_grad_compact_shards, nr_grad_lists_apply, nr_grad_lists_restore) against their torch
restatements (the stand-ins the 2-rank gloo tests of tests/test_distributed_gloo.py run the same exchange on), bit for bit --
the lists are integer indices and copied floats, the adds are plain fp32 adds of unique rows.
Reference semantics: the reduction half of DDP's gradient all-reduce, pipelines/base_pipeline.py:305-307."""
import pytest
import torch

from tests.test_distributed_gloo import _torch_compact_shards, _torch_lists_apply, _torch_lists_restore

pytestmark = pytest.mark.gpu


def _gradient(rows, F, density, seed, dev):
    g = torch.Generator().manual_seed(seed)
    hit = torch.rand(rows, generator=g) < density
    v = torch.where(hit[:, None], torch.randn(rows, F, generator=g), torch.zeros(rows, F))
    return v.reshape(-1).to(dev)


def _segments(idx, val, caps, counts):
    """Per destination: the list's rows sorted by index (the kernel's order depends on which block reserved its span first)."""
    out, off = [], 0
    for d, c in enumerate(caps):
        n = min(int(counts[d]), int(c))
        order = torch.argsort(idx[off:off + n])
        out.append((idx[off:off + n][order].cpu(), val[off:off + n][order].cpu()))
        off += int(c)
    return out


@pytest.mark.parametrize("F", [1, 2, 4, 8])
@pytest.mark.parametrize("world,caps_frac", [(1, [0.5]), (2, [0.3, 0.02]), (4, [0.2, 0.2, 0.0, 0.05]), (8, [0.15] * 8)])
def test_compact_shards_vs_torch(F, world, caps_frac):
    from neuradar_amd import ops

    dev = torch.device("cuda", 0)
    rows = 3000 * world + 0  # per shard 3000 rows (not a multiple of the wave / block sizes)
    per = rows // world
    g = _gradient(rows, F, 0.1, 5 + F + world, dev)
    g_ref = g.clone()
    caps = torch.tensor([int(f * per) for f in caps_frac], dtype=torch.int32)
    total = int(caps.sum())
    idx, val = torch.full((max(total, 1),), -1, dtype=torch.int32, device=dev), torch.zeros(max(total, 1), F, device=dev)
    counts = torch.zeros(world, dtype=torch.int32, device=dev)
    ops.grad_compact_shards(g, F, world, caps.to(dev), idx, val, counts)
    idx_r, val_r, counts_r = torch.full_like(idx, -1), torch.zeros_like(val), torch.zeros_like(counts)
    _torch_compact_shards(g_ref, F, world, caps, idx_r, val_r, counts_r)
    assert torch.equal(counts.cpu(), counts_r.cpu())
    overflow = [int(counts[d]) > int(caps[d]) for d in range(world)]
    assert any(overflow) or world == 1 or F >= 1  # (the parametrisation holds overflowing and fitting segments)
    # which rows an overflowing segment takes is the kernel's choice (span order): compare what is order-independent --
    # lists + remainder together are the original gradient, every listed row was non-zero and is cleared, capacities respected
    full = g.clone().view(world, per, F)
    off = 0
    for d in range(world):
        n = min(int(counts[d]), int(caps[d]))
        rows_d = idx[off:off + n].long()
        assert rows_d.numel() == rows_d.unique().numel() and (n == 0 or (int(rows_d.min()) >= 0 and int(rows_d.max()) < per))
        assert float(full[d][rows_d].abs().max()) == 0.0 if n else True
        full[d].index_add_(0, rows_d, val[off:off + n])
        if not overflow[d]:
            (ia, va), (ib, vb) = _segments(idx, val, caps, counts)[d], _segments(idx_r, val_r, caps, counts_r)[d]
            assert torch.equal(ia, ib) and torch.equal(va, vb)
            assert float(g.view(world, per, F)[d].abs().max()) == 0.0, "a segment that fits leaves its shard cleared"
        assert bool((idx[off + n:off + int(caps[d])] == -1).all()), "nothing written beyond the list's end"
        off += int(caps[d])
    assert torch.equal(full.reshape(-1), _gradient(rows, F, 0.1, 5 + F + world, dev))


@pytest.mark.parametrize("overflow,found", [(False, False), (True, False), (True, True), (False, True)])
def test_lists_apply_and_restore_vs_torch(overflow, found):
    """One rank's view of a 4-rank exchange: its send lists (compacted here), four received lists (made up: each source's rows of
    this rank's shard), the gathered counts.  Without overflow the owner's shard becomes the rank-order sum and restore is a
    no-op; with one overflowing (source, destination) pair NOTHING is applied, the flag says 2 and restore makes the local
    gradient whole again -- unless the loss scaler's found-inf flag is raised as well: then the whole local gradient (here with a
    few rows the lists had no room for left in it) is cleared instead."""
    from neuradar_amd import ops

    dev = torch.device("cuda", 0)
    world, F, per, own = 4, 4, 5000, 2
    caps = torch.tensor([700, 650, 800, 600], dtype=torch.int32)
    g = _gradient(world * per, F, 0.1, 77, dev)
    g0 = g.clone()
    total = int(caps.sum())
    idx_s, val_s = torch.zeros(total, dtype=torch.int32, device=dev), torch.zeros(total, F, device=dev)
    counts = torch.zeros(world, dtype=torch.int32, device=dev)
    ops.grad_compact_shards(g, F, world, caps.to(dev), idx_s, val_s, counts)
    cm = torch.zeros(world, world, dtype=torch.int32)
    mine = int(caps[own])
    idx_r, val_r = torch.zeros(world * mine, dtype=torch.int32, device=dev), torch.zeros(world * mine, F, device=dev)
    gen = torch.Generator().manual_seed(3)
    for src in range(world):
        n = 400 + 50 * src if src != own else int(counts[own])  # (the own list's length is what the compaction counted)
        rows_ = torch.randperm(per, generator=gen)[:n].to(torch.int32)
        idx_r[src * mine:src * mine + n] = rows_.to(dev)
        val_r[src * mine:src * mine + n] = torch.randn(n, F, generator=gen).to(dev)
        cm[src] = torch.tensor([500, 480, n, 450], dtype=torch.int32)
    cm[own] = counts.cpu()
    if overflow:
        cm[1, 3] = int(caps[3]) + 1  # some OTHER pair's list did not fit: every rank must take the same branch
    cm_dev = cm.reshape(-1).to(dev)
    flag = torch.full((1,), -1.0, device=dev)
    ref_g, ref_flag = g.clone().cpu(), torch.full((1,), -1.0)
    shard = g[own * per * F:(own + 1) * per * F]
    for src in range(world):
        ops.grad_lists_apply(idx_r[src * mine:(src + 1) * mine], val_r[src * mine:(src + 1) * mine], cm_dev, caps.to(dev), src, own, F, shard, flag)
        _torch_lists_apply(idx_r[src * mine:(src + 1) * mine].cpu(), val_r[src * mine:(src + 1) * mine].cpu(), cm.reshape(-1), caps, src, own, F,
                           ref_g[own * per * F:(own + 1) * per * F], ref_flag)
    found_inf = torch.full((1,), 1.0 if found else 0.0, device=dev)
    if overflow and found:
        g[5 * F] = float("nan")  # (a row the compaction had no room for)
        ref_g[5 * F] = float("nan")
    ops.grad_lists_restore(idx_s, val_s, int(caps.max()), cm_dev, caps.to(dev), own, F, g, found_inf)
    _torch_lists_restore(idx_s.cpu(), val_s.cpu(), int(caps.max()), cm.reshape(-1), caps, own, F, ref_g, found_inf.cpu())
    assert float(flag) == float(ref_flag) == (2.0 if overflow else 0.0)
    assert torch.equal(g.cpu(), ref_g)
    if overflow and found:
        assert float(g.abs().max()) == 0.0, "a rejected step's gradient is discarded everywhere"
    elif overflow:
        assert torch.equal(g, g0), "after an overflowed exchange the local gradient is whole again"
    else:
        assert float(g.view(world, -1)[torch.arange(world) != own].abs().max()) == 0.0
