"""CPU, world_size 2 over gloo: the data-parallel plumbing of the path (rank bootstrap, parameter
broadcast, bucketed gradient mean all-reduce incl. the unused-parameter case) -- the N>1 path that
runs over RCCL on the GPU box."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, bf16):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from neuradar_amd.parallel import GradAllReducer, broadcast_parameters, init_distributed

    r, w, _ = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # deliberately different init per rank
    table = torch.nn.Parameter(torch.randn(1 << 17, 2))       # "hash table": its own bucket
    unused = torch.nn.Parameter(torch.randn(1 << 17, 1))      # proposal_fields[0]: never gets a gradient
    small = [torch.nn.Parameter(torch.randn(33, 32)), torch.nn.Parameter(torch.randn(33))]
    mod = torch.nn.ParameterList([table, unused, *small])
    broadcast_parameters(mod)
    ref = [p.detach().clone() for p in mod]
    gathered = [torch.empty_like(ref[0]) for _ in range(world)]
    dist.all_gather(gathered, ref[0])
    assert all(torch.equal(g, gathered[0]) for g in gathered), "broadcast did not synchronise the replicas"

    table.grad = torch.full_like(table, float(rank + 1))
    unused.grad = None  # never evaluated -> no gradient at all; the reducer must still issue the collective (as zeros)
    small[0].grad = torch.full_like(small[0], 10.0 * (rank + 1))
    small[1].grad = None  # e.g. a bias that did not take part
    red = GradAllReducer(list(mod), table_dtype=torch.bfloat16 if bf16 else None)
    assert red.bytes_per_step() == (table.numel() + unused.numel()) * (2 if bf16 else 4) + (33 * 32 + 33) * 4
    red.all_reduce()
    mean = sum(range(1, world + 1)) / world
    assert torch.allclose(table.grad, torch.full_like(table, mean), rtol=1e-2 if bf16 else 1e-6)
    assert float(unused.grad.abs().max()) == 0.0
    assert torch.allclose(small[0].grad, torch.full_like(small[0], 10.0 * mean))
    assert small[1].grad is not None and float(small[1].grad.abs().max()) == 0.0
    # fine-grained interface of the fused step: issue early (start), wait late (wait_all)
    bufs = [torch.full((1 << 17,), float(rank + 1)), torch.full((100,), 2.0 * (rank + 1))]
    red2 = GradAllReducer(None, buffers=bufs, table_dtype=torch.bfloat16 if bf16 else None)
    for b in bufs:
        red2.start(b)
    red2.wait_all()
    total = float(sum(range(1, world + 1)))
    assert torch.allclose(bufs[0], torch.full_like(bufs[0], total), rtol=1e-2 if bf16 else 1e-6)
    assert torch.allclose(bufs[1], torch.full_like(bufs[1], 2.0 * total))
    dist.barrier()
    dist.destroy_process_group()


def _torch_compact(grad, row_width, idx, val, count):
    """Test-side CPU stand-in for nr_grad_compact (the product default is the HIP kernel)."""
    g = grad.view(-1, row_width)
    rows = (g != 0).any(dim=1).nonzero()[:, 0]
    k = min(rows.numel(), idx.numel())
    idx[:k] = rows[:k].to(torch.int32)
    val[:k * row_width] = g[rows[:k]].reshape(-1)
    g[rows[:k]] = 0
    count += rows.numel()


def _torch_apply_guarded(idx, val, counts, list_rank, own_rank, m, row_width, grad, flag):
    """... and for nr_grad_apply_guarded."""
    ovf = bool((counts > m).any())
    if flag is not None and list_rank == 0:
        flag.fill_(2.0 if ovf else 0.0)
    if ovf and list_rank != own_rank:
        return
    k = min(int(counts[list_rank]), m)
    grad.view(-1, row_width).index_add_(0, idx[:k].long(), val[:k * row_width].view(k, row_width))


def _sparse_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from neuradar_amd.parallel import GradAllReducer, init_distributed

    init_distributed(backend="gloo")
    rows, F = 1 << 14, 2
    ops_ = (_torch_compact, _torch_apply_guarded)

    def grads(seed, density):
        g = torch.Generator().manual_seed(seed + rank)
        grad = torch.zeros(rows, F)
        hit = torch.randperm(rows, generator=g)[:int(rows * density)]
        grad[hit] = torch.randn(hit.numel(), F, generator=g)
        return grad

    def identical(flat):
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(t, gathered[0]) for t in gathered), "ranks disagree bitwise"

    # (a) a first call that fits: lists; (b) a first call that does not: the dense fallback (the one synchronous decision)
    for trial, density in enumerate((0.01, 0.5)):
        red = GradAllReducer(None, buffers=[])
        grad = grads(7 * trial, density)
        dense = grad.clone()
        dist.all_reduce(dense)
        flat = grad.view(-1)
        info = red.reduce_sparse(flat, F, ops=ops_)
        assert info["mode"] == ("sparse" if density < 0.05 else "dense"), info
        assert float(info["flag"]) == 0.0
        assert torch.allclose(flat.view(rows, F), dense, rtol=1e-6, atol=1e-6)
        identical(flat)
    # (c) steady state without a host read: the list length of step k comes from the counts of step k-1 (2x headroom) ...
    red = GradAllReducer(None, buffers=[])
    buf = torch.zeros(rows * F)
    for k, density in enumerate((0.01, 0.012, 0.015)):
        buf.view(rows, F).add_(grads(100 + k, density))
        want = buf.view(rows, F).clone()
        dist.all_reduce(want)
        info = red.reduce_sparse(buf, F, ops=ops_)
        assert info["mode"] == "sparse" and float(info["flag"]) == 0.0
        if k > 0:
            assert info["list_rows"] < 1000, info  # ~2 x 1 % of 16 384 rows, not the capacity
        assert torch.allclose(buf.view(rows, F), want, rtol=1e-6, atol=1e-6)
        identical(buf)
        buf.zero_()  # (the optimizer consumes the gradient)
    # (d) ... and a step whose count jumps past it: nobody applies anything foreign, every rank keeps its own gradient whole,
    # the flag says "skip and keep"; the NEXT exchange (sized from the counts known by then) carries both steps' gradients
    mine = grads(200, 0.2 if rank == 0 else 0.01)  # rank 0 touches 20 % of the rows: far beyond 2 x 1.5 %
    buf.view(rows, F).add_(mine)
    info = red.reduce_sparse(buf, F, ops=ops_)
    assert info["mode"] == "sparse" and float(info["flag"]) == 2.0, info
    assert torch.equal(buf.view(rows, F), mine), "an overflowed exchange must leave the local gradient whole"
    more = grads(300, 0.01)
    buf.view(rows, F).add_(more)  # the skipped optimizer kept the gradient; the next step's scatter adds onto it
    want = (mine + more).clone()
    dist.all_reduce(want)
    info = red.reduce_sparse(buf, F, ops=ops_)
    assert float(info["flag"]) == 0.0 and info["mode"] in ("sparse", "dense"), info
    assert torch.allclose(buf.view(rows, F), want, rtol=1e-6, atol=1e-6)
    identical(buf)
    # (e) dense is not for ever: one outlier batch sends the exchange dense; the touched rows are re-counted every
    # `dense_probe_every` dense steps and the exchange returns to lists once they fit again (with 2x room)
    red = GradAllReducer(None, buffers=[])
    red.dense_probe_every = 2
    modes = []
    for k, density in enumerate((0.5, 0.01, 0.01, 0.01, 0.01)):
        buf.zero_()
        buf.view(rows, F).add_(grads(400 + k, density))
        want = buf.view(rows, F).clone()
        dist.all_reduce(want)
        info = red.reduce_sparse(buf, F, ops=ops_)
        modes.append(info["mode"])
        assert float(info["flag"]) == 0.0
        assert torch.allclose(buf.view(rows, F), want, rtol=1e-6, atol=1e-6), (k, info["mode"])
        identical(buf)
    assert modes[0] == "dense" and modes[1] == "dense" and modes[-1] == "sparse", modes
    dist.barrier()
    dist.destroy_process_group()


class _StubOpt:
    """What GradAllReducer.shard_step needs of step.FlatAdam (whose Adam kernel needs a GPU): buffers, shards, step_buffer --
    here plain SGD on the rank's rows, which also clears their gradient like the fused Adam does; with `delta16` the update
    leaves as a bf16 delta that the owner applies itself (nr_adam_step's delta16 semantics)."""

    def __init__(self, p, g, rank, world):
        self.buffers = [(p, g)]
        per = p.numel() // world
        self.shards = {0: (rank * per, (rank + 1) * per)}

    def step_buffer(self, i, grad_scale, delta16=None):
        (p, g), (lo, hi) = self.buffers[i], self.shards[i]
        upd = -0.1 * grad_scale * g[lo:hi]
        if delta16 is not None:
            delta16.copy_(upd.to(torch.bfloat16))
            upd = delta16.float()
        p[lo:hi] += upd
        g[lo:hi] = 0


def _torch_to16_clear(g, low):
    low.copy_(g.to(low.dtype))
    g.zero_()


def _torch_apply_delta(p, delta, lo, hi):
    d = delta.float()
    d[lo:hi] = 0
    p += d


def _shard_worker(rank, world, port, bf16, delta):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from neuradar_amd.parallel import GradAllReducer, init_distributed

    init_distributed(backend="gloo")
    n = 1 << 14
    p = torch.linspace(-1, 1, n).clone()  # identical replicas
    p0 = p.clone()
    red = GradAllReducer(None, buffers=[], table_mode="shard")
    opt = None
    for stepno in range(3):
        g = torch.Generator().manual_seed(10 * stepno + rank)
        grad = torch.where(torch.rand(n, generator=g) < 0.3, torch.randn(n, generator=g), torch.zeros(n))
        total = grad.clone()
        dist.all_reduce(total)
        if opt is None:
            opt = _StubOpt(p, grad, rank, world)
        opt.buffers[0] = (p, grad)
        info = red.shard_step(opt, 0, 1.0 / world, transport=torch.bfloat16 if bf16 else None,
                              delta_dtype=torch.bfloat16 if delta else None, defer=True,  # (defer: a no-op off the GPU)
                              kernels=(_torch_to16_clear, _torch_apply_delta))
        assert info["mode"] == "shard" and info["bytes"] > 0
        f = (world - 1) / world
        assert info["reduce_scatter_bytes_per_gpu"] == int(f * n * (2 if bf16 else 4))
        assert info["all_gather_bytes_per_gpu"] == int(f * n * (2 if delta else 4))
        p0 -= 0.1 / world * total  # the replicated step on the summed gradient
        assert float(grad.abs().max()) == 0.0, "the local gradient must be cleared everywhere"
        gathered = [torch.empty_like(p) for _ in range(world)]
        dist.all_gather(gathered, p)
        assert all(torch.equal(t, gathered[0]) for t in gathered), "replicas differ after the sharded step"
        # bf16 transport rounds the summed gradient (2^-9 relative per hop), the bf16 delta rounds the UPDATE (2^-9 of 0.1 * g):
        # against the exact replicated step the parameters stay within 2^-8 of the largest update so far
        tol = 1e-6 if not (bf16 or delta) else 2.0 ** -8 * 0.1 * 4.0 * (stepno + 1)
        assert torch.allclose(p, p0, rtol=0, atol=tol), float((p - p0).abs().max())
    red.flush()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_table_step_fp32():
    mp.spawn(_shard_worker, args=(2, _free_port(), False, False), nprocs=2, join=True)


def test_two_rank_sharded_table_step_bf16_transport():
    mp.spawn(_shard_worker, args=(2, _free_port(), True, False), nprocs=2, join=True)


def test_two_rank_sharded_table_step_bf16_both_halves():
    """bf16 reduce-scatter AND bf16 update-delta all-gather: owner and receivers apply the same rounded delta, so the replicas
    stay bit-identical (checked inside the worker) although neither half of the exchange carries fp32."""
    mp.spawn(_shard_worker, args=(2, _free_port(), True, True), nprocs=2, join=True)


def test_two_rank_sharded_table_step_fp32_gradient_bf16_delta():
    mp.spawn(_shard_worker, args=(2, _free_port(), False, True), nprocs=2, join=True)


# ---- the sharded step's gradient half as row lists to the shard owners (GradAllReducer._lists_to_owners) --------------------
def _torch_compact_shards(g, F, world, caps, idx, val, counts):
    """nr_grad_compact_shards in torch: the non-zero rows of shard d -> segment d of the lists (capacity caps[d]), cleared in g."""
    G = g.view(world, -1, F)
    off = 0
    for d in range(world):
        nz = (G[d] != 0).any(dim=1).nonzero().flatten()
        counts[d] += nz.numel()
        take = nz[:int(caps[d])]
        if take.numel():
            idx[off:off + take.numel()] = take.to(torch.int32)
            val[off:off + take.numel()] = G[d][take]
            G[d][take] = 0
        off += int(caps[d])


def _lists_overflowed(cm, caps):
    w = caps.numel()
    return bool((cm.view(w, w) > caps.view(1, w)).any())


def _torch_lists_apply(idx, val, cm, caps, src, own, F, shard, flag):
    ovf = _lists_overflowed(cm, caps)
    if src == 0:
        flag[0] = 2.0 if ovf else 0.0
    if ovf:
        return
    n = int(cm[src * caps.numel() + own])
    shard.view(-1, F).index_add_(0, idx[:n].long(), val[:n])  # (unique rows within one list)


def _torch_lists_restore(idx, val, max_cap, cm, caps, own, F, g, found_inf=None):
    if not _lists_overflowed(cm, caps):
        return
    if found_inf is not None and float(found_inf[0]) != 0.0:  # the loss scaler rejects the step: discard, do not keep
        g.zero_()
        return
    w = caps.numel()
    G = g.view(w, -1, F)
    off = 0
    for d in range(w):
        n = min(int(cm[own * w + d]), int(caps[d]))
        G[d].index_add_(0, idx[off:off + n].long(), val[off:off + n])
        off += int(caps[d])


class _StubOptSkip(_StubOpt):
    """_StubOpt + nr_adam_step's skip flag: 2 = no update, the gradient is KEPT (zero deltas for the replicas)."""

    def step_buffer(self, i, grad_scale, delta16=None, skip_extra=None):
        if skip_extra is not None and float(skip_extra[0]) == 2.0:
            if delta16 is not None:
                delta16.zero_()
            return
        super().step_buffer(i, grad_scale, delta16=delta16)


def _lists_worker(rank, world, port, delta, densities):
    """Steps with the given fraction of non-zero rows per rank.  Checked every step: replicas bit-identical; the parameters equal the
    replicated step on the all-reduced gradient (world 2: a + b is the same float in either order, so EXACTLY); the exchange mode
    that ran; a step whose lists overflow moves nothing and its gradient arrives with the next step's."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from neuradar_amd.parallel import GradAllReducer, init_distributed

    init_distributed(backend="gloo")
    F, rows = 4, 1 << 12
    n = rows * F
    p = torch.linspace(-1, 1, n).clone()
    p0 = p.clone()
    red = GradAllReducer(None, buffers=[], table_mode="shard")
    red.list_granularity = 16
    red.dense_probe_every = 2
    grad = torch.zeros(n)
    opt = _StubOptSkip(p, grad, rank, world)
    modes = []
    for stepno, dens in enumerate(densities):
        g = torch.Generator().manual_seed(10 * stepno + rank)
        hit = torch.rand(rows, generator=g) < dens
        fresh = torch.where(hit[:, None], torch.randn(rows, F, generator=g), torch.zeros(rows, F)).reshape(-1)
        grad += fresh  # (the scatter ADDS onto whatever an overflowed step left behind)
        # what the exchange must deliver if it delivers: the sum of the ranks' local gradients IN RANK ORDER (the owner adds the lists
        # 0, 1, ..., world - 1 with plain adds; for two ranks any order gives the same float)
        parts = [torch.empty_like(grad) for _ in range(world)]
        dist.all_gather(parts, grad.clone())
        total = torch.zeros_like(grad)
        for part in parts:
            total += part
        info = red.shard_step(opt, 0, 1.0 / world, delta_dtype=torch.bfloat16 if delta else None, defer=True, row_width=F,
                              kernels=(_torch_to16_clear, _torch_apply_delta),
                              list_kernels=(_torch_compact_shards, _torch_lists_apply, _torch_lists_restore))
        st = next(iter(red._lists_state.values()))
        overflowed = info["gradient_half"].startswith("row lists") and float(st["flag"][0]) == 2.0
        modes.append("overflow" if overflowed else ("lists" if info["gradient_half"].startswith("row lists") else "dense"))
        if overflowed:
            assert float(grad.abs().max()) > 0, "an overflowed step must keep the local gradient"
            kept = grad.clone()
            dist.all_reduce(kept)
            assert torch.equal(kept, total), "an overflowed step must leave every rank's gradient whole"
        else:
            upd = -0.1 / world * total
            p0 += upd.to(torch.bfloat16).float() if delta else upd
            assert float(grad.abs().max()) == 0.0, "the local gradient must be cleared everywhere"
            if modes[-1] == "lists":
                f = (world - 1) / world
                assert info["reduce_scatter_bytes_per_gpu"] == sum(c for d, c in enumerate(info["list_rows_per_destination"]) if d != rank) * (4 + 4 * F)
                assert info["reduce_scatter_bytes_per_gpu"] < info["dense_reduce_scatter_would_be"] == int(f * n * 4)
        gathered = [torch.empty_like(p) for _ in range(world)]
        dist.all_gather(gathered, p)
        assert all(torch.equal(t, gathered[0]) for t in gathered), f"replicas differ after step {stepno} ({modes[-1]})"
        assert torch.equal(p, p0), (stepno, modes[-1], float((p - p0).abs().max()))
    red.flush()
    dist.barrier()
    dist.destroy_process_group()
    return modes


def _lists_entry(rank, world, port, delta, densities, check):
    modes = _lists_worker(rank, world, port, delta, densities)
    if check == "all_lists":
        assert modes == ["lists"] * len(densities), modes
    else:  # the overflow story (see the test's docstring); a step is sized from the counts of TWO steps earlier (count_lag)
        assert modes[:5] == ["lists", "lists", "overflow", "overflow", "lists"], modes
        assert modes[5] == "overflow" and "dense" in modes[6:] and modes[-1] == "lists", modes


def test_two_rank_sharded_step_gradient_as_row_lists_to_the_owners():
    """15 % of the rows per rank, steady: every step goes as lists, results EXACTLY the replicated step's."""
    mp.spawn(_lists_entry, args=(2, _free_port(), False, [0.15] * 4, "all_lists"), nprocs=2, join=True)


def test_four_rank_row_lists_exact_in_rank_order():
    """Four ranks: per-destination segment sizes and offsets that differ (shard d of rank s holds what it holds), the owner's
    sum in rank order -- bit for bit the sum 0 + g0 + g1 + g2 + g3."""
    mp.spawn(_lists_entry, args=(4, _free_port(), False, [0.12] * 4, "all_lists"), nprocs=4, join=True)


def test_two_rank_row_lists_with_bf16_update_deltas():
    mp.spawn(_lists_entry, args=(2, _free_port(), True, [0.15] * 3, "all_lists"), nprocs=2, join=True)


def test_two_rank_row_lists_overflow_keeps_the_gradient_then_dense_then_back():
    """0.1 -> 0.14 of the rows: the lists sized from the 0.1 steps (x 1.3) overflow -- nothing moves, every rank's gradient stays
    whole, the following steps' scatters add onto it.  A step is sized from the counts of TWO steps earlier (count_lag: the host
    never waits for the previous step's copy), so the step after the overflow overflows too and the third one's lists (sized
    3.6 x the overflowed step's true counts) carry the union of three batches -- every step's result is still EXACTLY the
    replicated step's on whatever the gradients held.  -> 0.7: overflow again, then -- beyond half a shard's rows -- the dense
    reduce-scatter; back to lists once a re-count (every 2 dense steps here) sees the 0.05 steps."""
    dens = [0.1, 0.1, 0.14, 0.14, 0.14, 0.7, 0.7, 0.05, 0.05, 0.05, 0.05, 0.05]
    mp.spawn(_lists_entry, args=(2, _free_port(), False, dens, "overflow_story"), nprocs=2, join=True)


class _StubAmp:
    def __init__(self):
        self.flag = torch.zeros(1)

    def found(self, group):
        return self.flag


def _lists_found_inf_worker(rank, world, port):
    """A step the loss scaler rejects AND whose lists overflow (an inf loss makes every touched vertex non-zero: the row count jumps
    on exactly such a step): the optimizer skips and clears -- and so must every rank's whole local gradient, although the exchange
    alone would have kept it (seen on the GPU: the kept NaN rows poisoned the next step too and the replicas diverged)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from neuradar_amd.parallel import GradAllReducer, init_distributed

    init_distributed(backend="gloo")
    F, rows = 4, 1 << 12
    n = rows * F
    p = torch.linspace(-1, 1, n).clone()
    red = GradAllReducer(None, buffers=[], table_mode="shard")
    red.list_granularity = 16
    grad = torch.zeros(n)

    class Opt(_StubOptSkip):
        def step_buffer(self, i, grad_scale, delta16=None, skip_extra=None):
            (p_, g_), (lo, hi) = self.buffers[i], self.shards[i]
            if float(self.amp.flag[0]) != 0.0:  # found-inf takes precedence (FlatAdam.step_buffer): skip and CLEAR the shard's gradient
                g_[lo:hi] = 0
                return
            super().step_buffer(i, grad_scale, delta16=delta16, skip_extra=skip_extra)

    opt = Opt(p, grad, rank, world)
    opt.amp, opt.amp_group = _StubAmp(), 0
    kern = dict(kernels=(_torch_to16_clear, _torch_apply_delta), list_kernels=(_torch_compact_shards, _torch_lists_apply, _torch_lists_restore))
    for stepno, (dens, poisoned) in enumerate([(0.1, False), (0.1, False), (0.3, True), (0.1, False)]):
        g = torch.Generator().manual_seed(10 * stepno + rank)
        hit = torch.rand(rows, generator=g) < dens
        fresh = torch.where(hit[:, None], torch.randn(rows, F, generator=g), torch.zeros(rows, F)).reshape(-1)
        if poisoned and rank == 1:
            fresh = torch.where(fresh != 0, torch.full_like(fresh, float("nan")), fresh)
        grad += fresh
        opt.amp.flag[0] = 1.0 if poisoned else 0.0  # (already summed over the ranks: raised on both)
        total = torch.nan_to_num(grad.clone())
        dist.all_reduce(total)
        before = p.clone()
        info = red.shard_step(opt, 0, 1.0 / world, defer=True, row_width=F, **kern)
        assert info["gradient_half"].startswith("row lists")
        assert float(grad.abs().nan_to_num(nan=1.0).max()) == 0.0, f"step {stepno}: the local gradient must be clear on every rank"
        if poisoned:
            assert float(next(iter(red._lists_state.values()))["flag"][0]) == 2.0, "the poisoned step was meant to overflow the lists"
            assert torch.equal(p, before)
        else:
            assert torch.equal(p, before - 0.1 / world * total)
        gathered = [torch.empty_like(p) for _ in range(world)]
        dist.all_gather(gathered, p)
        assert all(torch.equal(t, gathered[0]) for t in gathered) and bool(torch.isfinite(p).all())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_lists_overflow_on_a_step_the_loss_scaler_rejects_discards_the_gradient():
    mp.spawn(_lists_found_inf_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_two_rank_sparse_table_exchange():
    mp.spawn(_sparse_worker, args=(2, _free_port()), nprocs=2, join=True)


def _run(bf16):
    mp.spawn(_worker, args=(2, _free_port(), bf16), nprocs=2, join=True)


def test_two_rank_grad_allreduce_fp32():
    _run(False)


def test_two_rank_grad_allreduce_bf16_tables():
    _run(True)
