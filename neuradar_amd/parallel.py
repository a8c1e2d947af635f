"""Data parallelism of the path: rays shard across ranks, parameters are replicated, one gradient
all-reduce (mean) per step -- the behaviour of the reference's DDP wrap
(pipelines/base_pipeline.py:305-307, scripts/train.py:110-164) over RCCL/xGMI.

MI355X-first choices: >99 % of the gradient bytes are hash tables, so they get
their own large flat buckets (one collective per table, no DDP bucketing/copy), all small parameters
are packed into a single fused bucket, and unused parameters (proposal_fields[0]) are simply
all-reduced as zeros instead of DDP's find_unused_parameters bitmap exchange.
"""
import datetime
import os
import weakref
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn

# reducers with a deferred table all-gather possibly in flight (shard_step(defer=True)): whatever reads a table outside the
# fused step -- the rendering entry, state_dict(), optimizer checkpoints -- calls wait_all_table_syncs() first instead of relying
# on its caller to remember reducer.flush().  A stream-side wait: no host block, no-op when nothing is pending.
_DEFERRED = weakref.WeakSet()


def wait_all_table_syncs() -> None:
    for r in list(_DEFERRED):
        r.wait_table_sync()


SMALL_PARAM_NUMEL = 1 << 16


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def init_distributed(backend: Optional[str] = None, timeout_s: Optional[float] = None) -> tuple:
    """(rank, world, local_rank) from the torchrun environment; no-op for a single process.  timeout_s: the process group's
    collective timeout (None: the backend's default -- 10 minutes for RCCL, 30 for gloo)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        elif torch.cuda.is_available():
            torch.cuda.set_device(min(local_rank, torch.cuda.device_count() - 1))
        kw = {} if timeout_s is None else {"timeout": datetime.timedelta(seconds=timeout_s)}
        dist.init_process_group(backend=backend, **kw)
    return rank, world, local_rank


class _GradHolder:
    """Adapter: a bare gradient buffer seen as an object with .grad / .numel()."""

    def __init__(self, grad: torch.Tensor):
        self.grad = grad

    def numel(self) -> int:
        return self.grad.numel()


class GradAllReducer:
    """Mean all-reduce of `.grad` over the world: big tensors in place, small ones via one flat bucket."""

    def __init__(self, params: Optional[Iterable[nn.Parameter]], group=None, table_dtype: Optional[torch.dtype] = None,
                 buffers: Optional[List[torch.Tensor]] = None, sparse_tables: bool = True, separate_sparse_group: bool = False,
                 table_mode: Optional[str] = None):
        """`params`: parameters whose .grad is reduced; or `buffers`: ready-made flat gradient buffers
        (FlatAdam.grad_buffers(): one per hash table + one holding every small parameter).
        sparse_tables: the fused step exchanges the main table's gradient as (row, value) lists
        (reduce_sparse) instead of all-reducing the dense table.
        separate_sparse_group: give the list exchange a communicator of its own.  Off by default: two communicators
        with collectives in flight at the same time on different streams is the documented NCCL/RCCL deadlock hazard
        (each rank may launch them in a different order); with ONE communicator the collectives execute in issue
        order, and the fused step issues the main table's exchange first."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # how the fused step exchanges the MAIN table's gradient: "sparse" (row lists, reduce_sparse), "dense" (all-reduce),
        # "shard" (reduce-scatter -> Adam on 1/world of the rows -> all-gather, shard_step)
        # world == 1 normally skips every collective; force_collectives issues them anyway (a one-rank RCCL group executes the
        # same all_gather_into_tensor / reduce_scatter_tensor calls: how the GPU tests reach the "nccl" branches on a one-GPU box)
        self.force_collectives = False
        self.table_mode = table_mode if table_mode is not None else ("sparse" if sparse_tables else "dense")
        assert self.table_mode in ("sparse", "dense", "shard")
        sparse_tables = self.table_mode == "sparse"
        self.sparse_tables = sparse_tables
        self.last_sparse: dict = {}
        self.sparse_group = dist.new_group() if (self.world > 1 and sparse_tables and separate_sparse_group) else group
        self.table_dtype = table_dtype
        self.dense_probe_every = 50  # reduce_sparse: dense steps between two re-counts of the touched rows
        self._flat: Optional[torch.Tensor] = None
        if buffers is not None:
            self.params = [_GradHolder(b) for b in buffers]
            self.big, self.small = self.params, []
            return
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        self.big = [p for p in self.params if p.numel() > SMALL_PARAM_NUMEL]
        self.small = [p for p in self.params if p.numel() <= SMALL_PARAM_NUMEL]

    # ---- fine-grained interface used by FusedTrainStep: issue early, overlap, wait late -------------
    def start(self, grad: torch.Tensor) -> None:
        """Asynchronous SUM all-reduce of one gradient buffer, ordered after the current stream's work."""
        if self.world == 1 and not self.force_collectives:
            return
        if not hasattr(self, "_pending"):
            self._pending = []
        if self.table_dtype is not None and grad.numel() > SMALL_PARAM_NUMEL:
            low = grad.to(self.table_dtype)
            self._pending.append((dist.all_reduce(low, op=dist.ReduceOp.SUM, group=self.group, async_op=True), grad, low))
        else:
            self._pending.append((dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True), grad, None))

    def wait_all(self) -> None:
        """Make the current stream wait for every collective issued with start()."""
        for work, grad, low in getattr(self, "_pending", []):
            work.wait()
            if low is not None:
                grad.copy_(low)
        self._pending = []

    # ---- sharded table step: reduce-scatter -> Adam on the owned rows -> all-gather ------------------
    def shard_step(self, opt, i: int, grad_scale: float, transport: Optional[torch.dtype] = None,
                   delta_dtype: Optional[torch.dtype] = None, defer: bool = False, kernels=None, row_width: Optional[int] = None,
                   list_kernels=None) -> dict:
        """The data-parallel step of ONE big table whose gradient is dense across the ranks (the mixed batch touches 9-22 % of
        the main table's rows per rank: the union over 8 ranks is most of the table, so row lists no longer pay).  Instead of
        all-reduce + a replicated Adam over the whole table on every GPU:
          1. reduce-scatter (SUM) of the gradient: rank r receives the reduced rows [lo, hi) = its 1/world of the table.
             `transport` = torch.bfloat16 carries it in bf16 (half the bytes); the send buffer is written and the spent local
             gradient cleared in ONE pass (nr_grad_to16_clear) -- no 537-MB memset of the rows outside the shard;
          2. Adam on those rows only (opt.shard_buffer: moments exist for the shard only; the dense 4.3 GB stream of the
             replicated step becomes 1/world of it on every GPU);
          3. the updated rows reach every replica by all-gather, either as fp32 parameters in place (delta_dtype None) or --
             delta_dtype = torch.bfloat16 -- as the UPDATE rounded to bf16: the owner applies p_old + float(delta) itself
             (nr_adam_step's delta16) and the receivers do the same (nr_apply_delta16), so every replica computes the same sum
             from identical inputs -- bit-identical tables at half the bytes; what is lost is 2^-9 of each update, not of the
             parameter;
          4. defer=True: step 3 runs on a communication stream of its own and is only waited for where the table is read next
             (wait_table_sync: the next step's main-grid gather, ~0.4 ms into that step; flush() before anything else reads the
             parameters) -- the all-gather hides behind the next step's sampling rounds instead of ending this one.
        The same bytes on the wire as a ring all-reduce of the table in the transport types (reduce-scatter + all-gather IS that
        all-reduce), but the optimizer sits between the halves.  The owner's Adam is 1/world of the replicated one (~70 us of
        550 at 8 ranks), so nothing is gained by pipelining it against a chunked exchange: what matters is where the wire time
        of the two halves hides (the reduce-scatter behind the proposal scatters, the all-gather in the next step).
        No host read, no allocation after the first call.  opt: step.FlatAdam with shard_buffer(i) applied.
        kernels: (to16_clear, apply_delta16) callables; default = the HIP kernels (CPU tests pass torch stand-ins).

        row_width (the table's features per row) given, fp32 transport and self.shard_lists (the default): step 1 is NOT a
        dense reduce-scatter -- the gradient is sparse (9-22 % of the rows per rank on the mixed batch), so every rank sends
        each owner the (row, values) lists of that owner's shard (_lists_to_owners: one compaction launch, one all-to-all of
        the index parts and one of the value parts, the owner adds the lists in rank order with plain fp32 adds -- exact, no
        rounding of partial sums, the same sum whatever the arrival order), with the dense reduce-scatter as the fallback
        when the lists stop paying and a device-side guard for a list that overflows."""
        p, g = opt.buffers[i]
        if self.world == 1 and not self.force_collectives:
            opt.step_buffer(i, grad_scale)
            return {"mode": "single"}
        if kernels is None:
            from . import ops as hip_ops
            kernels = (hip_ops.grad_to16_clear, hip_ops.apply_delta16)
        to16_clear, apply_delta = kernels
        lo, hi = opt.shards[i]
        n, per = p.numel(), hi - lo
        nccl = dist.get_backend(self.group) == "nccl"
        key = ("shard", g.data_ptr(), transport, delta_dtype)
        st = self.__dict__.setdefault("_shard_state", {}).get(key)
        if st is None:
            st = {}
            if transport is not None:
                assert transport == torch.bfloat16, "the 16-bit transport of the reduce-scatter is bf16"
                st["low"] = torch.empty(n, device=g.device, dtype=transport)
                st["low_out"] = torch.empty(per, device=g.device, dtype=transport)
            if delta_dtype is not None:
                assert delta_dtype == torch.bfloat16, "the update delta travels in bf16"
                st["delta"] = torch.zeros(n, device=g.device, dtype=delta_dtype)
            if not nccl:  # gloo (CPU / one-GPU tests) has neither reduce_scatter nor in-place all_gather
                st["mine"] = torch.empty(per, device=g.device, dtype=delta_dtype or torch.float32)
            self._shard_state[key] = st
        # 1. reduce-scatter
        lists, keep = None, None
        if transport is None and row_width is not None and getattr(self, "shard_lists", True) and per % row_width == 0:
            amp = getattr(opt, "amp", None)  # (the loss scaler's found-inf flag of this optimizer, summed over the ranks by now)
            lists = self._lists_to_owners(g, lo, hi, int(row_width), list_kernels,
                                          found_inf=None if amp is None else amp.found(opt.amp_group))  # None: this step goes dense
            keep = None if lists is None else lists["flag"]
        if lists is not None:
            pass
        elif transport is not None:
            to16_clear(g, st["low"])  # low = bf16(g); g = 0
            if nccl:
                dist.reduce_scatter_tensor(st["low_out"], st["low"], op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.all_reduce(st["low"], op=dist.ReduceOp.SUM, group=self.group)
                st["low_out"].copy_(st["low"][lo:hi])
            g[lo:hi].copy_(st["low_out"])
        elif nccl:
            dist.reduce_scatter_tensor(g[lo:hi], g, op=dist.ReduceOp.SUM, group=self.group)  # in place: output = input's own slice
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)  # (test backends: same result on the shard)
        # 2. the owner's Adam (zeroes g[lo:hi]).  With a delta buffer it must not be overwritten while a deferred all-gather of
        #    the previous step still reads it (and the receivers' apply of that step must be done before this step's arrives)
        self.wait_table_sync()
        extra = {} if keep is None else {"skip_extra": keep}  # (an overflowed list exchange: the step is skipped, the gradient kept)
        if delta_dtype is not None:
            opt.step_buffer(i, grad_scale, delta16=st["delta"][lo:hi], **extra)
        else:
            opt.step_buffer(i, grad_scale, **extra)
        # 3. all-gather of the updated rows (parameters, or their bf16 update deltas), 4. optionally on its own stream
        cur = torch.cuda.current_stream() if g.is_cuda else None
        comm = None
        if defer and cur is not None:
            if getattr(self, "_comm_stream", None) is None:
                self._comm_stream = torch.cuda.Stream(device=g.device)
            comm = self._comm_stream
            comm.wait_stream(cur)
        with (torch.cuda.stream(comm) if comm is not None else _nullcontext()):
            buf = st["delta"] if delta_dtype is not None else p
            if nccl:
                dist.all_gather_into_tensor(buf, buf[lo:hi], group=self.group)  # in place: input = output's own slice
            else:
                st["mine"].copy_(buf[lo:hi])
                dist.all_gather([buf[r * per:(r + 1) * per] for r in range(self.world)], st["mine"], group=self.group)
            if delta_dtype is not None:
                apply_delta(p, st["delta"], lo, hi)  # p[outside the shard] += float(delta)
            if comm is not None:
                self._table_sync = torch.cuda.Event()
                self._table_sync.record(comm)
                _DEFERRED.add(self)
        if transport is None and lists is None:  # (the bf16 send pass / the compaction has cleared the local gradient already)
            if lo > 0:
                g[:lo].zero_()
            if hi < n:
                g[hi:].zero_()
        e_rs = 4 if transport is None else 2
        e_ag = 4 if delta_dtype is None else 2
        f = (self.world - 1) / self.world
        self.last_sparse = {"mode": "shard", "rows": [per] * self.world, "bytes": int(f * n * (e_rs + e_ag)),
                            "reduce_scatter_bytes_per_gpu": int(f * n * e_rs), "all_gather_bytes_per_gpu": int(f * n * e_ag),
                            "reduce_scatter_dtype": "float32" if transport is None else "bfloat16",
                            "all_gather": "float32 parameters" if delta_dtype is None else "bfloat16 update deltas", "deferred": bool(comm is not None),
                            "gradient_half": "dense reduce-scatter"}
        if lists is not None:
            self.last_sparse.update({"gradient_half": "row lists to the shard owners (all-to-all, fp32, owner adds in rank order)",
                                     "reduce_scatter_bytes_per_gpu": lists["bytes"], "bytes": lists["bytes"] + int(f * n * e_ag),
                                     "list_rows_per_destination": lists["caps"], "dense_reduce_scatter_would_be": int(f * n * 4)})
        return self.last_sparse

    def _lists_to_owners(self, g: torch.Tensor, lo: int, hi: int, F: int, kernels=None, found_inf=None) -> Optional[dict]:
        """The gradient half of the sharded step for a SPARSE gradient: afterwards g[lo:hi] holds the sum over the ranks of the
        rows this rank owns and the rest of g is zero -- what the dense reduce-scatter (+ clearing) leaves -- without moving
        the zeros.  g: flat [world * per] fp32 gradient of a [rows, F] table, shard d = elements [d * per, (d + 1) * per).

          1. nr_grad_compact_shards: ONE launch; destination d's non-zero rows -> segment d (capacity caps[d] rows) of the send
             lists, cleared in g; counts[d] = all non-zero rows of shard d.
          2. the counts are all-gathered ([world, world] on every rank) and copied to pinned host memory asynchronously: they
             size the segments of the step `count_lag` (2) steps later -- by then the copy has landed long ago and the host is
             never held within a step of the device -- at 1.3 x the largest list a destination received, kept while it stays
             within [1.1, 1.6] x; (1 + 1.3 lag) x after an overflow, whose kept gradient the following steps add onto.  The same
             counts decide, `count_lag` steps late and alike on every rank, to fall back to the dense reduce-scatter when a list
             would exceed half a shard's rows (there the lists' 4 + 4F bytes per row stop paying against 4F per row of the dense
             ring).  No host read after the first call (which counts once, synchronously, before any timed region).
          3. all-to-all of the index parts, all-to-all of the value parts (fixed sizes known to every rank: segment d of every
             rank goes to rank d).
          4. nr_grad_lists_apply per source rank, IN RANK ORDER, plain adds onto g[lo:hi]: every owner computes a sum that
             depends on the lists only -- fp32, exact to the addition order 0 .. world-1, identical however the ranks' lists
             arrive.  (The dense ring adds in an order that depends on the ring position; both are fp32 sums of the same
             `world` addends.)
          5. a step whose count jumps past a segment's capacity is caught on the device from the gathered counts: nobody applies
             anything, nr_grad_lists_restore puts every rank's own rows back, the returned flag (= 2) makes the owner's Adam skip
             the step and keep the gradient; the next step's scatter adds onto it and the next exchange (sized from the counts
             known by then) carries both.  Replicas stay bit-identical, nothing is lost.  Exception: a step the loss scaler
             rejects (found_inf raised: the optimizer skips AND clears) -- an inf / NaN loss makes every touched vertex non-zero,
             so the count jumps on exactly such a step; its gradient is then discarded on every rank instead of kept.
        Returns {"flag", "bytes" (per GPU on the wire), "caps"} or None when this step is to go dense (the caller then runs the
        reduce-scatter).  Reference semantics: the fp32 gradient mean of DDP, pipelines/base_pipeline.py:305-307."""
        world, per = self.world, hi - lo
        rows = per // F  # rows per shard
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        if kernels is None:
            from . import ops as hip_ops
            kernels = (hip_ops.grad_compact_shards, hip_ops.grad_lists_apply, hip_ops.grad_lists_restore)
        compact_shards, lists_apply, lists_restore = kernels
        dev = g.device
        key = ("lists", g.data_ptr(), F)
        st = self.__dict__.setdefault("_lists_state", {}).get(key)
        i32 = dict(device=dev, dtype=torch.int32)
        if st is None:
            st = dict(counts=torch.zeros(world, **i32), cm=torch.zeros(world * world, **i32), flag=torch.zeros(1, device=dev, dtype=torch.float32),
                      host=[], pending=[], turn=0, calls=0,
                      zero_caps=torch.zeros(world, **i32), need=None, caps=None, caps_dev=None, dense_steps=0,
                      idx_s=None, val_s=None, idx_r=None, val_r=None)
            self._lists_state[key] = st
        cap_max = max(rows // 2, 1)

        lag = max(1, int(getattr(self, "count_lag", 2)))  # a step is sized from the counts of `lag` steps ago

        def gather_counts(sync: bool):
            self._all_gather_g(st["cm"], st["counts"])
            if sync:
                st["host"][0].copy_(st["cm"])
                st["need"] = st["host"][0].view(world, world).max(dim=0).values.tolist()
                return
            buf = st["host"][st["turn"] % len(st["host"])]  # (lag + 1 pinned buffers in turn: the one written now is read `lag` steps on)
            st["turn"] += 1
            ev = None
            if dev.type == "cuda":
                buf.copy_(st["cm"], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:
                buf.copy_(st["cm"])
            st["pending"].append((st["calls"], ev, buf))

        if len(st["host"]) != lag + 1:
            mk = lambda: torch.zeros(world * world, dtype=torch.int32)  # noqa: E731
            st["host"] = [mk().pin_memory() if dev.type == "cuda" else mk() for _ in range(lag + 1)]
        # The counts that size THIS step: those gathered `lag` steps ago (default 2).  Waiting for the previous step's copy
        # would hold the host within one step of the device on every step; two steps back the copy has landed long ago and the
        # host keeps its run-ahead (graph segments: tests/test_gpu_dp.py::test_segment_replay_takes_the_host_out...).  Every
        # rank applies the same rule, so every rank sizes its segments from the same counts.
        st["calls"] += 1
        while st["pending"] and st["pending"][0][0] <= st["calls"] - lag:  # (by call number: the same entries on every rank)
            _, ev, buf = st["pending"].pop(0)
            if ev is not None and not ev.query():
                # the host has run `lag` steps ahead of the device: it waits here for the counts -- back-pressure, not work; the time
                # is accounted separately (bench.py reports the launch loop's host time without it)
                import time

                t0 = time.perf_counter()
                ev.synchronize()
                self.host_wait_s = getattr(self, "host_wait_s", 0.0) + time.perf_counter() - t0
            st["need"] = buf.view(world, world).max(dim=0).values.tolist()
        if st["need"] is None:  # first call: count once, synchronously (rows stay where they are: capacities of zero)
            st["counts"].zero_()
            compact_shards(g, F, world, st["zero_caps"], st["zero_caps"], st["flag"], st["counts"])
            gather_counts(sync=True)
        need = st["need"]
        dense = max(need) > cap_max or (st["dense_steps"] > 0 and 2 * max(need) > cap_max)  # (coming back needs 2x room)
        if dense:
            st["dense_steps"] += 1
            if st["dense_steps"] % self.dense_probe_every == 0:  # dense is not for ever: re-count now and then (one read of g)
                st["counts"].zero_()
                compact_shards(g, F, world, st["zero_caps"], st["zero_caps"], st["flag"], st["counts"])
                gather_counts(sync=False)
            return None
        st["dense_steps"] = 0
        caps = st["caps"]
        gr = int(getattr(self, "list_granularity", 1024))
        if caps is not None and any(nd > c for nd, c in zip(need, caps)):
            # a step overflowed: its gradient was kept and the following steps' scatters have added onto it -- room for the union
            caps = [min(cap_max, max(gr, (int((1.0 + 1.3 * lag) * nd) + gr - 1) // gr * gr)) for nd in need]  # (lag more batches pile up meanwhile)
        elif caps is None or any(not (1.1 * nd <= c) or c > max(1.6 * nd, gr) for nd, c in zip(need, caps)):
            caps = [min(cap_max, max(gr, (int(1.3 * nd) + gr - 1) // gr * gr)) for nd in need]
        if caps != st["caps"]:
            # new capacities reach the device through a ring of pinned staging buffers and an asynchronous copy on the exchange's
            # own stream (a pageable host-to-device copy would hold the host until the stream has drained -- and a fresh model's
            # counts grow by tens of per cent per step for its first hundred steps: measured 9-18 ms of host time per step)
            if st["caps_dev"] is None:
                st["caps_dev"] = torch.zeros(world, **i32)
                mk = lambda: torch.zeros(world, dtype=torch.int32)  # noqa: E731
                st["caps_stage"] = [mk().pin_memory() if dev.type == "cuda" else mk() for _ in range(8)]
            stage = st["caps_stage"][st["turn"] % len(st["caps_stage"])]
            stage.copy_(torch.tensor(caps, dtype=torch.int32))
            st["caps_dev"].copy_(stage, non_blocking=True)
            st["caps"] = caps
        total, mine = sum(caps), caps[rank]
        if st["idx_s"] is None:
            # the lists' buffers, ONCE, for the largest capacities the exchange can ever choose (cap_max rows per destination: 20
            # bytes x rows of the whole table for sending and as much for receiving -- 0.67 GB for the NeuRadar main table, whatever
            # the world size): re-sizing them as the counts drift re-allocates hundreds of megabytes on the step's critical path
            st["idx_s"], st["val_s"] = torch.zeros(world * cap_max, **i32), torch.zeros(world * cap_max, F, device=dev, dtype=torch.float32)
            st["idx_r"], st["val_r"] = torch.zeros(world * cap_max, **i32), torch.zeros(world * cap_max, F, device=dev, dtype=torch.float32)
        idx_s, val_s = st["idx_s"][:total], st["val_s"][:total]
        idx_r, val_r = st["idx_r"][:world * mine], st["val_r"][:world * mine]
        st["counts"].zero_()
        compact_shards(g, F, world, st["caps_dev"], idx_s, val_s, st["counts"])
        gather_counts(sync=False)
        self._all_to_all_rows(idx_r, idx_s, caps, mine)
        self._all_to_all_rows(val_r, val_s, caps, mine)
        shard = g[lo:hi]
        for src in range(world):  # rank order, plain adds
            lists_apply(idx_r[src * mine:(src + 1) * mine], val_r[src * mine:(src + 1) * mine], st["cm"], st["caps_dev"], src, rank, F,
                        shard, st["flag"])
        lists_restore(idx_s, val_s, max(caps), st["cm"], st["caps_dev"], rank, F, g, found_inf)
        return {"flag": st["flag"], "bytes": (total - mine) * (4 + 4 * F), "caps": list(caps)}

    def _all_gather_g(self, out: torch.Tensor, mine: torch.Tensor) -> None:
        """out [world * len(mine)] <- every rank's `mine` in rank order, over self.group (flat form on RCCL, list form on gloo)."""
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out, mine, group=self.group)
        else:
            n = mine.numel()
            dist.all_gather([out[r * n:(r + 1) * n] for r in range(self.world)], mine, group=self.group)

    def _all_to_all_rows(self, out: torch.Tensor, inp: torch.Tensor, caps, mine: int) -> None:
        """Segment d (caps[d] leading-dimension rows) of every rank's `inp` goes to rank d; `out` receives world segments of
        `mine` = caps[own rank] rows in rank order.  RCCL: one all_to_all_single with the split sizes every rank knows.  gloo (the
        CPU / one-GPU tests): every rank gathers all send buffers and copies its own segments out -- same result."""
        world = self.world
        if dist.get_backend(self.group) == "nccl":
            dist.all_to_all_single(out, inp, output_split_sizes=[mine] * world, input_split_sizes=list(caps), group=self.group)
            return
        rank = dist.get_rank(self.group)
        parts = [torch.empty_like(inp) for _ in range(world)]
        dist.all_gather(parts, inp.contiguous(), group=self.group)
        off = sum(caps[:rank])
        for r in range(world):
            out[r * mine:(r + 1) * mine].copy_(parts[r][off:off + mine])

    def wait_table_sync(self) -> None:
        """Make the current stream wait for a deferred all-gather (+ delta apply) of the previous sharded step: call it in
        front of whatever reads the table next.  No-op when nothing is pending."""
        ev = getattr(self, "_table_sync", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def flush(self) -> None:
        """Host-side: block until a deferred all-gather has landed (before checkpoints, evaluation, replica checks)."""
        ev = getattr(self, "_table_sync", None)
        if ev is not None:
            ev.synchronize()
            self._table_sync = None

    # ---- sparse table exchange ---------------------------------------------------------------------
    def reduce_sparse(self, grad: torch.Tensor, row_width: int, cap_rows: Optional[int] = None, ops=None) -> dict:
        """SUM of `grad` (flat view of a [rows, row_width] table gradient) over the ranks by exchanging the
        non-zero rows only; result bit-identical on every rank (lists are added in rank order).  `cap_rows`: the list length
        beyond which the dense all-reduce is cheaper.  Default: the length at which the all-gather of the lists ((world-1) * m *
        (1+w) floats into every rank) costs half of what the ring all-reduce of the table moves (2 (world-1)/world * rows * w
        floats): rows * w / ((1+w) * world) -- rows/10 for the NeuRadar main grid (w = 4) on 8 GPUs, 0.4 rows on 2.

        NO HOST READ after the first call.  The lists travel with a fixed length m chosen from the PREVIOUS step's row counts
        (all-gathered on the device, copied to pinned host memory asynchronously: by the next step the copy has long landed) with
        2x headroom; the same counts decide -- one step late, identically on every rank -- to go dense when the lists no longer
        pay.  A step whose count jumps past m is handled on the device: every rank sees the gathered counts, so every rank takes
        the same branch of the guarded apply (nr_grad_apply_guarded) -- nobody applies foreign rows, every rank puts its own rows
        back, the returned `flag` (= 2) makes the table's optimizer launch skip the step and KEEP the gradient, which the next
        step's scatter adds onto and the next exchange (sized from the counts that are known by then) carries.  The update of that
        table is postponed by one step and sums two batches -- replicas stay bit-identical, nothing is lost.  The very first call
        has no history: it reads the counts once (before any graph capture / timed region).
        Returns {"mode": "sparse" | "dense" | "single", "flag": device float or None, ...}; pass `flag` as the optimizer's skip.
        `ops`: (compact, apply_guarded) callables; default = the HIP kernels (neuradar_amd.ops)."""
        if self.world == 1 and not self.force_collectives:
            return {"mode": "single", "flag": None}
        count_rows = None
        if ops is None:
            from . import ops as hip_ops
            ops = (hip_ops.grad_compact, hip_ops.grad_apply_guarded)

            def count_rows(count_, st_):  # count-only pass of the compaction kernel (capacity 0: nothing moves, no temporaries)
                hip_ops.grad_compact_shards(grad, row_width, 1, st_["zero_cap"], st_["zero_cap"], st_["flag"], count_)
        compact, apply_guarded = ops
        rows = grad.numel() // row_width
        cap = int(cap_rows) if cap_rows is not None else max(rows * row_width // ((1 + row_width) * self.world), 1)
        key = (grad.data_ptr(), cap, row_width)
        st = getattr(self, "_sparse_state", {}).get(key)
        world = self.world
        rank = dist.get_rank(self.sparse_group) if dist.is_initialized() else 0
        if st is None:
            dev = grad.device
            st = dict(idx=torch.zeros(cap, device=dev, dtype=torch.int32), val=torch.zeros(cap * row_width, device=dev, dtype=torch.float32),
                      count=torch.zeros(1, device=dev, dtype=torch.int32), counts=torch.zeros(world, device=dev, dtype=torch.int32),
                      out_idx=torch.zeros(world * cap, device=dev, dtype=torch.int32),
                      out_val=torch.zeros(world * cap * row_width, device=dev, dtype=torch.float32),
                      flag=torch.zeros(1, device=dev, dtype=torch.float32), zero_cap=torch.zeros(1, device=dev, dtype=torch.int32),
                      host=torch.zeros(world, dtype=torch.int32).pin_memory() if dev.type == "cuda" else torch.zeros(world, dtype=torch.int32),
                      event=None, prev_max=None)
            self.__dict__.setdefault("_sparse_state", {})[key] = st
        # ---- what the previous step's counts say about this one (host values, no wait in practice)
        prev = st["prev_max"]
        if st["event"] is not None:
            st["event"].synchronize()  # (recorded a whole step ago)
            prev = int(st["host"].max())
            st["event"] = None
        count, idx, val = st["count"], st["idx"], st["val"]
        first = prev is None
        # lists no longer pay (decided from the last known counts, alike on every rank); coming BACK from dense needs 2x room
        dense = (not first) and (prev > cap or (st.get("dense_steps", 0) > 0 and 2 * prev > cap))
        m = cap if first else min(cap, max(256, (2 * prev + 255) // 256 * 256))
        if dense:
            # (a gradient kept from an overflowed step is part of `grad` and travels with it)
            # Dense is not for ever: one outlier batch must not turn every later step into a full all-reduce.  Every
            # `dense_probe_every` dense steps the non-zero rows are counted again (one read of the gradient) and all-gathered;
            # the host sees them one step later, like the lists' counts, and the exchange returns to lists once they fit twice.
            st["dense_steps"] = st.get("dense_steps", 0) + 1
            if st["dense_steps"] % self.dense_probe_every == 0:
                if count_rows is not None:
                    count.zero_()
                    count_rows(count, st)
                else:  # (torch stand-ins of the CPU tests)
                    count.copy_((grad.view(-1, row_width) != 0).any(dim=1).sum().to(torch.int32).reshape(1))
                self._all_gather(st["counts"], count)
                if grad.is_cuda:
                    st["host"].copy_(st["counts"], non_blocking=True)
                    st["event"] = torch.cuda.Event()
                    st["event"].record()
                else:
                    st["host"].copy_(st["counts"])
                    prev = int(st["host"].max())
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.sparse_group)
            st["prev_max"] = prev
            st["flag"].zero_()
            self.last_sparse = {"mode": "dense", "rows": None, "bytes": grad.numel() * 4, "flag": st["flag"]}
            return self.last_sparse
        st["dense_steps"] = 0
        count.zero_()
        compact(grad, row_width, idx[:m], val[:m * row_width], count)  # rows beyond m stay in grad; count = ALL non-zero rows
        self._all_gather(st["counts"], count)
        if first:
            n_rows = st["counts"].cpu()  # the one synchronous read: sizes the very first exchange
            if int(n_rows.max()) > cap:  # a rank's list does not fit: put the local rows back, reduce densely
                apply_guarded(idx[:m], val[:m * row_width], st["counts"], rank, rank, m, row_width, grad, None)  # (own list: always applied)
                dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.sparse_group)
                st["prev_max"] = int(n_rows.max())
                st["flag"].zero_()
                self.last_sparse = {"mode": "dense", "rows": n_rows.tolist(), "bytes": grad.numel() * 4, "flag": st["flag"]}
                return self.last_sparse
            st["prev_max"] = int(n_rows.max())
            m_x = min(m, max(256, (int(n_rows.max()) + 255) // 256 * 256))  # (known exactly this once)
        else:
            if grad.is_cuda:
                st["host"].copy_(st["counts"], non_blocking=True)
                st["event"] = torch.cuda.Event()
                st["event"].record()
            else:
                st["host"].copy_(st["counts"])
                st["prev_max"] = int(st["host"].max())
            m_x = m
        # two collectives over preallocated buffers: the index parts and the value parts of every rank's list
        oi, ov = st["out_idx"][:world * m_x], st["out_val"][:world * m_x * row_width]
        self._all_gather(oi, idx[:m_x])
        self._all_gather(ov, val[:m_x * row_width])
        for r in range(world):  # rank order, plain adds: every rank computes the same sums
            apply_guarded(oi[r * m_x:(r + 1) * m_x], ov[r * m_x * row_width:(r + 1) * m_x * row_width], st["counts"], r, rank, m_x,
                          row_width, grad, st["flag"])
        self.last_sparse = {"mode": "sparse", "rows": None if not first else n_rows.tolist(), "list_rows": m_x,
                            "bytes": world * m_x * (1 + row_width) * 4, "flag": st["flag"]}
        return self.last_sparse

    def _all_gather(self, out: torch.Tensor, mine: torch.Tensor) -> None:
        """out [world * len(mine)] <- every rank's `mine`, in rank order.  RCCL gathers straight into the flat buffer; gloo
        (CPU tests) only knows the list form."""
        if dist.get_backend(self.sparse_group) == "nccl":
            dist.all_gather_into_tensor(out, mine, group=self.sparse_group)
        else:
            n = mine.numel()
            dist.all_gather([out[r * n:(r + 1) * n] for r in range(self.world)], mine, group=self.sparse_group)

    def bytes_per_step(self) -> int:
        esz = 4 if self.table_dtype is None else torch.empty((), dtype=self.table_dtype).element_size()
        return sum(p.numel() for p in self.big) * esz + sum(p.numel() for p in self.small) * 4

    @torch.no_grad()
    def all_reduce(self) -> None:
        if self.world == 1:
            return
        works = []
        for p in self.big:
            if p.grad is None:  # e.g. proposal_fields[0].hash_table (never evaluated): every rank must still issue the
                p.grad = torch.zeros_like(p)  # matching collective -- all-reduced as zeros
            g = p.grad
            if self.table_dtype is not None and g.numel() > SMALL_PARAM_NUMEL:
                low = g.to(self.table_dtype)
                dist.all_reduce(low, op=dist.ReduceOp.SUM, group=self.group)
                g.copy_(low).div_(self.world)
            else:
                works.append((dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True), g))
        if self.small:
            if self._flat is None:
                self._flat = torch.empty(sum(p.numel() for p in self.small), device=self.small[0].device,
                                         dtype=torch.float32)
            off = 0
            for p in self.small:
                n = p.numel()
                self._flat[off:off + n].copy_(p.grad.reshape(-1) if p.grad is not None else torch.zeros(n, device=p.device))
                off += n
            dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
            self._flat.div_(self.world)
            off = 0
            for p in self.small:
                n = p.numel()
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                p.grad.copy_(self._flat[off:off + n].view_as(p))
                off += n
        for w, g in works:
            w.wait()
            g.div_(self.world)


def broadcast_parameters(module: nn.Module, src: int = 0, group=None) -> None:
    """DDP's initial parameter sync (rank 0's weights everywhere)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
