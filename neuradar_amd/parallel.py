"""Data parallelism of the path: rays shard across ranks, parameters are replicated, one gradient
all-reduce (mean) per step -- the behaviour of the reference's DDP wrap
(pipelines/base_pipeline.py:305-307, scripts/train.py:110-164) over RCCL/xGMI.

MI355X-first choices: >99 % of the gradient bytes are hash tables, so they get
their own large flat buckets (one collective per table, no DDP bucketing/copy), all small parameters
are packed into a single fused bucket, and unused parameters (proposal_fields[0]) are simply
all-reduced as zeros instead of DDP's find_unused_parameters bitmap exchange.
"""
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn

SMALL_PARAM_NUMEL = 1 << 16


def init_distributed(backend: Optional[str] = None) -> tuple:
    """(rank, world, local_rank) from the torchrun environment; no-op for a single process."""
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
        dist.init_process_group(backend=backend)
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
    def shard_step(self, opt, i: int, grad_scale: float, transport: Optional[torch.dtype] = None, chunks: int = 1) -> dict:
        """The data-parallel step of ONE big table whose gradient is dense across the ranks (the mixed batch touches 9-22 % of
        the main table's rows per rank: the union over 8 ranks is most of the table, so row lists no longer pay).  Instead of
        all-reduce + a replicated Adam over the whole table on every GPU:
          1. reduce-scatter (SUM) of the gradient: rank r receives the reduced rows [lo, hi) = its 1/world of the table
             (`transport`: optionally carried in bf16 -- half the bytes of the step's largest exchange);
          2. Adam on those rows only (opt.shard_buffer: moments exist for the shard only; the dense 4.3 GB stream of the
             replicated step becomes 1/world of it on every GPU);
          3. all-gather of the updated rows into every replica (fp32, in place: replicas stay bit-identical by construction --
             everybody receives the owner's result);
          4. the gradient outside the shard is cleared (the owner's Adam kernel clears its own rows).
        The same bytes on the wire as a ring all-reduce of the table (reduce-scatter + all-gather IS that all-reduce), but the
        optimizer sits between the halves.  `chunks`: the shard is exchanged in that many pieces so that Adam of piece k runs
        while piece k+1 is still being reduced (collectives execute in issue order on the communicator).
        No host read, no allocation after the first call.  opt: step.FlatAdam with shard_buffer(i) applied."""
        p, g = opt.buffers[i]
        if self.world == 1 and not self.force_collectives:
            opt.step_buffer(i, grad_scale)
            return {"mode": "single"}
        lo, hi = opt.shards[i]
        n, per = p.numel(), hi - lo
        nccl = dist.get_backend(self.group) == "nccl"
        key = ("shard", g.data_ptr(), transport)
        st = self.__dict__.setdefault("_shard_state", {}).get(key)
        if st is None:
            st = {}
            if transport is not None:
                st["low"] = torch.empty(n, device=g.device, dtype=transport)
                st["low_out"] = torch.empty(per, device=g.device, dtype=transport)
            if not nccl:  # gloo (CPU / one-GPU tests) has neither reduce_scatter nor in-place all_gather
                st["mine"] = torch.empty(per, device=g.device, dtype=torch.float32)
            self._shard_state[key] = st
        # 1. reduce-scatter
        if transport is not None:
            st["low"].copy_(g)
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
        # 2. the owner's Adam (zeroes g[lo:hi])
        opt.step_buffer(i, grad_scale)
        # 3. all-gather of the updated rows
        if nccl:
            dist.all_gather_into_tensor(p, p[lo:hi], group=self.group)  # in place: input = output's own slice
        else:
            st["mine"].copy_(p[lo:hi])
            dist.all_gather([p[r * per:(r + 1) * per] for r in range(self.world)], st["mine"], group=self.group)
        # 4. the rest of the local gradient
        if lo > 0:
            g[:lo].zero_()
        if hi < n:
            g[hi:].zero_()
        esz = 4 if transport is None else torch.empty((), dtype=transport).element_size()
        self.last_sparse = {"mode": "shard", "rows": [per] * self.world,
                            "bytes": int((self.world - 1) / self.world * n * (esz + 4))}
        return self.last_sparse

    # ---- sparse table exchange ---------------------------------------------------------------------
    def reduce_sparse(self, grad: torch.Tensor, row_width: int, cap_rows: Optional[int] = None, ops=None) -> dict:
        """SUM of `grad` (flat view of a [rows, row_width] table gradient) over the ranks by exchanging the
        non-zero rows only; result bit-identical on every rank (lists are added in rank order).  Falls back
        to the dense all-reduce when some rank's list exceeds `cap_rows`.  Default: the list length at which the
        all-gather of the lists ((world-1) * m * (1+w) floats into every rank) costs half of what the ring all-reduce of the
        table moves (2 (world-1)/world * rows * w floats): rows * w / ((1+w) * world) -- rows/10 for the NeuRadar main grid
        (w = 4) on 8 GPUs, 0.4 rows on 2; the mixed 16 384-ray batch touches 9 % of that table's rows freshly initialised,
        22 % after 600 steps.  Synchronous
        with respect to the current stream; one small host read (the ranks' row counts) per call.
        `ops`: (compact, apply) callables; default = the HIP kernels (neuradar_amd.ops)."""
        if self.world == 1 and not self.force_collectives:
            return {"mode": "single"}
        if ops is None:
            from . import ops as hip_ops
            ops = (hip_ops.grad_compact, hip_ops.grad_apply)
        compact, apply = ops
        rows = grad.numel() // row_width
        cap = int(cap_rows) if cap_rows is not None else max(rows * row_width // ((1 + row_width) * self.world), 1)
        key = (grad.data_ptr(), cap, row_width)
        st = getattr(self, "_sparse_state", {}).get(key)
        if st is None:
            dev = grad.device
            st = dict(send=torch.zeros(cap * (1 + row_width) + 1, device=dev, dtype=torch.float32),
                      counts=torch.zeros(self.world, device=dev, dtype=torch.int32))
            self.__dict__.setdefault("_sparse_state", {})[key] = st
        send = st["send"]  # [count | idx (int32 bits) x cap | val x cap*row_width]
        count = send[:1].view(torch.int32)
        idx = send[1:1 + cap].view(torch.int32)
        val = send[1 + cap:]
        count.zero_()
        compact(grad, row_width, idx, val, count)
        self._all_gather(st["counts"], count)
        n_rows = st["counts"].cpu()  # the step's one host read: sizes the list exchange (all ranks read the same numbers)
        max_rows = int(n_rows.max())
        if max_rows == 0:  # nobody touched the table: nothing to exchange, nothing to apply
            self.last_sparse = {"mode": "sparse", "rows": n_rows.tolist(), "bytes": 0}
            return self.last_sparse
        if max_rows > cap:  # a rank's list does not fit: put the local rows back, reduce densely
            apply(idx, val, count, row_width, grad)
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.sparse_group)
            self.last_sparse = {"mode": "dense", "rows": n_rows.tolist(), "bytes": grad.numel() * 4}
            return self.last_sparse
        m = min(cap, (max_rows + 255) // 256 * 256)
        # one collective: [count, idx[:m], val[:m*row_width]] of every rank
        piece = 1 + m + m * row_width
        out = torch.empty(self.world * piece, device=grad.device, dtype=torch.float32)
        mine = torch.cat([send[:1 + m], val[:m * row_width]])
        self._all_gather(out, mine)
        for r in range(self.world):
            seg = out[r * piece:(r + 1) * piece]
            apply(seg[1:1 + m].view(torch.int32), seg[1 + m:], seg[:1].view(torch.int32), row_width, grad)
        self.last_sparse = {"mode": "sparse", "rows": n_rows.tolist(), "bytes": self.world * piece * 4}
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
