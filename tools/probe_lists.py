"""GPU probe: the phases of the sharded step's row-list exchange (GradAllReducer._lists_to_owners) in a ONE-RANK RCCL group on the
NeuRadar main table's size (33.5 M rows x 4 floats), gradient with PROBE_DENSITY of the rows non-zero: compaction, the counts'
all-gather + pinned copy, the two all_to_all_single calls, apply + restore -- device time by events, host time by the clock."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuradar_amd import ops  # noqa: E402
from neuradar_amd.parallel import GradAllReducer  # noqa: E402

with socket.socket() as s_:
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
dev = torch.device("cuda", 0)
rows, F = 8 << 22, 4
dens = float(os.environ.get("PROBE_DENSITY", "0.09"))
g0 = torch.where(torch.rand(rows, 1, device=dev) < dens, torch.randn(rows, F, device=dev), torch.zeros(rows, F, device=dev)).reshape(-1)
g = g0.clone()
red = GradAllReducer(None, buffers=[], table_mode="shard")
red.force_collectives = True
red.world = 1


def timed(fn, reps=5):
    out = []
    for _ in range(reps):
        g.copy_(g0)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        fn()
        b.record()
        host = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        out.append((a.elapsed_time(b), host))
    out.sort()
    return out[len(out) // 2]


whole = timed(lambda: red._lists_to_owners(g, 0, rows * F, F))
st = next(iter(red._lists_state.values()))
caps, total = st["caps"], sum(st["caps"])
print(f"rows {rows}, density {dens}: list capacity {caps} rows; whole exchange {whole[0]:.3f} ms device, {whole[1]:.3f} ms host")
idx_s, val_s, idx_r, val_r = st["idx_s"][:total], st["val_s"][:total], st["idx_r"][:total], st["val_r"][:total]


def compact():
    st["counts"].zero_()
    ops.grad_compact_shards(g, F, 1, st["caps_dev"], idx_s, val_s, st["counts"])


print("compaction            %.3f ms device, %.3f ms host" % timed(compact))
print("counts all-gather     %.3f ms device, %.3f ms host" % timed(lambda: red._all_gather_g(st["cm"], st["counts"])))
print("all_to_all (indices)  %.3f ms device, %.3f ms host" % timed(lambda: red._all_to_all_rows(idx_r, idx_s, caps, caps[0])))
print("all_to_all (values)   %.3f ms device, %.3f ms host" % timed(lambda: red._all_to_all_rows(val_r, val_s, caps, caps[0])))
print("apply                 %.3f ms device, %.3f ms host" % timed(lambda: ops.grad_lists_apply(idx_r, val_r, st["cm"], st["caps_dev"], 0, 0, F, g, st["flag"])))
print("restore (+ discard)   %.3f ms device, %.3f ms host" % timed(lambda: ops.grad_lists_restore(idx_s, val_s, max(caps), st["cm"], st["caps_dev"], 0, F, g, st["flag"])))
dist.destroy_process_group()
