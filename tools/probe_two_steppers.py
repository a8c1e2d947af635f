"""How much of the chip does one training step leave idle?  Two INDEPENDENT models stepped concurrently (each its own hipGraph of
two steps on its own stream) against one alone: the aggregate rays/s bounds what overlapping step k+1's latency-bound front
(sampling rounds, gathers) with step k's throughput-bound tail (scatters, Adam) could gain (development tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_graph_replay import _setup  # noqa: E402

WL, DT = sys.argv[1] if len(sys.argv) > 1 else "mixed16384_neuradar", "bfloat16"
N = 2 if len(sys.argv) < 3 else int(sys.argv[2])
import bench  # noqa: E402

n_rays = bench.WORKLOADS[WL]["rays"]
jobs = []
for j in range(N):
    model, fwd_bwd, stepper = _setup(WL, DT)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(4):
            fwd_bwd()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fwd_bwd()
            fwd_bwd()
    torch.cuda.synchronize()
    jobs.append((model, fwd_bwd, stepper, s, g))


def run(active, reps=60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for (_, _, _, s, g) in active:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return len(active) * reps * 2 * n_rays / dt, dt / (reps * 2) * 1e3


for _ in range(2):
    one, ms1 = run(jobs[:1])
    both, ms2 = run(jobs)
    print(f"{WL}: one stepper {one / 1e6:.3f} M rays/s ({ms1:.3f} ms/step); {N} concurrent {both / 1e6:.3f} M rays/s aggregate "
          f"({ms2:.3f} ms per step pair) -> x{both / one:.3f}", flush=True)
