"""The data-parallel fused step exercised on a GPU (round 1 only reasoned about it): two ranks share ONE device, gloo
carries the collectives, each rank renders half of a fixed global batch.  Checked: (1) the replicas stay bit-identical
over three optimizer steps; (2) after the first step Adam's first moment -- (1 - beta1) x the reduced mean gradient --
equals the single-process run on the whole batch, for every parameter buffer; (3) both exchange paths of the main table
are hit: (row, value) lists, and the dense fallback when a step touches more than rows / 16."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(tmp, tag, world, extra):
    out = os.path.join(tmp, tag)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 1:
        cmd = [sys.executable, WORKER, "--out", out, *extra]
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), WORKER, "--out", out, *extra]
    try:
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        # two processes sharing one device through gloo's host path: seen once in ~20 runs of the suite to stall in the
        # rendezvous of a freshly started pair (the same command by itself takes 8 s); one more attempt on a new port
        import warnings

        warnings.warn(f"{tag}: the {world}-process run did not finish within 240 s; retrying once")
        if world > 1:
            cmd[cmd.index("--master-port") + 1] = str(_free_port())
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, f"{tag}: rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return [torch.load(f"{out}.rank{k}", weights_only=False) for k in range(world)]


@pytest.mark.parametrize("path,extra", [("sparse", ["--log2t", "20", "--rays", "256"]), ("dense_fallback", ["--log2t", "14", "--rays", "512"]),
                                        ("dense", ["--log2t", "16", "--rays", "512", "--dense"]),
                                        ("shard", ["--log2t", "16", "--rays", "512", "--shard"]),
                                        ("shard_lists", ["--log2t", "20", "--rays", "256", "--shard"]),
                                        ("shard_dense", ["--log2t", "20", "--rays", "256", "--shard", "--dense-shard"]),
                                        ("shard_bf16", ["--log2t", "16", "--rays", "512", "--shard", "--bf16"])])
def test_two_rank_fused_step_on_one_gpu(tmp_path, path, extra):
    dp = _run(str(tmp_path), "dp", 2, extra)
    single = _run(str(tmp_path), "single", 1, extra + ["--steps", "1"])[0]
    first = _run(str(tmp_path), "dp1", 2, extra + ["--steps", "1"])
    # (1) replicas bit-identical after three steps
    for n, p in dp[0]["params"].items():
        assert torch.equal(p, dp[1]["params"][n]), f"{path}: parameter {n} differs between the ranks after 3 steps"
    # (3) the exchange path that was meant to run did run
    modes = {e.get("mode") for e in dp[0]["exchange"]}
    want = {"sparse": {"sparse"}, "dense_fallback": {"dense"}, "dense": {None}, "shard": {"shard"}, "shard_bf16": {"shard"},
            "shard_lists": {"shard"}, "shard_dense": {"shard"}}[path]
    if path in ("shard_lists", "shard_dense"):
        # the gradient half of the sharded step: (row, values) lists to the shard owners by default (a step touches a few per cent
        # of this table's rows), the dense fp32 reduce-scatter on request; the lists move a fraction of the dense bytes
        for e in dp[0]["exchange"]:
            assert e["gradient_half"].startswith("row lists" if path == "shard_lists" else "dense reduce-scatter"), e
            if path == "shard_lists":
                assert e["reduce_scatter_bytes_per_gpu"] < 0.5 * e["dense_reduce_scatter_would_be"], e
    if path == "shard_bf16":  # bf16 on both halves of the exchange, the all-gather deferred (and still: replicas identical, above)
        e0 = dp[0]["exchange"][0]
        assert e0["reduce_scatter_dtype"] == "bfloat16" and e0["all_gather"].startswith("bfloat16") and e0["deferred"]
        assert e0["all_gather_bytes_per_gpu"] * 2 == dp[0]["params"]["field.hashgrid.static_grid.hash_table"].numel() * 2
    assert modes == want, f"{path}: main-table exchange modes {modes}"
    # (2) the reduced gradient of step 1 == the single-process gradient on the whole batch
    for i, (a, b) in enumerate(zip(first[0]["exp_avg"], single["exp_avg"])):
        if path.startswith("shard") and i == first[0]["main_buffer"]:
            # the sharded table: every rank holds the moments of ITS rows only; together they are the single-process moments
            (lo0, hi0), (lo1, hi1) = first[0]["shard"], first[1]["shard"]
            assert (lo0, hi1) == (0, b.numel()) and hi0 == lo1 and a.numel() == hi0 - lo0
            both = torch.cat([a, first[1]["exp_avg"][i]])
            err = float((both - b.reshape(-1)).norm() / b.norm().clamp_min(1e-30))
            tol = 1e-4 if path != "shard_bf16" else 2.0 ** -8  # (the gradient travelled in bf16: 2^-9 per entry)
            assert err < tol, f"{path}: Adam first moment of the main table: relative L2 error {err:.3e} against the single-process run"
            continue
        err = float((a - b).norm() / b.norm().clamp_min(1e-30))
        tol = 2.0 ** -8 if (path == "shard_bf16" and a.numel() > (1 << 16)) else 1e-4  # (tables all-reduced in bf16 there)
        assert err < tol, f"{path}: Adam first moment of buffer {i}: relative L2 error {err:.3e} against the single-process run"
        assert torch.equal(a, first[1]["exp_avg"][i])


def test_rccl_branches_in_a_one_rank_group(tmp_path):
    """The "nccl" (= RCCL) code paths that two ranks on one GPU cannot reach through gloo -- all_gather_into_tensor of the
    row lists, the in-place reduce_scatter_tensor / all_gather_into_tensor of the sharded table step in fp32 and with bf16 on both
    halves (bf16 send buffer, bf16 update deltas gathered in place on the communication stream and applied, deferred into the next
    step) -- executed for real in a one-rank RCCL group (force_collectives): same result as the step without any collective."""
    base = ["--log2t", "18", "--rays", "256", "--steps", "2"]
    plain = _run(str(tmp_path), "plain", 1, base)[0]
    env_port = str(_free_port())
    os.environ["MASTER_PORT"] = env_port
    try:
        for tag, extra in (("rccl_sparse", []), ("rccl_shard", ["--shard"]), ("rccl_shard_dense", ["--shard", "--dense-shard"]),
                           ("rccl_shard_bf16", ["--shard", "--bf16"])):
            got = _run(str(tmp_path), tag, 1, base + extra + ["--backend", "nccl", "--force-collectives"])[0]
            assert {e.get("mode") for e in got["exchange"]} == ({"shard"} if extra else {"sparse"}), got["exchange"]
            if tag == "rccl_shard":  # the row lists' all_to_all_single (+ the counts' all_gather_into_tensor) ran on RCCL
                assert all(e["gradient_half"].startswith("row lists") for e in got["exchange"]), got["exchange"]
            for n, p in plain["params"].items():  # (float atomics: two runs agree to rounding, not bitwise)
                torch.testing.assert_close(got["params"][n], p, rtol=1e-4, atol=1e-6, msg=lambda m, n=n: f"{tag}: parameter {n}: {m}")
    finally:
        os.environ.pop("MASTER_PORT", None)


@pytest.mark.parametrize("workload", ["mixed8192_vod_nll", "mixed16384_neuradar_full", "mixed16384_neuradar_full_fp16"])
def test_two_rank_decoder_workload_replicas_stay_identical(tmp_path, workload):
    """A DECODER workload (BASELINE configs[3]'s per-GPU shape: RGB CNN, lidar MLP, radar transformer + heads, the device-side
    assignment, all inside the step) data-parallel on two ranks of one device through bench.py itself: the proposal chains
    beside the decoder segment with a reducer, the main table sharded with (opt-in) bf16 on both halves of its exchange and the
    all-gather deferred, cnn / transformer optimizers all-reduced, the step REPLAYED as hipGraph segments cut at the collectives
    (fused_step.SegmentedStep) -- bench.py --check-replicas exits non-zero unless every parameter is bit-identical on both ranks
    after the run.  The three decoder workloads (BASELINE configs[3] / [2] full / [4] per-GPU shapes); the line must report graph
    segments."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--workload", workload, "--secondary", "", "--full-model", "", "--trained-steps", "0", "--min-seconds", "0",
           "--no-cpu-baseline", "--no-roofline", "--no-render", "--dist-backend", "gloo", "--single-device", "--check-replicas",
           "--table-transport", "bf16", "--table-delta", "bf16"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    assert "replicas identical on 2 ranks" in r.stderr, r.stderr[-2000:]
    import json

    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    ex = line["config"]["gradient_exchange"]
    assert ex["main_table_mode"] == "shard" and ex["main_table"]["reduce_scatter_dtype"] == "bfloat16" and ex["main_table"]["deferred"]
    cfg = line["config"]
    print(f"{workload}: world 2 on one device: {cfg['graph_segments_per_step']} graph segments per step, host {cfg['host_ms_per_step']} ms / step, "
          f"step {line['ms_per_step']} ms")
    assert cfg["graph_segments_per_step"] >= 2, cfg["graph_segments_per_step"]
    # (no statement about the host time here: gloo blocks the host in every collective -- 250 ms per step for the 537-MB table;
    # test_segment_replay_takes_the_host_out_of_the_data_parallel_decoder_step measures it with real RCCL calls)


def test_two_rank_loss_scaler_skips_on_every_rank(tmp_path):
    """fp16 operands under the device-side loss scaler, two ranks: rank 1's batch of step 1 overflows, rank 0's does not -- the
    found-inf flags are summed over the ranks before the first Adam launch, so BOTH ranks skip that step (and halve the scale):
    replicas bit-identical, nothing non-finite in any parameter or moment, one skipped step counted on both."""
    dp = _run(str(tmp_path), "amp", 2, ["--log2t", "16", "--rays", "512", "--shard", "--fp16-amp"])
    for r in dp:
        assert r["amp"] == {"scale": 512.0, "skipped": 1}, r["amp"]
        for n, p in r["params"].items():
            assert bool(torch.isfinite(p).all()), n
        for m in r["exp_avg"]:
            assert bool(torch.isfinite(m).all())
    for n, p in dp[0]["params"].items():
        assert torch.equal(p, dp[1]["params"][n]), f"parameter {n} differs between the ranks"


@pytest.mark.parametrize("path,extra", [("shard", ["--log2t", "16", "--rays", "512", "--shard"]),
                                        ("shard_lists", ["--log2t", "20", "--rays", "256", "--shard"]),
                                        ("shard_bf16_deferred", ["--log2t", "16", "--rays", "512", "--shard", "--bf16"]),
                                        ("sparse", ["--log2t", "20", "--rays", "256"]),
                                        ("amp", ["--log2t", "16", "--rays", "512", "--shard", "--fp16-amp"])])
def test_two_rank_step_replayed_as_graph_segments_equals_the_eager_step(tmp_path, path, extra):
    """World > 1 no longer means eager launches (VERDICT r04 missing #2): fused_step.SegmentedStep captures the data-parallel step
    ONCE as hipGraph segments cut at the host-side actions -- wait for the deferred all-gather | main gather ... main scatter |
    main table exchange + Adam on the communication stream | proposal scatters | remaining all-reduces + Adam launches -- and
    replays them.  Two ranks on one device, four steps (one eager, three replayed): the replicas stay bit-identical, the
    parameters equal the all-eager run of the same steps (to the rounding of the scatters' float atomics), the exchange modes
    are the same, and for the loss-scaler case both ranks skip the poisoned step."""
    steps = ["--steps", "4"]
    seg = _run(str(tmp_path), "seg", 2, extra + steps + ["--segments"])
    eager = _run(str(tmp_path), "eager", 2, extra + steps)
    assert seg[0]["segments"] in (2, 3), seg[0]["segments"]  # camera-only batches start the proposal chains early: one cut less
    for n, p in seg[0]["params"].items():
        assert torch.equal(p, seg[1]["params"][n]), f"{path}: parameter {n} differs between the ranks after the replayed steps"
        # (float atomics: two runs of the same steps agree to rounding, and Adam turns the rounding of a near-zero gradient into a
        # visible difference of that entry's update -- lr x sign-like ratio: all but 1e-3 of the entries within rtol 1e-4, none beyond steps x lr;
        # one run measured 11 of 393 216 entries off, 5.5e-5 at most.  A missing collective or a stale buffer moves MOST entries)
        q = eager[0]["params"][n]
        off = (p - q).abs() > 1e-6 + 1e-4 * q.abs()
        assert float(off.float().mean()) <= 1e-3 and float((p - q).abs().max()) <= 4e-2, (
            f"{path}: parameter {n} vs the eager run: {int(off.sum())} of {off.numel()} entries off, worst {float((p - q).abs().max()):.3e}")
    assert [e.get("mode") for e in seg[0]["exchange"]] == [e.get("mode") for e in eager[0]["exchange"]]
    if path == "amp":
        for r in seg:
            assert r["amp"] == eager[0]["amp"], (r["amp"], eager[0]["amp"])


def test_segment_replay_takes_the_host_out_of_the_data_parallel_decoder_step():
    """What graph segments are for, measured where it can be measured on one GPU: bench.py --one-rank-collectives runs the
    DATA-PARALLEL step of a decoder workload (reducer, sharded table exchange, all-reduces of every optimizer's buffers) in a
    one-rank RCCL group -- every collective is issued for real and returns at once -- replayed as hipGraph segments.  The host
    spends well under the step's duration launching it (measured: 1.0 of 3.2 ms; launched eagerly, NR_SEGMENTS=0: 3.6 of 3.7 ms,
    host-bound -- profiles/r05_host_ms_per_step.txt)."""
    import json

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--one-rank-collectives", "--workload", "mixed8192_vod_nll", "--steps", "20",
           "--warmup", "20", "--secondary", "", "--full-model", "", "--trained-steps", "0", "--min-seconds", "0.3", "--no-cpu-baseline",
           "--no-roofline", "--no-render"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = line["config"]
    print(f"one-rank RCCL group: {cfg['graph_segments_per_step']} segments, host {cfg['host_ms_per_step']} ms / step, step {line['ms_per_step']} ms")
    assert cfg["graph_segments_per_step"] == 3
    assert cfg["host_ms_per_step"] < 0.6 * line["ms_per_step"], (cfg["host_ms_per_step"], line["ms_per_step"])


def test_two_rank_bf16_table_exchange_trains_like_the_fp32_exchange(tmp_path):
    """The opt-in bf16 variant of the sharded table exchange (bf16 reduce-scatter, bf16 update-delta all-gather, deferred) against
    the default fp32 one over 150 optimizer steps on two ranks (a reduced-precision exchange needs more than a
    3-step tolerance check before anyone turns it on): the loss follows the same trajectory -- every step within 15 %, the mean of the last 20 within 3 % -- and the
    tables end up as far from the fp32 run's as a SECOND fp32 run does (measured: 0.120 vs 0.119 in relative L2 after 150 steps --
    the scatters' float atomics make this sensitive trajectory diverge between identical runs; the bf16 exchange adds nothing
    visible on top; what it rounds is 2^-9 of every update and of every partial gradient sum)."""
    base = ["--log2t", "16", "--rays", "512", "--shard", "--steps", "150"]
    f32 = _run(str(tmp_path), "f32", 2, base)
    f32b = _run(str(tmp_path), "f32b", 2, base)  # the yardstick: a second fp32 run (the scatters' float atomics make runs differ)
    b16 = _run(str(tmp_path), "b16", 2, base + ["--bf16"])
    lf, lb = torch.tensor(f32[0]["losses"]), torch.tensor(b16[0]["losses"])
    print(f"loss fp32 exchange {float(lf[:5].mean()):.5f} -> {float(lf[-20:].mean()):.5f}; bf16 exchange {float(lb[:5].mean()):.5f} -> {float(lb[-20:].mean()):.5f}")
    # (the worker's per-ray random targets cannot be fitted -- its loss moves by an order of magnitude over the run, which makes it
    # a sensitive trajectory: what is asserted is that the two exchanges FOLLOW THE SAME ONE, step for step)
    assert float((lf - lf[0]).abs().max()) > 0.5 * float(lf[0]), "the parameters did not move"
    dev_ = ((lb - lf).abs() / lf.abs().clamp_min(1e-6))
    print(f"largest per-step relative deviation of the loss {float(dev_.max()):.3e}, over the last 20 steps {float(dev_[-20:].mean()):.3e}")
    # (measured: 5.0e-2 at the worst step, 1.8e-2 over the last 20; two fp32 runs differ by 0.5 % in the last-20 mean -- float atomics)
    assert float(dev_.max()) <= 0.15 and abs(float(lb[-20:].mean()) - float(lf[-20:].mean())) <= 0.03 * float(lf[-20:].mean())
    t32 = f32[0]["params"]["field.hashgrid.static_grid.hash_table"].double()
    t16 = b16[0]["params"]["field.hashgrid.static_grid.hash_table"].double()
    t32b = f32b[0]["params"]["field.hashgrid.static_grid.hash_table"].double()
    rel, noise = float((t16 - t32).norm() / t32.norm()), float((t32b - t32).norm() / t32.norm())
    print(f"main table after 150 steps: relative L2 distance bf16 vs fp32 exchange {rel:.3e}; two fp32 runs {noise:.3e}")
    assert rel <= 3.0 * noise + 2e-2, (rel, noise)
    for n, p in b16[0]["params"].items():
        assert torch.equal(p, b16[1]["params"][n]), f"bf16 exchange: parameter {n} differs between the ranks"

