"""Worker of tests/test_gpu_dp.py: one data-parallel rank (or the single-process reference run) of the fused training step
on ONE GPU, gloo between the ranks.  Not a test module."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--rays", type=int, default=512, help="GLOBAL batch (split over the ranks)")
    ap.add_argument("--log2t", type=int, default=20)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--shard", action="store_true", help="main table: reduce-scatter -> Adam on this rank's rows -> all-gather")
    ap.add_argument("--dense-shard", action="store_true", help="shard: dense fp32 reduce-scatter instead of row lists to the owners")
    ap.add_argument("--bf16", action="store_true", help="shard: bf16 reduce-scatter, bf16 update-delta all-gather, deferred into the next step")
    ap.add_argument("--fp16-amp", action="store_true", help="fp16 MFMA operands under the device-side loss scaler; rank 1's batch of "
                    "step 1 is poisoned (an overflowing target): BOTH ranks must skip that step")
    ap.add_argument("--segments", action="store_true", help="step 0 eager, then the step captured as hipGraph segments cut at the "
                    "collectives (fused_step.SegmentedStep) and REPLAYED for the remaining steps, from static input buffers")
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--force-collectives", action="store_true", help="world 1: issue the collectives anyway (one-rank RCCL group)")
    args = ap.parse_args()
    from neuradar_amd.fused_step import FusedTrainStep
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.parallel import GradAllReducer, broadcast_parameters, init_distributed
    from neuradar_amd.step import FlatAdam, HotPathConfig, NeuRadarHotPath

    if args.force_collectives and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        import torch.distributed as dist

        torch.cuda.set_device(0)
        dist.init_process_group(backend=args.backend, init_method=f"tcp://127.0.0.1:{os.environ.get('MASTER_PORT', '29533')}", rank=0, world_size=1)
        rank, world = 0, 1
    else:
        rank, world, _ = init_distributed(backend=args.backend)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = HotPathConfig(field=NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=args.log2t))))
    for pc in (cfg.proposal_field_1, cfg.proposal_field_2):
        pc.grid.static.log2_hashmap_size = 16
    torch.manual_seed(0)
    model = NeuRadarHotPath(cfg).to(dev).train()
    with torch.no_grad():
        model.field.hashgrid.static_grid.hash_table.mul_(200.0)
        model.proposal_fields[1].hashgrid.static_grid.hash_table.mul_(1000.0)
    broadcast_parameters(model)
    groups = model.get_param_groups()
    unused = list(model.proposal_fields[0].parameters())
    opts = [FlatAdam(groups["hashgrids"], lr=1e-2, eps=1e-15, skip=unused),
            FlatAdam(groups["fields"], lr=1e-2, eps=1e-15, weight_decay=1e-7, adamw=True, skip=unused)]
    reducer = GradAllReducer(None, buffers=[g for o in opts for g in o.grad_buffers()],
                             table_mode="shard" if args.shard else ("dense" if args.dense else "sparse"))
    reducer.force_collectives = args.force_collectives
    reducer.shard_lists = not args.dense_shard
    if args.bf16:
        reducer.table_dtype, reducer.table_delta, reducer.defer_gather = torch.bfloat16, torch.bfloat16, True
    shard = None
    if args.shard and (world > 1 or args.force_collectives):
        i_main = opts[0].buffer_of(model.field.hashgrid.static_grid.hash_table)
        if args.force_collectives and world == 1:
            reducer.world = 1
            shard = opts[0].shard_buffer(i_main, 0, 1, force=True)  # one rank owns everything
        else:
            shard = opts[0].shard_buffer(i_main, rank, world)
            assert shard is not None
    B = args.rays
    g = torch.Generator().manual_seed(42)  # the GLOBAL batch, identical in every process
    o = torch.cat([-50 + 100 * torch.rand(B, 1, generator=g), torch.randn(B, 1, generator=g), torch.full((B, 1), 1.6)], -1)
    d = torch.nn.functional.normalize(torch.cat([torch.ones(B, 1), 0.4 * torch.randn(B, 2, generator=g)], -1), dim=-1)
    area = torch.full((B,), 2.25e-6)
    tf, td = 0.1 * torch.randn(B, 32, generator=g), 5.0 + 50.0 * torch.rand(B, generator=g)
    draws = [(torch.rand(B, 129, generator=g), torch.rand(B, generator=g), torch.rand(B, generator=g)) for _ in range(args.steps)]
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    sl = lambda t: t[lo:hi].contiguous().to(dev)  # noqa: E731
    amp = None
    if args.fp16_amp:
        from neuradar_amd.step import GradScalerState

        model.field.config.mlp_dtype, model.field.config.mlp_grad_scale = "float16", 1024.0
    step = FusedTrainStep(model, hi - lo)
    if args.fp16_amp:
        amp = GradScalerState(dev, init_scale=1024.0).attach(opts)
        step.set_grad_scaler(amp)
    fars = torch.full((hi - lo,), 1e6, device=dev)
    info, losses = [], []
    use_reducer = reducer if (world > 1 or args.force_collectives) else None
    static = [sl(o), sl(d), sl(area), fars, sl(tf), sl(td), sl(draws[0][0]), sl(draws[0][1]), sl(draws[0][2])]  # --segments: the graphs' inputs
    seg = None
    for k in range(args.steps):
        tr, j1, j2 = draws[k]
        tf_k = sl(tf)
        if args.fp16_amp and k == 1 and rank == 1:
            tf_k = torch.full_like(tf_k, 1e30)  # only THIS rank's gradients overflow
        if args.segments:
            from neuradar_amd.fused_step import SegmentedStep

            for buf, src in zip(static[4:], (tf_k, sl(td), sl(tr), sl(j1), sl(j2))):
                buf.copy_(src)
            run = lambda: step.forward_backward(*static, optimizers=tuple(opts), reducer=use_reducer)  # noqa: E731
            if k == 0:
                run()  # eager: lazy allocations, the exchange's one host read
            else:
                if seg is None:
                    seg = SegmentedStep(step).capture(run)  # (nothing executes during capture)
                seg.replay()
        else:
            step.forward_backward(sl(o), sl(d), sl(area), fars, tf_k, sl(td), sl(tr), sl(j1), sl(j2), optimizers=tuple(opts), reducer=use_reducer)
        if k == args.steps - 1:
            reducer.flush()  # (a deferred all-gather of the last step; earlier ones are waited for by the next step's gather)
        torch.cuda.synchronize()
        info.append(dict(reducer.last_sparse))
        losses.append(float(step.loss.sum()))
    out = {"rank": rank, "world": world, "exchange": [{k_: v_ for k_, v_ in e.items() if k_ != "flag"} for e in info],
           "amp": None if amp is None else {"scale": amp.get_scale(), "skipped": amp.skipped_steps()},
           "params": {n: p.detach().cpu() for n, p in model.named_parameters()},
           "exp_avg": [m.reshape(-1).cpu() for o_ in opts for m, _ in o_.state], "shard": shard,
           "main_buffer": opts[0].buffer_of(model.field.hashgrid.static_grid.hash_table),
           "segments": None if seg is None else len(seg.parts), "losses": losses}
    torch.save(out, f"{args.out}.rank{rank}")
    if world > 1 or args.force_collectives:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
