"""Convergence beside speed (BASELINE's metric is "rays/sec + PSNR"): a small scene -- 256 camera rays into the analytic street
canyon bench.py's `trained` block uses (ground plane, two walls; per-ray feature = 0.5 sin(W . hit point), depth = distance to
the surface) -- trained 200 steps (a) by the fused HIP step with its fused Adam and (b) by the CPU oracle (oracle/pipeline.py:
the reference's torch math) with torch.optim.Adam / AdamW, from the same parameters, on the same rays, with the same jitter
draws.  Required: both fit the scene (PSNR well above the untrained model's), and the two PSNRs agree within 0.5 dB; so does
the fused step on bf16 MFMA operands.  This is what the speed numbers are allowed to claim: the fast path learns what the
reference's arithmetic learns."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
STEPS, B = 200, 256


def _scene(gen):
    """Rays of a forward camera rig in the canyon and their targets."""
    o = torch.stack([-40.0 + 80.0 * torch.rand(B, generator=gen), 2.0 * torch.randn(B, generator=gen), torch.full((B,), 1.6)], dim=-1)
    d = torch.nn.functional.normalize(torch.stack([torch.ones(B), 0.5 * torch.randn(B, generator=gen), 0.25 * torch.randn(B, generator=gen) - 0.05], -1), dim=-1)
    big = torch.full((B,), 200.0)
    tg = torch.where(d[:, 2] < -1e-6, -o[:, 2] / d[:, 2].clamp(max=-1e-6), big)
    tw = torch.where(d[:, 1].abs() > 1e-6, (12.0 * torch.sign(d[:, 1]) - o[:, 1]) / torch.where(d[:, 1].abs() > 1e-6, d[:, 1], big), big)
    depth = torch.minimum(torch.minimum(tg, tw), big)
    Wf = 0.3 * torch.randn(3, 32, generator=gen)
    feats = 0.5 * torch.sin((o + depth[:, None] * d) @ Wf)
    return o, d, torch.full((B,), 2.25e-6), feats, depth


def _psnr(pred, target):
    return -10.0 * math.log10(max(float(((pred - target) ** 2).mean()), 1e-12))  # peak-to-peak 1


def _model():
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.step import HotPathConfig, NeuRadarHotPath

    cfg = HotPathConfig(field=NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=15))))
    for pc in (cfg.proposal_field_1, cfg.proposal_field_2):
        pc.grid.static.log2_hashmap_size = 13
    torch.manual_seed(11)
    return NeuRadarHotPath(cfg).to(DEV).train()


def _train_fused(model, o, d, area, tf, td, draws, mlp_dtype):
    from neuradar_amd.fused_step import FusedTrainStep
    from neuradar_amd.step import FlatAdam

    model.field.config.mlp_dtype = mlp_dtype
    groups = model.get_param_groups()
    unused = list(model.proposal_fields[0].parameters())
    # constant learning rates (warm-up 0, lr_final = lr): the oracle side uses plain torch.optim with the same numbers
    opts = [FlatAdam(groups["hashgrids"], lr=1e-2, eps=1e-15, warmup_steps=0, skip=unused),
            FlatAdam(groups["fields"], lr=2e-3, eps=1e-15, weight_decay=1e-7, adamw=True, warmup_steps=0, skip=unused)]
    fused = FusedTrainStep(model, B)
    dv = lambda t: t.to(DEV).contiguous()  # noqa: E731
    args = (dv(o), dv(d), dv(area), torch.full((B,), 1e6, device=DEV), dv(tf), dv(td))
    curve = []
    for k in range(STEPS):
        tr, j1, j2 = draws[k]
        fused.forward_backward(*args, dv(tr), dv(j1), dv(j2), optimizers=opts)
        if k % 50 == 0 or k == STEPS - 1:
            curve.append(_psnr(fused.outputs()["features"].cpu(), tf))
    out = fused.outputs()
    return curve, float((out["depth"][:, 0].cpu() - td).abs().mean())


def _train_oracle(model, o, d, area, tf, td, draws):
    from oracle import field as of, pipeline as op

    c = lambda t: t.detach().cpu().clone().requires_grad_(True)  # noqa: E731
    f, p = model.field, model.proposal_fields[1]
    sg, pg = f.hashgrid.static_grid, p.hashgrid.static_grid
    fp = of.FieldParams(of.GridParams(c(sg.hash_table), sg.scalings.cpu(), sg.log2_hashmap_size),
                        [(c(l.weight), c(l.bias)) for l in f.mlp_geo.layers], [(c(l.weight), c(l.bias)) for l in f.mlp_feature.layers],
                        c(f.sdf_to_density.beta), f.hashgrid.static_scale)
    pp = of.ProposalParams(of.GridParams(c(pg.hash_table), pg.scalings.cpu(), pg.log2_hashmap_size), c(p.density_decoder.weight),
                           p.hashgrid.static_scale)
    tables = [fp.grid.table, pp.grid.table]
    small = [t for t in fp.tensors() + pp.tensors() if all(t is not q for q in tables)]
    o_t = torch.optim.Adam(tables, lr=1e-2, eps=1e-15)
    o_s = torch.optim.AdamW(small, lr=2e-3, eps=1e-15, weight_decay=1e-7)
    bundle = {"origins": o, "directions": d, "pixel_area": area[:, None], "fars": torch.full((B, 1), 1e6)}
    curve = []
    for k in range(STEPS):
        tr, j1, j2 = draws[k]
        out = op.nff_outputs(fp, [pp, pp], bundle, tr, (j1[:, None], j2[:, None]))
        loss = op.train_loss(out, tf, td[:, None])
        o_t.zero_grad(set_to_none=True)
        o_s.zero_grad(set_to_none=True)
        loss.backward()
        o_t.step()
        o_s.step()
        if k % 50 == 0 or k == STEPS - 1:
            curve.append(_psnr(out["features"].detach(), tf))
    return curve, float((out["depth"].detach()[:, 0] - td).abs().mean())


def test_fused_step_converges_like_the_oracle():
    gen = torch.Generator().manual_seed(2024)
    o, d, area, tf, td = _scene(gen)
    draws = [(torch.rand(B, 129, generator=gen), torch.rand(B, generator=gen), torch.rand(B, generator=gen)) for _ in range(STEPS)]
    start = {k: v.detach().clone() for k, v in _model().state_dict().items()}

    def fresh():
        m = _model()
        m.load_state_dict(start)
        return m

    oracle_curve, oracle_l1 = _train_oracle(fresh(), o, d, area, tf, td, draws)
    f32_curve, f32_l1 = _train_fused(fresh(), o, d, area, tf, td, draws, "float32")
    bf16_curve, bf16_l1 = _train_fused(fresh(), o, d, area, tf, td, draws, "bfloat16")
    fmt = lambda c: " -> ".join(f"{v:.2f}" for v in c)  # noqa: E731
    print(f"feature PSNR [dB] at steps 0/50/100/150/199: oracle (CPU, torch.optim) {fmt(oracle_curve)} | fused fp32 {fmt(f32_curve)} | "
          f"fused bf16 {fmt(bf16_curve)};  depth L1 [m] {oracle_l1:.2f} / {f32_l1:.2f} / {bf16_l1:.2f}")
    # the untrained model renders ~0 features: PSNR = -10 log10(mean target^2) ~ 9 dB; trained: well above
    assert oracle_curve[-1] > oracle_curve[0] + 6.0 and f32_curve[-1] > f32_curve[0] + 6.0 and bf16_curve[-1] > bf16_curve[0] + 6.0
    assert abs(f32_curve[0] - oracle_curve[0]) < 0.05, "the two sides did not start from the same model"
    assert abs(f32_curve[-1] - oracle_curve[-1]) < 0.5, (f32_curve, oracle_curve)
    assert abs(bf16_curve[-1] - oracle_curve[-1]) < 0.5, (bf16_curve, oracle_curve)
