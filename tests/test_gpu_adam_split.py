"""The main table's Adam in two launches around the scatter (nr_hash_mark_vertices + nr_adam_step_split, ABI v26; DESIGN.md section
13): the step's critical path ended main scatter -> main Adam; the groups this step cannot touch get their zero-gradient update
beside the forward instead.  Reference behaviour: torch.optim.Adam over the dense table (configs/method_configs.py:384-409) --
every entry with a history moves every step, which is what the two launches together must do, bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _state(n_groups, seed):
    g = torch.Generator().manual_seed(seed)
    n = 4 * n_groups
    p = torch.randn(n, generator=g) * 0.1
    hist = torch.rand(n_groups, generator=g) < 0.6            # groups with a history (moments non-zero)
    m = torch.randn(n, generator=g) * 1e-3 * hist.repeat_interleave(4)
    v = torch.rand(n, generator=g) * 1e-6 * hist.repeat_interleave(4)
    touched = torch.rand(n_groups, generator=g) < 0.25        # groups this step's gradient reaches
    grad = torch.randn(n, generator=g) * 1e-3 * touched.repeat_interleave(4)
    grad[::7] = 0.0                                           # (zero components inside touched groups)
    zero_but_stamped = torch.rand(n_groups, generator=g) < 0.1  # stamped vertices whose weight / gradient is 0
    seen = (hist | touched).to(torch.uint8)                   # the scatter marks what it writes
    return p, grad, m, v, seen, touched | zero_but_stamped


@pytest.mark.parametrize("n_groups,epoch", [(1 << 18, 0.0), (100_003, 254.0), (4097, 255.0), (1 << 16, 1234.0)])
def test_two_launches_equal_the_single_marked_launch_bit_for_bit(n_groups, epoch):
    from neuradar_amd import ops

    p0, g0, m0, v0, seen, stamped = _state(n_groups, n_groups)
    now = int(epoch) % 255 + 1
    stale = torch.randint(0, 256, (n_groups,), dtype=torch.int64)
    stale[stale == now] = (now + 7) % 256 if (now + 7) % 256 != now else 3
    stamp = torch.where(stamped, torch.full_like(stale, now), stale).to(torch.uint8)
    hyper = torch.tensor([3e-3, 1.0 - 0.9 ** 5, (1.0 - 0.999 ** 5) ** 0.5], device=DEV)
    ep = torch.tensor([epoch], device=DEV)
    d = lambda t: t.clone().to(DEV)  # noqa: E731
    a = [d(p0), d(g0), d(m0), d(v0)]
    ops.adam_step(*a, 3e-3, 1, (0.9, 0.999), 1e-15, 0.0, False, grad_scale=1.0, zero_grad=True, dev_hyper=hyper, seen_grad=d(seen), marked=True)
    b = [d(p0), d(g0), d(m0), d(v0)]
    for phase in (1, 2):
        ops.adam_step_split(*b, (0.9, 0.999), 1e-15, 1.0, hyper, d(seen), d(stamp), ep, phase)
    for x, y, what in zip(a, b, ("parameters", "gradient (cleared)", "exp_avg", "exp_avg_sq")):
        assert torch.equal(x, y), f"{what}: {int((x != y).sum())} of {x.numel()} elements differ"
    assert float(a[1].abs().max()) == 0.0 and not torch.equal(a[0].cpu(), p0)
    # phase 1 alone leaves every stamped group and the whole gradient buffer untouched
    c = [d(p0), d(g0), d(m0), d(v0)]
    ops.adam_step_split(*c, (0.9, 0.999), 1e-15, 1.0, hyper, d(seen), d(stamp), ep, 1)
    sel = stamped.repeat_interleave(4).to(DEV)
    assert torch.equal(c[1], d(g0)) and torch.equal(c[0][sel], d(p0)[sel]) and torch.equal(c[2][sel], d(m0)[sel])


@pytest.mark.parametrize("n,log2t", [(4099, 12), (65536, 16), (3, 10)])
def test_stamps_cover_everything_the_scatter_writes(n, log2t):
    """Every table entry that nr_hash_encode_bwd_shared writes for these rows carries the step's stamp (the stamps may cover more:
    vertices with weight 0, rows without a gradient), for positions at and across cell boundaries and outside [0, 1]."""
    from neuradar_amd import _lib, ops

    L, F = 8, 4
    gen = torch.Generator().manual_seed(n)
    x = torch.rand(n, 3, generator=gen)
    x[: n // 8] = torch.round(x[: n // 8] * 16) / 16  # exact cell boundaries at the coarse levels
    x[-1] = torch.tensor([1.0, 0.0, 0.999999])
    x = x.to(DEV)
    scalings = torch.floor(16.0 * (8192.0 / 16.0) ** (torch.arange(L) / (L - 1))).to(DEV)
    gout = torch.randn(L, n, F, generator=gen).to(DEV)
    gout[:, ::5] = 0.0
    table_grad = torch.zeros(L << log2t, F, device=DEV)
    lib, p = _lib.lib(), ops._p
    ops.check(lib.nr_hash_encode_bwd_shared(p(x), None, p(scalings), L, F, log2t, p(gout), F, n * F, p(table_grad), n, None, ops._stream()),
              "nr_hash_encode_bwd_shared")
    stamp = torch.zeros(L << log2t, device=DEV, dtype=torch.uint8)
    epoch = torch.tensor([41.0], device=DEV)
    ops.hash_mark_vertices(x, scalings, log2t, stamp, epoch)
    written = (table_grad != 0).any(dim=1)
    assert int(written.sum()) > 0
    assert bool((stamp[written] == 42).all()), f"{int((stamp[written] != 42).sum())} written entries without the step's stamp"
    assert int((stamp == 42).sum()) <= 8 * n * L and bool(((stamp == 0) | (stamp == 42)).all())


def _train(split: str, steps: int, monkeypatch):
    from neuradar_amd.fused_step import FusedTrainStep
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.step import FlatAdam, HotPathConfig, NeuRadarHotPath

    monkeypatch.setenv("NR_ADAM_SPLIT", split)
    cfg = HotPathConfig(field=NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=17))))
    for pc in (cfg.proposal_field_1, cfg.proposal_field_2):
        pc.grid.static.log2_hashmap_size = 14
    torch.manual_seed(0)
    model = NeuRadarHotPath(cfg).to(DEV).train()
    model.field.config.mlp_dtype = "bfloat16"  # (the block-shared scatter + marked Adam: the bench's headline configuration)
    with torch.no_grad():
        model.field.hashgrid.static_grid.hash_table.mul_(200.0)
        model.proposal_fields[1].hashgrid.static_grid.hash_table.mul_(1000.0)
    groups = model.get_param_groups()
    unused = list(model.proposal_fields[0].parameters())
    opts = (FlatAdam(groups["hashgrids"], lr=1e-2, eps=1e-15, warmup_steps=0, skip=unused),
            FlatAdam(groups["fields"], lr=1e-3, eps=1e-15, weight_decay=1e-7, adamw=True, warmup_steps=0, skip=unused))
    B = 512
    g = torch.Generator().manual_seed(42)
    step = FusedTrainStep(model, B)
    fars = torch.full((B,), 1e6, device=DEV)
    used = []
    for k in range(steps):  # a different batch every step: groups touched once keep moving on their moments afterwards
        o = torch.cat([-50 + 100 * torch.rand(B, 1, generator=g), torch.randn(B, 1, generator=g), torch.full((B, 1), 1.6)], -1).to(DEV)
        d = torch.nn.functional.normalize(torch.cat([torch.ones(B, 1), 0.4 * torch.randn(B, 2, generator=g)], -1), dim=-1).to(DEV)
        tf, td = (0.1 * torch.randn(B, 32, generator=g)).to(DEV), (5.0 + 50.0 * torch.rand(B, generator=g)).to(DEV)
        tr, j1, j2 = torch.rand(B, 129, generator=g).to(DEV), torch.rand(B, generator=g).to(DEV), torch.rand(B, generator=g).to(DEV)
        step.forward_backward(o, d, torch.full((B,), 2.25e-6, device=DEV), fars, tf, td, tr, j1, j2, optimizers=opts)
        used.append(getattr(step, "_stamp", None) is not None)
    torch.cuda.synchronize()
    table = model.field.hashgrid.static_grid.hash_table
    m_, v_ = opts[0].state[opts[0].buffer_of(table)]
    return {"table": table.detach().clone(), "m": m_.reshape(-1).clone(), "v": v_.reshape(-1).clone(), "split": all(used),
            "others": {n: p.detach().clone() for n, p in model.named_parameters() if "hash_table" not in n}}


def test_training_steps_with_the_split_adam_equal_the_single_launch(monkeypatch):
    """Six optimizer steps of the fused step (block-shared main scatter, marked Adam) with the Adam split around the scatter
    (NR_ADAM_SPLIT=1) and with the single launch behind it, different batches every step.  The scatters' float atomics make two
    runs of the SAME configuration differ (and Adam turns the rounding of a near-zero gradient into a visible difference of that
    entry's update), so the yardstick is measured: the split run may differ from a single-launch run in at most three times as many
    entries as two single-launch runs differ from each other (+ a floor, the yardstick's own spread), for the table and both of its moments; every
    other parameter within the usual tolerance; moments of groups a later step no longer touches keep decaying (the
    zero-gradient phase did run)."""
    a, b, b2 = _train("1", 6, monkeypatch), _train("0", 6, monkeypatch), _train("0", 6, monkeypatch)
    assert a["split"] and not b["split"]

    def off(x, y):
        x, y = x.reshape(-1), y.reshape(-1)
        return float(((x - y).abs() > 1e-6 * float(y.abs().max()) + 1e-3 * y.abs()).float().mean())

    # floors: the yardstick is itself noisy (single vs single over five runs: table 1.6e-2 ... 2.5e-2, m 3.8e-4 ... 1.4e-3, v ~5e-5);
    # a phase that did not run would show as ~1e-1 of the entries (every group with a history that a later step does not touch)
    floor = {"table": 2e-2, "m": 2e-3, "v": 2e-4}
    for key in ("table", "m", "v"):
        noise, got = off(b2[key], b[key]), off(a[key], b[key])
        print(f"{key}: entries off -- split vs single {got:.3e}, single vs single {noise:.3e}")
        assert got <= 3.0 * noise + floor[key], f"{key}: {got:.3e} of the entries differ (two single-launch runs: {noise:.3e})"
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-30))  # noqa: E731
    for n_, p in a["others"].items():  # (the same yardstick: MLP weights of two single-launch runs differ too -- 16-bit operands, float atomics, Adam)
        noise, got = rel(b2["others"][n_], b["others"][n_]), rel(p, b["others"][n_])
        assert got <= 3.0 * noise + 1e-3, f"{n_}: relative L2 distance {got:.3e} (two single-launch runs: {noise:.3e})"
    assert float((a["m"] != 0).float().mean()) > 0.01
