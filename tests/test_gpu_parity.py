"""GPU parity: the HIP path (through the C ABI) against the reference-generated golden vectors and the
CPU oracle on seeded inputs.  Tolerance: the north-star's 1e-4 relative (fp32), written per check;
integer/index work (hash slots, did_return, scan ids) is bit-exact."""
import numpy as np
import pytest
import torch

from helpers import assert_close, load_golden, oracle_field_params, oracle_prop_params

pytestmark = pytest.mark.gpu

DEV = "cuda"


def dev(x):
    return x.to(DEV) if isinstance(x, torch.Tensor) else x


def cpu(x):
    return x.detach().cpu()


# ------------------------------------------------------------------------------------------------ helpers
def make_field(g, prefix="", log2t_key="log2t", hidden=32, static=None):
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig

    L = g[f"{prefix}scalings"].numel()
    F = g[f"{prefix}table"].shape[1]
    st = static or StaticSettings(hashgrid_dim=F, num_levels=L, log2_hashmap_size=int(g[log2t_key]))
    fld = NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=st), geo_hidden_dim=hidden, nff_hidden_dim=hidden
                            ).setup(actors=None, static_scale=100.0).to(DEV)
    with torch.no_grad():
        fld.hashgrid.static_grid.hash_table.copy_(g[f"{prefix}table"])
        fld.hashgrid.static_grid.scalings.copy_(g[f"{prefix}scalings"])
        for i, l in enumerate(fld.mlp_geo.layers):
            l.weight.copy_(g[f"{prefix}geo_w{i}"]); l.bias.copy_(g[f"{prefix}geo_b{i}"])
        for i, l in enumerate(fld.mlp_feature.layers):
            l.weight.copy_(g[f"{prefix}feat_w{i}"]); l.bias.copy_(g[f"{prefix}feat_b{i}"])
        fld.sdf_to_density.beta.copy_(g[f"{prefix}beta"])
    return fld


def make_prop(g, prefix="prop_", log2t_key="prop_log2t"):
    from neuradar_amd.neurad_field import NeuRADProposalFieldConfig

    pc = NeuRADProposalFieldConfig()
    pc.grid.static.log2_hashmap_size = int(g[log2t_key])
    pf = pc.setup(actors=None, static_scale=100.0).to(DEV)
    with torch.no_grad():
        pf.hashgrid.static_grid.hash_table.copy_(g[f"{prefix}table"])
        pf.hashgrid.static_grid.scalings.copy_(g[f"{prefix}scalings"])
        pf.density_decoder.weight.copy_(g[f"{prefix}decoder"])
    return pf


def samples_from_edges(g, edges):
    from neuradar_amd.rays import RaySamples

    B = edges.shape[0]
    return RaySamples(dev(g["origins"]), dev(g["directions"]), dev(g["pixel_area"]), dev(torch.zeros_like(edges)),
                      dev(edges), dev(torch.zeros(B, 1)), dev(torch.full((B, 1), 1e6)))


# ------------------------------------------------------------------------------------------------ a8
@pytest.mark.parametrize("tag", ["l8f4", "l6f1", "l16f2", "l4f4"])
@pytest.mark.parametrize("level_major", [False, True])
def test_hash_encode_fwd_bwd_vs_reference_golden(tag, level_major):
    from neuradar_amd import ops

    g = load_golden("hash_encode")
    x, table, sc = dev(g[f"{tag}_x"]), dev(g[f"{tag}_table"]).requires_grad_(True), dev(g[f"{tag}_scalings"])
    n, L, F = x.shape[0], sc.numel(), table.shape[1]
    sm = 60 if level_major else 0  # 300 = 5 "rays" x 60 samples: exercises the sample-major lane mapping
    out = ops.hash_encode(x, table, sc, int(g[f"{tag}_log2t"]), level_major=level_major, sample_major=sm)
    flat = out.permute(1, 0, 2).reshape(n, L * F) if level_major else out
    assert_close(cpu(flat), g[f"{tag}_out"], rtol=1e-5, atol_scale=1e-6, what=tag)
    gout = dev(g[f"{tag}_gout"])
    gbuf = gout.view(n, L, F).permute(1, 0, 2).contiguous() if level_major else gout
    (gt,) = torch.autograd.grad(out, table, gbuf)
    assert_close(cpu(gt), g[f"{tag}_gtable"], rtol=1e-4, atol_scale=1e-5, what=tag + " grad")


@pytest.mark.parametrize("F", [8, 4, 2, 1])
@pytest.mark.parametrize("coherent", [True, False])
def test_hash_encode_every_width_vs_oracle(F, coherent):
    """All four feature widths of the kernels (the goldens hold F = 1, 2, 4) against the oracle, on clustered
    positions (the on-chip dedup path of the scatter) and on uniformly random ones (table pressure, flushes)."""
    from neuradar_amd import ops
    from oracle import hashgrid

    torch.manual_seed(F)
    L, log2t, n = 3, 12, 20_000
    sc = hashgrid.level_scalings(L, 16, 512)
    table = hashgrid.init_table(L, log2t, F, scale=1.0)
    if coherent:  # a few tight clusters: many samples per cell
        centres = torch.rand(40, 3)
        x = (centres[torch.randint(0, 40, (n,))] + 0.002 * torch.randn(n, 3)).clamp(0, 1)
    else:
        x = torch.rand(n, 3)
    x[:7] = torch.tensor([0.0, 0.25, 0.5])  # exact grid planes
    gout = torch.randn(n, L * F)
    tr = table.clone().requires_grad_(True)
    ref = hashgrid.encode(x, tr, sc, 2**log2t)
    (gref,) = torch.autograd.grad(ref, tr, gout)
    td = dev(table).requires_grad_(True)
    out = ops.hash_encode(dev(x), td, dev(sc), log2t)
    (gt,) = torch.autograd.grad(out, td, dev(gout))
    assert_close(cpu(out), ref.detach(), rtol=1e-5, atol_scale=1e-6, what=f"F={F} fwd")
    assert_close(cpu(gt), gref, rtol=1e-4, atol_scale=1e-5, what=f"F={F} table grad")


def test_hash_encode_module_is_dropin_and_linear_in_table():
    """HashEncoding keeps the reference's surface (scalings buffer, hash_table parameter, out dim) and
    the encoding is linear in the table (size-independent property, checked at a large n)."""
    from neuradar_amd.encodings import HashEncoding

    torch.manual_seed(0)
    enc = HashEncoding(num_levels=8, min_res=32, max_res=8192, log2_hashmap_size=19, features_per_level=4).to(DEV)
    assert enc.get_out_dim() == 32 and enc.scalings[-1].item() == 8191.0
    assert enc.hash_table.shape == (8 * 2**19, 4) and isinstance(enc.hash_table, torch.nn.Parameter)
    x = torch.rand(200_000, 3, device=DEV)
    y1 = enc(x)
    with torch.no_grad():
        t0 = enc.hash_table.clone()
        enc.hash_table.mul_(-2.5)
    y2 = enc(x)
    torch.testing.assert_close(y2, -2.5 * y1, rtol=1e-5, atol=1e-9)
    with torch.no_grad():
        enc.hash_table.copy_(torch.ones_like(t0))
    y3 = enc(x)  # partition of unity: trilinear weights sum to one
    torch.testing.assert_close(y3, torch.ones_like(y3), rtol=1e-5, atol=1e-6)
    assert enc(torch.empty(0, 3, device=DEV)).shape == (0, 32)  # empty input


@pytest.mark.parametrize("cfg", [("prop", 6, 1, 20, 16, 512, 4096 * 128), ("l16f2", 16, 2, 19, 16, 1024, 4096 * 32),
                                 ("l8f4", 8, 4, 22, 32, 8192, 16384 * 32)])
def test_scatter_checksums_and_linearity_at_baseline_sizes(cfg):
    """Size-independent properties of the table gradient at BASELINE.json's table and batch sizes, where the
    oracle is too slow to be the checker: (1) checksum of checksums -- the trilinear weights of a sample sum to
    one, so per level and feature sum_rows grad_table == sum_samples rescale * grad_out; (2) linearity in
    grad_out; (3) every touched row is one of the 8 corners of some sample (nothing lands elsewhere: rows the
    forward never reads stay exactly zero).  Clustered positions, so the on-chip dedup path is the one running."""
    from neuradar_amd import ops
    from oracle import hashgrid

    tag, L, F, log2t, rmin, rmax, n = cfg
    torch.manual_seed(L * F)
    sc = dev(hashgrid.level_scalings(L, rmin, rmax))
    centres = torch.rand(2000, 3, device=DEV)
    x = (centres[torch.randint(0, 2000, (n,), device=DEV)] + 0.01 * torch.randn(n, 3, device=DEV)).clamp(0, 1)
    std = 0.002 * torch.rand(n, device=DEV)
    table = (torch.rand(L << log2t, F, device=DEV) * 2e-3 - 1e-3).requires_grad_(True)

    def grad_of(gout):
        out = ops.hash_encode(x, table, sc, log2t, std=std, level_major=True)  # [L, n, F]
        (gt,) = torch.autograd.grad(out, table, gout)
        return gt

    g1, g2 = torch.randn(L, n, F, device=DEV), torch.randn(L, n, F, device=DEV)
    gt1, gt2 = grad_of(g1), grad_of(g2)
    resc = 1.0 / torch.clamp(2.0 * sc[:, None] * std[None, :], min=1.0)  # neurad_encoding.py:309-316
    want = (g1.double() * resc[:, :, None].double()).sum(1)  # [L, F]
    got = gt1.view(L, 1 << log2t, F).double().sum(1)
    scale = (g1.double().abs() * resc[:, :, None].double()).sum(1)  # magnitude the float sums ran over
    assert float(((got - want).abs() / scale).max()) < 1e-5, f"{tag}: checksum of checksums"
    gt12 = grad_of(2.0 * g1 - 0.5 * g2)
    ref = 2.0 * gt1 - 0.5 * gt2
    assert_close(cpu(gt12), cpu(ref), rtol=1e-4, atol_scale=1e-5, what=f"{tag}: linearity in grad_out")
    # rows written == rows the forward reads: perturbing ONLY untouched rows must not change the encoding
    touched = (gt1 != 0).any(1) | (gt2 != 0).any(1)
    assert 0 < int(touched.sum()) < touched.numel()
    with torch.no_grad():
        out0 = ops.hash_encode(x, table, sc, log2t, std=std, level_major=True)
        t2 = table.detach().clone()
        t2[~touched] += 1.0
        out1 = ops.hash_encode(x, t2, sc, log2t, std=std, level_major=True)
    assert torch.equal(out0, out1), f"{tag}: the forward read a row the backward never wrote"


def test_empty_and_single_inputs_through_the_abi():
    """Empty batches are legal everywhere on the path (a sensor without rays in a mixed batch, an actor nobody hits)
    and return empty results without touching memory; a single ray / single sample works too."""
    from neuradar_amd import ops
    from neuradar_amd.sensors import Cameras, Lidars
    from oracle import hashgrid

    e = lambda *shape: torch.empty(*shape, device=DEV)  # noqa: E731
    sc = dev(hashgrid.level_scalings(4, 16, 128))
    table = (torch.rand(4 << 12, 2, device=DEV) - 0.5).requires_grad_(True)
    out = ops.hash_encode(e(0, 3), table, sc, 12)
    assert out.shape == (0, 8)
    (gt,) = torch.autograd.grad(out.sum(), table)
    assert float(gt.abs().max()) == 0.0
    x01, std = ops.contract_gaussians(e(0, 3), e(0, 3), e(0, 1), e(0, 33), 100.0)
    assert x01.shape[0] == 0 and std.shape[0] == 0
    sp, eu = ops.power_bins(e(0), e(0), 16)
    assert sp.shape == (0, 17) and eu.shape == (0, 17)
    assert ops.weights_from_density(e(0, 16), e(0, 17)).shape == (0, 16)
    sp2, eu2 = ops.pdf_resample(e(0, 16), e(0, 17), e(0), e(0), 8)
    assert sp2.shape == (0, 9)
    w, acc, f, dep = ops.composite(e(0, 8), e(0, 8, 4), e(0, 9))
    assert w.shape == (0, 8) and f.shape == (0, 4) and dep.shape[0] == 0
    assert ops.sh4(e(0, 3)).shape == (0, 16)
    cams = Cameras(torch.eye(3, 4, device=DEV)[None], *(torch.tensor([v], device=DEV) for v in (500.0, 500.0, 320.0, 240.0, 480.0, 0.0)))
    rb = cams.generate_rays(torch.empty(0, 3, dtype=torch.int64, device=DEV))
    assert rb.origins.shape == (0, 3)
    lid = Lidars(torch.eye(3, 4, device=DEV)[None], torch.zeros(1, device=DEV))
    assert lid.generate_rays(torch.empty(0, dtype=torch.int64, device=DEV), e(0, 5)).origins.shape == (0, 3)
    # one ray, one sample
    one = ops.hash_encode(torch.full((1, 3), 0.3, device=DEV), table, sc, 12)
    ref = hashgrid.encode(torch.full((1, 3), 0.3), cpu(table.detach()), cpu(sc), 1 << 12)
    assert_close(cpu(one.detach()), ref, rtol=1e-5, atol_scale=1e-6, what="single sample")
    w1, acc1, _, _ = ops.composite(torch.full((1, 1), 0.25, device=DEV), torch.ones(1, 1, 3, device=DEV),
                                   torch.tensor([[1.0, 2.0]], device=DEV))
    assert abs(float(w1.sum()) - 1.0) < 1e-6 and abs(float(acc1.reshape(-1)[0]) - 0.25) < 1e-6


def test_sampling_chain_properties_at_baseline_sizes():
    """16 384 rays (configs[2] per-GPU size): power bins and both PDF resampling rounds are sorted and stay
    inside [near, far]; get_weights and the alpha compositing conserve probability mass (w >= 0, sum <= 1;
    after the sky fix-up the weights of a ray sum to one)."""
    from neuradar_amd import ops

    torch.manual_seed(3)
    B = 16384
    nears, fars = torch.zeros(B, device=DEV), torch.full((B,), 20000.0, device=DEV)
    sp, eu = ops.power_bins(nears, fars, 128, t_rand=torch.rand(B, 129, device=DEV))
    for S_in, S_out in ((128, 64), (64, 32)):
        assert bool((eu[:, 1:] >= eu[:, :-1]).all()) and bool((sp[:, 1:] >= sp[:, :-1]).all())
        assert float(eu.min()) >= 0.0 and float(eu.max()) <= 20000.0 * (1 + 1e-6)
        dens = torch.exp(2.0 * torch.randn(B, S_in, device=DEV))
        w = ops.weights_from_density(dens, eu)
        assert bool((w >= 0).all()) and float(w.sum(1).max()) <= 1.0 + 1e-5
        sp, eu = ops.pdf_resample(w, sp, nears, fars, S_out, jitter=torch.rand(B, device=DEV))
    assert bool((eu[:, 1:] >= eu[:, :-1]).all())
    alpha = torch.rand(B, 32, device=DEV) ** 4
    feat = torch.randn(B, 32, 32, device=DEV)
    w, acc, feats, depth = ops.composite(alpha, feat, eu)
    assert float(w.min()) >= -1e-6  # the fix-up `w_last += 1 - acc` may round a hair below zero, as in the reference
    # sky fix-up (neuradar.py:505-508): the last weight takes the remaining 1 - accumulation
    torch.testing.assert_close(w.sum(1), torch.ones(B, device=DEV), rtol=0, atol=4e-6)
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-6
    torch.testing.assert_close(feats, (w[:, :, None] * feat).sum(1), rtol=1e-4, atol=1e-5)
    assert bool(torch.isfinite(depth).all()) and float(depth.min()) >= 0.0


# ------------------------------------------------------------------------------------------------ a6 a7 a9
def test_contraction_and_rescaled_grid_features():
    from neuradar_amd import ops
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings

    g = load_golden("gaussian_contraction")
    e = g["edges"]
    x01, std01 = ops.contract_gaussians(dev(g["origins"]), dev(g["directions"]), dev(g["pixel_area"]), dev(e), 100.0)
    B, S = e.shape[0], e.shape[1] - 1
    assert_close(cpu(x01).view(B, S, 3), g["mean01"], rtol=1e-5, atol_scale=1e-6, what="mean01")
    assert_close(cpu(std01).view(B, S, 1), g["std01"], rtol=2e-5, atol_scale=1e-6, what="std01")
    x01a, std01a = ops.contract_gaussians(dev(g["origins"]), dev(g["directions"]), dev(g["pixel_area"]), dev(e), 10.0)
    assert_close(cpu(x01a).view(B, S, 3), g["mean01_actor"], rtol=1e-5, atol_scale=1e-6)
    assert_close(cpu(std01a).view(B, S, 1), g["std01_actor"], rtol=2e-5, atol_scale=1e-6)
    hg = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=int(g["log2t"]))).setup(
        dynamic_actors=None, static_scale=100.0).to(DEV)
    with torch.no_grad():
        hg.static_grid.hash_table.copy_(g["table"])
    feats, _ = hg(samples_from_edges(g, e))
    assert_close(cpu(feats).view(B, S, -1), g["grid_features"], rtol=1e-4, atol_scale=1e-5, what="grid features")


# ------------------------------------------------------------------------------------------------ a15 a11
@pytest.mark.parametrize("tag,hidden", [("field_neurad", 32), ("field_l16f2w64", 64)])
def test_field_fwd_bwd_vs_reference_golden(tag, hidden):
    from neuradar_amd.field_heads import FieldHeadNames
    from neuradar_amd.neurad_encoding import StaticSettings

    g = load_golden(tag)
    static = None
    if hidden == 64:
        static = StaticSettings(hashgrid_dim=2, num_levels=16, base_res=16, max_res=1024, log2_hashmap_size=int(g["log2t"]))
    fld = make_field(g, hidden=hidden, static=static)
    rs = samples_from_edges(g, g["edges"])
    out = fld(rs)
    assert_close(cpu(out[FieldHeadNames.FEATURE]), g["feature"], rtol=1e-4, atol_scale=1e-5, what="feature")
    assert_close(cpu(out[FieldHeadNames.SDF]), g["sdf"], rtol=1e-4, atol_scale=1e-5, what="sdf")
    assert_close(cpu(out[FieldHeadNames.ALPHA]), g["alpha"], rtol=1e-4, atol_scale=1e-5, what="alpha")
    loss = (out[FieldHeadNames.FEATURE] * dev(g["g_feature"])).sum() + (out[FieldHeadNames.ALPHA] * dev(g["g_alpha"])).sum()
    named = dict(fld.named_parameters())
    keys = ["hashgrid.static_grid.hash_table"]
    gold = ["grad_table"]
    for i in range(2):
        keys += [f"mlp_geo.layers.{i}.weight", f"mlp_geo.layers.{i}.bias"]; gold += [f"grad_geo_w{i}", f"grad_geo_b{i}"]
    for i in range(3):
        keys += [f"mlp_feature.layers.{i}.weight", f"mlp_feature.layers.{i}.bias"]; gold += [f"grad_feat_w{i}", f"grad_feat_b{i}"]
    keys.append("sdf_to_density.beta"); gold.append("grad_beta")
    grads = torch.autograd.grad(loss, [named[k] for k in keys])
    for k, gk, gr in zip(keys, gold, grads):
        assert_close(cpu(gr), g[gk], rtol=2e-4, atol_scale=2e-5, what="grad " + k)


def test_proposal_density_fwd_bwd_vs_reference_golden():
    g = load_golden("field_neurad")
    pf = make_prop(g)
    rs = samples_from_edges(g, g["edges"])
    dens, _ = pf.get_density(rs)
    assert_close(cpu(dens), g["prop_density"], rtol=1e-4, atol_scale=1e-5)
    gt, gw = torch.autograd.grad((dens * dev(g["prop_g_density"])).sum(),
                                 [pf.hashgrid.static_grid.hash_table, pf.density_decoder.weight])
    assert_close(cpu(gt), g["prop_grad_table"], rtol=2e-4, atol_scale=2e-5)
    assert_close(cpu(gw), g["prop_grad_decoder"], rtol=2e-4, atol_scale=2e-5)


def test_sh_and_generic_mlp_vs_reference_golden_and_oracle_backward():
    from neuradar_amd.encodings import SHEncoding
    from neuradar_amd.mlp import MLP
    from oracle import field as ofield

    g = load_golden("sh_mlp")
    assert_close(cpu(SHEncoding(levels=4)(dev(g["dirs"]))), g["sh_raw"], rtol=1e-5, atol_scale=1e-6)
    m = MLP(in_dim=48, num_layers=3, layer_width=32, out_dim=2).to(DEV)  # lidar decoder shape (K7)
    with torch.no_grad():
        for i, l in enumerate(m.layers):
            l.weight.copy_(g[f"mlp_w{i}"]); l.bias.copy_(g[f"mlp_b{i}"])
    x = dev(g["mlp_x"]).requires_grad_(True)
    y = m(x)
    assert_close(cpu(y), g["mlp_y"], rtol=1e-4, atol_scale=1e-5)
    # backward vs oracle autograd, ragged n (not a multiple of the 32-sample MFMA tile)
    torch.manual_seed(3)
    for (ind, width, outd, nl, n) in [(48, 32, 2, 3, 77), (32, 64, 33, 2, 301), (64, 64, 64, 3, 1000), (7, 16, 5, 2, 33)]:
        m = MLP(in_dim=ind, num_layers=nl, layer_width=width, out_dim=outd).to(DEV)
        x = torch.randn(n, ind, device=DEV, requires_grad=True)
        gy = torch.randn(n, outd, device=DEV)
        y = m(x)
        params = list(m.parameters())
        grads = torch.autograd.grad(y, [x] + params, gy)
        xo = cpu(x).requires_grad_(True)
        layers = [(cpu(l.weight).requires_grad_(True), cpu(l.bias).requires_grad_(True)) for l in m.layers]
        yo = ofield.mlp(xo, layers)
        assert_close(cpu(y), yo.detach(), rtol=1e-4, atol_scale=1e-5, what=f"mlp {ind}-{width}-{outd}")
        flat = [xo] + [t for wb in layers for t in wb]
        gos = torch.autograd.grad(yo, flat, cpu(gy))
        got = [grads[0]] + [grads[1 + 2 * i + j] for i in range(nl) for j in range(2)]
        for a, b in zip(got, gos):
            assert_close(cpu(a), b, rtol=2e-4, atol_scale=2e-5, what=f"mlp grad {ind}-{width}-{outd}")


# ------------------------------------------------------------------------------------------------ a5 a12 a13
def test_sampler_kernels_vs_reference_golden():
    from neuradar_amd import ops
    from oracle import sampler as osampler

    g = load_golden("sampler")
    nears, fars = dev(g["nears"]), dev(g["fars"])
    sp, eu = ops.power_bins(nears, fars, 128)
    assert_close(cpu(sp), g["power_eval_spacing"], rtol=1e-6, atol_scale=1e-7)
    assert_close(cpu(eu), g["power_eval_euclid"], rtol=1e-4, atol_scale=1e-6)
    sp_t, eu_t = ops.power_bins(nears, fars, 128, dev(g["power_train_t_rand"]))
    assert_close(cpu(sp_t), g["power_train_spacing"], rtol=1e-5, atol_scale=1e-6)
    assert_close(cpu(eu_t), g["power_train_euclid"], rtol=1e-4, atol_scale=1e-6)
    dens = dev(g["gw_density"]).requires_grad_(True)
    w = ops.weights_from_density(dens, dev(g["power_train_euclid"]))
    assert_close(cpu(w), g["gw_weights"], rtol=1e-4, atol_scale=1e-6)
    # backward vs oracle autograd
    gw = torch.randn_like(w)
    (gd,) = torch.autograd.grad(w, dens, gw)
    d_o = g["gw_density"].clone().requires_grad_(True)
    e = g["power_train_euclid"]
    w_o = osampler.weights_from_density(e[:, 1:] - e[:, :-1], d_o)
    (gd_o,) = torch.autograd.grad(w_o, d_o, cpu(gw))
    assert_close(cpu(gd), gd_o, rtol=2e-4, atol_scale=2e-5, what="weights bwd")
    for S in (64, 32, 40):  # other widths incl. a ragged one
        d2 = torch.rand(50, S, device=DEV).requires_grad_(True)
        e2 = torch.cumsum(torch.rand(50, S + 1, device=DEV), -1)
        w2 = ops.weights_from_density(d2, e2)
        w2o = osampler.weights_from_density(cpu(e2[:, 1:] - e2[:, :-1]), cpu(d2).detach())
        assert_close(cpu(w2), w2o, rtol=1e-4, atol_scale=1e-6, what=f"weights S={S}")
    sp_e, eu_e = ops.pdf_resample(dev(g["gw_weights"]), dev(g["power_train_spacing"]), nears, fars, 64)
    assert_close(cpu(sp_e), g["pdf_eval_spacing"], rtol=1e-4, atol_scale=1e-5)
    assert_close(cpu(eu_e), g["pdf_eval_euclid"], rtol=1e-3, atol_scale=1e-5)  # euclid amplifies ds near s->1
    sp_j, eu_j = ops.pdf_resample(dev(g["gw_weights"]), dev(g["power_train_spacing"]), nears, fars, 64,
                                  dev(g["pdf_train_jitter"]))
    assert_close(cpu(sp_j), g["pdf_train_spacing"], rtol=1e-4, atol_scale=1e-5)
    assert_close(cpu(eu_j), g["pdf_train_euclid"], rtol=1e-3, atol_scale=1e-5)
    assert bool((sp_j[:, 1:] >= sp_j[:, :-1]).all())  # sortedness of the resampled edges


# ------------------------------------------------------------------------------------------------ a16-a18
def test_composite_fwd_bwd_vs_oracle():
    from neuradar_amd import ops
    from oracle import render as orender

    torch.manual_seed(5)
    for (B, S, C) in [(37, 32, 32), (8, 7, 48), (5, 64, 3)]:
        alpha = torch.rand(B, S, device=DEV) ** 3
        alpha[0] = 0.0
        alpha[1, 3] = 1.0  # full occlusion mid-ray: exercises the division-free backward
        alpha.requires_grad_(True)
        feat = torch.randn(B, S, C, device=DEV, requires_grad=True)
        eu = torch.cumsum(torch.rand(B, S + 1, device=DEV) * 3, -1)
        w, acc, fo, d = ops.composite(alpha, feat, eu)
        a_o, f_o = cpu(alpha).requires_grad_(True), cpu(feat).requires_grad_(True)
        ref = orender.composite(a_o[..., None], f_o, cpu(eu)[:, :-1], cpu(eu)[:, 1:])
        assert_close(cpu(w), ref["weights"].detach(), rtol=1e-4, atol_scale=1e-6, what="weights")
        assert_close(cpu(acc), ref["accumulation"][:, 0].detach(), rtol=1e-4, atol_scale=1e-6)
        assert_close(cpu(fo), ref["features"].detach(), rtol=1e-4, atol_scale=1e-5)
        assert_close(cpu(d), ref["depth"][:, 0].detach(), rtol=1e-4, atol_scale=1e-5)
        gw, ga, gf, gd = torch.randn_like(w), torch.randn_like(acc), torch.randn_like(fo), torch.randn_like(d)
        gw[:, -1] = 0
        loss = (w * gw).sum() + (acc * ga).sum() + (fo * gf).sum() + (d * gd).sum()
        g_alpha, g_feat = torch.autograd.grad(loss, [alpha, feat])
        loss_o = ((ref["weights"] * cpu(gw)).sum() + (ref["accumulation"][:, 0] * cpu(ga)).sum()
                  + (ref["features"] * cpu(gf)).sum() + (ref["depth"][:, 0] * cpu(gd)).sum())
        go_alpha, go_feat = torch.autograd.grad(loss_o, [a_o, f_o])
        assert_close(cpu(g_feat), go_feat, rtol=1e-4, atol_scale=1e-5, what="g_feature")
        mask = cpu(alpha) < 1.0  # torch's cumprod backward is itself ill-defined exactly at alpha == 1
        assert_close(cpu(g_alpha)[mask], go_alpha[mask], rtol=2e-4, atol_scale=2e-5, what="g_alpha")
    # size-independent property at full size: weights are a partition of unity after the sky fix-up
    alpha = torch.rand(16384, 32, device=DEV)
    w, acc, fo, _ = ops.composite(alpha, torch.ones(16384, 32, 32, device=DEV), torch.zeros(16384, 33, device=DEV))
    torch.testing.assert_close(w.sum(-1), torch.ones(16384, device=DEV), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fo, torch.ones_like(fo), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------ a4 a14 end to end
def build_hot_path(g):
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.step import HotPathConfig, NeuRadarHotPath

    cfg = HotPathConfig(field=NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=int(g["main_log2t"])))))
    cfg.proposal_field_1.grid.static.log2_hashmap_size = int(g["prop_log2t"])
    cfg.proposal_field_2.grid.static.log2_hashmap_size = int(g["prop_log2t"])
    model = NeuRadarHotPath(cfg).to(DEV)
    src = make_field(g, prefix="main_", log2t_key="main_log2t")
    model.field.load_state_dict(src.state_dict())
    model.proposal_fields[1].load_state_dict(make_prop(g, prefix="prop1_", log2t_key="prop_log2t").state_dict())
    return model


def test_pipeline_end_to_end_vs_reference_golden():
    from neuradar_amd.rays import RayBundle

    g = load_golden("pipeline")
    model = build_hot_path(g).train()
    bundle = RayBundle(dev(g["origins"]), dev(g["directions"]), dev(g["pixel_area"]), fars=dev(g["fars"]))
    out = model.get_nff_outputs(bundle, t_rand=dev(g["t_rand"]), jitters=(dev(g["jitter1"]), dev(g["jitter2"])))
    rsl = out["ray_samples_list"]
    for i in (0, 1):
        assert_close(cpu(rsl[i].spacing), g[f"prop_spacing_{i}"], rtol=1e-4, atol_scale=1e-5, what=f"prop spacing {i}")
        assert_close(cpu(rsl[i].euclid), g[f"prop_euclid_{i}"], rtol=1e-3, atol_scale=1e-5, what=f"prop euclid {i}")
        assert_close(cpu(out["weights_list"][i][..., 0]), g[f"prop_weights_{i}"], rtol=1e-3, atol_scale=1e-4)
        assert_close(cpu(out[f"prop_depth_{i}"]), g[f"prop_depth_{i}"], rtol=1e-3, atol_scale=1e-4)
    assert_close(cpu(out["ray_samples"].spacing), g["final_spacing"], rtol=1e-4, atol_scale=1e-5, what="final spacing")
    assert_close(cpu(out["ray_samples"].euclid), g["final_euclid"], rtol=1e-3, atol_scale=1e-5, what="final euclid")
    # north-star bar: rendered features / depth within 1e-4 relative (of the tensor's scale)
    assert_close(cpu(out["weights"][..., 0]), g["weights"], rtol=1e-4, atol_scale=1e-4, what="weights")
    assert_close(cpu(out["accumulation"]), g["accumulation"], rtol=1e-4, atol_scale=1e-4, what="accumulation")
    assert_close(cpu(out["features"]), g["features"], rtol=1e-4, atol_scale=1e-4, what="features")
    assert_close(cpu(out["depth"]), g["depth"], rtol=1e-4, atol_scale=1e-4, what="depth")
    loss = model.bench_loss(out, dev(g["target_features"]), dev(g["target_depth"]))
    assert_close(cpu(loss), g["loss"], rtol=1e-4, atol_scale=1e-5, what="loss")
    names = {
        "main_hashgrid_static_grid_hash_table": model.field.hashgrid.static_grid.hash_table,
        "main_sdf_to_density_beta": model.field.sdf_to_density.beta,
        "prop1_hashgrid_static_grid_hash_table": model.proposal_fields[1].hashgrid.static_grid.hash_table,
        "prop1_density_decoder_weight": model.proposal_fields[1].density_decoder.weight,
    }
    for i, l in enumerate(model.field.mlp_geo.layers):
        names[f"main_mlp_geo_layers_{i}_weight"], names[f"main_mlp_geo_layers_{i}_bias"] = l.weight, l.bias
    for i, l in enumerate(model.field.mlp_feature.layers):
        names[f"main_mlp_feature_layers_{i}_weight"], names[f"main_mlp_feature_layers_{i}_bias"] = l.weight, l.bias
    # the reference quirk: proposal_fields[0] never runs, so it must have no gradient
    p0 = model.proposal_fields[0].hashgrid.static_grid.hash_table
    grads = torch.autograd.grad(loss, list(names.values()) + [p0], allow_unused=True)
    assert grads[-1] is None
    for (n, _), gr in zip(names.items(), grads):
        assert_close(cpu(gr), g["grad_" + n], rtol=1e-3, atol_scale=1e-4, what="grad " + n)


def test_regularisers_product_vs_reference_golden():
    """Both forms of the regularisers behind torch.autograd: the op-by-op torch restatement (tight) and
    the one-launch HIP kernels the modular path uses (kernel tolerances, see test_loss_kernels_*)."""
    from neuradar_amd import losses

    g = load_golden("losses")
    cs = [dev(g[f"c{i}"]) for i in range(3)]
    for inter_fn, dist_fn, tol1 in ((losses.zipnerf_interlevel_loss_torch, losses.distortion_loss_torch, 1e-3),
                                    (losses.zipnerf_interlevel_loss, losses.distortion_loss, 1.001e-3)):
        ws = [dev(g[f"w{i}"]).requires_grad_(True) for i in range(3)]
        inter = inter_fn(cs, ws)
        dist = dist_fn(cs[-1], ws[-1])
        assert_close(cpu(inter), g["interlevel"], rtol=1e-4, atol_scale=1e-6)
        assert_close(cpu(dist), g["distortion"], rtol=1e-4, atol_scale=1e-6)
        gi = torch.autograd.grad(2.0 * inter, ws[:2])  # upstream factor 2 exercises backward's scaling
        assert_close(cpu(gi[0]) / 2, g["g_inter_w0"], rtol=1e-3, atol_scale=1e-4 if tol1 > 1e-3 else 1e-5)
        assert_close(cpu(gi[1]) / 2, g["g_inter_w1"], rtol=tol1, atol_scale=1e-4 if tol1 > 1e-3 else 1e-5)
        (gd,) = torch.autograd.grad(dist, [ws[2]])
        assert_close(cpu(gd), g["g_dist_w2"], rtol=1e-4, atol_scale=1e-5)


# ------------------------------------------------------------------------------------------------ a1 a2 a3
def test_raygen_vs_reference_golden():
    from neuradar_amd.sensors import Cameras, Lidars, Radars

    g = load_golden("raygen")
    cams = Cameras(dev(g["cam_c2w"]), dev(g["cam_fx"]), dev(g["cam_fy"]), dev(g["cam_cx"]), dev(g["cam_cy"]),
                   dev(g["cam_heights"]), dev(g["cam_times_in"]), dev(g["cam_vel"]), dev(g["cam_rs_offsets"]))
    b = cams.generate_rays(dev(g["cam_ray_indices"]))
    assert_close(cpu(b.origins), g["cam_origins"], rtol=1e-5, atol_scale=1e-6, what="cam origins")
    assert_close(cpu(b.directions), g["cam_directions"], rtol=1e-5, atol_scale=1e-6, what="cam directions")
    assert_close(cpu(b.pixel_area), g["cam_pixel_area"], rtol=1e-3, atol_scale=1e-4, what="cam pixel_area")
    assert_close(cpu(b.times), g["cam_times"], rtol=1e-6, atol_scale=1e-7, what="cam times")
    assert_close(cpu(b.metadata["directions_norm"]), g["cam_directions_norm"], rtol=1e-5, atol_scale=1e-6)
    cams2 = Cameras(dev(g["cam_c2w"]), dev(g["cam_fx"]), dev(g["cam_fy"]), dev(g["cam_cx"]), dev(g["cam_cy"]),
                    dev(g["cam_heights"]), dev(g["cam_times_in"]))
    b2 = cams2.generate_rays(dev(g["cam_ray_indices"]))
    assert_close(cpu(b2.origins), g["cam_nors_origins"], rtol=1e-6, atol_scale=1e-7)
    assert_close(cpu(b2.times), g["cam_nors_times"], rtol=1e-6, atol_scale=1e-7)

    # lens undistortion + fisheye (the ZOD camera model), and undistorted pinhole
    for ctype, key in ((1, "fe"), (0, "pd")):
        cams3 = Cameras(dev(g["cam_c2w"]), dev(g["cam_fx"]), dev(g["cam_fy"]), dev(g["cam_cx"]), dev(g["cam_cy"]),
                        dev(g["cam_heights"]), dev(g["cam_times_in"]), distortion_params=dev(g["cam_dist"]),
                        camera_type=torch.full((5,), ctype, dtype=torch.int32, device=DEV))
        b3 = cams3.generate_rays(dev(g["cam_ray_indices"]))
        assert_close(cpu(b3.directions), g[f"cam_{key}_directions"], rtol=1e-5, atol_scale=2e-6, what=f"{key} directions")
        assert_close(cpu(b3.pixel_area), g[f"cam_{key}_pixel_area"], rtol=2e-3, atol_scale=1e-4, what=f"{key} pixel_area")

    lid = Lidars(dev(g["lid_l2w"]), dev(g["lid_times_in"]), dev(g["lid_vel"]))
    lb = lid.generate_rays(dev(g["lid_indices"]), dev(g["lid_points"]))
    assert_close(cpu(lb.origins), g["lid_origins"], rtol=1e-5, atol_scale=1e-6, what="lidar origins")
    assert_close(cpu(lb.directions), g["lid_directions"], rtol=1e-4, atol_scale=2e-5, what="lidar directions")
    assert_close(cpu(lb.metadata["directions_norm"]), g["lid_directions_norm"], rtol=1e-5, atol_scale=1e-6)
    assert_close(cpu(lb.times), g["lid_times"], rtol=1e-6, atol_scale=1e-7)
    assert_close(cpu(lb.pixel_area), g["lid_pixel_area"], rtol=1e-6, atol_scale=1e-7)
    assert torch.equal(cpu(lb.metadata["did_return"]), g["lid_did_return"])

    rad = Radars(dev(g["rad_r2w"]), dev(g["rad_times_in"]), 0.015, 0.015, -0.80, 0.80, -0.08, 0.4)
    assert rad.grid_shape() == (107, 33)  # ZOD: 3 531 rays per scan
    rb = rad.generate_rays(dev(g["rad_scans"]))
    assert_close(cpu(rb.metadata["directions_spher"]), g["rad_directions_spher"], rtol=1e-6, atol_scale=1e-7, what="az/el grid")
    assert torch.equal(cpu(rb.origins), g["rad_origins"])
    # (R d + t) - t cancellation noise of the reference is ~eps*|t| = 5e-6 absolute on a unit vector
    assert_close(cpu(rb.directions), g["rad_directions"], rtol=1e-4, atol_scale=2e-5, what="radar directions")
    assert_close(cpu(rb.pixel_area), g["rad_pixel_area"], rtol=1e-6, atol_scale=1e-7)
    assert torch.equal(cpu(rb.times).reshape(-1), g["rad_times"].reshape(-1))
    assert torch.equal(cpu(rb.camera_indices[:, 0]), g["rad_scan_of_ray"])
    vod = Radars(dev(g["rad_r2w"]), dev(g["rad_times_in"]), 0.02, 0.02, -1.0, 1.0, -0.39, 0.49)
    assert vod.grid_shape()[0] * vod.grid_shape()[1] == 4545
    vb = vod.generate_rays(torch.tensor([1], device=DEV))
    assert_close(cpu(vb.metadata["directions_spher"]), g["rad_vod_directions_spher"], rtol=1e-6, atol_scale=1e-7)
    assert_close(cpu(vb.directions), g["rad_vod_directions"], rtol=1e-4, atol_scale=2e-5)


# ------------------------------------------------------------------------------------------------ optimizer
def test_fused_adam_matches_torch_optim():
    from neuradar_amd import ops

    torch.manual_seed(1)
    for adamw, wd in ((False, 0.0), (True, 1e-2), (False, 1e-3)):
        p = torch.randn(100_003, device=DEV)
        ref = torch.nn.Parameter(p.clone())
        opt = (torch.optim.AdamW if adamw else torch.optim.Adam)([ref], lr=1e-2, eps=1e-15, weight_decay=wd)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for step in range(1, 6):
            g = torch.randn_like(p) * (step % 2)  # includes an all-zero gradient step (dense Adam still moves)
            ref.grad = g.clone()
            opt.step()
            gbuf = g.clone()
            ops.adam_step(p, gbuf, m, v, 1e-2, step, eps=1e-15, weight_decay=wd, adamw=adamw)
            assert float(gbuf.abs().max()) == 0.0  # grad zeroed in the same pass
        torch.testing.assert_close(p, ref.data, rtol=1e-5, atol=1e-7)


def test_adam_sparse_gradients_are_exact():
    """Hash-table gradients are sparse: the kernel leaves entries at Adam's fixed point (g = m = v = 0) alone and, with
    the caller's seen_grad bytes, does so without reading the moments.  Both variants give bit-identical parameters and
    moments, match torch.optim.Adam, and never-touched entries keep their initial bits -- across steps whose touched
    rows move and a step with no gradient at all (entries with momentum still move)."""
    from neuradar_amd import ops

    torch.manual_seed(9)
    n = 1 << 18
    p0 = torch.randn(n, device=DEV)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-2, eps=1e-15)
    runs = [dict(p=p0.clone(), m=torch.zeros(n, device=DEV), v=torch.zeros(n, device=DEV), seen=None),
            dict(p=p0.clone(), m=torch.zeros(n, device=DEV), v=torch.zeros(n, device=DEV),
                 seen=torch.zeros(n // 4, dtype=torch.uint8, device=DEV)),
            # nr_adam_step_marked: the bytes are set by whoever writes the gradient (here the test, in the step the scatter
            # kernel of the fused step): never-marked groups are skipped without reading their gradient
            dict(p=p0.clone(), m=torch.zeros(n, device=DEV), v=torch.zeros(n, device=DEV),
                 seen=torch.zeros(n // 4, dtype=torch.uint8, device=DEV), marked=True)]
    # the tables' layout in FlatAdam: the two moments as the halves of ONE array of [exp_avg x 4 | exp_avg_sq x 4] records
    for marked in (False, True):
        mv = torch.zeros(n // 4, 2, 4, device=DEV)
        runs.append(dict(p=p0.clone(), m=mv[:, 0, :], v=mv[:, 1, :], seen=torch.zeros(n // 4, dtype=torch.uint8, device=DEV), marked=marked))
    ever = torch.zeros(n, dtype=torch.bool, device=DEV)
    for step in range(1, 7):
        g = torch.zeros(n, device=DEV)
        if step != 4:
            rows = torch.randint(0, n // 8, (3000,), device=DEV) + (step % 3) * (n // 8)  # clustered, moving window
            g[rows] = torch.randn(rows.numel(), device=DEV)
        ever |= g != 0
        ref.grad = g.clone()
        opt.step()
        for r in runs:
            gbuf = g.clone()
            if r.get("marked"):
                r["seen"] |= (g != 0).view(-1, 4).any(1).to(torch.uint8)
            ops.adam_step(r["p"], gbuf, r["m"], r["v"], 1e-2, step, eps=1e-15, seen_grad=r["seen"], marked=bool(r.get("marked")))
            assert float(gbuf.abs().max()) == 0.0
    a_, b_, c_ = runs[:3]
    for r in runs[3:]:  # interleaved moments: bit-identical to the two-array launches
        assert torch.equal(a_["p"], r["p"]) and torch.equal(a_["m"], r["m"].reshape(-1)) and torch.equal(a_["v"], r["v"].reshape(-1))
    assert torch.equal(a_["p"], b_["p"]) and torch.equal(a_["m"], b_["m"]) and torch.equal(a_["v"], b_["v"])
    assert torch.equal(a_["p"], c_["p"]) and torch.equal(a_["m"], c_["m"]) and torch.equal(a_["v"], c_["v"])
    assert torch.equal(b_["seen"], c_["seen"])
    st = opt.state[ref]
    torch.testing.assert_close(a_["p"], ref.data, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(a_["m"], st["exp_avg"], rtol=1e-4, atol=1e-9)
    torch.testing.assert_close(a_["v"], st["exp_avg_sq"], rtol=1e-4, atol=1e-12)
    assert torch.equal(a_["p"][~ever], p0[~ever]) and float(a_["m"][~ever].abs().max()) == 0.0
    assert torch.equal(b_["seen"].bool(), ever.view(-1, 4).any(1))


@pytest.mark.parametrize("F,cells", [(4, 256), (4, 0), (2, 0), (1, 0), (8, 0)])
def test_marked_scatter_sets_the_bytes_of_every_group_it_adds_to(F, cells):
    """nr_hash_encode_bwd_marked: the same gradient as nr_hash_encode_bwd_tuned, and seen_grad byte (row * F + f) / 4 is set for
    every group that received a non-zero sum and for no group outside the cells the samples touch."""
    from neuradar_amd import _lib, ops

    torch.manual_seed(F + cells)
    L, log2T, n, S = 3, 12, 64 * 40, 8
    T = 1 << log2T
    x = torch.rand(n, 3, device=DEV)
    scal = torch.tensor([4.0, 9.0, 23.0], device=DEV)
    gout = torch.randn(L, n, F, device=DEV)
    gout[:, ::3] = 0.0  # rows without a gradient mark nothing
    lib, p = _lib.lib(), ops._p
    g_ref = torch.zeros(L * T, F, device=DEV)
    g_mk = torch.zeros_like(g_ref)
    seen = torch.zeros(L * T * F // 4, dtype=torch.uint8, device=DEV)
    _lib.check(lib.nr_hash_encode_bwd_tuned(p(x), None, p(scal), L, F, log2T, p(gout), F, n * F, p(g_ref), n, S, cells, ops._stream()), "tuned")
    _lib.check(lib.nr_hash_encode_bwd_marked(p(x), None, p(scal), L, F, log2T, p(gout), F, n * F, p(g_mk), n, S, cells, p(seen),
                                             ops._stream()), "marked")
    torch.testing.assert_close(g_mk, g_ref, rtol=1e-5, atol=1e-6)  # (float atomics: the order of the sums differs per launch)
    nonzero = (g_mk.view(-1, 4) != 0).any(1)
    assert bool((seen.bool() | ~nonzero).all()), "a group with a gradient was not marked"
    # marks only where a corner of a row WITH a gradient lands: the same scatter of ones
    ones = torch.zeros_like(g_ref)
    _lib.check(lib.nr_hash_encode_bwd_tuned(p(x), None, p(scal), L, F, log2T, p((gout != 0).float().contiguous()), F, n * F, p(ones), n,
                                            S, cells, ops._stream()), "tuned")
    touched = (ones.view(-1, 4) != 0).any(1)
    assert bool((touched | ~seen.bool()).all()), "a group no sample touches was marked"
    assert int(seen.sum()) > 0


# ------------------------------------------------------------------------------------------------ f-3 loss kernels
def test_loss_kernels_vs_reference_golden_and_oracle():
    from neuradar_amd import ops
    from oracle import losses as olosses

    g = load_golden("losses")
    c2, w2 = dev(g["c2"]), dev(g["w2"])  # final level: 31 samples, 32 edges
    # store them the way the step does: [B,33] edges / [B,32] weights incl. the (ignored) sky sample
    B = c2.shape[0]
    c_full = torch.cat([c2, torch.ones(B, 1, device=DEV)], 1).contiguous()
    w_full = torch.cat([w2, torch.full((B, 1), 0.123, device=DEV)], 1).contiguous()
    loss = torch.zeros(1024, device=DEV)  # NR_LOSS_SLOTS partial sums
    gd = ops.distortion_loss(c_full, w_full, 31, 1.0, loss)
    assert_close(cpu(loss).sum(), g["distortion"], rtol=1e-4, atol_scale=1e-6, what="distortion")
    assert_close(cpu(gd[:, :31]), g["g_dist_w2"], rtol=1e-4, atol_scale=1e-5, what="distortion grad")
    assert float(gd[:, 31].abs().max()) == 0.0
    total = 0.0
    for i, pulse in enumerate((0.03, 0.003)):
        loss.zero_()
        gi = ops.interlevel_loss(c_full, w_full, 31, dev(g[f"c{i}"]), dev(g[f"w{i}"]), pulse, 1.0, loss)
        # the blurred-histogram slopes are O(w/dc/pulse) ~ 1e5 with alternating signs; the kernel's running sums are
        # sequential like the reference's torch.cumsum (a tree scan differed by up to 3e-2 on the fine pulse)
        assert_close(cpu(gi), g[f"g_inter_w{i}"], rtol=1e-3, atol_scale=1e-4, what=f"interlevel grad {i}")
        total += float(loss.sum())
    assert abs(total - float(g["interlevel"])) <= 1e-4 * abs(float(g["interlevel"]))
    # supervision loss vs plain torch
    torch.manual_seed(0)
    f, tf = torch.randn(100, 48, device=DEV), torch.randn(100, 32, device=DEV)
    d, td = torch.rand(100, device=DEV) * 50, torch.rand(100, device=DEV) * 50
    loss.zero_()
    gf, gdp = ops.supervision_loss(f, tf, d, td, 5.0, 0.01, loss)
    fr, dr = f.clone().requires_grad_(True), d.clone().requires_grad_(True)
    ref = 5.0 * torch.mean((fr[:, :32] - tf) ** 2) + 0.01 * (dr - td).abs().mean()
    gfr, gdr = torch.autograd.grad(ref, [fr, dr])
    torch.testing.assert_close(loss.sum(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gf, gfr, rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(gdp, gdr, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("S,C", [(32, 32), (48, 32), (64, 16), (7, 3)])
def test_fused_render_matches_separate_kernels(S, C):
    """nr_render_train == nr_composite_fwd -> nr_supervision_loss -> nr_distortion_loss -> nr_composite_bwd."""
    from neuradar_amd import _lib, ops

    lib, p, st = _lib.lib(), ops._p, ops._stream
    torch.manual_seed(S * 100 + C)
    B = 257
    f32 = dict(device=DEV, dtype=torch.float32)
    alpha = torch.rand(B, S, **f32)
    alpha[::7] = alpha[::7] * 0.05  # some nearly empty rays: the sky fix-up carries most of the weight
    feature = torch.randn(B * S, C, **f32)
    sp = torch.sort(torch.rand(B, S + 1, **f32), dim=1).values
    eu = sp * 80.0 + 0.5
    tf, td = torch.rand(B, C, **f32), torch.rand(B, **f32) * 60
    mults = (1.3, 0.02, 0.07)

    def outs():
        return dict(w=torch.empty(B, S, **f32), acc=torch.empty(B, **f32), f=torch.empty(B, C, **f32), d=torch.empty(B, **f32),
                    ga=torch.empty(B, S, **f32), gf=torch.empty(B * S, C, **f32), loss=torch.zeros(_lib.NR_LOSS_SLOTS, **f32))

    a, b = outs(), outs()
    _lib.check(lib.nr_render_train(p(alpha), p(feature), p(eu), p(sp), p(tf), p(td), B, S, C, *mults, p(a["w"]), p(a["acc"]),
                                   p(a["f"]), p(a["d"]), p(a["ga"]), p(a["gf"]), p(a["loss"]), None, None, None, st()), "render_train")
    g_f, g_d, g_w = torch.empty(B, C, **f32), torch.empty(B, **f32), torch.empty(B, S, **f32)
    _lib.check(lib.nr_composite_fwd(p(alpha), p(feature), p(eu), B, S, C, p(b["w"]), p(b["acc"]), p(b["f"]), p(b["d"]), st()), "fwd")
    _lib.check(lib.nr_supervision_loss(p(b["f"]), C, p(tf), C, p(b["d"]), p(td), B, mults[0], mults[1], p(g_f), p(g_d),
                                       p(b["loss"]), st()), "sup")
    _lib.check(lib.nr_distortion_loss(p(sp), S + 1, p(b["w"]), S, S - 1, B, mults[2], p(g_w), p(b["loss"]), st()), "dist")
    _lib.check(lib.nr_composite_bwd(p(alpha), p(feature), p(eu), p(b["w"]), p(g_f), p(g_d), None, p(g_w), B, S, C, p(b["ga"]),
                                    p(b["gf"]), st()), "bwd")
    for k in ("w", "acc", "f", "d", "ga", "gf"):
        assert_close(cpu(a[k]), cpu(b[k]), rtol=2e-5, atol_scale=2e-6, what=k)
    assert_close(cpu(a["loss"].sum()), cpu(b["loss"].sum()), rtol=1e-5, atol_scale=1e-6, what="loss")


@pytest.mark.parametrize("Sp", [64, 128, 200])
def test_fused_interlevel_to_density_matches_separate_kernels(Sp):
    from neuradar_amd import _lib, ops

    lib, p, st = _lib.lib(), ops._p, ops._stream
    torch.manual_seed(Sp)
    B, S = 130, 32
    f32 = dict(device=DEV, dtype=torch.float32)
    c = torch.sort(torch.rand(B, S + 1, **f32), dim=1).values
    w = torch.softmax(torch.randn(B, S, **f32), dim=1) * 0.9
    cp = torch.sort(torch.rand(B, Sp + 1, **f32), dim=1).values
    eup = cp * 70 + 0.1
    dens = torch.rand(B, Sp, **f32) * 0.3
    wp = torch.empty(B, Sp, **f32)
    _lib.check(lib.nr_weights_from_density_fwd(p(dens), p(eup), B, Sp, p(wp), st()), "w")
    g_w, g_d1, g_d2 = torch.empty(B, Sp, **f32), torch.empty(B, Sp, **f32), torch.empty(B, Sp, **f32)
    l1, l2 = torch.zeros(_lib.NR_LOSS_SLOTS, **f32), torch.zeros(_lib.NR_LOSS_SLOTS, **f32)
    _lib.check(lib.nr_interlevel_loss(p(c), S + 1, p(w), S, S - 1, p(cp), p(wp), Sp, B, 0.03, 1.7, p(g_w), p(l1), st()), "il")
    _lib.check(lib.nr_weights_from_density_bwd(p(dens), p(eup), p(g_w), B, Sp, p(g_d1), st()), "wb")
    _lib.check(lib.nr_interlevel_loss_to_density(p(c), S + 1, p(w), S, S - 1, p(cp), p(wp), p(dens), p(eup), Sp, B, 0.03, 1.7,
                                                 p(g_d2), p(l2), None, st()), "fused")
    assert_close(cpu(g_d2), cpu(g_d1), rtol=1e-6, atol_scale=1e-7, what="g_density")
    assert_close(cpu(l2.sum()), cpu(l1.sum()), rtol=1e-6, atol_scale=1e-7, what="loss")


@pytest.mark.parametrize("S,S_out,sky", [(128, 64, 0.0), (64, 32, 20000.0), (48, 48, 0.0)])
@pytest.mark.parametrize("rows_sm", [0, 203, 77])  # rays stored sample-major: none, all, the leading 77 (mixed batch)
def test_fused_proposal_round_matches_separate_kernels(S, S_out, sky, rows_sm):
    from neuradar_amd import _lib, ops

    lib, p, st = _lib.lib(), ops._p, ops._stream
    torch.manual_seed(S + S_out)
    B = 203
    f32 = dict(device=DEV, dtype=torch.float32)
    o, d = torch.randn(B, 3, **f32) * 5, torch.nn.functional.normalize(torch.randn(B, 3, **f32), dim=-1)
    area = torch.rand(B, **f32) * 1e-5 + 1e-6
    nears, fars = torch.zeros(B, **f32), torch.full((B,), 20000.0, **f32)
    t_rand, jit = torch.rand(B, S + 1, **f32), torch.rand(B, **f32)
    lam, scal, scale = -1.0, 0.1, 100.0
    # level 0: bins (+ contraction)
    sp_a, eu_a = torch.empty(B, S + 1, **f32), torch.empty(B, S + 1, **f32)
    sp_b, eu_b = torch.empty_like(sp_a), torch.empty_like(eu_a)
    x_a, sd_a, x_b, sd_b = torch.empty(B * S, 3, **f32), torch.empty(B * S, **f32), torch.empty(B * S, 3, **f32), torch.empty(B * S, **f32)
    _lib.check(lib.nr_power_bins(p(nears), p(fars), p(t_rand), B, S, lam, scal, p(sp_a), p(eu_a), st()), "pb")
    _lib.check(lib.nr_contract_gaussians(p(o), p(d), p(area), p(eu_a), B, S, scale, rows_sm, p(x_a), p(sd_a), st()), "cg")
    _lib.check(lib.nr_power_bins_contract(p(nears), p(fars), p(t_rand), p(o), p(d), p(area), B, S, lam, scal, scale, rows_sm,
                                          p(sp_b), p(eu_b), p(x_b), p(sd_b), st()), "pbc")
    for got, ref, what in ((sp_b, sp_a, "spacing"), (eu_b, eu_a, "euclid"), (x_b, x_a, "x01"), (sd_b, sd_a, "std01")):
        assert_close(cpu(got), cpu(ref), rtol=1e-6, atol_scale=1e-7, what=what)
    # one round
    dens = torch.rand(B, S, **f32) * 0.2
    dens[::5] = 0.0
    w_a, dep_a = torch.empty(B, S, **f32), torch.empty(B, **f32)
    sp2_a, eu2_a = torch.empty(B, S_out + 1, **f32), torch.empty(B, S_out + 1, **f32)
    x2_a, sd2_a = torch.empty(B * S_out, 3, **f32), torch.empty(B * S_out, **f32)
    _lib.check(lib.nr_weights_from_density_fwd(p(dens), p(eu_a), B, S, p(w_a), st()), "w")
    _lib.check(lib.nr_depth_from_weights(p(w_a), p(eu_a), B, S, p(dep_a), st()), "d")
    _lib.check(lib.nr_pdf_resample(p(w_a), p(sp_a), p(jit), p(nears), p(fars), B, S, S_out, lam, scal, sky, p(sp2_a), p(eu2_a), st()), "pdf")
    _lib.check(lib.nr_contract_gaussians(p(o), p(d), p(area), p(eu2_a), B, S_out, scale, rows_sm, p(x2_a), p(sd2_a), st()), "cg2")
    w_b, dep_b = torch.empty_like(w_a), torch.empty_like(dep_a)
    sp2_b, eu2_b, x2_b, sd2_b = torch.empty_like(sp2_a), torch.empty_like(eu2_a), torch.empty_like(x2_a), torch.empty_like(sd2_a)
    _lib.check(lib.nr_proposal_round(p(dens), p(eu_a), p(sp_a), p(jit), p(nears), p(fars), p(o), p(d), p(area), B, S, S_out, lam,
                                     scal, sky, scale, rows_sm, p(w_b), p(dep_b), p(sp2_b), p(eu2_b), p(x2_b), p(sd2_b), st()), "round")
    for got, ref, what in ((w_b, w_a, "weights"), (dep_b, dep_a, "depth"), (sp2_b, sp2_a, "spacing_out"), (eu2_b, eu2_a, "euclid_out"),
                           (x2_b, x2_a, "x01_out"), (sd2_b, sd2_a, "std01_out")):
        assert_close(cpu(got), cpu(ref), rtol=2e-6, atol_scale=1e-6, what=what)


@pytest.mark.parametrize("F", [1, 2, 4])
def test_sparse_gradient_lists_roundtrip(F):
    """nr_grad_compact moves exactly the non-zero rows out of the table; nr_grad_apply adds a list back."""
    from neuradar_amd import ops

    torch.manual_seed(F)
    rows = 300_001
    grad = torch.zeros(rows, F, device=DEV)
    hit = torch.randperm(rows, device=DEV)[:5000]
    grad[hit] = torch.randn(5000, F, device=DEV)
    grad[hit[:10], 0] = 0.0  # rows with a zero component still count when another one is non-zero
    if F == 1:
        grad[hit[:10]] = 1.0
    ref = grad.clone()
    cap = 8192
    idx = torch.zeros(cap, dtype=torch.int32, device=DEV)
    val = torch.zeros(cap * F, device=DEV)
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.grad_compact(grad.view(-1), F, idx, val, count)
    n = int(count)
    assert n == int((ref != 0).any(dim=1).sum()) and float(grad.abs().max()) == 0.0
    assert torch.equal(torch.sort(idx[:n]).values.long(), torch.sort((ref != 0).any(dim=1).nonzero()[:, 0]).values)
    ops.grad_apply(idx, val, count, F, grad.view(-1))
    assert torch.equal(grad, ref)
    # overflow: rows beyond the capacity stay in the table, count reports the true number
    small_idx = torch.zeros(1000, dtype=torch.int32, device=DEV)
    small_val = torch.zeros(1000 * F, device=DEV)
    count.zero_()
    ops.grad_compact(grad.view(-1), F, small_idx, small_val, count)
    assert int(count) == n
    ops.grad_apply(small_idx, small_val, count, F, grad.view(-1))
    assert torch.equal(grad, ref)


def test_uniform_fill_is_uniform_and_counter_based():
    from neuradar_amd import ops

    n = 1 << 20
    a, b, c = (torch.empty(n, device=DEV) for _ in range(3))
    epoch = torch.tensor([3.0], device=DEV)
    ops.uniform_fill(a, 1234, epoch)
    ops.uniform_fill(b, 1234, epoch)
    assert torch.equal(a, b)  # same (seed, epoch) -> same numbers
    epoch += 1
    ops.uniform_fill(c, 1234, epoch)
    assert not torch.equal(a, c)
    for t in (a, c):
        assert float(t.min()) >= 0.0 and float(t.max()) < 1.0
        assert abs(float(t.mean()) - 0.5) < 2e-3 and abs(float(t.var()) - 1.0 / 12.0) < 1e-3
        hist = torch.histc(t, bins=64, min=0.0, max=1.0) / n
        assert float((hist - 1.0 / 64).abs().max()) < 1e-3
    assert abs(float(torch.corrcoef(torch.stack([a, c]))[0, 1])) < 5e-3
    assert abs(float(torch.corrcoef(torch.stack([a[:-1], a[1:]]))[0, 1])) < 5e-3


def test_reference_pinned_vectors_on_the_hip_path():
    """The reference's own value-pinning tests through the C ABI: frustum midpoint
    (tests/cameras/test_rays.py:12-34), SH orthonormality (tests/utils/test_math.py:7-16), pinhole
    origin (tests/cameras/test_cameras.py:109-121)."""
    from neuradar_amd import ops
    from neuradar_amd.sensors import Cameras

    o, d = torch.tensor([[0.0, 1.0, 2.0]], device=DEV), torch.tensor([[0.0, 1.0, 0.0]], device=DEV)
    x01, _ = ops.contract_gaussians(o, d, torch.ones(1, device=DEV), torch.tensor([[2.0, 3.0]], device=DEV), 100.0)
    assert torch.allclose(x01[0] * 4.0 - 2.0, torch.tensor([0.0, 3.5, 2.0], device=DEV) / 100.0, atol=1e-6)  # inside the unit box: no contraction
    torch.manual_seed(0)
    n = 1_000_000
    dirs = torch.nn.functional.normalize(torch.normal(0, 1, size=(n, 3), device=DEV), dim=-1)
    sh = ops.sh4(dirs).double()
    torch.testing.assert_close(cpu((sh.T @ sh) / n * 4 * torch.pi), torch.eye(16, dtype=torch.float64), rtol=0, atol=1.5e-2)
    one = torch.ones(1, device=DEV)
    cam = Cameras(torch.eye(4, device=DEV)[None, :3, :].contiguous(), 10 * one, 10 * one, 400 * one, 400 * one, 800 * one, 0 * one)
    b = cam.generate_rays(torch.tensor([[0, 0, 0], [0, 400, 400], [0, 799, 13]], device=DEV))
    assert torch.allclose(b.origins, torch.zeros(3, 3, device=DEV))
    assert torch.allclose(b.directions.norm(dim=-1), torch.ones(3, device=DEV), atol=1e-6)


@pytest.mark.parametrize("sm_frac", [1.0, 0.0, 0.37])
def test_prop_field_fwd_matches_two_launches(sm_frac):
    """nr_prop_field_fwd == nr_hash_encode_fwd -> nr_prop_density_fwd (features and densities), and the reference."""
    from neuradar_amd import _lib, ops

    lib, p, st = _lib.lib(), ops._p, ops._stream
    g = load_golden("field_neurad")
    pf = make_prop(g)
    rs = samples_from_edges(g, g["edges"])
    B, S = rs.shape
    sm = int(sm_frac * B)
    grid = pf.hashgrid.static_grid
    L, F, n = grid.num_levels, grid.features_per_level, B * S
    x01, std = ops.contract_gaussians(rs.origins, rs.directions, rs.pixel_area, rs.euclid, pf.hashgrid.static_scale, sample_major_rows=sm)
    table = grid.hash_table.detach()
    w = pf.density_decoder.weight.detach().reshape(-1).contiguous()
    f1, d1 = torch.empty(L, n, F, device=DEV), torch.empty(n, device=DEV)
    _lib.check(lib.nr_hash_encode_fwd(p(x01), p(std), p(table), p(grid.scalings), L, F, grid.log2_hashmap_size, p(f1), F, n * F, n, 0,
                                      st()), "hash")
    _lib.check(lib.nr_prop_density_fwd(p(f1), F, n * F, F, p(w), w.numel(), n, S, sm, p(d1), st()), "density")
    f2, d2 = torch.empty_like(f1), torch.empty_like(d1)
    _lib.check(lib.nr_prop_field_fwd(p(x01), p(std), p(table), p(grid.scalings), L, F, grid.log2_hashmap_size, p(w), p(f2), F, n * F, n,
                                     S, sm, p(d2), st()), "fused")
    assert torch.equal(f2, f1)
    assert_close(cpu(d2), cpu(d1), rtol=1e-6, atol_scale=1e-7, what="density")
    assert_close(cpu(d2.view(B, S, 1)), g["prop_density"], rtol=1e-4, atol_scale=1e-5, what="density vs reference")


def test_row_orders_are_permutations_of_one_result():
    """Sample-major / hybrid / ray-major row storage (nr_contract_gaussians' sample_major_rows) only permutes
    rows: grid features, densities and the table gradient agree exactly."""
    from neuradar_amd import ops

    g = load_golden("field_neurad")
    pf = make_prop(g)
    rs = samples_from_edges(g, g["edges"])
    B, S = rs.shape
    grid = pf.hashgrid.static_grid
    ref_d, ref_gt = None, None
    for sm in (0, B, B // 3 + 1):
        x01, std = ops.contract_gaussians(rs.origins, rs.directions, rs.pixel_area, rs.euclid, pf.hashgrid.static_scale,
                                          sample_major_rows=sm)
        table = grid.hash_table.detach().clone().requires_grad_(True)
        buf = ops.hash_encode(x01, table, grid.scalings, grid.log2_hashmap_size, std=std, level_major=True)
        n = B * S
        dens = ops.prop_density(buf, (grid.features_per_level, n * grid.features_per_level), grid.features_per_level,
                                pf.density_decoder.weight, n, n_samples=S, rows_sample_major=sm)
        (gt,) = torch.autograd.grad((dens * dev(g["prop_g_density"]).reshape(-1)).sum(), [table])
        if ref_d is None:
            ref_d, ref_gt = dens, gt
            assert_close(cpu(dens.view(B, S, 1)), g["prop_density"], rtol=1e-4, atol_scale=1e-5)
        else:
            assert torch.equal(dens, ref_d), f"densities differ for sm={sm}"
            assert_close(cpu(gt), cpu(ref_gt), rtol=1e-5, atol_scale=1e-6, what=f"table grad sm={sm}")


def test_field_row_orders_are_permutations_of_one_result():
    """nr_field_fwd/bwd with ray-major, sample-major and hybrid rows: identical outputs, matching gradients."""
    from neuradar_amd import ops

    g = load_golden("field_neurad")
    fld = make_field(g)
    rs = samples_from_edges(g, g["edges"])
    B, S = rs.shape
    grid = fld.hashgrid.static_grid
    n, F = B * S, grid.features_per_level
    ref = None
    for sm in (0, B, B // 3 + 1):
        x01, std = ops.contract_gaussians(rs.origins, rs.directions, rs.pixel_area, rs.euclid, fld.hashgrid.static_scale,
                                          sample_major_rows=sm)
        table = grid.hash_table.detach().clone().requires_grad_(True)
        buf = ops.hash_encode(x01, table, grid.scalings, grid.log2_hashmap_size, std=std, level_major=True)
        params = [p_.detach().clone().requires_grad_(True) for p_ in (*fld.mlp_geo.weights()[0], *fld.mlp_geo.weights()[1],
                                                                       *fld.mlp_feature.weights()[0], *fld.mlp_feature.weights()[1])]
        ng, nf = len(fld.mlp_geo.weights()[0]), len(fld.mlp_feature.weights()[0])
        geo = (params[:ng], params[ng:2 * ng])
        feat = (params[2 * ng:2 * ng + nf], params[2 * ng + nf:])
        beta = fld.sdf_to_density.beta.detach().clone().requires_grad_(True)
        feature, sdf, alpha = ops.field_mlp(buf, (F, n * F), F, rs.directions, S, n, geo, feat, beta, rows_sample_major=sm)
        loss = (feature.view(B, S, -1) * dev(g["g_feature"])).sum() + (alpha.view(B, S, 1) * dev(g["g_alpha"])).sum()
        grads = torch.autograd.grad(loss, [table, beta, *params])
        if ref is None:
            ref = (feature, sdf, alpha, grads)
            assert_close(cpu(feature.view(B, S, -1)), g["feature"], rtol=1e-4, atol_scale=1e-5, what="feature")
        else:
            assert torch.equal(feature, ref[0]) and torch.equal(sdf, ref[1]) and torch.equal(alpha, ref[2]), f"outputs differ, sm={sm}"
            for a_, b_ in zip(grads, ref[3]):
                assert_close(cpu(a_), cpu(b_), rtol=1e-4, atol_scale=1e-5, what=f"grad sm={sm}")


@pytest.mark.parametrize("coherent", [None, 0.4, 0.0])  # all rays sample-major (camera), a mixed batch, all ray-major
def test_fused_step_matches_autograd_path(coherent):
    """The autograd-free fused step (what bench.py times) reproduces outputs, loss and EVERY parameter
    gradient of the modular autograd path, which the tests above pin to the reference goldens."""
    from neuradar_amd.fused_step import FusedTrainStep
    from neuradar_amd.rays import RayBundle

    g = load_golden("pipeline")
    model = build_hot_path(g).train()
    o, d, area, fars = dev(g["origins"]), dev(g["directions"]), dev(g["pixel_area"]), dev(g["fars"])
    t_rand, j1, j2 = dev(g["t_rand"]), dev(g["jitter1"]), dev(g["jitter2"])
    tf, td = dev(g["target_features"]), dev(g["target_depth"])
    out = model.get_nff_outputs(RayBundle(o, d, area, fars=fars.clone()), t_rand=t_rand, jitters=(j1, j2))
    loss = model.bench_loss(out, tf, td)
    params = [p for p in model.parameters() if p.requires_grad]
    ref_grads = torch.autograd.grad(loss, params, allow_unused=True)
    fused = FusedTrainStep(model, o.shape[0], coherent_rays=None if coherent is None else int(coherent * o.shape[0]))
    for p in params:
        p.grad.zero_()
    floss = fused.forward_backward(o, d, area[:, 0].contiguous(), fars[:, 0].contiguous(), tf, td[:, 0].contiguous(), t_rand,
                                   j1[:, 0].contiguous(), j2[:, 0].contiguous())
    fo = fused.outputs()
    assert_close(cpu(fo["features"]), g["features"], rtol=1e-4, atol_scale=1e-4, what="features")
    assert_close(cpu(fo["depth"]), g["depth"], rtol=1e-4, atol_scale=1e-4, what="depth")
    assert_close(cpu(fo["accumulation"]), g["accumulation"], rtol=1e-4, atol_scale=1e-4, what="accumulation")
    assert_close(cpu(fo["final_euclid"]), g["final_euclid"], rtol=1e-3, atol_scale=1e-5, what="final euclid")
    assert_close(cpu(fo["prop_depth_0"]), g["prop_depth_0"], rtol=1e-3, atol_scale=1e-4)
    assert_close(cpu(floss).sum(), g["loss"], rtol=1e-4, atol_scale=1e-5, what="loss")
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    for n, p, rg in zip(names, params, ref_grads):
        if rg is None:
            assert float(p.grad.abs().max()) == 0.0, n  # proposal_fields[0]: never evaluated
            continue
        assert_close(cpu(p.grad), cpu(rg), rtol=1e-3, atol_scale=1e-4, what="fused grad " + n)
    # a second call accumulates (+=) -- the contract the Adam kernel's zero_grad relies on
    fused.forward_backward(o, d, area[:, 0].contiguous(), fars[:, 0].contiguous(), tf, td[:, 0].contiguous(), t_rand,
                           j1[:, 0].contiguous(), j2[:, 0].contiguous())
    t = model.field.hashgrid.static_grid.hash_table
    rg = ref_grads[names.index("field.hashgrid.static_grid.hash_table")]
    assert_close(cpu(t.grad), 2 * cpu(rg), rtol=1e-3, atol_scale=1e-4, what="accumulated grad")
    # the split step (prepare() early into the second buffer set, then forward_backward(prepared=True)): what the
    # cross-step pipelining of bench.py runs; bit-identical outputs, and slot 0 is left untouched by it
    x0 = fused._slots[0]["x01"].clone()
    fo = {k_: v_.clone() for k_, v_ in fo.items()}  # outputs() are views of the step's buffers
    fused.prepare(1, o, d, area[:, 0].contiguous(), fars[:, 0].contiguous(), t_rand)
    fused.forward_backward(o, d, area[:, 0].contiguous(), None, tf, td[:, 0].contiguous(), None, j1[:, 0].contiguous(),
                           j2[:, 0].contiguous(), slot=1, prepared=True)
    fo2 = fused.outputs()
    for k_ in ("features", "depth", "accumulation", "final_euclid"):
        assert torch.equal(fo2[k_], fo[k_]), k_
    assert torch.equal(fused._slots[0]["x01"], x0) and torch.equal(fused._slots[1]["x01"], x0)


@pytest.mark.parametrize("schedule", ["0", "1", "2", "3", "4"])
def test_fused_step_schedules_agree(schedule, monkeypatch):
    """Every backward schedule of the fused step (NR_EARLY_FORK: which chains start beside nr_field_bwd, where the
    weight-gradient slabs are reduced -- nr_field_bwd(grads=NULL) + nr_field_grad_reduce) yields the same outputs and
    parameter gradients as the serial, single-stream step."""
    from neuradar_amd.fused_step import FusedTrainStep

    g = load_golden("pipeline")
    o, d, area, fars = dev(g["origins"]), dev(g["directions"]), dev(g["pixel_area"]), dev(g["fars"])
    t_rand, j1, j2 = dev(g["t_rand"]), dev(g["jitter1"]), dev(g["jitter2"])
    tf, td = dev(g["target_features"]), dev(g["target_depth"])

    def run(overlap):
        model = build_hot_path(g).train()
        fused = FusedTrainStep(model, o.shape[0], overlap=overlap)
        loss = fused.forward_backward(o, d, area[:, 0].contiguous(), fars[:, 0].contiguous(), tf, td[:, 0].contiguous(), t_rand,
                                      j1[:, 0].contiguous(), j2[:, 0].contiguous())
        torch.cuda.synchronize()
        return float(loss.sum()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    monkeypatch.setenv("NR_EARLY_FORK", "0")
    monkeypatch.setenv("NR_FIELD_STASH", "0")  # the reference run also recomputes the field's forward in its backward
    ref_loss, ref = run(False)
    monkeypatch.setenv("NR_EARLY_FORK", schedule)
    monkeypatch.delenv("NR_FIELD_STASH")  # ... the scheduled runs read it from the activation stash (nr_field_t.stash)
    loss, grads = run(True)
    assert abs(loss - ref_loss) <= 1e-6 * abs(ref_loss)
    assert grads.keys() == ref.keys()
    for n in ref:  # scatter-adds and loss slots are summed in a different order: float noise only
        assert_close(cpu(grads[n]), cpu(ref[n]), rtol=1e-4, atol_scale=1e-5, what=f"schedule {schedule}: {n}")


def test_fused_training_matches_torch_training():
    """Three whole training steps -- fused forward/backward with the optimizers stepped inside it (Adam on the
    tables, AdamW on the small parameters, the reference's LR schedule) -- against the modular autograd path
    stepped by torch.optim with the same hyper-parameters and schedule (method_configs.py:384-409): losses and
    final parameters agree.  Adam normalises the update, so float noise in a tiny gradient can flip a whole
    lr-sized step; parameters are therefore compared on the entries whose gradient is above the noise floor."""
    import math

    from neuradar_amd.fused_step import FusedTrainStep
    from neuradar_amd.rays import RayBundle
    from neuradar_amd.step import FlatAdam

    g = load_golden("pipeline")
    o, d, area, fars = dev(g["origins"]), dev(g["directions"]), dev(g["pixel_area"]), dev(g["fars"])
    t_rand, j1, j2 = dev(g["t_rand"]), dev(g["jitter1"]), dev(g["jitter2"])
    tf, td = dev(g["target_features"]), dev(g["target_depth"])
    lr, lr_final, warm, max_steps = 1e-2, 1e-3, 2, 10

    def schedule(k):  # ExponentialDecayScheduler (schedulers.py:112-143), k = 0-based step
        if k < warm:
            return 1e-8 + (lr - 1e-8) * math.sin(0.5 * math.pi * min(max(k / warm, 0.0), 1.0))
        t = min(max((k - warm) / (max_steps - warm), 0.0), 1.0)
        return math.exp(math.log(lr) * (1 - t) + math.log(lr_final) * t)

    # torch side
    ref = build_hot_path(g).train()
    groups = ref.get_param_groups()
    unused = {id(p) for p in ref.proposal_fields[0].parameters()}
    tabs = [p for p in groups["hashgrids"] if id(p) not in unused]
    small = [p for p in groups["fields"] if id(p) not in unused]
    o_tab = torch.optim.Adam(tabs, lr=lr, eps=1e-15)
    o_small = torch.optim.AdamW(small, lr=lr, eps=1e-15, weight_decay=1e-7)
    ref_losses, first_grads = [], None
    for k in range(3):
        for opt in (o_tab, o_small):
            for gr in opt.param_groups:
                gr["lr"] = schedule(k)
        out = ref.get_nff_outputs(RayBundle(o, d, area, fars=fars.clone()), t_rand=t_rand, jitters=(j1, j2))
        loss = ref.bench_loss(out, tf, td)
        for p in tabs + small:
            p.grad = None
        loss.backward()
        if first_grads is None:
            first_grads = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        o_tab.step()
        o_small.step()
        ref_losses.append(float(loss.detach()))
    # fused side
    model = build_hot_path(g).train()
    mg = model.get_param_groups()
    skip = list(model.proposal_fields[0].parameters())
    opts = [FlatAdam(mg["hashgrids"], lr=lr, eps=1e-15, lr_final=lr_final, max_steps=max_steps, warmup_steps=warm, skip=skip),
            FlatAdam(mg["fields"], lr=lr, eps=1e-15, weight_decay=1e-7, adamw=True, lr_final=lr_final, max_steps=max_steps,
                     warmup_steps=warm, skip=skip)]
    fused = FusedTrainStep(model, o.shape[0])
    start = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses = []
    for k in range(3):
        l_ = fused.forward_backward(o, d, area[:, 0].contiguous(), fars[:, 0].contiguous(), tf, td[:, 0].contiguous(), t_rand,
                                    j1[:, 0].contiguous(), j2[:, 0].contiguous(), optimizers=opts)
        losses.append(float(l_.sum()))
    for a_, b_ in zip(losses, ref_losses):
        assert abs(a_ - b_) <= 2e-4 * abs(b_), (losses, ref_losses)
    assert losses[2] < losses[0]  # it trains
    rp = dict(ref.named_parameters())
    for n, p in model.named_parameters():
        if n not in first_grads:  # proposal_fields[0]: never evaluated (reference quirk), never stepped
            assert torch.equal(p.detach(), start[n]), n
            continue
        g0 = first_grads[n]
        solid = g0.abs() > 1e-3 * g0.abs().max()  # entries whose Adam direction is not decided by float noise
        diff = (p.detach() - rp[n].detach()).abs()
        assert float(diff[solid].max()) < 2e-4, f"{n}: {float(diff[solid].max()):.3e}"  # steps are ~1e-2 each
        assert float(diff.max()) <= 3 * lr * 1.0001, n  # nothing can differ by more than the three steps themselves


def test_lidar_decoder_and_losses_vs_oracle():
    """SURVEY 8f-2, first item: the per-ray lidar decoder (neuradar.py:241-248, 432-452) on the MFMA MLP kernels and
    the lidar losses (:612-636, multipliers :690-700) against the oracle restatement, forward and parameter /
    feature gradients; the decoder MLP itself is pinned by the reference-generated sh_mlp golden."""
    from neuradar_amd.step import HotPathConfig, NeuRadarHotPath
    from oracle import decoders

    g = load_golden("sh_mlp")
    cfg = HotPathConfig(appearance_dim=16, lidar_decoder=True, num_sensors=2)
    cfg.field.grid.static.log2_hashmap_size = 12
    cfg.proposal_field_1.grid.static.log2_hashmap_size = 12
    cfg.proposal_field_2.grid.static.log2_hashmap_size = 12
    model = NeuRadarHotPath(cfg).to(DEV).train()
    with torch.no_grad():
        for i, layer in enumerate(model.lidar_decoder.layers):
            layer.weight.copy_(dev(g[f"mlp_w{i}"]))
            layer.bias.copy_(dev(g[f"mlp_b{i}"]))
    assert any(p is model.lidar_decoder.layers[0].weight for p in model.get_param_groups()["fields"])
    # the reference's own vector first: all rows are lidar rays
    x = dev(g["mlp_x"])
    inten, drop = model.decode_lidar(x, torch.ones(x.shape[0], 1, dtype=torch.bool, device=DEV))
    y = torch.as_tensor(g["mlp_y"])
    assert_close(cpu(inten), torch.sigmoid(y[:, :1]), rtol=1e-4, atol_scale=1e-5, what="intensity vs reference MLP")
    assert_close(cpu(drop), y[:, 1:], rtol=1e-4, atol_scale=1e-5, what="ray drop logit vs reference MLP")
    # a mixed batch: 4 661 lidar rays among camera rays, 10 % non-returns
    torch.manual_seed(5)
    B, n_l = 8192, 4661
    feats = torch.randn(B, 48)
    is_lidar = torch.zeros(B, 1, dtype=torch.bool)
    is_lidar[torch.randperm(B)[:n_l]] = True
    depth = 2.0 + 148.0 * torch.rand(n_l, 1)
    term = depth + 0.5 * torch.randn(n_l, 1)
    did_return = torch.rand(n_l) > 0.1
    pts_int = torch.rand(n_l, 1)
    ws = [cpu(l.weight.detach()) for l in model.lidar_decoder.layers]
    bs = [cpu(l.bias.detach()) for l in model.lidar_decoder.layers]
    fr, dr = feats.clone().requires_grad_(True), depth.clone().requires_grad_(True)
    wr, br = [w.clone().requires_grad_(True) for w in ws], [b.clone().requires_grad_(True) for b in bs]
    i_ref, l_ref = decoders.lidar_decode(fr, is_lidar, wr, br)
    ref = decoders.lidar_losses(dr, i_ref, l_ref, term, did_return, pts_int)
    ref_total = 0.01 * ref["depth_loss"] + 0.1 * ref["intensity_loss"] + 0.01 * ref["ray_drop_loss"]
    ref_g = torch.autograd.grad(ref_total, [fr, dr, *wr, *br])
    fd, dd = dev(feats).requires_grad_(True), dev(depth).requires_grad_(True)
    i_hip, l_hip = model.decode_lidar(fd, dev(is_lidar))
    out = model.lidar_losses(dd, i_hip, l_hip, dev(term), dev(did_return), dev(pts_int))
    assert_close(cpu(i_hip.detach()), i_ref.detach(), rtol=1e-4, atol_scale=1e-5, what="intensity")
    for k, mult in (("depth_loss", 0.01), ("intensity_loss", 0.1), ("ray_drop_loss", 0.01)):
        assert abs(float(out[k].detach()) - mult * float(ref[k].detach())) <= 1e-4 * abs(mult * float(ref[k].detach())) + 1e-9, k
    params = [l.weight for l in model.lidar_decoder.layers] + [l.bias for l in model.lidar_decoder.layers]
    got = torch.autograd.grad(sum(out.values()), [fd, dd, *params])
    for a_, b_, n_ in zip(got, ref_g, ["features", "depth", "w0", "w1", "w2", "b0", "b1", "b2"]):
        assert_close(cpu(a_), b_, rtol=1e-3, atol_scale=1e-5, what="lidar loss grad " + n_)
    assert model.decode_lidar(fd, torch.zeros(B, 1, dtype=torch.bool, device=DEV)) == (None, None)


def test_flat_adam_flattening_keeps_parameters_and_matches_torch():
    from neuradar_amd.step import FlatAdam

    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.Linear(64, 33)).to(DEV)
    big = torch.nn.Parameter(torch.randn(1 << 17, 2, device=DEV))
    ref = [p.detach().clone().requires_grad_(True) for p in list(lin.parameters()) + [big]]
    topt = torch.optim.AdamW(ref, lr=1e-2, eps=1e-15, weight_decay=1e-7)
    opt = FlatAdam(list(lin.parameters()) + [big], lr=1e-2, eps=1e-15, weight_decay=1e-7, adamw=True, warmup_steps=0)
    x = torch.randn(16, 32, device=DEV)
    assert torch.equal(lin(x), torch.nn.functional.linear(torch.nn.functional.linear(x, ref[0], ref[1]), ref[2], ref[3]))
    for _ in range(3):
        for p, r in zip(list(lin.parameters()) + [big], ref):
            gr = torch.randn_like(p)
            p.grad.copy_(gr)
            r.grad = gr.clone()
        opt.step()
        topt.step()
    for p, r in zip(list(lin.parameters()) + [big], ref):
        torch.testing.assert_close(p.data, r.data, rtol=1e-5, atol=1e-7)
        assert float(p.grad.abs().max()) == 0.0
    assert set(dict(lin.named_parameters())) == {"0.weight", "0.bias", "1.weight", "1.bias"}


def test_patch_ray_assembly_matches_index_path_and_adam_hyper_schedule():
    """f-1: the one-kernel patch sampler + ray generator equals generate_rays on the indices it
    reports; nr_adam_hyper reproduces the reference LR schedule and bias corrections."""
    import math

    import numpy as np

    from neuradar_amd import ops
    from neuradar_amd.sensors import Cameras

    g = load_golden("raygen")
    cams = Cameras(dev(g["cam_c2w"]), dev(g["cam_fx"]), dev(g["cam_fy"]), dev(g["cam_cx"]), dev(g["cam_cy"]),
                   dev(g["cam_heights"]), dev(g["cam_times_in"]), dev(g["cam_vel"]), dev(g["cam_rs_offsets"]))
    torch.manual_seed(0)
    u = torch.rand(7, 3, device=DEV)
    u[0] = 0.0
    u[1] = 0.999999
    b, idx = cams.generate_patch_rays(u, 32, 3, 1080, 1920, area_scale=9.0, return_indices=True)
    assert idx.shape == (7 * 1024, 3)
    assert int(idx[:, 0].max()) < 5 and int(idx[:, 1].max()) < 1080 and int(idx[:, 2].max()) < 1920 and int(idx.min()) >= 0
    p0 = idx[:1024].view(32, 32, 3)
    assert torch.equal(p0[:, :, 1], (p0[0, 0, 1] + 3 * torch.arange(32, device=DEV))[:, None].expand(32, 32))
    assert torch.equal(p0[:, :, 2], (p0[0, 0, 2] + 3 * torch.arange(32, device=DEV))[None, :].expand(32, 32))
    ref = cams.generate_rays(idx)
    assert torch.equal(b.origins, ref.origins) and torch.equal(b.directions, ref.directions)
    torch.testing.assert_close(b.pixel_area, ref.pixel_area * 9.0, rtol=1e-6, atol=0)
    assert torch.equal(b.times, ref.times)

    step_t, hyper = torch.zeros(2, device=DEV), torch.zeros(3, device=DEV)  # (scheduler steps, optimizer updates)
    for k in range(1, 4):
        ops.check(ops._lib.lib().nr_adam_hyper(ops._p(step_t), ops._p(hyper), 1e-2, 1e-3, 500, 20001, 0.9, 0.999, None, 0, ops._stream()), "hyper")
        lr = 1e-8 + (1e-2 - 1e-8) * np.sin(0.5 * np.pi * (k - 1) / 500)
        want = torch.tensor([lr, 1 - 0.9**k, math.sqrt(1 - 0.999**k)])
        torch.testing.assert_close(cpu(hyper), want.float(), rtol=1e-5, atol=1e-12)
    step_t.fill_(10000.0)
    ops.check(ops._lib.lib().nr_adam_hyper(ops._p(step_t), ops._p(hyper), 1e-2, 1e-3, 500, 20001, 0.9, 0.999, None, 0, ops._stream()), "hyper")
    t = (10000 - 500) / (20001 - 500)
    assert abs(float(hyper[0]) - math.exp(math.log(1e-2) * (1 - t) + math.log(1e-3) * t)) < 1e-8


# ------------------------------------------------------------------------------------------------ a10 dynamic actors
def test_dynamic_actors_vs_reference_golden():
    """Culling, world->box transform, per-ray flip, one 3-D grid per actor, overwrite of the static
    features, per-sample view directions, trajectory gradients (through nr_hash_encode_bwd_input)."""
    from neuradar_amd.dynamic_actors import DynamicActors
    from neuradar_amd.field_heads import FieldHeadNames
    from neuradar_amd.neurad_encoding import ActorSettings, NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig, NeuRADProposalFieldConfig
    from neuradar_amd.rays import RaySamples

    g = load_golden("actors")
    actors = DynamicActors.from_state(g["actor_positions"], g["actor_rotations_6d"], g["actor_timestamps"],
                                      g["actor_present"], g["actor_sizes"]).to(DEV)
    grid_cfg = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=int(g["log2t"])),
                                        actor=ActorSettings(flip_prob=0.25, log2_hashmap_size=int(g["actor_log2t"])))
    fld = NeuRADFieldConfig(grid=grid_cfg).setup(actors=actors, static_scale=100.0).to(DEV)
    pcfg = NeuRADProposalFieldConfig()
    pcfg.grid.static.log2_hashmap_size = int(g["prop_log2t"])
    pcfg.grid.actor.log2_hashmap_size = int(g["actor_log2t"])
    prop = pcfg.setup(actors=actors, static_scale=100.0).to(DEV)
    with torch.no_grad():
        fld.hashgrid.static_grid.hash_table.copy_(g["table"])
        for i, gr in enumerate(fld.hashgrid.actor_grids):
            gr.hash_table.copy_(g[f"actor{i}_table"])
        for i, l in enumerate(fld.mlp_geo.layers):
            l.weight.copy_(g[f"geo_w{i}"]); l.bias.copy_(g[f"geo_b{i}"])
        for i, l in enumerate(fld.mlp_feature.layers):
            l.weight.copy_(g[f"feat_w{i}"]); l.bias.copy_(g[f"feat_b{i}"])
        fld.sdf_to_density.beta.copy_(g["beta"])
        prop.hashgrid.static_grid.hash_table.copy_(g["prop_table"])
        for i, gr in enumerate(prop.hashgrid.actor_grids):
            gr.hash_table.copy_(g[f"prop_actor{i}_table"])
        prop.density_decoder.weight.copy_(g["prop_decoder"])
    e = g["edges"]
    B = e.shape[0]
    rs = RaySamples(dev(g["origins"]), dev(g["directions"]), dev(g["pixel_area"]), dev(torch.zeros_like(e)), dev(e),
                    dev(torch.zeros(B, 1)), dev(torch.full((B, 1), 1e6)), times=dev(g["times"]))
    fld.eval(); prop.eval()
    out = fld(rs)
    assert_close(cpu(out[FieldHeadNames.FEATURE]), g["eval_feature"], rtol=1e-4, atol_scale=1e-5, what="eval feature")
    assert_close(cpu(out[FieldHeadNames.ALPHA]), g["eval_alpha"], rtol=1e-4, atol_scale=1e-5, what="eval alpha")
    assert_close(cpu(prop.get_density(rs)[0]), g["eval_prop_density"], rtol=1e-4, atol_scale=1e-5, what="eval prop density")
    # the (ray, sample, actor) triples of the reference's culling (neurad_encoding.py:231-275), from the device-side lookup
    from neuradar_amd import ops

    hg = fld.hashgrid
    geom = hg.actor_geometry(rs)
    S = e.shape[1] - 1
    slot = torch.empty(B * S, device=DEV, dtype=torch.int32)
    xa, sa = torch.empty(B * S, 3, device=DEV), torch.empty(B * S, device=DEV)
    lib, p = ops._lib.lib(), ops._p
    ops.check(lib.nr_actor_assign(p(rs.origins), p(rs.directions), p(rs.pixel_area.reshape(-1).contiguous()), p(rs.euclid), B, S, 0,
                                  p(geom["cand"]), hg.MAX_CANDIDATES, p(geom["w2b"].detach()), p(geom["centres"]), p(geom["bounds"]),
                                  hg.config.actor.actor_scale, None, p(slot), p(xa), p(sa), None, ops._stream()), "assign")
    assert int(hg.actor_overflow) == 0
    hit = (slot.view(B, S) >= 0).nonzero()
    actor = geom["cand"][hit[:, 0], slot.view(B, S)[hit[:, 0], hit[:, 1]].long()]
    got = set(zip(cpu(hit[:, 0]).tolist(), cpu(hit[:, 1]).tolist(), cpu(actor).tolist()))
    want = {}
    for r, s_, a in zip(g["actor_ray_idx"].tolist(), g["actor_sample_idx"].tolist(), g["actor_actor_idx"].tolist()):
        want[(r, s_)] = max(a, want.get((r, s_), -1))  # a sample inside two boxes: the later (higher) actor's write wins
    assert got == {(r, s_, a) for (r, s_), a in want.items()} and len(got) > 0
    fld.train()
    out = fld(rs, flip=dev(g["flip"]))
    assert_close(cpu(out[FieldHeadNames.FEATURE]), g["train_feature"], rtol=1e-4, atol_scale=1e-5, what="train feature")
    assert_close(cpu(out[FieldHeadNames.ALPHA]), g["train_alpha"], rtol=1e-4, atol_scale=1e-5, what="train alpha")
    loss = (out[FieldHeadNames.FEATURE] * dev(g["g_feature"])).sum() + (out[FieldHeadNames.ALPHA] * dev(g["g_alpha"])).sum()
    wrt = {"hashgrid_static_grid_hash_table": fld.hashgrid.static_grid.hash_table,
           "hashgrid_actor_grids_0_hash_table": fld.hashgrid.actor_grids[0].hash_table,
           "hashgrid_actor_grids_1_hash_table": fld.hashgrid.actor_grids[1].hash_table,
           "hashgrid_actors_actor_positions": actors.actor_positions,
           "hashgrid_actors_actor_rotations_6d": actors.actor_rotations_6d,
           "mlp_geo_layers_0_weight": fld.mlp_geo.layers[0].weight}
    for (k, _), gr in zip(wrt.items(), torch.autograd.grad(loss, list(wrt.values()))):
        assert_close(cpu(gr), g["grad_" + k], rtol=1e-3, atol_scale=1e-4, what="grad " + k)


def test_dynamic_actors_honour_actor_to_id():
    """neurad_encoding.py:183: the hash grid of actor a is actor_grids[actor_to_id[a]].  With the two actors' tables
    swapped AND actor_to_id = [1, 0] the reference golden must come out unchanged -- modular path (values, table
    gradients land in the swapped grids) and the fused step's launches (same kernels, same table_of_actor array)."""
    from neuradar_amd.dynamic_actors import DynamicActors
    from neuradar_amd.field_heads import FieldHeadNames
    from neuradar_amd.neurad_encoding import ActorSettings, NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.rays import RaySamples

    g = load_golden("actors")
    actors = DynamicActors.from_state(g["actor_positions"], g["actor_rotations_6d"], g["actor_timestamps"],
                                      g["actor_present"], g["actor_sizes"]).to(DEV)
    assert actors.n_actors == 2
    grid_cfg = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=int(g["log2t"])),
                                        actor=ActorSettings(flip_prob=0.25, log2_hashmap_size=int(g["actor_log2t"])))
    fld = NeuRADFieldConfig(grid=grid_cfg).setup(actors=actors, static_scale=100.0).to(DEV)
    with torch.no_grad():
        fld.hashgrid.static_grid.hash_table.copy_(g["table"])
        for i, gr in enumerate(fld.hashgrid.actor_grids):
            gr.hash_table.copy_(g[f"actor{1 - i}_table"])  # swapped
        for i, l in enumerate(fld.mlp_geo.layers):
            l.weight.copy_(g[f"geo_w{i}"]); l.bias.copy_(g[f"geo_b{i}"])
        for i, l in enumerate(fld.mlp_feature.layers):
            l.weight.copy_(g[f"feat_w{i}"]); l.bias.copy_(g[f"feat_b{i}"])
        fld.sdf_to_density.beta.copy_(g["beta"])
        actors.actor_to_id.copy_(torch.tensor([1, 0]))
    e = g["edges"]
    B = e.shape[0]
    rs = RaySamples(dev(g["origins"]), dev(g["directions"]), dev(g["pixel_area"]), dev(torch.zeros_like(e)), dev(e),
                    dev(torch.zeros(B, 1)), dev(torch.full((B, 1), 1e6)), times=dev(g["times"]))
    fld.train()
    out = fld(rs, flip=dev(g["flip"]))
    assert_close(cpu(out[FieldHeadNames.FEATURE]), g["train_feature"], rtol=1e-4, atol_scale=1e-5, what="train feature")
    assert_close(cpu(out[FieldHeadNames.ALPHA]), g["train_alpha"], rtol=1e-4, atol_scale=1e-5, what="train alpha")
    loss = (out[FieldHeadNames.FEATURE] * dev(g["g_feature"])).sum() + (out[FieldHeadNames.ALPHA] * dev(g["g_alpha"])).sum()
    g0, g1 = torch.autograd.grad(loss, [fld.hashgrid.actor_grids[0].hash_table, fld.hashgrid.actor_grids[1].hash_table])
    assert_close(cpu(g0), g["grad_hashgrid_actor_grids_1_hash_table"], rtol=1e-3, atol_scale=1e-4, what="grad of grid 0 = actor 1's")
    assert_close(cpu(g1), g["grad_hashgrid_actor_grids_0_hash_table"], rtol=1e-3, atol_scale=1e-4, what="grad of grid 1 = actor 0's")
    # without the indirection (identity map on swapped tables) the features differ: the check above is not vacuous
    with torch.no_grad():
        actors.actor_to_id.copy_(torch.tensor([0, 1]))
        wrong = fld(rs, flip=dev(g["flip"]))[FieldHeadNames.FEATURE]
    assert float((cpu(wrong) - g["train_feature"]).abs().max()) > 1e-3 * float(g["train_feature"].abs().max())


# ------------------------------------------------------------------------------------------------ mixed sensors, a19, a20
def test_mixed_sensor_batch_with_appearance_and_lidar_masks_vs_oracle():
    """configs[2]-style batch: camera patch + lidar points + one ZOD radar scan generated on the device,
    pixel-area scaling (a4), appearance embedding (a19), is_close_to_lidar / carving side outputs (a20);
    everything compared with the CPU oracle on the same rays and jitter."""
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.sensors import Cameras, Lidars, Radars, merge_bundles, scale_pixel_area
    from neuradar_amd.step import HotPathConfig, NeuRadarHotPath
    from oracle import field as of, pipeline as op, raygen as orr

    g = load_golden("raygen")
    torch.manual_seed(0)
    cams = Cameras(dev(g["cam_c2w"]), dev(g["cam_fx"]), dev(g["cam_fy"]), dev(g["cam_cx"]), dev(g["cam_cy"]),
                   dev(g["cam_heights"]), dev(g["cam_times_in"]), dev(g["cam_vel"]), dev(g["cam_rs_offsets"]))
    cam_b, _ = cams.generate_patch_rays(torch.rand(1, 3, device=DEV), 8, 3, 1080, 1920)
    lid_b = Lidars(dev(g["lid_l2w"]), dev(g["lid_times_in"]), dev(g["lid_vel"])).generate_rays(dev(g["lid_indices"]), dev(g["lid_points"]))
    rad = Radars(dev(g["rad_r2w"]), dev(g["rad_times_in"]), 0.0625, 0.0625, -0.5, 0.5, -0.5, 0.5)
    rad_b = rad.generate_rays(torch.tensor([1], device=DEV))
    bundle = merge_bundles(cam_b, lid_b, rad_b)
    n_cam, n_lid, n_rad = len(cam_b), len(lid_b), len(rad_b)
    assert n_rad == 256 and len(bundle) == n_cam + n_lid + n_rad
    scale_pixel_area(bundle)
    assert torch.equal(bundle.pixel_area[:n_cam], cam_b.pixel_area * 9) and torch.equal(bundle.pixel_area[n_cam:n_cam + n_lid], lid_b.pixel_area)
    B = len(bundle)
    bundle.times = bundle.times.clamp(0, 19.9)
    bundle.metadata["sensor_idxs"] = torch.cat([torch.zeros(n_cam, 1), torch.ones(n_lid, 1), torch.full((n_rad, 1), 2.0)]).long().to(DEV)
    cfg = HotPathConfig(field=NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=12))),
                        appearance_dim=16, duration=20.0, num_sensors=3)
    cfg.proposal_field_1.grid.static.log2_hashmap_size = 11
    cfg.proposal_field_2.grid.static.log2_hashmap_size = 11
    model = NeuRadarHotPath(cfg).to(DEV).train()
    with torch.no_grad():
        model.field.hashgrid.static_grid.hash_table.mul_(300.0)
        model.proposal_fields[1].hashgrid.static_grid.hash_table.mul_(2000.0)
    t_rand, j1, j2 = torch.rand(B, 129, device=DEV), torch.rand(B, 1, device=DEV), torch.rand(B, 1, device=DEV)
    o, d, area, fars, times = (cpu(x).clone() for x in (bundle.origins, bundle.directions, bundle.pixel_area, bundle.fars, bundle.times))
    out = model.get_nff_outputs(bundle, t_rand=t_rand, jitters=(j1, j2))

    def grid(m):
        gg = m.hashgrid.static_grid
        return of.GridParams(cpu(gg.hash_table), cpu(gg.scalings), gg.log2_hashmap_size)

    fp = of.FieldParams(grid(model.field), [(cpu(l.weight), cpu(l.bias)) for l in model.field.mlp_geo.layers],
                        [(cpu(l.weight), cpu(l.bias)) for l in model.field.mlp_feature.layers],
                        cpu(model.field.sdf_to_density.beta), 100.0)
    pf = model.proposal_fields[1]
    pp = of.ProposalParams(grid(pf), cpu(pf.density_decoder.weight), 100.0)
    md = {k: cpu(v) for k, v in bundle.metadata.items()}
    ref = op.nff_outputs(fp, [pp, pp], {"origins": o, "directions": d, "pixel_area": area, "fars": fars, "times": times,
                                        "sensor_idx": md["sensor_idxs"], "is_lidar": md["is_lidar"],
                                        "directions_norm": md["directions_norm"], "did_return": md["did_return"]},
                         cpu(t_rand), (cpu(j1), cpu(j2)),
                         appearance={"table": cpu(model.appearance_embedding.weight), "duration": 20.0, "embeds_per_sensor": 20})
    assert out["features"].shape == (B, 48)
    assert_close(cpu(out["features"]), ref["features"].detach(), rtol=1e-4, atol_scale=1e-4, what="features+appearance")
    assert_close(cpu(out["depth"]), ref["depth"].detach(), rtol=1e-4, atol_scale=1e-4, what="depth")
    for i in (0, 1):
        assert_close(cpu(out[f"prop_weights_loss_{i}"]), ref[f"prop_weights_loss_{i}"].detach(), rtol=1e-3, atol_scale=1e-4,
                     what=f"prop_weights_loss_{i}")
    # radar sanity on the rendered depths of the radar rays (256-ray scan): finite and inside the sky range
    rd = out["depth"][n_cam + n_lid:]
    assert bool(torch.isfinite(rd).all()) and float(rd.max()) < 20000.0
