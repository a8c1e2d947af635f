"""tiny-cuda-nn-compatible grid.  CPU part: the oracle restatement against what can be known without the
library -- the published parameter count of the instant-ngp default grid, an explicit n-linear interpolation of a dense
array on levels that are indexed densely, partition of unity -- and the FullyFusedMLP parameter layout.  GPU part: the HIP
kernels against the oracle, 3-D and 4-D, values and parameter gradients; the NeuRAD encoding with layout="tcnn" and the
4-D actor grid against a composition of oracle pieces; a tcnn-style state dict loaded into the field."""
import itertools

import pytest
import torch

from oracle import tcnn_grid as tg

DEV = "cuda"


def test_parameter_count_of_the_instant_ngp_default_grid():
    """instant-ngp logs `n_params=12196240` for its default HashGrid (16 levels, 2 features, T=2^19, base 16,
    per_level_scale 1.38191): the offset table (dense levels rounded up to 8, hashed levels capped at T) reproduces it."""
    g = tg.geometry(3, 16, 2, 19, 16, 1.38191)
    assert g.n_params == 12_196_240
    assert g.resolutions[0] == 16 and g.resolutions[-1] == 2048
    assert g.offsets[1] == 4096 and g.offsets[2] - g.offsets[1] == 12168  # 16^3; 23^3 = 12167 -> next multiple of 8


@pytest.mark.parametrize("D", [3, 4])
def test_dense_levels_are_an_nlinear_interpolation_of_a_dense_array(D):
    """Where resolution^D fits the table the index is the row-major position in a [res]^D array (x fastest) and the
    encoding must equal the n-linear interpolation of that array at x * scale + 0.5."""
    torch.manual_seed(D)
    g = tg.geometry(D, 1, 2, 16, 6, 1.0)  # one level, resolution 6: 6^4 = 1296 <= 2^16
    r = g.resolutions[0]
    assert r**D <= g.offsets[1]
    params = torch.randn(g.n_params, dtype=torch.float64)
    x = torch.rand(200, D, dtype=torch.float64) * 0.8  # (cells whose upper corner stays inside the array)
    got = tg.encode(x.float(), params, g).double()
    arr = params.view(-1, 2)[: r**D].view(*([r] * D), 2)  # index = sum c_d r^d: the LAST array axis is x
    pos = (x.float().double() * g.scales[0] + 0.5).float().double()
    cell, w = torch.floor(pos).long(), pos - torch.floor(pos)
    want = torch.zeros(200, 2, dtype=torch.float64)
    for corner in itertools.product((0, 1), repeat=D):
        wt = torch.ones(200, dtype=torch.float64)
        idx = []
        for d in range(D):
            wt = wt * (w[:, d] if corner[d] else 1 - w[:, d])
            idx.append(cell[:, d] + corner[d])
        want += wt[:, None] * arr[tuple(reversed(idx))]
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)


def test_constant_table_gives_constant_features_on_hashed_levels():
    g = tg.geometry(4, 3, 1, 10, 32, 2.0)  # 33^4 > 2^10: every level hashed
    out = tg.encode(torch.rand(64, 4), torch.full((g.n_params,), 0.25), g)
    torch.testing.assert_close(out, torch.full_like(out, 0.25), rtol=1e-5, atol=1e-6)


def test_fully_fused_mlp_layout():
    """[width, pad16(in)] | [width, width] * (hidden - 1) | [pad16(out), width], row-major; y = W_last relu(... relu(W_0 x))."""
    from neuradar_amd.tcnn_compat import fully_fused_mlp_weights

    torch.manual_seed(0)
    in_dim, width, hidden, out_dim = 20, 32, 2, 3
    w0, w1, w2 = torch.randn(width, 32), torch.randn(width, width), torch.randn(16, width)
    params = torch.cat([w0.reshape(-1), w1.reshape(-1), w2.reshape(-1)])
    for unpack in (fully_fused_mlp_weights, tg.mlp_weights):
        a, b, c = unpack(params, in_dim, width, hidden, out_dim)
        assert a.shape == (width, in_dim) and b.shape == (width, width) and c.shape == (out_dim, width)
        assert torch.equal(a, w0[:, :in_dim]) and torch.equal(b, w1) and torch.equal(c, w2[:out_dim])
    with pytest.raises(ValueError):
        fully_fused_mlp_weights(params[:-1], in_dim, width, hidden, out_dim)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(3, 8, 4, 15, 16, 1.6), (4, 4, 4, 12, 64, 2.5198), (3, 16, 2, 19, 16, 1.38191), (4, 2, 1, 8, 4, 1.5),
                                 (3, 3, 8, 10, 5, 2.0)])
def test_hip_tcnn_grid_matches_oracle(cfg):
    """Values (rtol 1e-5) and parameter gradients (rtol 1e-4 of their scale: atomics sum in another order); exact grid
    positions, 0 and 1 included."""
    from neuradar_amd import ops

    D, L, F, log2t, base, pls = cfg
    torch.manual_seed(L + D)
    g = tg.geometry(D, L, F, log2t, base, pls)
    lib, p, st = ops._lib.lib(), ops._p, ops._stream
    assert lib.nr_tcnn_grid_param_count(D, L, F, log2t, base, pls) == g.n_params
    # the library's level geometry: resolutions and offsets exactly the oracle's; the scales to a few float32 ulp (exp2f of
    # two libms) -- the value comparison below runs the oracle ON the library's scales
    import ctypes

    import numpy as np
    sc, rs_, of = np.zeros(L, np.float32), np.zeros(L, np.uint32), np.zeros(L + 1, np.uint32)
    ops.check(lib.nr_tcnn_grid_geometry(D, L, log2t, base, pls, sc.ctypes.data_as(ctypes.c_void_p), rs_.ctypes.data_as(ctypes.c_void_p),
                                        of.ctypes.data_as(ctypes.c_void_p)), "geometry")
    assert list(rs_) == g.resolutions and list(of) == g.offsets
    np.testing.assert_allclose(sc, np.array(g.scales, np.float32), rtol=8 * 1.2e-7)
    g.scales = [float(v) for v in sc]
    n = 3001
    x = torch.rand(n, D)
    x[:4] = torch.tensor([[0.0] * D, [1.0] * D, [0.5] * D, [0.25] * D])
    params = torch.randn(g.n_params, requires_grad=True)
    ref = tg.encode(x, params, g)
    go = torch.randn_like(ref)
    (gref,) = torch.autograd.grad(ref, params, go)
    xd, pd, out = x.to(DEV), params.detach().to(DEV), torch.empty(n, L * F, device=DEV)
    ops.check(lib.nr_tcnn_grid_fwd(p(xd), p(pd), D, L, F, log2t, base, pls, p(out), n, st()), "fwd")
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    gp = torch.zeros(g.n_params, device=DEV)
    ops.check(lib.nr_tcnn_grid_bwd(p(xd), D, L, F, log2t, base, pls, p(go.to(DEV)), p(gp), n, st()), "bwd")
    torch.testing.assert_close(gp.cpu(), gref, rtol=1e-4, atol=1e-5 * float(gref.abs().max()))
    assert torch.equal(gp.cpu() != 0, gref != 0)


@pytest.mark.gpu
def test_tcnn_grid_rejects_unsupported_configurations():
    from neuradar_amd import _lib

    lib = _lib.lib()
    assert lib.nr_tcnn_grid_param_count(2, 4, 2, 15, 16, 2.0) == -1  # 2-D
    assert lib.nr_tcnn_grid_param_count(3, 33, 2, 15, 16, 2.0) == -1  # > 32 levels
    assert lib.nr_tcnn_grid_param_count(3, 4, 2, 15, 16, 0.0) == -1


@pytest.mark.gpu
def test_tcnn_layout_field_with_4d_actor_grid_and_checkpoint_loading():
    """NeuRADField built with layout="tcnn" and the golden's two actors: (i) a reference-style tcnn state dict (grid
    vectors, a FullyFusedMLP vector, plain tensors) loads key by key; (ii) the encoding equals the oracle pieces composed by
    hand -- the static 3-D grid, and for samples inside an actor box the ONE 4-D grid at (box coordinates, actor id /
    n_actors), both with NeuRAD's per-level rescale (neurad_encoding.py:282-293,309-316); (iii) the gradients of both
    parameter vectors equal the oracle's."""
    from helpers import load_golden
    from neuradar_amd import ops
    from neuradar_amd.dynamic_actors import DynamicActors
    from neuradar_amd.neurad_encoding import ActorSettings, NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.rays import RaySamples
    from neuradar_amd.tcnn_compat import load_tcnn_state_dict

    g = load_golden("actors")
    actors = DynamicActors.from_state(g["actor_positions"], g["actor_rotations_6d"], g["actor_timestamps"], g["actor_present"],
                                      g["actor_sizes"])
    st_, ac_ = (StaticSettings(hashgrid_dim=4, num_levels=8, base_res=16, max_res=2048, log2_hashmap_size=12),
                ActorSettings(hashgrid_dim=4, num_levels=4, base_res=8, max_res=64, log2_hashmap_size=10, use_4d_hashgrid=True, flip_prob=0.0))
    cfg = NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=st_, actor=ac_, layout="tcnn"))
    torch.manual_seed(3)
    field = cfg.setup(actors=actors, static_scale=100.0, implementation="hip").to(DEV).eval()
    hg = field.hashgrid
    assert len(hg.actor_grids) == 1 and hg.actor_grids[0].in_dim == 4
    # (i) a tcnn-style state dict
    n_s, n_a = hg.static_grid.tcnn_encoding.params.numel(), hg.actor_grids[0].tcnn_encoding.params.numel()
    sd = {"hashgrid.static_grid.tcnn_encoding.params": torch.randn(n_s), "hashgrid.actor_grids.0.tcnn_encoding.params": torch.randn(n_a),
          "mlp_geo.tcnn_encoding.params": 0.1 * torch.randn(32 * 32 + 48 * 32), "sdf_to_density.beta": torch.tensor([17.0])}
    used = load_tcnn_state_dict(field, sd)
    assert sorted(used) == sorted(sd), (used, list(sd))
    assert float(field.sdf_to_density.beta) == 17.0 and float(field.mlp_geo.layers[0].bias.abs().max()) == 0.0
    assert torch.equal(field.mlp_geo.layers[1].weight.cpu(), sd["mlp_geo.tcnn_encoding.params"][32 * 32:].view(48, 32)[:33])
    # (ii) rays towards the actors
    gen = torch.Generator().manual_seed(11)
    B, S = 64, 48
    pos = g["actor_positions"].reshape(-1, 3)
    tgt = pos[torch.randint(0, pos.shape[0], (B,), generator=gen)] + 0.5 * torch.randn(B, 3, generator=gen)
    o = torch.cat([torch.randn(B, 2, generator=gen) * 2.0, torch.full((B, 1), 1.5)], dim=1)
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    ts = g["actor_timestamps"]
    times = ts.min() + (ts.max() - ts.min()) * torch.rand(B, generator=gen)
    dv = lambda t: t.to(DEV)  # noqa: E731
    nears, fars = torch.zeros(B, 1), torch.full((B, 1), 80.0)
    sp, eu = ops.power_bins(dv(nears), dv(fars), S, None)
    rs = RaySamples(dv(o), dv(d), torch.full((B, 1), 2.25e-6, device=DEV), sp, eu, dv(nears), dv(fars), dv(times)[:, None])
    with torch.no_grad():  # the whole field runs on the tcnn-layout tables (grid -> MFMA MLPs -> heads)
        from neuradar_amd.field_heads import FieldHeadNames

        outs = field(rs)
        assert outs[FieldHeadNames.FEATURE].shape == (B, S, 32) and bool(torch.isfinite(outs[FieldHeadNames.ALPHA]).all())
    feats, strides, _, rows_sm = hg.encode_samples(rs, level_major=False)
    assert not rows_sm and feats.shape == (B * S, 32)
    go = torch.randn(B * S, 32, generator=gen)
    (feats * dv(go)).sum().backward()
    # the same by hand from oracle pieces
    x01, std01 = ops.contract_gaussians(rs.origins, rs.directions, rs.pixel_area, rs.euclid, 100.0, sample_major_rows=False)
    geom = hg.actor_geometry(rs, None)
    lib, p = ops._lib.lib(), ops._p
    n = B * S
    slot = torch.empty(n, device=DEV, dtype=torch.int32)
    x01a, std01a = torch.empty(n, 3, device=DEV), torch.empty(n, device=DEV)
    ops.check(lib.nr_actor_assign(p(rs.origins.contiguous()), p(rs.directions.contiguous()), p(rs.pixel_area.reshape(-1).contiguous()),
                                  p(rs.euclid.contiguous()), B, S, 0, p(geom["cand"]), hg.MAX_CANDIDATES, p(geom["w2b"].detach()),
                                  p(geom["centres"]), p(geom["bounds"]), ac_.actor_scale, None, p(slot), p(x01a), p(std01a), None,
                                  ops._stream()), "assign")
    inside = (slot >= 0).cpu()
    assert 0.02 < float(inside.float().mean()) < 0.9, "the test rays must cross actor boxes"
    actor = geom["cand"][torch.arange(n, device=DEV) // S, slot.clamp(min=0).long()].long().cpu()
    assert not hasattr(actors, "actor_to_id") or actors.actor_to_id is None or torch.equal(actors.actor_to_id.cpu(), torch.arange(actors.n_actors))
    g3 = tg.geometry(3, st_.num_levels, 4, st_.log2_hashmap_size, st_.base_res, float(hg.static_grid.growth_factor))
    g4 = tg.geometry(4, ac_.num_levels, 4, ac_.log2_hashmap_size, ac_.base_res, float(hg.actor_grids[0].growth_factor))
    ps = sd["hashgrid.static_grid.tcnn_encoding.params"].clone().requires_grad_(True)
    pa = sd["hashgrid.actor_grids.0.tcnn_encoding.params"].clone().requires_grad_(True)

    def rescale(f, std, scalings):
        w = 1.0 / torch.clamp(2.0 * scalings[None, :] * std[:, None], min=1.0)
        return (f.view(n, scalings.numel(), -1) * w[:, :, None]).view(n, -1)

    fs = rescale(tg.encode(x01.cpu(), ps, g3), std01.cpu(), hg.static_grid.scalings.cpu())
    pos4 = torch.cat([x01a.cpu(), (actor.float() / actors.n_actors)[:, None]], dim=-1)
    # (rows outside every box carry whatever the buffers held: NaN there would reach pa.grad as 0 * NaN through `where`)
    std_a = torch.where(inside, std01a.cpu(), torch.ones_like(std01a.cpu()))
    fa = rescale(tg.encode(torch.where(inside[:, None], pos4, torch.zeros_like(pos4)), pa, g4), std_a, hg.actor_grids[0].scalings.cpu())
    want = torch.where(inside[:, None], torch.nn.functional.pad(fa, (0, 16)), fs)  # 4 actor levels of 8: zero-padded (:186)
    torch.testing.assert_close(feats.detach().cpu(), want.detach(), rtol=1e-4, atol=1e-5)
    (want * go).sum().backward()
    for got, ref, what in ((hg.static_grid.tcnn_encoding.params.grad, ps.grad, "static"), (hg.actor_grids[0].tcnn_encoding.params.grad, pa.grad, "4-D actor")):
        assert float(ref.abs().max()) > 0
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()), msg=lambda m, w=what: f"{w} grid gradient: {m}")
