"""CPU checks of oracle/field_lp.py (the reduced-precision restatement the GPU tests of mlp_lp.hip lean on)."""
import numpy as np
import torch

from helpers import load_golden, oracle_field_params
from oracle import field as of, field_lp


def _setup(hidden=64, n=500, seed=0):
    g = torch.Generator().manual_seed(seed)

    def lin(o, i):
        k = 1 / np.sqrt(i)
        return ((torch.rand(o, i, generator=g) * 2 - 1) * k).requires_grad_(True), ((torch.rand(o, generator=g) * 2 - 1) * k).requires_grad_(True)

    geo, feat = [lin(hidden, 32), lin(33, hidden)], [lin(hidden, 48), lin(hidden, hidden), lin(32, hidden)]
    feats = (torch.randn(n, 32, generator=g) * 0.5).requires_grad_(True)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    return g, geo, feat, feats, dirs, torch.tensor([3.0], requires_grad=True)


def test_handwritten_backward_equals_autograd_without_rounding():
    """dtype "float32" switches every rounding off: the hand-written backward must then be the autograd gradient of the
    fp32 oracle (oracle/field.py, itself pinned by the reference goldens)."""
    g, geo, feat, feats, dirs, beta = _setup()
    n = feats.shape[0]
    gf, ga = torch.randn(n, 32, generator=g), torch.randn(n, generator=g)
    det = lambda ls: [(w.detach(), b.detach()) for w, b in ls]  # noqa: E731
    out = field_lp.field_mlp_lp(feats.detach(), dirs, det(geo), det(feat), beta.detach(), "float32", gf, ga, 8.0)
    h = of.mlp(feats, geo)
    sdf, e = h[:, 0], h[:, 1:]
    feature = e + of.mlp(torch.cat([e, of.direction_encoding(dirs)], -1), feat)
    alpha = of.sigmoid_density(sdf, beta)
    ps = [feats, geo[0][0], geo[1][0], geo[0][1], geo[1][1], feat[0][0], feat[1][0], feat[2][0], feat[0][1], feat[1][1], feat[2][1], beta]
    names = ["g_feats", "g_geo_w0", "g_geo_w1", "g_geo_b0", "g_geo_b1", "g_feat_w0", "g_feat_w1", "g_feat_w2", "g_feat_b0",
             "g_feat_b1", "g_feat_b2", "g_beta"]
    grads = torch.autograd.grad((feature * gf).sum() + (alpha * ga).sum(), ps)
    torch.testing.assert_close(out["feature"], feature.detach(), rtol=1e-5, atol=1e-6)
    for k, v in zip(names, grads):
        torch.testing.assert_close(out[k], v, rtol=1e-4, atol=1e-5 * float(v.abs().max()))


def test_rounding_oracle_vs_reference_autocast_golden():
    """The restatement against the reference itself under torch.autocast (field_autocast.npz): forward values within
    2.5 unit roundoffs of the tensor's scale (the reference additionally rounds every layer OUTPUT to 16 bit)."""
    for tag in ("field_neurad", "field_l16f2w64"):
        g, ga = load_golden(tag), load_golden("field_autocast")
        p = oracle_field_params(g)
        e = g["edges"]
        B, S = e.shape[0], e.shape[1] - 1
        mean, std = of.isotropic_gaussian(g["origins"], g["directions"], e[:, :-1], e[:, 1:], g["pixel_area"])
        feats = of.static_grid_features(mean, std, p.grid, p.static_scale).reshape(B * S, -1)
        dirs = g["directions"][:, None, :].expand(B, S, 3).reshape(-1, 3)
        for dt, u in (("bfloat16", 2.0 ** -8), ("float16", 2.0 ** -11)):
            out = field_lp.field_mlp_lp(feats, dirs, p.geo, p.feat, p.beta, dt)
            for k in ("feature", "sdf", "alpha"):
                want = ga[f"{tag[6:]}_{dt}_{k}"].reshape(out[k].shape)
                assert float((out[k] - want).abs().max() / want.abs().max()) < 2.5 * u, (tag, dt, k)
