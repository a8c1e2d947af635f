"""nerfacc / renderer drop-ins of SURVEY 8b (models/neuradar.py:1010-1023, models/neurad.py:721-728,
model_components/renderers.py:59-90,322-350) through the C ABI against oracle/render.py (values and gradients)."""
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _leaf(x):
    return x.clone().requires_grad_(True)


@pytest.mark.parametrize("B,S", [(257, 32), (33, 64), (5, 1), (19, 200), (1, 129)])
def test_render_weight_from_alpha_and_density_vs_oracle(B, S):
    from neuradar_amd import renderers
    from oracle import render as orc

    g = torch.Generator().manual_seed(S * 1000 + B)
    alphas = torch.rand(B, S, generator=g)
    alphas[0, : min(S, 3)] = 0.0  # transmittance where alpha = 0 is still prod(1 - alpha_j), not 1
    if S > 4:
        alphas[1 % B, 2] = 1.0  # opaque sample: everything behind it gets T = 0, gradients stay finite
    gw, gT = torch.randn(B, S, generator=g), torch.randn(B, S, generator=g)

    a_ref = _leaf(alphas)
    w_ref, T_ref = orc.render_weight_from_alpha(a_ref)
    ((w_ref * gw).sum() + (T_ref * gT).sum()).backward()
    a_hip = _leaf(alphas.cuda())
    w, T = renderers.render_weight_from_alpha(a_hip)
    ((w * gw.cuda()).sum() + (T * gT.cuda()).sum()).backward()
    assert_close(w.cpu(), w_ref.detach(), what="weights")
    assert_close(T.cpu(), T_ref.detach(), what="transmittance")
    assert_close(a_hip.grad.cpu(), a_ref.grad, rtol=2e-4, what="grad alphas")
    assert float(T[0, 0]) == 1.0 and (S < 4 or B == 1 or float(T[0, 3].cpu()) == 1.0)  # (B == 1: row 0 holds the opaque sample)

    sig = 3.0 * torch.rand(B, S, generator=g)
    ts = torch.sort(50.0 * torch.rand(B, S + 1, generator=g), dim=-1).values
    s_ref = _leaf(sig)
    w_ref, T_ref, al_ref = orc.render_weight_from_density(ts[:, :-1], ts[:, 1:], s_ref)
    ga = torch.randn(B, S, generator=g)
    ((w_ref * gw).sum() + (T_ref * gT).sum() + (al_ref * ga).sum()).backward()
    s_hip = _leaf(sig.cuda())
    w, T, al = renderers.render_weight_from_density(ts[:, :-1].cuda(), ts[:, 1:].cuda(), s_hip)
    ((w * gw.cuda()).sum() + (T * gT.cuda()).sum() + (al * ga.cuda()).sum()).backward()
    assert_close(w.cpu(), w_ref.detach(), what="weights (density)")
    assert_close(T.cpu(), T_ref.detach(), what="transmittance (density)")
    assert_close(al.cpu(), al_ref.detach(), what="alphas (density)")
    assert_close(s_hip.grad.cpu(), s_ref.grad, rtol=2e-4, what="grad sigmas")


@pytest.mark.parametrize("B,S,C", [(130, 32, 32), (7, 31, 48), (3, 70, 3), (65, 128, 1)])
def test_accumulate_and_renderers_vs_oracle(B, S, C):
    from neuradar_amd import renderers
    from oracle import render as orc

    g = torch.Generator().manual_seed(B + S + C)
    w, v, go = torch.rand(B, S, generator=g), torch.randn(B, S, C, generator=g), torch.randn(B, C, generator=g)
    w_ref, v_ref = _leaf(w), _leaf(v)
    out_ref = orc.accumulate_along_rays(w_ref, v_ref)
    (out_ref * go).sum().backward()
    w_hip, v_hip = _leaf(w.cuda()), _leaf(v.cuda())
    out = renderers.accumulate_along_rays(w_hip, v_hip, None, None)
    (out * go.cuda()).sum().backward()
    assert_close(out.cpu(), out_ref.detach(), what="accumulate")
    assert_close(w_hip.grad.cpu(), w_ref.grad, what="grad weights")
    assert_close(v_hip.grad.cpu(), v_ref.grad, what="grad values")
    # the nerfstudio renderer modules on the same kernels (weights [B,S,1])
    assert_close(renderers.FeatureRenderer()(v.cuda(), w.cuda()[..., None]).cpu(), out_ref.detach(), what="FeatureRenderer")
    acc = renderers.AccumulationRenderer()(w.cuda()[..., None])
    assert acc.shape == (B, 1)
    assert_close(acc.cpu(), orc.accumulate_along_rays(w), what="AccumulationRenderer")
    w2 = _leaf(w.cuda())
    renderers.accumulate_along_rays(w2).sum().backward()
    assert torch.equal(w2.grad.cpu(), torch.ones(B, S))


def test_renderer_dropins_empty_and_packed_branch():
    from neuradar_amd import renderers

    z = torch.zeros(0, 32, device="cuda")
    w, T = renderers.render_weight_from_alpha(z)
    assert w.shape == (0, 32) and T.shape == (0, 32)
    assert renderers.accumulate_along_rays(z, torch.zeros(0, 32, 4, device="cuda")).shape == (0, 4)
    with pytest.raises(NotImplementedError):
        renderers.render_weight_from_alpha(torch.rand(8, device="cuda"), ray_indices=torch.zeros(8, dtype=torch.long, device="cuda"))
    with pytest.raises(RuntimeError):
        renderers.render_weight_from_alpha(torch.rand(4, 8))  # CPU tensor: no fallback
