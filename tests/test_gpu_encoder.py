"""The radar decoder's encoder layer as six launches each way (encoder.hip: nr_encoder_pre_fwd / post_fwd / post_bwd / pre_bwd
around nr_attention_fwd / bwd; reference: detr/models/transformer.py:176-189 forward_pre + the final LayerNorm :66-68) against
the modular layer it replaces (torch layer norms / linears / dropouts around the same attention kernel): output, input
gradient and every parameter gradient; ragged token counts, several scans, the three widths the kernels are built for; the
direct accumulation into .grad buffers; dropout (training) through a directional derivative of the deterministic masked
function and through its dependence on seed and step counter."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _models(C, FF, seed=0, p=0.1):
    from neuradar_amd.decoders import Transformer

    torch.manual_seed(seed)
    m = Transformer(d_model=C, dim_feedforward=FF, dropout=p, attention="hip").to(DEV)
    with torch.no_grad():  # parameters away from their init (layer norms at 1 / 0, biases at 0 would hide mistakes)
        for prm in m.parameters():
            prm.add_(0.2 * torch.randn_like(prm))
    return m


def _run(m, src, pos, wgt, fused, monkeypatch):
    monkeypatch.setenv("NR_FUSED_ENCODER", "1" if fused else "0")
    for prm in m.parameters():
        prm.grad = None
    x = src.clone().requires_grad_(True)
    y = m(x, pos)
    (y * wgt).sum().backward()
    return y.detach(), x.grad.detach(), {k: v.grad.detach().clone() for k, v in m.named_parameters()}


def _close(a, b, rtol, what):
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= rtol * scale + 1e-30, f"{what}: max |diff| {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("C,FF,N,n", [(48, 64, 1, 3531), (48, 64, 3, 301), (32, 64, 2, 64), (64, 64, 1, 97), (48, 64, 1, 1)])
def test_fused_encoder_layer_matches_modular_layer(monkeypatch, C, FF, N, n):
    m = _models(C, FF).eval()
    torch.manual_seed(1)
    src, pos, wgt = (torch.randn(N, n, C, device=DEV) for _ in range(3))
    y0, g0, p0 = _run(m, src, pos, wgt, False, monkeypatch)
    y1, g1, p1 = _run(m, src, pos, wgt, True, monkeypatch)
    _close(y1, y0, 1e-4, "output")
    _close(g1, g0, 1e-3, "input gradient")  # (the attention backward's atomics: summation order)
    for k in p0:
        _close(p1[k], p0[k], 1e-3, "gradient of " + k)


def test_fused_encoder_layer_adds_into_grad_buffers(monkeypatch):
    """Under ops.direct_param_grads the kernels add into the parameters' .grad buffers (the training step's mode): twice the
    same backward = twice the gradient, autograd receives None for the parameters."""
    from neuradar_amd import ops

    m = _models(48, 64).eval()
    torch.manual_seed(2)
    src, pos, wgt = (torch.randn(2, 200, 48, device=DEV) for _ in range(3))
    _, g0, p0 = _run(m, src, pos, wgt, True, monkeypatch)
    for prm in m.parameters():
        prm.grad = torch.zeros_like(prm)
    for _ in range(2):
        x = src.clone().requires_grad_(True)
        with ops.direct_param_grads():
            (m(x, pos) * wgt).sum().backward()
        _close(x.grad, g0, 1e-4, "input gradient")
    for k, v in m.named_parameters():
        _close(v.grad, 2.0 * p0[k], 1e-3, "accumulated gradient of " + k)


def test_fused_encoder_layer_dropout(monkeypatch):
    """Training mode: the masks are a function of (seed, step counter) -- same pair, same output bit for bit; another counter,
    another output -- and on average the layer is the eval layer (inverted dropout keeps expectations: the mean deviation from
    eval over many tokens is small against its spread).  The backward uses the forward's masks: directional derivatives of the
    (deterministic) masked function at p = 0.5, where a mask the backward got wrong would change them by tens of percent;
    central differences in fp32 with relu kinks are good to ~3 % here (tools/probe_encoder_fd.py: the same at p = 0)."""
    from neuradar_amd import ops

    monkeypatch.setenv("NR_FUSED_ENCODER", "1")
    m = _models(48, 64, p=0.1).train()
    lyr = m.encoder.layers[0]
    torch.manual_seed(3)
    src, pos, wgt = (torch.randn(1, 256, 48, device=DEV) for _ in range(3))
    epoch = torch.zeros(1, device=DEV)

    def f(x, e, p=0.1):
        epoch.fill_(e)
        return ops.encoder_layer(x, pos, lyr, m.encoder.norm, p, seed=1234, seed_epoch=epoch)

    with torch.no_grad():
        y_a, y_b, y_c = f(src, 5.0), f(src, 5.0), f(src, 6.0)
        assert torch.equal(y_a, y_b)
        assert float((y_a - y_c).abs().max()) > 1e-3
        y_eval = ops.encoder_layer(src, pos, lyr, m.encoder.norm, 0.0)
        dev = y_a - y_eval
        assert float(dev.abs().mean()) > 1e-3 and abs(float(dev.mean())) < 0.1 * float(dev.std())
    x = src.clone().requires_grad_(True)
    (f(x, 5.0, 0.5).double() * wgt.double()).sum().backward()
    eps = 1e-2
    for trial in range(3):
        d = torch.randn_like(src)
        d /= d.norm()
        with torch.no_grad():
            fp = (f(src + eps * d, 5.0, 0.5).double() * wgt.double()).sum()
            fm = (f(src - eps * d, 5.0, 0.5).double() * wgt.double()).sum()
        num, ana = float((fp - fm) / (2 * eps)), float((x.grad.double() * d.double()).sum())
        assert abs(num - ana) <= 5e-2 * abs(ana) + 2e-2, (trial, num, ana)
