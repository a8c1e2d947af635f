"""The RGB decoder's 7 x 7 convolutions on the matrix cores (conv7.hip: nr_conv7_pack / nr_conv7_fwd; reference:
model_components/cnns.py:21-47 as instantiated by models/neuradar.py:225-240 -- Conv2d(32, 32, 7, padding=3)) against torch's
convolution in fp32 on the same 16-bit operands: forward, the data gradient through the flipped / transposed weight image, the
fused ReLU / residual epilogue, ragged and tiny images, and a whole eval-mode BasicBlock with its batch norms folded in."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
U = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def _close(got, want, u, what):
    """|got - want| <= 2u |want| + 2u * (typical magnitude): the output is rounded once to 16 bits (u relative), the fp32
    accumulation of 1 568 products differs from torch's in order only (1e-6)."""
    scale = float(want.abs().mean())
    err = (got.float() - want).abs()
    bound = 2 * u * want.abs() + 2 * u * scale
    assert bool((err <= bound).all()), f"{what}: worst excess {float((err - bound).max()):.3e} at scale {scale:.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("P,H,W", [(2, 32, 32), (1, 96, 96), (1, 17, 45), (3, 5, 3), (1, 8, 33)])
def test_conv7_forward_and_data_gradient_vs_torch(dtype, P, H, W):
    from neuradar_amd import ops

    torch.manual_seed(P * 1000 + H * 10 + W)
    x = torch.randn(P, 32, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(32, 32, 7, 7, device=DEV) / 40.0).to(dtype)
    b = torch.randn(32, device=DEV).to(dtype)
    # the parameter's channels-last memory [O, kh, kw, I] + the bias behind it, as they sit in the optimizer's flat buffer
    flat = torch.cat([w.permute(0, 2, 3, 1).reshape(-1), b])
    images = ops.conv7_pack(flat, [0], [w.numel()])
    y = ops.conv7_forward(x, images[0, 0])
    xr = x.float().requires_grad_(True)
    ref = F.conv2d(xr, w.float(), b.float(), padding=3)
    _close(y, ref.detach(), U[dtype], f"forward {dtype} {P}x{H}x{W}")
    assert y.is_contiguous(memory_format=torch.channels_last) and y.dtype == dtype
    g = torch.randn_like(ref).to(dtype).contiguous(memory_format=torch.channels_last)
    (gx_ref,) = torch.autograd.grad(ref, xr, g.float())
    gx = ops.conv7_forward(g, images[0, 1])  # the data-gradient image: flipped taps, channels swapped, no bias
    _close(gx, gx_ref, U[dtype], f"data gradient {dtype} {P}x{H}x{W}")
    # fused epilogue: ReLU, residual + ReLU
    _close(ops.conv7_forward(x, images[0, 0], None, relu=True), torch.relu(ref.detach()), U[dtype], "relu")
    res = torch.randn_like(x)
    want = torch.relu(ref.detach() + res.float())
    _close(ops.conv7_forward(x, images[0, 0], res, relu=True), want, U[dtype], "residual + relu")


def test_conv7_autograd_function_matches_torch_conv():
    """ops.conv7 (forward + data gradient on the kernel, weight / bias gradient from the library) against torch's Conv2d on the
    same 16-bit parameters: output, d x, d weight, d bias."""
    from neuradar_amd import ops

    torch.manual_seed(1)
    dtype, u = torch.float16, U[torch.float16]
    x = torch.randn(2, 32, 24, 40, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    flat = torch.cat([(torch.randn(32, 7, 7, 32, device=DEV) / 40.0).reshape(-1), torch.randn(32, device=DEV)]).to(dtype)
    w = flat[:32 * 49 * 32].view(32, 7, 7, 32).permute(0, 3, 1, 2).detach().requires_grad_(True)  # logical [O, I, 7, 7], channels-last memory
    b = flat[32 * 49 * 32:].detach().requires_grad_(True)
    images = ops.conv7_pack(flat, [0], [32 * 49 * 32])
    y = ops.conv7(x, w, b, images[0])
    g = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], g)
    xr, wr, br = x.detach().float().requires_grad_(True), w.detach().float().requires_grad_(True), b.detach().float().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=3)
    rx, rw, rb = torch.autograd.grad(ref, [xr, wr, br], g.float())
    _close(y.detach(), ref.detach(), u, "output")
    _close(gx, rx, u, "d x")
    rel = lambda a, b_: float((a.float() - b_).norm() / b_.norm())  # noqa: E731
    assert rel(gw, rw) < 2 * u and rel(gb, rb) < 2 * u, (rel(gw, rw), rel(gb, rb))  # (fp32 sums over 1 920 pixels, rounded once)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("P,H,W", [(8, 32, 32), (2, 96, 96), (1, 17, 45), (3, 5, 3), (70, 8, 32)])
def test_conv7_weight_gradient_vs_torch(dtype, P, H, W):
    """nr_conv7_wgrad (operands read column-wise from LDS with ds_read_b64_tr_b16, per-block partials + a reduce launch): every
    entry of d weight [32, 32, 7, 7] and d bias against torch's convolution backward in fp32 on the same 16-bit operands -- at
    the training step's sizes (8 patches of 32 x 32 / 96 x 96), ragged and tiny images, and more tiles than blocks (70 x 1)."""
    from neuradar_amd import ops

    torch.manual_seed(P + H + W)
    x = torch.randn(P, 32, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    g = (torch.randn(P, 32, H, W, device=DEV) / 8.0).to(dtype).contiguous(memory_format=torch.channels_last)
    gw, gb = ops.conv7_wgrad(x, g)
    w = torch.zeros(32, 32, 7, 7, device=DEV, requires_grad=True)
    b = torch.zeros(32, device=DEV, requires_grad=True)
    ref = F.conv2d(x.float(), w, b, padding=3)
    rw, rb = torch.autograd.grad(ref, [w, b], g.float())
    assert gw.shape == (32, 32, 7, 7) and gw.stride() == (32 * 49, 1, 7 * 32, 32)
    _close(gw, rw, U[dtype], f"d weight {dtype} {P}x{H}x{W}")
    _close(gb, rb, U[dtype], f"d bias {dtype} {P}x{H}x{W}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rendering_blocks_with_folded_batch_norm_equal_the_eval_modules(dtype):
    """Decoders.prepare_conv7_eval: an eval-mode BasicBlock as two launches (batch norms folded into the convolutions' weights
    and biases, ReLU / residual in the epilogue) against the torch modules in fp32 -- the whole RGB decoder on a feature image."""
    from neuradar_amd.decoders import Decoders

    torch.manual_seed(4)
    dec = Decoders(n_features=48).to(DEV).eval()
    with torch.no_grad():
        for m in dec.rgb_decoder.modules():
            if isinstance(m, torch.nn.BatchNorm2d):  # statistics of a trained model: not the identity
                m.running_mean.normal_(0.0, 0.3)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.2)
        x = torch.randn(1, 48, 20, 37, device=DEV).contiguous(memory_format=torch.channels_last)
        dec.prepare_conv7_eval(None)
        ref = dec.rgb_decoder(x)
        with torch.autocast("cuda", dtype=dtype):
            plain = dec.rgb_decoder(x).float()
        dec.prepare_conv7_eval(dtype)
        assert all(b.conv7_eval is not None for b in dec.rgb_decoder.modules() if hasattr(b, "conv7_eval"))
        with torch.autocast("cuda", dtype=dtype):
            got = dec.rgb_decoder(x).float()
    assert got.shape == ref.shape == (1, 3, 60, 111)
    e_conv7, e_plain = float((got - ref).abs().max()), float((plain - ref).abs().max())
    print(f"{dtype}: max |rgb - fp32| with conv7 {e_conv7:.3e}, with the library's autocast convolutions {e_plain:.3e}")
    # eleven 16-bit layers in a row: a few units of u on values in (0, 1); never worse than twice what autocast itself does
    assert e_conv7 <= max(2.0 * e_plain, 8 * U[dtype]), (e_conv7, e_plain)
