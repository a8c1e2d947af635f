"""The RGB decoder's pointwise convolutions (pw.hip: nr_pw_fwd / nr_pw_bwd_data / nr_pw_bwd_weight; reference:
models/neuradar.py:225-240 -- Conv2d(48, 32, 1) + ReLU, ConvTranspose2d(32, 32, 3, stride=3), Conv2d(32, 3, 1) + Sigmoid) against
torch's convolutions in fp32 on the same 16-bit operands: output, input gradient, weight and bias gradient; ragged pixel counts;
accumulation into existing gradient buffers; the loss-scale factor on the input gradient."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
U = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def _close(got, want, u, what, k=2.0):
    scale = float(want.abs().mean())
    err = (got.float() - want).abs()
    bound = k * u * want.abs() + k * u * scale
    assert bool((err <= bound).all()), f"{what}: worst excess {float((err - bound).max()):.3e} at scale {scale:.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("P,H,W", [(8, 32, 32), (1, 5, 7), (3, 1, 1)])
def test_head_conv1x1_relu_from_fp32_rows(dtype, P, H, W):
    from neuradar_amd import ops

    torch.manual_seed(P + H)
    u = U[dtype]
    x = torch.randn(P * H * W, 48, device=DEV, requires_grad=True)
    w = (torch.randn(32, 48, 1, 1, device=DEV) / 7.0).to(dtype).requires_grad_(True)
    b = torch.randn(32, device=DEV).to(dtype).requires_grad_(True)
    scale = torch.full((1,), 0.25, device=DEV)
    y = ops.pointwise(x, w, b, act=1, grad_scale=scale)
    assert y.dtype == dtype and y.shape == (P * H * W, 32)
    xr = x.detach().clone().requires_grad_(True)
    wr, br = w.detach().float().requires_grad_(True), b.detach().float().requires_grad_(True)
    ref = torch.relu(F.linear(xr, wr.view(32, 48), br))
    _close(y.detach(), ref.detach(), u, "output")
    g = torch.randn_like(ref).to(dtype)
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], g)
    # the reference's gradient flows where the 16-bit output is positive (what the kernel sees)
    mask = (y.detach().float() > 0).float()
    rx, rw, rb = torch.autograd.grad(F.linear(xr, wr.view(32, 48), br), [xr, wr, br], g.float() * mask)
    _close(gx, 0.25 * rx, 1e-5, "d x (fp32, times the scale)", k=4.0)
    _close(gw.float(), rw, u, "d weight")
    _close(gb.float(), rb, u, "d bias")


@pytest.mark.parametrize("mfma", ["1", "0"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("P,H,W", [(8, 32, 32), (2, 3, 5), (1, 7, 9)])
def test_transposed_convolution_3x3_stride_3(request, dtype, P, H, W, mfma):
    """Forward and data gradient on the matrix cores (nine 32 x 32 GEMMs per 32 pixels) and on the generic pointwise kernels
    (nr_set_tuning NR_TUNE_PW_MFMA_OFF); the weight gradient is the generic kernel's in both."""
    from neuradar_amd import _lib, ops

    _lib.set_tuning("NR_PW_MFMA_OFF", 0 if mfma == "1" else 1)
    request.addfinalizer(lambda: _lib.set_tuning("NR_PW_MFMA_OFF", 0))
    torch.manual_seed(P * H)
    u = U[dtype]
    x = torch.randn(P, 32, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(32, 32, 3, 3, device=DEV) / 6.0).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(32, device=DEV).to(dtype).requires_grad_(True)
    y = ops.conv_transpose3(x, w, b)
    assert y.shape == (P, 32, 3 * H, 3 * W) and y.is_contiguous(memory_format=torch.channels_last)
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    ref = F.conv_transpose2d(xr, wr, br, stride=3)
    _close(y.detach(), ref.detach(), u, "output")
    g = torch.randn_like(ref).to(dtype).contiguous(memory_format=torch.channels_last)
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], g)
    rx, rw, rb = torch.autograd.grad(ref, [xr, wr, br], g.float())
    _close(gx.float(), rx, u, "d x")
    _close(gw.float(), rw, u, "d weight")
    _close(gb.float(), rb, u, "d bias")
    assert gw.stride() == w.stride()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tail_conv1x1_sigmoid_to_fp32_and_accumulation(dtype):
    from neuradar_amd import ops

    torch.manual_seed(5)
    u = U[dtype]
    n = 8 * 96 * 96 + 3
    x = torch.randn(n, 32, device=DEV).to(dtype).requires_grad_(True)
    w = (torch.randn(3, 32, 1, 1, device=DEV) / 5.0).to(dtype).requires_grad_(True)
    b = torch.randn(3, device=DEV).to(dtype).requires_grad_(True)
    y = ops.pointwise(x, w, b, act=2, out_f32=True)
    assert y.dtype == torch.float32 and y.shape == (n, 3)
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    ref = torch.sigmoid(F.linear(xr, wr.view(3, 32), br))
    assert float((y.detach() - ref.detach()).abs().max()) < 1e-5
    g = torch.randn_like(ref) / 64.0  # (a loss-scaled gradient: d x stays inside fp16's normal range)
    rx, rw, rb = torch.autograd.grad(ref, [xr, wr, br], g)
    # accumulation into preallocated 16-bit .grad buffers (the training step's mode): twice the gradient after two backwards
    w.grad, b.grad = torch.zeros_like(w), torch.zeros_like(b)
    with ops.direct_param_grads():
        for _ in range(2):
            (gx,) = torch.autograd.grad(ops.pointwise(x, w, b, act=2, out_f32=True), [x], g)
    _close(gx.float(), rx, u, "d x")
    _close(w.grad.float(), 2.0 * rw, u, "accumulated d weight", k=3.0)
    _close(b.grad.float(), 2.0 * rb, u, "accumulated d bias", k=3.0)
