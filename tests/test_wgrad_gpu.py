"""Weight-gradient kernels of the training step (kpf_conv2d_wgrad_f32, kpf_dwconv7_wgrad_f32, kpf_dwconv7_f32) against
torch's CPU float64 convolution_backward on the same seeded inputs.  fp32 tolerance: 2e-5 of the gradient's range
(sums of up to 1.3e5 products, different summation order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x_nhwc, dy_nhwc, wshape, stride, pad, groups=1):
    x = x_nhwc.double().cpu().permute(0, 3, 1, 2).requires_grad_(False)
    dy = dy_nhwc.double().cpu().permute(0, 3, 1, 2)
    w = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(wshape[0], dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, b, stride=stride, padding=pad, groups=groups)
    assert y.shape == dy.shape, (y.shape, dy.shape)
    y.backward(dy)
    return w.grad, b.grad


CASES = [  # B, H, W, Cin, N, k, stride, pad
    (2, 8, 8, 384, 1536, 1, 1, 0),     # pwconv1 at stage 3
    (2, 8, 8, 1536, 384, 1, 1, 0),     # pwconv2
    (4, 32, 32, 96, 384, 1, 1, 0),
    (4, 16, 16, 64, 64, 3, 1, 1),      # Residual conv2
    (3, 5, 7, 8, 12, 3, 1, 1),         # ragged everything
    (2, 16, 16, 96, 192, 2, 2, 0),     # ConvNeXt downsample (patchify)
    (2, 12, 12, 16, 48, 4, 4, 0),
    (1, 1, 21, 128, 512, 1, 1, 0),     # a Linear over 21 tokens
    (5, 9, 9, 4, 4, 1, 1, 0),
    (2, 16, 16, 144, 72, 1, 1, 0),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_wgrad_matches_torch(case):
    from keypointfusion_amd.training import conv_wgrad_hip
    B, H, W, Cin, N, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    OH = (H + 2 * pad - k) // stride + 1
    OW = (W + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, W, Cin, generator=g)
    dy = torch.randn(B, OH, OW, N, generator=g)
    dw, db = conv_wgrad_hip(dy.cuda(), x.cuda(), (N, Cin, k, k), stride, pad, True)
    rw, rb = _ref(x, dy, (N, Cin, k, k), stride, pad)
    assert dw.shape == rw.shape
    assert float((dw.cpu().double() - rw).abs().max()) <= 2e-5 * float(rw.abs().max())
    assert float((db.cpu().double() - rb).abs().max()) <= 2e-5 * max(float(rb.abs().max()), 1.0)
    dw2, _ = conv_wgrad_hip(dy.cuda(), x.cuda(), (N, Cin, k, k), stride, pad, False)
    assert torch.equal(dw, dw2), "fixed-order reduction: run-to-run bit-identical"


def test_conv_wgrad_large_pixel_count():
    from keypointfusion_amd.training import conv_wgrad_hip
    g = torch.Generator().manual_seed(7)
    x = torch.randn(32, 64, 64, 48, generator=g)
    dy = torch.randn(32, 64, 64, 128, generator=g)
    dw, db = conv_wgrad_hip(dy.cuda(), x.cuda(), (128, 48, 1, 1), 1, 0, True)
    rw = (dy.double().view(-1, 128).t() @ x.double().view(-1, 48)).view(128, 48, 1, 1)
    assert float((dw.cpu().double() - rw).abs().max()) <= 2e-5 * float(rw.abs().max())
    assert float((db.cpu().double() - dy.double().view(-1, 128).sum(0)).abs().max()) <= 2e-5 * 2000


@pytest.mark.parametrize("shape", [(2, 32, 32, 96), (3, 16, 16, 192), (2, 8, 8, 384), (2, 4, 4, 768), (1, 5, 11, 8)])
def test_dwconv7_forward_backward_match_torch(shape):
    from keypointfusion_amd.training import dwconv7_nhwc
    B, H, W, Cc = shape
    g = torch.Generator().manual_seed(B * H + Cc)
    x = torch.randn(B, H, W, Cc, generator=g)
    w = torch.randn(Cc, 1, 7, 7, generator=g) * 0.2
    b = torch.randn(Cc, generator=g)
    dy = torch.randn(B, H, W, Cc, generator=g)
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = dwconv7_nhwc(xd, wd, bd)
    y.backward(dy.cuda())
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=3, groups=Cc)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    tol = lambda r: 2e-5 * max(float(r.detach().abs().max()), 1e-3)
    assert float((y.detach().cpu().double() - yr.detach().permute(0, 2, 3, 1)).abs().max()) <= tol(yr)
    assert float((xd.grad.cpu().double() - xr.grad.permute(0, 2, 3, 1)).abs().max()) <= tol(xr.grad)
    assert float((wd.grad.cpu().double() - wr.grad).abs().max()) <= tol(wr.grad)
    assert float((bd.grad.cpu().double() - br.grad).abs().max()) <= tol(br.grad)


@pytest.mark.parametrize("M,Cc,relu", [(32 * 32 * 32, 128, True), (2048, 384, True), (4097, 48, True), (37, 8, False), (512, 576, False),
                                        (32 * 16 * 16, 144, True)])
def test_batchnorm_relu_rows_matches_torch(M, Cc, relu):
    from keypointfusion_amd.training import batchnorm_relu_rows
    g = torch.Generator().manual_seed(M + Cc)
    x = torch.randn(M, Cc, generator=g) * 2.0 + 5.0 * torch.randn(Cc, generator=g)  # per-channel means well away from zero
    w = torch.rand(Cc, generator=g) + 0.5
    b = torch.randn(Cc, generator=g)
    dy = torch.randn(M, Cc, generator=g)
    rm, rv = torch.randn(Cc, generator=g), torch.rand(Cc, generator=g) + 0.5
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    rmd, rvd = rm.cuda(), rv.cuda()
    y = batchnorm_relu_rows(xd, wd, bd, rmd, rvd, 0.1, 1e-5, relu)
    y.backward(dy.cuda())
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    rmr, rvr = rm.double(), rv.double()
    yr = F.batch_norm(xr, rmr, rvr, wr, br, True, 0.1, 1e-5)
    if relu:
        yr = F.relu(yr)
    yr.backward(dy.double())
    tol = lambda r: 3e-5 * max(float(r.detach().abs().max()), 1e-3)
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) <= tol(yr)
    assert float((rmd.cpu().double() - rmr).abs().max()) <= tol(rmr)
    assert float((rvd.cpu().double() - rvr).abs().max()) <= tol(rvr)
    assert float((xd.grad.cpu().double() - xr.grad).abs().max()) <= tol(xr.grad)
    assert float((wd.grad.cpu().double() - wr.grad).abs().max()) <= tol(wr.grad)
    assert float((bd.grad.cpu().double() - br.grad).abs().max()) <= tol(br.grad)


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 8, 8, 384, 1536, 1, 1, 0), (4, 16, 16, 64, 64, 3, 1, 1), (3, 5, 7, 8, 16, 3, 1, 1), (2, 16, 16, 96, 192, 2, 2, 0),
                                  (1, 1, 21, 128, 512, 1, 1, 0), (2, 16, 16, 144, 72, 1, 1, 0), (32, 32, 32, 48, 128, 1, 1, 0)])
def test_conv_wgrad_16bit_operands_match_fp64_on_the_rounded_values(case, tdt):
    """kpf_conv2d_wgrad_h16: dY and X in 16-bit storage, fp32 products and sums — equal to the fp64 gradient of the ROUNDED operands to
    fp32 accumulation accuracy (nothing else is rounded)."""
    from keypointfusion_amd.training import conv_wgrad_hip
    B, H, W, Cin, N, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    OH = (H + 2 * pad - k) // stride + 1
    OW = (W + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, W, Cin, generator=g).to(tdt)
    dy = torch.randn(B, OH, OW, N, generator=g).to(tdt)
    dw, db = conv_wgrad_hip(dy.cuda(), x.cuda(), (N, Cin, k, k), stride, pad, True)
    assert dw.dtype == torch.float32
    rw, rb = _ref(x.float(), dy.float(), (N, Cin, k, k), stride, pad)
    assert float((dw.cpu().double() - rw).abs().max()) <= 2e-5 * float(rw.abs().max())
    assert float((db.cpu().double() - rb).abs().max()) <= 2e-5 * max(float(rb.abs().max()), 1.0)


@pytest.mark.parametrize("xdt,ydt", [(torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32),
                                     (torch.float16, torch.float16), (torch.float32, torch.float16)])
@pytest.mark.parametrize("M,Cc,relu", [(4096, 128, True), (777, 48, False)])
def test_batchnorm_rows_mixed_storage_types(M, Cc, relu, xdt, ydt):
    """16-bit input and / or output rows: statistics and arithmetic are fp32 on the stored values; only the written y / dx are
    rounded to their storage type (2^-8 bf16, 2^-11 f16)."""
    from keypointfusion_amd.training import batchnorm_relu_rows
    g = torch.Generator().manual_seed(M + Cc + 5)
    x = (torch.randn(M, Cc, generator=g) * 2.0 + 3.0 * torch.randn(Cc, generator=g)).to(xdt)
    w, b = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    dy = torch.randn(M, Cc, generator=g).to(ydt)
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    rmd, rvd = rm.cuda(), rv.cuda()
    y = batchnorm_relu_rows(xd, wd, bd, rmd, rvd, 0.1, 1e-5, relu, ydt)
    assert y.dtype == ydt
    y.backward(dy.cuda())
    assert xd.grad.dtype == xdt
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.batch_norm(xr, rm.double(), rv.double(), wr, br, True, 0.1, 1e-5)
    if relu:
        # the mask the kernel uses is the stored (rounded) output's sign; identical except where |y| rounds to zero
        yr = F.relu(yr)
    yr.backward(dy.double())
    eps = {torch.float32: 3e-5, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}
    rel = lambda a, r: float((a.detach().cpu().double() - r.detach()).abs().max()) / max(float(r.detach().abs().max()), 1e-3)
    assert rel(y, yr) <= 1.01 * eps[ydt] + 3e-5
    assert rel(xd.grad, xr.grad) <= 1.01 * eps[xdt] + 3e-5
    assert rel(wd.grad, wr.grad) <= 1e-4 and rel(bd.grad, br.grad) <= 1e-4
    assert rel(rmd, rm.double() * 0.9 + 0.1 * x.double().mean(0)) <= 1e-5


def test_conv_wgrad_full_size_properties():
    """At the training benchmark's own sizes (B = 32, stage-1 / stage-3 ConvNeXt-T shapes and a 3x3 decoder convolution), where a CPU
    reference would take minutes: the gradient is additive over a split of the batch, linear in dY, run-to-run identical, and its
    bias part equals the pixel sum of dY."""
    from keypointfusion_amd.training import conv_wgrad_hip
    g = torch.Generator(device="cuda").manual_seed(11)
    for (B, H, Cin, N, k) in ((32, 32, 96, 384, 1), (32, 8, 1536, 384, 1), (32, 32, 64, 64, 3)):
        x = torch.randn(B, H, H, Cin, device="cuda", generator=g)
        dy = torch.randn(B, H, H, N, device="cuda", generator=g)
        args = ((N, Cin, k, k), 1, k // 2, True)
        dw, db = conv_wgrad_hip(dy, x, *args)
        dw2, db2 = conv_wgrad_hip(dy, x, *args)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
        da, ba = conv_wgrad_hip(dy[:16].contiguous(), x[:16].contiguous(), *args)
        dc, bc = conv_wgrad_hip(dy[16:].contiguous(), x[16:].contiguous(), *args)
        scale = float(dw.abs().max())
        assert float((dw - (da + dc)).abs().max()) <= 2e-5 * scale
        assert float((db - (ba + bc)).abs().max()) <= 2e-5 * float(db.abs().max())
        d3, _ = conv_wgrad_hip(dy * 0.5, x, *args)  # exact: a power-of-two scale commutes with every rounding
        assert torch.equal(d3, dw * 0.5)
        assert float((db - dy.double().sum((0, 1, 2)).float()).abs().max()) <= 2e-5 * float(db.abs().max())
