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
    (2, 16, 16, 64, 128, 3, 2, 1),     # ResNet 3x3 / s2
    (2, 16, 16, 64, 128, 1, 2, 0),     # ResNet 1x1 / s2 downsample
    (2, 32, 32, 4, 64, 7, 2, 3),       # ResNet stem
    (2, 15, 17, 8, 16, 3, 2, 1),
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


@pytest.mark.parametrize("shape", [(2, 32, 32, 96), (3, 16, 16, 192), (2, 8, 8, 384), (2, 4, 4, 768), (1, 5, 11, 8), (8, 32, 32, 192), (2, 64, 64, 96), (3, 7, 20, 36),
                                   (1, 3, 70, 12), (5, 9, 24, 100), (1, 2, 150, 8)])
def test_dwconv7_forward_backward_match_torch(shape):
    # (the weight gradient: the LDS-staged kernel for maps up to ~144 columns — every width class of its 8-column segments and channel-block sizes, chunks that
    #  start inside an image — and the register-window kernel beyond)
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
                                  (1, 1, 21, 128, 512, 1, 1, 0), (2, 16, 16, 144, 72, 1, 1, 0), (32, 32, 32, 48, 128, 1, 1, 0),
                                  (2, 9, 9, 264, 136, 1, 1, 0), (2, 14, 14, 8, 64, 7, 2, 3), (1, 3, 3, 8, 8, 1, 1, 0),
                                  (2, 4, 4, 768, 1536, 1, 1, 0), (1, 5, 5, 1096, 1032, 1, 1, 0)])  # (the last two: >= 256 tiles of 64 x 64 -> the unsplit 64-tile form)
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


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, N, k, stride, pad
    (2, 16, 16, 96, 384, 1, 1, 0),
    (2, 12, 10, 64, 48, 3, 1, 1),
    (1, 16, 16, 48, 105, 1, 1, 0),    # N not a multiple of 4 (heads): the data gradient pads dY's channels
    (2, 16, 16, 96, 192, 2, 2, 0),    # 2x2 / s2 patchify (ConvNeXt downsample)
    (2, 32, 32, 4, 96, 4, 4, 0),      # 4x4 / s4 stem
    (2, 16, 16, 64, 128, 3, 2, 1),    # ResNet stage entry: 3x3 / s2 (data gradient = transposed convolution of the dilated dY)
    (2, 16, 16, 64, 128, 1, 2, 0),    # ResNet downsample: 1x1 / s2
    (2, 32, 32, 4, 64, 7, 2, 3),      # ResNet stem 7x7 / s2 on the channel-padded image
    (2, 15, 17, 8, 16, 3, 2, 1),      # odd sizes
    (1, 10, 9, 8, 8, 3, 2, 0),        # rows / columns the strided window never reaches
])
def test_conv2d_nhwc_autograd_matches_torch(case):
    from keypointfusion_amd.training import conv2d_nhwc
    B, H, W, Cin, N, k, stride, pad = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    xr, wr, br = (t.clone().double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, stride=stride, padding=pad).permute(0, 2, 3, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xd, wd, bd = (t.clone().to(dev).requires_grad_(True) for t in (x, w, b))
    y = conv2d_nhwc(xd, wd, bd, stride, pad)
    y.backward(gy.to(dev))
    rel = lambda a, r: float((a.detach().cpu().double() - r).abs().max() / (r.abs().max() + 1e-12))
    assert rel(y, yr.detach()) < 1e-5
    assert rel(xd.grad, xr.grad) < 1e-5, "data gradient"
    assert rel(wd.grad, wr.grad) < 1e-4, "weight gradient"
    assert rel(bd.grad, br.grad) < 1e-5


def test_linear_hip_autograd_and_sgd_step_reduce_the_loss():
    from keypointfusion_amd.training import SmoothL1Loss, linear_hip, make_optimizer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    w = torch.nn.Parameter((torch.randn(32, 64) / 8).to(dev))
    b = torch.nn.Parameter(torch.zeros(32, device=dev))
    x, y = torch.randn(40, 64, device=dev), torch.randn(40, 32, device=dev)
    opt, _ = make_optimizer([w, b], lr=1e-2)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        loss = SmoothL1Loss()(linear_hip(x, w, b), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 4, 4, 8), (3, 5, 7, 12), (2, 16, 16, 128), (1, 1, 1, 4), (2, 1, 6, 4)])
def test_upsample2x_forward_backward_match_torch(shape, dt):
    """kpf_upsample2x_* / kpf_upsample2x_bwd vs F.interpolate(scale_factor=2, bilinear, align_corners=False) in float64; the backward is a
    gather (no atomics): two runs return the same bits."""
    from keypointfusion_amd.training import upsample2x_nhwc
    B, H, W, Cc = shape
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn(B, H, W, Cc, generator=g).to(dt)
    dy = torch.randn(B, 2 * H, 2 * W, Cc, generator=g).to(dt)
    xd = x.cuda().requires_grad_(True)
    y = upsample2x_nhwc(xd)
    y.backward(dy.cuda())
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=False)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    eps = 2e-6 if dt == torch.float32 else 2.0 ** -8
    assert float((y.detach().cpu().double() - yr.detach().permute(0, 2, 3, 1)).abs().max()) <= eps * max(float(yr.abs().max()), 1.0)
    assert float((xd.grad.cpu().double() - xr.grad.permute(0, 2, 3, 1)).abs().max()) <= eps * max(float(xr.grad.abs().max()), 1.0)
    g1 = xd.grad.clone()
    xd.grad = None
    upsample2x_nhwc(xd).backward(dy.cuda())
    assert torch.equal(g1, xd.grad)


@pytest.mark.parametrize("shape", [(2, 8, 8, 8), (2, 7, 9, 12), (1, 64, 64, 64), (2, 1, 1, 4), (1, 2, 5, 4)])
def test_maxpool3x3s2_forward_backward_match_torch_including_ties(shape):
    """kpf_maxpool3x3s2_fwd/_bwd vs F.max_pool2d(3, 2, 1): values drawn from a small integer set so that most windows hold ties — the
    gradient must go to the FIRST maximum in scan order, as ATen routes it."""
    from keypointfusion_amd.training import maxpool3x3s2_nhwc
    B, H, W, Cc = shape
    g = torch.Generator().manual_seed(H + W * 7)
    x = torch.randint(0, 3, (B, H, W, Cc), generator=g).float()
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, OH, OW, Cc, generator=g)
    xd = x.cuda().requires_grad_(True)
    y = maxpool3x3s2_nhwc(xd)
    y.backward(dy.cuda())
    xr = x.clone().permute(0, 3, 1, 2).contiguous().requires_grad_(True)  # CPU fp32: ATen's reference tie-breaking
    yr = F.max_pool2d(xr, 3, 2, 1)
    yr.backward(dy.permute(0, 3, 1, 2).contiguous())
    assert torch.equal(y.detach().cpu(), yr.detach().permute(0, 2, 3, 1))
    assert float((xd.grad.cpu() - xr.grad.permute(0, 2, 3, 1)).abs().max()) <= 1e-6 * float(xr.grad.abs().max())


@pytest.mark.parametrize("case", [(3, 1024, 1024, 4, 128, True), (2, 1045, 21 * 64, 1, 128, False), (2, 37, 50, 3, 8, True), (1, 5, 4096, 2, 4, True)])
def test_row_gather_forward_backward_match_torch(case):
    """kpf_row_gather_fwd/_bwd_f32 vs torch.gather + weighted sum in float64 (heavily repeated indices included); the backward adds
    in entry order without atomics: bit-identical across runs."""
    from keypointfusion_amd.training import row_gather
    B, P, R, G, Cc, weighted = case
    g = torch.Generator().manual_seed(P + R)
    src = torch.randn(B, P, Cc, generator=g)
    idx = torch.randint(0, P, (B, R, G), generator=g)
    idx[0, : R // 2] = idx[0, 0, 0]  # one source row gathered by many entries
    w = torch.rand(B, R, G, generator=g) if weighted else None
    dout = torch.randn(B, R, Cc, generator=g)
    sd = src.cuda().requires_grad_(True)
    out = row_gather(sd, idx.int().cuda(), w.cuda() if weighted else None)
    out.backward(dout.cuda())
    sr = src.double().requires_grad_(True)
    gr = torch.gather(sr, 1, idx.reshape(B, R * G, 1).expand(-1, -1, Cc)).view(B, R, G, Cc)
    outr = (gr * (w.double().unsqueeze(-1) if weighted else 1.0)).sum(2)
    outr.backward(dout.double())
    assert float((out.detach().cpu().double() - outr.detach()).abs().max()) <= 2e-6 * float(outr.abs().max())
    assert float((sd.grad.cpu().double() - sr.grad).abs().max()) <= 2e-5 * float(sr.grad.abs().max())
    g1 = sd.grad.clone()
    sd.grad = None
    row_gather(sd, idx.int().cuda(), w.cuda() if weighted else None).backward(dout.cuda())
    assert torch.equal(g1, sd.grad)


@pytest.mark.parametrize("shape,ydt", [((4, 16, 16, 96), torch.float32), ((3, 21, 128), torch.float32), ((2, 4, 4, 768), torch.bfloat16),
                                       ((777, 1024), torch.float32), ((5, 8), torch.float16), ((2, 9, 9, 384), torch.float32)])
def test_layer_norm_rows_forward_backward_match_torch(shape, ydt):
    """kpf_ln_train_forward / _backward vs F.layer_norm in float64 (per-channel means far from zero); the parameter gradients are column
    sums added in a fixed order: two runs return the same bits."""
    from keypointfusion_amd.training import layer_norm_rows
    Cc = shape[-1]
    g = torch.Generator().manual_seed(Cc + len(shape))
    x = torch.randn(*shape, generator=g) * 2.0 + 4.0 * torch.randn(Cc, generator=g)
    w, b = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    dy = torch.randn(*shape, generator=g)
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = layer_norm_rows(xd, wd, bd, 1e-6, ydt)
    assert y.dtype == ydt
    y.backward(dy.cuda().to(ydt))
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.layer_norm(xr, (Cc,), wr, br, 1e-6)
    yr.backward(dy.to(ydt).double())
    eps = {torch.float32: 2e-5, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[ydt]
    rel = lambda a, r: float((a.detach().cpu().double() - r.detach()).abs().max()) / max(float(r.detach().abs().max()), 1e-3)
    assert rel(y, yr) <= 1.01 * eps
    assert rel(xd.grad, xr.grad) <= 3e-5 and rel(wd.grad, wr.grad) <= 3e-5 and rel(bd.grad, br.grad) <= 3e-5
    g1 = (xd.grad.clone(), wd.grad.clone(), bd.grad.clone())
    xd.grad = wd.grad = bd.grad = None
    layer_norm_rows(xd, wd, bd, 1e-6, ydt).backward(dy.cuda().to(ydt))
    assert torch.equal(g1[0], xd.grad) and torch.equal(g1[1], wd.grad) and torch.equal(g1[2], bd.grad)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gelu_rows_forward_backward_match_torch(dt):
    from keypointfusion_amd.training import gelu_rows
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(1024, 36, generator=g) * 3).to(dt)
    dy = torch.randn(1024, 36, generator=g).to(dt)
    xd = x.cuda().requires_grad_(True)
    y = gelu_rows(xd)
    y.backward(dy.cuda())
    xr = x.double().requires_grad_(True)
    yr = F.gelu(xr)
    yr.backward(dy.double())
    eps = 3e-6 if dt == torch.float32 else 2.0 ** -8
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) <= eps * max(float(yr.abs().max()), 1.0)
    assert float((xd.grad.cpu().double() - xr.grad).abs().max()) <= eps * max(float(xr.grad.abs().max()), 1.0)


@pytest.mark.parametrize("B,scale", [(3, 32 ** -0.5), (1, 1.0)])
def test_attn21_forward_backward_match_torch(B, scale):
    """kpf_attn21_forward / _backward (softmax(scale QK^T) V per head, heads read in place) vs the matmul / softmax chain in float64."""
    from keypointfusion_amd.training import attn21
    g = torch.Generator().manual_seed(B)
    q, k, v, dctx = (torch.randn(B, 21, 128, generator=g) for _ in range(4))
    qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = attn21(qd, kd, vd, 4, scale)
    out.backward(dctx.cuda())
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    sp = lambda t: t.view(B, 21, 4, 32).transpose(1, 2)
    a = torch.softmax(torch.matmul(sp(qr), sp(kr).transpose(-1, -2)) * scale, -1)
    ref = torch.matmul(a, sp(vr)).transpose(1, 2).reshape(B, 21, 128)
    ref.backward(dctx.double())
    rel = lambda x, r: float((x.detach().cpu().double() - r.detach()).abs().max()) / float(r.detach().abs().max())
    assert rel(out, ref) < 1e-5
    assert rel(qd.grad, qr.grad) < 1e-5 and rel(kd.grad, kr.grad) < 1e-5 and rel(vd.grad, vr.grad) < 1e-5


def test_attn21_dropout_is_a_bernoulli_mask_consistent_between_forward_and_backward():
    """Dropout inside the fused attention: the kept fraction is 1 - p, kept probabilities are scaled by 1/(1-p), the backward uses the same
    mask (finite-difference-free check: with V = identity-like rows the output reveals the dropped probabilities), and the counter
    changes the mask."""
    from keypointfusion_amd.training import attn21
    B, p = 64, 0.25
    g = torch.Generator().manual_seed(0)
    q, k = torch.randn(B, 21, 128, generator=g).cuda(), torch.randn(B, 21, 128, generator=g).cuda()
    v = torch.zeros(B, 21, 128)
    for h in range(4):
        v[:, :, 32 * h:32 * h + 21] = torch.eye(21)  # ctx[b, i, 32h + j] = P'[b, h, i, j]
    v = v.cuda().requires_grad_(True)
    rng = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")
    out = attn21(q, k, v, 4, 32 ** -0.5, p, rng, 1)
    ref = attn21(q, k, v, 4, 32 ** -0.5, 0.0, None, 1)
    pd = torch.stack([out[:, :, 32 * h:32 * h + 21] for h in range(4)], 1)   # [B, 4, 21, 21] dropped probabilities
    pr = torch.stack([ref[:, :, 32 * h:32 * h + 21] for h in range(4)], 1)
    kept = pd != 0
    frac = float(kept.float().mean())
    assert abs(frac - (1 - p)) < 0.01, frac
    assert float((pd[kept] - pr[kept] / (1 - p)).abs().max()) < 1e-6
    out.sum().backward()  # d ctx = 1: dV[b, j, 32h + d] = sum_i P'[b,h,i,j] for every d
    dv = v.grad
    want = pd.sum(2)  # [B, 4, 21(j)]
    for h in range(4):
        assert float((dv[:, :, 32 * h] - want[:, h]).abs().max()) < 1e-5
    rng2 = torch.tensor([1234, 1], dtype=torch.int64, device="cuda")
    out2 = attn21(q, k, v, 4, 32 ** -0.5, p, rng2, 1)
    assert not torch.equal(out2 != 0, out != 0)


@pytest.mark.parametrize("B,J,P,Cc", [(3, 21, 1024, 128), (2, 21, 50, 8), (5, 21, 52, 36), (2, 24, 260, 200), (1, 7, 33, 128), (2, 21, 64, 400)])
def test_bmm_small_k_forward_backward_match_torch(B, J, P, Cc):
    # (P % 4 == 0: the four-rows-per-trip forward; C <= 216: the LDS-staged A-gradient; other shapes: the first forms)
    from keypointfusion_amd.training import bmm_small_k
    g = torch.Generator().manual_seed(P)
    A, X, dout = torch.randn(B, J, P, generator=g), torch.randn(B, P, Cc, generator=g), torch.randn(B, J, Cc, generator=g)
    Ad, Xd = A.cuda().requires_grad_(True), X.cuda().requires_grad_(True)
    out = bmm_small_k(Ad, Xd)
    out.backward(dout.cuda())
    Ar, Xr = A.double().requires_grad_(True), X.double().requires_grad_(True)
    ref = torch.bmm(Ar, Xr)
    ref.backward(dout.double())
    rel = lambda x, r: float((x.detach().cpu().double() - r.detach()).abs().max()) / float(r.detach().abs().max())
    assert rel(out, ref) < 1e-5 and rel(Ad.grad, Ar.grad) < 1e-5 and rel(Xd.grad, Xr.grad) < 1e-5


@pytest.mark.parametrize("ydt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(4, 16, 16, 96), (2, 4, 4, 768), (3, 5, 7, 192)])
def test_layer_scale_residual_forward_backward_match_torch(shape, ydt):
    from keypointfusion_amd.training import layer_scale_residual
    Cc = shape[-1]
    g = torch.Generator().manual_seed(Cc)
    x, y, gm, dout = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g).to(ydt), torch.randn(Cc, generator=g), torch.randn(*shape, generator=g)
    xd, yd, gd = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True), gm.cuda().requires_grad_(True)
    out = layer_scale_residual(xd, gd, yd)
    out.backward(dout.cuda())
    xr, yr, gr = x.double().requires_grad_(True), y.double().requires_grad_(True), gm.double().requires_grad_(True)
    ref = xr + gr * yr
    ref.backward(dout.double())
    rel = lambda a, r: float((a.detach().cpu().double() - r.detach()).abs().max()) / float(r.detach().abs().max())
    eps = 2e-6 if ydt == torch.float32 else 2.0 ** -8
    assert out.dtype == torch.float32 and rel(out, ref) < 2e-6
    assert torch.equal(xd.grad.cpu(), dout) and yd.grad.dtype == ydt and rel(yd.grad, yr.grad) <= 1.01 * eps and rel(gd.grad, gr.grad) < 2e-5
    g1 = gd.grad.clone()
    gd.grad = None
    layer_scale_residual(xd, gd, yd).backward(dout.cuda())
    assert torch.equal(g1, gd.grad)


@pytest.mark.parametrize("seed", [0, 1])
def test_dense_stage_loss_kernel_matches_the_torch_codec(seed):
    """kpf_dense_loss_forward / _backward against the torch restatement of GFM.joint2offset / offset2joint_weight + SmoothL1 (itself pinned to
    the reference's fixtures on the CPU): both loss terms and the gradient of all 105 maps, on crops with background, a joint far from
    the hand (empty target mask) and an all-background sample."""
    from keypointfusion_amd import training as T
    from keypointfusion_amd.weights import synthetic_batch
    B = 3
    g = torch.Generator().manual_seed(seed)
    img = torch.from_numpy(synthetic_batch(B, 128, seed=seed + 2)["img"])
    img[2] = 1.0  # all background: uniform soft-argmax weights, empty masks
    pd = torch.randn(B, 105, 32, 32, generator=g) * 0.5
    gt = torch.rand(B, 21, 3, generator=g) * 1.6 - 0.8
    gt[0, 0] = torch.tensor([3.0, 3.0, 3.0])  # farther than the kernel size from every pixel
    pr = pd.clone().double().requires_grad_(True)
    l1 = T.SmoothL1Loss()
    from oracle import train_oracle as TO
    pg = TO.joint2offset(gt.double(), img.double(), 0.8, 32)
    lp_r = l1(pr[:, :84], pg)
    lc_r = l1(TO.offset2joint_weight(pr, img.double(), 0.8), gt.double())
    (lp_r * 1.0 + lc_r * 100.0).backward()
    pdd = pd.cuda().requires_grad_(True)
    lp, lc = T.DenseStageLoss.apply(pdd, img.cuda(), gt.cuda(), 0.8)
    (lp * 1.0 + lc * 100.0).backward()
    assert abs(float(lp) - float(lp_r)) < 2e-6 * abs(float(lp_r)) and abs(float(lc) - float(lc_r)) < 2e-6 * abs(float(lc_r)), (float(lp), float(lp_r), float(lc), float(lc_r))
    d = (pdd.grad.cpu().double() - pr.grad).abs().max()
    assert float(d) < 2e-5 * float(pr.grad.abs().max()), float(d)


@pytest.mark.parametrize("epoch", [0, 25, "dev0", "dev25"])
def test_fused_loss_matches_the_torch_schedule(epoch):
    """training.FusedLoss (the whole loss of train.py:211-261 as one autograd node: kpf_dense_loss_* + kpf_loss_tail_*) against
    oracle/train_oracle.py::kpfusion_loss (the torch restatement) in float64 on the CPU (itself pinned to the reference's train_loss.npz): the total, every named
    term and the gradient of all six results and both spatial weights; host epoch gate (term present / absent) and device epoch gate;
    spatial weights in the permuted [B, H, W, J] memory the model produces."""
    from keypointfusion_amd import training as T
    from keypointfusion_amd.weights import synthetic_batch
    B, J, Fs = 3, 21, 32
    g = torch.Generator().manual_seed(7)
    img = torch.from_numpy(synthetic_batch(B, 128, seed=3)["img"])
    uvd = torch.rand(B, J, 3, generator=g) * 1.6 - 0.8
    xyz = torch.rand(B, J, 3, generator=g) * 1.6 - 0.8
    res = [torch.randn(B, 5 * J, Fs, Fs, generator=g) * 0.5 for _ in range(2)] + [xyz + torch.randn(B, J, 3, generator=g) * s for s in (0.003, 0.02, 0.1, 0.008)]
    sws = [torch.rand(B, Fs, Fs, J, generator=g).permute(0, 3, 1, 2) for _ in range(2)]
    ep_host = int(epoch[3:]) if isinstance(epoch, str) else epoch

    ref_in = [r.clone().double().requires_grad_(True) for r in res + sws]
    from oracle import train_oracle as TO
    lr, pr = TO.kpfusion_loss(ref_in[:6], ref_in[6:], img.double(), uvd.double(), xyz.double(), epoch=ep_host)
    lr.backward()
    dev_in = [r.cuda().requires_grad_(True) for r in res]
    dev_sw = [torch.empty(B, Fs, Fs, J, device="cuda").copy_(s.permute(0, 2, 3, 1)).permute(0, 3, 1, 2).requires_grad_(True) for s in sws]
    ep = torch.tensor(ep_host, device="cuda") if isinstance(epoch, str) else epoch
    ld, pdv = T.kpfusion_loss(dev_in, dev_sw, img.cuda(), uvd.cuda(), xyz.cuda(), epoch=ep)
    assert ld.grad_fn is not None and type(ld.grad_fn).__name__.startswith("FusedLoss"), ld.grad_fn
    (ld * 1.5).backward()
    assert abs(float(ld) - float(lr)) < 3e-6 * abs(float(lr)), (float(ld), float(lr))
    if isinstance(epoch, str):  # device gate: the spatial entries exist and are 0 behind the gate
        assert set(pdv) == set(pr) | {"loss_spatial_0", "loss_spatial_1"}
        if ep_host > 24:
            assert float(pdv["loss_spatial_0"]) == 0.0 and float(pdv["loss_spatial_1"]) == 0.0
    else:
        assert set(pdv) == set(pr)
    for k in pr:
        assert abs(float(pdv[k]) - float(pr[k])) < 5e-6 * max(abs(float(pr[k])), 1e-6), (k, float(pdv[k]), float(pr[k]))
    for i, (a, b) in enumerate(zip(dev_in + dev_sw, ref_in)):
        want = torch.zeros_like(b) if b.grad is None else b.grad * 1.5
        if a.grad is None:
            assert float(want.abs().max()) == 0.0, i
            continue
        d = float((a.grad.cpu().double() - want).abs().max())
        assert d <= 3e-5 * float(want.abs().max()) + 1e-12, (i, d, float(want.abs().max()))


def test_fused_adamw_matches_torch_adamw_and_exchanges_state():
    """training.FusedAdamW (kpf_adamw_step_multi) against torch.optim.AdamW(fused=True, capturable=True) over several steps: parameters and
    both moments, odd sizes (scalar tail, > one workgroup, more tensors than one launch carries), a parameter without a gradient, a
    learning rate that changes between steps (device scalar), and a state_dict() that the library class loads and continues from."""
    from keypointfusion_amd import training as T
    g = torch.Generator().manual_seed(11)
    shapes = [(1,), (5,), (4099,), (33, 7), (64, 64, 3, 3), (8193,)] + [(17 + i,) for i in range(90)]
    ref_p = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    dead_r, dead_o = torch.nn.Parameter(torch.ones(3).cuda()), torch.nn.Parameter(torch.ones(3).cuda())
    lr_r, lr_o = torch.tensor(8e-4, device="cuda"), torch.tensor(8e-4, device="cuda")
    ref = torch.optim.AdamW(ref_p + [dead_r], lr=lr_r, weight_decay=0.01, capturable=True, fused=True)
    our = T.FusedAdamW(our_p + [dead_o], lr=lr_o, weight_decay=0.01, capturable=True)
    for it in range(6):
        if it == 3:
            lr_r.fill_(8e-5), lr_o.fill_(8e-5)
        for a, b in zip(ref_p, our_p):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** (it % 3 - 1))
            a.grad, b.grad = gr.clone(), gr.clone()
        ref.step(), our.step()
    torch.cuda.synchronize()
    assert dead_o.grad is None and torch.equal(dead_o, dead_r) and not our.state.get(dead_o)
    for a, b in zip(ref_p, our_p):
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-9, a.shape
        for k in ("exp_avg", "exp_avg_sq"):
            x, y = ref.state[a][k], our.state[b][k]
            assert float((x - y).abs().max()) <= 2e-6 * float(x.abs().max()) + 1e-12, (a.shape, k)
        assert float(our.state[b]["step"]) == float(ref.state[a]["step"]) == 6.0
    # the library class continues from our state
    cont = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone()) for p in our_p] + [torch.nn.Parameter(torch.ones(3).cuda())], lr=torch.tensor(8e-5, device="cuda"),
                             weight_decay=0.01, capturable=True, fused=True)
    cont.load_state_dict(our.state_dict())
    cp = cont.param_groups[0]["params"]
    for a, b, c in zip(ref_p, our_p, cp):
        gr = torch.randn(a.shape, generator=g).cuda()
        a.grad, c.grad = gr.clone(), gr.clone()
    ref.step(), cont.step()
    for a, c in zip(ref_p, cp):
        assert float((a - c).abs().max()) <= 4e-6 * float(a.abs().max()) + 1e-9, a.shape


def test_grouped_linear_weight_gradients_after_backward():
    """training.GroupedLinearWgrad (kpf_linear_wgrad_grouped): the weight / bias gradients of many small Linear layers in one launch per 80
    layers after backward, against fp64 — more layers than one launch carries, ragged row counts (1, 21, 672, 1000), widths that are not
    whole tiles, layers without bias; every dW must have been ADOPTED as the parameter's .grad (no copy of the unwritten tensor); layers
    outside the rule (too many rows, sliced weight key) take the immediate path inside the same pass; a weight used twice is refused."""
    from keypointfusion_amd import training as T
    g = torch.Generator().manual_seed(5)
    cache = T.PackCache()
    layers, named = [], {}
    for i in range(90):
        rows, k, n = [(672, 128, 128), (21, 132, 64), (1000, 64, 516), (1, 8, 4), (672, 512, 128), (2048, 32, 32)][i % 6]
        w = torch.nn.Parameter((torch.randn(n, k, generator=g) * 0.1).cuda())
        b = torch.nn.Parameter(torch.randn(n, generator=g).cuda()) if i % 4 else None
        x = torch.randn(rows, k, generator=g).cuda().requires_grad_(True)
        named["l%d" % i] = w
        if b is not None:
            named["l%d.bias" % i] = b  # (the bias gradient is handed over unwritten too: it must be a parameter the map knows)
        layers.append((w, b, x))

    def run(grouped):
        for w, b, x in layers:
            w.grad = x.grad = None
            if b is not None:
                b.grad = None
        ctx = T.GroupedLinearWgrad(named) if grouped else __import__("contextlib").nullcontext()
        with ctx as grp:
            tot = 0
            for i, (w, b, x) in enumerate(layers):
                tot = tot + T.linear_hip(x, w, b, "f32", None, "l%d" % i, cache).square().sum()
            tot.backward()
            n_grouped = len(grp.items) if grouped else 0
        torch.cuda.synchronize()
        return n_grouped

    assert run(False) == 0
    imm = [(w.grad.clone(), None if b is None else b.grad.clone()) for w, b, x in layers]
    assert run(True) == 75  # (the 2048-row layers take the immediate path)
    for i, ((w, b, x), (gw, gb)) in enumerate(zip(layers, imm)):  # same bits as the per-layer split + reduce form (same summation order)
        assert torch.equal(w.grad, gw), (i, float((w.grad - gw).abs().max()))
        assert b is None or torch.equal(b.grad, gb), i
    for i, (w, b, x) in enumerate(layers):
        xr, wr = x.detach().double().cpu(), w.detach().double().cpu()
        y = xr @ wr.t() + (b.detach().double().cpu() if b is not None else 0)
        dy = 2 * y
        rw, rb = dy.t() @ xr, dy.sum(0)
        assert float((w.grad.cpu().double() - rw).abs().max()) <= 2e-5 * float(rw.abs().max()), i
        if b is not None:
            assert float((b.grad.cpu().double() - rb).abs().max()) <= 2e-5 * max(float(rb.abs().max()), 1.0), i
        assert float((x.grad.cpu().double() - dy @ wr).abs().max()) <= 2e-5 * float((dy @ wr).abs().max()), i
    with pytest.raises(RuntimeError, match="second gradient"):
        with T.GroupedLinearWgrad(named):
            w, b, x = layers[0]
            w.grad = None  # (a parameter that still holds a gradient is not deferred at all: autograd would add to it)
            if b is not None:
                b.grad = None
            (T.linear_hip(x, w, b, "f32", None, "l0", cache).sum() + T.linear_hip(x, w, b, "f32", None, "l0", cache).sum()).backward()
    assert T.GroupedLinearWgrad.active is None
    w, b, x = layers[0]
    # a stale gradient: autograd would ADD the still-unwritten tensor to it instead of adopting it -> such a parameter is not deferred at all (ADVICE
    # r03: decided before anything is handed to autograd, not detected afterwards); it takes the per-layer kernels and accumulates correctly
    want = imm[0][0].clone()
    w.grad = torch.ones_like(w)
    if b is not None:
        b.grad = None
    x.grad = None
    with T.GroupedLinearWgrad(named) as grp:
        T.linear_hip(x, w, b, "f32", None, "l0", cache).square().sum().backward()
        assert len(grp.items) == 0
    assert torch.equal(w.grad, want + 1.0)


def test_deferred_layernorm_and_layer_scale_parameter_sums_are_bit_identical():
    """training.DeferredParamGrads, column-sum half (kpf_ln_train_backward_partial / kpf_layer_scale_backward_partial +
    kpf_colsum_reduce_grouped): d gamma / d beta of many LayerNorm and layer-scale layers reduced in grouped launches after backward —
    same bits as the per-layer reduce, more layers than one launch carries, both storage types; parameters the model does not own
    (not in the name map) keep the immediate path."""
    from keypointfusion_amd import training as T
    g = torch.Generator().manual_seed(9)
    lns, lss, named = [], [], {}
    for i in range(70):
        rows, c = [(672, 128), (2048, 96), (21, 512), (4096, 384)][i % 4]
        w, b = torch.nn.Parameter((torch.rand(c, generator=g) + 0.5).cuda()), torch.nn.Parameter(torch.randn(c, generator=g).cuda())
        lns.append((w, b, torch.randn(rows, c, generator=g).cuda().requires_grad_(True)))
        if i < 66:
            named["ln%d.weight" % i], named["ln%d.bias" % i] = w, b
    for i in range(40):
        rows, c = [(1024, 96), (256, 768)][i % 2]
        gm = torch.nn.Parameter((torch.rand(c, generator=g) * 0.1).cuda())
        dt = torch.bfloat16 if i % 3 == 0 else torch.float32
        lss.append((gm, torch.randn(rows, c, generator=g).cuda().requires_grad_(True), torch.randn(rows, c, generator=g).cuda().to(dt).requires_grad_(True)))
        named["ls%d.gamma" % i] = gm

    def run(deferred):
        for w, b, x in lns:
            w.grad = b.grad = x.grad = None
        for gm, x, y in lss:
            gm.grad = x.grad = y.grad = None
        ctx = T.DeferredParamGrads(named) if deferred else __import__("contextlib").nullcontext()
        with ctx as grp:
            tot = 0
            for w, b, x in lns:
                tot = tot + T.layer_norm_rows(x, w, b, 1e-6).square().sum()
            for gm, x, y in lss:
                tot = tot + T.layer_scale_residual(x, gm, y).square().sum()
            tot.backward()
            n = len(grp.colsums) if deferred else 0
        torch.cuda.synchronize()
        return n, [t.grad.clone() for w, b, x in lns for t in (w, b, x)] + [t.grad.clone() for gm, x, y in lss for t in (gm, x, y)]

    n0, ref = run(False)
    n1, got = run(True)
    assert n0 == 0 and n1 == 66 + 40  # (the four LayerNorms outside the name map reduce at once)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_batched_weight_gradient_reduces_are_bit_identical(dt):
    """training.DeferredParamGrads, reduce half (kpf_conv2d_wgrad_deferred / kpf_dwconv7_wgrad_deferred + kpf_wgrad_reduce_multi): the fixed-order
    reduces behind the split weight-gradient GEMMs of a backward pass issued in one launch per 24 calls — same bits as a reduce launch per
    call; dense / 3x3 / strided / grouped / depthwise layers, more calls than one launch carries, a paired parameter (two parameters side by
    side behind one weight tensor) and a parameter outside the name map (keeps its own reduce)."""
    from keypointfusion_amd import training as T
    g = torch.Generator().manual_seed(12)
    prec = "f32" if dt == torch.float32 else "bf16"
    layers, named = [], {}
    shapes = [(2, 24, 24, 32, 64, 1, 1, 0, 1), (2, 24, 24, 64, 32, 3, 1, 1, 1), (4, 16, 16, 32, 64, 2, 2, 0, 1), (2, 24, 24, 64, 64, 1, 1, 0, 2),
              (8, 32, 32, 96, 384, 1, 1, 0, 1), (2, 20, 20, 48, 24, 3, 2, 1, 1)]
    for i in range(30):
        B, H, W, cin, n, k, s, pd, G = shapes[i % len(shapes)]
        w = torch.nn.Parameter((torch.randn(n, cin // G, k, k, generator=g) * 0.1).cuda())
        b = torch.nn.Parameter(torch.randn(n, generator=g).cuda())
        x = torch.randn(B, H, W, cin, generator=g).cuda().requires_grad_(True)
        layers.append(("conv", w, b, x, s, pd, G))
        if i != 7:
            named["c%d.weight" % i], named["c%d.bias" % i] = w, b
    for i in range(6):
        c = [32, 96][i % 2]
        w = torch.nn.Parameter((torch.randn(c, 1, 7, 7, generator=g) * 0.1).cuda())
        b = torch.nn.Parameter(torch.randn(c, generator=g).cuda())
        layers.append(("dw", w, b, torch.randn(2, 24, 24, c, generator=g).cuda().requires_grad_(True), 1, 3, 1))
        named["d%d.weight" % i], named["d%d.bias" % i] = w, b
    reg = {}
    pa, pb = [torch.nn.Parameter((torch.randn(64, 32, 1, 1, generator=g) * 0.1).cuda()) for _ in range(2)]
    ba, bb = [torch.nn.Parameter(torch.randn(64, generator=g).cuda()) for _ in range(2)]
    T.pair_storage(reg, "pw", pa, pb), T.pair_storage(reg, "pbias", ba, bb)
    named.update({"a.weight": pa, "b.weight": pb, "a.bias": ba, "b.bias": bb})
    xp = torch.randn(2, 24, 24, 64, generator=g).cuda().requires_grad_(True)
    params = [t for l in layers for t in l[1:4]] + [pa, pb, ba, bb, xp]

    def run(deferred):
        for t in params:
            t.grad = None
        ctx = T.DeferredParamGrads(named) if deferred else __import__("contextlib").nullcontext()
        with ctx as grp:
            tot = 0
            for kind, w, b, x, s, pd, G in layers:
                if kind == "conv":
                    y = T.conv2d_nhwc(x.to(dt) if dt != torch.float32 else x, w, b, stride=s, pad=pd, prec=prec, groups=G)
                else:
                    y = T.dwconv7_nhwc(x, w, b)
                tot = tot + y.float().square().sum()
            wp = T.pair_params(reg, "pw", pa, pb).view(128, 32, 1, 1)
            bp = T.pair_params(reg, "pbias", ba, bb).view(128)
            tot = tot + T.conv2d_nhwc(xp.to(dt) if dt != torch.float32 else xp, wp, bp, prec=prec, groups=2).float().square().sum()
            tot.backward()
            n = grp.n_reduces if deferred else 0
        torch.cuda.synchronize()
        return n, [t.grad.clone() for t in params]

    n0, ref = run(False)
    n1, got = run(True)
    assert n0 == 0 and 25 <= n1 <= 36, n1  # (a call whose tiles fill the chip writes its gradient directly; the unnamed layer reduces at once)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))


def test_joint_heatmap_and_geometry_gate_match_torch():
    """training.JointHeatmap / training.GeomGate (one launch each way) against the torch expressions they replace in the fusion block
    (GFM.joint2heatmap; 1 / (10 |pixel - joint|^2 + 1)) in float64: values and the gradients towards the joints, joints on / off the map."""
    from keypointfusion_amd import training as T
    g = torch.Generator().manual_seed(4)
    B, J, F_ = 3, 21, 32
    uvd = torch.rand(B, J, 3, generator=g) * 2.4 - 1.2
    w = torch.randn(B, J, F_, F_, generator=g)
    ur = uvd.clone().double().requires_grad_(True)
    from oracle import train_oracle as TO
    hr = TO.joint2heatmap(ur[:, :, :2], 0.8, F_, sigma=1)
    (hr * w.double()).sum().backward()
    ud = uvd.cuda().requires_grad_(True)
    hd = T.JointHeatmap.apply(ud, 0.8, F_, 1.0)
    (hd * w.cuda()).sum().backward()
    assert float((hd.detach().cpu().double() - hr.detach()).abs().max()) <= 2e-6
    assert float((ud.grad.cpu().double() - ur.grad).abs().max()) <= 2e-5 * float(ur.grad.abs().max()) and float(ud.grad[..., 2].abs().max()) == 0.0
    ix = torch.randn(B, F_ * F_, 3, generator=g) * 0.5
    jx = torch.randn(B, J, 3, generator=g) * 0.5
    wg = torch.randn(B, J, F_ * F_, generator=g)
    jr = jx.clone().double().requires_grad_(True)
    gr = 1 / (10 * torch.sum(torch.pow(ix.double().unsqueeze(1) - jr.unsqueeze(2), 2), dim=-1) + 1)
    (gr * wg.double()).sum().backward()
    jd = jx.cuda().requires_grad_(True)
    gd = T.GeomGate.apply(ix.cuda(), jd)
    (gd * wg.cuda()).sum().backward()
    assert float((gd.detach().cpu().double() - gr.detach()).abs().max()) <= 2e-6
    assert float((jd.grad.cpu().double() - jr.grad).abs().max()) <= 2e-5 * float(jr.grad.abs().max())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_skip_path_gradient_is_folded_into_the_producing_backward_kernel(dt):
    """BatchNormReLU / DwConv7NHWC with alias=True return their input a second time; the gradient that reaches the alias (the skip path of a
    Residual / ConvNeXt block) is added to dx by kpf_bn_train_backward_add / kpf_dwconv7_add_f32 inside the backward kernel.  Against
    torch in float64 on the same (rounded) operands: y, dx = layer gradient + skip gradient, parameter gradients; alias unused or used
    alone also work."""
    from keypointfusion_amd import training as T
    g = torch.Generator().manual_seed(6)
    M, Cc = 1500, 48
    x = (torch.randn(M, Cc, generator=g) * 1.5 + 0.3).to(dt)
    w, b = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    c1, c2 = torch.randn(M, Cc, generator=g), torch.randn(M, Cc, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.relu(F.batch_norm(xr, None, None, wr, br, True, 0.1, 1e-5))
    ((yr * c1.double()).sum() + (xr * c2.double()).sum()).backward()
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y, xa = T.batchnorm_relu_rows(xd, wd, bd, torch.zeros(Cc).cuda(), torch.ones(Cc).cuda(), 0.1, 1e-5, True, None, True)
    assert xa.data_ptr() == xd.data_ptr() and y.dtype == dt
    ((y.float() * c1.cuda()).sum() + (xa.float() * c2.cuda()).sum()).backward()
    eps = 3e-5 if dt == torch.float32 else 2.0 ** -7
    rel = lambda a, r: float((a.detach().cpu().double() - r.detach()).abs().max()) / max(float(r.detach().abs().max()), 1e-3)
    ptol = 2e-4 if dt == torch.float32 else 6e-3  # (16-bit: dy itself is rounded to the storage type on the way in)
    assert rel(y, yr) <= eps and rel(xd.grad, xr.grad) <= 1.5 * eps and rel(wd.grad, wr.grad) <= ptol and rel(bd.grad, br.grad) <= ptol
    # alias alone / alias unused
    xd2 = x.cuda().requires_grad_(True)
    y2, xa2 = T.batchnorm_relu_rows(xd2, wd, bd, None, None, 0.1, 1e-5, True, None, True)
    (xa2.float() * c2.cuda()).sum().backward()
    assert rel(xd2.grad, c2.double()) <= eps
    if dt == torch.float32:
        B, H, W_, C2 = 2, 9, 11, 32
        xi = torch.randn(B, H, W_, C2, generator=g)
        wk, bk = torch.randn(C2, 1, 7, 7, generator=g) * 0.1, torch.randn(C2, generator=g)
        d1, d2 = torch.randn(B, H, W_, C2, generator=g), torch.randn(B, H, W_, C2, generator=g)
        xr2, wr2, br2 = xi.double().requires_grad_(True), wk.double().requires_grad_(True), bk.double().requires_grad_(True)
        yr2 = F.conv2d(xr2.permute(0, 3, 1, 2), wr2, br2, padding=3, groups=C2).permute(0, 2, 3, 1)
        ((yr2 * d1.double()).sum() + (xr2 * d2.double()).sum()).backward()
        xd3, wd3, bd3 = xi.cuda().requires_grad_(True), wk.cuda().requires_grad_(True), bk.cuda().requires_grad_(True)
        y3, xa3 = T.dwconv7_nhwc(xd3, wd3, bd3, None, None, True)
        ((y3 * d1.cuda()).sum() + (xa3 * d2.cuda()).sum()).backward()
        assert rel(y3, yr2) <= 3e-5 and rel(xd3.grad, xr2.grad) <= 3e-5 and rel(wd3.grad, wr2.grad) <= 1e-4 and rel(bd3.grad, br2.grad) <= 1e-4


# ---- round 4: channel-stacked groups (the paired backbones of the training step: kpf_conv_desc::groups, kpf_conv2d_wgrad_groups,
# kpf_ln_train_*_g, kpf_layer_scale_backward_g).  Same kernels, tile choice and summation order per group as the ungrouped calls on the
# channel slices, so the comparison is BIT-EXACT, forward and every gradient (16-bit weight gradients: to rounding, see the test).
@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(4, 16, 16, 96, 384, 1, 1, 0), (4, 16, 16, 384, 96, 1, 1, 0), (3, 8, 8, 64, 64, 3, 1, 1), (2, 16, 16, 96, 192, 2, 2, 0),
                                  (2, 4, 4, 128, 112, 1, 1, 0), (32, 4, 4, 768, 3072, 1, 1, 0)])
def test_grouped_convolution_is_bit_identical_to_per_group_calls(case, prec):
    from keypointfusion_amd import training as T
    B, H, W, Cin, N, k, stride, pad = case
    G = 2
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(case))
    tdt = torch.float32 if prec == "f32" else torch.bfloat16
    x = torch.randn(B, H, W, G * Cin, generator=g).to(dev).to(tdt)
    w = (torch.randn(G * N, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5).to(dev)
    b = torch.randn(G * N, generator=g).to(dev)
    xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = T.conv2d_nhwc(xg, wg, bg, stride, pad, prec, None, None, None, G)
    dy = torch.randn(y.shape, generator=g).to(dev).to(y.dtype)
    y.backward(dy)
    for i in range(G):
        xi = x[..., i * Cin:(i + 1) * Cin].contiguous().requires_grad_(True)
        wi, bi = w[i * N:(i + 1) * N].clone().requires_grad_(True), b[i * N:(i + 1) * N].clone().requires_grad_(True)
        yi = T.conv2d_nhwc(xi, wi, bi, stride, pad, prec)
        yi.backward(dy[..., i * N:(i + 1) * N].contiguous())
        assert torch.equal(y[..., i * N:(i + 1) * N], yi), "forward, group %d" % i
        assert torch.equal(xg.grad[..., i * Cin:(i + 1) * Cin], xi.grad), "data gradient, group %d" % i
        if prec == "f32":
            assert torch.equal(wg.grad[i * N:(i + 1) * N], wi.grad), "weight gradient, group %d" % i
            assert torch.equal(bg.grad[i * N:(i + 1) * N], bi.grad), "bias gradient, group %d" % i
        else:  # (16-bit operands: a grouped launch splits the pixel range for half the chip per group — the same sums in another order)
            for a, r in ((wg.grad[i * N:(i + 1) * N], wi.grad), (bg.grad[i * N:(i + 1) * N], bi.grad)):
                assert float((a - r).abs().max()) <= 2e-5 * max(1.0, float(r.abs().max())), "weight / bias gradient, group %d" % i


@pytest.mark.parametrize("ydt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(4, 16, 16, 96), (2, 4, 4, 768), (3, 5, 7, 192), (32, 32, 32, 96)])
def test_grouped_layer_norm_and_layer_scale_match_per_group_calls(shape, ydt):
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    G, Cc = 2, shape[-1]
    g = torch.Generator().manual_seed(Cc + shape[0])
    x = torch.randn(shape[:-1] + (G * Cc,), generator=g).to(dev)
    w, b = (1 + 0.1 * torch.randn(G * Cc, generator=g)).to(dev), (0.1 * torch.randn(G * Cc, generator=g)).to(dev)
    dy = torch.randn(x.shape, generator=g).to(dev)
    xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = T.layer_norm_rows(xg, wg, bg, 1e-6, ydt, G)
    y.backward(dy.to(ydt))
    # layer scale: out = x + gamma * y2
    y2 = torch.randn(x.shape, generator=g).to(dev).to(ydt)
    xs, gs, ys = x.clone().requires_grad_(True), w.clone().requires_grad_(True), y2.clone().requires_grad_(True)
    o = T.layer_scale_residual(xs, gs, ys, G)
    o.backward(dy)
    for i in range(G):
        sl = slice(i * Cc, (i + 1) * Cc)
        xi, wi, bi = x[..., sl].contiguous().requires_grad_(True), w[sl].clone().requires_grad_(True), b[sl].clone().requires_grad_(True)
        yi = T.layer_norm_rows(xi, wi, bi, 1e-6, ydt)
        yi.backward(dy[..., sl].contiguous().to(ydt))
        assert torch.equal(y[..., sl], yi) and torch.equal(xg.grad[..., sl], xi.grad), "LayerNorm output / input gradient, group %d" % i
        # parameter gradients: column sums over the same rows, but the grouped launch cuts them into workgroup partials differently
        for a, r in ((wg.grad[sl], wi.grad), (bg.grad[sl], bi.grad)):
            assert float((a - r).abs().max()) <= 2e-5 * max(1.0, float(r.abs().max())), "LayerNorm parameter gradient, group %d" % i
        xi2, gi, yi2 = x[..., sl].contiguous().requires_grad_(True), w[sl].clone().requires_grad_(True), y2[..., sl].contiguous().requires_grad_(True)
        oi = T.layer_scale_residual(xi2, gi, yi2)
        oi.backward(dy[..., sl].contiguous())
        assert torch.equal(o[..., sl], oi) and torch.equal(ys.grad[..., sl], yi2.grad) and torch.equal(xs.grad[..., sl], xi2.grad), "layer scale, group %d" % i
        assert float((gs.grad[sl] - gi.grad).abs().max()) <= 2e-5 * max(1.0, float(gi.grad.abs().max())), "layer-scale gamma gradient, group %d" % i


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_paired_backbones_match_the_two_pass_training_graph(prec, monkeypatch):
    """KPF_TRAIN_PAIR: the two ConvNeXt backbones as one grouped network against the two separate passes — same loss, same gradients for every
    parameter (bit-identical GEMMs per group; the per-channel column sums of LayerNorm / layer scale are cut into different partials)."""
    from conftest import synthetic_sd
    from keypointfusion_amd import train_graph as TG
    from keypointfusion_amd import training as T
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.weights import synthetic_batch
    net, B, dev = "KPFusion-convnext-tiny", 4, torch.device("cuda:0")
    batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=5).items()}
    g = torch.Generator().manual_seed(1)
    uvd, xyz = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev), (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)

    class Loader:
        img_size, flip = 128, 1

    def run(pair):
        monkeypatch.setattr(TG, "PAIR_BACKBONES", pair)
        m = KPFusion(net, "", 21, "dexycb", "")
        m.load_state_dict(synthetic_sd(net), strict=True)
        m = m.to(dev).train()
        m.train_dropout, m.precision = 0.0, prec
        r, s, _ = m(batch["img_rgb"], batch["img"], batch["pcl"], Loader(), batch["center"], batch["M"], batch["cube"], batch["cam_para"], 0.8)
        loss = T.kpfusion_loss(r, s, batch["img"], uvd, xyz, epoch=0)[0]
        loss.backward()
        bufs = {k: v.detach().clone() for k, v in m.named_buffers()}
        return float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}, bufs, [t.detach().clone() for t in r]

    l0, g0, b0, r0 = run(False)
    l1, g1, b1, r1 = run(True)
    rel = lambda a, b: float((a.float() - b.float()).abs().max()) / max(float(a.float().abs().max()), 1e-6)
    assert set(g0) == set(g1)
    if prec == "f32":
        # Not bit-identical (BatchNorm / LayerNorm statistics are cut into different partial sums over 2C-wide rows, the patchify convolutions take
        # the general strided form) but equal to fp32 rounding: this case pins the structure of the paired pass.
        for a, b in zip(r0, r1):
            assert rel(a, b) <= 1e-4, rel(a, b)
        assert abs(l0 - l1) <= 1e-4 * abs(l0), (l0, l1)
        # (biases in front of a batch-statistics BatchNorm — FA.conv_l0_blocks.*.bias, *_emb.0.bias — have an exactly zero gradient: what the two passes
        #  hold there is rounding noise of ~1e-8, whose RELATIVE difference means nothing; every tensor is therefore measured against at least 1e-5 of the
        #  largest gradient of the model)
        floor = 1e-5 * max(float(v.float().abs().max()) for v in g0.values())
        relf = lambda a, b: float((a.float() - b.float()).abs().max()) / max(float(a.float().abs().max()), floor)
        errs = sorted((relf(g0[k], g1[k]), k) for k in g0)
        assert errs[len(errs) // 2][0] <= 1e-3 and errs[-1][0] <= 5e-2, (errs[len(errs) // 2], errs[-8:])
        for k in b0:
            assert torch.allclose(b0[k], b1[k], rtol=1e-4, atol=1e-5), k
        return
    # bf16: the two passes round 16-bit activations at different places, and at B = 4 the gradients behind batch statistics of 64 samples amplify
    # that; what must hold is that the paired pass is as close to the fp32 gradients as the two-pass form is (a structural error would not be).
    prec = "f32"
    lf, gf, _, rf = run(False)
    for a, b in zip(r0[:2], r1[:2]):
        assert rel(a, b) <= 5e-2, rel(a, b)
    assert abs(l1 - lf) <= max(2.0 * abs(l0 - lf), 2e-2 * abs(lf)), (l0, l1, lf)
    med = lambda g: sorted(rel(gf[k], g[k]) for k in gf)[len(gf) // 2]
    assert med(g1) <= 1.5 * med(g0) + 1e-2, (med(g0), med(g1))



@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(672, 131, 128, False), (672, 128, 3, False), (32768, 149, 21, True), (32768, 105, 128, True), (1344, 3, 64, True), (50, 7, 5, False)])
def test_odd_width_linear_matches_torch(case, prec):
    """Conv2dNHWC's odd-width form (round 4): input / output widths that are not whole channel groups, x given unpadded or already padded by its
    producer; forward, dX, dW, db against torch on the (rounded) operands."""
    from keypointfusion_amd import training as T
    M, K, N, prepadded = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + K + N)
    tdt = torch.float32 if prec == "f32" else torch.bfloat16
    cm = 4 if prec == "f32" else 8
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    dy = torch.randn(M, N, generator=g).to(dev)
    rd = lambda t: t.to(tdt).double()
    xr, wr = rd(x).requires_grad_(True), rd(w).requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = xr @ wr.t() + br
    yr.backward(rd(dy))
    xin = x.clone()
    if prepadded:
        xin = torch.cat([x, torch.zeros(M, (-K) % cm, device=dev)], 1)
    xg, wg, bg = xin.requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = T.linear_hip(xg, wg, bg, prec)
    assert y.shape == (M, N)
    y.backward(dy.to(y.dtype))
    tol = 2e-5 if prec == "f32" else 1.5e-2
    chk = lambda a, r, what: (float((a.double() - r).abs().max()) <= tol * max(float(r.abs().max()), 1.0), what)
    for ok, what in (chk(y, yr, "y"), chk(xg.grad[:, :K], xr.grad, "dx"), chk(wg.grad, wr.grad, "dw"), chk(bg.grad, br.grad, "db")):
        assert ok, what
    assert wg.grad.shape == (N, K) and bg.grad.shape == (N,)


def test_pose_tokens_and_geometry_gate_from_uvd_match_torch():
    from keypointfusion_amd import train_graph as TG
    from keypointfusion_amd import training as T
    from keypointfusion_amd import lib as L
    dev = torch.device("cuda:0")
    B, N, J, P, S = 3, 257, 21, 1024, 128
    g = torch.Generator().manual_seed(11)
    pcl = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
    pcl[:, ::7, 2] = 0.995  # (points beyond the depth gate)
    joint = (torch.rand(B, J, 3, generator=g) * 1.2 - 0.6).to(dev)
    pw = torch.randn(B, N, J, generator=g).to(dev)
    ref = torch.cat((pw, TG.TrainGraph.pcl_joint2offset(joint, pcl, 0.8)), -1)
    out = torch.empty(B, N, 108, device=dev)
    L.check(L.load().kpf_pose_tokens_f32(pw.data_ptr(), joint.data_ptr(), pcl.data_ptr(), out.data_ptr(), B, N, J, 108, 0.8, torch.cuda.current_stream().cuda_stream), "pose")
    assert float((out[..., :105] - ref).abs().max()) <= 2e-6 and float(out[..., 105:].abs().max()) == 0.0
    # geometry gate with the uvd -> xyz map inside the kernel, against the torch expression (TrainGraph.uvd2xyz + GeomGate)
    ix = (torch.rand(B, P, 3, generator=g) * 2 - 1).to(dev)
    uvd = (torch.rand(B, J, 3, generator=g) * 1.6 - 0.8).to(dev).requires_grad_(True)
    center = (torch.tensor([[10.0, -20.0, 600.0]]) + torch.randn(B, 3, generator=g) * 20).to(dev)
    cube = torch.tensor([[250.0, 250.0, 250.0]]).repeat(B, 1).to(dev)
    cam = torch.tensor([[615.0, 615.0, 320.0, 240.0]]).repeat(B, 1).to(dev)
    Minv = (torch.eye(3).view(1, 3, 3) + 0.05 * torch.randn(B, 3, 3, generator=g)).to(dev)
    Minv[:, :2, 2] += 100.0
    flip = -1
    jx = TG.TrainGraph.uvd2xyz(uvd, center, Minv, cube, cam, S, flip)
    gam_ref = T.GeomGate.apply(ix, jx)
    dg = torch.randn(B, J, P, generator=g).to(dev)
    gam_ref.backward(dg)
    gref = uvd.grad.clone()
    uvd.grad = None
    par = torch.cat((Minv.reshape(B, 9)[:, :6], cam, center, cube), 1)
    gam = T.GeomGateUVD.apply(ix, uvd, par, S, flip)
    gam.backward(dg)
    assert float((gam - gam_ref).abs().max()) <= 2e-6
    assert float((uvd.grad - gref).abs().max()) <= 2e-5 * max(1.0, float(gref.abs().max()))


def test_self_attention21_matches_three_linears_and_the_attention_core():
    """training.SelfAttention21 (one q | k | v projection, attention on its column slices, one data-gradient GEMM) against three linear_hip +
    attn21: same outputs and gradients (the projections are the same dot products in the same order: bit-identical forward)."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    B, Tn, Cc, H = 5, 21, 128, 4
    g = torch.Generator().manual_seed(3)
    h = torch.randn(B, Tn, Cc, generator=g).to(dev)
    ws = [(torch.randn(Cc, Cc, generator=g) * Cc ** -0.5).to(dev) for _ in range(3)]
    bs = [(0.1 * torch.randn(Cc, generator=g)).to(dev) for _ in range(3)]
    dctx = torch.randn(B, Tn, Cc, generator=g).to(dev)
    hr = h.clone().requires_grad_(True)
    wr, br = [w.clone().requires_grad_(True) for w in ws], [b.clone().requires_grad_(True) for b in bs]
    q, k, v = (T.linear_hip(hr, w, b) for w, b in zip(wr, br))
    ref = T.attn21(q, k, v, H, 32 ** -0.5)
    ref.backward(dctx)
    hg = h.clone().requires_grad_(True)
    wg, bg = [torch.nn.Parameter(w.clone()) for w in ws], [torch.nn.Parameter(b.clone()) for b in bs]
    cache = T.PackCache()
    out, h_alias = T.self_attention21(hg, wg[0], bg[0], wg[1], bg[1], wg[2], bg[2], ("q.weight", "k.weight", "v.weight"), cache, H, 32 ** -0.5)
    # the alias carries a second consumer's gradient (the layer's residual path) into the same backward: d h = data gradient + that gradient
    g2 = torch.randn(B, Tn, Cc, generator=g).to(dev)
    torch.autograd.backward([out, h_alias], [dctx, g2])
    hr.grad += g2
    assert torch.equal(out, ref) and torch.equal(h_alias, h)
    rel = lambda a, r: float((a - r).abs().max()) / max(float(r.abs().max()), 1e-6)
    assert rel(hg.grad, hr.grad) <= 2e-6
    for a, r in zip(wg + bg, wr + br):
        assert rel(a.grad, r.grad) <= 2e-6
    # the persistent operand follows the parameters: change them in place, refresh, compare again
    with torch.no_grad():
        for p_, r_ in zip(wg + bg, wr + br):
            p_.mul_(0.5)
            r_.mul_(0.5)
    cache.refresh()
    out2 = T.self_attention21(h, wg[0], bg[0], wg[1], bg[1], wg[2], bg[2], ("q.weight", "k.weight", "v.weight"), cache, H, 32 ** -0.5)[0]
    q, k, v = (T.linear_hip(h, w, b) for w, b in zip(wr, br))
    assert torch.equal(out2, T.attn21(q, k, v, H, 32 ** -0.5))


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_drop_add_ln_matches_torch_with_the_drawn_mask(p):
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    o, h = torch.randn(7, 21, 128, generator=g).to(dev), torch.randn(7, 21, 128, generator=g).to(dev)
    w, b = (1 + 0.1 * torch.randn(128, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
    dy = torch.randn(7, 21, 128, generator=g).to(dev)
    rng = torch.tensor([1234, 5], dtype=torch.int64, device=dev) if p > 0 else None
    og, hgr, wg, bg = (t.clone().requires_grad_(True) for t in (o, h, w, b))
    y = T.drop_add_ln(og, hgr, wg, bg, 1e-12, p, rng, 3)
    y.backward(dy)
    # recover the mask the kernel drew from its effect on d o / d h (d o = d h * mask / (1 - p)), then replay the expression in torch
    if p > 0:
        mask = (og.grad != 0).double()
        assert 0.6 < float(mask.mean()) < 0.9, "keep rate ~ 1 - p"
        y2 = T.drop_add_ln(o, h, w, b, 1e-12, p, rng, 3)
        assert torch.equal(y2, y.detach()), "same (seed, counter, call): same mask"
        y3 = T.drop_add_ln(o, h, w, b, 1e-12, p, rng, 4)
        assert not torch.equal(y3, y.detach()), "another call id: another mask"
    else:
        mask = torch.ones_like(o).double()
    od, hd, wd, bd = (t.double().clone().requires_grad_(True) for t in (o, h, w, b))
    yr = F.layer_norm(hd + od * mask / (1 - p), (128,), wd, bd, 1e-12)
    yr.backward(dy.double())
    for a, r, what in ((y, yr, "y"), (og.grad, od.grad, "do"), (hgr.grad, hd.grad, "dh"), (wg.grad, wd.grad, "dw"), (bg.grad, bd.grad, "db")):
        assert float((a.double() - r).abs().max()) <= 3e-5 * max(1.0, float(r.abs().max())), what


@pytest.mark.parametrize("prec,G", [("f32", 1), ("bf16", 1), ("f32", 2), ("bf16", 2)])
def test_conv_with_residual_epilogue_matches_conv_plus_add(prec, G):
    """Conv2dNHWC(res=...): y = conv(x) + b + res in the GEMM's epilogue, the residual's gradient = dY (round 4: Residual.conv3 + skip)."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, N = 4, 16, 16, 64, 128
    tdt = torch.float32 if prec == "f32" else torch.bfloat16
    x = torch.randn(B, H, W, G * Cin, generator=g).to(dev).to(tdt)
    r = torch.randn(B, H, W, G * N, generator=g).to(dev).to(tdt)
    w = (torch.randn(G * N, Cin, 1, 1, generator=g) * Cin ** -0.5).to(dev)
    b = torch.randn(G * N, generator=g).to(dev)
    dy = torch.randn(B, H, W, G * N, generator=g).to(dev).to(tdt)
    outs = []
    for fused in (True, False):
        xg, rg, wg, bg = x.clone().requires_grad_(True), r.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = T.conv2d_nhwc(xg, wg, bg, 1, 0, prec, None, None, None, G, rg) if fused else T.conv2d_nhwc(xg, wg, bg, 1, 0, prec, None, None, None, G) + rg
        y.backward(dy)
        outs.append((y.detach(), xg.grad, rg.grad, wg.grad, bg.grad))
    tol = 1e-6 if prec == "f32" else 1e-2  # (bf16: the fused form rounds once, conv + add rounds twice)
    for a, r_, what in zip(outs[0], outs[1], ("y", "dx", "dres", "dw", "db")):
        assert float((a.float() - r_.float()).abs().max()) <= tol * max(1.0, float(r_.float().abs().max())), what


def test_ball_group_matches_the_op_by_op_grouping_on_its_own_index_sets():
    """training.BallGroup (round 4): grouped feature differences, scaled offsets and the gradients towards point / joint features against the
    torch expression of model/model.py:166-175 evaluated on the index sets the kernel chose; and a Linear reading the strided outputs in place."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    B, N, Jn, Cc = 3, 512, 21, 128
    g = torch.Generator().manual_seed(21)
    pcl = (torch.rand(B, N, 3, generator=g) * 1.0 - 0.5).to(dev)
    node = (torch.rand(B, Jn, 3, generator=g) * 0.8 - 0.4).to(dev)
    pf = torch.randn(B, N, Cc, generator=g).to(dev).requires_grad_(True)
    nf = torch.randn(B, Jn, Cc, generator=g).to(dev).requires_grad_(True)
    outs = T.ball_group(pcl, node, pf, nf)
    idx = outs[6]
    w = (torch.randn(64, Cc, generator=g) * Cc ** -0.5).to(dev).requires_grad_(True)
    ys = [T.linear_hip(outs[2 * i], w) for i in range(3)]  # (reads the [.., 132]-strided rows in place, forward and weight gradient)
    dys = [torch.randn(B * Jn * 64, 64, generator=g).to(dev) for _ in range(3)]
    torch.autograd.backward(ys, dys)
    gp, gn, gw = pf.grad.clone(), nf.grad.clone(), w.grad.clone()
    pf.grad = nf.grad = w.grad = None
    xyz, feat = torch.cat((pcl, node), 1).double(), torch.cat((pf, nf), 1).double()
    ref_ys = []
    for i, r in enumerate((0.1, 0.2, 0.4)):
        flat = idx[i].view(B, Jn * 64).long()
        gx = torch.gather(xyz, 1, flat.unsqueeze(-1).expand(-1, -1, 3)).view(B, Jn, 64, 3) - node.double().unsqueeze(2)
        gf = torch.gather(feat, 1, flat.unsqueeze(-1).expand(-1, -1, Cc)).view(B, Jn, 64, Cc) - nf.double().unsqueeze(2)
        assert float((outs[2 * i].double() - gf.reshape(-1, Cc)).abs().max()) <= 1e-6
        assert float((outs[2 * i + 1][:, :3].double() - (gx / r).reshape(-1, 3)).abs().max()) <= 1e-5 and float(outs[2 * i + 1][:, 3].abs().max()) == 0.0
        ref_ys.append(gf.reshape(-1, Cc) @ w.double().t())
    torch.autograd.backward(ref_ys, [d.double() for d in dys])
    for a, r_, what in ((ys[0], ref_ys[0], "y"), (gp, pf.grad, "d point features"), (gn, nf.grad, "d joint features"), (gw, w.grad, "dw")):
        assert float((a.double() - r_.double()).abs().max()) <= 3e-5 * max(1.0, float(r_.abs().max())), what


@pytest.mark.parametrize("prec,G", [("f32", 1), ("bf16", 1), ("bf16", 2), ("f32", 2)])
def test_linear_of_gelu_with_the_gelu_gradient_in_the_data_gradient_epilogue(prec, G):
    """Conv2dNHWC(gelu_in=True) (round 4: KPF_RES_GELU_GRAD): y = Linear(gelu(z)), d z = (dY W) * gelu'(z) from the GEMM's epilogue, against
    gelu_rows + linear_hip (the two-launch form) and, in fp32, against torch."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(17)
    M, K, N = 2048, 384, 96
    tdt = torch.float32 if prec == "f32" else torch.bfloat16
    z = (1.5 * torch.randn(M, G * K, generator=g)).to(dev).to(tdt)
    w = (torch.randn(G * N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(G * N, generator=g).to(dev)
    dy = torch.randn(M, G * N, generator=g).to(dev).to(tdt)
    res = []
    for fused in (True, False):
        zg, wg, bg = z.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = T.linear_hip(zg, wg, bg, prec, None, None, None, G, True) if fused else T.linear_hip(T.gelu_rows(zg), wg, bg, prec, None, None, None, G)
        y.backward(dy)
        res.append((y.detach(), zg.grad, wg.grad, bg.grad))
    tol = 2e-6 if prec == "f32" else 1e-2  # (bf16: the fused form rounds d z once)
    for a, r, what in zip(res[0], res[1], ("y", "dz", "dw", "db")):
        assert float((a.float() - r.float()).abs().max()) <= tol * max(1.0, float(r.float().abs().max())), what
    if prec == "f32" and G == 1:
        zr, wr = z.double().requires_grad_(True), w.double().requires_grad_(True)
        (F.gelu(zr) @ wr.t() + b.double()).backward(dy.double())
        assert float((res[0][1].double() - zr.grad).abs().max()) <= 2e-5 * float(zr.grad.abs().max())


def test_linear_alias_folds_the_residual_gradient_into_the_data_gradient():
    """linear_hip(alias=True): y and x itself; a gradient arriving through the alias is added in the data-gradient GEMM's residual epilogue."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(23)
    x = torch.randn(7, 21, 128, generator=g).to(dev)
    w, b = (torch.randn(512, 128, generator=g) * 128 ** -0.5).to(dev), torch.randn(512, generator=g).to(dev)
    dy, g2 = torch.randn(7, 21, 512, generator=g).to(dev), torch.randn(7, 21, 128, generator=g).to(dev)
    xa, wa, ba = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y, xal = T.linear_hip(xa, wa, ba, alias=True)
    torch.autograd.backward([y, xal], [dy, g2])
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = T.linear_hip(xr, wr, br)
    torch.autograd.backward([yr, xr * 1.0], [dy, g2])
    assert torch.equal(y, yr) and torch.equal(xal, x) and torch.equal(wa.grad, wr.grad) and torch.equal(ba.grad, br.grad)
    assert float((xa.grad - xr.grad).abs().max()) <= 2e-6 * float(xr.grad.abs().max())
    x2 = x.clone().requires_grad_(True)  # only the alias used
    T.linear_hip(x2, w, b, alias=True)[1].backward(g2)
    assert torch.equal(x2.grad, g2)


def test_add_relu_matches_torch():
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    ts = [torch.randn(5, 33, 128, generator=g).to(dev) for _ in range(3)]
    dy = torch.randn(5, 33, 128, generator=g).to(dev)
    for n, scale in ((2, 1.0), (3, 1.0), (2, 0.5), (1, 1.0)):
        a = [t.clone().requires_grad_(True) for t in ts[:n]]
        r = [t.clone().requires_grad_(True) for t in ts[:n]]
        y = T.add_relu(*a, scale=scale)
        y.backward(dy)
        yr = F.relu(sum(r) * scale)
        yr.backward(dy)
        assert torch.allclose(y, yr, rtol=0, atol=1e-6)
        for x, xr in zip(a, r):
            assert torch.allclose(x.grad, xr.grad, rtol=0, atol=1e-6)


def test_gate_mix_matches_torch():
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    B, J, P = 3, 21, 1024
    logits = torch.randn(B * P, J, generator=g).to(dev)
    gam = torch.rand(B, J, P, generator=g).to(dev)
    wd = torch.tensor([0.3]).to(dev)
    wf = (0.05 * torch.randn(1, P, generator=g)).to(dev)
    d_sw, d_gw = torch.randn(B, J, P, generator=g).to(dev), torch.randn(B, J, P, generator=g).to(dev)
    a = [t.clone().requires_grad_(True) for t in (logits, gam, wd, wf)]
    sw, gw = T.GateMix.apply(*a)
    torch.autograd.backward([sw, gw], [d_sw, d_gw])
    r = [t.double().clone().requires_grad_(True) for t in (logits, gam, wd, wf)]
    swr = torch.sigmoid(r[0].view(B, P, J).permute(0, 2, 1))
    w = torch.sigmoid(r[2])
    gwr = (w * r[1] + (1 - w) * swr) * r[3].view(1, 1, P)
    torch.autograd.backward([swr, gwr], [d_sw.double(), d_gw.double()])
    assert float((sw.double() - swr).abs().max()) <= 1e-6 and float((gw.double() - gwr).abs().max()) <= 1e-6
    for x, xr, what in zip(a, r, ("d logits", "d gam", "d weight_dis", "d w_fc")):
        assert float((x.grad.double() - xr.grad).abs().max()) <= 3e-5 * max(1.0, float(xr.grad.abs().max())), what


def test_row_gather_with_a_shared_inversion_is_bit_identical():
    """RowGather(inv=row_gather_invert(idx, P)): the backward that reuses one inversion for several gathers returns the same bits as the one that inverts itself."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(8)
    B, P, R, G, Cc = 3, 1024, 700, 4, 128
    src = torch.randn(B, P, Cc, generator=g).to(dev)
    idx = torch.randint(0, P, (B, R, G), generator=g, dtype=torch.int32).to(dev)
    w = torch.rand(B, R, G, generator=g).to(dev)
    dout = torch.randn(B, R, Cc, generator=g).to(dev)
    inv = T.row_gather_invert(idx, P)
    grads = []
    for use in (None, inv):
        s_ = src.clone().requires_grad_(True)
        T.row_gather(s_, idx, w, use).backward(dout)
        grads.append(s_.grad)
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("prec,G", [("f32", 1), ("bf16", 1), ("bf16", 2)])
def test_linear_with_gelu_forward_in_its_epilogue(prec, G):
    """Conv2dNHWC(gelu_out=True) (round 4: KPF_ACT_GELU_SAVE): (z, gelu(z)) from one GEMM launch; chained with the next layer's gelu_in / g_pre the pair
    Linear -> GELU -> Linear equals the three-launch form, forward and every gradient."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(19)
    M, K, Hd = 2048, 96, 384
    tdt = torch.float32 if prec == "f32" else torch.bfloat16
    x = torch.randn(M, G * K, generator=g).to(dev).to(tdt)
    w1, b1 = (torch.randn(G * Hd, K, generator=g) * K ** -0.5).to(dev), torch.randn(G * Hd, generator=g).to(dev)
    w2, b2 = (torch.randn(G * K, Hd, generator=g) * Hd ** -0.5).to(dev), torch.randn(G * K, generator=g).to(dev)
    dy = torch.randn(M, G * K, generator=g).to(dev).to(tdt)
    res = []
    for fused in (True, False):
        t = [v.clone().requires_grad_(True) for v in (x, w1, b1, w2, b2)]
        if fused:
            z, gz = T.linear_hip(t[0], t[1], t[2], prec, None, None, None, G, False, False, True)
            y = T.linear_hip(z, t[3], t[4], prec, None, None, None, G, True, False, False, gz)
        else:
            y = T.linear_hip(T.gelu_rows(T.linear_hip(t[0], t[1], t[2], prec, None, None, None, G)), t[3], t[4], prec, None, None, None, G)
        y.backward(dy)
        res.append([y.detach()] + [v.grad for v in t])
    tol = 3e-6 if prec == "f32" else 1.5e-2
    for a, r, what in zip(res[0], res[1], ("y", "dx", "dw1", "db1", "dw2", "db2")):
        assert float((a.float() - r.float()).abs().max()) <= tol * max(1.0, float(r.float().abs().max())), what


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_operand_refresh_forms_equal_the_single_operand_kernel(prec):
    """PackCache's one-launch refresh (kpf_pack_conv_weights_multi: LDS-staged forms per operand geometry, round 4) against kpf_pack_conv_weight (one element per
    thread) for every mode and a spread of shapes: bit-identical operands, padding included."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(41)
    cases = [((96, 48, 3, 3), 0, {}), ((128, 256, 3, 3), 1, dict(pad=1, n_pad=128)), ((100, 40, 3, 3), 1, dict(pad=1, n_pad=104)), ((192, 96, 2, 2), 0, dict(stride=2)),
             ((192, 96, 2, 2), 2, dict(n_pad=192)), ((384, 1, 7, 7), 2, dict(n_pad=384)), ((200, 1, 7, 7), 3, dict(n_pad=200)), ((64, 8, 4, 4), 0, dict(stride=4)),
             ((384, 96, 1, 1), 0, {}), ((384, 96, 1, 1), 1, dict(n_pad=384)), ((3, 128, 1, 1), 1, dict(n_pad=4)), ((105, 128, 1, 1), 0, {}), ((64, 20, 5, 5), 0, dict(pad=2))]
    cache = T.PackCache()
    ws, refs = [], []
    for i, (shape, mode, kw) in enumerate(cases):
        w = torch.randn(shape, generator=g).to(dev)
        ws.append(w)
        pc = cache.get(("k%d" % i, mode), w, None, mode, prec, **kw)           # registered: packed by the single-operand kernel
        refs.append((pc.w if pc.w is not None else pc.w16).clone())
    for e in cache.entries.values():                                            # poison the operands, then let the one-launch refresh rewrite them
        d = e["desc"]
        buf = e["pc"].w if e["pc"].w is not None else e["pc"].w16
        buf.fill_(float("nan"))
    cache.refresh()
    torch.cuda.synchronize()
    for (shape, mode, kw), e, r in zip(cases, cache.entries.values(), refs):
        buf = e["pc"].w if e["pc"].w is not None else e["pc"].w16
        assert torch.equal(buf.view(torch.int16 if buf.dtype != torch.float32 else torch.int32), r.view(torch.int16 if r.dtype != torch.float32 else torch.int32)), (shape, mode)


def _bert_stack_reference(e, pos, prm, masks=None):
    """The four post-LN BERT layers in plain torch (any dtype): model/model.py:30-126 with dropout off, or with given keep masks (None: none)."""
    B = e.shape[0]
    h = e + pos
    for l in range(4):
        Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, b1, Wi, bi, Wo2, bo2, g2, b2 = prm[16 * l:16 * l + 16]
        q, k, v = (F.linear(h, W, b_).view(B, 21, 4, 32).transpose(1, 2) for W, b_ in ((Wq, bq), (Wk, bk), (Wv, bv)))
        p = torch.softmax(q @ k.transpose(-1, -2) / 32 ** 0.5, -1)
        ctx = (p @ v).transpose(1, 2).reshape(B, 21, 128)
        h1 = F.layer_norm(h + F.linear(ctx, Wo, bo), (128,), g1, b1, 1e-12)
        it = F.linear(h1, Wi, bi)
        g = 0.5 * it * (1 + torch.erf(it / 2 ** 0.5))
        h = F.layer_norm(h1 + F.linear(g, Wo2, bo2), (128,), g2, b2, 1e-12)
    return h


def _bert_stack_params(gen, dev, scale=1.0):
    shapes = [(128, 128), (128,), (128, 128), (128,), (128, 128), (128,), (128, 128), (128,), (128,), (128,), (16, 128), (16,), (128, 16), (128,), (128,), (128,)]
    prm = []
    for l in range(4):
        for i, sh in enumerate(shapes):
            if i in (8, 14):  # LayerNorm weights around 1
                t = 1.0 + 0.2 * torch.randn(*sh, generator=gen)
            elif len(sh) == 2:
                t = torch.randn(*sh, generator=gen) * (scale / sh[1] ** 0.5)
            else:
                t = 0.1 * torch.randn(*sh, generator=gen)
            prm.append(t.to(dev).requires_grad_(True))
    return prm


@pytest.mark.parametrize("B", [1, 3, 32])
def test_bert_stack21_forward_backward_match_fp64_torch(B):
    """training.BertStack21 (csrc/kpf_trstack.hip: the four BERT layers of a 21-token stack as one launch each way) against the same layers written out in fp64
    torch, dropout off: output, input / position gradients and all 64 parameter gradients (24 through the grouped weight-gradient launch, 8 LayerNorm
    parameter sums through the grouped column-sum reduce), and two calls give the same bits."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7 + B)
    prm = _bert_stack_params(gen, dev, scale=1.5)
    e = torch.randn(B, 21, 128, generator=gen).to(dev).requires_grad_(True)
    pos = (0.5 * torch.randn(21, 128, generator=gen)).to(dev).requires_grad_(True)
    names = ["stack.%d.%s" % (l, k) for l in range(4) for k in T.BertStack21.ORDER]
    wsum = torch.randn(B, 21, 128, generator=gen).to(dev)

    def run():
        for t in [e, pos] + prm:
            t.grad = None
        out = T.bert_stack21(e, pos, names, None, 0.0, None, 1, prm)
        (out * wsum).sum().backward()
        return out.detach().clone(), [t.grad.detach().clone() for t in [e, pos] + prm]

    out, grads = run()
    out2, grads2 = run()
    assert torch.equal(out, out2) and all(torch.equal(a, c) for a, c in zip(grads, grads2)), "two calls on the same operands must give the same bits"
    e64, pos64 = e.detach().double().cpu().requires_grad_(True), pos.detach().double().cpu().requires_grad_(True)
    prm64 = [t.detach().double().cpu().requires_grad_(True) for t in prm]
    ref = _bert_stack_reference(e64, pos64, prm64)
    (ref * wsum.double().cpu()).sum().backward()
    assert float((out.double().cpu() - ref.detach()).abs().max() / ref.detach().abs().max()) < 2e-5
    for i, (gd, t64) in enumerate(zip(grads, [e64, pos64] + prm64)):
        # (the key biases' gradient is analytically ZERO — a constant added to every key shifts a softmax row uniformly — so the fp64 reference holds 1e-16
        #  there and the fp32 kernel rounding noise of sums of O(10) terms: checked in absolute terms)
        if i >= 2 and names[i - 2].endswith("key.bias"):
            assert float(gd.abs().max()) < 1e-4 and float(t64.grad.abs().max()) < 1e-10, "key bias: zero gradient"
            continue
        err = float((gd.double().cpu() - t64.grad).abs().max() / (t64.grad.abs().max() + 1e-4))
        assert err < 2e-4, "gradient %d (%s): relative error %.2e" % (i, "e pos".split()[i] if i < 2 else names[i - 2], err)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_bert_stack21_16bit_products_stay_close_to_the_fp32_stack(prec):
    """The mixed-precision form of the fused stacks (operands of every product rounded to bf16 / f16 in registers, fp32 accumulation, everything else fp32):
    output and gradients within a few 16-bit ulps (times the depth of the stack) of the fp32 form on the same operands, bit-identical between two calls."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    B = 8
    gen = torch.Generator().manual_seed(11)
    prm = _bert_stack_params(gen, dev)
    e = torch.randn(B, 21, 128, generator=gen).to(dev).requires_grad_(True)
    pos = (0.5 * torch.randn(21, 128, generator=gen)).to(dev).requires_grad_(True)
    names = ["stack.%d.%s" % (l, k) for l in range(4) for k in T.BertStack21.ORDER]
    wsum = torch.randn(B, 21, 128, generator=gen).to(dev)

    def run(p):
        for t in [e, pos] + prm:
            t.grad = None
        out = T.bert_stack21(e, pos, names, None, 0.0, None, 1, prm, p)
        (out * wsum).sum().backward()
        return out.detach().clone(), [t.grad.detach().clone() for t in [e, pos] + prm]

    o32, g32 = run("f32")
    o16, g16 = run(prec)
    o16b, g16b = run(prec)
    assert torch.equal(o16, o16b) and all(torch.equal(a, c) for a, c in zip(g16, g16b))
    ulp = 2.0 ** -8 if prec == "bf16" else 2.0 ** -11
    rel = lambda a, c: float((a - c).norm() / (c.norm() + 1e-12))
    assert 1e-6 < rel(o16, o32) < 40 * ulp, rel(o16, o32)  # (not the fp32 kernel by accident; within the rounding of ~25 chained products)
    errs = sorted(rel(a, c) for i, (a, c) in enumerate(zip(g16, g32)) if not (i >= 2 and names[i - 2].endswith("key.bias")))
    print("fused stack %s vs fp32: output %.2e, gradients median %.2e max %.2e" % (prec, rel(o16, o32), errs[len(errs) // 2], errs[-1]))
    assert errs[len(errs) // 2] < 60 * ulp and errs[-1] < 400 * ulp, (errs[len(errs) // 2], errs[-1])


def test_bert_stack21_dropout_masks_are_consistent_between_forward_and_backward():
    """p = 0.1: (a) about a tenth of H[0] = dropout(e + pos) is zero and the rest is scaled by 1 / 0.9; (b) the same (seed, counter) gives the same bits, another
    counter another mask; (c) the backward uses the forward's masks: a directional derivative of sum(out * w) along random directions in e and in two weights
    (central differences with the masks held fixed: they depend on (seed, counter, call, element) only) matches <grad, direction>."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    B = 4
    gen = torch.Generator().manual_seed(3)
    prm = _bert_stack_params(gen, dev)
    e = torch.randn(B, 21, 128, generator=gen).to(dev).requires_grad_(True)
    pos = (0.5 * torch.randn(21, 128, generator=gen)).to(dev).requires_grad_(True)
    names = ["stack.%d.%s" % (l, k) for l in range(4) for k in T.BertStack21.ORDER]
    wsum = torch.randn(B, 21, 128, generator=gen).to(dev)
    rng = torch.tensor([1234567, 5], dtype=torch.int64, device=dev)
    f = lambda: T.bert_stack21(e, pos, names, None, 0.1, rng, 1, prm)
    out = f()
    h0 = out.grad_fn.saved_tensors[0][:B * 21 * 128].view(B, 21, 128).clone()  # H[0] is the first block of the saved buffer
    (out * wsum).sum().backward()
    g_e, g_wq, g_wo2 = e.grad.clone(), prm[0].grad.clone(), prm[16 * 2 + 12].grad.clone()
    assert torch.equal(out, f())
    rng2 = rng.clone()
    rng2[1] += 1
    assert not torch.equal(out, T.bert_stack21(e, pos, names, None, 0.1, rng2, 1, prm))
    # zero fraction and scale of H[0]
    with torch.no_grad():
        zero = (h0 == 0)
        assert 0.07 < float(zero.float().mean()) < 0.13
        assert torch.allclose(h0[~zero], ((e + pos) / 0.9)[~zero], rtol=1e-6, atol=1e-6)
        for t, g in ((e, g_e), (prm[0], g_wq), (prm[16 * 2 + 12], g_wo2)):
            d = torch.randn(t.shape, generator=gen).to(dev)
            d = d / d.norm()
            eps = 2e-2 * float(t.norm()) / 10
            base = t.detach().clone()
            t.copy_(base + eps * d)
            lp = float((f().double() * wsum.double()).sum())
            t.copy_(base - eps * d)
            lm = float((f().double() * wsum.double()).sum())
            t.copy_(base)
            fd, an = (lp - lm) / (2 * eps), float((g.double() * d.double()).sum())
            assert abs(fd - an) < 3e-2 * max(abs(an), abs(fd)) + 1e-3, (fd, an)


@pytest.mark.parametrize("net", ["convnext-tiny", "resnet-18"])
def test_grouped_desa_and_fused_stacks_match_the_layer_by_layer_training_graph(net, monkeypatch):
    """Round 6: DESA's three radii as one channel-stacked chain (KPF_DESA_GROUPED: grouped Linears, BatchNorm / ReLU / group maximum over 3 x 128 channels,
    BallGroup3 / LinearSlices / GroupMax), the sibling point / joint embeddings as one channel-stacked tensor (KPF_EMB_GROUPED: LinearCat, one BatchNorm pass,
    SlicesSumRelu) and the 21-token stacks as one launch each way (KPF_TR_FUSED: BertStack21) against the radius-by-radius, layer-by-layer
    graph: same outputs, loss, BatchNorm running statistics and gradients of EVERY parameter to fp32 rounding.  (The reference-gradient test runs with given
    ball-query sets, which takes the op-by-op DESA: this test is what pins the grouped form, on the computed sets.)"""
    from conftest import synthetic_sd
    from keypointfusion_amd import train_graph as TG
    from keypointfusion_amd import training as T
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.weights import synthetic_batch
    net, B, dev = "KPFusion-" + net, 4, torch.device("cuda:0")
    batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=6).items()}
    g = torch.Generator().manual_seed(2)
    uvd, xyz = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev), (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)

    class Loader:
        img_size, flip = 128, 1

    def run(new):
        monkeypatch.setattr(TG, "DESA_GROUPED", new)
        monkeypatch.setattr(TG, "EMB_GROUPED", new)
        monkeypatch.setattr(TG, "TR_FUSED", new)
        monkeypatch.setattr(TG, "XATTN_FUSED", new)
        monkeypatch.setattr(TG, "UNSTACK_FUSED", new)
        monkeypatch.setattr(TG, "BN2_FUSED", new)
        m = KPFusion(net, "", 21, "dexycb", "")
        m.load_state_dict(synthetic_sd(net), strict=True)
        m = m.to(dev).train()
        m.train_dropout = 0.0
        r, s, _ = m(batch["img_rgb"], batch["img"], batch["pcl"], Loader(), batch["center"], batch["M"], batch["cube"], batch["cam_para"], 0.8)
        loss = T.kpfusion_loss(r, s, batch["img"], uvd, xyz, epoch=0)[0]
        loss.backward()
        bufs = {k: v.detach().clone() for k, v in m.named_buffers()}
        return float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}, bufs, [t.detach().clone() for t in r + s]

    l0, g0, b0, r0 = run(False)
    l1, g1, b1, r1 = run(True)
    rel = lambda a, b: float((a.float() - b.float()).abs().max()) / max(float(a.float().abs().max()), 1e-6)
    assert set(g0) == set(g1), sorted(set(g0) ^ set(g1))[:8]
    for a, b in zip(r0, r1):
        assert rel(a, b) <= 1e-4, rel(a, b)
    assert abs(l0 - l1) <= 1e-4 * abs(l0), (l0, l1)
    floor = 1e-5 * max(float(v.float().abs().max()) for v in g0.values())  # (biases in front of a batch-statistics BatchNorm have an exactly zero gradient)
    relf = lambda a, b: float((a.float() - b.float()).abs().max()) / max(float(a.float().abs().max()), floor)
    errs = sorted((relf(g0[k], g1[k]), k) for k in g0)
    print("grouped DESA + fused stacks vs layer by layer (%s): loss %.6f / %.6f, gradient error median %.2e, worst %s" % (net, l0, l1, errs[len(errs) // 2][0], errs[-3:]))
    assert errs[len(errs) // 2][0] <= 1e-3 and errs[-1][0] <= 5e-2, (errs[len(errs) // 2], errs[-8:])
    for k in b0:
        assert torch.allclose(b0[k].float(), b1[k].float(), rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 105, 112, 224, True), (2, 128, 128, 256, False), (2, 105, 112, 232, True), (3, 20, 24, 80, False), (2, 7, 7, 14, False), (2, 36, 40, 80, True)])
def test_unstack_rows_forward_backward(cfg, dt):
    """training.UnstackRows (kpf_unstack_rows / kpf_restack_rows): the channel-stacked maps of G networks as G dense fp32 maps, NHWC or NCHW, and the gradients back
    into one stacked tensor with zero pad columns — every form of the kernels (four columns per thread, element per thread, tiled NCHW transposes) against slicing."""
    from keypointfusion_amd import training as T
    G, Cc, gs, ld, nchw = cfg
    g = torch.Generator().manual_seed(Cc + ld)
    B, H, W = 3, 9, 8
    y = torch.randn(B, H, W, ld, generator=g).to(dt).cuda().requires_grad_(True)
    outs = T.unstack_rows(y, G, Cc, gs, nchw)
    ws = [torch.randn(o.shape, generator=g).cuda() for o in outs]
    want = [y.detach().float()[..., i * gs:i * gs + Cc] for i in range(G)]
    if nchw:
        want = [w_.permute(0, 3, 1, 2) for w_ in want]
    for o, w_ in zip(outs, want):
        assert o.dtype == torch.float32 and o.is_contiguous() and torch.equal(o, w_.contiguous())
    sum((o * w_).sum() for o, w_ in zip(outs[:-1], ws[:-1])).backward()  # (the last map receives no gradient: its columns must come back zero)
    ref = torch.zeros(B, H, W, ld)
    for i in range(G - 1):
        gi = ws[i].cpu()
        ref[..., i * gs:i * gs + Cc] = gi.permute(0, 2, 3, 1) if nchw else gi
    assert y.grad.dtype == dt and torch.equal(y.grad.float().cpu(), ref.to(dt).float())


@pytest.mark.parametrize("M,Cc", [(43008 // 8, 384), (4097, 48), (37, 8)])
def test_bn2_add_relu_matches_two_batchnorms_an_add_and_a_relu(M, Cc):
    """training.Bn2AddRelu (kpf_bn2_add_relu_forward / _backward) against relu(F.batch_norm(xa) + F.batch_norm(xb)) in float64: output, both input gradients, all four
    parameter gradients, both pairs of running statistics."""
    from keypointfusion_amd import training as T
    g = torch.Generator().manual_seed(M + Cc)
    mk = lambda *sh: torch.randn(*sh, generator=g)
    xa, xb = mk(M, Cc) * 2.0 + 3.0 * mk(Cc), mk(M, Cc) * 0.5 - 2.0 * mk(Cc)
    wa, ba, wb, bb = torch.rand(Cc, generator=g) + 0.5, mk(Cc), torch.rand(Cc, generator=g) + 0.5, mk(Cc)
    dy = mk(M, Cc)
    dev = lambda t: t.cuda().requires_grad_(True)
    xad, xbd, wad, bad, wbd, bbd = map(dev, (xa, xb, wa, ba, wb, bb))
    rs = [torch.zeros(Cc).cuda(), torch.ones(Cc).cuda(), torch.zeros(Cc).cuda(), torch.ones(Cc).cuda()]
    out = T.bn2_add_relu(xad, xbd, wad, bad, wbd, bbd, rs[0], rs[1], rs[2], rs[3], 0.1, 1e-5)
    out.backward(dy.cuda())
    ref_in = [t.double().requires_grad_(True) for t in (xa, xb, wa, ba, wb, bb)]
    rr = [torch.zeros(Cc).double(), torch.ones(Cc).double(), torch.zeros(Cc).double(), torch.ones(Cc).double()]
    ref = torch.relu(F.batch_norm(ref_in[0], rr[0], rr[1], ref_in[2], ref_in[3], True, 0.1, 1e-5) + F.batch_norm(ref_in[1], rr[2], rr[3], ref_in[4], ref_in[5], True, 0.1, 1e-5))
    ref.backward(dy.double())
    rel = lambda a, r: float((a.detach().cpu().double() - r.detach()).abs().max()) / max(float(r.detach().abs().max()), 1e-6)
    assert rel(out, ref) < 2e-5
    for got, want in zip((xad, xbd, wad, bad, wbd, bbd), ref_in):
        assert rel(got.grad, want.grad) < 2e-4, (tuple(want.shape), rel(got.grad, want.grad))
    for got, want in zip(rs, rr):
        assert rel(got, want) < 1e-5


@pytest.mark.parametrize("G,group,Cc,ld", [(672 // 4, 64, 384, 512), (33, 64, 48, 48), (5, 7, 8, 12)])
def test_bn_relu_group_max_matches_batchnorm_relu_and_max(G, group, Cc, ld):
    """training.BnReluGroupMax (kpf_bn_relu_gmax_forward / _backward) against F.batch_norm -> relu -> max over `group` consecutive rows in float64: values, the input
    gradient (dense), the parameter gradients and the running statistics; the incoming gradient is a column slice of a wider matrix (read in place)."""
    from keypointfusion_amd import training as T
    g = torch.Generator().manual_seed(G + Cc)
    M = G * group
    x = torch.randn(M, Cc, generator=g) * 1.5 + 2.0 * torch.randn(Cc, generator=g)
    w, b = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) - 0.3
    w[::5] *= -1.0  # (negative scales: the winner is then the member with the SMALLEST pre-activation)
    dyw = torch.randn(G, ld, generator=g)
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    rm, rv = torch.zeros(Cc).cuda(), torch.ones(Cc).cuda()
    y = T.bn_relu_group_max(xd, wd, bd, rm, rv, 0.1, 1e-5, group)
    y.backward(dyw.cuda()[:, :Cc])
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    rmr, rvr = torch.zeros(Cc).double(), torch.ones(Cc).double()
    yr = torch.relu(F.batch_norm(xr, rmr, rvr, wr, br, True, 0.1, 1e-5)).view(G, group, Cc).max(1)[0]
    yr.backward(dyw.double()[:, :Cc])
    rel = lambda a, r: float((a.detach().cpu().double() - r.detach()).abs().max()) / max(float(r.detach().abs().max()), 1e-6)
    assert rel(y, yr) < 2e-5
    assert rel(xd.grad, xr.grad) < 2e-4 and rel(wd.grad, wr.grad) < 2e-4 and rel(bd.grad, br.grad) < 2e-4, (rel(xd.grad, xr.grad), rel(wd.grad, wr.grad), rel(bd.grad, br.grad))
    assert rel(rm, rmr) < 1e-5 and rel(rv, rvr) < 1e-5


@pytest.mark.parametrize("rows,Cc,n1,n2", [(4096, 128, 3, 1), (672, 128, 2, 0), (77, 24, 1, 2), (33, 8, 4, 0)])
def test_bn_slices_sum_relu_matches_batchnorm_then_sums(rows, Cc, n1, n2):
    """training.BnSlicesSumRelu (kpf_bn_ssr_forward / _backward) against F.batch_norm over the n C stacked columns followed by relu(S1) / relu(relu(S1) + S2) in float64:
    output, input gradient, parameter gradients, running statistics."""
    from keypointfusion_amd import training as T
    g = torch.Generator().manual_seed(rows + Cc + n1)
    n = n1 + n2
    x = torch.randn(rows, n * Cc, generator=g) * 1.5 + 2.0 * torch.randn(n * Cc, generator=g)
    w, b = torch.rand(n * Cc, generator=g) + 0.5, torch.randn(n * Cc, generator=g) * 0.5
    dy = torch.randn(rows, Cc, generator=g)
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    rm, rv = torch.zeros(n * Cc).cuda(), torch.ones(n * Cc).cuda()
    out = T.bn_slices_sum_relu(xd, wd, bd, rm, rv, 0.1, 1e-5, Cc, n1, n2)
    out.backward(dy.cuda())
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    rmr, rvr = torch.zeros(n * Cc).double(), torch.ones(n * Cc).double()
    yr = F.batch_norm(xr, rmr, rvr, wr, br, True, 0.1, 1e-5).view(rows, n, Cc)
    ref = torch.relu(yr[:, :n1].sum(1))
    if n2:
        ref = torch.relu(ref + yr[:, n1:].sum(1))
    ref.backward(dy.double())
    rel = lambda a, r: float((a.detach().cpu().double() - r.detach()).abs().max()) / max(float(r.detach().abs().max()), 1e-6)
    assert rel(out, ref) < 2e-5
    assert rel(xd.grad, xr.grad) < 2e-4 and rel(wd.grad, wr.grad) < 2e-4 and rel(bd.grad, br.grad) < 2e-4, (rel(xd.grad, xr.grad), rel(wd.grad, wr.grad), rel(bd.grad, br.grad))
    assert rel(rm, rmr) < 1e-5 and rel(rv, rvr) < 1e-5


@pytest.mark.parametrize("prec,tdt", [("bf16", torch.bfloat16), ("f16", torch.float16)])
def test_head_mma_products_of_rounded_operands(prec, tdt):
    """training.head_mma (KPF_MMA_BF16 / _F16, KPF_DT_F32_MMA_*): a grouped Linear on fp32 tensors whose forward, data gradient and weight gradient multiply the
    operands ROUNDED to the 16-bit type (fp32 accumulation) — against float64 products of the rounded operands (an fp32-product result would miss this bound by two
    orders of magnitude: the test also proves the 16-bit path was taken), DESA's shape (three groups of 128 -> 128 over B * 21 * 64 rows)."""
    from keypointfusion_amd import training as T
    g = torch.Generator().manual_seed(17)
    rows, G, K, N = 43008 // 8, 3, 128, 128
    x = torch.randn(rows, G * K, generator=g)
    w = torch.randn(G * N, K, generator=g) * 0.1
    b = torch.randn(G * N, generator=g)
    dy = torch.randn(rows, G * N, generator=g)
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    with T.head_mma(prec):
        y = T.linear_hip(xd, wd, bd, "f32", None, None, None, G)
    y.backward(dy.cuda())
    r = lambda t: t.to(tdt).double()
    xr, wr, dyr = r(x).view(rows, G, K), r(w).view(G, N, K), r(dy).view(rows, G, N)
    yr = torch.einsum("rgk,gnk->rgn", xr, wr).reshape(rows, G * N) + b.double()
    dxr = torch.einsum("rgn,gnk->rgk", dyr, wr).reshape(rows, G * K)
    dwr = torch.einsum("rgn,rgk->gnk", dyr, xr).reshape(G * N, K)
    rel = lambda a, ref: float((a.detach().cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert rel(y, yr) < 2e-5 and rel(xd.grad, dxr) < 2e-5 and rel(wd.grad, dwr) < 5e-5, (rel(y, yr), rel(xd.grad, dxr), rel(wd.grad, dwr))
    assert rel(bd.grad, dy.double().sum(0)) < 2e-5  # (the bias gradient sums the UNROUNDED dY)
    # outside the block the same call multiplies fp32 operands
    y32 = T.linear_hip(xd.detach(), wd.detach(), bd.detach(), "f32", None, None, None, G)
    y64 = torch.einsum("rgk,gnk->rgn", x.double().view(rows, G, K), w.double().view(G, N, K)).reshape(rows, G * N) + b.double()
    assert rel(y32, y64) < 2e-5 and rel(y, y64) > 1e-4


def test_group_max_and_ball_group3_match_torch():
    """GroupMax (max over 64 consecutive rows with the winner kept) against torch.max and its autograd; BallGroup3 (the three radii channel-stacked, one backward
    launch) against BallGroup (radius by radius): same grouped rows and offsets, same index sets, same gradients towards the point / joint features."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(5 * 64, 384, generator=gen).to(dev).requires_grad_(True)
    w = torch.randn(5, 384, generator=gen).to(dev)
    y = T.group_max(x, 64)
    (y * w).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = x2.view(5, 64, 384).max(1)[0]
    (y2 * w).sum().backward()
    assert torch.equal(y, y2) and torch.equal(x.grad, x2.grad)
    B, N, J = 3, 1024, 21
    pcl = (torch.rand(B, N, 3, generator=gen) * 1.2 - 0.6).to(dev)
    node = (torch.rand(B, J, 3, generator=gen) * 0.8 - 0.4).to(dev)
    pf = torch.randn(B, N, 128, generator=gen).to(dev).requires_grad_(True)
    nf = torch.randn(B, J, 128, generator=gen).to(dev).requires_grad_(True)
    GF3, GX3, idx3 = T.ball_group3(pcl, node, pf, nf)
    wgt = torch.randn(B * J * 64, 384, generator=gen).to(dev)
    (GF3 * wgt).sum().backward()
    g3 = (pf.grad.clone(), nf.grad.clone())
    pf.grad = nf.grad = None
    outs = T.ball_group(pcl, node, pf, nf)
    assert torch.equal(outs[6], idx3)
    for i in range(3):
        assert torch.equal(outs[2 * i], GF3[:, 128 * i:128 * i + 128]) and torch.equal(outs[2 * i + 1], GX3[:, 4 * i:4 * i + 4])
    sum((outs[2 * i] * wgt[:, 128 * i:128 * i + 128]).sum() for i in range(3)).backward()
    for a, c in zip(g3, (pf.grad, nf.grad)):
        assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max()) + 1e-6, float((a - c).abs().max())


@pytest.mark.parametrize("B", [1, 5, 32])
def test_xattn_layer21_forward_backward_match_fp64_torch(B):
    """training.XAttnLayer21 (the decoder layer of a fusion block as one launch each way) against the layer written out in fp64 torch with nn.MultiheadAttention's
    arithmetic (model/transfusion_head.py:437-554: q scaled by head_dim^-1/2, softmax over the 21 keys), dropout off: output, gradients of query / key / both
    position tables and all 12 parameters; two calls give the same bits."""
    from keypointfusion_amd import training as T
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(30 + B)
    mk = lambda *sh, sc=1.0: (sc * torch.randn(*sh, generator=gen)).to(dev).requires_grad_(True)
    prm = [mk(384, 128, sc=0.12), mk(384, sc=0.1), mk(128, 128, sc=0.12), mk(128, sc=0.1), (1 + 0.2 * torch.randn(128, generator=gen)).to(dev).requires_grad_(True), mk(128, sc=0.1),
           mk(128, 128, sc=0.12), mk(128, sc=0.1), mk(128, 128, sc=0.12), mk(128, sc=0.1), (1 + 0.2 * torch.randn(128, generator=gen)).to(dev).requires_grad_(True), mk(128, sc=0.1)]
    query, key, qpos, kpos = mk(B, 21, 128), mk(B, 21, 128), mk(21, 128, sc=0.5), mk(21, 128, sc=0.5)
    names = ["dec." + k for k in T.XAttnLayer21.ORDER]
    wsum = torch.randn(B, 21, 128, generator=gen).to(dev)
    leaves = [query, key, qpos, kpos] + prm

    def run():
        for t in leaves:
            t.grad = None
        out = T.xattn_layer21(query, key, qpos, kpos, names, None, 0.0, None, 1, prm)
        (out * wsum).sum().backward()
        return out.detach().clone(), [t.grad.detach().clone() for t in leaves]

    out, grads = run()
    out2, grads2 = run()
    assert torch.equal(out, out2) and all(torch.equal(a, c) for a, c in zip(grads, grads2))
    l64 = [t.detach().double().cpu().requires_grad_(True) for t in leaves]
    q64, k64, qp64, kp64 = l64[:4]
    Win, bin_, Wo, bo, g2, b2, W1, b1, W2, bb2, g3, b3 = l64[4:]
    qe, ke = q64 + qp64, k64 + kp64
    q = F.linear(qe, Win[:128], bin_[:128]).view(B, 21, 4, 32).transpose(1, 2) * 32 ** -0.5
    k = F.linear(ke, Win[128:256], bin_[128:256]).view(B, 21, 4, 32).transpose(1, 2)
    v = F.linear(ke, Win[256:], bin_[256:]).view(B, 21, 4, 32).transpose(1, 2)
    ctx = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B, 21, 128)
    x = F.layer_norm(q64 + F.linear(ctx, Wo, bo), (128,), g2, b2, 1e-5)
    ref = F.layer_norm(x + F.linear(torch.relu(F.linear(x, W1, b1)), W2, bb2), (128,), g3, b3, 1e-5)
    (ref * wsum.double().cpu()).sum().backward()
    assert float((out.double().cpu() - ref.detach()).abs().max() / ref.detach().abs().max()) < 2e-5
    lab = ["query", "key", "qpos", "kpos"] + names
    for i, (gd, t64) in enumerate(zip(grads, l64)):
        err = float((gd.double().cpu() - t64.grad).abs().max() / (t64.grad.abs().max() + 1e-4))
        assert err < 2e-4, "gradient of %s: relative error %.2e" % (lab[i], err)
