"""Reduced-precision path (BASELINE configs[2] bf16 eval, configs[4] ConvNeXt-B 512x512 fp16): 16-bit storage, fp32 accumulation.
The reference has no such mode, so the contract is a stated tolerance against the fp32 oracle:
  * per-op: a 16-bit GEMM / depthwise+LN equals the fp32 op on the SAME 16-bit-rounded operands up to output rounding (2^-8 bf16,
    2^-11 f16 relative) — i.e. accumulation is fp32 and nothing else is lost;
  * end to end (the tolerance study, on synthetic weights and crops, cube 250 mm): MEAN joint deviation from the fp32 oracle — the
    quantity BASELINE.json's accuracy metric averages — below 4 mm (bf16) / 1.5 mm (f16) per stage, final stage below 1.5 / 0.6 mm;
    dense maps within 6e-2 / 8e-3 of their range.  The MAX over joints is reported, not bounded tightly: the head contains integer
    decisions (ball-query membership, top-4 pixels) taken around network outputs, and a joint whose neighbourhood changes by one point
    moves by centimetres with untrained weights (measured: f16 mean 0.2-0.7 mm, max 18 mm on one joint of 84);
  * size-independent properties at the full configs[4] size (ConvNeXt-B, 512x512): sample independence, batch-position independence,
    determinism, finite outputs; the fp32 oracle on a subset.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import synthetic_sd
from keypointfusion_amd.weights import synthetic_batch

pytestmark = pytest.mark.gpu
PREC = {"bf16": (torch.bfloat16, 2.0 ** -8), "f16": (torch.float16, 2.0 ** -11)}


def _dev():
    assert torch.cuda.is_available()
    from keypointfusion_amd import lib
    lib.load()
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("case", [
    # B, Cin, H, W, N, k, stride, pad, flags
    (2, 96, 16, 16, 384, 1, 1, 0, "gelu"),     # Cin % 64 != 0: general staging path with a K mask
    (2, 384, 16, 16, 96, 1, 1, 0, "res"),
    (1, 128, 32, 32, 105, 1, 1, 0, "nchw"),    # head: N tail, fp32 NCHW output
    (2, 64, 9, 7, 64, 3, 1, 1, "relu"),        # ragged M, 3x3
    (2, 48, 16, 16, 48, 3, 1, 1, "relu"),
    (1, 192, 16, 16, 96, 1, 1, 0, "pro"),      # BatchNorm + ReLU operand prologue
    (3, 768, 4, 4, 3072, 1, 1, 0, "gelu"),     # M = 48
    (2, 256, 16, 16, 512, 2, 2, 0, "patch"),   # 2x2 / s2 patchify (downsample)
    (4, 1024, 64, 64, 256, 1, 1, 0, "res"),    # M = 16384, long K, N % 256 == 0: the 256 x 256-tile configuration (ConvNeXt-B pwconv2)
    (5, 2048, 60, 60, 512, 1, 1, 0, "res"),    # ... with a ragged last M tile (M = 18000)
    (16, 512, 32, 32, 2048, 1, 1, 0, "gelu"),  # M = 16384, ConvNeXt-B pwconv1 at stage 3
    # k x k convolutions large enough for the round-5 tile rules (kpf_conv2d_h16: >= 16384 pixels): 64 channels -> 128 x 64 tiles, N % 128 == 0 with a
    # round of 256 x 128 tiles, N % 256 == 0 with a round of 256 x 256 tiles; a ragged last row tile; the residual epilogue of a ResNet block
    (4, 64, 64, 64, 64, 3, 1, 1, "relu"),
    (16, 32, 64, 64, 128, 3, 1, 1, "relu"),
    (17, 32, 60, 65, 256, 3, 1, 1, "relu"),
    (16, 64, 64, 64, 256, 3, 1, 1, "res"),
    (16, 64, 128, 128, 256, 2, 2, 0, "patch"),
])
def test_conv2d_h16_is_the_fp32_conv_of_the_rounded_operands(case, prec):
    from keypointfusion_amd import lib as L
    from keypointfusion_amd.engine import Act, PackedConv
    from keypointfusion_amd.engine16 import DTYPES, Packed16, conv16, empty16
    dev = _dev()
    tdt, eps = PREC[prec]
    kdt = DTYPES[prec][1]
    B, Cin, H, W, N, k, stride, pad, kind = case
    g = torch.Generator().manual_seed(B * 1000 + Cin + N)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    xr, wr = x.to(tdt).float(), w.to(tdt).float()  # what the kernel sees
    pro = None
    if kind == "pro":
        ps, pt = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
        pro = (ps.double(), pt.double())
        xin = F.relu(xr * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1)).to(tdt).float()  # the prologue rounds back to storage precision
    else:
        xin = xr
    ref = F.conv2d(xin.double(), wr.double(), bias.double(), stride=stride, padding=pad)
    pc = PackedConv(wr, bias, dev, stride=stride, pad=pad, prologue=pro, patchify=(kind == "patch"))
    p16 = Packed16(pc, tdt)
    xa = Act(x.permute(0, 2, 3, 1).contiguous().to(tdt).view(-1).to(dev), B, H, W, Cin)
    OH, OW = ref.shape[-2:]
    flags, kw = 0, {}
    if kind == "gelu":
        flags, ref = L.KPF_ACT_GELU, F.gelu(ref)
    elif kind in ("relu", "pro"):
        flags, ref = L.KPF_ACT_RELU, F.relu(ref)
    elif kind == "res":
        r = torch.randn(B, N, OH, OW, generator=g)
        gam = torch.rand(N, generator=g)
        ra = Act(r.permute(0, 2, 3, 1).contiguous().to(tdt).view(-1).to(dev), B, OH, OW, N)
        kw = dict(res=ra, gamma=gam.to(dev))
        ref = ref * gam.view(1, -1, 1, 1).double() + r.to(tdt).double()
    if kind == "nchw":
        out = torch.empty(B, N, OH, OW, device=dev)
        conv16(p16, xa, kdt, out_nchw=out, flags=flags)
        got = out.cpu().double()
        tol = 2e-6  # fp32 output: only the accumulation order differs
    else:
        o = conv16(p16, xa, kdt, flags=flags, **kw)
        got = o.buf.view(B, OH, OW, N).permute(0, 3, 1, 2).float().cpu().double()
        tol = 1.01 * eps
    if kind == "gelu":
        # 16-bit outputs use the 9-operation GELU (csrc/kpf_conv.hip gelu_h16): |error| <= 2.6e-5 ABSOLUTE (what reaches pwconv2's sums), on top of
        # the output rounding — bounded as an absolute term, not against the 1e-3-of-range floor of the other epilogues
        bad = (got - ref).abs() - (1.3 * tol * ref.abs() + 3.5e-5)
        assert float(bad.max()) <= 0, (case, prec, float(bad.max()))
        return
    err = float(((got - ref).abs() / (ref.abs() + ref.abs().max() * 1e-3)).max()) if kind != "nchw" else rel(got, ref)
    assert err < 2.5 * tol + 1e-6, (case, prec, err)  # elementwise relative (floor at 1e-3 of the range): output rounding only


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(4096, 2048, 512), (300, 384, 96), (65536, 2048, 512)])
def test_gelu_of_the_16bit_path_at_large_activations(M, N, K, prec):
    """The 9-operation GELU of the 16-bit kernels (csrc/kpf_common.h kpf_gelu_h16) far outside the range its polynomial was fitted on.  Until round 5 the
    polynomial's argument was not clamped: beyond |x| = 11.1 its x^4 term flips the exponent's sign and GELU(12) came out as 0, GELU(-12) as -12 — silent
    garbage for the outlier activations a trained ConvNeXt's hidden layers do produce (synthetic weights never left |x| < 6, so no test saw it).  Pre-activations
    here have a standard deviation of ~10 (|x| up to ~50) and go through every tile family the dispatcher picks for these shapes (two-stage 128 x 128, the 32 x 64
    default, the eight-phase persistent kernel); same bound as the unit-range test: output rounding + 3.5e-5 absolute."""
    from keypointfusion_amd import lib as L
    from keypointfusion_amd.engine import Act, PackedConv
    from keypointfusion_amd.engine16 import DTYPES, Packed16, conv16
    dev = _dev()
    tdt, eps = PREC[prec]
    kdt = DTYPES[prec][1]
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 10.0).to(tdt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(tdt)
    bias = torch.randn(N, generator=g)
    pre = x.double() @ w.double().t() + bias.double()
    assert float(pre.abs().max()) > 30.0 and float((pre.abs() > 11.1).double().mean()) > 0.1  # the regime the old form got wrong
    ref = F.gelu(pre)
    o = conv16(Packed16(PackedConv(w.float().view(N, K, 1, 1), bias, dev), tdt), Act(x.view(-1).to(dev), 1, 1, M, K), kdt, flags=L.KPF_ACT_GELU)
    got = o.buf.view(M, N).float().cpu().double()
    assert bool(torch.isfinite(got).all())
    bad = (got - ref).abs() - (1.3 * 1.01 * eps * ref.abs() + 3.5e-5)
    assert float(bad.max()) <= 0, (prec, float(bad.max()), float(got[pre > 12].min()), float(got[pre < -12].abs().max()))
    assert float(got[pre < -12].abs().max()) < 1e-6 and float((got[pre > 12] / pre[pre > 12]).min()) > 0.99


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 16, 16, 96), (1, 32, 32, 128), (2, 16, 16, 192), (1, 8, 8, 384), (1, 4, 4, 768), (1, 16, 16, 256), (1, 8, 8, 1024)])
def test_dwconv7_ln_and_layernorm_h16(shape, prec):
    from keypointfusion_amd import lib as L
    from keypointfusion_amd.engine import _ptr, _stream
    from keypointfusion_amd.engine16 import DTYPES
    dev = _dev()
    lib = L.load()
    tdt, eps = PREC[prec]
    kdt = DTYPES[prec][1]
    B, H, W, Cc = shape
    g = torch.Generator().manual_seed(Cc + H)
    x = (torch.randn(B, H, W, Cc, generator=g) * 2 + 0.3).to(tdt)
    wdw, bdw = torch.randn(49, Cc, generator=g) / 7, torch.randn(Cc, generator=g) * 0.1
    lw, lb = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    d = [t.to(dev).contiguous() for t in (x, wdw, bdw, lw, lb)]
    y = torch.empty_like(d[0])
    L.check(lib.kpf_dwconv7_ln_h16(_ptr(d[0]), _ptr(d[1]), _ptr(d[2]), _ptr(d[3]), _ptr(d[4]), _ptr(y), B, H, W, Cc, 1e-6, kdt, _stream()), "dw16")
    xr = x.float().permute(0, 3, 1, 2).double()
    conv = F.conv2d(xr, wdw.t().reshape(Cc, 1, 7, 7).double(), bdw.double(), padding=3, groups=Cc).permute(0, 2, 3, 1)
    ref = F.layer_norm(conv, (Cc,), lw.double(), lb.double(), 1e-6)
    assert float((y.float().cpu().double() - ref).abs().max() / ref.abs().max()) < 1.5 * eps
    # LayerNorm alone: fp32 -> 16-bit (stem) and 16-bit -> 16-bit (downsample)
    x32 = torch.randn(B * H * W, Cc, generator=g) * 3 + 1
    for src, sk in ((x32, L.KPF_DT_F32), (x32.to(tdt), kdt)):
        z = torch.empty(B * H * W, Cc, device=dev, dtype=tdt)
        L.check(lib.kpf_layernorm_h16(_ptr(src.to(dev)), sk, _ptr(d[3]), _ptr(d[4]), _ptr(z), kdt, B * H * W, Cc, 1e-6, _stream()), "ln16")
        want = F.layer_norm(src.float().double(), (Cc,), lw.double(), lb.double(), 1e-6)
        assert float((z.float().cpu().double() - want).abs().max() / want.abs().max()) < 1.5 * eps


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (1, 32, 32, 128), (2, 20, 24, 192), (1, 16, 16, 512), (3, 40, 17, 256), (1, 64, 64, 128)])
def test_dwconv7_stats_and_ln_apply_h16(shape, prec):
    """kpf_dwconv7_stats_h16 + kpf_ln_apply_stats_h16 (LDS-tiled stencil with per-chunk LayerNorm statistics, then the normalisation from them)
    against float64 on the same rounded input (convNeXT/convnext.py:41-44): the un-normalised output up to its rounding, the chunk statistics
    against their definition, the normalised result within the two roundings of the pair; ragged tiles (H, W not multiples of 16) and a
    large common offset (|mean| >> sigma: the statistics must not cancel)."""
    from keypointfusion_amd import lib as L
    from keypointfusion_amd.engine import _ptr, _stream
    from keypointfusion_amd.engine16 import DTYPES
    dev = _dev()
    lib = L.load()
    tdt, ulp = PREC[prec]
    kdt = DTYPES[prec][1]
    B, H, W, Cc = shape
    assert lib.kpf_dwconv7_stats_supported(H, W, Cc)
    g = torch.Generator().manual_seed(Cc + H + W)
    x = (torch.randn(B, H, W, Cc, generator=g) * 2 + 0.3).to(tdt)
    wdw, bdw = torch.randn(49, Cc, generator=g) / 7, torch.randn(Cc, generator=g) * 0.1 + (5.0 if H == 20 else 0.0)  # (one case with a big offset)
    lw, lb = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    d = [t.to(dev).contiguous() for t in (x, wdw, bdw, lw, lb)]
    y = torch.full_like(d[0], float("nan"))
    st = torch.full((lib.kpf_dwconv7_stats_floats(B, H, W, Cc),), float("nan"), device=dev)
    L.check(lib.kpf_dwconv7_stats_h16(_ptr(d[0]), _ptr(d[1]), _ptr(d[2]), _ptr(y), _ptr(st), B, H, W, Cc, kdt, _stream()), "dwstats")
    xr = x.float().permute(0, 3, 1, 2).double()
    conv = F.conv2d(xr, wdw.t().reshape(Cc, 1, 7, 7).double(), bdw.double(), padding=3, groups=Cc).permute(0, 2, 3, 1)  # B H W C
    raw = y.float().cpu().double()
    assert bool(torch.isfinite(raw).all())
    assert float((raw - conv).abs().max() / conv.abs().max()) < 1.05 * ulp  # fp32 accumulation, one rounding (half an ulp is up to 2^-8 / 2^-11 relative)
    chunks = conv.reshape(B, H, W, Cc // 64, 64)
    sm = st.view(B, H, W, Cc // 64, 2).cpu().double()
    assert float((sm[..., 0] - chunks.mean(-1)).abs().max()) < 2e-5 * float(conv.abs().max())
    m2 = ((chunks - chunks.mean(-1, keepdim=True)) ** 2).sum(-1)
    assert float(((sm[..., 1] - m2).abs() / (m2 + 1e-3)).max()) < 1e-4
    L.check(lib.kpf_ln_apply_stats_h16(_ptr(y), _ptr(st), _ptr(d[3]), _ptr(d[4]), B * H * W, Cc, 1e-6, kdt, _stream()), "lnapply")
    ref = F.layer_norm(conv, (Cc,), lw.double(), lb.double(), 1e-6)
    got = y.float().cpu().double()
    # two roundings: the stored un-normalised value (relative ulp/2 of |conv|, amplified by |conv| / sigma on normalisation) and the result
    amp = float((conv.abs().amax(-1) / conv.std(-1)).max())
    assert float((got - ref).abs().max() / ref.abs().max()) < (1.0 + 0.5 * amp) * ulp, (float((got - ref).abs().max() / ref.abs().max()), amp)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("C,M", [(128, 1000), (128, 4096 + 48), (256, 777), (256, 2048)])
def test_fused_convnext_mlp_h16(C, M, prec):
    """kpf_convnext_mlp_h16 (out = x + gamma * (W2 GELU(W1 y + b1) + b2), convNeXT/convnext.py:44-51, hidden tensor in registers) against
    float64 on the SAME 16-bit-rounded operands with the exact erf GELU: what is lost is the rounding of the hidden activations to the
    storage type (they are GEMM2's operand), the 2.6e-5 of the 9-operation GELU and the output rounding.  Ragged M (last tile partial),
    in place (out aliases x) like the engine calls it, and equal to the two-GEMM path within the same tolerance."""
    import ctypes as C_
    from keypointfusion_amd import engine as E, lib as L
    from keypointfusion_amd.engine import MLP_HIDDEN_PERM
    from keypointfusion_amd.engine16 import DTYPES, Packed16, conv16
    dev = _dev()
    tdt, ulp = PREC[prec]
    kdt = DTYPES[prec][1]
    g = torch.Generator().manual_seed(C + M)
    y = torch.randn(M, C, generator=g).to(tdt)
    x = (torch.randn(M, C, generator=g) * 2).to(tdt)
    w1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(tdt)
    w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(tdt)
    b1, b2 = torch.randn(4 * C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    gamma = torch.rand(C, generator=g) + 0.5
    hid = y.double() @ w1.double().t() + b1.double()
    hid = 0.5 * hid * (1 + torch.erf(hid / 2 ** 0.5))
    ref = x.double() + gamma.double() * (hid @ w2.double().t() + b2.double())
    perm = torch.tensor(MLP_HIDDEN_PERM)
    w2c = w2.view(C, 4 * C // 32, 32)[:, :, perm].permute(1, 0, 2).contiguous().to(dev)
    yd, xd, w1d = y.to(dev), x.to(dev).clone(), w1.to(dev)
    b1d, b2d, gd = b1.to(dev), b2.to(dev), gamma.to(dev)
    P = lambda t: C_.c_void_p(t.data_ptr())
    L.check(L.load().kpf_convnext_mlp_h16(P(yd), P(xd), P(w1d), P(b1d), P(w2c), P(b2d), P(gd), P(xd), M, C, kdt, E._stream()), "kpf_convnext_mlp_h16")
    torch.cuda.synchronize()
    got = xd.double().cpu()
    assert bool(torch.isfinite(got).all())
    err = float(((got - ref).abs() / (ref.abs() + 1.0)).max())
    # hidden rounding: 4C products with relative error <= ulp/2 each, random signs -> ~ulp * |w2 row| * rms(hidden) ~ 1 ulp of an O(1) sum
    assert err < 3.0 * ulp + 1e-4, err
    # the unfused pair on the same operands (hidden tensor through HBM, fp32-accurate GELU): same numbers up to the same roundings
    pc1 = Packed16(E.PackedConv(w1.float(), b1, dev), tdt)
    pc2 = Packed16(E.PackedConv(w2.float(), b2, dev), tdt)
    ya = E.Act(yd.reshape(-1), 1, 1, M, C)
    ha = E.Act(torch.empty(M * 4 * C, device=dev, dtype=tdt), 1, 1, M, 4 * C)
    xa = E.Act(x.to(dev).reshape(-1).clone(), 1, 1, M, C)
    conv16(pc1, ya, kdt, out=ha, flags=L.KPF_ACT_GELU)
    conv16(pc2, ha, kdt, out=xa, gamma=gd, res=xa)
    torch.cuda.synchronize()
    two = xa.buf.view(M, C).double().cpu()
    assert float(((got - two).abs() / (ref.abs() + 1.0)).max()) < 4.0 * ulp + 1e-4


def _model(net, prec):
    from keypointfusion_amd.model.model import KPFusion
    m = KPFusion("KPFusion-" + net, "", 21, "dexycb", "")
    m.load_state_dict(synthetic_sd("KPFusion-" + net), strict=True)
    m.precision = prec
    return m.to(_dev()).eval()


@pytest.mark.parametrize("net,prec,mean_tol,final_tol,map_tol", [("convnext-tiny", "bf16", 4.0, 2.0, 6e-2), ("convnext-tiny", "f16", 1.5, 0.7, 8e-3),
                                                                  ("resnet-18", "bf16", 6.0, 3.0, 6e-2), ("resnet-18", "f16", 1.5, 0.8, 8e-3)])
def test_full_model_reduced_precision_tolerance_in_mm(net, prec, mean_tol, final_tol, map_tol):
    """configs[2]: full model (ConvNeXt-T and ResNet-18, 128x128), 16-bit backbones + fp32 head, against the fp32 oracle: the tolerance study."""
    from oracle import kpf_oracle as O
    dev = _dev()
    sd = synthetic_sd("KPFusion-" + net)
    B = 4
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, 128, seed=3).items()}
    ref, rsw = O.kpfusion_forward(sd, b["img_rgb"], b["img"], b["pcl"], b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
    m = _model(net, prec)

    class Loader:
        img_size, flip = 128, 1

    with torch.no_grad():
        res, sws, _ = m(b["img_rgb"].to(dev), b["img"].to(dev), b["pcl"].to(dev), Loader(), b["center"].to(dev), b["M"].to(dev),
                        b["cube"].to(dev), b["cam_para"].to(dev), 0.8)
    assert all(t.dtype == torch.float32 and bool(torch.isfinite(t).all()) for t in res + sws)
    for k in (0, 1):
        assert rel(res[k], ref[k]) < map_tol, (prec, k, rel(res[k], ref[k]))
    dev_mm = [(res[k].cpu() - ref[k]).norm(dim=-1).reshape(-1) * 125.0 for k in range(2, 6)]  # cube 250 mm: x * 125 mm; B * 21 joints per stage
    mm = [float(d.max()) for d in dev_mm]
    mean_mm = [float(d.mean()) for d in dev_mm]
    med_mm = [float(d.median()) for d in dev_mm]
    p90_mm = [float(d.kthvalue(int(0.9 * d.numel()))[0]) for d in dev_mm]
    jumps = [int((d > 5.0).sum()) for d in dev_mm]
    print("reduced precision %s %s per stage: mean %s  median %s  p90 %s  max %s mm; joints > 5 mm: %s of %d" % (
        net, prec, ["%.3f" % v for v in mean_mm], ["%.3f" % v for v in med_mm], ["%.3f" % v for v in p90_mm], ["%.2f" % v for v in mm], jumps, dev_mm[0].numel()))
    assert max(mean_mm) < mean_tol and mean_mm[3] < final_tol, (prec, mean_mm, mm)
    # Flip-aware bounds instead of a loose maximum (tools/bf16_sensitivity.py): the deviation has a continuous part — bounded through the
    # median — and JUMPS of single joints by 5-25 mm when a ball-query neighbourhood or a top-4 pixel set changes by one point (discrete decisions
    # taken around network outputs; with untrained weights the intermediate 3-D estimate of block 2, which is decoded around block 1's refined
    # joints, jumps for 4-12 % of the joints, whichever layers are rounded).  Jumps are counted, not averaged away; the FINAL estimate (what the
    # accuracy metric of BASELINE.json is computed on) must not jump at all.
    med_tol, fin_p90 = (1.5, 2.5) if prec == "bf16" else (0.7, 2.0)
    assert max(med_mm) < med_tol, (prec, med_mm)
    assert max(jumps) <= int(0.15 * dev_mm[0].numel()), (prec, jumps)
    assert jumps[3] == 0 and mm[3] < 5.0 and p90_mm[3] < fin_p90, (prec, jumps, mm, p90_mm)


@pytest.mark.parametrize("net,prec,lim", [("convnext-tiny", "bf16", dict(mean=1.3, median=0.9, p90=2.8, vmax=8.0, jump_mid=0.10, jump_2d=0.005)),
                                          ("convnext-tiny", "f16", dict(mean=0.4, median=0.2, p90=0.9, vmax=5.0, jump_mid=0.02, jump_2d=0.005)),
                                          # ResNet-18 (round 6, measured on these 64 crops: bf16 final mean 1.00 / median 0.82 / p90 1.88 / max 4.2 mm, intermediate 3-D estimates
                                          # jump for 2.6 % / 13.4 % of the joints and block 1's refined estimate for 1.0 % — BatchNorm backbones round harder in bf16 than the
                                          # LayerNorm ones; the FINAL estimate does not jump)
                                          ("resnet-18", "bf16", dict(mean=1.5, median=1.2, p90=2.8, vmax=8.0, jump_mid=0.20, jump_2d=0.02)),
                                          ("resnet-18", "f16", dict(mean=0.5, median=0.3, p90=1.2, vmax=6.0, jump_mid=0.04, jump_2d=0.005))])
def test_reduced_precision_deviation_statistics_over_many_batches(net, prec, lim):
    """The 16-bit accuracy claim with statistics under it: 8 input seeds x B = 8 = 1344 joints per stage against the fp32 DEVICE forward
    (which the parity tests pin to the oracle within 1e-4 mm), instead of one batch of four.  Bounds from the distribution measured over 16 x 8 crops
    (tools/precision_stats.py -> profiles/r05_precision_stats.txt): final estimate bf16 mean 0.86 +- 0.32 (worst batch 1.69), median 0.58, p90 1.84,
    max 4.2 mm, no joint beyond 5 mm; f16 mean 0.22 +- 0.12, median 0.10, p90 0.50, max 2.2 mm.  The intermediate 3-D estimates jump (> 5 mm: a ball-query
    neighbourhood or top-4 pixel set changed by one point) for 5.2 % (bf16) / 0.7 % (f16) of the joints: counted, and bounded, not averaged away."""
    import numpy as np
    dev = _dev()
    m32, m16 = _model(net, "f32"), _model(net, prec)

    class Loader:
        img_size, flip = 128, 1

    devs = [[] for _ in range(4)]
    for seed in range(21, 29):
        b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(8, 128, seed=seed).items()}
        with torch.no_grad():
            r32 = m32(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)[0]
            r16 = m16(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)[0]
        half = b["cube"].view(8, 1, 3) / 2
        for st in range(4):
            devs[st].append(((r16[2 + st] - r32[2 + st]) * half).norm(dim=-1).reshape(-1).cpu().numpy())
    v = [np.concatenate(d) for d in devs]
    fin = v[3]
    stats = dict(mean=float(fin.mean()), median=float(np.median(fin)), p90=float(np.percentile(fin, 90)), vmax=float(fin.max()))
    jumps = [float((x > 5.0).mean()) for x in v]
    print("reduced precision %s %s over %d joints per stage: final %s; joints > 5 mm per stage %s" % (net, prec, fin.size, {k: round(x, 3) for k, x in stats.items()}, ["%.3f" % j for j in jumps]))
    for k in ("mean", "median", "p90", "vmax"):
        assert stats[k] < lim[k], (prec, k, stats)
    assert jumps[3] <= 0.005 and jumps[1] <= lim["jump_2d"], (prec, jumps)    # the final estimate (what the metric reads) does not jump; block 1's refined one rarely
    assert max(jumps[0], jumps[2]) <= lim["jump_mid"], (prec, jumps)          # the intermediate 3-D estimates do, rarely
    # f16 is the recommended reduced-precision mode: its final estimate is at least twice as close as bf16's on the same crops (measured: 4x)
    if prec == "f16":
        assert stats["mean"] < 0.5


@pytest.mark.parametrize("net", ["convnext-tiny", "resnet-18"])
def test_reduced_precision_deviation_on_the_reference_demo_crop(net):
    """The one REAL crop this repo can hold (BASELINE configs[0]: visualization/box* cropped as demo_RGBD.py does; tests/golden/demo_box_*): the 16-bit
    forwards against the fp32 device forward on it, in mm, per stage — a real hand's depth statistics instead of the synthetic disc.
    Weights are still synthetic (no checkpoint exists here), so the bound is the many-batch distribution's, not an accuracy claim."""
    import numpy as np
    from test_preprocess import P, _frame
    dev = _dev()
    rgb, depth, bbox, cam = _frame()
    pre = P.prepare_rgbd(rgb, depth, bbox, cam)
    b = {k: torch.from_numpy(np.ascontiguousarray(pre[k]))[None].to(dev) for k in ("img_rgb", "img", "pcl", "center", "M", "cube", "cam_para")}

    class Loader:
        img_size, flip = 128, 1

    out = {}
    for prec in ("f32", "bf16", "f16"):
        with torch.no_grad():
            out[prec] = _model(net, prec)(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)[0]
    half = b["cube"].view(1, 1, 3) / 2
    rep = {}
    for prec in ("bf16", "f16"):
        d = [((out[prec][2 + st] - out["f32"][2 + st]) * half).norm(dim=-1).reshape(-1).cpu().numpy() for st in range(4)]
        rep[prec] = d
        print("demo crop %s %s vs fp32: per stage mean %s max %s mm" % (net, prec, ["%.3f" % x.mean() for x in d], ["%.3f" % x.max() for x in d]))
        assert all(np.isfinite(x).all() for x in d)
    # final estimate (stage 4, what the metric reads): inside the worst single batch of the synthetic distribution (bf16 1.69 mm mean / f16 0.5), no jump
    assert rep["bf16"][3].mean() < 3.0 and rep["bf16"][3].max() < 8.0, rep["bf16"][3]
    assert rep["f16"][3].mean() < 0.8 and rep["f16"][3].max() < 5.0, rep["f16"][3]
    assert rep["f16"][3].mean() <= rep["bf16"][3].mean() + 0.05, "f16 is the recommended reduced-precision mode"


def test_full_model_bf16_at_the_stated_batch_of_configs2():
    """configs[2] at its stated batch: B = 32 full model in bf16.  Tile selection depends on M = B H W, so the B = 32 launches are not the B = 4
    ones: every sample of the B = 32 forward must still be bit-identical to the same sample inside a B = 4 forward (eight groups of four),
    all outputs finite, two runs identical."""
    dev = _dev()
    m = _model("convnext-tiny", "bf16")
    b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(32, 128, seed=11).items()}

    class Loader:
        img_size, flip = 128, 1

    def fwd(sl):
        with torch.no_grad():
            res, sws, _ = m(b["img_rgb"][sl], b["img"][sl], b["pcl"][sl], Loader(), b["center"][sl], b["M"][sl], b["cube"][sl], b["cam_para"][sl], 0.8)
        return [t.clone() for t in res + sws]

    big = fwd(slice(0, 32))
    assert all(t.shape[0] == 32 and bool(torch.isfinite(t).all()) for t in big)
    again = fwd(slice(0, 32))
    assert all(torch.equal(a, c) for a, c in zip(big, again)), "two B = 32 forwards differ"
    for g in range(8):
        small = fwd(slice(4 * g, 4 * g + 4))
        for i, (a, c) in enumerate(zip(big, small)):
            assert torch.equal(a[4 * g:4 * g + 4], c), "output %d: samples %d..%d of the B = 32 forward differ from their B = 4 forward" % (i, 4 * g, 4 * g + 3)


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_convnext_base_512_backbones_properties_and_oracle_subset(prec):
    """configs[4]: ConvNeXt-B backbones at 512x512.  Full-size properties (finite, deterministic, independent of batch composition and
    position) at B = 8, and the fp32 oracle on one sample."""
    from oracle import kpf_oracle as O
    dev = _dev()
    net = "convnext-base"
    sd = synthetic_sd("KPFusion-" + net)
    B, S = 8, 512
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, S, seed=2).items()}
    m = _model(net, prec)
    with torch.no_grad():
        o1 = [t.clone() for t in m.forward_backbones(b["img_rgb"].to(dev), b["img"].to(dev))]
        o2 = m.forward_backbones(b["img_rgb"].to(dev), b["img"].to(dev))
        perm = torch.tensor([3, 0, 7, 1, 5, 2, 6, 4])
        o3 = m.forward_backbones(b["img_rgb"][perm].to(dev), b["img"][perm].to(dev))
        o4 = m.forward_backbones(b["img_rgb"][2:3].to(dev), b["img"][2:3].to(dev))
    assert [tuple(t.shape) for t in o1] == [(B, 105, 128, 128), (B, 128, 128, 128)] * 2
    for a, c, p, s1 in zip(o1, o2, o3, o4):
        assert bool(torch.isfinite(a).all())
        assert torch.equal(a, c), "not deterministic"
        assert torch.equal(a[perm], p), "a sample's result depends on its batch position / neighbours"
        assert torch.equal(a[2:3], s1), "a sample's result depends on the batch size"
    if prec == "f16":
        # the bench's own batch size (configs[4], B = 64): tile configurations are chosen from M = B*H*W, so the launches differ from the
        # B = 8 ones above — every sample must still come out bit-identical to its B = 8 result
        with torch.no_grad():
            o64 = m.forward_backbones(b["img_rgb"].repeat(8, 1, 1, 1).to(dev), b["img"].repeat(8, 1, 1, 1).to(dev))
        for a, big in zip(o1, o64):
            assert big.shape[0] == 64 and bool(torch.isfinite(big).all())
            for k in range(8):
                assert torch.equal(big[8 * k:8 * k + 8], a), "B = 64 launches compute different values than B = 8 (copy %d)" % k
        del o64
    ref = O.backbones_forward(sd, b["img_rgb"][2:3], b["img"][2:3])
    tol = 6e-2 if prec == "bf16" else 8e-3
    for a, r, name in zip(o1, ref, ("img_offset", "img_feat", "img_offset_rgb", "img_feat_rgb")):
        e = rel(a[2:3], r)
        print("convnext-base 512 %s %s: rel err vs fp32 oracle %.2e" % (prec, name, e))
        assert e < tol, (name, e)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("C_,M", [(512, 8192), (512, 16384), (512, 8640), (1024, 4096), (512, 1024), (384, 16384)])
def test_layernorm_folded_into_pwconv1_h16(C_, M, prec):
    """KPF_PRO_LN (convNeXT/convnext.py:42-44: pwconv1(norm(x)) then GELU): the GEMM multiplies the RAW 16-bit tensor by W diag(ln_w) and its epilogue applies
    rstd * (acc - mean * s) + (W ln_b + b) — against float64 (a) of the same expression on the same rounded operands (the kernel's arithmetic: output rounding
    only) and (b) of LayerNorm -> Linear -> GELU itself on the same raw tensor (what the fold may lose: the rounding of W diag(ln_w) to the storage type,
    a relative 2^-9 / 2^-12 per weight, where the two-pass form rounds the normalised activations instead).  kpf_ln_stats_merge against its definition.
    One-tile-per-workgroup form (256 tiles), persistent form (512 tiles), ragged last row tile, the 1024-channel stage, and fewer tiles than CUs (the
    fold is a rule over the layer's shape, so small batches take the same kernel)."""
    from keypointfusion_amd import lib as L
    from keypointfusion_amd.engine import Act, PackedConv, _ptr, _stream
    from keypointfusion_amd.engine16 import DTYPES, Packed16, conv16
    dev = _dev()
    lib = L.load()
    tdt, ulp = PREC[prec]
    kdt = DTYPES[prec][1]
    g = torch.Generator().manual_seed(C_ + M)
    N = 4 * C_
    x = (torch.randn(M, C_, generator=g) * (torch.rand(M, 1, generator=g) * 2 + 0.5) + torch.randn(M, 1, generator=g)).to(tdt)  # rows of different scale and offset
    w1 = torch.randn(N, C_, generator=g) / C_ ** 0.5
    b1 = torch.randn(N, generator=g) * 0.3
    lw, lb = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.2
    xd = x.double().to(dev)  # (the float64 references run on the GPU: 17 GFLOP each at the largest case)
    w1, b1, lw, lb = w1.to(dev), b1.to(dev), lw.to(dev), lb.to(dev)
    # chunk statistics as kpf_dwconv7_stats_h16 defines them: (mean, sum of centred squares) of every 64-channel chunk
    ch = xd.view(M, C_ // 64, 64)
    st = torch.stack((ch.mean(-1), ((ch - ch.mean(-1, keepdim=True)) ** 2).sum(-1)), -1).float().contiguous()
    mr = torch.full((M, 2), float("nan"), device=dev)
    L.check(lib.kpf_ln_stats_merge(_ptr(st), _ptr(mr), M, C_, 1e-6, _stream()), "merge")
    mean, rstd = xd.mean(-1), 1.0 / torch.sqrt(xd.var(-1, unbiased=False) + 1e-6)
    assert float((mr[:, 0].double() - mean).abs().max()) < 1e-5 * float(xd.abs().max())
    assert float(((mr[:, 1].double() - rstd).abs() / rstd).max()) < 1e-5
    p16 = Packed16(PackedConv((w1.double() * lw.double()[None, :]).float().cpu().reshape(N, C_, 1, 1), None, dev), tdt)
    s = p16.w[:, :C_].double().sum(1).float().contiguous()
    bf = (w1.double() @ lb.double() + b1.double()).float()
    xa = Act(x.to(dev).contiguous().view(-1), 1, 1, M, C_)
    out = Act(torch.full((M * N,), float("nan"), device=dev, dtype=tdt), 1, 1, M, N)
    assert conv16(p16, xa, kdt, out=out, flags=L.KPF_ACT_GELU, probe=True)
    conv16(p16, xa, kdt, out=out, flags=L.KPF_ACT_GELU, ln=(mr, s, bf))
    got = out.buf.view(M, N).double()
    assert bool(torch.isfinite(got).all())
    gelu = lambda v: 0.5 * v * (1 + torch.erf(v / 2 ** 0.5))
    mrd = mr.double()
    same = gelu(mrd[:, 1:2] * (xd @ p16.w[:, :C_].double().t() - mrd[:, 0:1] * s.double()[None, :]) + bf.double()[None, :])
    scale = float(same.abs().max())
    assert float((got - same).abs().max()) / scale < 1.2 * ulp, float((got - same).abs().max()) / scale  # fp32 accumulation and epilogue, one output rounding
    true = gelu(F.layer_norm(xd, (C_,), lw.double(), lb.double(), 1e-6) @ w1.double().t() + b1.double())
    assert float((got - true).abs().max()) / scale < 3 * ulp, float((got - true).abs().max()) / scale
    # without the statistics the call is refused, as is a layer the eight-phase kernel does not cover
    d_bad = Packed16(PackedConv(torch.randn(N + 64, C_, 1, 1, generator=g), None, dev), tdt)
    assert not conv16(d_bad, xa, kdt, flags=L.KPF_ACT_GELU, probe=True)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_convnext_block16_with_and_without_the_folded_layernorm(prec):
    """engine16.Block16 at a ConvNeXt-B stage-3 shape (C = 512, 32 x 32 maps): the schedule with the LayerNorm folded into pwconv1 (stencil + statistics,
    merge, GEMM with KPF_PRO_LN, pwconv2) against the three-pass one (stencil + statistics, normalisation pass, GEMM) and against float64 — the folded form
    must not be further from float64 than the three-pass form by more than an output rounding."""
    from keypointfusion_amd import engine16 as E16, lib as L
    from keypointfusion_amd.engine import Act
    dev = _dev()
    lib = L.load()
    tdt, ulp = PREC[prec]
    kdt = E16.DTYPES[prec][1]
    B, H, W, Cc = 8, 32, 32, 512
    g = torch.Generator().manual_seed(7)
    sd = {"b.dwconv.weight": torch.randn(Cc, 1, 7, 7, generator=g) / 7, "b.dwconv.bias": torch.randn(Cc, generator=g) * 0.1,
          "b.norm.weight": torch.rand(Cc, generator=g) + 0.5, "b.norm.bias": torch.randn(Cc, generator=g) * 0.1,
          "b.pwconv1.weight": torch.randn(4 * Cc, Cc, generator=g) / Cc ** 0.5, "b.pwconv1.bias": torch.randn(4 * Cc, generator=g) * 0.2,
          "b.pwconv2.weight": torch.randn(Cc, 4 * Cc, generator=g) / (4 * Cc) ** 0.5, "b.pwconv2.bias": torch.randn(Cc, generator=g) * 0.2,
          "b.gamma": torch.rand(Cc, generator=g) + 0.5}
    blk = E16.Block16(sd, "b", dev, tdt)
    assert not blk.fused
    x0 = (torch.randn(B, H, W, Cc, generator=g) * 1.5).to(tdt)
    outs = {}
    keep = E16.LN_FOLD
    try:
        for fold in (True, False):
            E16.LN_FOLD = fold
            x = Act(x0.to(dev).contiguous().view(-1), B, H, W, Cc)
            y, h = E16.empty16(B, H, W, Cc, dev, tdt), E16.empty16(B, H, W, 4 * Cc, dev, tdt)
            st = torch.empty(lib.kpf_dwconv7_stats_floats(B, H, W, Cc), device=dev, dtype=torch.float32)
            blk(x, y, h, kdt, st)
            outs[fold] = x.buf.view(B, H, W, Cc).float().cpu().double()
    finally:
        E16.LN_FOLD = keep
    xd = x0.double().permute(0, 3, 1, 2)
    t = F.conv2d(xd, sd["b.dwconv.weight"].double(), sd["b.dwconv.bias"].double(), padding=3, groups=Cc).permute(0, 2, 3, 1)
    t = F.layer_norm(t, (Cc,), sd["b.norm.weight"].double(), sd["b.norm.bias"].double(), 1e-6)
    t = t @ sd["b.pwconv1.weight"].double().t() + sd["b.pwconv1.bias"].double()
    t = 0.5 * t * (1 + torch.erf(t / 2 ** 0.5))
    ref = x0.double() + sd["b.gamma"].double() * (t @ sd["b.pwconv2.weight"].double().t() + sd["b.pwconv2.bias"].double())
    scale = float(ref.abs().max())
    e_fold, e_three = float((outs[True] - ref).abs().max()) / scale, float((outs[False] - ref).abs().max()) / scale
    assert e_fold < e_three + ulp and e_fold < 6 * ulp, (e_fold, e_three)
    assert float((outs[True] - outs[False]).abs().max()) / scale < 6 * ulp


@pytest.mark.parametrize("prec,map_tol", [("f16", 8e-3), ("bf16", 6e-2)])
def test_convnext_tiny_backbones_at_256_take_the_folded_layernorm(prec, map_tol):
    """At 256 x 256 crops ConvNeXt-T's stage 3 (16 x 16 maps, C = 384) runs stencil + statistics, and its pwconv1 takes the LayerNorm folded into the GEMM
    (KPF_PRO_LN) — the same model with the fold switched off (three passes) and the fp32 plan bound it: dense maps and features of both backbones within the
    16-bit tolerance of the fp32 ones, and the folded and the three-pass results within that tolerance of each other."""
    from keypointfusion_amd import engine16 as E16, lib as L
    from keypointfusion_amd.engine import ModelPlan
    dev = _dev()
    net = "KPFusion-convnext-tiny"
    sd = synthetic_sd(net)
    g = torch.Generator().manual_seed(5)
    img, rgb = torch.rand(2, 1, 256, 256, generator=g).to(dev) * 2 - 1, torch.rand(2, 3, 256, 256, generator=g).to(dev)
    assert E16._ln_fold_shape_ok(384) and E16._ln_fold_shape_ok(768) and not E16._ln_fold_shape_ok(96) and not E16._ln_fold_shape_ok(192)
    assert L.load().kpf_dwconv7_stats_supported(16, 16, 384)
    outs = {}
    keep = E16.LN_FOLD
    try:
        with torch.no_grad():
            for mode in ("f32", "fold", "three"):
                E16.LN_FOLD = mode == "fold"
                plan = ModelPlan(sd, net, dev, precision="f32" if mode == "f32" else prec)
                (od, fd), (orgb, frgb) = plan.backbones(img, rgb)
                outs[mode] = [t.float().cpu() for t in (od, fd.buf.view(-1), orgb, frgb.buf.view(-1))]
    finally:
        E16.LN_FOLD = keep
    for a, b, r in zip(outs["fold"], outs["three"], outs["f32"]):
        assert bool(torch.isfinite(a).all())
        assert rel(a, r) < map_tol and rel(b, r) < map_tol, (rel(a, r), rel(b, r))
        assert rel(a, b) < map_tol
    assert any(not torch.equal(a, b) for a, b in zip(outs["fold"], outs["three"]))  # (the switch did select two schedules)
