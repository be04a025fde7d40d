"""GPU box: correctness of kpf_conv2d_h16's dense 1x1 path against an fp64 reference on the rounded operands, then timings of the heavy ConvNeXt-B
512^2 shapes (configs[4]).  Run once per tile configuration:  KPF_FORCE_CFG16=<case> python tools/gemm16_check.py [--bench-only|--check-only]
(no KPF_FORCE_CFG16 = the dispatcher's own choice: the eight-phase 256 x 256 kernel wherever it applies)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from keypointfusion_amd import engine as E, lib as L  # noqa: E402
from keypointfusion_amd.engine16 import DTYPES, Packed16, conv16  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def gelu64(x):
    return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def check(prec, M, N, K, kind, in_place=False):
    tdt, kdt = DTYPES[prec]
    xh = (torch.randn(M, K, generator=g) * 1.5).to(tdt)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    x = E.Act(xh.reshape(-1).to(dev), 1, 1, M, K)
    p16 = Packed16(E.PackedConv(w, b, dev), tdt)
    wr = p16.w[:, :K].double().cpu()
    ref = xh.double() @ wr.t() + b.double()
    gam = None
    if kind == "gelu":
        ref, fl, res = gelu64(ref), L.KPF_ACT_GELU, None
    elif kind == "relu":
        ref, fl, res = ref.clamp_min(0), L.KPF_ACT_RELU, None
    elif kind == "lin":
        fl, res = 0, None
    else:
        rh = torch.randn(M, N, generator=g).to(tdt)
        gam = (torch.rand(N, generator=g) + 0.5).to(dev)
        ref = ref * gam.double().cpu() + rh.double()
        fl = 0
        res = E.Act(rh.reshape(-1).to(dev), 1, 1, M, N)
    out = res if (in_place and res is not None) else E.Act(torch.full((M * N,), float("nan"), device=dev, dtype=tdt), 1, 1, M, N)
    conv16(p16, x, kdt, out=out, flags=fl, gamma=gam, res=res)
    torch.cuda.synchronize()
    got = out.buf.view(M, N).double().cpu()
    ulp = 2.0 ** (-8 if prec == "bf16" else -11)
    err = ((got - ref).abs() / (ref.abs() + 1.0)).max().item()
    ok = err < 1.5 * ulp + (6e-5 if kind == "gelu" else 0) and bool(torch.isfinite(got).all())
    print("%-4s M=%-6d N=%-5d K=%-5d %-5s%s max err / (|ref|+1) = %.2e  %s" % (prec, M, N, K, kind, " in-place" if in_place else "", err, "ok" if ok else "FAIL"), flush=True)
    return ok


def bench(prec="f16"):
    tdt, kdt = DTYPES[prec]
    shapes = [(65536, 2048, 512, "gelu"), (65536, 512, 2048, "res"), (16384, 4096, 1024, "gelu"), (16384, 1024, 4096, "res"), (262144, 1024, 256, "gelu"),
              (262144, 256, 1024, "res"), (1048576, 512, 128, "gelu"), (1048576, 128, 512, "res"), (8192, 8192, 8192, "lin")]
    for M, N, K, kind in shapes:
        x = E.Act((torch.randn(M * K, generator=g)).to(tdt).to(dev), 1, 1, M, K)
        p16 = Packed16(E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev), tdt)
        out = E.Act(torch.empty(M * N, device=dev, dtype=tdt), 1, 1, M, N)
        res = E.Act(torch.randn(M * N, generator=g).to(tdt).to(dev), 1, 1, M, N) if kind == "res" else None
        fl = L.KPF_ACT_GELU if kind == "gelu" else 0
        for _ in range(3):
            conv16(p16, x, kdt, out=out, flags=fl, res=res)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                conv16(p16, x, kdt, out=out, flags=fl, res=res)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        byt = 2.0 * (M * K + N * K + M * N * (2 if res is not None else 1))
        print("%s M=%-8d N=%-5d K=%-5d %-4s %.3f ms  %.0f TF  %.2f TB/s" % (prec, M, N, K, kind, best, 2.0 * M * N * K / best / 1e9, byt / best / 1e9), flush=True)


if __name__ == "__main__":
    L.load()
    print("KPF_FORCE_CFG16 =", os.environ.get("KPF_FORCE_CFG16"))
    good = True
    if "--bench-only" not in sys.argv:
        for prec in ("f16", "bf16"):
            for M, N, K, kind, ip in [(512, 256, 128, "lin", False), (1000, 256, 128, "gelu", False), (256, 512, 256, "relu", False), (2048, 256, 1024, "res", True),
                                      (777, 512, 384, "res", False), (65536, 512, 512, "gelu", False), (4096, 1024, 2048, "res", True)]:
                good &= check(prec, M, N, K, kind, ip)
        print("ALL OK" if good else "SOME FAILED")
    if "--check-only" not in sys.argv:
        bench(os.environ.get("KPF_PREC", "f16"))
    sys.exit(0 if good else 1)
