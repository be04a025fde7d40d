"""GPU box: are gemm16_8ph_kernel's results bit-identical to the round-3 tile shapes'?  (a sample's result must not depend on which kernel its batch
size selects.)  Runs itself twice (KPF_NO_8PH=1 / default) and compares the saved outputs."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from keypointfusion_amd import engine as E, lib as L
    from keypointfusion_amd.engine16 import DTYPES, Packed16, conv16
    dev = torch.device("cuda:0")
    out = {}
    for prec in ("f16", "bf16"):
        tdt, kdt = DTYPES[prec]
        for M, N, K, kind in [(65536, 512, 2048, "res"), (65536, 2048, 512, "gelu"), (65536, 256, 256, "lin"), (65536, 256, 128, "res0"),
                              (65536, 256, 128, "relu"), (65536, 256, 384, "resself"), (65536, 256, 128, "slices")]:
            g = torch.Generator().manual_seed(M + N + K)
            if kind == "slices":  # input = channel slice of a wider row, output = channel slice of a wider row (UNet concat buffers)
                xb = torch.randn(M * (K + 64), generator=g).to(tdt).to(dev)
                x = E.Act(xb, 1, 1, M, K, ld=K + 64, coff=64)
            else:
                x = E.Act(torch.randn(M * K, generator=g).to(tdt).to(dev), 1, 1, M, K)
            p16 = Packed16(E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev), tdt)
            res = E.Act(torch.randn(M * N, generator=g).to(tdt).to(dev), 1, 1, M, N) if kind in ("res", "res0", "resself") else None
            gam = (torch.rand(N, generator=g) + 0.5).to(dev) if kind == "res" else None
            outa = None
            if kind == "resself":
                outa = res  # in place (Residual16: conv3 adds into the buffer the skip convolution wrote)
            if kind == "slices":
                outa = E.Act(torch.zeros(M * (N + 128), device=dev, dtype=tdt), 1, 1, M, N, ld=N + 128, coff=128)
            fl = L.KPF_ACT_GELU if kind == "gelu" else (L.KPF_ACT_RELU if kind == "relu" else 0)
            o = conv16(p16, x, kdt, out=outa, flags=fl, res=res, gamma=gam)
            torch.cuda.synchronize()
            out["%s_%s" % (prec, kind)] = o.buf.view(torch.int16).cpu()
    torch.save(out, sys.argv[1])
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "/tmp/g8_new.pt"], env=env)
    env["KPF_NO_8PH"] = "1"
    subprocess.check_call([sys.executable, __file__, "/tmp/g8_old.pt"], env=env)
    import torch
    a, b = torch.load("/tmp/g8_new.pt"), torch.load("/tmp/g8_old.pt")
    for k in a:
        d = (a[k] != b[k])
        print(k, "differing elements:", int(d.sum()), "of", d.numel())
