"""GPU box: gemm16_dfe_kernel (256 x 128 tiles, deferred epilogue) against the library's default choice for the same launch — bit-for-bit — and timed
beside it (graph of 10 launches).  usage: python tools/dfe_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import engine as E, engine16 as E16, lib as L
from keypointfusion_amd.engine16 import DTYPES, Packed16, conv16
dev = torch.device("cuda:0")


def timed(fn):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3


ONLY_FIRST = len(sys.argv) > 1  # ablation runs (KPF_G8_DBG=...): the K = 512 GELU layer in f16 only
for prec in (("f16",) if ONLY_FIRST else ("f16", "bf16")):
    tdt, kdt = DTYPES[prec]
    for M, N, K, kind in [(65536, 2048, 512, "gelu")] if ONLY_FIRST else [(65536, 2048, 512, "gelu"), (16384, 4096, 1024, "gelu"), (65536, 1024, 512, "relu"), (32768, 512, 1024, "lin"), (65536, 512, 512, "slices"),
                          (131072, 1024, 2048, "gelu")]:
        g = torch.Generator().manual_seed(M + N + K)
        if kind == "slices":
            x = E.Act(torch.randn(M * (K + 64), generator=g).to(tdt).to(dev), 1, 1, M, K, ld=K + 64, coff=64)
        else:
            x = E.Act(torch.randn(M * K, generator=g).to(tdt).to(dev), 1, 1, M, K)
        p16 = Packed16(E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev), tdt)
        fl = L.KPF_ACT_GELU if kind == "gelu" else (L.KPF_ACT_RELU if kind == "relu" else 0)
        outs = []
        times = []
        for cfg in (0, 51):
            E16.FORCE_TILE16 = cfg
            if kind == "slices":
                o = E.Act(torch.zeros(M * (N + 128), device=dev, dtype=tdt), 1, 1, M, N, ld=N + 128, coff=128)
            else:
                o = E.Act(torch.zeros(M * N, device=dev, dtype=tdt), 1, 1, M, N)
            times.append(timed(lambda: conv16(p16, x, kdt, out=o, flags=fl)))
            outs.append(o.buf.view(torch.int16).clone())
        E16.FORCE_TILE16 = 0
        d = outs[0] != outs[1]
        fin = bool(torch.isfinite(outs[1].view(tdt).float()).all())
        print("%-5s M=%-7d N=%-5d K=%-5d %-6s default %7.1f us  dfe %7.1f us  (%.0f -> %.0f TF)  differing elements %d of %d  finite %s" % (
            prec, M, N, K, kind, times[0], times[1], 2.0 * M * N * K / times[0] / 1e6, 2.0 * M * N * K / times[1] / 1e6, int(d.sum()), d.numel(), fin), flush=True)
