"""Tuning aid (GPU box): kpf_conv2d_h16 on the heavy 1x1 shapes of ConvNeXt-B at 512^2 (configs[4]).  KPF_FORCE_CFG16=<case> forces a tile
configuration (0 1 2 5 6 8 = two-stage shapes, 20-22 = LDS-ring variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import engine as E, lib as L
from keypointfusion_amd.engine16 import DTYPES, Packed16, conv16
dev = torch.device("cuda:0")
tdt, kdt = DTYPES[os.environ.get("KPF_PREC", "f16")]
g = torch.Generator().manual_seed(0)
SHAPES = [(65536, 2048, 512, "gelu"), (65536, 512, 2048, "res"), (16384, 4096, 1024, "gelu"), (16384, 1024, 4096, "res"), (262144, 1024, 256, "gelu"),
          (262144, 256, 1024, "res"), (1048576, 512, 128, "gelu"), (1048576, 128, 512, "res"), (8192, 8192, 8192, "lin")]
for M, N, K, kind in SHAPES:
    x = E.Act(torch.randn(M * K, generator=g).to(tdt).to(dev), 1, 1, M, K)
    p16 = Packed16(E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev), tdt)
    out = E.Act(torch.empty(M * N, device=dev, dtype=tdt), 1, 1, M, N)
    res = E.Act(torch.randn(M * N, generator=g).to(tdt).to(dev), 1, 1, M, N) if kind == "res" else None
    fl = L.KPF_ACT_GELU if kind == "gelu" else 0
    for _ in range(3):
        conv16(p16, x, kdt, out=out, flags=fl, res=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        conv16(p16, x, kdt, out=out, flags=fl, res=res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byt = 2.0 * (M * K + N * K + M * N * (2 if res is not None else 1))
    print("M=%-8d N=%-5d K=%-5d %-4s %.3f ms  %.0f TF  %.2f TB/s" % (M, N, K, kind, ms, 2.0 * M * N * K / ms / 1e9, byt / ms / 1e9), flush=True)
