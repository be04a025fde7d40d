"""Tuning aid (GPU box): kpf_conv2d_h16 on the shapes of ConvNeXt-T at B = 32, 128 x 128 (configs[2] / configs[3]: grids that do not fill the
chip), every candidate tile case per shape, replayed from a hipGraph of 20 launches (what the model's step does).  Prints us per launch.
usage: python tools/h16_small_sweep.py [bf16|f16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import engine as E, engine16 as E16, lib as L
from keypointfusion_amd.engine16 import DTYPES, Packed16, conv16
dev = torch.device("cuda:0")
tdt, kdt = DTYPES[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
g = torch.Generator().manual_seed(0)
# (B, H, W, Cin, N, k, kind)
SHAPES = [(32, 8, 8, 384, 1536, 1, "gelu"), (32, 8, 8, 1536, 384, 1, "res"), (32, 4, 4, 768, 3072, 1, "gelu"), (32, 4, 4, 3072, 768, 1, "res"),
          (32, 16, 16, 192, 768, 1, "gelu"), (32, 16, 16, 768, 192, 1, "res"), (32, 32, 32, 96, 384, 1, "gelu"), (32, 32, 32, 384, 96, 1, "res"),
          (32, 8, 8, 384, 384, 3, "lin"), (32, 16, 16, 192, 192, 3, "lin"), (32, 32, 32, 96, 96, 3, "lin"), (32, 4, 4, 384, 384, 3, "lin"), (32, 128, 128, 64, 64, 3, "lin"),
          (32, 8, 8, 1152, 384, 1, "lin"), (32, 16, 16, 576, 192, 1, "lin"), (32, 16, 16, 384, 192, 1, "res"), (32, 8, 8, 192, 384, 1, "res"),
          (32, 128, 128, 64, 128, 1, "res"), (32, 128, 128, 128, 64, 1, "lin")]
CASES = [0, 7, 3, 6, 1, 42, 45]  # value = case + 1 (0 = the library's choice)
print("%-44s" % "shape" + "".join("%8s" % ("auto" if c == 0 else "c%d" % (c - 1)) for c in CASES))
for B, H, W, Cin, N, k, kind in SHAPES:
    x = E.Act(torch.randn(B * H * W * Cin, generator=g).to(tdt).to(dev), B, H, W, Cin)
    wt = torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    p16 = Packed16(E.PackedConv(wt, torch.randn(N, generator=g), dev, pad=k // 2), tdt)
    out = E.Act(torch.empty(B * H * W * N, device=dev, dtype=tdt), B, H, W, N)
    res = E.Act(torch.randn(B * H * W * N, generator=g).to(tdt).to(dev), B, H, W, N) if kind == "res" else None
    fl = L.KPF_ACT_GELU if kind == "gelu" else 0
    row, ref = [], None
    for c in CASES:
        E16.FORCE_TILE16 = c
        try:
            conv16(p16, x, kdt, out=out, flags=fl, res=res)
            torch.cuda.synchronize()
            got = out.buf.float().clone()
            if ref is None:
                ref = got
            bad = not torch.equal(got, ref)  # every tile shape accumulates in the same k order: bit-identical results
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(20):
                    conv16(p16, x, kdt, out=out, flags=fl, res=res)
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            row.append("%7.1f%s" % (e0.elapsed_time(e1) * 10.0, "!" if bad else " "))
        except Exception as e:  # noqa: BLE001
            row.append("    err ")
    E16.FORCE_TILE16 = 0
    print("%-44s" % ("M=%d N=%d K=%d %dx%d %s" % (B * H * W, N, Cin * k * k, k, k, kind)) + "".join(row), flush=True)
