import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import engine as E, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
for M, N, K in [(2048, 2048, 4096), (4096, 4096, 4096)]:
    x = E.Act(torch.randn(M * K, generator=g).to(dev), 1, 1, M, K)
    pc = E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev)
    for _ in range(3):
        out = E.conv(pc, x)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.load().kpf_debug_stamps(buf)
    st, mm, bar, _, tot, nk = [buf[i] for i in range(6)]
    print("M=%d N=%d K=%d nk=%d per-Kstep cycles: dma-issue %.0f mfma %.0f wait+barrier %.0f | total/nk %.0f" % (M, N, K, nk, st / nk, mm / nk, bar / nk, tot / nk))
