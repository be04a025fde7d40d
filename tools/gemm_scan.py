"""Tuning aid: efficiency of the GEMM kernel vs number of tiles per CU (tail / occupancy effects)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import engine as E, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
for M, N, K in [(2048, 2048, 8192), (4096, 2048, 8192), (4096, 3072, 4096), (4096, 4096, 4096), (8192, 4096, 4096), (16384, 4096, 2048), (16384, 8192, 1024), (32768, 8192, 512)]:
    x = E.Act(torch.randn(M * K, generator=g).to(dev), 1, 1, M, K)
    pc = E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev)
    out = E.conv(pc, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        E.conv(pc, x, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("M=%-6d N=%-5d K=%-5d tiles/CU=%.1f  %.3f ms  %.1f TF" % (M, N, K, (M / 128) * (N / 128) / 256, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
