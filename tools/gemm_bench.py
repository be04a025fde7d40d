"""Tuning aid (GPU box): time kpf_conv2d_f32 on isolated GEMM/conv shapes.  usage: gemm_bench.py [reps]
Shapes = the heavy hitters of configs[1] plus a square reference.  KPF_FORCE_CFG=<i> forces a tile configuration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import engine as E, lib as L

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [  # (M, N, K, k, flags)
    (16384, 1536, 384, 1, L.KPF_ACT_GELU), (16384, 384, 1536, 1, 0), (262144, 384, 96, 1, L.KPF_ACT_GELU), (262144, 96, 384, 1, 0),
    (65536, 768, 192, 1, L.KPF_ACT_GELU), (65536, 192, 768, 1, 0), (4096, 3072, 768, 1, L.KPF_ACT_GELU), (4096, 768, 3072, 1, 0),
    (16384, 192, 192 * 9, 3, L.KPF_ACT_RELU), (262144, 64, 64 * 9, 3, L.KPF_ACT_RELU), (262144, 128, 64, 1, 0), (4096, 4096, 4096, 1, 0),
]
only = os.environ.get("KPF_SHAPES")
if only:
    SHAPES = [SHAPES[int(i)] for i in only.split(",")]
g = torch.Generator(device="cpu").manual_seed(0)
for M, N, K, k, fl in SHAPES:
    if k == 1:
        x = E.Act(torch.randn(M * K, generator=g).to(dev), 1, 1, M, K)
        pc = E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev)
    else:
        C = K // 9
        H = int((M // 64) ** 0.5)
        x = E.Act(torch.randn(64 * H * H * C, generator=g).to(dev), 64, H, H, C)
        pc = E.PackedConv(torch.randn(N, C, 3, 3, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev, pad=1)
    out = E.conv(pc, x, flags=fl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        E.conv(pc, x, out=out, flags=fl)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("M=%-7d N=%-5d K=%-5d k=%d fl=%d  %.3f ms  %.1f TF" % (M, N, K, k, fl, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
