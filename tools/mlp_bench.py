"""Tuning aid: fused ConvNeXt MLP kernel vs the two plain GEMM launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import engine as E, lib as L
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for C, M in ((96, 262144), (192, 65536), (128, 262144)):
    y = torch.randn(M, C, generator=g).to(dev); x = torch.randn(M, C, generator=g).to(dev)
    w1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(dev); b1 = torch.randn(4 * C, generator=g).to(dev)
    w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(dev); b2 = torch.randn(C, generator=g).to(dev); gam = torch.rand(C, generator=g).to(dev)
    out = torch.empty_like(x)
    def fused():
        L.check(L.load().kpf_convnext_mlp_f32(E._ptr(y), E._ptr(x), E._ptr(w1), E._ptr(b1), E._ptr(w2), E._ptr(b2), E._ptr(gam), E._ptr(out), M, C, E._stream()))
    pc1 = E.PackedConv(w1, b1, dev); pc2 = E.PackedConv(w2, b2, dev)
    ya = E.Act(y.view(-1), 1, 1, M, C); xa = E.Act(x.view(-1), 1, 1, M, C); h = E.Act.empty(1, 1, M, 4 * C, dev); oa = E.Act(out.view(-1), 1, 1, M, C)
    def plain():
        E.conv(pc1, ya, out=h, flags=L.KPF_ACT_GELU); E.conv(pc2, h, out=oa, gamma=gam, res=xa)
    for name, fn in (("fused", fused), ("plain", plain)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("C=%d M=%d %s %.3f ms %.1f TF" % (C, M, name, ms, 16.0 * M * C * C / ms / 1e9), flush=True)
