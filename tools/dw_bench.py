"""Tuning aid: depthwise 7x7 + LN at the four stage shapes of configs[1]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import engine as E, lib as L
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for C, H in ((96, 64), (192, 32), (384, 16), (768, 8)):
    B = 64
    x = torch.randn(B * H * H * C, generator=g).to(dev); y = torch.empty_like(x)
    w = torch.randn(49, C, generator=g).to(dev); b = torch.randn(C, generator=g).to(dev); lw = torch.rand(C, generator=g).to(dev); lb = torch.randn(C, generator=g).to(dev)
    fn = lambda: L.check(L.load().kpf_dwconv7_ln_f32(E._ptr(x), E._ptr(w), E._ptr(b), E._ptr(lw), E._ptr(lb), E._ptr(y), B, H, H, C, 1e-6, E._stream()))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("C=%d H=%d dw+ln %.1f us  (%.2f TB/s of the 4 passes)" % (C, H, ms * 1e3, 4 * B * H * H * C * 4 / ms / 1e9), flush=True)
