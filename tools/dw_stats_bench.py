"""GPU box: depthwise 7x7 + LayerNorm on 16-bit storage, the one-pass kernels (kpf_dwconv7_ln_h16) against the round-4 pair (kpf_dwconv7_stats_h16 +
kpf_ln_apply_stats_h16) on the ConvNeXt-B 512^2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import lib as L
from keypointfusion_amd.engine import _ptr, _stream
from keypointfusion_amd.engine16 import DTYPES
dev = torch.device("cuda:0")
lib = L.load()
prec = os.environ.get("KPF_PREC", "f16")
tdt, kdt = DTYPES[prec]
g = torch.Generator().manual_seed(0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


for B, H, W, C in [(64, 32, 32, 512), (64, 128, 128, 128), (64, 64, 64, 256), (64, 16, 16, 1024), (32, 32, 32, 512)]:
    x = torch.randn(B, H, W, C, generator=g).to(tdt).to(dev)
    wdw, bdw = (torch.randn(49, C, generator=g) / 7).to(dev), torch.randn(C, generator=g).to(dev)
    lw, lb = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    y = torch.empty_like(x)
    st = torch.empty(lib.kpf_dwconv7_stats_floats(B, H, W, C), device=dev)
    t_old = timeit(lambda: L.check(lib.kpf_dwconv7_ln_h16(_ptr(x), _ptr(wdw), _ptr(bdw), _ptr(lw), _ptr(lb), _ptr(y), B, H, W, C, 1e-6, kdt, _stream())))
    t_st = timeit(lambda: L.check(lib.kpf_dwconv7_stats_h16(_ptr(x), _ptr(wdw), _ptr(bdw), _ptr(y), _ptr(st), B, H, W, C, kdt, _stream())))
    t_ap = timeit(lambda: L.check(lib.kpf_ln_apply_stats_h16(_ptr(y), _ptr(st), _ptr(lw), _ptr(lb), B * H * W, C, 1e-6, kdt, _stream())))
    mb = 2.0 * B * H * W * C * 2 / 1e6
    print("%s %dx%dx%dx%d: one-pass %.1f us | stencil+stats %.1f us (%.2f TB/s) + apply %.1f us (%.2f TB/s) = %.1f us" % (
        prec, B, H, W, C, t_old, t_st, mb / t_st, t_ap, mb / t_ap, t_st + t_ap), flush=True)
