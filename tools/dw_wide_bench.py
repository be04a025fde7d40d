import os, sys, torch
sys.path.insert(0, os.getcwd())
from keypointfusion_amd import lib as L
from keypointfusion_amd.engine import _ptr, _stream
lib = L.load()
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, H, W, C in ((64, 16, 16, 384), (64, 8, 8, 768), (64, 32, 32, 512), (64, 16, 16, 1024), (64, 8, 8, 384), (64, 4, 4, 768)):
    x = torch.randn(B, H, W, C, device="cuda"); y = torch.empty_like(x)
    w = torch.randn(49, C, device="cuda"); b = torch.randn(C, device="cuda"); lw = torch.randn(C, device="cuda"); lb = torch.randn(C, device="cuda")
    us = t(lambda: lib.kpf_dwconv7_ln_f32(_ptr(x), _ptr(w), _ptr(b), _ptr(lw), _ptr(lb), _ptr(y), B, H, W, C, 1e-6, _stream()))
    xh = x.half(); yh = torch.empty_like(xh)
    ush = t(lambda: lib.kpf_dwconv7_ln_h16(_ptr(xh), _ptr(w), _ptr(b), _ptr(lw), _ptr(lb), _ptr(yh), B, H, W, C, 1e-6, L.KPF_DT_F16, _stream()))
    print(os.environ.get("KPF_DW_WIDE", "1"), (B, H, W, C), "f32 %.1f us %.2f TB/s | f16 %.1f us %.2f TB/s" % (
        us, 2 * x.numel() * 4 / us / 1e6, ush, 2 * x.numel() * 2 / ush / 1e6))
