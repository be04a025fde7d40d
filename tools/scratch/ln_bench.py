import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from keypointfusion_amd.training import layer_norm_rows, gelu_rows
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, C in ((32768, 96), (8192, 192), (2048, 384), (512, 768), (672, 128)):
    x = torch.randn(rows, C, device="cuda", requires_grad=True); w = torch.randn(C, device="cuda", requires_grad=True); b = torch.randn(C, device="cuda", requires_grad=True)
    dy = torch.randn(rows, C, device="cuda")
    def f_h(): return layer_norm_rows(x, w, b, 1e-6)
    def f_t(): return F.layer_norm(x, (C,), w, b, 1e-6)
    yh, yt = f_h(), f_t()
    def b_h(): torch.autograd.grad(yh, (x, w, b), dy, retain_graph=True)
    def b_t(): torch.autograd.grad(yt, (x, w, b), dy, retain_graph=True)
    print("LN %6d x %4d  fwd hip %.1f torch %.1f us | bwd hip %.1f torch %.1f us" % (rows, C, t(f_h), t(f_t), t(b_h), t(b_t)))
for n in (32768 * 384, 8192 * 768, 2048 * 1536, 672 * 16):
    x = torch.randn(n, device="cuda", requires_grad=True); dy = torch.randn(n, device="cuda")
    yh, yt = gelu_rows(x), F.gelu(x)
    print("GELU %9d fwd hip %.1f torch %.1f us | bwd hip %.1f torch %.1f us" % (n, t(lambda: gelu_rows(x)), t(lambda: F.gelu(x)),
          t(lambda: torch.autograd.grad(yh, x, dy, retain_graph=True)), t(lambda: torch.autograd.grad(yt, x, dy, retain_graph=True))))
