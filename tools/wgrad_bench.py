"""Times kpf_conv2d_wgrad_f32 / kpf_dwconv7_wgrad_f32 on the training step's shapes (B = 32, 128x128 crops, ConvNeXt-T) next to the
library paths they replace (rocBLAS dY^T X, torch.nn.grad.conv2d_weight)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointfusion_amd.training import conv_wgrad_hip, DwConv7NHWC  # noqa: E402


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 32
shapes = [(32, 96, 384, 1), (32, 384, 96, 1), (16, 192, 768, 1), (16, 768, 192, 1), (8, 384, 1536, 1), (8, 1536, 384, 1), (4, 768, 3072, 1),
          (4, 3072, 768, 1), (32, 64, 64, 3), (16, 96, 96, 3), (64, 128, 128, 1), (64, 64, 64, 3), (32, 144, 48, 1)]
for hw, cin, n, k in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda")
    dy = torch.randn(B, hw, hw, n, device="cuda")
    us = t(lambda: conv_wgrad_hip(dy, x, (n, cin, k, k), 1, k // 2, True))
    if k == 1:
        ref = t(lambda: dy.view(-1, n).t() @ x.view(-1, cin))
    else:
        xc, dyc = x.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2)
        ref = t(lambda: torch.nn.grad.conv2d_weight(xc, (n, cin, k, k), dyc, stride=1, padding=k // 2))
    fl = 2.0 * B * hw * hw * n * cin * k * k
    print(f"wgrad {hw}x{hw} cin={cin} n={n} k={k}: hip {us:8.1f} us ({fl / us / 1e6:6.1f} TF)   library {ref:8.1f} us")
for hw, c in [(32, 96), (16, 192), (8, 384), (4, 768)]:
    x = torch.randn(B, hw, hw, c, device="cuda", requires_grad=True)
    w = torch.randn(c, 1, 7, 7, device="cuda", requires_grad=True)
    b = torch.randn(c, device="cuda", requires_grad=True)
    dy = torch.randn(B, hw, hw, c, device="cuda")

    def hip():
        y = DwConv7NHWC.apply(x, w, b)
        y.backward(dy)

    xc = x.detach().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    dyc = dy.permute(0, 3, 1, 2).contiguous()

    def lib():
        y = F.conv2d(xc, w, b, padding=3, groups=c)
        y.backward(dyc)

    print(f"dw7 fwd+bwd {hw}x{hw} C={c}: hip {t(hip):8.1f} us   library {t(lib):8.1f} us")
