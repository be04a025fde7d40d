import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import lib as L, engine as E
from keypointfusion_amd.engine import _ptr, _stream
from keypointfusion_amd.engine16 import DTYPES
dev = torch.device("cuda:0"); lib = L.load()
tdt, kdt = DTYPES["f16"]
g = torch.Generator().manual_seed(0)
for (H, W, C) in [(32, 32, 512), (64, 64, 256), (128, 128, 128), (16, 16, 1024)]:
    x1 = torch.randn(2, H, W, C, generator=g).to(tdt).to(dev)
    wdw, bdw = (torch.randn(49, C, generator=g) / 7).to(dev), torch.randn(C, generator=g).to(dev)
    lw, lb = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    outs = []
    for rep in (1, 16):
        x = x1.repeat(rep, 1, 1, 1).contiguous(); B = x.shape[0]
        y = torch.empty_like(x); st = torch.empty(lib.kpf_dwconv7_stats_floats(B, H, W, C), device=dev)
        L.check(lib.kpf_dwconv7_stats_h16(_ptr(x), _ptr(wdw), _ptr(bdw), _ptr(y), _ptr(st), B, H, W, C, kdt, _stream()))
        raw = y.clone()
        L.check(lib.kpf_ln_apply_stats_h16(_ptr(y), _ptr(st), _ptr(lw), _ptr(lb), B * H * W, C, 1e-6, kdt, _stream()))
        torch.cuda.synchronize()
        outs.append((raw, st.view(B, -1).clone(), y))
    (r1, s1, y1), (r2, s2, y2) = outs
    print(H, W, C, "raw eq", bool(torch.equal(r2[:2], r1) and torch.equal(r2[-2:], r1)), "stats eq", bool(torch.equal(s2[:2], s1) and torch.equal(s2[-2:], s1)),
          "ln eq", bool(torch.equal(y2[:2], y1) and torch.equal(y2[-2:], y1)))
# fused MLP
import ctypes as C_
from keypointfusion_amd.engine import MLP_HIDDEN_PERM
for Cc in (128, 256):
    M1 = 5000
    y = torch.randn(M1, Cc, generator=g).to(tdt).to(dev); x = torch.randn(M1, Cc, generator=g).to(tdt).to(dev)
    w1 = (torch.randn(4 * Cc, Cc, generator=g) / Cc ** 0.5).to(tdt).to(dev); w2 = (torch.randn(Cc, 4 * Cc, generator=g) / (4 * Cc) ** 0.5).to(tdt)
    w2c = w2.view(Cc, 4 * Cc // 32, 32)[:, :, torch.tensor(MLP_HIDDEN_PERM)].permute(1, 0, 2).contiguous().to(dev)
    b1, b2, gm = torch.randn(4 * Cc, generator=g).to(dev), torch.randn(Cc, generator=g).to(dev), torch.rand(Cc, generator=g).to(dev)
    res = []
    for rep in (1, 7):
        yy, xx = y.repeat(rep, 1).contiguous(), x.repeat(rep, 1).contiguous()
        L.check(lib.kpf_convnext_mlp_h16(_ptr(yy), _ptr(xx), _ptr(w1), _ptr(b1), _ptr(w2c), _ptr(b2), _ptr(gm), _ptr(xx), yy.shape[0], Cc, kdt, _stream()))
        torch.cuda.synchronize(); res.append(xx)
    print("mlp", Cc, bool(torch.equal(res[1][:M1], res[0]) and torch.equal(res[1][-M1:], res[0])))
