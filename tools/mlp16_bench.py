"""GPU box: kpf_convnext_mlp_h16 on the ConvNeXt-B 512^2 stage-1 / stage-2 shapes (KPF_MLP16_DBG = ablation bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import lib as L
from keypointfusion_amd.engine import _ptr, _stream, MLP_HIDDEN_PERM
from keypointfusion_amd.engine16 import DTYPES
dev = torch.device("cuda:0"); lib = L.load()
tdt, kdt = DTYPES[os.environ.get("KPF_PREC", "f16")]
g = torch.Generator().manual_seed(0)
for Cc, M in ((128, 1048576), (256, 262144)):
    y = torch.randn(M, Cc, generator=g).to(tdt).to(dev); x = torch.randn(M, Cc, generator=g).to(tdt).to(dev)
    w1 = (torch.randn(4 * Cc, Cc, generator=g) / Cc ** 0.5).to(tdt).to(dev); w2 = (torch.randn(Cc, 4 * Cc, generator=g) / (4 * Cc) ** 0.5).to(tdt)
    w2c = w2.view(Cc, 4 * Cc // 32, 32)[:, :, torch.tensor(MLP_HIDDEN_PERM)].permute(1, 0, 2).contiguous().to(dev)
    b1, b2, gm = torch.randn(4 * Cc, generator=g).to(dev), torch.randn(Cc, generator=g).to(dev), torch.rand(Cc, generator=g).to(dev)
    out = torch.empty_like(x)
    f = lambda: L.check(lib.kpf_convnext_mlp_h16(_ptr(y), _ptr(x), _ptr(w1), _ptr(b1), _ptr(w2c), _ptr(b2), _ptr(gm), _ptr(out), M, Cc, kdt, _stream()))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print("C=%d M=%d: %.1f us  %.0f TF  %.2f TB/s" % (Cc, M, best * 1e3, 16.0 * M * Cc * Cc / best / 1e9, 6.0 * M * Cc / best / 1e9), flush=True)
