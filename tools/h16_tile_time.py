"""Tuning aid (GPU box): where an igemm_h16_kernel workgroup spends its cycles and the clock it runs at.  Needs a library whose
kpf_conv16.hip was compiled with -DKPF_DBG_TIME:
  hipcc <CXXFLAGS of the Makefile> -DKPF_DBG_TIME -c kpf_conv16.hip -o /tmp/c16dbg.o; link it instead of kpf_conv16.o into libkpf_hip_dbg16.so
usage: KPF_LIB_PATH=keypointfusion_amd/libkpf_hip_dbg16.so [KPF_FORCE_CFG16=..] python tools/h16_tile_time.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keypointfusion_amd import engine as E, lib as L
from keypointfusion_amd.engine16 import DTYPES, Packed16, conv16
dev = torch.device("cuda:0")
lib = L.load()
lib.kpf_dbg_read.argtypes = [C.c_void_p, C.c_int]
tdt, kdt = DTYPES[os.environ.get("KPF_PREC", "bf16")]
g = torch.Generator().manual_seed(0)
SHAPES = [(65536, 2048, 512, "gelu"), (65536, 512, 2048, "res"), (16384, 1024, 4096, "res"), (8192, 8192, 8192, "lin")]
for M, N, K, kind in SHAPES:
    x = E.Act(torch.randn(M * K, generator=g).to(tdt).to(dev), 1, 1, M, K)
    p16 = Packed16(E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev), tdt)
    out = E.Act(torch.empty(M * N, device=dev, dtype=tdt), 1, 1, M, N)
    res = E.Act(torch.randn(M * N, generator=g).to(tdt).to(dev), 1, 1, M, N) if kind == "res" else None
    fl = L.KPF_ACT_GELU if kind == "gelu" else 0
    for _ in range(20):
        conv16(p16, x, kdt, out=out, flags=fl, res=res)
    torch.cuda.synchronize()
    lib.kpf_dbg_clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    conv16(p16, x, kdt, out=out, flags=fl, res=res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    n = 8 * 8192
    buf = (C.c_ulonglong * n)()
    assert lib.kpf_dbg_read(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    t = t[(t[:, 0] > 0) & (t[:, 3] > t[:, 0])]
    clk = (t[:, 3] - t[:, 0]).sum() / max(1, (t[:, 5] - t[:, 4]).sum()) * 100.0
    span = t[:, 3].max() - t[:, 0].min()
    print("M=%d N=%d K=%d %s: %.3f ms %.0f TF | %d WGs | clock %.0f MHz | cycles/WG: prologue %.0f main %.0f epilogue %.0f total %.0f | span %.0f cyc"
          % (M, N, K, kind, ms, 2.0 * M * N * K / ms / 1e9, len(t), clk, (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(),
             (t[:, 3] - t[:, 2]).mean(), (t[:, 3] - t[:, 0]).mean(), span), flush=True)
