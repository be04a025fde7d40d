"""Tuning aid (GPU box; needs `make -C keypointfusion_amd/csrc dbg`): where an igemm_f32_kernel workgroup spends its cycles.
Per shape: mean cycles of prologue (launch -> main loop), main loop, epilogue; the in-kernel clock (shader cycles per 100-MHz tick);
MFMA-issue-bound cycles of the main loop for comparison; tiles per CU and how long each CU was busy.
usage: KPF_LIB_PATH=keypointfusion_amd/libkpf_hip_dbg.so python tools/f32_tile_time.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keypointfusion_amd import engine as E, lib as L
assert E.GEMM_MODE == "f32"
dev = torch.device("cuda:0")
lib = L.load()
lib.kpf_dbg_read.argtypes = [C.c_void_p, C.c_int]
g = torch.Generator().manual_seed(0)
SHAPES = [(16384, 1536, 384, "gelu"), (16384, 384, 1536, "res"), (4096, 3072, 768, "gelu"), (4096, 768, 3072, "res"), (65536, 768, 192, "gelu"),
          (65536, 192, 768, "res"), (262144, 128, 64, "res"), (262144, 96, 48, "res"), (16384, 384, 192, "res"), (4096, 4096, 4096, "lin")]
for M, N, K, kind in SHAPES:
    x = E.Act(torch.randn(M * K, generator=g).to(dev), M, 1, 1, K)
    pc = E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev)
    out = E.Act.empty(M, 1, 1, N, dev)
    res = E.Act(torch.randn(M * N, generator=g).to(dev), M, 1, 1, N)
    kw = dict(flags=L.KPF_ACT_GELU) if kind == "gelu" else (dict(res=res) if kind == "res" else {})
    for _ in range(20):
        E.conv(pc, x, out=out, **kw)
    torch.cuda.synchronize()
    lib.kpf_dbg_clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    E.conv(pc, x, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    n = 8 * 8192
    buf = (C.c_ulonglong * n)()
    assert lib.kpf_dbg_read(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    t = t[(t[:, 0] > 0) & (t[:, 3] > t[:, 0])]
    nb = len(t)
    clk = (t[:, 3] - t[:, 0]).sum() / max(1, (t[:, 5] - t[:, 4]).sum()) * 100.0  # MHz
    hw = t[:, 6] & 0xFFFFFFFF
    cu = ((t[:, 6] >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)  # xcc, se, sh, cu
    per_cu = {}
    for c, row in zip(cu, t):
        per_cu.setdefault(int(c), []).append(row)
    tiles = np.array([len(v) for v in per_cu.values()])
    busy = np.array([max(r[3] for r in v) - min(r[0] for r in v) for v in per_cu.values()])
    span = t[:, 3].max() - t[:, 0].min()
    tot = (t[:, 3] - t[:, 0])
    print("M=%d N=%d K=%d %s: %.3f ms %.1f TF | blocks %d on %d CUs (tiles/CU min %d max %d) | clock %.0f MHz | cycles/WG: prologue %.0f main %.0f "
          "epilogue %.0f total %.0f | launch span %.0f cyc = %.3f ms, CU busy mean %.0f" % (
              M, N, K, kind, ms, 2.0 * M * N * K / ms / 1e9, nb, len(per_cu), tiles.min(), tiles.max(), clk, (t[:, 1] - t[:, 0]).mean(),
              (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean(), tot.mean(), span, span / clk / 1e3, busy.mean()), flush=True)
