"""Tuning aid (needs a -DKPF_DBG_TIME build, KPF_LIB_PATH): per-workgroup cycle stamps of the igemm kernel — prologue / main loop / epilogue."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keypointfusion_amd import engine as E, lib as L
dev = torch.device("cuda:0")
lib = L.load()
g = torch.Generator().manual_seed(0)
for M, N, K, kind in [(16384, 1536, 384, "gelu"), (16384, 384, 1536, "res")]:
    a = torch.randn(M, K, generator=g)
    hi = a.half(); lo = (a - hi.float()).half()
    sp = torch.stack([hi.reshape(-1, K // 32, 32), lo.reshape(-1, K // 32, 32)], 2).contiguous().view(torch.float32).reshape(M, K)
    xs = E.Act(sp.to(dev).view(-1), M, 1, 1, K, split=True)
    pc = E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev)
    out = E.Act.empty(M, 1, 1, N, dev)
    res = E.Act(torch.randn(M * N, generator=g).to(dev), M, 1, 1, N)
    kw = dict(flags=L.KPF_ACT_GELU, out_split=True) if kind == "gelu" else dict(res=res)
    for _ in range(3):
        E.conv(pc, xs, out=out, **kw)
    torch.cuda.synchronize()
    n = 4 * 8192
    buf = (C.c_ulonglong * n)()
    lib.kpf_dbg_read.argtypes = [C.c_void_p, C.c_int]
    assert lib.kpf_dbg_read(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
    t = t[t[:, 0] > 0]
    t = t[(t[:, 3] > t[:, 0])]
    print(M, N, K, kind, "blocks", len(t), "cycles: prologue %.0f  mainloop %.0f  epilogue %.0f  total %.0f ; span of launch %.0f" % (
        (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean(), (t[:, 3] - t[:, 0]).mean(), t[:, 3].max() - t[:, 0].min()))
