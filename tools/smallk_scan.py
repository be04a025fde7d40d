"""Tuning aid (GPU box): tile configuration scan for the memory-bound small-K 1x1 layers of the decoder (K <= 192), f32 arithmetic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd import engine as E, lib as L
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
SHAPES = [(262144, 128, 64, "res"), (262144, 96, 48, "res"), (262144, 64, 128, "pro"), (262144, 48, 96, "pro"), (65536, 192, 96, "res"), (65536, 96, 192, "pro"),
          (16384, 384, 192, "res"), (16384, 192, 384, "pro"), (262144, 105, 128, "nchw")]
for M, N, K, kind in SHAPES:
    x = E.Act(torch.randn(M * K, generator=g).to(dev), M // 4096, 64, 64, K) if M % 4096 == 0 else E.Act(torch.randn(M * K, generator=g).to(dev), 1, 1, M, K)
    pro = (torch.rand(K, generator=g).double() + 0.5, torch.randn(K, generator=g).double()) if kind == "pro" else None
    pc = E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev, prologue=pro)
    out = E.Act.empty(x.B, x.H, x.W, N, dev)
    res = E.Act(torch.randn(M * N, generator=g).to(dev), x.B, x.H, x.W, N)
    nchw = torch.empty(x.B, N, x.H, x.W, device=dev) if kind == "nchw" else None
    line = "M=%d N=%d K=%d %s:" % (M, N, K, kind)
    for cfg in (0, 1, 2, 5, 6, 7, 8, 3):
        pc.tuned = {}
        E.AUTOTUNE = False
        d_cfg = cfg + 1
        def run():
            import ctypes as C
            # go through conv() with a forced tile_cfg via the autotune cache key mechanism: simplest is the env override per call
            os.environ["KPF_FORCE_CFG_PY"] = str(cfg)
            if kind == "res":
                return E.conv(pc, x, out=out, res=res)
            if kind == "nchw":
                return E.conv(pc, x, out_nchw=nchw)
            return E.conv(pc, x, out=out, flags=L.KPF_ACT_RELU)
        try:
            E.FORCE_TILE = d_cfg
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += "  cfg%d %.1fus" % (cfg, ms * 1e3)
        except Exception as ex:  # noqa: BLE001
            line += "  cfg%d ERR" % cfg
        finally:
            E.FORCE_TILE = 0
    print(line, flush=True)
