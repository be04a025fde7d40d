"""Tuning aid: time the split-arithmetic GEMM on the model's shapes for every tile configuration (KPF_FORCE_CFG), one subprocess per
configuration (the override is read once per process).  python tools/split_scan.py [cfg]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(4096, 768, 3072, "res"), (4096, 384, 3456, "lin"), (4096, 768, 1536, "lin"), (4096, 3072, 768, "gelu"), (16384, 192, 1728, "lin"), (16384, 1536, 384, "gelu"), (16384, 384, 1536, "res"), (65536, 768, 192, "gelu"), (65536, 192, 768, "res"), (4096, 3072, 768, "gelu"),
          (4096, 768, 3072, "res"), (262144, 128, 288, "lin"), (262144, 64, 288, "lin"), (65536, 192, 576, "lin"), (16384, 384, 1152, "lin"),
          (262144, 128, 64, "res"), (65536, 192, 96, "res"), (16384, 384, 192, "res")]
CFGS = ["128x128", "128x96", "128x64", "256x48", "128x112", "64x128", "64x64", "32x64", "256x128", "r256x128", "(retired)", "(retired)", "r128x64", "o128x128", "o128x96", "o128x64", "o64x64"]
ONLY = [int(c) for c in os.environ.get("SCAN_CFGS", "").split(",") if c]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from keypointfusion_amd import engine as E, lib as L
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for M, N, K, kind in SHAPES:
        a = torch.randn(M, K, generator=g)
        xs = E.Act(a.to(dev).view(-1), M, 1, 1, K)
        if os.environ.get("PRESPLIT", "1") == "1":
            hi = a.half()
            lo = (a - hi.float()).half()
            sp = torch.stack([hi.reshape(-1, K // 32, 32), lo.reshape(-1, K // 32, 32)], 2).contiguous().view(torch.float32).reshape(M, K)
            xs = E.Act(sp.to(dev).view(-1), M, 1, 1, K, split=True)

        pc = E.PackedConv(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev)
        out = E.Act.empty(M, 1, 1, N, dev)
        res = E.Act(torch.randn(M * N, generator=g).to(dev), M, 1, 1, N)
        kw = dict(flags=L.KPF_ACT_GELU) if kind == "gelu32" else dict(flags=L.KPF_ACT_GELU, out_split=True) if kind == "gelu" else (dict(res=res) if kind == "res" else dict(flags=L.KPF_ACT_RELU, out_split=True))
        E.conv(pc, xs, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            E.conv(pc, xs, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%d %d %d %s %.4f %.1f" % (M, N, K, kind, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    sys.exit(0)

res = {}
for ci, name in enumerate(CFGS):
    if ONLY and ci not in ONLY:
        continue
    env = dict(os.environ, KPF_FORCE_CFG=str(ci), KPF_GEMM="split")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    out = r.stdout
    if r.returncode:
        print(r.stderr[-600:])
    for line in out.strip().splitlines():
        M, N, K, kind, ms, tf = line.split()
        res.setdefault((int(M), int(N), int(K), kind), {})[name] = float(ms)
CFGS = [c for i, c in enumerate(CFGS) if not ONLY or i in ONLY]
print("%-28s" % "shape" + "".join("%9s" % c for c in CFGS))
for k, v in res.items():
    best = min(v.values())
    print("%-28s" % str(k) + "".join("%9s" % ("%.3f%s" % (v.get(c, 0), "*" if v.get(c) == best else " ")) for c in CFGS) +
          "  best %.0f TF-eq" % (2.0 * k[0] * k[1] * k[2] / best / 1e9))
