"""Per-shape tile search over the 16-bit launches of one forward (default: configs[4], ConvNeXt-B 512 x 512 f16, B = 64): time with the dispatcher's choice
against every tile case kpf_conv2d_h16 accepts for the launch (isolated launches, us)."""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
from conftest import synthetic_sd
from keypointfusion_amd import engine as E, engine16 as E16, lib as L
dev = torch.device("cuda:0")
net, B, S, prec = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]) if len(sys.argv) > 4 else ("KPFusion-convnext-base", 64, 512, "f16")
lib = L.load()
CASES = (0, 1, 2, 5, 6, 8, 20, 21, 22, 26, 30, 41, 44)
seen = {}
orig_check = L.check
real = lib.kpf_conv2d_h16
def timed_call(d, *args):
    rc = real(d, *args)
    dd = d._obj
    M = dd.B * dd.OH * dd.OW
    key = (M, dd.N, dd.KH * dd.KW * dd.Cin, dd.KH, dd.flags)
    if key in seen or torch.cuda.is_current_stream_capturing() or (dd.flags & L.KPF_PRO_LN) or getattr(dd, "groups", 0) > 1:
        return rc
    times = {}
    keep = dd.tile_cfg
    for c in (None,) + CASES:
        dd.tile_cfg = 0 if c is None else c + 1
        if real(d, *args) != 0:
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            real(d, *args)
        e1.record(); e1.synchronize()
        times[c] = e0.elapsed_time(e1) / 5 * 1e3
    dd.tile_cfg = keep
    seen[key] = times
    return rc
lib.kpf_conv2d_h16 = timed_call
plan = E.ModelPlan(synthetic_sd(net), net, dev, precision=prec)
plan.serial_streams = True
g = torch.Generator().manual_seed(0)
img = torch.randn(B, 1, S, S, generator=g).to(dev); rgb = torch.randn(B, 3, S, S, generator=g).to(dev)
with torch.no_grad():
    plan.backbones(img, rgb)
    seen.clear()  # (first pass: warm-up of every kernel)
    plan.backbones(img, rgb)
for key, times in sorted(seen.items(), key=lambda kv: -kv[1].get(None, 0)):
    M, N, K, kh, fl = key
    cand = {c: t for c, t in times.items() if c is not None}
    if not cand or None not in times:
        continue
    best = min(cand, key=cand.get)
    print("M=%-8d N=%-5d K=%-5d k%d fl=%-5d default %7.1f us | best case %2d %7.1f us (%+5.1f %%) | %s" % (
        M, N, K, kh, fl, times[None], best, cand[best], 100 * (cand[best] / times[None] - 1), " ".join("%d:%.0f" % (c, t) for c, t in sorted(cand.items()))))
