"""Which tile configuration does the per-shape search pick for every launch shape of the headline, against the cost model's choice (times in us, isolated launches)?"""
import os, sys, ctypes as C
os.environ["KPF_AUTOTUNE"] = "1"
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
from conftest import synthetic_sd
from keypointfusion_amd import engine as E, lib as L
dev = torch.device("cuda:0")
net = "KPFusion-convnext-tiny"
lib = L.load()
seen = {}
def tuned(lib_, d, x, w, pc, gamma, res, optr, flags):
    ncfg = int(lib_.kpf_conv_num_tile_cfgs())
    cands = [0] + [i + 1 for i in range(ncfg) if i < 9 or i == 17]
    out_t = optr
    if res is not None and res.buf.data_ptr() == optr.data_ptr():
        out_t = torch.empty_like(optr)
    args = (E._ptr(x.buf), E._ptr(w), E._ptr(pc.b), E._ptr(pc.ps), E._ptr(pc.pt), E._ptr(gamma), E._ptr(res.buf if res is not None else None), E._ptr(out_t), E._stream())
    torch.cuda.synchronize()
    times = {}
    for c in cands:
        d.tile_cfg = c
        if lib_.kpf_conv2d_f32(C.byref(d), *args) != 0:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib_.kpf_conv2d_f32(C.byref(d), *args)
        e1.record(); e1.synchronize()
        times[c] = e0.elapsed_time(e1) / 5 * 1e3
    best = min((c for c in times if c), key=lambda c: times[c])
    M = d.B * d.OH * d.OW
    key = (M, d.N, d.KH * d.KW * d.Cin, d.KH, d.flags & 0x3f, res is not None, pc.ps is not None)
    if key not in seen:
        seen[key] = (times, best)
    return best
E._autotune = tuned
plan = E.ModelPlan(synthetic_sd(net), net, dev, precision="f32")
g = torch.Generator().manual_seed(0)
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 256)
img = torch.randn(B, 1, S, S, generator=g).to(dev); rgb = torch.randn(B, 3, S, S, generator=g).to(dev)
plan.serial_streams = True
with torch.no_grad():
    plan.backbones(img, rgb)
tot_d = tot_b = 0.0
for key, (times, best) in sorted(seen.items(), key=lambda kv: -kv[1][0][0]):
    M, N, K, kh, fl, hres, hpro = key
    print("M=%-7d N=%-5d K=%-5d k%d fl=%-2d res=%d pro=%d  default %7.1f us | best cfg %2d %7.1f us (%+5.1f %%) | %s" % (
        M, N, K, kh, fl, hres, hpro, times[0], best - 1, times[best], 100 * (times[best] / times[0] - 1), " ".join("%d:%.0f" % (c - 1, t) for c, t in sorted(times.items()) if c)))
