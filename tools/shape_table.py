"""Per-shape table of the backbones' MFMA launches (GPU box): both backbones on one stream, HIP events per launch, 5 passes
averaged; sorted by total time.  Shows where the GEMM kernel's step time goes and which shapes sit furthest below the roof.
Columns: time per launch, TFLOP/s, algorithmic TB/s (inputs + weights + outputs once), and the fraction of EACH roof the launch reaches —
mfma = TFLOP/s / dense MFMA peak of its arithmetic (157.3 f32, 2500 16-bit), hbm = TB/s / 8 — `roof` names the higher of the two: the roof
that bounds the shape as executed.
usage: python tools/shape_table.py [B=64] [S=256] [precision=f32] [net=KPFusion-convnext-tiny]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keypointfusion_amd import engine as E  # noqa: E402
from keypointfusion_amd.model.model import KPFusion  # noqa: E402
from keypointfusion_amd.weights import synthetic_batch, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
net = sys.argv[4] if len(sys.argv) > 4 else "KPFusion-convnext-tiny"
m = KPFusion(net, "", 21, "dexycb", "")
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(net, 0).items()}, strict=True)
m = m.to(dev).eval()
B_, S_ = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 256
m.precision = sys.argv[3] if len(sys.argv) > 3 else "f32"
b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B_, S_, seed=1).items()}
plan = m._plan(dev)
plan.serial_streams = True
acc = collections.OrderedDict()
with torch.no_grad():
    for it in range(7):
        E.PROFILE = []
        plan.backbones(b["img"], b["img_rgb"])
        torch.cuda.synchronize()
        if it >= 2:
            for name, e0, e1, fl, nb, shp in E.PROFILE:
                d = acc.setdefault((name, shp), [0, 0.0, fl, nb])
                d[0] += 1
                d[1] += e0.elapsed_time(e1)
E.PROFILE = None
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows) / 5
print("%s B=%d %dx%d %s: total MFMA-kernel ms/step %.3f" % (net, B_, S_, S_, m.precision, tot))
cum = 0.0
for (name, shp), (n, ms, fl, nb) in rows[:60]:
    per = ms / n
    tf, tb = fl / per / 1e9, nb / per / 1e9
    peak = 2500.0 if ("h16" in name or "gemm16" in name) else 157.3
    fm, fh = tf / peak, tb / 8.0
    cum += ms / 5
    print("%-22s M=%-7d N=%-5d K=%-5d %dx%d  x%-3d %7.1f us  %6.1f TF  %5.2f TB/s  mfma %.2f hbm %.2f roof %-4s %5.2f ms/step  cum %5.2f" % (
        name[:22], shp[0], shp[1], shp[2], shp[3], shp[4], n // 5, per * 1e3, tf, tb, fm, fh, "mfma" if fm >= fh else "hbm", ms / 5, cum))
