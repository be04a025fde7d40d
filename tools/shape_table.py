"""Per-shape table of the backbones' MFMA launches (GPU box): both backbones on one stream, HIP events per launch, 5 passes
averaged; sorted by total time.  Shows where the GEMM kernel's step time goes and which shapes sit furthest below the roof.
usage: python tools/shape_table.py [B=64] [S=256] [precision=f32]"""
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
net = "KPFusion-convnext-tiny"
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
print("total MFMA-kernel ms/step %.3f" % tot)
for (name, shp), (n, ms, fl, nb) in rows[:40]:
    per = ms / n
    print("%-22s M=%-7d N=%-5d K=%-5d %dx%d  x%-3d %7.1f us  %6.1f TF  %5.2f TB/s  %5.2f ms/step" % (
        name[:22], shp[0], shp[1], shp[2], shp[3], shp[4], n // 5, per * 1e3, fl / per / 1e9, nb / per / 1e9, ms / 5))
