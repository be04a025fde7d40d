"""Statistics under the 16-bit accuracy claim: the deviation of the bf16 / f16 modes from the fp32 path in millimetres over MANY
batches — S input seeds x B crops (+ the reference's own sample frame, tests/golden/demo_box_*) — instead of the one batch of four the tolerance test
used.  Runs on the GPU box: the fp32 DEVICE forward is the comparison target (it sits within 1e-4 mm of the CPU oracle: tests/test_parity_gpu.py), so
hundreds of crops cost seconds.  Per precision and stage (3-D / 2-D estimates of the two fusion blocks; the last one is what BASELINE's accuracy metric
is computed on): mean +- sd over seeds of the per-seed mean, median of the per-joint deviations, 90th / 99th percentile, maximum, and JUMPS (> 5 mm:
a ball-query neighbourhood or top-4 pixel set changed by a point) counted, not averaged away.
usage: python tools/precision_stats.py [net=convnext-tiny] [seeds=16] [B=8] [weight_seeds=1]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keypointfusion_amd.model.model import KPFusion  # noqa: E402
from keypointfusion_amd.weights import synthetic_batch, synthetic_state_dict  # noqa: E402

net = "KPFusion-" + (sys.argv[1] if len(sys.argv) > 1 else "convnext-tiny")
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
WS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")


class Loader:
    img_size, flip = 128, 1


def forward(m, b):
    with torch.no_grad():
        res, _, _ = m(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
    return [r.float() for r in res[2:6]]


def box_batch():
    """the reference's sample frame (visualization/box*: committed as windows under tests/golden/) cropped as demo_RGBD.py does"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_preprocess import _frame
    from keypointfusion_amd import preprocess as P
    rgb, depth, bbox, cam = _frame()
    pre = P.prepare_rgbd(rgb, depth, bbox, cam)
    return {k: torch.from_numpy(np.ascontiguousarray(pre[k]))[None].to(dev) for k in ("img_rgb", "img", "pcl", "center", "M", "cube", "cam_para")}


print("%s, %d input seeds x B = %d (+ the sample frame), %d weight seed(s); deviation from the fp32 device forward in mm (x * cube / 2)" % (net, S, B, WS))
for prec in ("bf16", "f16"):
    per_seed_mean = [[] for _ in range(4)]
    allv = [[] for _ in range(4)]
    box = None
    for ws in range(WS):
        sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(net, ws).items()}
        m32 = KPFusion(net, "", 21, "dexycb", "")
        m32.load_state_dict(sd, strict=True)
        m32 = m32.to(dev).eval()
        m16 = KPFusion(net, "", 21, "dexycb", "")
        m16.load_state_dict(sd, strict=True)
        m16 = m16.to(dev).eval()
        m16.precision = prec
        for seed in range(1, S + 1):
            b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=100 * ws + seed).items()}
            r32, r16 = forward(m32, b), forward(m16, b)
            half = b["cube"].view(B, 1, 3) / 2
            for st in range(4):
                d = ((r16[st] - r32[st]) * half).norm(dim=-1).reshape(-1).cpu().numpy()
                per_seed_mean[st].append(d.mean())
                allv[st].append(d)
        if ws == 0:
            bb = box_batch()
            r32, r16 = forward(m32, bb), forward(m16, bb)
            box = [float((((r16[st] - r32[st]) * (bb["cube"].view(1, 1, 3) / 2)).norm(dim=-1)).mean()) for st in range(4)]
        del m32, m16
        torch.cuda.empty_cache()
    print("-- %s" % prec)
    for st, name in enumerate(("block1 3-D", "block1 2-D", "block2 3-D", "block2 2-D (final)")):
        v = np.concatenate(allv[st])
        ps = np.array(per_seed_mean[st])
        print("  %-20s mean %.3f +- %.3f (sd over %d batches; worst batch %.3f)  median %.3f  p90 %.3f  p99 %.3f  max %.2f  jumps > 5 mm: %d of %d (%.2f %%)  sample frame mean %.3f" % (
            name, ps.mean(), ps.std(), len(ps), ps.max(), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max(), int((v > 5).sum()), v.size,
            100.0 * (v > 5).mean(), box[st]))
