"""Robustness sweep (GPU box): one training iteration and one eval forward over model families / precisions / odd batch sizes the
benchmarks do not touch; prints loss and finiteness.  Not a parity test (tests/ hold those) — it looks for crashes and NaNs."""
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keypointfusion_amd import training as T  # noqa: E402
from keypointfusion_amd.model.model import KPFusion  # noqa: E402
from keypointfusion_amd.weights import synthetic_batch, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")


class Loader:
    img_size, flip = 128, 1


def run(net, prec, B, train):
    m = KPFusion("KPFusion-" + net, "", 21, "dexycb", "")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict("KPFusion-" + net, 0).items()}, strict=True)
    m = m.to(dev)
    m.precision = prec
    b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=B).items()}
    if train:
        m.train()
        g = torch.Generator().manual_seed(B)
        uvd = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)
        xyz = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)
        m.train_dropout = 0.0
        opt = torch.optim.SGD(m.parameters(), lr=2e-3)  # (plain SGD, no dropout: the loss must go down step over step)
        losses = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            res, sws, _ = m(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
            loss, _ = T.kpfusion_loss(res, sws, b["img"], uvd, xyz, epoch=0)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        bad = [n for n, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        return "loss %s %s, non-finite grads: %d" % (["%.4f" % v for v in losses], "decreasing" if losses[-1] < losses[0] else "NOT DECREASING", len(bad))
    m.eval()
    with torch.no_grad():
        res, sws, _ = m(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
    return "finite %s, joints %s" % (all(bool(torch.isfinite(t).all()) for t in res + sws), tuple(res[5].shape))


CASES = [("convnext-tiny", "f32", 3, True), ("resnet-18", "bf16", 4, True), ("convnext-small", "bf16", 2, True), ("convnext-base", "f32", 2, True),
         ("resnet-50", "f32", 2, True), ("convnext-small", "f16", 5, False), ("convnext-large", "bf16", 1, False), ("resnet-101", "f32", 3, False),
         ("convnext-base", "f32", 7, False)]
fails = 0
for net, prec, B, train in CASES:
    try:
        print("%-15s %-4s B=%d %-5s: %s" % (net, prec, B, "train" if train else "eval", run(net, prec, B, train)), flush=True)
    except Exception:  # noqa: BLE001
        fails += 1
        print("%-15s %-4s B=%d %-5s: FAILED\n%s" % (net, prec, B, "train" if train else "eval", traceback.format_exc()[-1500:]), flush=True)
    torch.cuda.empty_cache()
sys.exit(1 if fails else 0)
