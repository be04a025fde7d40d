import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from conftest import synthetic_sd, GOLDEN
from keypointfusion_amd.model.model import KPFusion
from keypointfusion_amd import training as T
from keypointfusion_amd.weights import synthetic_batch
net = sys.argv[1] if len(sys.argv) > 1 else "convnext-tiny"
Zs = np.load(os.path.join(GOLDEN, "train_step_%s.npz" % net))
dev = torch.device("cuda:0")
m = KPFusion("KPFusion-" + net, "", 21, "dexycb", "")
m.load_state_dict(synthetic_sd("KPFusion-" + net), strict=True)
m = m.to(dev).train(); m.train_dropout = 0.0
b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(3, 128, seed=11).items()}
uvd_gt, xyz_gt = torch.from_numpy(Zs["uvd_gt"]).to(dev), torch.from_numpy(Zs["xyz_gt"]).to(dev)
class Loader: img_size, flip = 128, 1
m._ball_override = [torch.from_numpy(Zs["ball_idx"][i].astype(np.int64)) for i in range(6)]
results, sws, _ = m(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
loss, parts = T.kpfusion_loss(results, sws, b["img"], uvd_gt, xyz_gt, epoch=0)
print("ball flips", m._last_ball_flips)
print("r3d1 maxdiff", float((results[2].detach().cpu() - torch.from_numpy(Zs["r3d1"])).abs().max()), "r2d2", float((results[5].detach().cpu() - torch.from_numpy(Zs["r2d2"])).abs().max()))
print("loss", float(loss), float(Zs["loss"]))
for k, v in parts.items(): print(k, float(v), float(Zs[k]))
loss.backward()
ref = dict(zip([str(n) for n in Zs["grad_names"]], Zs["grad_norms"]))
got = {n: float(p.grad.double().norm()) for n, p in m.named_parameters() if p.grad is not None}
print("sets equal", set(got) == set(ref), len(got), len(ref), sorted(set(got) ^ set(ref))[:6])
rel = sorted(((abs(got[n] - ref[n]) / (ref[n] + 1e-12), n, got[n], ref[n]) for n in ref if n in got), reverse=True)
for r in rel[:12]: print("%.3e %s %.4e %.4e" % r)
import statistics
print("median rel", statistics.median(r[0] for r in rel))
for top in ("backbone_d", "backbone_rgb", "block1", "block2"):
    a = sum(v * v for n, v in got.items() if n.startswith(top)) ** 0.5
    r = sum(v * v for n, v in ref.items() if n.startswith(top)) ** 0.5
    print(top, a, r, abs(a - r) / r)
