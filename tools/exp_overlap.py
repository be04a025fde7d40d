"""Do two hipGraph replays on two HIP streams overlap on this runtime?  Backbone graph (A) and head graph (B) of DIFFERENT slots (no data dependency), each alone,
both together, and the same kernels issued eagerly on the two streams."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
from conftest import synthetic_sd
from keypointfusion_amd.engine import ModelPlan
from keypointfusion_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
net = "KPFusion-convnext-tiny"
plan = ModelPlan(synthetic_sd(net), net, dev, precision="bf16")
b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(32, 128, seed=3).items()}
args = (b["img_rgb"], b["img"], b["pcl"], b["center"], b["M"], b["cube"], b["cam_para"], 0.8, 128, 1)
with torch.no_grad():
    (ga0, gb0, st0, _, _), _ = plan.staged_graphs(*args, slot=0)
    (ga1, gb1, st1, _, _), _ = plan.staged_graphs(*args, slot=1)
    ga0.replay(); gb0.replay(); ga1.replay(); gb1.replay()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def wall(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def on(s, g):
    with torch.cuda.stream(s): g.replay()
print("A alone %.3f ms | B alone %.3f ms" % (wall(lambda: on(sa, ga0)), wall(lambda: on(sb, gb1))))
print("A (stream a) || B of the other slot (stream b): %.3f ms" % wall(lambda: (on(sa, ga0), on(sb, gb1))))
print("A || A' (two backbone graphs): %.3f ms ;  B || B': %.3f ms" % (wall(lambda: (on(sa, ga0), on(sb, ga1))), wall(lambda: (on(sa, gb0), on(sb, gb1)))))
# the same work without graphs
def eager_a():
    with torch.cuda.stream(sa), torch.no_grad(): return plan.backbones(st0[1], st0[0])
bb = eager_a(); torch.cuda.synchronize()
def eager_b():
    with torch.cuda.stream(sb), torch.no_grad(): plan._head(bb, st1[1], st1[2], st1[3], st1[4], st1[5], st1[6], 0.8, 128, 1)
print("eager: A alone %.3f ms | B alone %.3f ms | A || B %.3f ms" % (wall(eager_a), wall(eager_b), wall(lambda: (eager_a(), eager_b()))))
