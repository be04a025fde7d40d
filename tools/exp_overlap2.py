"""Host time of each call of the stage pipeline's submit (where does the host block?)"""
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
ents = []
with torch.no_grad():
    for s in range(2):
        e, ins = plan.staged_graphs(*args, slot=s)
        ents.append(e)
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
head_done = [None, None]
T = {"wait": 0.0, "copy": 0.0, "A": 0.0, "evA": 0.0, "B": 0.0, "clone": 0.0}
def tick(k, t0):
    t1 = time.perf_counter(); T[k] += t1 - t0; return t1
N = 40
torch.cuda.synchronize()
t_start = time.perf_counter()
for i in range(N):
    slot = i % 2
    ga, gb, static, res, sws = ents[slot]
    t0 = time.perf_counter()
    if head_done[slot] is not None: sa.wait_event(head_done[slot])
    t0 = tick("wait", t0)
    with torch.cuda.stream(sa):
        for d, t in zip(static, ins): d.copy_(t)
        t0 = tick("copy", t0)
        ga.replay()
        t0 = tick("A", t0)
        ev = torch.cuda.Event(); ev.record(sa)
        t0 = tick("evA", t0)
    with torch.cuda.stream(sb):
        sb.wait_event(ev)
        gb.replay()
        t0 = tick("B", t0)
        out = [t.clone() for t in res]
        e2 = torch.cuda.Event(); e2.record(sb)
        t0 = tick("clone", t0)
    head_done[slot] = e2
t_issue = time.perf_counter() - t_start
torch.cuda.synchronize()
t_all = time.perf_counter() - t_start
print("per batch: issue %.3f ms, total %.3f ms; host ms per call:" % (t_issue / N * 1e3, t_all / N * 1e3), {k: round(v / N * 1e3, 3) for k, v in T.items()})
