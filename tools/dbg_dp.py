import os, sys, socket, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch.distributed as dist
from keypointfusion_amd import training as T
from keypointfusion_amd.parallel import live_parameters
import test_training as TT
dev = torch.device("cuda:0")
net = "KPFusion-resnet-18"
sd, batch, loss_fn = TT._train_fixture(net, 4, dev)
def run(dist_mod, **kw):
    torch.manual_seed(0)
    m = TT._fresh(net, sd).to(dev).train(); m.train_dropout = 0.0
    live = live_parameters(m)
    names = [n for n, p in m.named_parameters() if any(p is x for x in live)]
    opt = torch.optim.SGD(live, lr=0.0)
    step = T.GraphedTrainStep(m, opt, loss_fn, batch, warmup=1, dist_mod=dist_mod, params=live, **kw)
    for _ in range(3): step(batch)
    torch.cuda.synchronize()
    return names, [None if p.grad is None else p.grad.detach().clone() for p in live], step
n1, p1, _ = run(None)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
n3, p3, st = run(dist, dp_mode="overlap", bucket_mb=8.0)
bad = [(i, n1[i]) for i, (a, b) in enumerate(zip(p1, p3)) if a is not None and not torch.equal(a, b)]
print(bad)
for i, n in bad[:10]:
    print(n, tuple(p1[i].shape), float((p1[i] - p3[i]).abs().max()), float(p1[i].abs().max()), float(p3[i].abs().max()), "acc count", st._acc_counts.get(id(dict(zip(n3, [None]*len(n3))).get(n)), "?"))
print("early", len(st._early), "late", len(st._late))
for bi, b in enumerate(st._early):
    nm = [n for n, q in zip(n3, [None]*0)]
dist.destroy_process_group()
