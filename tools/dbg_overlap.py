import sys, socket, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
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
    opt = torch.optim.SGD(live, lr=0.0)
    step = T.GraphedTrainStep(m, opt, loss_fn, batch, warmup=1, dist_mod=dist_mod, params=live, **kw)
    losses = [float(step(batch)) for _ in range(3)]
    torch.cuda.synchronize()
    names = [n for n, p in m.named_parameters() if any(p is q for q in live)]
    return losses, [None if p.grad is None else p.grad.detach().clone() for p in live], step, names, live
l1, p1, _, names, _ = run(None)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
l3, p3, st, names3, live3 = run(dist, dp_mode="overlap", bucket_mb=8.0)
print("mode", st.dp_mode, "early", len(st._early), "late", len(st._late), l1, l3)
lateids = {id(q) for b in st._late for q in b["params"]}
for i, (a, b) in enumerate(zip(p1, p3)):
    if a is not None and not torch.equal(a, b):
        print(i, names[i], tuple(a.shape), "late" if id(live3[i]) in lateids else "early", float((a - b).abs().max()), float(a.abs().max()), float(b.abs().max()))
dist.destroy_process_group()
