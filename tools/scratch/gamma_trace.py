"""When does y of backbone_rgb.stages.1.0 become zero in graph replays >= 1?  Checksums of that tensor recorded inside the captured graph
after every forward component and before every backward component."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch.nn.functional as F
from conftest import synthetic_sd
from keypointfusion_amd import training as T, train_graph as TG
from keypointfusion_amd.model.model import KPFusion
from keypointfusion_amd.parallel import live_parameters
from keypointfusion_amd.weights import synthetic_batch
from keypointfusion_amd.training import dwconv7_nhwc
net = "KPFusion-convnext-tiny"; B = 4; dev = torch.device("cuda:0")
sd = synthetic_sd(net)
batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=5).items()}
g = torch.Generator().manual_seed(1)
batch["uvd_gt"] = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)
batch["xyz_gt"] = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)
class Loader: img_size, flip = 128, 1
SUMS, ORDER, Y = {}, [], {}
def rec(key):
    if "y" not in Y: return
    if key not in SUMS:
        SUMS[key] = torch.zeros((), device=dev, dtype=torch.float64); ORDER.append(key)
    SUMS[key].copy_(Y["y"].detach().double().abs().sum())
class Mark(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, key):
        ctx.key = key
        rec("F " + key)
        return t.view_as(t)
    @staticmethod
    def backward(ctx, gt):
        rec("B " + ctx.key)
        return gt, None
CNT = [0]
def wrap(name):
    orig = getattr(TG.TrainGraph, name)
    def f(self, p, x, *a, **k):
        out = orig(self, p, x, *a, **k)
        CNT[0] += 1
        if isinstance(out, torch.Tensor) and out.requires_grad:
            out = Mark.apply(out, "%03d %s %s" % (CNT[0], name, p if isinstance(p, str) else x))
        return out
    setattr(TG.TrainGraph, name, f)
def block(self, p, x):
    c = x.shape[-1]
    y = dwconv7_nhwc(x.float(), self.t[p + ".dwconv.weight"], self.t[p + ".dwconv.bias"])
    y = F.layer_norm(y, (c,), self.t[p + ".norm.weight"], self.t[p + ".norm.bias"], 1e-6)
    y = F.gelu(self.linear(y, p + ".pwconv1.weight", p + ".pwconv1.bias"))
    y = self.linear(y, p + ".pwconv2.weight", p + ".pwconv2.bias")
    if p == "backbone_rgb.backbone.stages.1.0":
        Y["y"] = y
    return x + self.t[p + ".gamma"] * y
TG.TrainGraph.convnext_block = block
for n in ("convnext_block", "residual", "conv_l", "bn_l"):
    wrap(n)
def loss_fn(mdl, bt):
    CNT[0] = 0; Y.clear()
    results, sws, _ = mdl(bt["img_rgb"], bt["img"], bt["pcl"], Loader(), bt["center"], bt["M"], bt["cube"], bt["cam_para"], 0.8)
    rec("F end of forward")
    l = T.kpfusion_loss(results, sws, bt["img"], bt["uvd_gt"], bt["xyz_gt"], epoch=0)[0]
    rec("F loss")
    return l
torch.manual_seed(0)
m = KPFusion(net, "", 21, "dexycb", ""); m.load_state_dict(sd, strict=True); m = m.to(dev).train(); m.train_dropout = 0.0
live = live_parameters(m)
opt = torch.optim.SGD(live, lr=0.0)
step = T.GraphedTrainStep(m, opt, loss_fn, batch, warmup=1, params=live)
hist = []
for r in range(3):
    step(batch); torch.cuda.synchronize()
    hist.append({k: float(v) for k, v in SUMS.items()})
prev = None
for k in ORDER:
    vals = [h[k] for h in hist]
    sig = tuple(v == vals[0] for v in vals)
    if sig != prev:
        print(k, vals)
    prev = sig
