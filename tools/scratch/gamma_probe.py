"""Which tensor of the ConvNeXt block is damaged between forward and backward in graph replays >= 1?"""
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
SUMS = {}
class Probe(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, key, other):
        ctx.key = key; ctx.other = other
        SUMS.setdefault(key + ".fwd", torch.zeros((), device=t.device, dtype=torch.float64)).copy_(other.detach().double().abs().sum())
        return t.view_as(t)
    @staticmethod
    def backward(ctx, gt):
        SUMS.setdefault(ctx.key + ".bwd_saved", torch.zeros((), device=gt.device, dtype=torch.float64)).copy_(ctx.other.detach().double().abs().sum())
        SUMS.setdefault(ctx.key + ".bwd_grad", torch.zeros((), device=gt.device, dtype=torch.float64)).copy_(gt.double().abs().sum())
        return gt, None, None
def block(self, p, x):
    c = x.shape[-1]
    y = dwconv7_nhwc(x.float(), self.t[p + ".dwconv.weight"], self.t[p + ".dwconv.bias"])
    y = F.layer_norm(y, (c,), self.t[p + ".norm.weight"], self.t[p + ".norm.bias"], 1e-6)
    y = F.gelu(self.linear(y, p + ".pwconv1.weight", p + ".pwconv1.bias"))
    y = self.linear(y, p + ".pwconv2.weight", p + ".pwconv2.bias")
    out = x + self.t[p + ".gamma"] * y
    if "stages.1." in p:
        out = Probe.apply(out, p, y)
    return out
TG.TrainGraph.convnext_block = block
SITES = os.environ.get("KEEPSITES", "")
if SITES:
    import traceback
    KEEP = []
    COUNT = {}
    re, rel = torch.empty, torch.empty_like
    def site():
        st = traceback.extract_stack(limit=4)
        return "%s:%s" % (os.path.basename(st[-3].filename), st[-3].name)
    def e(*a, **k):
        t = re(*a, **k)
        s_ = site(); COUNT[s_] = COUNT.get(s_, 0) + 1
        if SITES == "all" or any(x in s_ for x in SITES.split(",")):
            KEEP.append(t.detach())
        return t
    def el(*a, **k):
        t = rel(*a, **k)
        s_ = site(); COUNT[s_] = COUNT.get(s_, 0) + 1
        if SITES == "all" or any(x in s_ for x in SITES.split(",")):
            KEEP.append(t.detach())
        return t
    torch.empty, torch.empty_like = e, el
def loss_fn(mdl, bt):
    results, sws, _ = mdl(bt["img_rgb"], bt["img"], bt["pcl"], Loader(), bt["center"], bt["M"], bt["cube"], bt["cam_para"], 0.8)
    return T.kpfusion_loss(results, sws, bt["img"], bt["uvd_gt"], bt["xyz_gt"], epoch=0)[0]
torch.manual_seed(0)
m = KPFusion(net, "", 21, "dexycb", ""); m.load_state_dict(sd, strict=True); m = m.to(dev).train(); m.train_dropout = 0.0
live = live_parameters(m)
names = [n for n, p in m.named_parameters() if any(p is q for q in live)]
opt = torch.optim.SGD(live, lr=0.0)
step = T.GraphedTrainStep(m, opt, loss_fn, batch, warmup=1, params=live)
hist = []
for r in range(3):
    step(batch); torch.cuda.synchronize()
    hist.append({k: float(v) for k, v in SUMS.items()})
    gi = names.index("backbone_rgb.backbone.stages.1.0.gamma")
    print("replay", r, "gamma grad sum", float(live[gi].grad.double().abs().sum()))
for k in sorted(hist[0]):
    vals = [h[k] for h in hist]
    if len(set(vals)) > 1 or k.endswith(".fwd"):
        print(k, vals)

if SITES:
    print("sites:", sorted(COUNT.items(), key=lambda kv: -kv[1])[:20])
