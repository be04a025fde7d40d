"""Which layers still pack their operand per use (DevPack.packed outside the PackCache)?  One eager training iteration, calls counted by caller."""
import os, sys, collections, traceback, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import synthetic_sd
from keypointfusion_amd import training as T
from keypointfusion_amd.model.model import KPFusion
from keypointfusion_amd.weights import synthetic_batch
net, B, dev = "KPFusion-convnext-tiny", 4, torch.device("cuda:0")
batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=5).items()}
g = torch.Generator().manual_seed(1)
uvd, xyz = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev), (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)
class Loader: img_size, flip = 128, 1
m = KPFusion(net, "", 21, "dexycb", ""); m.load_state_dict(synthetic_sd(net), strict=True); m = m.to(dev).train(); m.precision = "bf16"
def it():
    for p in m.parameters(): p.grad = None
    r, s, _ = m(batch["img_rgb"], batch["img"], batch["pcl"], Loader(), batch["center"], batch["M"], batch["cube"], batch["cam_para"], 0.8)
    T.kpfusion_loss(r, s, batch["img"], uvd, xyz, epoch=0)[0].backward()
it(); it()
cnt = collections.Counter()
orig = T.DevPack.packed.__func__
def spy(cls, weight, bias, mode, prec="f32", **kw):
    fr = [f for f in traceback.extract_stack(limit=12) if "keypointfusion_amd" in f.filename and f.name not in ("spy", "packed", "get")]
    cnt[(" <- ".join("%s:%d" % (f.name, f.lineno) for f in fr[-3:]), tuple(weight.shape), mode)] += 1
    return orig(cls, weight, bias, mode, prec, **kw)
T.DevPack.packed = classmethod(spy)
it()
for k, v in cnt.most_common(): print(v, k)
