"""Where does the one-launch operand refresh (pack_weights_multi_kernel) spend its time?  After one training iteration has registered every operand,
the table is rebuilt from subsets of the descriptors (by pack mode / filter size) and each subset's launch is timed.  usage: python tools/pack_breakdown.py [bf16|f32]"""
import os, sys, time, torch
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
m = KPFusion(net, "", 21, "dexycb", ""); m.load_state_dict(synthetic_sd(net), strict=True); m = m.to(dev).train()
m.precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
r, s, _ = m(batch["img_rgb"], batch["img"], batch["pcl"], Loader(), batch["center"], batch["M"], batch["cube"], batch["cam_para"], 0.8)
T.kpfusion_loss(r, s, batch["img"], uvd, xyz, epoch=0)[0].backward()
cache = list(m.__dict__["_pack_cache"].values())[0]
ents = dict(cache.entries)
def timed(sub):
    cache.entries = sub; cache.dirty = True; cache.build_table()
    for _ in range(3): cache.refresh()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): cache.refresh()
    torch.cuda.synchronize()
    el = sum(e["desc"][8] * e["desc"][9] for e in sub.values())
    return (time.perf_counter() - t) / 20 * 1e6, len(sub), el
kinds = {"all": lambda d: True, "mode0 1x1": lambda d: d[6] == 0 and d[4] * d[5] == 1, "mode0 kxk": lambda d: d[6] == 0 and d[4] * d[5] > 1,
         "mode1 1x1": lambda d: d[6] == 1 and d[4] * d[5] == 1, "mode1 kxk": lambda d: d[6] == 1 and d[4] * d[5] > 1, "mode2/3": lambda d: d[6] in (2, 3), "mode4": lambda d: d[6] == 4}
for name, f in kinds.items():
    sub = {k: e for k, e in ents.items() if f(e["desc"])}
    if sub:
        us, n, el = timed(sub)
        print("%-10s %4d operands %8.2f M elements  %7.1f us" % (name, n, el / 1e6, us))
