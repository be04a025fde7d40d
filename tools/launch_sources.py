"""Where do the small ATen launches of one training iteration come from?  One eager iteration under torch.profiler (CPU side, with Python
stacks); ops grouped by the innermost frame inside keypointfusion_amd / the test harness."""
import os, sys, collections, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import synthetic_sd
from keypointfusion_amd import training as T
from keypointfusion_amd.model.model import KPFusion
from keypointfusion_amd.weights import synthetic_batch
net = "KPFusion-convnext-tiny"; B = 8; dev = torch.device("cuda:0")
sd = synthetic_sd(net)
batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=5).items()}
g = torch.Generator().manual_seed(1)
uvd = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev); xyz = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)
class Loader: img_size, flip = 128, 1
m = KPFusion(net, "", 21, "dexycb", ""); m.load_state_dict(sd, strict=True); m = m.to(dev).train(); m.train_dropout = 0.1
m.precision = sys.argv[1] if len(sys.argv) > 1 else "f32"
def it():
    for p in m.parameters(): p.grad = None
    r, s, _ = m(batch["img_rgb"], batch["img"], batch["pcl"], Loader(), batch["center"], batch["M"], batch["cube"], batch["cam_para"], 0.8)
    T.kpfusion_loss(r, s, batch["img"], uvd, xyz, epoch=0)[0].backward()
it(); torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
SKIP = {"aten::view", "aten::_unsafe_view", "aten::permute", "aten::transpose", "aten::reshape", "aten::detach", "aten::slice", "aten::expand", "aten::t",
        "aten::unsqueeze", "aten::squeeze", "aten::alias", "aten::as_strided", "aten::select", "aten::empty", "aten::empty_like", "aten::empty_strided",
        "aten::narrow", "aten::unbind", "aten::split", "aten::view_as", "aten::_reshape_alias", "aten::size", "aten::stride", "aten::is_contiguous", "aten::unfold"}
cnt = collections.Counter()
COPIES = len(sys.argv) > 2 and sys.argv[2] == "copies"
ANOMALY = len(sys.argv) > 2 and sys.argv[2] == "anomaly"
import re
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if name not in SKIP:
            node = torch._C._current_autograd_node()
            if node is not None:
                where = "BWD " + node.name()
                if ANOMALY:  # where the forward created this node (anomaly mode keeps the forward traceback in the node's metadata)
                    tb = node.metadata.get("traceback_", [])
                    fr = [l.strip().split("\n")[0] for l in tb if "keypointfusion_amd" in l]
                    if fr:
                        m = re.search(r'File ".*?([\w.]+)", line (\d+), in (\w+)', fr[-1])
                        where += "  <- fwd " + ("%s:%s %s" % m.groups() if m else fr[-1][-80:])
                shp = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
                where += "  " + str(shp)
            else:
                where = "?"
                for fr in reversed(traceback.extract_stack(limit=18)):
                    if "keypointfusion_amd" in fr.filename:
                        where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                        break
            if COPIES and name in ("aten::clone", "aten::_to_copy", "aten::copy_", "aten::contiguous"):
                src = args[1] if name == "aten::copy_" else args[0]
                if torch.is_tensor(src) and (not src.is_contiguous() or (name == "aten::copy_" and not args[0].is_contiguous())):
                    fr = [f for f in traceback.extract_stack(limit=18) if "keypointfusion_amd" in f.filename]
                    loc = "%s:%d" % (os.path.basename(fr[-1].filename), fr[-1].lineno) if fr else "?"
                    cnt[("STRIDED " + name, "%s %s %s numel %d" % (where, loc, tuple(src.shape), src.numel()))] += 1
            cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))
if ANOMALY:
    with torch.autograd.detect_anomaly(check_nan=False), Mode():
        it()
else:
    with Mode():
        it()
torch.cuda.synchronize()
print("ATen ops (non-view) in one iteration:", sum(cnt.values()))
for (name, where), c in cnt.most_common(400):
    print("%5d  %-30s %s" % (c, name, where))
