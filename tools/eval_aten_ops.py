"""Which ATen ops (each at least one launch or memcpy) does one eval forward of the full model issue besides the library's kernels?
  python tools/eval_aten_ops.py [f32|bf16]"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import synthetic_sd
from keypointfusion_amd.model.model import KPFusion
from keypointfusion_amd.weights import synthetic_batch
from torch.utils._python_dispatch import TorchDispatchMode
net = "KPFusion-convnext-tiny"; B = 32; dev = torch.device("cuda:0")
m = KPFusion(net, "", 21, "dexycb", ""); m.load_state_dict(synthetic_sd(net), strict=True); m = m.to(dev).eval()
m.precision = sys.argv[1] if len(sys.argv) > 1 else "f32"
b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=1).items()}
class Loader: img_size, flip = 128, 1
def fwd():
    with torch.no_grad():
        return m(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
fwd(); torch.cuda.synchronize()
SKIP = {"aten::view", "aten::_unsafe_view", "aten::permute", "aten::transpose", "aten::reshape", "aten::detach", "aten::slice", "aten::expand", "aten::t",
        "aten::unsqueeze", "aten::squeeze", "aten::alias", "aten::as_strided", "aten::select", "aten::empty", "aten::empty_like", "aten::empty_strided",
        "aten::narrow", "aten::unbind", "aten::split", "aten::view_as", "aten::_reshape_alias"}
cnt = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if name not in SKIP:
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=16)):
                if "keypointfusion_amd" in fr.filename:
                    where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    fwd()
print("ATen ops in one eval forward (%s): %d" % (m.precision, sum(cnt.values())))
for (name, where), c in cnt.most_common(40):
    print("%5d  %-26s %s" % (c, name, where))
