import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
from conftest import synthetic_sd
from keypointfusion_amd.model.model import KPFusion
from keypointfusion_amd.serving import PipelinedEval
from keypointfusion_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
net = "KPFusion-convnext-tiny"
m = KPFusion(net, "", 21, "dexycb", ""); m.load_state_dict(synthetic_sd(net), strict=True); m = m.to(dev).eval(); m.precision = "bf16"
b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(32, 128, seed=3).items()}
class Loader: img_size, flip = 128, 1
a = (b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
def run(pe, collect_lag, N=40):
    pend = []
    for _ in range(6):
        pend.append(pe.submit(*a))
    while pend: pe.collect(pend.pop(0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        pend.append(pe.submit(*a))
        if len(pend) > collect_lag: pe.collect(pend.pop(0))
    ti = time.perf_counter() - t0
    while pend: pe.collect(pend.pop(0))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3, ti / N * 1e3
loop = torch.cuda.Stream() if os.environ.get("LOOP_STREAM") else torch.cuda.current_stream()
loop.wait_stream(torch.cuda.current_stream())
with torch.no_grad(), torch.cuda.stream(loop):
    for stages in (True, False):
        pe = PipelinedEval(m, depth=2, stages=stages)
        for lag in (2, 3, 8, 1000):
            print("stages=%s collect lag %4d: %.3f ms per batch (host issue %.3f)" % (stages, lag, *run(pe, lag)), flush=True)
