"""Would a batch-stacked grouped backbone pair pay at configs[2]'s size?  One ConvNeXt-T UNet stream at B = 64 (the tile counts a grouped pair at B = 32 would have)
against the two streams at B = 32 (two HIP streams, as ModelPlan.backbones runs them), both as hipGraph replays, bf16, 128 x 128."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
from conftest import synthetic_sd
from keypointfusion_amd.engine import ModelPlan
dev = torch.device("cuda:0")
net = "KPFusion-convnext-tiny"
sd = synthetic_sd(net)
plan = ModelPlan(sd, net, dev, precision="bf16")
g = torch.Generator().manual_seed(0)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def graphed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = fn()
    return gr
for B in (32, 64):
    img = torch.randn(B, 1, 128, 128, generator=g).to(dev)
    rgb = torch.randn(B, 3, 128, 128, generator=g).to(dev)
    with torch.no_grad():
        g_pair = graphed(lambda: plan.backbones(img, rgb))
        plan.serial_streams = True
        g_ser = graphed(lambda: plan.backbones(img, rgb))
        plan.serial_streams = False
        g_d = graphed(lambda: plan.backbone_d(img))
        g_rgb = graphed(lambda: plan.backbone_rgb(rgb))
    print("B=%d: pair on two streams %.3f ms | pair on one stream %.3f ms | depth stream alone %.3f ms | rgb stream alone %.3f ms" % (
        B, timeit(g_pair.replay), timeit(g_ser.replay), timeit(g_d.replay), timeit(g_rgb.replay)), flush=True)
