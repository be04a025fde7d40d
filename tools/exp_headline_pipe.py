"""Headline workload (both backbones, B = 64, 256 x 256, fp32): one graph replay per step on one stream against two slots on two streams fed from a loop stream."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
from conftest import synthetic_sd
from keypointfusion_amd.engine import ModelPlan
dev = torch.device("cuda:0")
net = "KPFusion-convnext-tiny"
plan = ModelPlan(synthetic_sd(net), net, dev, precision="f32")
g = torch.Generator().manual_seed(0)
img = torch.randn(64, 1, 256, 256, generator=g).to(dev); rgb = torch.randn(64, 3, 256, 256, generator=g).to(dev)
N = 30
with torch.no_grad():
    for s in (0, 1): plan.backbones_graphed(img, rgb, slot=s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): plan.backbones_graphed(img, rgb, slot=0)
    torch.cuda.synchronize()
    print("one slot, default stream: %.3f ms per step" % ((time.perf_counter() - t0) / N * 1e3))
    loop = torch.cuda.Stream(); ss = [torch.cuda.Stream(), torch.cuda.Stream()]
    loop.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(loop):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(N):
                s = ss[i % 2]
                s.wait_stream(loop)
                with torch.cuda.stream(s): plan.backbones_graphed(img, rgb, slot=i % 2)
            for s in ss: loop.wait_stream(s)
            torch.cuda.synchronize()
            print("two slots on two streams, loop stream: %.3f ms per step" % ((time.perf_counter() - t0) / N * 1e3))
