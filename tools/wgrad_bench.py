"""Weight-gradient kernels on the training step's own shapes (B = 32, 128^2 ConvNeXt-T): the 16-bit-MFMA form (kpf_conv2d_wgrad_h16:
wgrad_h16_kernel + reduce), the round-2 widening form (KPF_WGRAD_H16_WIDEN=1: run this script twice) and the fp32 form, timed with HIP
events over 5 replays of a captured graph of 20 calls; algorithmic bytes = dY + X read once + dW written once, FLOP = 2 M N K.
  python3 tools/wgrad_bench.py > profiles/r03_wgrad_bench.txt"""
import os, sys, torch
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.getcwd())
from keypointfusion_amd.training import conv_wgrad_hip
dev = torch.device("cuda:0")
shapes = [  # (rows as B,H,W), Cin, N, k   — pw1 / pw2 of the four ConvNeXt-T stages at 128^2 input, a decoder 3x3, the stem
    ((32, 32, 32), 96, 384, 1), ((32, 32, 32), 384, 96, 1), ((32, 16, 16), 192, 768, 1), ((32, 16, 16), 768, 192, 1),
    ((32, 8, 8), 384, 1536, 1), ((32, 8, 8), 1536, 384, 1), ((32, 4, 4), 768, 3072, 1), ((32, 4, 4), 3072, 768, 1),
    ((32, 32, 32), 64, 64, 3), ((32, 16, 16), 192, 96, 3)]
print("mode: %s" % ("widening (round 2)" if os.environ.get("KPF_WGRAD_H16_WIDEN") == "1" else "16-bit MFMA + transposed LDS reads"))
print("%-28s %10s %10s %10s %10s" % ("M x N x K", "bf16 us", "TFLOP/s", "GB/s", "fp32 us"))
for (B, H, W), cin, n, k in shapes:
    res = {}
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(B, H, W, cin, device=dev).to(dt)
        dy = torch.randn(B, H, W, n, device=dev).to(dt)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                conv_wgrad_hip(dy, x, (n, cin, k, k), 1, k // 2, True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()  # 20 calls per replay: the host side of the wrapper (~20 us per call) stays out of the measurement
        with torch.cuda.graph(g):
            for _ in range(20):
                conv_wgrad_hip(dy, x, (n, cin, k, k), 1, k // 2, True)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        res[dt] = e0.elapsed_time(e1) / 100 * 1e3
    M, K = B * H * W, cin * k * k
    fl = 2.0 * M * n * K
    by = (M * n + M * cin) * 2 + n * K * 4
    t = res[torch.bfloat16]
    print("%-28s %10.1f %10.1f %10.0f %10.1f" % ("%d x %d x %d" % (M, n, K), t, fl / t / 1e6, by / t / 1e3, res[torch.float32]))
print("(per call: GEMM launch + reduce launch, replayed from a graph; bf16 MFMA peak 2500 TFLOP/s, HBM 8000 GB/s: these layers are "
      "launch- and latency-bound at B = 32 — 0.1-2.4 GFLOP each)")
