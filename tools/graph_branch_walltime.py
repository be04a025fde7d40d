"""Wall time of a captured graph with two independent chains of N small kernels: both on one stream vs forked onto a side stream, and
packet capture on / off (run once per DEBUG_CLR_GRAPH_PACKET_CAPTURE value).  No profiler."""
import os, sys, time, torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
numel = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
dev = torch.device("cuda:0")
a, b = torch.ones(numel, device=dev), torch.ones(numel, device=dev)
side = torch.cuda.Stream()
def chain(x):
    for _ in range(N):
        x.mul_(1.0001)
def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        chain(b)
    chain(a)
    main.wait_stream(side)
def serial():
    chain(b); chain(a)
for name, body in (("serial", serial), ("forked", forked)):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: PACKET_CAPTURE=%s  %d+%d kernels of %d elements: %.3f ms per replay (host issue %.3f ms) = %.2f us per kernel" % (
        name, os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "default"), N, N, numel, (t2 - t0) / 10 * 1e3, (t1 - t0) / 10 * 1e3, (t2 - t0) / 10 / (2 * N) * 1e6))
