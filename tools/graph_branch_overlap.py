"""Do two independent branches of a captured hipGraph overlap at replay on this runtime?  Two chains of N small element-wise kernels, one on
the capture stream and one on a forked side stream; replayed under `rocprofv3 --kernel-trace`, tools/replay_timeline.py-style overlap is
computed by tools/branch_overlap_report.py from the trace.  usage: python3 tools/graph_branch_overlap.py [n_kernels] [numel] [mode]
mode: plain | alloc (each step allocates its output, like a real forward)"""
import os, sys, torch
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
numel = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
dev = torch.device("cuda:0")
a, b = torch.ones(numel, device=dev), torch.ones(numel, device=dev)
side = torch.cuda.Stream()

def chain(x):
    if mode == "alloc":
        for _ in range(N):
            x = x * 1.0001
        return x
    for _ in range(N):
        x.mul_(1.0001)
    return x

def body():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        rb = chain(b)
    ra = chain(a)
    main.wait_stream(side)
    return ra + rb

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
torch.cuda.synchronize()
marker = torch.zeros(7, device=dev)
for _ in range(4):
    marker.add_(1)  # (delimits replays in the trace: a 7-element add)
    g.replay()
torch.cuda.synchronize()
print("ok", float(out[0]))
