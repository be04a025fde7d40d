import os, sys, torch
sys.path.insert(0, os.getcwd())
dev = torch.device("cuda:0")
torch.manual_seed(0)
def run(shape, tag):
    g = torch.nn.Parameter(torch.randn(shape[-1], device=dev))
    y = torch.randn(*shape, device=dev)
    x = torch.randn(*shape, device=dev)
    w = torch.randn(*shape, device=dev)
    def it():
        g.grad = None
        out = x + g * y
        (out * w).sum().backward()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        it()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    ref = g.grad.clone()
    g.grad = None
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        it()
    outs = []
    for _ in range(4):
        gr.replay(); torch.cuda.synchronize(); outs.append(g.grad.clone())
    print(tag, shape, [float((o - ref).abs().max() / ref.abs().max()) for o in outs])
for shape in [(4, 16, 16, 192), (4, 32, 32, 96), (4, 8, 8, 384), (32, 16, 16, 192), (4, 4, 4, 768)]:
    run(shape, "mul-backward under capture")
