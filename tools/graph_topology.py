"""Capture one training iteration and dump the hipGraph's topology (hipGraphDebugDotPrint): node kinds, in / out degrees (a single
chain?), and the memset / memcpy nodes with their neighbours -> gpurun_out/graph_nodes.txt.  Used to pin the replay defect of
DESIGN.md §4.5 on the runtime (5244 nodes in one chain: 5120 kernels, 112 D2D copies, 12 eight-byte memsets)."""
import os, sys, re, torch, collections
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import synthetic_sd
from keypointfusion_amd import training as T
from keypointfusion_amd.model.model import KPFusion
from keypointfusion_amd.parallel import live_parameters
from keypointfusion_amd.weights import synthetic_batch
net = "KPFusion-convnext-tiny"; B = 4; dev = torch.device("cuda:0")
sd = synthetic_sd(net)
batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=5).items()}
g = torch.Generator().manual_seed(1)
batch["uvd_gt"] = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)
batch["xyz_gt"] = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)
class Loader: img_size, flip = 128, 1
def loss_fn(mdl, bt):
    results, sws, _ = mdl(bt["img_rgb"], bt["img"], bt["pcl"], Loader(), bt["center"], bt["M"], bt["cube"], bt["cam_para"], 0.8)
    return T.kpfusion_loss(results, sws, bt["img"], bt["uvd_gt"], bt["xyz_gt"], epoch=0)[0]
torch.manual_seed(0)
m = KPFusion(net, "", 21, "dexycb", ""); m.load_state_dict(sd, strict=True); m = m.to(dev).train(); m.train_dropout = 0.0
live = live_parameters(m)
opt = torch.optim.SGD(live, lr=0.0)
orig = torch.cuda.CUDAGraph
class G(orig):
    def __new__(cls, *a, **k):
        return super().__new__(cls, keep_graph=False)
step = T.GraphedTrainStep.__new__(T.GraphedTrainStep)
# replicate __init__ with debug mode on
import gc
step.model, step.opt, step.loss_fn, step.dist, step.group = m, opt, loss_fn, None, None
step.static = {k: v.detach().clone() for k, v in batch.items()}
step.params = live
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step._forward_backward(); opt.step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
opt.zero_grad(set_to_none=True)
step.graph = torch.cuda.CUDAGraph(keep_graph=True); step.graph_b = None
step.graph.enable_debug_mode()
gc.collect(); gc.disable()
with torch.cuda.graph(step.graph):
    step.loss = step._forward_backward(); opt.step()
gc.enable()
os.makedirs("gpurun_out", exist_ok=True)
step.graph.debug_dump("/tmp/graph.dot"); print(os.path.exists("/tmp/graph.dot"), os.listdir("/tmp")[:20])
txt = open("/tmp/graph.dot").read()
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
outd, ind = collections.Counter(a for a, b in edges), collections.Counter(b for a, b in edges)
nodes = set(a for a, b in edges) | set(b for a, b in edges)
print("nodes", len(nodes), "edges", len(edges), "max out-degree", max(outd.values()), "max in-degree", max(ind.values()))
print("nodes with out-degree > 1:", sum(1 for v in outd.values() if v > 1), "in-degree > 1:", sum(1 for v in ind.values() if v > 1))
print("roots", sum(1 for n in nodes if ind[n] == 0), "leaves", sum(1 for n in nodes if outd[n] == 0))
kinds = collections.Counter(re.findall(r'label="[^"]*?(MEMSET|MEMCPY|KERNEL|EMPTY|HOST|EVENT)', txt))
print("kinds", kinds)
blocks = re.split(r'\n(?="graph_0_node_)', txt)
short = []
for b in blocks:
    mm = re.match(r'"graph_0_node_(\d+)"', b)
    if not mm:
        continue
    kind = re.search(r'label="\{\s*(\w+)', b)
    kind = kind.group(1) if kind else "?"
    if kind == "KERNEL":
        nm = re.search(r"\| \{ID \| \d+ \| (\S+?)\\<", b)
        short.append((int(mm.group(1)), "K", (nm.group(1) if nm else "?")[:90]))
    else:
        short.append((int(mm.group(1)), kind, " ".join(b.split())[:700]))
short.sort()
with open("gpurun_out/graph_nodes.txt", "w") as f:
    for i, (n, k, d) in enumerate(short):
        if k != "K":
            for j in range(max(0, i - 2), min(len(short), i + 3)):
                f.write("%s%d %s %s\n" % ("  >> " if j == i else "     ", short[j][0], short[j][1], short[j][2]))
            f.write("\n")

