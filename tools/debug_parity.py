"""Debug helper (GPU box): per-tensor deviation of the HIP forward from the oracle, incl. index flips."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointfusion_amd.model.model import KPFusion
from keypointfusion_amd.weights import synthetic_batch, synthetic_state_dict
from oracle import kpf_oracle as O

net = sys.argv[1] if len(sys.argv) > 1 else "KPFusion-convnext-tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(net, 0).items()}
m = KPFusion(net, "", 21, "dexycb", ""); m.load_state_dict(sd); m = m.to(dev).eval()
b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, 128, seed=seed).items()}
aux = {}
ref, rsw = O.kpfusion_forward(sd, b["img_rgb"], b["img"], b["pcl"], b["center"], b["M"], b["cube"], b["cam_para"], 0.8, aux=aux)
plan = m._plan(dev)
with torch.no_grad():
    out, sws, ctx = plan.forward(*[b[k].to(dev) for k in ("img_rgb", "img", "pcl", "center", "M", "cube", "cam_para")], 0.8, 128, 1, want_aux=True)
rel = lambda a, r: float((a.cpu().float() - r).abs().max() / (r.abs().max() + 1e-12))
for n, o, r in zip(["off_d", "off_rgb", "r3d1", "r2d1", "r3d2", "r2d2"], out, ref):
    print("%-8s rel %.2e  abs %.2e" % (n, rel(o, r), float((o.cpu() - r).abs().max())))
for i in range(2):
    print("sw%d rel %.2e" % (i, rel(sws[i], rsw[i])))
print("joint_uvd", rel(ctx["joint_uvd"], aux["joint_uvd"]), "joint_xyz0", rel(ctx["joint_xyz0"], aux["joint_xyz0"]))
print("top4 idx mismatches", int((ctx["index"].cpu().long() != aux["pcl_index"]).sum()))
for i in range(2):
    a, g = aux["block%d" % (i + 1)], ctx["aux"][i]
    print("block", i + 1, "X", rel(g["X"], a["pcl_feat"]), "D", rel(g["D"], a["joint_feat_desa"]), "h_init", rel(g["h_init"], a["h_init"]),
          "fj", "dec", rel(g["dec"], a["dec"]))
    for r in range(3):
        gi = g["ball_idx"][r].cpu().long().view(B, 21, 64)
        print("   ball r%d mismatches %d" % (r, int((gi != a["ball_idx"][r]).sum())))
