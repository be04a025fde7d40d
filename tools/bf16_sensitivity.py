"""Where does the 16-bit path's deviation come from?  (per-layer-group sensitivity.)

CPU analysis tool, not product code: the fp32 oracle (oracle/kpf_oracle.py) is run with 16-bit STORAGE emulated per layer group — the input, the
weight and the output of every convolution / Linear of a group are rounded to the storage type (fp32 accumulation in between, which is what the
HIP kernels do), block outputs (the residual stream) are rounded too — and the deviation of the four joint estimates from the all-fp32 oracle is
measured in millimetres (cube 250 mm), on the synthetic weights and crops the tolerance test uses.  Sweeps:
  all16            every group in 16-bit storage (the shipped bf16 / f16 modes; should land near DESIGN 4.3's measured table)
  only <g>         ONLY group g in 16 bits, everything else fp32      -> what each group contributes alone
  all-but <g>      everything in 16 bits except group g               -> what fp32 in that group would buy
  fp32-stream      16-bit GEMM operands, fp32 residual stream (block outputs / skip sums not rounded): the usual mixed-precision assignment
usage:  python tools/bf16_sensitivity.py [bf16|f16] [B]      (from the repo root; ~3 min on 8 cores)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from keypointfusion_amd.weights import synthetic_batch, synthetic_state_dict  # noqa: E402
from oracle import kpf_oracle as O  # noqa: E402

GROUPS = ["enc.stage0", "enc.stage1", "enc.stage2", "enc.stage3", "dec.level4", "dec.level3", "dec.level2", "dec.result_emb", "heads"]


def group_of(name):
    if ".backbone." in name:
        for i in range(4):
            if ".stages.%d." % i in name or ".downsample_layers.%d." % i in name:
                return "enc.stage%d" % i
    for lvl in (4, 3, 2):
        if ".up%d." % lvl in name or ".skip_layer%d." % lvl in name or ".fusion_layer%d." % lvl in name:
            return "dec.level%d" % lvl
    if ".result_emb." in name:
        return "dec.result_emb"
    if ".finals." in name:
        return "heads"
    return None  # fusion head: fp32 in every mode


class Emu:
    def __init__(self, sd, tdt):
        self.names = {id(v): k for k, v in sd.items()}
        self.tdt = tdt
        self.low = set()          # groups in 16-bit storage
        self.stream16 = True      # round block outputs (the residual stream) as well
        self.cur = None
        self._conv, self._lin = F.conv2d, F.linear

    def r(self, t):
        return t.to(self.tdt).float()

    def conv2d(self, x, w, b=None, **kw):
        g = group_of(self.names.get(id(w), ""))
        self.cur = g
        if g in self.low:
            return self.r(self._conv(self.r(x), self.r(w), b, **kw))
        return self._conv(x, w, b, **kw)

    def linear(self, x, w, b=None):
        g = group_of(self.names.get(id(w), ""))
        self.cur = g
        if g in self.low:
            return self.r(self._lin(self.r(x), self.r(w), b))
        return self._lin(x, w, b)


def run(sd, b, emu):
    orig = (F.conv2d, F.linear, O.convnext_block, O.residual)
    cb, rs = O.convnext_block, O.residual

    def block(sd_, p, x):
        y = cb(sd_, p, x)
        return emu.r(y) if (emu.stream16 and group_of(p + ".") in emu.low) else y

    def resid(sd_, p, x):
        y = rs(sd_, p, x)
        return emu.r(y) if (emu.stream16 and group_of(p + ".") in emu.low) else y

    F.conv2d, F.linear, O.convnext_block, O.residual = emu.conv2d, emu.linear, block, resid
    try:
        res, _ = O.kpfusion_forward(sd, b["img_rgb"], b["img"], b["pcl"], b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
    finally:
        F.conv2d, F.linear, O.convnext_block, O.residual = orig
    return res


def dev_mm(res, ref):
    mean = [float((res[k] - ref[k]).norm(dim=-1).mean()) * 125.0 for k in range(2, 6)]
    mx = [float((res[k] - ref[k]).norm(dim=-1).max()) * 125.0 for k in range(2, 6)]
    return mean, mx


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    tdt = torch.bfloat16 if prec == "bf16" else torch.float16
    net = "KPFusion-convnext-tiny"
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(net, 0).items()}
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, 128, seed=3).items()}
    emu = Emu(sd, tdt)
    ref = run(sd, b, emu)  # nothing low: the fp32 oracle
    fmt = lambda v: " / ".join("%.3f" % x for x in v)

    def show(label):
        mean, mx = dev_mm(run(sd, b, emu), ref)
        print("%-28s mean %s   max %s" % (label, fmt(mean), fmt(mx)), flush=True)

    print("%s storage, %s, B = %d: joint deviation from the fp32 oracle per stage (r3d1 / r2d1 / r3d2 / r2d2), mm" % (prec, net, B))
    emu.low, emu.stream16 = set(GROUPS), True
    show("all16")
    emu.stream16 = False
    show("all16, fp32 residual stream")
    emu.stream16 = True
    for g in GROUPS:
        emu.low = {g}
        show("only " + g)
    for g in GROUPS:
        emu.low = set(GROUPS) - {g}
        show("all-but " + g)
    emu.low = set(GROUPS) - {"dec.level2", "dec.result_emb", "heads"}
    show("all-but dec.level2+emb+heads")
    emu.low = {"enc.stage0", "enc.stage1", "enc.stage2", "enc.stage3"}
    show("encoder only")


if __name__ == "__main__":
    main()
