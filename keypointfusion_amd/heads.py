"""Execution plans of the stand-alone heads (SURVEY.md §8 a17 CBAM, a18 Hourglass/PoseNet, a19 MANO head) — same conventions as
engine.py: NHWC activations, weights repacked once, every launch goes through the C ABI of include/kpf.h on the current stream."""
import ctypes as C

import torch

from . import lib as L
from .engine import Act, PackedConv, ResidualPlan, bn_scale_shift, conv, nchw_to_nhwc, nhwc_to_nchw, _ptr, _stream


# ----------------------------------------------------------------------------------------------------------------
# a17 CBAM (model/cbam.py:84-94)
# ----------------------------------------------------------------------------------------------------------------
class CbamPlan:
    def __init__(self, sd, device, p=""):
        f = lambda k: sd[p + k].detach().float().contiguous().to(device)  # noqa: E731
        self.w1, self.b1 = f("ChannelGate.mlp.1.weight"), f("ChannelGate.mlp.1.bias")
        self.w2, self.b2 = f("ChannelGate.mlp.3.weight"), f("ChannelGate.mlp.3.bias")
        self.Cr, self.C = self.w1.shape
        self.spatial = (p + "SpatialGate.spatial.conv.weight") in sd
        if self.spatial:
            self.w7 = f("SpatialGate.spatial.conv.weight").reshape(98)
            s, t = bn_scale_shift(sd, p + "SpatialGate.spatial.bn")
            self.bn_s, self.bn_t = float(s.item()), float(t.item())

    def channel_scale(self, x):
        """x: dense Act -> scale tensor [B, C]."""
        lib = L.load()
        assert x.ld == x.C and x.coff == 0 and x.C == self.C
        dev = x.buf.device
        HW = x.H * x.W
        ws = torch.empty(int(lib.kpf_cbam_workspace_floats(x.B, HW, x.C)), device=dev, dtype=torch.float32)
        scale = torch.empty(x.B, x.C, device=dev, dtype=torch.float32)
        L.check(lib.kpf_cbam_channel_gate_f32(_ptr(x.buf), _ptr(self.w1), _ptr(self.b1), _ptr(self.w2), _ptr(self.b2), _ptr(ws), _ptr(scale),
                                              x.B, HW, x.C, self.Cr, _stream()), "kpf_cbam_channel_gate_f32")
        return scale

    def __call__(self, x):
        """x: dense Act -> Act (no_spatial) or (Act, Act)."""
        lib = L.load()
        dev = x.buf.device
        scale = self.channel_scale(x)
        HW = x.H * x.W
        out0 = Act.empty(x.B, x.H, x.W, x.C, dev)
        if not self.spatial:
            L.check(lib.kpf_cbam_apply_f32(_ptr(x.buf), _ptr(scale), None, _ptr(out0.buf), None, x.B, HW, x.C, _stream()), "kpf_cbam_apply_f32")
            return out0
        comp = torch.empty(x.B * HW * 2, device=dev, dtype=torch.float32)
        sg = torch.empty(x.B * HW, device=dev, dtype=torch.float32)
        L.check(lib.kpf_cbam_spatial_gate_f32(_ptr(x.buf), _ptr(scale), _ptr(self.w7), self.bn_s, self.bn_t, _ptr(comp), _ptr(sg), x.B, x.H, x.W,
                                              x.C, _stream()), "kpf_cbam_spatial_gate_f32")
        out1 = Act.empty(x.B, x.H, x.W, x.C, dev)
        L.check(lib.kpf_cbam_apply_f32(_ptr(x.buf), _ptr(scale), _ptr(sg), _ptr(out0.buf), _ptr(out1.buf), x.B, HW, x.C, _stream()),
                "kpf_cbam_apply_f32")
        return out0, out1


# ----------------------------------------------------------------------------------------------------------------
# a18 Hourglass / PoseNet (model/hourglass.py:122-229)
# ----------------------------------------------------------------------------------------------------------------
def maxpool2x2(x):
    lib = L.load()
    assert x.ld == x.C and x.coff == 0 and x.H % 2 == 0 and x.W % 2 == 0, "MaxPool2d(2,2) on odd maps is not needed by the hourglass"
    out = Act.empty(x.B, x.H // 2, x.W // 2, x.C, x.buf.device)
    L.check(lib.kpf_maxpool2x2_f32(_ptr(x.buf), _ptr(out.buf), x.B, x.H, x.W, x.C, _stream()), "kpf_maxpool2x2_f32")
    return out


def upnearest2x_add(low, up1):
    lib = L.load()
    assert (up1.H, up1.W, up1.C) == (2 * low.H, 2 * low.W, low.C) and up1.ld == up1.C and low.ld == low.C
    out = Act.empty(up1.B, up1.H, up1.W, up1.C, up1.buf.device)
    L.check(lib.kpf_upnearest2x_add_f32(_ptr(low.buf), _ptr(up1.buf), _ptr(out.buf), low.B, low.H, low.W, low.C, _stream()),
            "kpf_upnearest2x_add_f32")
    return out


class HourglassPlan:
    def __init__(self, sd, p, n, device):
        self.up1 = ResidualPlan(sd, p + ".up1", device)
        self.low1 = ResidualPlan(sd, p + ".low1", device)
        self.low2 = HourglassPlan(sd, p + ".low2", n - 1, device) if n > 1 else ResidualPlan(sd, p + ".low2", device)
        self.low3 = ResidualPlan(sd, p + ".low3", device)

    def __call__(self, x):
        up1 = self.up1(x)
        low = self.low3(self.low2(self.low1(maxpool2x2(x))))
        return upnearest2x_add(low, up1)


class PoseNetPlan:
    """PoseNet.forward (model/hourglass.py:211-229).  The 7x7/s2 stem has its BatchNorm+ReLU folded; the three 1x1 heads of a stack
    are one N = 5J GEMM (their concatenation is what the reference returns); the inter-stack merge
    x + merge_preds(preds) + merge_features(feature) is two GEMMs with residual epilogues."""

    def __init__(self, sd, nstack, device):
        self.nstack = nstack
        self.stem = PackedConv(sd["pre.0.conv.weight"], sd["pre.0.conv.bias"], device, stride=2, pad=3, fold_bn=bn_scale_shift(sd, "pre.0.bn"),
                               cin_pad=4)
        self.pre1 = ResidualPlan(sd, "pre.1", device)
        self.pre3 = ResidualPlan(sd, "pre.3", device)
        self.pre4 = ResidualPlan(sd, "pre.4", device)
        self.hgs, self.feat_res, self.feat_conv, self.heads, self.merge_p, self.merge_f = [], [], [], [], [], []
        for i in range(nstack):
            self.hgs.append(HourglassPlan(sd, "hgs.%d" % i, 4, device))
            self.feat_res.append(ResidualPlan(sd, "features.%d.0" % i, device))
            self.feat_conv.append(PackedConv(sd["features.%d.1.conv.weight" % i], sd["features.%d.1.conv.bias" % i], device,
                                             fold_bn=bn_scale_shift(sd, "features.%d.1.bn" % i)))
            w = torch.cat([sd["outs_%d.%d.weight" % (k, i)] for k in (1, 2, 3)], 0)
            b = torch.cat([sd["outs_%d.%d.bias" % (k, i)] for k in (1, 2, 3)], 0)
            self.heads.append(PackedConv(w, b, device))
            if i < nstack - 1:
                # preds are kept NHWC with the channel count padded to a multiple of 4 for the merge GEMM
                npred = w.shape[0]
                self.merge_p.append(PackedConv(sd["merge_preds.%d.conv.conv.weight" % i], sd["merge_preds.%d.conv.conv.bias" % i], device,
                                               cin_pad=(npred + 3) // 4 * 4))
                self.merge_f.append(PackedConv(sd["merge_features.%d.conv.conv.weight" % i], sd["merge_features.%d.conv.conv.bias" % i], device))

    def __call__(self, img):
        """img: NCHW [B,1,S,S] tensor -> (preds NCHW tensor [B,5J,S/4,S/4], feature Act)."""
        dev = img.device
        x = nchw_to_nhwc(img, cpad=4)
        x = conv(self.stem, x, flags=L.KPF_ACT_RELU)
        x = self.pre4(self.pre3(maxpool2x2(self.pre1(x))))
        preds = feat = None
        for i in range(self.nstack):
            hg = self.hgs[i](x)
            feat = conv(self.feat_conv[i], self.feat_res[i](hg), flags=L.KPF_ACT_RELU)
            last = i == self.nstack - 1
            if last:
                N = self.heads[i].N
                preds = torch.empty(x.B, N, x.H, x.W, device=dev, dtype=torch.float32)
                conv(self.heads[i], feat, out_nchw=preds)
            else:
                N = self.heads[i].N
                Np = (N + 3) // 4 * 4
                pbuf = Act(torch.zeros(x.B * x.H * x.W * Np, device=dev, dtype=torch.float32), x.B, x.H, x.W, Np)
                conv(self.heads[i], feat, out=pbuf.slice(0, N))
                nx = Act.empty(x.B, x.H, x.W, x.C, dev)
                conv(self.merge_p[i], pbuf, out=nx, res=x)
                conv(self.merge_f[i], feat, out=nx, res=nx)
                x = nx
        return preds, feat


# ----------------------------------------------------------------------------------------------------------------
# a19 MANO regression head (model/mano_head.py:177-225)
# ----------------------------------------------------------------------------------------------------------------
class ManoHeadPlan:
    def __init__(self, sd, device):
        self.base = []
        i = 0
        while "mano_base_layer.%d.weight" % i in sd:
            self.base.append(PackedConv(sd["mano_base_layer.%d.weight" % i], sd["mano_base_layer.%d.bias" % i], device))
            i += 2
        # pose_reg (96) and shape_reg (10) as one GEMM: row = [pose6d 96 | betas 10 | 0 0]
        w = torch.cat([sd["pose_reg.weight"], sd["shape_reg.weight"]], 0)
        b = torch.cat([sd["pose_reg.bias"], sd["shape_reg.bias"]], 0)
        self.reg = PackedConv(w, b, device)
        f = lambda k: sd["mano_layer." + k].detach().float().to(device)  # noqa: E731
        self.shape_t = f("th_shapedirs").reshape(778 * 3, 10).t().contiguous()
        self.pose_t = f("th_posedirs").reshape(778 * 3, 135).t().contiguous()
        self.vtmpl = f("th_v_template").reshape(778 * 3).contiguous()
        self.jreg = f("th_J_regressor").contiguous()
        self.skin = f("th_weights").contiguous()
        self.hands_mean = f("th_hands_mean").reshape(45).contiguous()

    def __call__(self, features):
        lib = L.load()
        B = features.shape[0]
        dev = features.device
        h = Act(features.detach().float().contiguous().view(-1), B, 1, 1, features.shape[1])
        for pc in self.base:
            h = conv(pc, h, flags=L.KPF_ACT_LEAKY)
        reg = Act(torch.zeros(B * 108, device=dev, dtype=torch.float32), B, 1, 1, 108)
        conv(self.reg, h, out=reg.slice(0, 106))
        r = reg.buf.view(B, 108)
        verts = torch.empty(B, 778, 3, device=dev, dtype=torch.float32)
        joints = torch.empty(B, 21, 3, device=dev, dtype=torch.float32)
        rotmat = torch.empty(B, 16, 3, 3, device=dev, dtype=torch.float32)
        aa = torch.empty(B, 48, device=dev, dtype=torch.float32)
        betas = r[:, 96:106]
        L.check(lib.kpf_mano_forward_f32(_ptr(r), 108, C.c_void_p(r.data_ptr() + 96 * 4), 108, _ptr(self.shape_t), _ptr(self.pose_t), _ptr(self.vtmpl),
                                         _ptr(self.jreg), _ptr(self.skin), _ptr(self.hands_mean), _ptr(verts), _ptr(joints), _ptr(rotmat), _ptr(aa),
                                         B, _stream()), "kpf_mano_forward_f32")
        return {"verts3d": verts, "joints3d": joints, "mano_shape": betas.contiguous(), "mano_pose": rotmat, "mano_pose_aa": aa}
