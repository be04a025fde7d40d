"""Reduced-precision execution plan of the UNet backbones (BASELINE.json configs[2] bf16 — ConvNeXt and ResNet —, configs[4] fp16): 16-bit NHWC
activations and weights in HBM, fp32 accumulation on v_mfma_f32_16x16x32_{bf16,f16}, fp32 elementwise arithmetic (LayerNorm statistics,
GELU, residual adds, depthwise taps) inside the kernels.  Same schedule as engine.UNetPlan; the 4x4 stem convolution stays on the fp32
path (its input is the fp32 image, K = 16 / 48), the 105-channel head output is written fp32 NCHW, and the 128-channel feature map is
cast to fp32 for the fusion head, which stays fp32 (its integer decisions must not move).
The reference has no reduced-precision mode: this is the arithmetic a `torch.autocast`-style deployment of it would use,
and its contract is a tolerance in millimetres against the fp32 oracle (tests/test_reduced_precision_gpu.py), not bit parity.
"""
import ctypes as C

import torch

from . import lib as L
from .engine import Act, PackedConv, _launch, _ptr, _stream, bn_scale_shift, conv, maxpool3x3s2, nchw_to_nhwc
from .spec import CONVNEXT, parse_net

DTYPES = {"bf16": (torch.bfloat16, L.KPF_DT_BF16), "f16": (torch.float16, L.KPF_DT_F16)}
FORCE_UNFUSED_MLP16 = bool(int(__import__("os").environ.get("KPF_UNFUSED_MLP16", "0")))  # A/B switch for tuning
DW_STATS = bool(int(__import__("os").environ.get("KPF_DW_STATS", "1")))  # depthwise stencil + statistics pair instead of the one-pass dw+LN kernels
LN_FOLD = bool(int(__import__("os").environ.get("KPF_LN_FOLD", "1")))  # LayerNorm folded into pwconv1's epilogue where the eight-phase GEMM takes the layer (KPF_PRO_LN)
DW_STATS_MIN_C = int(__import__("os").environ.get("KPF_DW_STATS_MIN_C", "256"))  # (at C = 128 the one-pass wave kernel is still faster: 311 vs 363 us)


# experiment (tools/exp_rules16.sh): per-shape tile overrides "M:N:K:kh=case,..."
_TILE_RULES16 = {tuple(int(v) for v in r.split("=")[0].split(":")): int(r.split("=")[1]) + 1 for r in __import__("os").environ.get("KPF_TILE_RULES16", "").split(",") if r}
FORCE_TILE16 = 0  # tuning aid (tools/h16_small_sweep.py): tile case + 1 for every kpf_conv2d_h16 launch, 0 = the library's choice


def PROFILE_LABELS():
    from . import engine as _E
    return _E.PROFILE is not None  # (the label lookup is a library call: only when bench.py is collecting per-launch events)


def empty16(B, H, W, Cc, device, tdt):
    a = Act(torch.empty(B * H * W * Cc, device=device, dtype=tdt), B, H, W, Cc)
    return a


class Packed16:
    """16-bit image of a PackedConv: rows [N][Kp], Kp padded to 64 elements, in the storage dtype; bias stays fp32."""

    def __init__(self, pc, tdt):
        self.pc = pc
        self.Kp = (pc.K + 63) // 64 * 64
        w = torch.zeros(pc.N, self.Kp, dtype=torch.float32, device=pc.w.device)
        w[:, :pc.K] = pc.w[:, :pc.K]
        self.w = w.to(tdt).contiguous()


def conv16(p16, x, kdt, out=None, flags=0, gamma=None, res=None, out_nchw=None, out2=None, ln=None, probe=False):
    """kpf_conv2d_h16: x / out / res are 16-bit Acts (out_nchw: fp32 NCHW tensor).  ln = (mean_rstd [pixels][2], s [N], b' [N]): KPF_PRO_LN (p16 holds
    W diag(ln_w)).  probe: no launch — returns whether the library takes this layer with KPF_PRO_LN (a rule over the layer's shape, not its batch)."""
    lib = L.load()
    pc = p16.pc
    B = x.B
    if pc.merge > 1:
        assert x.ld == x.C and x.coff == 0 and x.W % pc.merge == 0
        IH, IW, in_ld, in_coff = x.H, x.W // pc.merge, x.C * pc.merge, 0
        assert in_ld == pc.Cin, (in_ld, pc.Cin)
    else:
        IH, IW, in_ld, in_coff = x.H, x.W, x.ld, x.coff
        assert x.C == pc.Cin, (x.C, pc.Cin)
    OH = (IH + 2 * pc.ph - pc.KH) // pc.sh + 1
    OW = (IW + 2 * pc.pw - pc.KW) // pc.sw + 1
    d = L.ConvDesc()
    d.B, d.IH, d.IW, d.Cin, d.in_ld, d.in_coff = B, IH, IW, pc.Cin, in_ld, in_coff
    d.OH, d.OW, d.N = OH, OW, pc.N
    d.KH, d.KW, d.sh, d.sw, d.ph, d.pw, d.Kp = pc.KH, pc.KW, pc.sh, pc.sw, pc.ph, pc.pw, p16.Kp
    if out_nchw is not None:
        flags |= L.KPF_OUT_NCHW
        optr = out_nchw
        d.out_ld, d.out_coff = pc.N, 0
    else:
        if out is None:
            out = empty16(B, OH, OW, pc.N, x.buf.device, x.buf.dtype)
        assert (out.B, out.H, out.W, out.C) == (B, OH, OW, pc.N)
        optr = out.buf
        d.out_ld, d.out_coff = out.ld, out.coff
    if res is not None:
        flags |= L.KPF_RES_ADD
        d.res_ld, d.res_coff = res.ld, res.coff
    if out2 is not None:  # KPF_ACT_GELU_SAVE: the pre-activation goes to a second buffer, passed in the residual's slot (nothing is read from it)
        assert res is None and (flags & L.KPF_ACT_GELU)
        flags |= L.KPF_ACT_GELU_SAVE
        d.res_ld, d.res_coff = out2.ld, out2.coff
        res = out2
    if gamma is not None:
        flags |= L.KPF_RES_GAMMA
    ps, pt, bias = pc.ps, pc.pt, pc.b
    if ln is not None:
        flags |= L.KPF_PRO_LN
        ps, pt, bias = ln
    d.flags = flags
    d.groups, d.w_gstride = getattr(pc, "groups", 0), getattr(pc, "w_gstride", 0)  # (grouped launch: training.GroupedPack; 0 = one convolution)
    if _TILE_RULES16:
        d.tile_cfg = _TILE_RULES16.get((B * OH * OW, pc.N, pc.K, pc.KH), 0)
    if FORCE_TILE16:
        d.tile_cfg = FORCE_TILE16
    if probe:
        return bool(lib.kpf_conv2d_h16_ln_fold_supported(C.byref(d)))
    M = B * OH * OW
    nbytes = 2.0 * (B * IH * IW * pc.Cin + pc.N * pc.K + M * pc.N * (2 if res is not None else 1))
    name = "gemm16_8ph_kernel" if (PROFILE_LABELS() and (ln is not None or lib.kpf_conv2d_h16_uses_8ph(C.byref(d), 1 if pc.ps is not None else 0))) else "igemm_h16_kernel"
    ng = max(1, d.groups)
    _launch(name, pc.flops(M) * ng, nbytes * ng, (M, pc.N, pc.K, pc.KH, pc.KW),
            lambda: L.check(lib.kpf_conv2d_h16(C.byref(d), _ptr(x.buf), _ptr(p16.w), _ptr(bias), _ptr(ps), _ptr(pt), _ptr(gamma),
                                               _ptr(res.buf if res is not None else None), _ptr(optr), kdt, _stream()), "kpf_conv2d_h16"))
    return out


class Residual16:
    """engine.ResidualPlan on 16-bit storage."""

    def __init__(self, sd, p, device, tdt):
        cin = sd[p + ".conv1.conv.weight"].shape[1]
        self.cout = sd[p + ".conv3.conv.weight"].shape[0]
        P = lambda *a, **k: Packed16(PackedConv(*a, **k), tdt)
        self.c1 = P(sd[p + ".conv1.conv.weight"], sd[p + ".conv1.conv.bias"], device, fold_bn=bn_scale_shift(sd, p + ".bn2"),
                    prologue=bn_scale_shift(sd, p + ".bn1"))
        self.c2 = P(sd[p + ".conv2.conv.weight"], sd[p + ".conv2.conv.bias"], device, pad=1, fold_bn=bn_scale_shift(sd, p + ".bn3"))
        self.c3 = P(sd[p + ".conv3.conv.weight"], sd[p + ".conv3.conv.bias"], device)
        self.skip = P(sd[p + ".skip_layer.conv.weight"], sd[p + ".skip_layer.conv.bias"], device) if cin != self.cout else None

    def __call__(self, x, kdt, out=None):
        h = conv16(self.c1, x, kdt, flags=L.KPF_ACT_RELU)
        h = conv16(self.c2, h, kdt, flags=L.KPF_ACT_RELU)
        if out is None:
            out = empty16(x.B, x.H, x.W, self.cout, x.buf.device, x.buf.dtype)
        if self.skip is not None:
            conv16(self.skip, x, kdt, out=out)
            return conv16(self.c3, h, kdt, out=out, res=out)
        return conv16(self.c3, h, kdt, out=out, res=x)


def _ln_fold_shape_ok(c):
    """kpf_conv2d_h16_ln_fold_supported for pwconv1 of a C-channel ConvNeXt block (any batch, any map: the rule does not look at them)."""
    d = L.ConvDesc()
    d.B, d.IH, d.IW, d.Cin, d.in_ld, d.in_coff = 1, 16, 16, c, c, 0
    d.OH, d.OW, d.N = 16, 16, 4 * c
    d.KH = d.KW = d.sh = d.sw = 1
    d.Kp = (c + 63) // 64 * 64
    d.out_ld, d.out_coff, d.flags = 4 * c, 0, L.KPF_ACT_GELU
    return bool(L.load().kpf_conv2d_h16_ln_fold_supported(C.byref(d)))


class Block16:
    """engine.ConvNeXtBlockPlan on 16-bit storage: dw7x7+LN kernel, pwconv1 GEMM (+GELU), pwconv2 GEMM (+gamma*y + x, in place)."""

    def __init__(self, sd, p, device, tdt):
        c = sd[p + ".gamma"].numel()
        f32 = lambda k: sd[p + k].detach().float().contiguous().to(device)
        self.wdw = sd[p + ".dwconv.weight"].detach().float().reshape(c, 49).t().contiguous().to(device)
        self.bdw, self.lnw, self.lnb, self.gamma = f32(".dwconv.bias"), f32(".norm.weight"), f32(".norm.bias"), f32(".gamma")
        self.pw1 = Packed16(PackedConv(sd[p + ".pwconv1.weight"], sd[p + ".pwconv1.bias"], device), tdt)
        self.pw2 = Packed16(PackedConv(sd[p + ".pwconv2.weight"], sd[p + ".pwconv2.bias"], device), tdt)
        # LayerNorm folded into pwconv1 (KPF_PRO_LN; used when the statistics come from the stencil and gemm16_8ph_kernel takes the layer):
        #   W (ln_w * xhat + ln_b) + b = rstd * (W' x - mean * s) + b',   W' = W diag(ln_w) rounded to the storage type, s = its row sums in fp32, b' = W ln_b + b
        self.pw1f = None
        if LN_FOLD and _ln_fold_shape_ok(c):  # (the rule is over the layer's shape: known here; blocks it does not cover keep one copy of pwconv1)
            w1 = sd[p + ".pwconv1.weight"].detach().double().reshape(4 * c, c)
            lnw, lnb = sd[p + ".norm.weight"].detach().double(), sd[p + ".norm.bias"].detach().double()
            self.pw1f = Packed16(PackedConv((w1 * lnw[None, :]).float().reshape(4 * c, c, 1, 1), None, device), tdt)
            self.pw1f_s = self.pw1f.w[:, :c].double().sum(1).float().contiguous()
            self.pw1f_b = (w1 @ lnb + sd[p + ".pwconv1.bias"].detach().double()).float().contiguous().to(device)
        # fused MLP (kpf_convnext_mlp_h16: the 4C-wide hidden tensor never reaches HBM) where the library has it: pwconv2's weight chunk-major
        # [4C/32][C][32] with the hidden index of a 32-block in the order GEMM1's accumulator registers form GEMM2's operand
        self.fused = (not FORCE_UNFUSED_MLP16) and bool(L.load().kpf_convnext_mlp_h16_supported(c))
        if self.fused:
            from .engine import MLP_HIDDEN_PERM
            w2 = sd[p + ".pwconv2.weight"].detach().float().reshape(c, 4 * c)
            perm = torch.tensor(MLP_HIDDEN_PERM, dtype=torch.long)
            w2c = w2.view(c, 4 * c // 32, 32)[:, :, perm].permute(1, 0, 2).contiguous()  # [chunk][c][k-slot]
            self.w2c = w2c.to(tdt).to(device)
            self.c = c

    def __call__(self, x, y, h, kdt, st=None):
        lib = L.load()
        fused = self.fused and x.ld == x.C and x.coff == 0 and y.ld == y.C and y.coff == 0
        if st is not None:
            # round 4: LDS-tiled stencil + per-chunk LayerNorm statistics, then the normalisation from them (two launches that together take
            # 0.5-0.65 of the one-pass kernel's time at C >= 256: tools/dw_stats_bench.py)
            L.check(lib.kpf_dwconv7_stats_h16(_ptr(x.buf), _ptr(self.wdw), _ptr(self.bdw), _ptr(y.buf), _ptr(st), x.B, x.H, x.W, x.C, kdt, _stream()),
                    "kpf_dwconv7_stats_h16")
            rows = x.B * x.H * x.W
            if LN_FOLD and not fused and self.pw1f is not None and conv16(self.pw1f, y, kdt, out=h, flags=L.KPF_ACT_GELU, probe=True):
                # round 5: no normalisation pass — the statistics are merged to (mean, rstd) per pixel (4 us) and pwconv1's epilogue applies them
                mr = torch.empty(2 * rows, device=x.buf.device, dtype=torch.float32)  # (per call, like y / h / st: batches in flight must not share it)
                L.check(lib.kpf_ln_stats_merge(_ptr(st), _ptr(mr), rows, x.C, 1e-6, _stream()), "kpf_ln_stats_merge")
                conv16(self.pw1f, y, kdt, out=h, flags=L.KPF_ACT_GELU, ln=(mr, self.pw1f_s, self.pw1f_b))
                conv16(self.pw2, h, kdt, out=x, gamma=self.gamma, res=x)
                return x
            L.check(lib.kpf_ln_apply_stats_h16(_ptr(y.buf), _ptr(st), _ptr(self.lnw), _ptr(self.lnb), rows, x.C, 1e-6, kdt, _stream()),
                    "kpf_ln_apply_stats_h16")
        else:
            L.check(lib.kpf_dwconv7_ln_h16(_ptr(x.buf), _ptr(self.wdw), _ptr(self.bdw), _ptr(self.lnw), _ptr(self.lnb), _ptr(y.buf), x.B, x.H,
                                           x.W, x.C, 1e-6, kdt, _stream()), "kpf_dwconv7_ln_h16")
        if fused:
            M, c = x.B * x.H * x.W, self.c
            _launch("convnext_mlp_h16_kernel", 16.0 * M * c * c, 2.0 * (3 * M * c + 8 * c * c), (M, c, 4 * c, 1, 1),
                    lambda: L.check(L.load().kpf_convnext_mlp_h16(_ptr(y.buf), _ptr(x.buf), _ptr(self.pw1.w), _ptr(self.pw1.pc.b), _ptr(self.w2c), _ptr(self.pw2.pc.b),
                                                                  _ptr(self.gamma), _ptr(x.buf), M, c, kdt, _stream()), "kpf_convnext_mlp_h16"))
            return x
        conv16(self.pw1, y, kdt, out=h, flags=L.KPF_ACT_GELU)
        conv16(self.pw2, h, kdt, out=x, gamma=self.gamma, res=x)
        return x


class UNetPlan16:
    """One UNet stream (ConvNeXt or ResNet encoder) on 16-bit storage (convNeXT/resnetUnet.py:129-152, model/resnetUnet.py:309-330).  __call__(img NCHW fp32) ->
    (img_result NCHW fp32 B x 105 x F x F, img_feature Act NHWC 128 fp32)."""

    def __init__(self, sd, p, net, device, precision):
        self.fam, size = parse_net(net)
        self.tdt, self.kdt = DTYPES[precision]
        self.device = device
        sdp = {k[len(p) + 1:]: v for k, v in sd.items() if k.startswith(p + ".")}
        b = "backbone"
        f32 = lambda k: sdp[k].detach().float().contiguous().to(device)
        P16 = lambda *a, **k: Packed16(PackedConv(*a, **k), self.tdt)
        if self.fam == "convnext":
            depths, dims = CONVNEXT[size]
            self.dims = dims
            self.stem = PackedConv(sdp[b + ".downsample_layers.0.0.weight"], sdp[b + ".downsample_layers.0.0.bias"], device, stride=4, patchify=True)
            self.stem_ln = (f32(b + ".downsample_layers.0.1.weight"), f32(b + ".downsample_layers.0.1.bias"))
            self.down, self.down_ln = [None], [None]
            for i in range(1, 4):
                self.down_ln.append((f32(b + ".downsample_layers.%d.0.weight" % i), f32(b + ".downsample_layers.%d.0.bias" % i)))
                self.down.append(P16(sdp[b + ".downsample_layers.%d.1.weight" % i], sdp[b + ".downsample_layers.%d.1.bias" % i], device, stride=2, patchify=True))
            self.stages = [[Block16(sdp, b + ".stages.%d.%d" % (i, j), device, self.tdt) for j in range(depths[i])] for i in range(4)]
        else:  # ResNet (model/resnet.py:232-244): fp32 7x7/s2 stem (BatchNorm folded) + max-pool, then 16-bit stages with folded BatchNorms
            self.stem = PackedConv(sdp[b + ".conv1.weight"], None, device, stride=2, pad=3, fold_bn=bn_scale_shift(sdp, b + ".bn1"), cin_pad=4)
            self.layers = []
            for li in range(1, 5):
                blocks = []
                j = 0
                while (b + ".layer%d.%d.conv1.weight" % (li, j)) in sdp:
                    q = b + ".layer%d.%d" % (li, j)
                    stride = 2 if (li > 1 and j == 0) else 1
                    ds = None
                    if (q + ".downsample.0.weight") in sdp:
                        ds = P16(sdp[q + ".downsample.0.weight"], None, device, stride=stride, fold_bn=bn_scale_shift(sdp, q + ".downsample.1"))
                    if (q + ".conv3.weight") in sdp:
                        blocks.append((P16(sdp[q + ".conv1.weight"], None, device, fold_bn=bn_scale_shift(sdp, q + ".bn1")),
                                       P16(sdp[q + ".conv2.weight"], None, device, stride=stride, pad=1, fold_bn=bn_scale_shift(sdp, q + ".bn2")),
                                       P16(sdp[q + ".conv3.weight"], None, device, fold_bn=bn_scale_shift(sdp, q + ".bn3")), ds))
                    else:
                        blocks.append((P16(sdp[q + ".conv1.weight"], None, device, stride=stride, pad=1, fold_bn=bn_scale_shift(sdp, q + ".bn1")),
                                       P16(sdp[q + ".conv2.weight"], None, device, pad=1, fold_bn=bn_scale_shift(sdp, q + ".bn2")), None, ds))
                    j += 1
                self.layers.append(blocks)
        R = lambda name: Residual16(sdp, name, device, self.tdt)
        self.up4, self.skip4, self.fus4 = R("up4.0"), R("skip_layer4"), R("fusion_layer4")
        self.up3, self.skip3, self.fus3 = R("up3.0"), R("skip_layer3"), R("fusion_layer3")
        self.up2, self.skip2, self.fus2 = R("up2.0"), R("skip_layer2"), R("fusion_layer2")
        self.result_emb = R("result_emb") if self.fam == "convnext" else None
        wf = torch.cat([sdp["finals.%d.weight" % i] for i in range(3)], 0)
        bf = torch.cat([sdp["finals.%d.bias" % i] for i in range(3)], 0)
        self.finals = Packed16(PackedConv(wf, bf, device), self.tdt)

    def _ln(self, x, wb, out, x_kdt):
        L.check(L.load().kpf_layernorm_h16(_ptr(x.buf), x_kdt, _ptr(wb[0]), _ptr(wb[1]), _ptr(out.buf), self.kdt, x.B * x.H * x.W, x.C, 1e-6,
                                           _stream()), "kpf_layernorm_h16")
        return out

    def _resnet(self, img):
        lib = L.load()
        kdt = self.kdt
        x = maxpool3x3s2(conv(self.stem, nchw_to_nhwc(img, cpad=4), flags=L.KPF_ACT_RELU))  # fp32 stem + pool
        x16 = empty16(x.B, x.H, x.W, x.C, self.device, self.tdt)
        L.check(lib.kpf_cast_f32_h16(_ptr(x.buf), _ptr(x16.buf), kdt, x.buf.numel(), _stream()), "kpf_cast_f32_h16")
        x = x16
        feats = []
        for blocks in self.layers:
            for c1, c2, c3, ds in blocks:
                h = conv16(c1, x, kdt, flags=L.KPF_ACT_RELU)
                idt = conv16(ds, x, kdt) if ds is not None else x
                if c3 is None:
                    x = conv16(c2, h, kdt, res=idt, flags=L.KPF_RELU_AFTER_RES)
                else:
                    x = conv16(c3, conv16(c2, h, kdt, flags=L.KPF_ACT_RELU), kdt, res=idt, flags=L.KPF_RELU_AFTER_RES)
            feats.append(x)
        return feats

    def _convnext(self, img):
        dev, tdt, kdt = self.device, self.tdt, self.kdt
        B, Cc, S, _ = img.shape
        x = Act(img.contiguous().float().view(-1), B, S, S, 1) if Cc == 1 else nchw_to_nhwc(img)
        feats = []
        cur = None
        for i in range(4):
            if i == 0:
                s32 = conv(self.stem, x)  # fp32 image -> fp32 (K = 16 / 48: not worth a 16-bit repack of the input)
                cur = self._ln(s32, self.stem_ln, empty16(s32.B, s32.H, s32.W, s32.C, dev, tdt), L.KPF_DT_F32)
            else:
                t = self._ln(cur, self.down_ln[i], empty16(cur.B, cur.H, cur.W, cur.C, dev, tdt), kdt)
                cur = conv16(self.down[i], t, kdt)
            y = empty16(cur.B, cur.H, cur.W, cur.C, dev, tdt)
            fused = all(blk.fused for blk in self.stages[i])
            h = None if fused else empty16(cur.B, cur.H, cur.W, 4 * cur.C, dev, tdt)  # (the fused MLP keeps the 4C-wide hidden tensor in registers)
            lib = L.load()
            st = None
            if DW_STATS and cur.C >= DW_STATS_MIN_C and lib.kpf_dwconv7_stats_supported(cur.H, cur.W, cur.C):
                st = torch.empty(lib.kpf_dwconv7_stats_floats(cur.B, cur.H, cur.W, cur.C), device=dev, dtype=torch.float32)
            for blk in self.stages[i]:
                blk(cur, y, h, kdt, st)
            feats.append(cur)
        return feats

    def __call__(self, img):
        lib = L.load()
        dev, tdt, kdt = self.device, self.tdt, self.kdt
        B = img.shape[0]
        c1, c2, c3, c4 = self._convnext(img) if self.fam == "convnext" else self._resnet(img)

        def level(up, skip, fus, lo, hi):
            cat = empty16(B, hi.H, hi.W, up.cout + skip.cout, dev, tdt)
            u = up(lo, kdt)
            dst = cat.slice(0, up.cout)
            L.check(lib.kpf_upsample2x_h16(_ptr(u.buf), _ptr(dst.buf), u.B, u.H, u.W, u.C, dst.ld, dst.coff, kdt, _stream()), "kpf_upsample2x_h16")
            skip(hi, kdt, out=cat.slice(up.cout, skip.cout))
            return fus(cat, kdt)

        c3f = level(self.up4, self.skip4, self.fus4, c4, c3)
        c2f = level(self.up3, self.skip3, self.fus3, c3f, c2)
        feat = level(self.up2, self.skip2, self.fus2, c2f, c1)
        if self.result_emb is not None:
            feat = self.result_emb(feat, kdt)
        res = torch.empty(B, 105, feat.H, feat.W, device=dev, dtype=torch.float32)
        conv16(self.finals, feat, kdt, out_nchw=res)
        f32 = Act.empty(B, feat.H, feat.W, feat.C, dev)
        L.check(lib.kpf_cast_h16_f32(_ptr(feat.buf), kdt, _ptr(f32.buf), B * feat.H * feat.W, feat.C, feat.ld, feat.coff, _stream()), "kpf_cast_h16_f32")
        return res, f32
