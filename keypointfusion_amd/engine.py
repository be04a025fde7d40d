"""Host-side execution plans for the KPFusion hot path on MI355X.

Python here is plumbing: it repacks the reference-layout state dict (NCHW/OIHW, BatchNorm as separate tensors) into
the kernel layouts once (OHWI rows padded to 32, eval-BatchNorm folded into the producing convolution or turned into
an operand prologue), owns activation buffers through torch's caching allocator, and issues the C-ABI calls of
include/kpf.h on torch's current HIP stream.  All arithmetic happens in libkpf_hip.so.

Activation layout: NHWC everywhere ("pixel rows, channels contiguous"), so 1x1 convolutions / Linear layers are plain
row-major GEMMs, channel concatenation is a slice of a wider pixel row (ld, coff) and patchify convolutions
(4x4/s4 stem, 2x2/s2 downsample) are KHx1 convolutions over a [H, W/KW, KW*C] view of the same memory.
"""
import ctypes as C
import math

import torch

from . import lib as L
from .spec import CONVNEXT, parse_net

BN_EPS = 1e-5
FORCE_UNFUSED_MLP = bool(int(__import__("os").environ.get("KPF_UNFUSED_MLP", "0")))  # A/B switch for tuning
# GEMM arithmetic.  "f32" (default): every GEMM on v_mfma_f32_16x16x4_f32 — IEEE fp32 products and accumulation, bit for bit an
# fmaf chain, the reference's arithmetic.  "split" (KPF_GEMM=split, opt-in): fp32 emulation on the f16 matrix cores — operands as
# f16 hi + lo (22 bits), 3 MFMAs per product (hi*hi + hi*lo + lo*hi), fp32 accumulate — used ONLY for layers whose operands have a
# bound proven at pack time (PackedConv.split_allowed; LayerNorm / GELU outputs of the ConvNeXt blocks): operands outside the f16
# range would saturate, so every other layer (pre-activation Residuals, ResNet stages, the fusion head) stays on the f32 MFMA.
GEMM_MODE = __import__("os").environ.get("KPF_GEMM", "f32")
assert GEMM_MODE in ("split", "f32"), "KPF_GEMM must be 'split' or 'f32'"
F16_MAX = 65504.0
# Optional tile-shape autotuning of the implicit GEMM (KPF_AUTOTUNE=1): the first time a (layer, shape, epilogue) combination runs,
# every tile configuration is timed on the real operands and the fastest is remembered (like a cuDNN/MIOpen "find").  Off by default:
# timed in isolation it makes the GEMMs 4 % faster (7.63 vs 7.96 ms per step), but with the two backbones overlapped on two streams
# the step gets slower (10.77 vs 10.42 ms) — the isolated optimum fills the whole chip and leaves nothing for the other stream.
AUTOTUNE = bool(int(__import__("os").environ.get("KPF_AUTOTUNE", "0")))
# experiment: per-shape tile overrides "M:N:K:kh=cfg,..." (cfg = configuration index)
_TILE_RULES = {tuple(int(v) for v in r.split("=")[0].split(":")): int(r.split("=")[1]) + 1 for r in __import__("os").environ.get("KPF_TILE_RULES", "").split(",") if r}


# Per-launch profiling hook (bench.py): when PROFILE is a list, every MFMA-kernel launch is bracketed by HIP events recorded on
# the launch stream and appended as (kernel, start_event, end_event, algorithmic_flops, algorithmic_bytes, shape).
PROFILE = None
FORCE_TILE = 0  # tuning aid (tools/smallk_scan.py): tile configuration index + 1 for every kpf_conv2d_f32 launch, 0 = the library's choice


def _launch(kernel, flops, nbytes, shape, fn):
    if PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROFILE.append((kernel, e0, e1, flops, nbytes, shape))
    return r


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Act:
    """A channel slice [coff, coff+C) of an NHWC buffer with pixel stride ld."""
    __slots__ = ("buf", "B", "H", "W", "C", "ld", "coff", "split")

    def __init__(self, buf, B, H, W, C, ld=None, coff=0, split=False):
        self.buf, self.B, self.H, self.W, self.C = buf, B, H, W, C
        self.ld = C if ld is None else ld
        self.coff = coff
        self.split = split  # rows hold [32 x f16 hi | 32 x f16 lo] blocks instead of fp32 (same bytes; include/kpf.h)

    @staticmethod
    def empty(B, H, W, C, device):
        return Act(torch.empty(B * H * W * C, device=device, dtype=torch.float32), B, H, W, C)

    def slice(self, coff, C_):
        return Act(self.buf, self.B, self.H, self.W, C_, self.ld, self.coff + coff)

    def dense(self):
        assert self.ld == self.C and self.coff == 0
        return self.buf.view(self.B, self.H, self.W, self.C)


# ----------------------------------------------------------------------------------------------------------------
# weight packing
# ----------------------------------------------------------------------------------------------------------------
def bn_scale_shift(sd, p):
    """eval BatchNorm as y = x*s + t (float64 fold, stored fp32)."""
    w, b = sd[p + ".weight"].double(), sd[p + ".bias"].double()
    m, v = sd[p + ".running_mean"].double(), sd[p + ".running_var"].double()
    s = w / torch.sqrt(v + BN_EPS)
    return s, b - m * s


def split_pack(w):
    """fp32/fp64 [N][K] (K % 32 == 0) -> (fp32-viewed [N][K] tensor holding [K/32][hi 32 | lo 32] f16 of w * 2^s, 2^-s): the split
    operand format of include/kpf.h; s keeps the lo halves out of the f16 subnormals."""
    w = w.double().cpu()
    N, K = w.shape
    assert K % 32 == 0
    amax = float(w.abs().max())
    s = 7 - math.floor(math.log2(amax)) if amax > 0 else 0
    ws = w * (2.0 ** s)
    hi = ws.half()
    lo = (ws - hi.double()).half()
    blk = torch.stack([hi.view(N, K // 32, 32), lo.view(N, K // 32, 32)], 2).contiguous()  # N, K/32, 2, 32
    return blk.view(torch.float32).reshape(N, K).contiguous(), 2.0 ** (-s)


# order in which GEMM1's accumulator registers of two neighbouring 16-wide hidden tiles form a 32-deep f16 MFMA operand
# (csrc/kpf_mlp.hip, convnext_mlp_split_kernel): k-slot 8g+j holds hidden 16*(j>>2) + 4g + (j&3)
MLP_HIDDEN_PERM = [16 * ((k & 7) >> 2) + 4 * (k >> 3) + (k & 3) for k in range(32)]


class PackedConv:
    """Weights of one convolution/linear in kernel layout: w [N][Kp] with k = (ky,kx,c), bias [N], optional input
    prologue (scale, shift) [Cin].  `view_kw` > 1 marks a patchify convolution executed as KHx1 over a merged view."""

    def __init__(self, weight, bias, device, stride=1, pad=0, fold_bn=None, prologue=None, cin_pad=None, patchify=False, n_pad=None):
        w = weight.detach().double().cpu()
        if w.dim() == 2:
            w = w[:, :, None, None]
        elif w.dim() == 3:
            w = w[:, :, :, None]
        N, Cin, KH, KW = w.shape
        b = bias.detach().double().cpu() if bias is not None else torch.zeros(N, dtype=torch.float64)
        if fold_bn is not None:  # conv -> BN : W' = W*s, b' = b*s + t
            s, t = fold_bn
            w = w * s.cpu()[:, None, None, None]
            b = b * s.cpu() + t.cpu()
        if n_pad is not None and n_pad > N:  # extra output channels that are identically zero (zero rows, zero bias)
            w = torch.cat([w, torch.zeros(n_pad - N, Cin, KH, KW, dtype=w.dtype)], 0)
            b = torch.cat([b, torch.zeros(n_pad - N, dtype=b.dtype)])
            N = n_pad
        if cin_pad is not None and cin_pad > Cin:
            w = torch.cat([w, torch.zeros(N, cin_pad - Cin, KH, KW, dtype=w.dtype)], 1)
            Cin = cin_pad
        w = w.permute(0, 2, 3, 1).contiguous()  # N KH KW Cin
        if patchify:  # kernel == stride, pad 0: merge (kx, c) into the channel axis of a [H, W/KW, KW*C] view
            assert stride == KH == KW and pad == 0
            self.KH, self.KW, self.Cin = KH, 1, KW * Cin
            self.sh, self.sw, self.ph, self.pw = KH, 1, 0, 0
            self.merge = KW
        else:
            self.KH, self.KW, self.Cin = KH, KW, Cin
            self.sh = self.sw = stride
            self.ph = self.pw = pad
            self.merge = 1
        assert self.Cin % 4 == 0, "input channels (after view) must be a multiple of 4"
        K = KH * KW * Cin
        self.N, self.K = N, K
        self.Kp = (K + 31) // 32 * 32
        wp = torch.zeros(N, self.Kp, dtype=torch.float64)
        wp[:, :K] = w.reshape(N, K)
        self.w = wp.float().to(device)
        self.b = b.float().to(device)
        # split (3 x f16) arithmetic only where the caller has proven |activation| < 65504 at pack time (ConvNeXtBlockPlan, the
        # downsample LayerNorms): everywhere else a large activation would saturate silently, so the default is the f32 MFMA
        self.split_allowed = False
        self.tuned = {}  # (shape, epilogue) -> tile configuration index + 1 (autotuning cache)
        self.ps = self.pt = None
        if prologue is not None:
            s, t = prologue
            assert s.numel() == Cin
            self.ps, self.pt = s.float().to(device).contiguous(), t.float().to(device).contiguous()

    def flops(self, M):
        return 2.0 * M * self.N * self.K

    def split_weights(self):
        """(w_split, w_unscale): rows of [Kp/32][hi 32 | lo 32] f16 of w * 2^s (s keeps the lo halves out of the f16 subnormals),
        viewed as fp32 [N][Kp]; built once from the fp32 pack (exactly representable inputs: the split is of the fp32 weights)."""
        if getattr(self, "_ws", None) is None:
            ws, self._wus = split_pack(self.w)
            self._ws = ws.to(self.w.device)
        return self._ws, self._wus

    def row_l1(self):
        return float(self.w.abs().sum(1).max())


def conv(pc, x, out=None, flags=0, gamma=None, res=None, out_nchw=None, out_split=False, out2=None):
    """Launch kpf_conv2d_f32.  x: Act; out: Act (or None to allocate dense); res: Act."""
    lib = L.load()
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
    d.KH, d.KW, d.sh, d.sw, d.ph, d.pw, d.Kp = pc.KH, pc.KW, pc.sh, pc.sw, pc.ph, pc.pw, pc.Kp
    if out_nchw is not None:
        flags |= L.KPF_OUT_NCHW
        optr = out_nchw
        d.out_ld, d.out_coff = pc.N, 0
    else:
        if out is None:
            out = Act.empty(B, OH, OW, pc.N, x.buf.device)
        assert (out.B, out.H, out.W, out.C) == (B, OH, OW, pc.N), ((out.B, out.H, out.W, out.C), (B, OH, OW, pc.N))
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
    w = pc.w
    if x.split:
        flags |= L.KPF_IN_SPLIT
        w, d.w_unscale = pc.split_weights()
    elif GEMM_MODE == "split" and pc.Cin % 32 == 0 and pc.split_allowed:
        flags |= L.KPF_W_SPLIT  # fp32 activations, split in registers
        w, d.w_unscale = pc.split_weights()
    if out_split:
        flags |= L.KPF_OUT_SPLIT
        out.split = True
    elif out is not None and out_nchw is None:
        out.split = False
    d.flags = flags
    d.groups, d.w_gstride = getattr(pc, "groups", 0), getattr(pc, "w_gstride", 0)  # (grouped launch: training.GroupedPack; 0 = one convolution)
    M = B * OH * OW
    if AUTOTUNE:
        key = (M, IH, IW, flags, in_ld, in_coff, d.out_ld, d.out_coff, d.res_ld, d.res_coff)
        cfg = pc.tuned.get(key)
        if cfg is None and PROFILE is None and not torch.cuda.is_current_stream_capturing():
            cfg = pc.tuned[key] = _autotune(lib, d, x, w, pc, gamma, res, optr, flags)
        d.tile_cfg = cfg or 0
    if _TILE_RULES and not AUTOTUNE:
        d.tile_cfg = _TILE_RULES.get((M, pc.N, pc.K, pc.KH), 0)
    if FORCE_TILE:
        d.tile_cfg = FORCE_TILE
    # algorithmic bytes: input pixels once, weights once, output once (+ residual once)
    nbytes = 4.0 * (B * IH * IW * pc.Cin + pc.N * pc.K + M * pc.N * (2 if res is not None else 1))
    ng = max(1, d.groups)
    _launch("igemm_split_kernel" if flags & (L.KPF_IN_SPLIT | L.KPF_W_SPLIT) else "igemm_f32_kernel", pc.flops(M) * ng, nbytes * ng, (M, pc.N, pc.K, pc.KH, pc.KW),
            lambda: L.check(lib.kpf_conv2d_f32(C.byref(d), _ptr(x.buf), _ptr(w), _ptr(pc.b), _ptr(pc.ps), _ptr(pc.pt), _ptr(gamma),
                                               _ptr(res.buf if res is not None else None), _ptr(optr), _stream()), "kpf_conv2d_f32"))
    return out


def _autotune(lib, d, x, w, pc, gamma, res, optr, flags):
    """Time every tile configuration on the real operands (output to a scratch buffer when the launch is an in-place residual
    update, so repeated runs do not accumulate) and return the fastest as index + 1."""
    ncfg = int(lib.kpf_conv_num_tile_cfgs())
    split = bool(flags & (L.KPF_IN_SPLIT | L.KPF_W_SPLIT))
    cands = [i for i in range(ncfg) if split or i < 9 or i == 17]  # 9-16 are LDS-ring / single-stage variants for split operands
    out_t = optr
    if res is not None and res.buf.data_ptr() == optr.data_ptr():
        out_t = torch.empty_like(optr)
    args = (_ptr(x.buf), _ptr(w), _ptr(pc.b), _ptr(pc.ps), _ptr(pc.pt), _ptr(gamma), _ptr(res.buf if res is not None else None), _ptr(out_t), _stream())
    torch.cuda.synchronize()  # nothing else (the other backbone's stream) runs while candidates are timed
    best, best_t = 0, float("inf")
    for c in cands:
        d.tile_cfg = c + 1
        if lib.kpf_conv2d_f32(C.byref(d), *args) != 0:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            lib.kpf_conv2d_f32(C.byref(d), *args)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1)
        if t < best_t:
            best, best_t = c + 1, t
    return best


def layernorm(x, w, b, eps, out=None, out_split=False):
    lib = L.load()
    assert x.ld == x.C and x.coff == 0
    out = x if out is None else out
    fn = lib.kpf_layernorm_split_f32 if out_split else lib.kpf_layernorm_f32
    L.check(fn(_ptr(x.buf), _ptr(w), _ptr(b), _ptr(out.buf), x.B * x.H * x.W, x.C, eps, _stream()), "kpf_layernorm_f32")
    out.split = out_split
    return out


def upsample2x(x, out):
    lib = L.load()
    assert x.ld == x.C and x.coff == 0 and out.C == x.C
    L.check(lib.kpf_upsample2x_f32(_ptr(x.buf), _ptr(out.buf), x.B, x.H, x.W, x.C, out.ld, out.coff, _stream()),
            "kpf_upsample2x_f32")
    return out


def crop_inverse(M):
    """M^-1 [B][3][3] on the device, in the rounding order of the reference's PyTorch-CPU forward on this host (torch.linalg.inv ->
    MKL getrf / getrs, dataloader/loader.py:781; inv3x3.host_mode picks the fused or the separately rounded variant).  No host round
    trip on hosts whose library follows one of the two known orders (every box seen so far), and never under stream capture."""
    from .inv3x3 import host_mode
    B = M.shape[0]
    mode = host_mode()
    if mode < 0:  # a host library that follows neither known order
        if not torch.cuda.is_current_stream_capturing():
            # eager: the reference's own call on the host (dataloader/loader.py:781) — exact, at the price of a synchronisation
            return torch.linalg.inv(M.detach().float().cpu().view(B, 1, 3, 3)).view(B, 3, 3).to(M.device)
        mode = 0  # under capture M must not leave the device
    out = torch.empty(B, 3, 3, device=M.device, dtype=torch.float32)
    L.check(L.load().kpf_inv3x3_f32(_ptr(M.detach().float().contiguous()), _ptr(out), B, mode, _stream()), "kpf_inv3x3_f32")
    return out


def nchw_to_nhwc(t, cpad=None):
    lib = L.load()
    B, Cc, H, W = t.shape
    cpad = Cc if cpad is None else cpad
    t = t.contiguous().float()
    out = Act.empty(B, H, W, cpad, t.device)
    L.check(lib.kpf_nchw_to_nhwc_f32(_ptr(t), _ptr(out.buf), B, Cc, H, W, cpad, _stream()), "kpf_nchw_to_nhwc_f32")
    return out


def nhwc_to_nchw(x):
    lib = L.load()
    out = torch.empty(x.B, x.C, x.H, x.W, device=x.buf.device, dtype=torch.float32)
    L.check(lib.kpf_nhwc_to_nchw_f32(_ptr(x.buf), _ptr(out), x.B, x.C, x.H, x.W, x.ld, x.coff, _stream()),
            "kpf_nhwc_to_nchw_f32")
    return out


def maxpool3x3s2(x):
    lib = L.load()
    assert x.ld == x.C and x.coff == 0
    OH, OW = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
    out = Act.empty(x.B, OH, OW, x.C, x.buf.device)
    L.check(lib.kpf_maxpool3x3s2_f32(_ptr(x.buf), _ptr(out.buf), x.B, x.H, x.W, x.C, _stream()), "kpf_maxpool3x3s2_f32")
    return out


# ----------------------------------------------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------------------------------------------
class ResidualPlan:
    """Pre-activation bottleneck of model/hourglass.py:87-119 as 3 (4 with a skip conv) MFMA launches:
    conv1 = [bn1+relu prologue] 1x1 [bn2 folded, relu] ; conv2 = 3x3 [bn3 folded, relu] ; conv3 = 1x1 + residual.
    Always on the f32-input MFMA, also under KPF_GEMM=split: its operands are unbounded ReLU outputs and residual sums (no pack-time
    bound on |activation| exists), which the f16 split format would saturate at 65504."""

    def __init__(self, sd, p, device):
        cin = sd[p + ".conv1.conv.weight"].shape[1]
        cout = sd[p + ".conv3.conv.weight"].shape[0]
        self.cin, self.cout = cin, cout
        self.c1 = PackedConv(sd[p + ".conv1.conv.weight"], sd[p + ".conv1.conv.bias"], device,
                             fold_bn=bn_scale_shift(sd, p + ".bn2"), prologue=bn_scale_shift(sd, p + ".bn1"))
        self.c2 = PackedConv(sd[p + ".conv2.conv.weight"], sd[p + ".conv2.conv.bias"], device, pad=1,
                             fold_bn=bn_scale_shift(sd, p + ".bn3"))
        self.c3 = PackedConv(sd[p + ".conv3.conv.weight"], sd[p + ".conv3.conv.bias"], device)
        self.skip = None
        if cin != cout:
            self.skip = PackedConv(sd[p + ".skip_layer.conv.weight"], sd[p + ".skip_layer.conv.bias"], device)

    def __call__(self, x, out=None):
        h = conv(self.c1, x, flags=L.KPF_ACT_RELU)
        h = conv(self.c2, h, flags=L.KPF_ACT_RELU)
        if out is None:
            out = Act.empty(x.B, x.H, x.W, self.cout, x.buf.device)
        if self.skip is not None:
            conv(self.skip, x, out=out)
            return conv(self.c3, h, out=out, res=out)
        return conv(self.c3, h, out=out, res=x)

    def flops(self, M):
        f = self.c1.flops(M) + self.c2.flops(M) + self.c3.flops(M)
        return f + (self.skip.flops(M) if self.skip else 0)


class ConvNeXtBlockPlan:
    """convNeXT/convnext.py:39-52 as: fused dw7x7+LN kernel, pw1 GEMM (+GELU), pw2 GEMM (+gamma*y + x, in place)."""

    def __init__(self, sd, p, device):
        c = sd[p + ".gamma"].numel()
        self.C = c
        self.wdw = sd[p + ".dwconv.weight"].detach().float().reshape(c, 49).t().contiguous().to(device)  # [49][C]
        self.bdw = sd[p + ".dwconv.bias"].detach().float().to(device)
        self.lnw = sd[p + ".norm.weight"].detach().float().to(device)
        self.lnb = sd[p + ".norm.bias"].detach().float().to(device)
        self.pw1 = PackedConv(sd[p + ".pwconv1.weight"], sd[p + ".pwconv1.bias"], device)
        self.pw2 = PackedConv(sd[p + ".pwconv2.weight"], sd[p + ".pwconv2.bias"], device)
        self.gamma = sd[p + ".gamma"].detach().float().to(device)
        # split (3 x f16) arithmetic needs activations inside the f16 range: |LN out| <= sqrt(C) max|w| + max|b| (a normalised row has
        # L2 norm sqrt(C)), |pw1 out| <= max row L1 norm of W1 times that, plus the bias; GELU does not grow magnitudes
        ln_bound = math.sqrt(c) * float(self.lnw.abs().max()) + float(self.lnb.abs().max())
        h_bound = self.pw1.row_l1() * ln_bound + float(self.pw1.b.abs().max())
        self.split_ok = c % 32 == 0 and ln_bound < F16_MAX and h_bound < F16_MAX
        self.pw1.split_allowed = self.pw2.split_allowed = self.split_ok  # operands proven inside the f16 range
        self.fused_split = GEMM_MODE == "split" and self.split_ok and bool(L.load().kpf_convnext_mlp_split_supported(c))
        if self.fused_split:
            self.w1s, self.us1 = self.pw1.split_weights()
            w2 = self.pw2.w.double().cpu()  # [C][4C]
            perm = torch.tensor([32 * q + k for q in range(4 * c // 32) for k in MLP_HIDDEN_PERM])
            w2s, self.us2 = split_pack(w2[:, perm])
            self.w2s = w2s.to(device)
        self.fused = bool(L.load().kpf_convnext_mlp_supported(c))
        if self.fused:  # fused MLP kernel takes the PyTorch layouts as they are
            self.w1 = sd[p + ".pwconv1.weight"].detach().float().contiguous().to(device)
            self.b1 = sd[p + ".pwconv1.bias"].detach().float().contiguous().to(device)
            self.w2 = sd[p + ".pwconv2.weight"].detach().float().contiguous().to(device)
            self.b2 = sd[p + ".pwconv2.bias"].detach().float().contiguous().to(device)

    def __call__(self, x, y, h):
        lib = L.load()
        if GEMM_MODE == "split" and self.fused_split and not FORCE_UNFUSED_MLP:
            L.check(lib.kpf_dwconv7_ln_split_f32(_ptr(x.buf), _ptr(self.wdw), _ptr(self.bdw), _ptr(self.lnw), _ptr(self.lnb),
                                                 _ptr(y.buf), x.B, x.H, x.W, x.C, 1e-6, _stream()), "kpf_dwconv7_ln_split_f32")
            M, Cc = x.B * x.H * x.W, x.C
            _launch("convnext_mlp_split_kernel", 16.0 * M * Cc * Cc, 4.0 * (3 * M * Cc + 8 * Cc * Cc), (M, Cc, 4 * Cc, 1, 1),
                    lambda: L.check(lib.kpf_convnext_mlp_split_f32(_ptr(y.buf), _ptr(x.buf), _ptr(self.w1s), _ptr(self.pw1.b), self.us1,
                                                                   _ptr(self.w2s), _ptr(self.pw2.b), self.us2, _ptr(self.gamma), _ptr(x.buf), M, Cc,
                                                                   _stream()), "kpf_convnext_mlp_split_f32"))
            return x
        if GEMM_MODE == "split" and self.split_ok:
            L.check(lib.kpf_dwconv7_ln_split_f32(_ptr(x.buf), _ptr(self.wdw), _ptr(self.bdw), _ptr(self.lnw), _ptr(self.lnb),
                                                 _ptr(y.buf), x.B, x.H, x.W, x.C, 1e-6, _stream()), "kpf_dwconv7_ln_split_f32")
            y.split = True
            conv(self.pw1, y, out=h, flags=L.KPF_ACT_GELU, out_split=True)
            conv(self.pw2, h, out=x, gamma=self.gamma, res=x)
            y.split = h.split = False
            return x
        L.check(lib.kpf_dwconv7_ln_f32(_ptr(x.buf), _ptr(self.wdw), _ptr(self.bdw), _ptr(self.lnw), _ptr(self.lnb),
                                       _ptr(y.buf), x.B, x.H, x.W, x.C, 1e-6, _stream()), "kpf_dwconv7_ln_f32")
        if self.fused and not FORCE_UNFUSED_MLP:
            M, Cc = x.B * x.H * x.W, x.C
            _launch("convnext_mlp_kernel", 16.0 * M * Cc * Cc, 4.0 * (3 * M * Cc + 8 * Cc * Cc), (M, Cc, 4 * Cc, 1, 1),
                    lambda: L.check(lib.kpf_convnext_mlp_f32(_ptr(y.buf), _ptr(x.buf), _ptr(self.w1), _ptr(self.b1), _ptr(self.w2),
                                                             _ptr(self.b2), _ptr(self.gamma), _ptr(x.buf), M, Cc, _stream()),
                                    "kpf_convnext_mlp_f32"))
            return x
        conv(self.pw1, y, out=h, flags=L.KPF_ACT_GELU)
        conv(self.pw2, h, out=x, gamma=self.gamma, res=x)
        return x


class UNetPlan:
    """One backbone stream (encoder + 3-level Residual UNet decoder + heads): convNeXT/resnetUnet.py:129-152 /
    model/resnetUnet.py:309-330.  __call__(img NCHW) -> (img_result NCHW B x 105 x F x F, img_feature Act NHWC 128)."""

    def __init__(self, sd, p, net, device):
        self.fam, size = parse_net(net)
        self.device = device
        sdp = {k[len(p) + 1:]: v for k, v in sd.items() if k.startswith(p + ".")}
        self.in_ch = (sdp["backbone.downsample_layers.0.0.weight"] if self.fam == "convnext" else sdp["backbone.conv1.weight"]).shape[1]
        if self.fam == "convnext":
            depths, dims = CONVNEXT[size]
            self.dims = dims
            b = "backbone"
            self.stem = PackedConv(sdp[b + ".downsample_layers.0.0.weight"], sdp[b + ".downsample_layers.0.0.bias"], device,
                                   stride=4, patchify=True)
            self.stem_ln = (sdp[b + ".downsample_layers.0.1.weight"].float().to(device), sdp[b + ".downsample_layers.0.1.bias"].float().to(device))
            self.down, self.down_ln, self.down_ln_bound = [None], [None], [None]
            for i in range(1, 4):
                lw_, lb_ = sdp[b + ".downsample_layers.%d.0.weight" % i], sdp[b + ".downsample_layers.%d.0.bias" % i]
                self.down_ln_bound.append(math.sqrt(lw_.numel()) * float(lw_.abs().max()) + float(lb_.abs().max()))
                self.down_ln.append((sdp[b + ".downsample_layers.%d.0.weight" % i].float().to(device),
                                     sdp[b + ".downsample_layers.%d.0.bias" % i].float().to(device)))
                self.down.append(PackedConv(sdp[b + ".downsample_layers.%d.1.weight" % i], sdp[b + ".downsample_layers.%d.1.bias" % i],
                                            device, stride=2, patchify=True))
            self.stages = [[ConvNeXtBlockPlan(sdp, b + ".stages.%d.%d" % (i, j), device) for j in range(depths[i])] for i in range(4)]
        else:
            self.dims = dims = (64, 128, 256, 512)
            b = "backbone"
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
                        ds = PackedConv(sdp[q + ".downsample.0.weight"], None, device, stride=stride, fold_bn=bn_scale_shift(sdp, q + ".downsample.1"))
                    if (q + ".conv3.weight") in sdp:  # Bottleneck (v1.5: the stride sits on the 3x3), model/resnet.py:78-135
                        c1 = PackedConv(sdp[q + ".conv1.weight"], None, device, fold_bn=bn_scale_shift(sdp, q + ".bn1"))
                        c2 = PackedConv(sdp[q + ".conv2.weight"], None, device, stride=stride, pad=1, fold_bn=bn_scale_shift(sdp, q + ".bn2"))
                        c3 = PackedConv(sdp[q + ".conv3.weight"], None, device, fold_bn=bn_scale_shift(sdp, q + ".bn3"))
                        blocks.append((c1, c2, c3, ds))
                        j += 1
                        continue
                    c1 = PackedConv(sdp[q + ".conv1.weight"], None, device, stride=stride, pad=1, fold_bn=bn_scale_shift(sdp, q + ".bn1"))
                    c2 = PackedConv(sdp[q + ".conv2.weight"], None, device, pad=1, fold_bn=bn_scale_shift(sdp, q + ".bn2"))
                    blocks.append((c1, c2, None, ds))
                    j += 1
                self.layers.append(blocks)
        d = dims
        R = lambda name: ResidualPlan(sdp, name, device)
        self.up4, self.skip4, self.fus4 = R("up4.0"), R("skip_layer4"), R("fusion_layer4")
        self.up3, self.skip3, self.fus3 = R("up3.0"), R("skip_layer3"), R("fusion_layer3")
        self.up2, self.skip2, self.fus2 = R("up2.0"), R("skip_layer2"), R("fusion_layer2")
        self.result_emb = R("result_emb") if self.fam == "convnext" else None
        wf = torch.cat([sdp["finals.%d.weight" % i] for i in range(3)], 0)
        bf = torch.cat([sdp["finals.%d.bias" % i] for i in range(3)], 0)
        self.finals = PackedConv(wf, bf, device)

    # -- encoders -------------------------------------------------------------------------------------------
    def _convnext(self, img):
        B, Cc, S, _ = img.shape
        x = Act(img.contiguous().float().view(-1), B, S, S, 1) if Cc == 1 else nchw_to_nhwc(img)
        feats = []
        cur = None
        for i in range(4):
            if i == 0:
                cur = conv(self.stem, x)
                layernorm(cur, self.stem_ln[0], self.stem_ln[1], 1e-6)
            else:
                t = Act.empty(cur.B, cur.H, cur.W, cur.C, self.device)
                # the normalised map feeds only the 2x2/s2 convolution: split it where it is produced (|LN out| is bounded)
                sp = GEMM_MODE == "split" and cur.C % 32 == 0 and self.down_ln_bound[i] < F16_MAX
                layernorm(cur, self.down_ln[i][0], self.down_ln[i][1], 1e-6, out=t, out_split=sp)
                cur = conv(self.down[i], t)
            y = Act.empty(cur.B, cur.H, cur.W, cur.C, self.device)
            h = Act.empty(cur.B, cur.H, cur.W, 4 * cur.C, self.device)
            for blk in self.stages[i]:
                blk(cur, y, h)
            feats.append(cur)
        return feats

    def _resnet(self, img):
        x = nchw_to_nhwc(img, cpad=4)
        x = conv(self.stem, x, flags=L.KPF_ACT_RELU)
        x = maxpool3x3s2(x)
        feats = []
        for blocks in self.layers:
            for c1, c2, c3, ds in blocks:
                h = conv(c1, x, flags=L.KPF_ACT_RELU)
                idt = conv(ds, x) if ds is not None else x
                if c3 is None:
                    x = conv(c2, h, res=idt, flags=L.KPF_RELU_AFTER_RES)
                else:
                    x = conv(c3, conv(c2, h, flags=L.KPF_ACT_RELU), res=idt, flags=L.KPF_RELU_AFTER_RES)
            feats.append(x)
        return feats

    def __call__(self, img):
        c1, c2, c3, c4 = self._convnext(img) if self.fam == "convnext" else self._resnet(img)
        d = self.dims
        dev = self.device
        B = c1.B

        def level(up, skip, fus, lo, hi, fus_out_c):
            cat = Act.empty(B, hi.H, hi.W, up.cout + skip.cout, dev)
            upsample2x(up(lo), cat.slice(0, up.cout))
            skip(hi, out=cat.slice(up.cout, skip.cout))
            return fus(cat)

        c3f = level(self.up4, self.skip4, self.fus4, c4, c3, d[2])
        c2f = level(self.up3, self.skip3, self.fus3, c3f, c2, d[1])
        feat = level(self.up2, self.skip2, self.fus2, c2f, c1, 128)
        if self.result_emb is not None:
            feat = self.result_emb(feat)
        res = torch.empty(B, 105, feat.H, feat.W, device=dev, dtype=torch.float32)
        conv(self.finals, feat, out_nchw=res)
        return res, feat


# ----------------------------------------------------------------------------------------------------------------
# keypoint-fusion head
# ----------------------------------------------------------------------------------------------------------------
J = 21


def _fold_emb(sd, p):
    """Conv1d(k=1)+BatchNorm1d (model/model.py:254-259) as (W*s [128][Cin], b*s+t) in float64."""
    s, t = bn_scale_shift(sd, p + ".1")
    w = sd[p + ".0.weight"].double()[:, :, 0] * s[:, None]
    return w, sd[p + ".0.bias"].double() * s + t


def pack_tr(sd, p, din, device):
    """Flat fp32 weight image of one KP_Interaction_TR for kpf_tr_encoder_f32 (layout: csrc/kpf_tr.hip)."""
    g = lambda k: sd[p + "." + k].detach().float().cpu()
    parts = [g("bert.img_embedding.weight").t().contiguous(), g("bert.img_embedding.bias"), g("bert.position_embeddings.weight")[:J]]
    for l in range(4):
        q = "bert.encoder.layer.%d." % l
        wqkv = torch.cat([g(q + "attention.self.%s.weight" % n) for n in ("query", "key", "value")], 0)  # 384 x 128
        bqkv = torch.cat([g(q + "attention.self.%s.bias" % n) for n in ("query", "key", "value")], 0)
        parts += [wqkv.t().contiguous(), bqkv, g(q + "attention.output.dense.weight").t().contiguous(), g(q + "attention.output.dense.bias"),
                  g(q + "attention.output.LayerNorm.weight"), g(q + "attention.output.LayerNorm.bias"),
                  g(q + "intermediate.dense.weight").t().contiguous(), g(q + "intermediate.dense.bias"),
                  g(q + "output.dense.weight").t().contiguous(), g(q + "output.dense.bias"),
                  g(q + "output.LayerNorm.weight"), g(q + "output.LayerNorm.bias")]
    parts += [g("cls_head.weight").t().contiguous(), g("cls_head.bias"), g("residual.weight").t().contiguous(), g("residual.bias")]
    flat = torch.cat([x.reshape(-1) for x in parts])
    assert flat.numel() == L.load().kpf_tr_encoder_weight_floats(din), (flat.numel(), din)
    return flat.to(device)


def pack_xattn(sd, p, device):
    """Flat weight image of decoder layer `p` for kpf_xattn_layer_f32."""
    g = lambda k: sd[p + "." + k].detach().float().cpu()
    w, b = g("multihead_attn.in_proj_weight"), g("multihead_attn.in_proj_bias")
    parts = [g("self_posembed.weight")[:J], g("cross_posembed.weight")[:J], w[:128].t().contiguous(), b[:128], w[128:].t().contiguous(), b[128:],
             g("multihead_attn.out_proj.weight").t().contiguous(), g("multihead_attn.out_proj.bias"), g("norm2.weight"), g("norm2.bias"),
             g("linear1.weight").t().contiguous(), g("linear1.bias"), g("linear2.weight").t().contiguous(), g("linear2.bias"),
             g("norm3.weight"), g("norm3.bias")]
    flat = torch.cat([x.reshape(-1) for x in parts])
    assert flat.numel() == L.load().kpf_xattn_weight_floats()
    return flat.to(device)


class FusionBlockPlan:
    """Block_KPFusion (model/model.py:207-351) in kernel layouts."""

    def __init__(self, sd, p, device):
        f64 = lambda k: sd[p + "." + k].double()
        wf, bf = _fold_emb(sd, p + ".pcl_feat_emb")
        wx, bx = _fold_emb(sd, p + ".pcl_xyz_emb")
        wp, bp = _fold_emb(sd, p + ".pcl_pose_emb")
        z4 = torch.zeros(128, 4, dtype=torch.float64, device=wf.device)
        self.e_point = PackedConv(torch.cat([wf, wx, wp, z4], 1), bf + bx + bp, device)  # K = 128+3+105+4 = 240
        wr, br = _fold_emb(sd, p + ".pcl_feat_emb_RGB")
        self.e_rgb = PackedConv(wr, br, device)
        wj, bj = _fold_emb(sd, p + ".joint_feat_emb")
        wjx, bjx = _fold_emb(sd, p + ".joint_xyz_emb")
        z1 = torch.zeros(128, 1, dtype=torch.float64, device=wf.device)
        self.e_joint = PackedConv(torch.cat([wj, wjx, z1], 1), bj + bjx, device)  # K = 132
        self.desa1, self.desa2 = [], []
        fa = p + ".FA"
        for i in range(3):
            sl, tl = bn_scale_shift(sd, fa + ".bn_l0_blocks.%d" % i)
            sf, tf = bn_scale_shift(sd, fa + ".bn_f0_blocks.%d" % i)
            wl = sd[fa + ".conv_l0_blocks.%d.weight" % i].double()[:, :, 0, 0] * sl[:, None]
            wfe = sd[fa + ".conv_f0_blocks.%d.weight" % i].double()[:, :, 0, 0] * sf[:, None]
            b1 = sd[fa + ".conv_l0_blocks.%d.bias" % i].double() * sl + tl + sd[fa + ".conv_f0_blocks.%d.bias" % i].double() * sf + tf
            self.desa1.append(PackedConv(torch.cat([wfe, wl, z1], 1), b1, device))  # rows of G: [feat 128 | xyz 3 | 0]
            self.desa2.append(PackedConv(sd[fa + ".conv_blocks.%d.0.weight" % i], sd[fa + ".conv_blocks.%d.0.bias" % i], device,
                                         fold_bn=bn_scale_shift(sd, fa + ".bn_blocks.%d.0" % i)))
        self.desa_fusion = PackedConv(sd[fa + ".fusion.0.weight"], sd[fa + ".fusion.0.bias"], device, fold_bn=bn_scale_shift(sd, fa + ".fusion.1"))
        self.init_tr = pack_tr(sd, p + ".init_TR", 128, device)
        self.final_tr = pack_tr(sd, p + ".final_TR", 131, device)
        self.xattn = pack_xattn(sd, p + ".crossTR.decoder.3", device)  # layers 0-2 never reach the output
        wa = sd[p + ".atten_spatial.weight"].detach().float()[:, :, 0, 0]
        self.att_feat = PackedConv(wa[:, :128].contiguous(), None, device)
        self.att_wh = wa[:, 128:].contiguous().to(device)
        self.att_b = sd[p + ".atten_spatial.bias"].detach().float().to(device)
        self.wfc = sd[p + ".fc_spatial2joint_feature.weight"].detach().float().reshape(-1).contiguous().to(device)
        self.bfc = sd[p + ".fc_spatial2joint_feature.bias"].detach().float().to(device)
        self.weight_dis = sd[p + ".weight_dis"].detach().float().to(device)

    def __call__(self, ctx, joint_xyz, prev):
        """ctx: per-forward shared tensors (see ModelPlan.forward).  Returns (r3d, r2d, img_feat_j, spatial_weight)."""
        lib = L.load()
        dev, B, N, P, F = ctx["dev"], ctx["B"], ctx["N"], ctx["P"], ctx["F"]
        st = _stream()
        f32 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        A1, A2 = f32(B * N, 240), f32(B * N, 128)
        L.check(lib.kpf_point_assemble_f32(_ptr(ctx["feat_d"].buf), _ptr(ctx["feat_rgb"].buf), _ptr(ctx["img_offset"]), _ptr(ctx["pcl"]),
                                           _ptr(joint_xyz), _ptr(ctx["closeness"]), _ptr(ctx["index"]), _ptr(A1), _ptr(A2), B, N, P,
                                           0.8, st), "kpf_point_assemble_f32")  # pcl_joint2offset(joint_xyz, pcl, 0.8): model/model.py:294 hard-codes the radius
        rows = lambda t, c, ld=None: Act(t.view(-1), 1, 1, t.numel() // (ld or c), c, ld or c)
        X = conv(self.e_point, rows(A1, 240), flags=L.KPF_ACT_RELU)
        conv(self.e_rgb, rows(A2, 128), out=X, res=X, flags=L.KPF_RELU_AFTER_RES)
        JA = f32(B * J, 132)
        L.check(lib.kpf_softmax_pool_f32(_ptr(A1), _ptr(X.buf), _ptr(joint_xyz), _ptr(JA), B, N, st), "kpf_softmax_pool_f32")
        DC = f32(B * J, 512)  # [r=0.1 | r=0.2 | r=0.4 | node_feat]
        dc = rows(DC, 512)
        JF = conv(self.e_joint, rows(JA, 132), out=dc.slice(384, 128), flags=L.KPF_ACT_RELU)
        G = f32(3, B * J * 64, 132)
        idx = torch.empty(3, B * J, 64, device=dev, dtype=torch.int32) if ctx.get("want_aux") else None
        L.check(lib.kpf_ball_group_f32(_ptr(ctx["pcl"]), _ptr(joint_xyz), _ptr(X.buf), C.c_void_p(DC.data_ptr() + 384 * 4), 512, _ptr(G),
                                       _ptr(idx), B, N, 0.1, 0.2, 0.4, st), "kpf_ball_group_f32")
        for r in range(3):
            y = conv(self.desa1[r], rows(G[r], 132), flags=L.KPF_ACT_RELU)
            y = conv(self.desa2[r], y, flags=L.KPF_ACT_RELU)
            L.check(lib.kpf_group_max_f32(_ptr(y.buf), _ptr(DC), B * J, 64, 128, 512, r * 128, st), "kpf_group_max_f32")
        D = conv(self.desa_fusion, dc, flags=L.KPF_ACT_RELU)  # B*J x 128
        h_init, r3d = f32(B, J, 128), f32(B, J, 3)
        FA = f32(B * J, 132)  # final_TR input rows: [r3d 3 | decoder out 128 | pad]
        L.check(lib.kpf_tr_encoder_f32(_ptr(D.buf), 128, 128, _ptr(self.init_tr), _ptr(h_init), _ptr(r3d), _ptr(FA), 132, B, st),
                "kpf_tr_encoder_f32")
        SF = Act(f32(B * P * 24), 1, 1, B * P, J, 24)
        conv(self.att_feat, ctx["feat_rgb_rows"], out=SF)
        sw, Gw = f32(B, J, F, F), f32(B, J, P)
        L.check(lib.kpf_heat_gam_gate_f32(_ptr(r3d), _ptr(ctx["img_xyz"]), _ptr(SF.buf), 24, _ptr(self.att_wh), _ptr(self.att_b),
                                          _ptr(self.weight_dis), _ptr(self.wfc), _ptr(ctx["center"]), _ptr(ctx["M"]), _ptr(ctx["cube"]),
                                          _ptr(ctx["cam"]), _ptr(sw), _ptr(Gw), B, F, ctx["img_size"], ctx["flip"], st), "kpf_heat_gam_gate_f32")
        fj = f32(B, J, 128)
        L.check(lib.kpf_gate_reduce_f32(_ptr(Gw), _ptr(ctx["feat_rgb"].buf), _ptr(self.bfc), _ptr(prev), _ptr(fj), B, P, st),
                "kpf_gate_reduce_f32")
        L.check(lib.kpf_xattn_layer_f32(_ptr(fj), _ptr(h_init), _ptr(self.xattn), _ptr(FA), 132, 3, B, st), "kpf_xattn_layer_f32")
        h_fin, r2d = f32(B, J, 128), f32(B, J, 3)
        L.check(lib.kpf_tr_encoder_f32(_ptr(FA), 132, 131, _ptr(self.final_tr), _ptr(h_fin), _ptr(r2d), None, 0, B, st), "kpf_tr_encoder_f32")
        if ctx.get("want_aux"):
            ctx.setdefault("aux", []).append(dict(X=X.buf.view(B, N, 128), A1=A1.view(B, N, 240), JA=JA.view(B, J, 132), D=D.buf.view(B, J, 128),
                                                  ball_idx=idx, h_init=h_init, fj=fj, dec=FA.view(B, J, 132)[:, :, 3:131], Gw=Gw))
        return r3d, r2d, fj, sw


class ModelPlan:
    """All kernel-layout weights of one KPFusion instance on one device, and the forward schedule
    (model/model.py:395-426)."""

    def __init__(self, sd, net, device, precision="f32"):
        self.net, self.device, self.precision = net, device, precision
        if precision == "f32":
            self.backbone_d = UNetPlan(sd, "backbone_d", net, device)
            self.backbone_rgb = UNetPlan(sd, "backbone_rgb", net, device)
        else:  # 16-bit storage in the backbones (engine16.py); the fusion head below stays fp32
            from .engine16 import UNetPlan16
            self.backbone_d = UNetPlan16(sd, "backbone_d", net, device, precision)
            self.backbone_rgb = UNetPlan16(sd, "backbone_rgb", net, device, precision)
        self.blocks = [FusionBlockPlan(sd, "block%d" % i, device) for i in (1, 2)]
        self._graphs = {}  # (B, S, N, img_size, flip, kernel) -> (hipGraph, static inputs, static outputs)
        self._graph_lock = __import__("threading").Lock()  # a graph's static buffers are shared by every caller of this plan
        self._side = None  # second HIP stream: the RGB backbone runs beside the depth backbone
        self.serial_streams = False  # profiling aid: issue both backbones on one stream so per-kernel timings are not shared

    def backbones(self, img, img_rgb):
        """Both UNet streams (model/model.py:397-398).  They are independent, so the RGB stream is issued on a second HIP
        stream: its small low-resolution layers (fewer tiles than CUs) fill the CUs the depth stream leaves idle."""
        cur = torch.cuda.current_stream(self.device)
        capturing = torch.cuda.is_current_stream_capturing()
        if self.serial_streams:
            return self.backbone_d(img), self.backbone_rgb(img_rgb)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out_rgb = self.backbone_rgb(img_rgb)
        out_d = self.backbone_d(img)
        cur.wait_stream(side)  # (under graph capture the fork / join become graph dependencies: two parallel branches)
        if not capturing:
            for t in (out_rgb[0], out_rgb[1].buf):  # allocated on the side stream, consumed on the caller's stream
                t.record_stream(cur)
        return out_d, out_rgb

    def backbones_graphed(self, img, img_rgb, slot=0):
        """backbones() replayed from a captured hipGraph (one graph per input shape; both streams are captured as parallel
        branches).  One submission per step instead of ~250 launches: the step no longer depends on the host keeping up.
        Inputs are copied into the graph's static buffers; the returned tensors are the graph's own and are overwritten by the
        next replay (the caller copies what it keeps)."""
        key = ("bb", tuple(img.shape), tuple(img_rgb.shape), slot)  # (slots: independent instances for several batches in flight)
        with self._graph_lock:  # capture, copy-in and replay are one critical section per plan (static buffers are shared; the returned
                                # tensors stay the graph's own: one consumer at a time, see the docstring)
            ent = self._graphs.get(key)
            if ent is None:
                static = [img.detach().float().contiguous().clone(), img_rgb.detach().float().contiguous().clone()]
                cur = torch.cuda.current_stream(self.device)
                warm = torch.cuda.Stream(device=self.device)
                warm.wait_stream(cur)
                with torch.cuda.stream(warm):  # eager warm-up: one-time weight packing / attribute lookups must not happen under capture
                    self.backbones(*static)
                    self.backbones(*static)
                cur.wait_stream(warm)
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self.backbones(*static)
                ent = (graph, static, out)
                self._graphs[key] = ent
            graph, static, out = ent
            static[0].copy_(img)
            static[1].copy_(img_rgb)
            graph.replay()
            return out

    def forward_graphed(self, img_rgb, img, pcl, center, M, cube, cam, kernel, img_size, flip, slot=0):
        """Same as forward(), replayed from a captured hipGraph: at small batch the ~300 launches of one forward are host-bound
        (3.9 ms at B=1 vs ~1 ms of device time), a graph replay costs one submission.  One graph per (input shape, slot); inputs are
        copied into the graph's static buffers, outputs are returned as copies.  Slots are independent instances (own static buffers
        and workspace pool) so that several batches can be in flight on different streams (serving.PipelinedEval)."""
        ins = [t.detach().to(device=self.device, dtype=torch.float32).contiguous() for t in (img_rgb, img, pcl, center, M, cube, cam)]
        key = tuple(tuple(t.shape) for t in ins) + (kernel, img_size, flip, slot)
        with self._graph_lock:  # capture, copy-in, replay and copy-out are one critical section per plan (static buffers are shared)
            ent = self._graphs.get(key)
            if ent is None:
                static = [t.clone() for t in ins]
                cur = torch.cuda.current_stream(self.device)
                warm = torch.cuda.Stream(device=self.device)
                warm.wait_stream(cur)
                with torch.cuda.stream(warm):  # eager warm-up: one-time attribute / symbol lookups must not happen under capture
                    self.forward(*static, kernel, img_size, flip)
                    self.forward(*static, kernel, img_size, flip)
                cur.wait_stream(warm)
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    res, sws, _ = self.forward(*static, kernel, img_size, flip)
                ent = (graph, static, res, sws)
                self._graphs[key] = ent
            graph, static, res, sws = ent
            torch._foreach_copy_(static, ins)  # (one multi-tensor launch for the seven inputs, one for the eight outputs: 15 copy launches were 4 % of a B = 32 step)
            graph.replay()
            outs = [torch.empty_like(t) for t in res + sws]
            torch._foreach_copy_(outs, res + sws)
            return outs[:len(res)], outs[len(res):], None

    def staged_graphs(self, img_rgb, img, pcl, center, M, cube, cam, kernel, img_size, flip, slot=0):
        """The forward as TWO captured hipGraphs over one set of static buffers — A: both backbones, B: everything behind them (reads A's outputs and the
        static inputs) — for serving.PipelinedEval's stage pipeline: A of batch i + 1 replays on one HIP stream beside B of batch i on another.  Returns
        (graph_a, graph_b, static inputs in the order (img_rgb, img, pcl, center, M, cube, cam), results, spatial weights); the caller copies the inputs in,
        orders the replays with events and copies the outputs out.  One entry per (input shapes, slot)."""
        ins = [t.detach().to(device=self.device, dtype=torch.float32).contiguous() for t in (img_rgb, img, pcl, center, M, cube, cam)]
        key = ("staged",) + tuple(tuple(t.shape) for t in ins) + (kernel, img_size, flip, slot)
        with self._graph_lock:
            ent = self._graphs.get(key)
            if ent is None:
                static = [t.clone() for t in ins]
                cur = torch.cuda.current_stream(self.device)
                warm = torch.cuda.Stream(device=self.device)
                warm.wait_stream(cur)
                with torch.cuda.stream(warm):  # eager warm-up: one-time attribute / symbol lookups must not happen under capture
                    self.forward(*static, kernel, img_size, flip)
                    self.forward(*static, kernel, img_size, flip)
                cur.wait_stream(warm)
                torch.cuda.synchronize(self.device)
                s_rgb, s_img, s_pcl, s_center, s_M, s_cube, s_cam = static
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga):
                    bb = self.backbones(s_img, s_rgb)
                with torch.cuda.graph(gb):
                    res, sws, _ = self._head(bb, s_img, s_pcl, s_center, s_M, s_cube, s_cam, kernel, img_size, flip)
                ent = (ga, gb, static, res, sws, bb)  # (bb: graph A's outputs stay referenced — graph B's kernels hold their addresses)
                self._graphs[key] = ent
        return ent[:5], ins

    def forward(self, img_rgb, img, pcl, center, M, cube, cam, kernel, img_size, flip, want_aux=False):
        lib = L.load()
        dev = self.device
        B, _, S, _ = img.shape
        N = pcl.shape[1]
        prep = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        img, img_rgb, pcl, center, M, cube, cam = map(prep, (img, img_rgb, pcl, center, M, cube, cam))
        return self._head(self.backbones(img, img_rgb), img, pcl, center, M, cube, cam, kernel, img_size, flip, want_aux)

    def _head(self, bb, img, pcl, center, M, cube, cam, kernel, img_size, flip, want_aux=False):
        """Everything behind the backbones (model/model.py:399-426): keypoint decode, pixel / point association, the two fusion blocks.  bb = backbones(...)."""
        lib = L.load()
        dev = self.device
        B, _, S, _ = img.shape
        N = pcl.shape[1]
        (img_offset, feat_d), (img_offset_rgb, feat_rgb) = bb
        F = feat_d.H
        P = F * F
        st = _stream()
        f32 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        joint_uvd, joint_xyz = f32(B, J, 3), f32(B, J, 3)
        M = crop_inverse(M)  # from here on M holds M^-1 (the only form the geometry kernels use)
        L.check(lib.kpf_offset2joint_f32(_ptr(img_offset), _ptr(img), _ptr(center), _ptr(M), _ptr(cube), _ptr(cam), _ptr(joint_uvd),
                                         _ptr(joint_xyz), B, S, F, kernel, img_size, flip, st), "kpf_offset2joint_f32")
        closeness, img_xyz = f32(B, N, 4), f32(B, P, 3)
        index = torch.empty(B, N, 4, device=dev, dtype=torch.int32)
        L.check(lib.kpf_img2pcl_top4_f32(_ptr(pcl), _ptr(img), _ptr(center), _ptr(M), _ptr(cube), _ptr(cam), _ptr(closeness), _ptr(index),
                                         _ptr(img_xyz), B, N, S, F, img_size, flip, st), "kpf_img2pcl_top4_f32")
        ctx = dict(dev=dev, B=B, N=N, P=P, F=F, kernel=kernel, img_size=img_size, flip=flip, feat_d=feat_d, feat_rgb=feat_rgb,
                   feat_rgb_rows=Act(feat_rgb.buf, 1, 1, B * P, 128), img_offset=img_offset, pcl=pcl, closeness=closeness, index=index,
                   img_xyz=img_xyz, center=center, M=M, cube=cube, cam=cam, want_aux=want_aux)
        result = [img_offset, img_offset_rgb]
        sws = []
        prev = None
        jx = joint_xyz
        for blk in self.blocks:
            r3d, r2d, prev, sw = blk(ctx, jx, prev)
            result += [r3d, r2d]
            sws.append(sw)
            jx = r2d
        if want_aux:
            ctx.update(joint_uvd=joint_uvd, joint_xyz0=joint_xyz)
            return result, sws, ctx
        return result, sws, None
