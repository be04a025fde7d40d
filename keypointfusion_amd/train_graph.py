"""Train-mode forward of KPFusion: the same graph as the inference engine, differentiable.

`KPFusion.forward` under `.train()` comes here.  The reference trains with batch-statistics BatchNorm, dropout in the transformer
layers and autograd through everything (`train.py:209-265`, `model/model.py:287-426`); this module builds that graph on the module's
own Parameters so that `loss.backward()` fills their `.grad`, out of autograd Functions whose forward AND backward are HIP kernels
(keypointfusion_amd/training.py):

  * every convolution / Linear (any stride, padding, channel count: odd widths are zero-padded to whole channel groups) — implicit GEMM
    forward and data gradient, split-K weight gradient; kernel-layout operands persistent across steps (`training.PackCache`);
  * depthwise 7x7, BatchNorm(+ReLU) with batch statistics, LayerNorm, GELU, layer scale + residual, bilinear x2, max-pool, the
    4-nearest-pixel sampling, DESA's grouping, the 21-token attention core (softmax(QK^T)V with dropout), the small-K batched products
    of the joint pooling;
  * the integer decisions (top-4 pixels, ball-query sets) and the detached soft-argmax decode come from the kernels of the inference path
    (`kpf_img2pcl_top4_f32`, `kpf_ball_group_f32`, `kpf_offset2joint_f32`): no gradient flows through them in the reference either
    (`model/model.py:308-309,324,327,404-405`).
What stays on torch ops: residual / embedding adds, ReLU / sigmoid, concatenations, dropout outside the attention, the geometry of the
gate (a few element-wise ops on B x 21 x 1024 maps) and the loss codec.  Every kernel adds in a fixed order: two iterations on the same
batch and weights give the same bits (tests/test_training.py::test_graphed_train_step_replays_are_bit_identical).

BatchNorm: per-replica batch statistics, running statistics updated in place with momentum 0.1 (PyTorch semantics, like the
reference under DataParallel).  Dropout: `module.train_dropout` (default 0.1 = config/config.json hidden_dropout_prob /
attention_probs_dropout_prob and the decoder layer's default, model/transfusion_head.py:95); the parity tests run with 0 because the
reference's random stream cannot be reproduced.  `module.precision`: "f32", or "bf16" mixed precision (fp32 master weights, statistics,
geometry and loss; GEMM operands rounded to bf16).
"""
import contextlib
import math
import os

import torch
import torch.nn.functional as F

from . import lib as L
from .engine import _ptr, _stream, crop_inverse
from .training import (add_relu, attn21, ball_group3, linear_cat, slices_sum_relu, batchnorm_relu_rows, bert_stack21, bmm_small_k, group_max, group_params, linear_slices, conv2d_nhwc, dwconv7_nhwc, gelu_rows, layer_norm_rows, layer_scale_residual, linear_hip, maxpool3x3s2_nhwc,
                       ball_group, drop_add_ln, pair_params, pair_storage, row_gather, self_attention21, upsample2x_nhwc)

_N_STREAMS = int(os.environ.get("KPF_TRAIN_STREAMS", "2"))  # 2: the RGB backbone (forward and backward) on a side stream (unpaired backbones only)
# 1 (default): the two ConvNeXt backbones — same architecture, two weight sets, independent until the fusion head (model/model.py:287-306) — run as ONE
# network over channel-stacked activations [B, H, W, 2C]: every convolution / Linear is a 2-group launch (kpf_conv_desc::groups), every LayerNorm /
# layer scale a 2-set launch, every per-channel kernel (BatchNorm, depthwise 7x7, GELU, bilinear x2) simply sees 2C channels.  Half the launches of
# the backbones, each twice as large — a captured iteration is latency-bound on ~3000 launches of 3-15 us.  0: two separate passes.
PAIR_BACKBONES = bool(int(os.environ.get("KPF_TRAIN_PAIR", "1")))
# 1 (default): the four BERT layers of a 21-token stack as ONE launch each way (training.BertStack21, csrc/kpf_trstack.hip); 0: layer by layer (bert_layer)
TR_FUSED = bool(int(os.environ.get("KPF_TR_FUSED", "1")))
XATTN_FUSED = bool(int(os.environ.get("KPF_XATTN_FUSED", "1")))  # the decoder layer as one launch each way (training.XAttnLayer21); 0: op by op
TR_MMA = os.environ.get("KPF_TR_MMA", "auto")  # GEMM arithmetic of the fused stacks: "auto" = the module's precision; "f32" | "bf16" | "f16" force one
# 1 (default): DESA's three radii as ONE channel-stacked chain — grouped Linears (G = 3), BatchNorm / add + ReLU / group maximum over 3 x 128 channels, one
# grouping launch each way (training.BallGroup3 / LinearSlices / GroupMax) — instead of three chains of small launches; 0: radius by radius
HEAD_MMA = os.environ.get("KPF_HEAD_MMA", "auto")  # products of DESA's wide Linears: "auto" = the module's precision (16-bit operands in the mixed-precision step), "f32"
BN2_FUSED = bool(int(os.environ.get("KPF_BN2_FUSED", "1")))  # DESA's relu(BatchNorm(local) + BatchNorm(features)) as one pass each way (training.Bn2AddRelu)
UNSTACK_FUSED = bool(int(os.environ.get("KPF_UNSTACK_FUSED", "1")))  # the paired backbones' maps -> dense fp32 maps in two launches each way (training.UnstackRows)
DESA_GROUPED = bool(int(os.environ.get("KPF_DESA_GROUPED", "1")))
# 1 (default): the sibling embeddings of a fusion block (Conv1d + BatchNorm1d each, then summed under a ReLU: model/model.py:254-259, 417-422) write ONE
# channel-stacked tensor — one BatchNorm pass over n x 128 channels, one sum + ReLU launch each way (training.LinearCat / SlicesSumRelu); 0: one by one
EMB_GROUPED = bool(int(os.environ.get("KPF_EMB_GROUPED", "1")))
PAIR = "PAIR."  # parameter-name prefix that stands for ("backbone_rgb.", "backbone_d.") while a paired pass is built

J = 21


class TrainGraph:
    def __init__(self, module):
        self.m = module
        self.t = dict(module.named_parameters())
        self.t.update(dict(module.named_buffers()))
        self.pd = float(getattr(module, "train_dropout", 0.1))
        self.momentum = 0.1
        # "f32" | "bf16" | "f16": 16-bit = mixed precision as torch.autocast does it — fp32 master weights and optimiser state, GEMM
        # operands rounded to 16 bits (HIP convolutions / Linears on the 16-bit MFMA, library ops under torch.autocast), fp32
        # normalisation statistics, softmaxes, geometry and loss
        self.prec = getattr(module, "precision", "f32")
        self.cmul = 4 if self.prec == "f32" else 8
        # DEBUG hook, off unless a test sets `module._debug_ball_override` (a list of index tensors): ball-query sets to use instead of
        # the computed ones — the sets are integer decisions taken around network outputs, and a test that compares gradients with the
        # reference's must compare on equal decisions — plus a count of the sets that differed, kept on the device.  The default path
        # touches neither.
        self.G = 1  # > 1 while the paired backbones are being built (names under PAIR resolve to both parameter sets)
        self.attn_calls = 0
        self.nbt = []  # BatchNorm num_batches_tracked counters touched by this forward: incremented by ONE multi-tensor launch at its end
        self.ball_override = list(getattr(module, "_debug_ball_override", None) or [])  # None / empty in every product path
        self.ball_flips = 0
        # kernel-layout operands of every convolution / Linear whose source is parameter storage: persistent buffers, rewritten from the
        # current parameter values by one multi-tensor launch here (training.PackCache); mixed precision rounds the fp32 master to the
        # 16-bit operand in the same kernel (no shadow copies)
        from .training import PackCache
        caches = module.__dict__.setdefault("_pack_cache", {})
        dev = next(module.parameters()).device
        ck = (dev.type, dev.index, self.prec)
        if ck not in caches:
            caches[ck] = PackCache()
        self.packs = caches[ck]
        self.packs.refresh()

    # ---- primitives ---------------------------------------------------------------------------------------------------------------
    def has(self, name):
        return (("backbone_rgb." + name[len(PAIR):]) if name.startswith(PAIR) else name) in self.t

    def w(self, name):
        """A parameter / buffer by name.  Under PAIR: the RGB and the depth backbone's tensors of that name as one group-major tensor
        [2*n, ...] over their (re-homed, adjacent) storage — differentiable towards both (training.pair_params)."""
        if not name.startswith(PAIR):
            return self.t[name]
        a, b = self.t["backbone_rgb." + name[len(PAIR):]], self.t["backbone_d." + name[len(PAIR):]]
        reg = self.m.__dict__.setdefault("_pair_registry", {})
        if a.requires_grad:
            t = pair_params(reg, name, a, b)
        else:
            t = pair_storage(reg, name, a, b)
            if t is None:
                # a buffer (BatchNorm running statistics) is updated IN PLACE by the kernels: a stacked copy would silently drop the update
                raise ValueError("paired backbones: buffer %r has %d elements, not a multiple of 4, so its two halves cannot share one 16-byte-"
                                 "aligned allocation; run this network with KPF_TRAIN_PAIR=0 (two passes)" % (name[len(PAIR):], a.numel()))
        return t.reshape((2 * a.shape[0],) + tuple(a.shape[1:]))

    def wg(self, names):
        """n same-shaped parameters / buffers (DESA's per-radius layers) as one group-major tensor [n * rows, ...] over their (re-homed, adjacent) storage,
        differentiable towards each (training.group_params)."""
        ts = [self.t[n] for n in names]
        reg = self.m.__dict__.setdefault("_pair_registry", {})
        t = group_params(reg, "grp:" + names[0], ts)
        if not ts[0].requires_grad and t.data_ptr() != ts[0].data_ptr():
            raise ValueError("grouped DESA: buffer %r cannot share one aligned allocation with its siblings; run with KPF_DESA_GROUPED=0" % names[0])
        return t.reshape((len(ts) * ts[0].shape[0],) + tuple(ts[0].shape[1:]))

    def bn_g(self, x, names, eps=1e-5, relu=False):
        """BatchNorm (batch statistics) [+ ReLU] of n sibling BatchNorm layers over the channel-stacked rows [M, n * C]: per-channel statistics, so the stacked
        tensor's BatchNorm IS the n layers' BatchNorms (one 3-launch pass instead of n)."""
        y = batchnorm_relu_rows(x, self.wg([n + ".weight" for n in names]), self.wg([n + ".bias" for n in names]), self.wg([n + ".running_mean" for n in names]),
                                self.wg([n + ".running_var" for n in names]), self.momentum, eps, relu, None, False)
        self.nbt += [self.t[n + ".num_batches_tracked"] for n in names]
        return y

    def zeros(self, shape, dev):
        """A constant all-zero tensor, kept on the module (zero channels of padded operands: a fill launch per forward otherwise)."""
        c = self.m.__dict__.setdefault("_zero_cache", {})
        k = (tuple(shape), dev.type, dev.index)
        if k not in c:
            c[k] = torch.zeros(*shape, device=dev)
        return c[k]

    def groups_of(self, name):
        return self.G if name.startswith(PAIR) else 1

    @staticmethod
    def key_of(name):
        """PackCache key of a weight: the parameter name; a paired weight's key carries a ':' (never a DeferredParamGrads candidate: its
        gradient is one tensor for two parameters)."""
        return ("pair:" + name[len(PAIR):]) if name.startswith(PAIR) else name

    def linear(self, x, p_w, p_b=None, gelu_in=False, alias=False, gelu_out=False, g_pre=None):
        w = self.w(p_w)
        b = self.w(p_b) if p_b is not None else None
        G = self.groups_of(p_w)
        n, cin = w.shape[0] // G, w.shape[1]
        assert G == 1 or (n % self.cmul == 0 and cin % self.cmul == 0), "paired Linear layers have whole channel groups"
        # (odd widths — the 3-wide joint heads, the 131-wide input of final_TR, model/model.py:99-104, 349 — are handled inside Conv2dNHWC: one pad
        #  launch on the activation / output gradient, weight gradient trimmed in its reduce; everything stays on the HIP kernels)
        return linear_hip(x.contiguous(), w, b, self.prec, None, self.key_of(p_w), self.packs, G, gelu_in, alias, gelu_out, g_pre)

    def bn(self, x, p, eps=1e-5):
        rm, rv = self.t[p + ".running_mean"], self.t[p + ".running_var"]
        y = F.batch_norm(x, rm, rv, self.t[p + ".weight"], self.t[p + ".bias"], True, self.momentum, eps)
        self.nbt.append(self.t[p + ".num_batches_tracked"])
        return y

    def ln(self, x, p_w, p_b, eps, to_gemm=False):
        """LayerNorm over the last axis on the HIP kernels (C % 4 == 0, C <= 1024), torch otherwise; to_gemm: the output feeds a GEMM, so under
        mixed precision it is written in the 16-bit operand type directly."""
        G = self.groups_of(p_w)
        c = x.shape[-1] // G
        if c % 4 == 0 and c <= 1024:  # (the kernel's shape limits; ConvNeXt-L's 1536-wide norms take the library's LayerNorm)
            from .training import _TDT
            return layer_norm_rows(x, self.w(p_w), self.w(p_b), eps, _TDT[self.prec] if (to_gemm and self.prec != "f32") else None, G)
        if G > 1:
            shp = x.shape
            y = F.layer_norm(x.reshape(shp[:-1] + (G, c)), (c,), None, None, eps) * self.w(p_w).view(G, c) + self.w(p_b).view(G, c)
            return y.reshape(shp)
        return F.layer_norm(x, (c,), self.t[p_w], self.t[p_b], eps)

    @staticmethod
    def gelu(x):
        return gelu_rows(x)

    def attention(self, q, k, v, heads, scale):
        """The 21-token attention core on the HIP kernel; its dropout masks come from the module's device-resident (seed, counter) pair,
        advanced once per forward (so every replay of a captured iteration draws new masks)."""
        self.attn_calls += 1
        return attn21(q, k, v, heads, scale, self.pd, self.rng(q.device), self.attn_calls)

    def rng(self, dev):
        """The module's device-resident (seed, counter) pair behind every HIP dropout mask (None when dropout is off)."""
        if self.pd <= 0:
            return None
        rng = self.m.__dict__.get("_drop_rng")
        if rng is None or rng.device != dev:
            rng = self.m.__dict__["_drop_rng"] = torch.tensor([torch.initial_seed() & 0x7fffffff, 0], dtype=torch.int64, device=dev)
        return rng

    def dropout_add_ln(self, o, h, p_w, p_b, eps):
        """LayerNorm(h + dropout(o)) as one launch each way (training.DropAddLN)."""
        self.attn_calls += 1
        return drop_add_ln(o, h, self.t[p_w], self.t[p_b], eps, self.pd, self.rng(h.device), self.attn_calls)

    def drop(self, x):
        return F.dropout(x, self.pd, True) if self.pd > 0 else x

    # ---- backbones (convNeXT/convnext.py, convNeXT/resnetUnet.py, model/resnet.py, model/resnetUnet.py, model/hourglass.py) -----------
    # Activations stay NHWC ([B, H, W, C] contiguous) from the stem to the heads: the HIP convolutions take and return that layout, a
    # BatchNorm2d over (B, H, W) is F.batch_norm on the [pixels, C] view, the channels-first LayerNorms of the ConvNeXt stem /
    # downsample layers are F.layer_norm over the last axis (same biased variance, eps inside the root), and the ops that want NCHW
    # (bilinear upsample, max-pool, the few library convolutions) see the same memory as a channels_last view: no layout copies.
    def conv_l(self, x, p_w, p_b=None, stride=1, pad=0, res=None):
        """NHWC in / out.  res: added in the convolution's epilogue (the skip path of a Residual block)."""
        w = self.w(p_w)
        b = self.w(p_b) if p_b is not None else None
        G = self.groups_of(p_w)
        cin, k = w.shape[1], w.shape[2]
        assert w.shape[2] == w.shape[3], "square kernels only (every convolution of the model)"
        # any stride / padding: forward, data- and weight-gradient on the HIP kernels
        cpad = (-cin) % self.cmul
        if cpad:  # the 3- / 1-channel images of the stems: zero channels on both operands (the weight's gradient is sliced back)
            assert G == 1
            from .training import PadRowsFn, nchw_to_nhwc_padded, pad_rows
            xn = x.permute(0, 3, 1, 2)
            if x.is_cuda and x.dtype == torch.float32 and not x.requires_grad and xn.is_contiguous():  # the image as the module received it: repack + pad in one launch
                x = nchw_to_nhwc_padded(xn, cpad)
            else:
                x = F.pad(x, (0, cpad))
            if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous():  # OIHW rows [N, cin * k * k]: the zero channels are the rows' tails — one launch each way
                w = PadRowsFn.apply(w.reshape(w.shape[0], cin * k * k), (cin + cpad) * k * k).view(w.shape[0], cin + cpad, k, k)
            else:
                w = F.pad(w, (0, 0, 0, 0, 0, cpad))
        return conv2d_nhwc(x.contiguous(), w, b, stride, pad, self.prec, None, None if cpad else self.key_of(p_w), self.packs, G, res)

    def bn_l(self, x, p, eps=1e-5, relu=False, out16=True, alias=False):
        """BatchNorm2d (batch statistics) [+ ReLU] on NHWC: the HIP kernels for fp32 rows, F.batch_norm otherwise.  alias: returns (y, x')
        with x' = x for a second consumer of x whose gradient the BatchNorm backward kernel then adds itself (training.BatchNormReLU)."""
        shp = x.shape
        rows = x.reshape(-1, shp[-1])
        if shp[-1] % 4 == 0:  # (every BatchNorm of the model; other widths take the library's batch_norm below)
            # mixed precision: 16-bit rows are read as stored (fp32 statistics and arithmetic) and the output is written in the compute
            # type the following convolution reads — no cast passes on either side
            from .training import _TDT
            y = batchnorm_relu_rows(rows, self.w(p + ".weight"), self.w(p + ".bias"), self.w(p + ".running_mean"), self.w(p + ".running_var"),
                                    self.momentum, eps, relu, _TDT[self.prec] if (self.prec != "f32" and out16) else None, alias)
            if p.startswith(PAIR):  # (per-channel statistics: the stacked tensor's BatchNorm IS the two backbones' BatchNorms)
                self.nbt += [self.t[q + p[len(PAIR):] + ".num_batches_tracked"] for q in ("backbone_rgb.", "backbone_d.")]
            else:
                self.nbt.append(self.t[p + ".num_batches_tracked"])
            return (y[0].view(shp), y[1].view(shp)) if alias else y.view(shp)
        assert not p.startswith(PAIR)
        y = self.bn(rows, p, eps).view(shp)
        y = F.relu(y) if relu else y
        return (y, x) if alias else y

    def residual(self, p, x):
        G = self.groups_of(p)
        cin = x.shape[-1] // G
        out, x = self.bn_l(x, p + ".bn1", relu=True, alias=True)  # (x: the skip path's handle on the input, see bn_l)
        out = self.conv_l(out, p + ".conv1.conv.weight", p + ".conv1.conv.bias")
        out = self.bn_l(out, p + ".bn2", relu=True)
        out = self.conv_l(out, p + ".conv2.conv.weight", p + ".conv2.conv.bias", pad=1)
        out = self.bn_l(out, p + ".bn3", relu=True)
        if cin != self.t[("backbone_rgb." + p[len(PAIR):] if p.startswith(PAIR) else p) + ".conv3.conv.weight"].shape[0]:
            x = self.conv_l(x, p + ".skip_layer.conv.weight", p + ".skip_layer.conv.bias")
        return self.conv_l(out, p + ".conv3.conv.weight", p + ".conv3.conv.bias", res=x)  # (conv3 + skip: the add rides in the GEMM's epilogue)

    def convnext_block(self, p, x):
        y, x = dwconv7_nhwc(x.float(), self.w(p + ".dwconv.weight"), self.w(p + ".dwconv.bias"), self.key_of(p + ".dwconv.weight"), self.packs, True)  # (x: the skip path's handle)
        y = self.ln(y, p + ".norm.weight", p + ".norm.bias", 1e-6, to_gemm=True)
        # pwconv1 -> GELU -> pwconv2: the GELU forward rides in pwconv1's epilogue (pre-activation kept as a second output), its backward in pwconv2's
        # data-gradient epilogue — no element-wise pass in either direction
        z, g = self.linear(y, p + ".pwconv1.weight", p + ".pwconv1.bias", gelu_out=True)
        y = self.linear(z, p + ".pwconv2.weight", p + ".pwconv2.bias", gelu_in=True, g_pre=g)
        return layer_scale_residual(x, self.w(p + ".gamma"), y, self.groups_of(p))  # (drop_path_rate is 0 in the reference's constructor call: identity)

    def convnext_features(self, p, x):
        feats = []
        i = 0
        while self.has(p + ".downsample_layers.%d.0.weight" % i):
            q = p + ".downsample_layers.%d" % i
            if i == 0 and p.startswith(PAIR):
                # the stems read different images (3 / 1 channels): two convolutions, their outputs stacked on the channel axis from here on
                rest = q[len(PAIR):]
                x = torch.cat([self.conv_l(xi, pre + rest + ".0.weight", pre + rest + ".0.bias", stride=4) for pre, xi in zip(("backbone_rgb.", "backbone_d."), x)], -1)
                x = self.ln(x, q + ".1.weight", q + ".1.bias", 1e-6)
            elif i == 0:
                x = self.conv_l(x, q + ".0.weight", q + ".0.bias", stride=4)
                x = self.ln(x, q + ".1.weight", q + ".1.bias", 1e-6)
            else:
                x = self.ln(x, q + ".0.weight", q + ".0.bias", 1e-6, to_gemm=True)
                x = self.conv_l(x, q + ".1.weight", q + ".1.bias", stride=2)
            j = 0
            while self.has(p + ".stages.%d.%d.gamma" % (i, j)):
                x = self.convnext_block(p + ".stages.%d.%d" % (i, j), x)
                j += 1
            feats.append(x)
            i += 1
        return feats

    def resnet_features(self, p, x):
        x = self.conv_l(x, p + ".conv1.weight", None, stride=2, pad=3)
        x = self.bn_l(x, p + ".bn1", relu=True)
        x = maxpool3x3s2_nhwc(x)
        feats = []
        for li in range(1, 5):
            j = 0
            while self.has(p + ".layer%d.%d.conv1.weight" % (li, j)):
                q = p + ".layer%d.%d" % (li, j)
                stride = 2 if (li > 1 and j == 0) else 1
                idt = x
                if self.has(q + ".conv3.weight"):
                    out = self.bn_l(self.conv_l(x, q + ".conv1.weight"), q + ".bn1", relu=True)
                    out = self.bn_l(self.conv_l(out, q + ".conv2.weight", None, stride, 1), q + ".bn2", relu=True)
                    out = self.bn_l(self.conv_l(out, q + ".conv3.weight"), q + ".bn3")
                else:
                    out = self.bn_l(self.conv_l(x, q + ".conv1.weight", None, stride, 1), q + ".bn1", relu=True)
                    out = self.bn_l(self.conv_l(out, q + ".conv2.weight", None, 1, 1), q + ".bn2")
                if self.has(q + ".downsample.0.weight"):
                    idt = self.bn_l(self.conv_l(x, q + ".downsample.0.weight", None, stride, 0), q + ".downsample.1")
                x = F.relu(out + idt)
                j += 1
            feats.append(x)
        return feats

    def heads(self, p, feat):
        """The three 1x1 heads (convNeXT/resnetUnet.py:95-97) as ONE convolution over the concatenated weights, output channels
        zero-padded to a multiple of 4 (the forward, data- and weight-gradient kernels take whole channel quads); autograd hands each
        head's slice of the gradient back to its own parameter."""
        G = self.groups_of(p + ".")
        if G > 1:  # paired: [2, n_i, C, 1, 1] per head -> [2, 112, C, 1, 1] (whole 8-channel groups per backbone); returns [B, H, W, 2, n]
            ws = [self.w(p + ".finals.%d.weight" % i) for i in range(3)]
            bs = [self.w(p + ".finals.%d.bias" % i) for i in range(3)]
            ws = [w.view((G, w.shape[0] // G) + tuple(w.shape[1:])) for w in ws]
            bs = [b.view(G, -1) for b in bs]
            n = sum(w.shape[1] for w in ws)
            npad = (n + 7) // 8 * 8
            if npad != n:
                ws.append(self.zeros((G, npad - n) + tuple(ws[0].shape[2:]), ws[0].device))
                bs.append(self.zeros((G, npad - n), bs[0].device))
            w, b = torch.cat(ws, 1), torch.cat(bs, 1)
            y = conv2d_nhwc(feat.contiguous(), w.view((G * npad,) + tuple(w.shape[2:])), b.view(-1), 1, 0, self.prec, None, None, None, G)
            if UNSTACK_FUSED:  # (unet() hands the stacked rows to training.unstack_rows: network g's n channels at column g * npad)
                return y, n, npad
            return y.view(y.shape[:-1] + (G, npad))[..., :n]
        ws = [self.t[p + ".finals.%d.weight" % i] for i in range(3)]
        bs = [self.t[p + ".finals.%d.bias" % i] for i in range(3)]
        return conv2d_nhwc(feat.contiguous(), torch.cat(ws, 0), torch.cat(bs, 0), 1, 0, self.prec)  # (105 output channels: the odd-width form)

    def unet(self, p, img):
        """img NCHW (the module boundary); returns (res, feat) NCHW-shaped views of NHWC memory."""
        convnext = self.has(p + ".backbone.downsample_layers.0.0.weight")
        G = self.groups_of(p + ".")
        x = [i.permute(0, 2, 3, 1) for i in img] if G > 1 else img.permute(0, 2, 3, 1)
        c1, c2, c3, c4 = self.convnext_features(p + ".backbone", x) if convnext else self.resnet_features(p + ".backbone", x)

        up = upsample2x_nhwc  # bilinear x2 on NHWC rows (HIP forward + deterministic gather-form backward)

        def cat(a, b):  # channel concatenation (per group when the activations are channel-stacked: [.., G, Ca] + [.., G, Cb] -> [.., G*(Ca+Cb)])
            if G == 1:
                return torch.cat((a, b), -1)
            s = a.shape[:-1]
            return torch.cat((a.reshape(s + (G, -1)), b.reshape(s + (G, -1))), -1).reshape(s + (-1,))

        c4_up = up(self.residual(p + ".up4.0", c4))
        c3_f = self.residual(p + ".fusion_layer4", cat(c4_up, self.residual(p + ".skip_layer4", c3)))
        c3_up = up(self.residual(p + ".up3.0", c3_f))
        c2_f = self.residual(p + ".fusion_layer3", cat(c3_up, self.residual(p + ".skip_layer3", c2)))
        c2_up = up(self.residual(p + ".up2.0", c2_f))
        feat = self.residual(p + ".fusion_layer2", cat(c2_up, self.residual(p + ".skip_layer2", c1)))
        if convnext:
            feat = self.residual(p + ".result_emb", feat)
        res = self.heads(p, feat)
        if G > 1 and UNSTACK_FUSED:  # -> ((res_rgb, feat_rgb), (res_d, feat_d)): one dense fp32 NHWC map per backbone, two launches for the four maps
            from .training import unstack_rows
            y, n, npad = res
            cf = feat.shape[-1] // G
            # (the offset maps leave as dense NCHW — what the decode kernel and the loss read, and the layout the reference returns them in; the features as NHWC rows)
            return tuple(zip(unstack_rows(y, G, n, npad, True), unstack_rows(feat, G, cf, cf)))
        if G > 1:  # -> ((res_rgb, feat_rgb), (res_d, feat_d)): each backbone's channel slice of the stacked maps, NHWC (strided views)
            fs = feat.view(feat.shape[:-1] + (G, -1))
            return tuple((res[..., g, :], fs[..., g, :]) for g in range(G))
        return res.permute(0, 3, 1, 2), feat.permute(0, 3, 1, 2)  # channels_last views: the consumers below gather pixel rows

    # ---- geometry -------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def uvd2xyz(uvd, center, Minv, cube, cam, img_size, flip):
        """dataloader/loader.py:775-789 with M^-1 given (differentiable in uvd)."""
        B = uvd.shape[0]
        uv = (uvd[:, :, 0:2] + 1) * (img_size / 2)
        d = uvd[:, :, 2:] * (cube.view(B, 1, 3)[:, :, 2:] / 2.0) + center.view(B, 1, 3)[:, :, 2:]
        # [u, v, 1] through the first two rows of M^-1, written out (as a batched 3x3 @ 3x1 product over B x P the library's GEMM takes
        # 440 us for the 1024 pixels of a map: one tiny matrix per batch entry)
        Mi = Minv.view(B, 1, 9)
        tx = Mi[:, :, 0] * uv[:, :, 0] + Mi[:, :, 1] * uv[:, :, 1] + Mi[:, :, 2]
        ty = Mi[:, :, 3] * uv[:, :, 0] + Mi[:, :, 4] * uv[:, :, 1] + Mi[:, :, 5]
        x = (tx - cam[:, 2:3]) * d[:, :, 0] / cam[:, 0:1]
        y = flip * (ty - cam[:, 3:4]) * d[:, :, 0] / cam[:, 1:2]
        xyz = torch.stack((x, y, d[:, :, 0]), -1)
        return (xyz - center.view(B, 1, 3)) / (cube.view(B, 1, 3) / 2.0)

    @staticmethod
    def pixel_grid(Fs, dev):
        c = 2.0 * (torch.arange(Fs, device=dev).float() + 0.5) / Fs - 1.0
        return c.view(1, Fs).expand(Fs, Fs).reshape(-1), c.view(Fs, 1).expand(Fs, Fs).reshape(-1)

    # ---- fusion head (model/model.py:129-351, model/transfusion_head.py:137-173) -----------------------------------------------------------
    def linear_rows(self, rows, w, b, key=None):
        """nn.Linear / Conv1d(k=1) / Conv2d(k=1) over rows [M, Cin] on the HIP GEMM (forward, data- and weight-gradient); input widths
        that are not whole channel groups (3-d coordinates, the 105 pose channels, the 149-channel gate input) see training.Conv2dNHWC."""
        return linear_hip(rows, w, b, self.prec, None, key, self.packs)  # (rows may arrive at the padded width: Conv2dNHWC's odd-width form)

    def emb1d(self, p, x):
        """Conv1d(k=1) + BatchNorm1d over (B, N) (model/model.py:254-259) on rows."""
        B, N, Cin = x.shape
        y = self.linear_rows(x.reshape(B * N, Cin), self.t[p + ".0.weight"].flatten(1), self.t[p + ".0.bias"], p + ".0.weight")
        return self.bn_l(y.view(B, N, -1), p + ".1", out16=False)  # (summed with the other embeddings: kept fp32)

    def emb_group(self, names, xs, n1, n2=0):
        """relu(sum of the first n1 embeddings) [then relu(. + the next n2)] of n sibling Conv1d(k=1) + BatchNorm1d branches over rows [B, N, K_i] (K_i % 4 == 0)."""
        B, N = xs[0].shape[:2]
        wb = []
        for nm in names:
            wb += [self.t[nm + ".0.weight"].flatten(1), self.t[nm + ".0.bias"]]
        y = linear_cat([t.reshape(B * N, t.shape[-1]) for t in xs], [nm + ".0.weight" for nm in names], self.packs, wb)
        Cc = y.shape[1] // len(names)
        if BN2_FUSED and Cc % 4 == 0 and Cc <= 256 and len(names) <= 4:  # the BatchNorms inside the sum + ReLU kernel (training.BnSlicesSumRelu)
            from .training import bn_slices_sum_relu
            bn = [nm + ".1" for nm in names]
            gw = lambda k: self.wg([n_ + k for n_ in bn])
            self.nbt += [self.t[n_ + ".num_batches_tracked"] for n_ in bn]
            return bn_slices_sum_relu(y, gw(".weight"), gw(".bias"), gw(".running_mean"), gw(".running_var"), self.momentum, 1e-5, Cc, n1, n2).view(B, N, -1)
        y = self.bn_g(y, [nm + ".1" for nm in names])
        return slices_sum_relu(y, y.shape[1] // len(names), n1, n2).view(B, N, -1)

    @staticmethod
    def gather_interp(feat, idx, clos, inv=None):
        """feat [B, C, H, W] (channels_last memory: the row view below is free) -> [B, N, C]: the 4 nearest pixels' feature rows
        weighted by clos (model/model.py:368-376).  Feature maps with whole channel quads go through kpf_row_gather_fwd/_bwd_f32 (the
        backward adds in a fixed order); the 21 weight-logit channels (no gradient: detached by the caller) use torch.gather."""
        B, C = feat.shape[:2]
        N, K = idx.shape[1:]
        rows = feat.permute(0, 2, 3, 1).reshape(B, -1, C)
        return row_gather(rows.float(), idx if idx.dtype == torch.int32 else idx.int(), clos, inv)  # (C % 4 != 0 — the 21 weight-logit channels, detached by the caller — and
        #                                                    shapes beyond the backward kernel's limits: torch.gather inside row_gather)

    @staticmethod
    def pcl_joint2offset(joint, pcl, kernel):
        B, Jn, _ = joint.shape
        N = pcl.shape[1]
        off = joint.unsqueeze(2) - pcl.unsqueeze(1)
        dis = torch.sqrt(torch.sum(torch.pow(off, 2), dim=-1))
        unit = (off / (dis.unsqueeze(-1) + 1e-8)).permute(0, 1, 3, 2).reshape(B, Jn * 3, N)
        clos = (kernel - dis) / kernel
        mask = (clos >= 0).float() * (pcl[:, :, 2] < 0.99).float().unsqueeze(1)
        clos = clos * mask
        unit = unit * mask.view(B, Jn, 1, N).expand(B, Jn, 3, N).reshape(B, -1, N)
        return torch.cat((unit, clos), 1).permute(0, 2, 1)

    def pose_tokens(self, pw, joint, pcl, kernel):
        """[B, N, 108]: the point's 21 weight logits, pcl_joint2offset above (63 unit offsets + 21 closenesses) and three zero channels — the
        input of pcl_pose_emb at the width its GEMM reads (kpf_pose_tokens_f32: one launch for ~25 element-wise ones; no gradient, like the
        reference's detach calls, model/model.py:308-316)."""
        B, N, _ = pcl.shape
        Jn = joint.shape[1]
        ld = (5 * Jn + 3) // 4 * 4
        out = torch.empty(B, N, ld, device=pcl.device, dtype=torch.float32)
        with torch.no_grad():
            L.check(L.load().kpf_pose_tokens_f32(_ptr(pw.detach().float().contiguous()), _ptr(joint.detach().float().contiguous()), _ptr(pcl.float().contiguous()),
                                                 _ptr(out), B, N, Jn, ld, float(kernel), _stream()), "kpf_pose_tokens_f32")
        return out

    def ball_query_hip(self, pcl_xyz, node_xyz, pcl_feat, node_feat):
        """The three ball-query index tensors of DESA from the inference path's kernel (kpf_ball_group_f32: same semantics as
        ball_query above, no host synchronisation, so a training iteration can be captured in a hipGraph).  Indices only."""
        with torch.no_grad():
            B, N, _ = pcl_xyz.shape
            dev = pcl_xyz.device
            X = pcl_feat.detach().float().contiguous()   # (the kernel takes fp32 rows; under mixed precision the features are 16-bit)
            JF = node_feat.detach().float().contiguous()
            G = torch.empty(3, B * J * 64, 132, device=dev)
            idx = torch.empty(3, B * J, 64, device=dev, dtype=torch.int32)
            L.check(L.load().kpf_ball_group_f32(_ptr(pcl_xyz.float().contiguous()), _ptr(node_xyz.detach().float().contiguous()), _ptr(X), _ptr(JF), 128, _ptr(G),
                                                _ptr(idx), B, N, 0.1, 0.2, 0.4, _stream()), "kpf_ball_group_f32")
            return [idx[i].view(B, J, 64).long() for i in range(3)]

    def desa(self, p, pcl_feat, node_feat, pcl_xyz, node_xyz):
        B, Jn, C = node_feat.shape
        outs = []
        assert C == 128, "DESA runs on the model's 128-channel features (model/model.py:166-204)"
        grouped = None
        if DESA_GROUPED and not self.ball_override and self.prec == "f32":
            # the three radii channel-stacked (round 6): [B*J*64, 3 * 128] rows through grouped launches
            GF3, GX3, _ = ball_group3(pcl_xyz, node_xyz, pcl_feat, node_feat)
            n3 = lambda fmt: [p + fmt % i for i in range(3)]
            wl = n3(".conv_l0_blocks.%d.weight")
            loc = linear_slices(GX3, wl, self.packs, [self.t[k] for i in range(3) for k in (p + ".conv_l0_blocks.%d.weight" % i, p + ".conv_l0_blocks.%d.bias" % i)])
            wf = n3(".conv_f0_blocks.%d.weight")
            # (mixed precision: the two wide Linears of the chain multiply operands rounded to the step's 16-bit type, fp32 storage and accumulation — KPF_HEAD_MMA)
            from .training import head_mma
            hm = (getattr(self.m, "precision", "f32") if HEAD_MMA == "auto" else HEAD_MMA)
            with head_mma(hm):
                ft = linear_hip(GF3, self.wg(wf).flatten(1), self.wg(n3(".conv_f0_blocks.%d.bias")), "f32", None, "grp:" + wf[0], self.packs, 3)
            if BN2_FUSED:  # relu(bn_l0(.) + bn_f0(.)): both normalisations, the sum and the ReLU in one pass over the two pre-activations (training.Bn2AddRelu)
                from .training import bn2_add_relu
                na, nb = n3(".bn_l0_blocks.%d"), n3(".bn_f0_blocks.%d")
                gw = lambda names, k: self.wg([n + k for n in names])
                g = bn2_add_relu(loc, ft, gw(na, ".weight"), gw(na, ".bias"), gw(nb, ".weight"), gw(nb, ".bias"), gw(na, ".running_mean"), gw(na, ".running_var"),
                                 gw(nb, ".running_mean"), gw(nb, ".running_var"), self.momentum, 1e-5)
                self.nbt += [self.t[n + ".num_batches_tracked"] for n in na + nb]
            else:
                loc = self.bn_g(loc, n3(".bn_l0_blocks.%d"))
                ft = self.bn_g(ft, n3(".bn_f0_blocks.%d"))
                g = add_relu(loc, ft)
            wb = n3(".conv_blocks.%d.0.weight")
            with head_mma(hm):
                g = linear_hip(g, self.wg(wb).flatten(1), self.wg(n3(".conv_blocks.%d.0.bias")), "f32", None, "grp:" + wb[0], self.packs, 3)
            if BN2_FUSED:  # BatchNorm + ReLU + the maximum over the ball in one pass (training.BnReluGroupMax): the normalised tensor is never written
                from .training import bn_relu_group_max
                nm = n3(".bn_blocks.%d.0")
                gw = lambda k: self.wg([n + k for n in nm])
                mx = bn_relu_group_max(g, gw(".weight"), gw(".bias"), gw(".running_mean"), gw(".running_var"), self.momentum, 1e-5, 64).view(B, Jn, 3 * C)
                self.nbt += [self.t[n + ".num_batches_tracked"] for n in nm]
            else:
                g = self.bn_g(g, n3(".bn_blocks.%d.0"), relu=True)
                mx = group_max(g, 64).view(B, Jn, 3 * C)  # == cat of the three radii's maxima on the channel axis
            cat = torch.cat((mx, node_feat), -1).reshape(B * Jn, -1)  # rows of 512
            y = self.linear_rows(cat, self.t[p + ".fusion.0.weight"].flatten(1), self.t[p + ".fusion.0.bias"], p + ".fusion.0.weight")
            return self.bn_l(y, p + ".fusion.1", relu=True, out16=False).view(B, Jn, -1)
        if self.ball_override:  # the debug hook of the gradient-parity tests (given index sets): the op-by-op form below
            xyz = torch.cat((pcl_xyz, node_xyz), 1)
            feat = torch.cat((pcl_feat, node_feat), 1)
            hip_idx = self.ball_query_hip(pcl_xyz, node_xyz, pcl_feat, node_feat)
        else:  # ball query + grouping + centring + scaling of all three radii: one launch (training.BallGroup)
            grouped = ball_group(pcl_xyz, node_xyz, pcl_feat, node_feat)
        for i, r in enumerate((0.1, 0.2, 0.4)):
            if grouped is not None:
                gf, gxr = grouped[2 * i], grouped[2 * i + 1]  # [B*J*64, 128] and [B*J*64, 4] = (offsets / r | 0)
            else:
                idx = hip_idx[i]
                given = self.ball_override.pop(0).to(idx.device).long()  # (KPFusion._debug_ball_override, see __init__)
                self.ball_flips = self.ball_flips + (given != idx).any(-1).sum()  # a device scalar: no host synchronisation here either
                flat = given.reshape(B, Jn * 64)
                gx = torch.gather(xyz, 1, flat.unsqueeze(-1).expand(-1, -1, 3)).view(B, Jn, 64, 3) - node_xyz.unsqueeze(2)
                gf = row_gather(feat.float(), flat.int().unsqueeze(-1)).view(B, Jn, 64, C) - node_feat.unsqueeze(2)  # group_points, gather-form backward
                gxr = (gx / r).reshape(-1, 3)
            # the three 1x1 Conv2d + BatchNorm2d of a scale (model/model.py:176-192) on rows [B*J*64, .]
            q = lambda name, k: self.t[p + ".%s.%d%s" % (name, i, k)]

            loc = self.bn_l(self.linear_rows(gxr, q("conv_l0_blocks", ".weight").flatten(1), q("conv_l0_blocks", ".bias"),
                                             p + ".conv_l0_blocks.%d.weight" % i), p + ".bn_l0_blocks.%d" % i, out16=False)
            ft = self.bn_l(self.linear_rows(gf.reshape(-1, C), q("conv_f0_blocks", ".weight").flatten(1), q("conv_f0_blocks", ".bias"),
                                            p + ".conv_f0_blocks.%d.weight" % i), p + ".bn_f0_blocks.%d" % i, out16=False)
            g = add_relu(loc, ft)
            g = self.bn_l(self.linear_rows(g, q("conv_blocks", ".0.weight").flatten(1), q("conv_blocks", ".0.bias"), p + ".conv_blocks.%d.0.weight" % i),
                          p + ".bn_blocks.%d.0" % i, relu=True)
            outs.append(g.view(B, Jn, 64, -1).max(2)[0])  # B x J x 128
        outs.append(node_feat)
        cat = torch.cat(outs, -1).reshape(B * Jn, -1)  # rows of 512
        y = self.linear_rows(cat, self.t[p + ".fusion.0.weight"].flatten(1), self.t[p + ".fusion.0.bias"], p + ".fusion.0.weight")
        return self.bn_l(y, p + ".fusion.1", relu=True, out16=False).view(B, Jn, -1)

    def bert_layer(self, p, h, heads=4):
        B, T, C = h.shape
        hd = C // heads

        assert T == 21 and hd == 32, "the fusion head's stacks are 21 tokens x 4 heads x 32 (config/config.json)"
        # q | k | v as one projection, the attention core on its column slices (dropout on the probabilities inside), one data-gradient GEMM back
        names = tuple(p + ".attention.self.%s.weight" % n for n in ("query", "key", "value"))
        wb = [self.t[p + ".attention.self.%s.%s" % (n, k)] for n in ("query", "key", "value") for k in ("weight", "bias")]
        self.attn_calls += 1
        # (h / h1 reach their residual adds through the aliases the first consumer returns: the residual's gradient is folded into that consumer's
        #  data-gradient GEMM instead of a separate accumulation launch)
        ctx, h = self_attention21(h, *wb, names, self.packs, heads, 1.0 / math.sqrt(hd), self.pd, self.rng(h.device), self.attn_calls)
        o = self.linear(ctx, p + ".attention.output.dense.weight", p + ".attention.output.dense.bias")
        h1 = self.dropout_add_ln(o, h, p + ".attention.output.LayerNorm.weight", p + ".attention.output.LayerNorm.bias", 1e-12)
        it, g, h1 = self.linear(h1, p + ".intermediate.dense.weight", p + ".intermediate.dense.bias", alias=True, gelu_out=True)
        o2 = self.linear(it, p + ".output.dense.weight", p + ".output.dense.bias", gelu_in=True, g_pre=g)  # output.dense(gelu(.)): GELU in both GEMMs' epilogues
        return self.dropout_add_ln(o2, h1, p + ".output.LayerNorm.weight", p + ".output.LayerNorm.bias", 1e-12)

    def kp_interaction_tr(self, p, x):
        T = x.shape[1]
        from .training import PrefixRows
        if TR_FUSED and T == 21 and self.has(p + ".bert.encoder.layer.3.output.LayerNorm.weight") and not self.has(p + ".bert.encoder.layer.4.output.LayerNorm.weight"):
            # embedding Linear, then [+ position, embedding dropout, four layers] as one launch each way
            from .training import BertStack21
            e = self.linear(x, p + ".bert.img_embedding.weight", p + ".bert.img_embedding.bias")
            names = [p + ".bert.encoder.layer.%d.%s" % (l, k) for l in range(4) for k in BertStack21.ORDER]
            call0 = self.attn_calls + 1
            self.attn_calls += 13
            # (mixed precision: the stack's products take 16-bit operands like the backbones' — what autocast gives the reference's Linears; KPF_TR_MMA=f32 keeps fp32)
            mma = getattr(self.m, "precision", "f32") if TR_MMA == "auto" else TR_MMA
            # (the whole position table goes in: the stack reads its first 21 rows and hands back the table's gradient through the grouped column-sum launch)
            h = bert_stack21(e, self.t[p + ".bert.position_embeddings.weight"], names, self.packs, self.pd, self.rng(x.device), call0, [self.t[n] for n in names], mma)
        else:
            h = self.linear(x, p + ".bert.img_embedding.weight", p + ".bert.img_embedding.bias") + PrefixRows.apply(self.t[p + ".bert.position_embeddings.weight"], T)
            h = self.drop(h)  # TR_Encoder applies the embedding dropout (model/model.py:84)
            for l in range(4):
                h = self.bert_layer(p + ".bert.encoder.layer.%d" % l, h)
        score = self.linear(h, p + ".cls_head.weight", p + ".cls_head.bias") + self.linear(x, p + ".residual.weight", p + ".residual.bias")
        return h, score

    def decoder_layer(self, p, query, key, heads=4):
        B, T, C = query.shape
        hd = C // heads
        from .training import PrefixRows, SplitRows
        if XATTN_FUSED and T == 21 and C == 128:  # the whole layer as one launch each way (training.XAttnLayer21)
            from .training import XAttnLayer21, xattn_layer21
            names = [p + "." + k for k in XAttnLayer21.ORDER]
            call0 = self.attn_calls + 1
            self.attn_calls += 4
            mma = getattr(self.m, "precision", "f32") if TR_MMA == "auto" else TR_MMA
            return xattn_layer21(query, key, self.t[p + ".self_posembed.weight"], self.t[p + ".cross_posembed.weight"], names,
                                 self.packs, self.pd, self.rng(query.device), call0, [self.t[n] for n in names], mma)
        qe = query + PrefixRows.apply(self.t[p + ".self_posembed.weight"], T)
        ke = key + PrefixRows.apply(self.t[p + ".cross_posembed.weight"], T)
        ipw = p + ".multihead_attn.in_proj_weight"
        Wq, Wk, Wv = SplitRows.apply(self.t[ipw], 3)  # (views of the packed parameter; their gradients come back as one concatenation)
        bq, bk, bv = SplitRows.apply(self.t[p + ".multihead_attn.in_proj_bias"], 3)
        q = linear_hip(qe.contiguous(), Wq, bq, self.prec, None, ipw + ":q", self.packs)
        k = linear_hip(ke.contiguous(), Wk, bk, self.prec, None, ipw + ":k", self.packs)
        v = linear_hip(ke.contiguous(), Wv, bv, self.prec, None, ipw + ":v", self.packs)
        assert T == 21 and hd == 32
        # (model/transfusion_head.py:468 scales q by 1/sqrt(hd) before the product: the same factor on the logits, inside the attention kernel)
        ctx = self.attention(q, k, v, heads, float(hd) ** -0.5)
        o = self.linear(ctx, p + ".multihead_attn.out_proj.weight", p + ".multihead_attn.out_proj.bias")
        x = self.dropout_add_ln(o, query, p + ".norm2.weight", p + ".norm2.bias", 1e-5)
        f = self.linear(self.drop(F.relu(self.linear(x, p + ".linear1.weight", p + ".linear1.bias"))), p + ".linear2.weight", p + ".linear2.bias")
        return self.dropout_add_ln(f, x, p + ".norm3.weight", p + ".norm3.bias", 1e-5)

    def block(self, p, img_feat, img_feat_rgb, pcl, joint_xyz, clos, idx, img_offset, prev_feat, img_down, center, Minv, cube, cam,
              img_size, flip):
        from .training import GateMix, GeomGateUVD, JointHeatmap
        B, C, H, W = img_feat.shape
        if self.samples is None:
            # the point samplings of the two feature maps and of the weight logits (model/model.py:368-376) read the same maps with the same indices and
            # weights in both blocks: computed once per forward (the reference recomputes identical tensors), as is the softmax of the logits
            pw = self.gather_interp(img_offset[:, J * 4:], idx, clos).detach()
            self.samples = (self.gather_interp(img_feat, idx, clos, self.idx_inv), self.gather_interp(img_feat_rgb, idx, clos, self.idx_inv), pw,
                            F.softmax(pw.permute(0, 2, 1), -1))
        pf, pf_rgb, pw, att = self.samples
        tok = self.pose_tokens(pw, joint_xyz, pcl, 0.8)  # [pw | pcl_joint2offset(joint, pcl) | 0 0 0]: 105 channels at the GEMM's width 108, no gradient
        grouped = EMB_GROUPED and self.prec == "f32"
        if grouped:  # relu(relu(feat + xyz + pose) + rgb feat): four Linears into one [B*N, 4 x 128] tensor, one BatchNorm pass, one sum + ReLU
            x = self.emb_group([p + ".pcl_feat_emb", p + ".pcl_xyz_emb", p + ".pcl_pose_emb", p + ".pcl_feat_emb_RGB"], [pf, self.pcl4, tok, pf_rgb], 3, 1)
        else:
            x = add_relu(self.emb1d(p + ".pcl_feat_emb", pf), self.emb1d(p + ".pcl_xyz_emb", self.pcl4), self.emb1d(p + ".pcl_pose_emb", tok))
            x = add_relu(x, self.emb1d(p + ".pcl_feat_emb_RGB", pf_rgb))
        jf = bmm_small_k(att, x)
        if grouped:
            from .training import pad_rows
            jf = self.emb_group([p + ".joint_feat_emb", p + ".joint_xyz_emb"], [jf, pad_rows(joint_xyz.detach(), 4)], 2)
        else:
            jf = add_relu(self.emb1d(p + ".joint_feat_emb", jf), self.emb1d(p + ".joint_xyz_emb", joint_xyz.detach()))
        jf = self.desa(p + ".FA", x, jf, pcl, joint_xyz.detach())
        h_init, r3d = self.kp_interaction_tr(p + ".init_TR", jf)
        r3d = r3d.float()  # geometry, heat map and the returned joints are fp32 in every precision
        hm = JointHeatmap.apply(r3d, 0.8, H, 1.0)
        # geometry adjacency map (dataloader/loader.py:791-819): the joints go through the uvd -> xyz map again, like the pixels
        ix = self.img_xyz  # pixel positions of the depth map (dataloader/loader.py:936-955): written by kpf_img2pcl_top4_f32, once per forward
        gam = GeomGateUVD.apply(ix, r3d, self.par16, img_size, flip).view(B, J, H, W)  # (uvd2xyz above, inside the kernel)
        # Conv2d(128 + 21 -> 21, k = 1) (model/model.py:262,336): rows of 149 channels, input and output channel counts zero-padded to
        # whole quads so that forward, data- and weight-gradient all run on the HIP kernels (fixed summation order): the three zero channels ride in
        # the concatenation, the 21 outputs are Conv2dNHWC's odd-width form
        sw_in = torch.cat([img_feat_rgb.permute(0, 2, 3, 1).float(), hm.permute(0, 2, 3, 1), self.zpad3], -1).reshape(B * H * W, -1)
        logits = self.linear_rows(sw_in, self.t[p + ".atten_spatial.weight"].flatten(1), self.t[p + ".atten_spatial.bias"], p + ".atten_spatial.weight")
        # sw = sigmoid(logits) [B, J, H, W]; gw = (sigmoid(weight_dis) * gam + (1 - sigmoid(weight_dis)) * sw) * w_fc: one launch (training.GateMix)
        sw, gw = GateMix.apply(logits, gam.reshape(B, J, H * W), self.t[p + ".weight_dis"], self.t[p + ".fc_spatial2joint_feature.weight"])
        sw = sw.view(B, J, H, W)
        # model/model.py:386-388: fj[b,j,c] = sum_hw relu(g[b,j,hw] * f[b,c,hw]) * w[hw] + bias.  g >= 0 (a convex mix of a positive
        # kernel and a sigmoid), so relu(g*f) == g*relu(f) exactly and the B x J x C x HW intermediate (352 MB at B = 32) collapses
        # into one batched GEMM [J x HW] @ [HW x C] — the same identity the inference kernel (kpf_gate_reduce_f32) uses
        if self.frows is None:  # relu(RGB features) as rows: the same operand in both blocks
            self.frows = F.relu(img_feat_rgb.float()).permute(0, 2, 3, 1).reshape(B, H * W, C)
        fj = bmm_small_k(gw, self.frows) + self.t[p + ".fc_spatial2joint_feature.bias"]
        if prev_feat is not None:
            fj = add_relu(fj, prev_feat, scale=0.5)
        dec = self.decoder_layer(p + ".crossTR.decoder.3", fj, h_init)
        _, r2d = self.kp_interaction_tr(p + ".final_TR", torch.cat([r3d, dec.float()], 2))
        return r3d, r2d.float(), fj, sw.float()

    def forward(self, img_rgb, img, pcl, center, M, cube, cam, kernel, img_size, flip):
        """model/model.py:395-426 in train mode.  Returns ([6 results], [2 spatial weights], None), autograd-connected."""
        lib = L.load()
        dev = img.device
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        img_rgb, img, pcl, center, M, cube, cam = map(f, (img_rgb, img, pcl, center, M, cube, cam))
        return self._forward(lib, dev, img_rgb, img, pcl, center, M, cube, cam, kernel, img_size, flip)

    def _side_stream(self, dev):
        st = self.m.__dict__.get("_train_side_stream")
        if st is None or st.device != dev:
            st = self.m.__dict__["_train_side_stream"] = torch.cuda.Stream(dev)
        return st

    def _forward(self, lib, dev, img_rgb, img, pcl, center, M, cube, cam, kernel, img_size, flip):
        # mixed precision covers the two backbones (where the FLOPs are); the fusion head runs fp32 like the inference path's (its tensors
        # are small and latency-bound, and a consistent type there removes several hundred cast launches per iteration)
        # The two backbones are independent until the fusion head: the RGB one is issued on a side stream (KPF_TRAIN_STREAMS=1: both on
        # the caller's stream).  autograd runs every backward node on its forward's stream, so the two backward chains overlap the same
        # way, inside a captured iteration too (fork / join are event nodes of the graph).  At 128^2 / B = 32 most launches are far
        # too small to fill 256 CUs, which is what the overlap buys back.
        main = torch.cuda.current_stream(dev)
        from .training import _TDT
        amp = (lambda: torch.autocast("cuda", dtype=_TDT[self.prec])) if self.prec != "f32" else contextlib.nullcontext
        paired = PAIR_BACKBONES and self.has("backbone_rgb.backbone.downsample_layers.0.0.weight") and self.has("backbone_d.backbone.downsample_layers.0.0.weight")
        side = self._side_stream(dev) if (_N_STREAMS > 1 and not paired) else main
        if paired:  # both ConvNeXt backbones as one grouped network (PAIR_BACKBONES above)
            self.G = 2
            with amp():
                (img_offset_rgb, img_feat_rgb), (img_offset, img_feat) = self.unet(PAIR[:-1], (img_rgb, img))
            self.G = 1
            # one dense fp32 copy per map (the slices are strided, 16-bit under mixed precision), then the channels_last view the head's consumers expect
            if UNSTACK_FUSED:  # (offset maps: dense NCHW already; features: the NCHW-shaped view of their NHWC rows)
                img_feat_rgb, img_feat = img_feat_rgb.permute(0, 3, 1, 2), img_feat.permute(0, 3, 1, 2)
            else:
                nchw = lambda t: t.float().contiguous().permute(0, 3, 1, 2)
                img_offset_rgb, img_feat_rgb, img_offset, img_feat = nchw(img_offset_rgb), nchw(img_feat_rgb), nchw(img_offset), nchw(img_feat)
        else:
            side.wait_stream(main) if side is not main else None
            with torch.cuda.stream(side), amp():
                img_offset_rgb, img_feat_rgb = self.unet("backbone_rgb", img_rgb)
                img_offset_rgb, img_feat_rgb = img_offset_rgb.float(), img_feat_rgb.float()
            with amp():
                img_offset, img_feat = self.unet("backbone_d", img)
            img_feat = img_feat.float()
        if side is not main:
            main.wait_stream(side)
            for t in (img_offset_rgb, img_feat_rgb):  # produced on the side stream, consumed by the head on the caller's
                t.record_stream(main)
        if self.prec != "f32":
            self.prec, self.cmul = "f32", 4
        img_offset, img_offset_rgb = img_offset.float(), img_offset_rgb.float()  # the dense maps are returned (and decoded) in fp32
        result = [img_offset, img_offset_rgb]
        B, _, S, _ = img.shape
        Fs = img_feat.shape[-1]
        N = pcl.shape[1]
        Minv = crop_inverse(M)
        off_d = img_offset.detach().contiguous()  # model/model.py:404-405: the decode and the pose tokens see detached maps
        off_v = img_offset.detach()  # (the same map as the NCHW-shaped view of its NHWC memory: the pose tokens' gather reads its 21 weight-logit columns in place)
        joint_uvd = torch.empty(B, J, 3, device=dev)
        joint_xyz = torch.empty(B, J, 3, device=dev)
        L.check(lib.kpf_offset2joint_f32(_ptr(off_d), _ptr(img), _ptr(center), _ptr(Minv), _ptr(cube), _ptr(cam), _ptr(joint_uvd), _ptr(joint_xyz),
                                         B, S, Fs, float(kernel), int(img_size), int(flip), _stream()), "kpf_offset2joint_f32")
        clos = torch.empty(B, N, 4, device=dev)
        index = torch.empty(B, N, 4, device=dev, dtype=torch.int32)
        self.img_xyz = torch.empty(B, Fs * Fs, 3, device=dev)
        L.check(lib.kpf_img2pcl_top4_f32(_ptr(pcl), _ptr(img), _ptr(center), _ptr(Minv), _ptr(cube), _ptr(cam), _ptr(clos), _ptr(index), _ptr(self.img_xyz),
                                         B, N, S, Fs, int(img_size), int(flip), _stream()), "kpf_img2pcl_top4_f32")
        idx = index  # (int32, as kpf_img2pcl_top4_f32 wrote it: the gather kernels read it as it is — six casts per iteration fewer)
        self.par16 = torch.cat((Minv.reshape(B, 9)[:, :6], cam.reshape(B, -1)[:, :4], center.reshape(B, 3), cube.reshape(B, 3)), 1)  # GeomGateUVD's per-sample numbers
        from .training import pad_rows
        self.pcl4 = pad_rows(pcl, 4)                                         # the points at the width pcl_xyz_emb's GEMM reads (both blocks)
        self.zpad3 = self.zeros((B, Fs, Fs, 3), dev)                         # the gate input's three zero channels (149 -> 152, both blocks)
        self.frows = self.samples = None
        from .training import ROW_GATHER_MAX_E, ROW_GATHER_MAX_P, row_gather_invert
        self.idx_inv = row_gather_invert(index, Fs * Fs) if (N * 4 <= ROW_GATHER_MAX_E and Fs * Fs <= ROW_GATHER_MAX_P) else None
        img_down = None  # (F.interpolate(img, [Fs, Fs]) in the reference, model/model.py:401: computed there and never read)
        sws = []
        prev = None
        for i in (1, 2):
            r3d, r2d, prev, sw = self.block("block%d" % i, img_feat, img_feat_rgb, pcl, joint_xyz, clos, idx, off_v, prev, img_down, center,
                                            Minv, cube, cam, img_size, flip)
            result += [r3d, r2d]
            sws.append(sw)
            joint_xyz = r2d
        if torch.is_tensor(self.ball_flips):  # (debug hook only)
            self.m.__dict__["_debug_ball_flips"] = self.ball_flips
        self.packs.build_table()  # (operands registered by this forward join the one-launch refresh from the next forward on — and a capture that follows)
        if self.nbt:
            torch._foreach_add_(self.nbt, 1)
        if self.pd > 0 and self.m.__dict__.get("_drop_rng") is not None:
            self.m.__dict__["_drop_rng"][1:2].add_(1)  # next forward (or replay): new dropout masks
        return result, sws, None
