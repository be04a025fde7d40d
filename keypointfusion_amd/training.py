"""Training-step pieces of the KPFusion path: the loss codec, the loss schedule of `train.py`, the optimiser
set-up, and autograd Functions that put convolution / Linear forward and data-gradient on the HIP implicit GEMM.  The train-mode
forward that uses them is keypointfusion_amd/train_graph.py (reached through `KPFusion.forward` under `.train()`).

What exists here
  * `SmoothL1Loss`      model/loss.py:3-26 (quadratic below 0.01, linear above; mean over the last dim, then over the rest).
  * `kpfusion_loss`     the stage-typed schedule of train.py:211-261 (stage_type [1,1,2,3,2,3], coord 100, deconv 1, spatial 10, sigma 3/2);
                        ONE autograd node, `FusedLoss` (kpf_dense_loss_* + kpf_loss_tail_*: 4 launches forward, 3 backward), which
                        computes GFM.joint2offset / offset2joint_weight / joint2heatmap (util/generateFeature.py:59-84,166-195,584-600)
                        inside the kernels.  There is no library-op fallback: the torch restatement of the codec and of the schedule is
                        test infrastructure and lives in oracle/train_oracle.py.
  * `make_optimizer`    AdamW(lr 8e-4, wd 0.01) + StepLR(10, 0.1) (train.py:84-91,120; config.py); `FusedAdamW` = the same optimiser
                        stepping all parameters in ~14 launches (kpf_adamw_step_multi), learning rate and step count on the device.
  * `Conv2dNHWC`        torch.autograd.Function: forward and data-gradient on kpf_conv2d_f32 / _h16 (dgrad = the forward kernel on
                        flipped / transposed weights, of the stride-dilated dY for strided convolutions; patchify convolutions: a
                        GEMM + pixel un-shuffle), weight and bias gradient on kpf_conv2d_wgrad_f32 / _h16 (fp32 MFMA, or the 16-bit MFMA
                        fed by transposed LDS reads for 16-bit dY / X; pixel index as the reduction dimension, fixed-order split
                        reduction, fp32 accumulation either way).
  * `DwConv7NHWC`, `BatchNormReLU`, `Upsample2xNHWC`, `MaxPool3x3s2NHWC`, `RowGather`: depthwise 7x7, train-mode BatchNorm(+ReLU),
                        bilinear x2, max-pool and the weighted row gathers with hand-written, run-to-run deterministic backward
                        kernels (csrc/kpf_wgrad.hip, csrc/kpf_train.hip).
  * `GraphedTrainStep`  the whole iteration replayed from captured hipGraphs (single process, or graph A -> bucket all-reduce over
                        RCCL -> graph B for data parallelism).
These are torch tensors in, torch tensors out (autograd and the optimiser are PyTorch-ROCm's: host-side plumbing, as BASELINE.json's
north_star puts it).  Mixed precision ("bf16"): 16-bit GEMM operands, fp32 master weights / statistics / loss / weight gradients.
  * `LayerNormRows`, `GeluRows`, `Attn21`, `BmmSmallK`, `layer_scale_residual`: the element-wise tail of the ConvNeXt block and the
                        21-token stacks' LayerNorm / GELU / attention core on HIP kernels with fixed-order parameter gradients.
Still on torch autograd: the residual / gate arithmetic of the fusion head, hidden dropout, zero-padding of odd channel
counts, gradient accumulation at fan-outs.
"""
import ctypes as C

import os

import torch
import torch.nn.functional as F

from .graphs import prepare_training_graphs

prepare_training_graphs()  # DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 unless the host chose a value: process-wide, training side only (graphs.py)

STAGE_TYPE = (1, 1, 2, 3, 2, 3)  # config.py: depth backbone, RGB backbone, (RGB KFAM, depth KFAM) x 2
COORD_WEIGHT, DECONV_WEIGHT = 100.0, 1.0
SPATIAL_WEIGHT, SPATIAL_EPOCH = (10.0, 10.0, 10.0), (24, 24, 24)
FEATURE_PARA = 0.8  # kernel size of the 'weight_offset' feature


class JointHeatmap(torch.autograd.Function):
    """GFM.joint2heatmap(uvd[..., :2], std, F, sigma) (util/generateFeature.py:584-600) with its gradient towards the joints:
    kpf_joint_heatmap_forward / _backward, one launch each (torch: ~16 element-wise launches forward, ~20 backward).  uvd [B, J, 3]."""

    @staticmethod
    def forward(ctx, uvd, std, size, sigma):
        from . import lib as L
        u = uvd.detach().float().contiguous()
        B, J, _ = u.shape
        hm = torch.empty(B, J, size, size, device=u.device, dtype=torch.float32)
        L.check(L.load().kpf_joint_heatmap_forward(u.data_ptr(), hm.data_ptr(), B, J, size, float(std), float(sigma), torch.cuda.current_stream().cuda_stream),
                "kpf_joint_heatmap_forward")
        ctx.save_for_backward(u)
        ctx.cfg = (size, float(std), float(sigma), uvd.dtype)
        return hm

    @staticmethod
    def backward(ctx, dhm):
        from . import lib as L
        (u,) = ctx.saved_tensors
        size, std, sigma, dt = ctx.cfg
        B, J, _ = u.shape
        d = torch.empty_like(u)
        L.check(L.load().kpf_joint_heatmap_backward(u.data_ptr(), dhm.float().contiguous().data_ptr(), d.data_ptr(), B, J, size, std, sigma,
                                                    torch.cuda.current_stream().cuda_stream), "kpf_joint_heatmap_backward")
        return d.to(dt), None, None, None


class GeomGate(torch.autograd.Function):
    """The geometry adjacency map of a fusion block, gam[b, j, p] = 1 / (10 |pix_xyz[b, p] - joint_xyz[b, j]|^2 + 1)
    (dataloader/loader.py:791-819; model/model.py:318-326), with its gradient towards the joints: kpf_geom_gate_forward / _backward
    (torch: 6 launches forward and ~10 backward over [B, J, P, 3] intermediates).  pix_xyz [B, P, 3] (data), joint_xyz [B, J, 3]."""

    @staticmethod
    def forward(ctx, pix_xyz, joint_xyz):
        from . import lib as L
        ix, jx = pix_xyz.detach().float().contiguous(), joint_xyz.detach().float().contiguous()
        B, P, _ = ix.shape
        J = jx.shape[1]
        gam = torch.empty(B, J, P, device=ix.device, dtype=torch.float32)
        L.check(L.load().kpf_geom_gate_forward(ix.data_ptr(), jx.data_ptr(), gam.data_ptr(), B, J, P, torch.cuda.current_stream().cuda_stream), "kpf_geom_gate_forward")
        ctx.save_for_backward(ix, jx)
        ctx.dt = joint_xyz.dtype
        return gam

    @staticmethod
    def backward(ctx, dgam):
        from . import lib as L
        ix, jx = ctx.saved_tensors
        B, P, _ = ix.shape
        J = jx.shape[1]
        d = torch.empty_like(jx)
        L.check(L.load().kpf_geom_gate_backward(ix.data_ptr(), jx.data_ptr(), dgam.float().contiguous().data_ptr(), d.data_ptr(), B, J, P,
                                                torch.cuda.current_stream().cuda_stream), "kpf_geom_gate_backward")
        return None, d.to(ctx.dt)


class GeomGateUVD(torch.autograd.Function):
    """GeomGate with the joints given as crop coordinates uvd: the uvd -> xyz map of dataloader/loader.py:775-789 (TrainGraph.uvd2xyz, ~25
    element-wise launches each way on [B, 21, 3] tensors) runs inside kpf_geom_gate_uvd_forward / _backward.  par16 [B, 16]: see include/kpf.h."""

    @staticmethod
    def forward(ctx, pix_xyz, joint_uvd, par16, img_size, flip):
        from . import lib as L
        ix, ju, par = pix_xyz.detach().float().contiguous(), joint_uvd.detach().float().contiguous(), par16.detach().float().contiguous()
        B, P, _ = ix.shape
        J = ju.shape[1]
        gam = torch.empty(B, J, P, device=ix.device, dtype=torch.float32)
        ctx.conf = (float(img_size) / 2.0, float(flip))
        L.check(L.load().kpf_geom_gate_uvd_forward(ix.data_ptr(), ju.data_ptr(), par.data_ptr(), gam.data_ptr(), B, J, P, ctx.conf[0], ctx.conf[1],
                                                   torch.cuda.current_stream().cuda_stream), "kpf_geom_gate_uvd_forward")
        ctx.save_for_backward(ix, ju, par)
        ctx.dt = joint_uvd.dtype
        return gam

    @staticmethod
    def backward(ctx, dgam):
        from . import lib as L
        ix, ju, par = ctx.saved_tensors
        B, P, _ = ix.shape
        J = ju.shape[1]
        d = torch.empty_like(ju)
        L.check(L.load().kpf_geom_gate_uvd_backward(ix.data_ptr(), ju.data_ptr(), par.data_ptr(), dgam.float().contiguous().data_ptr(), d.data_ptr(), B, J, P,
                                                    ctx.conf[0], ctx.conf[1], torch.cuda.current_stream().cuda_stream), "kpf_geom_gate_uvd_backward")
        return None, d.to(ctx.dt), None, None, None


class _SmoothL1(torch.autograd.Function):
    """model/loss.py:3-26 as one autograd node: the forward is the reference's own operation sequence (bit-identical values), the
    backward is the closed form dL/dz = scale * (z if |z| < 0.01 else 0.01 sign(z)) in four launches instead of the ~15 that autograd
    generates for the mask / pow / abs chain (ten loss terms per iteration)."""

    @staticmethod
    def forward(ctx, x, y, size_average):
        z = (x - y).float()
        mse_mask = (torch.abs(z) < 0.01).float()
        l1_mask = (torch.abs(z) >= 0.01).float()
        total = torch.mean(0.5 * torch.pow(mse_mask * z, 2) * mse_mask, dim=-1)
        total = total + torch.mean(0.01 * (torch.abs(l1_mask * z) - 0.005) * l1_mask, dim=-1)
        ctx.save_for_backward(z)
        ctx.scale = 1.0 / z.numel() if size_average else 1.0 / z.shape[-1]
        ctx.dtypes = (x.dtype, y.dtype)
        return total.mean() if size_average else total.sum()

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        gz = torch.where(torch.abs(z) < 0.01, z, 0.01 * torch.sign(z)) * (g * ctx.scale)
        return (gz.to(ctx.dtypes[0]) if ctx.needs_input_grad[0] else None, (-gz).to(ctx.dtypes[1]) if ctx.needs_input_grad[1] else None, None)


class SmoothL1Loss(torch.nn.Module):
    """model/loss.py:3-26 (not torch's SmoothL1Loss: the quadratic zone ends at 0.01 and the linear branch is 0.01 (|z| - 0.005))."""

    def __init__(self, size_average=True):
        super().__init__()
        self.size_average = size_average

    def forward(self, x, y):
        assert x.shape == y.shape
        return _SmoothL1.apply(x, y, self.size_average)


class _LayerScaleResidual(torch.autograd.Function):
    """out = x + gamma * y (convNeXT/convnext.py:48-51, drop_path = identity) on kpf_layer_scale_forward / _backward: x / out fp32 (the
    residual stream), y in the GEMM's storage type (no casts under mixed precision); dgamma added in a fixed order."""

    @staticmethod
    def forward(ctx, x, gamma, y, groups=1):
        """groups = G > 1: x / y [..., G*C] channel-stacked, gamma [G*C] group-major (the paired backbones)."""
        from . import lib as L
        ctx.groups = groups
        x, y = x.float().contiguous(), y.contiguous()
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        gm = gamma.detach().float().contiguous()
        assert y.dtype in _KDT and y.shape == x.shape and Cc % 4 == 0
        out = torch.empty_like(x)
        L.check(L.load().kpf_layer_scale_forward(x.data_ptr(), y.data_ptr(), _KDT[y.dtype], gm.data_ptr(), out.data_ptr(), rows, Cc,
                                                 torch.cuda.current_stream().cuda_stream), "kpf_layer_scale_forward")
        ctx.save_for_backward(gm, y)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import lib as L
        lib = L.load()
        gm, y = ctx.saved_tensors
        Cc = y.shape[-1]
        rows = y.numel() // Cc
        g = g.float().contiguous()
        dy = torch.empty_like(y)
        dgamma = torch.empty(Cc, device=y.device, dtype=torch.float32)
        nws = lib.kpf_layer_scale_ws_floats(rows * ctx.groups, Cc)  # (grouped: the kernel walks rows * G rows of C / G)
        ws = torch.empty(nws, device=y.device, dtype=torch.float32)
        st = torch.cuda.current_stream().cuda_stream
        if ctx.groups > 1:
            G = ctx.groups
            grp = DeferredParamGrads.wants_colsum(gm)
            desc = L.ColsumDesc() if grp is not None else None
            L.check(lib.kpf_layer_scale_backward_g(g.data_ptr(), y.data_ptr(), _KDT[y.dtype], gm.data_ptr(), dy.data_ptr(), dgamma.data_ptr(), ws.data_ptr(), nws,
                                                   rows * G, Cc // G, G, C.byref(desc) if desc is not None else None, st), "kpf_layer_scale_backward_g")
            if grp is not None:
                grp.add_colsum(gm, desc, ws, dgamma)
            return g, dgamma, dy, None
        grp = DeferredParamGrads.wants_colsum(gm)
        if grp is not None:  # d gamma: reduced with every other layer's after backward
            desc = L.ColsumDesc()
            L.check(lib.kpf_layer_scale_backward_partial(g.data_ptr(), y.data_ptr(), _KDT[y.dtype], gm.data_ptr(), dy.data_ptr(), dgamma.data_ptr(), ws.data_ptr(), nws,
                                                         rows, Cc, C.byref(desc), st), "kpf_layer_scale_backward_partial")
            grp.add_colsum(gm, desc, ws, dgamma)
        else:
            L.check(lib.kpf_layer_scale_backward(g.data_ptr(), y.data_ptr(), _KDT[y.dtype], gm.data_ptr(), dy.data_ptr(), dgamma.data_ptr(), ws.data_ptr(), nws, rows, Cc,
                                                 st), "kpf_layer_scale_backward")
        return g, dgamma, dy, None


LAYER_SCALE_MAX_C = 1024   # kpf_layer_scale_backward: a lane holds up to 4 channel quads (csrc/kpf_train.hip LN_MAXQ)
ROW_GATHER_MAX_E = 8192    # kpf_row_gather_bwd_f32: entries (R * G) per image that one workgroup sorts (csrc/kpf_train.hip GATHER_MAX_E)
ROW_GATHER_MAX_P = 4096    # ... and source rows per image (GATHER_MAX_P: the 64 x 64 feature map of the wide model)


def layer_scale_residual(x, gamma, y, groups=1):
    """x + gamma * y.  The backward kernel's limit is checked HERE, before autograd records a node: widths it does not cover (ConvNeXt-L's
    1536-channel stage) take the library expression forward and backward instead of failing in backward.  groups: channel-stacked
    parameter sets (the limit then applies to one group's width)."""
    c = x.shape[-1] // groups
    if c > LAYER_SCALE_MAX_C or c % 4:
        return x.float() + gamma * y.float()
    return _LayerScaleResidual.apply(x, gamma, y, groups)


class DenseStageLoss(torch.autograd.Function):
    """(loss_pixel, loss_coord) of one dense stage (train.py:211-224) before their weights: SmoothL1 between the first 4J maps and
    GFM.joint2offset(uvd_gt), and between the masked soft-argmax decode (GFM.offset2joint_weight) and uvd_gt — kpf_dense_loss_forward /
    _backward: one workgroup per (joint, sample), target maps computed on the fly, one kernel each way instead of ~100 element-wise
    launches per stage.  pixel_pd [B, 5J, F, F] (any strides), img [B, 1, S, S], uvd_gt [B, J, 3]."""

    @staticmethod
    def forward(ctx, pixel_pd, img, uvd_gt, kernel_size):
        from . import lib as L
        pd = pixel_pd.float().contiguous()
        im, gt = img.detach().float().contiguous(), uvd_gt.detach().float().contiguous()
        B, ch, Fs, _ = pd.shape
        J = ch // 5
        part = torch.empty(B, J, 2, device=pd.device, dtype=torch.float32)
        L.check(L.load().kpf_dense_loss_forward(pd.data_ptr(), im.data_ptr(), gt.data_ptr(), part.data_ptr(), B, J, Fs, im.shape[-1], float(kernel_size),
                                                torch.cuda.current_stream().cuda_stream), "kpf_dense_loss_forward")
        sums = part.sum((0, 1))
        ctx.save_for_backward(pd, im, gt)
        ctx.ks = float(kernel_size)
        return sums[0] / float(B * 4 * J * Fs * Fs), sums[1] / float(B * J * 3)

    @staticmethod
    def backward(ctx, g_pixel, g_coord):
        from . import lib as L
        pd, im, gt = ctx.saved_tensors
        B, ch, Fs, _ = pd.shape
        g2 = torch.stack((g_pixel, g_coord)).float().contiguous()
        dpd = torch.empty_like(pd)
        L.check(L.load().kpf_dense_loss_backward(pd.data_ptr(), im.data_ptr(), gt.data_ptr(), g2.data_ptr(), dpd.data_ptr(), B, ch // 5, Fs, im.shape[-1], ctx.ks,
                                                 torch.cuda.current_stream().cuda_stream), "kpf_dense_loss_backward")
        return dpd, None, None, None


class FusedLoss(torch.autograd.Function):
    """The whole loss of train.py:211-261 for the reference's stage schedule (STAGE_TYPE) as ONE autograd node: the two dense-stage kernels
    (kpf_dense_loss_forward), the spatial-weight terms and the final combination (kpf_loss_tail_forward: two launches), and three launches
    backward — the library path issued ~260 element-wise kernels each way for a few thousand numbers.  Returns (loss, out16) with out16 the
    named terms the reference logs (layout: include/kpf.h).  epoch: python number (host gate, like the reference's `if`) or a device scalar
    (hipGraph replay: gated on the device)."""

    @staticmethod
    def forward(ctx, img, uvd_gt, xyz_gt, epoch, r0, r1, r2, r3, r4, r5, sw0, sw1):
        from . import lib as L
        lib = L.load()
        dev = r0.device
        st = torch.cuda.current_stream().cuda_stream
        im, ugt, xgt = img.detach().float().contiguous(), uvd_gt.detach().float().contiguous(), xyz_gt.detach().float().contiguous()
        B, ch, Fs, _ = r0.shape
        J = ch // 5
        dense = [r.float().contiguous() for r in (r0, r1)]
        parts = torch.empty(2, B, J, 2, device=dev, dtype=torch.float32)
        for i, pd in enumerate(dense):
            L.check(lib.kpf_dense_loss_forward(pd.data_ptr(), im.data_ptr(), ugt.data_ptr(), parts[i].data_ptr(), B, J, Fs, im.shape[-1], float(FEATURE_PARA), st),
                    "kpf_dense_loss_forward")
        joints = [r.float().contiguous() for r in (r2, r3, r4, r5)]
        on_dev = torch.is_tensor(epoch)
        sws, strides = [], []
        for t, sw in enumerate((sw0, sw1)):
            if sw is not None and not on_dev and epoch > SPATIAL_EPOCH[t]:
                sw = None  # (host gate: the term is absent, train.py:251)
            if sw is not None:
                sw = sw.float()
                sb, sj, sy, sx = sw.stride()
                if sy != Fs * sx or sw.shape != (B, J, Fs, Fs):
                    sw = sw.contiguous()
                    sb, sj, sy, sx = sw.stride()
                strides += [sb, sj, sx]
            else:
                strides += [0, 0, 0]
            sws.append(sw)
        ep = epoch.detach().to(device=dev, dtype=torch.float32).reshape(1) if on_dev else None
        cfg = (C.c_float * 9)(FEATURE_PARA, 3.0, 2.0, COORD_WEIGHT, DECONV_WEIGHT, SPATIAL_WEIGHT[0], SPATIAL_WEIGHT[1], SPATIAL_EPOCH[0], SPATIAL_EPOCH[1])
        j4 = (C.c_void_p * 4)(*[j.data_ptr() for j in joints])
        s2 = (C.c_void_p * 2)(*[None if s is None else s.data_ptr() for s in sws])
        st6 = (C.c_long * 6)(*strides)
        sp_part = torch.empty(2 * B * J, device=dev, dtype=torch.float32)
        out = torch.empty(16, device=dev, dtype=torch.float32)
        L.check(lib.kpf_loss_tail_forward(parts[0].data_ptr(), parts[1].data_ptr(), j4, xgt.data_ptr(), ugt.data_ptr(), s2, st6, None if ep is None else ep.data_ptr(),
                                          cfg, sp_part.data_ptr(), out.data_ptr(), B, J, Fs, st), "kpf_loss_tail_forward")
        ctx.save_for_backward(im, ugt, xgt, out, *dense, *joints, *[s for s in sws if s is not None])
        ctx.have_sw = [s is not None for s in sws]
        ctx.strides = strides
        ctx.dims = (B, J, Fs)
        ctx.in_dtypes = [None if r is None else r.dtype for r in (r0, r1, r2, r3, r4, r5, sw0, sw1)]
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g, _g_out):
        from . import lib as L
        lib = L.load()
        im, ugt, xgt, out, d0, d1, j0, j1, j2, j3, *rest = ctx.saved_tensors
        B, J, Fs = ctx.dims
        st = torch.cuda.current_stream().cuda_stream
        sws = [rest.pop(0) if h else None for h in ctx.have_sw]
        need = ctx.needs_input_grad[4:]
        g = g.float().reshape(1).contiguous()
        dj = [torch.empty_like(j) if need[2 + i] else None for i, j in enumerate((j0, j1, j2, j3))]
        dsw = [torch.empty_like(s) if (s is not None and need[6 + t]) else None for t, s in enumerate(sws)]  # (preserve_format: the strides of sw)
        gd = torch.empty(2, device=g.device, dtype=torch.float32)
        cfg = (C.c_float * 9)(FEATURE_PARA, 3.0, 2.0, COORD_WEIGHT, DECONV_WEIGHT, SPATIAL_WEIGHT[0], SPATIAL_WEIGHT[1], SPATIAL_EPOCH[0], SPATIAL_EPOCH[1])
        j4 = (C.c_void_p * 4)(*[j.data_ptr() for j in (j0, j1, j2, j3)])
        s2 = (C.c_void_p * 2)(*[None if s is None else s.data_ptr() for s in sws])
        dj4 = (C.c_void_p * 4)(*[None if d is None else d.data_ptr() for d in dj])
        ds2 = (C.c_void_p * 2)(*[None if d is None else d.data_ptr() for d in dsw])
        st6 = (C.c_long * 6)(*ctx.strides)
        L.check(lib.kpf_loss_tail_backward(j4, xgt.data_ptr(), ugt.data_ptr(), s2, st6, cfg, out.data_ptr(), g.data_ptr(), dj4, ds2, gd.data_ptr(), B, J, Fs, st),
                "kpf_loss_tail_backward")
        dd = []
        for i, pd in enumerate((d0, d1)):
            if not need[i]:
                dd.append(None)
                continue
            dpd = torch.empty_like(pd)
            L.check(lib.kpf_dense_loss_backward(pd.data_ptr(), im.data_ptr(), ugt.data_ptr(), gd.data_ptr(), dpd.data_ptr(), B, J, Fs, im.shape[-1], float(FEATURE_PARA), st),
                    "kpf_dense_loss_backward")
            dd.append(dpd)
        grads = dd + dj + dsw
        grads = [None if x is None else (x if x.dtype == dt else x.to(dt)) for x, dt in zip(grads, ctx.in_dtypes)]
        return (None, None, None, None, *grads)


def _fused_loss_applies(results, spatial_weight, img, stage_type, l1):
    if l1 is not None or tuple(stage_type) != STAGE_TYPE or len(results) != 6 or len(spatial_weight) != 2:
        return False
    r0 = results[0]
    if not (r0.is_cuda and r0.dim() == 4 and results[1].shape == r0.shape and r0.shape[1] % 5 == 0):
        return False
    Fs, J = r0.shape[-1], r0.shape[1] // 5
    if Fs * Fs > 4096 or img.shape[-1] % Fs or r0.shape[-2] != Fs:
        return False
    if any(tuple(r.shape) != (r0.shape[0], J, 3) for r in results[2:]):
        return False
    return all(s is None or tuple(s.shape) == (r0.shape[0], J, Fs, Fs) for s in spatial_weight)


def kpfusion_loss(results, spatial_weight, img, uvd_gt, xyz_gt, epoch=0, stage_type=STAGE_TYPE, l1=None):
    """The loss of one training iteration, train.py:211-261.  results: the 6 forward outputs, spatial_weight: the 2 spatial weights.
    Returns (loss, parts) with parts a dict of the named scalar terms the reference logs."""
    if _fused_loss_applies(results, spatial_weight, img, stage_type, l1):
        loss, out = FusedLoss.apply(img, uvd_gt, xyz_gt, epoch, *results, *spatial_weight)
        parts = {"loss_pixel_0": out[1], "loss_coord_0": out[2], "loss_pixel_1": out[3], "loss_coord_1": out[4]}
        parts.update({"loss_coord_%d" % i: out[3 + i] for i in range(2, 6)})
        for t, sw in enumerate(spatial_weight):
            if sw is not None and (torch.is_tensor(epoch) or epoch <= SPATIAL_EPOCH[t]):
                parts["loss_spatial_%d" % t] = out[9 + t]
        return loss, parts
    raise ValueError("kpfusion_loss: the fused HIP loss covers the reference's schedule (train.py:211-261: stage_type %s, the module's "
                     "SmoothL1Loss, fp32 CUDA tensors, F*F <= 4096 feature maps) and nothing else; got device %s, stage_type %s, custom l1: %s.  "
                     "(The torch restatement used as its checker lives in oracle/train_oracle.py.)"
                     % (STAGE_TYPE, results[0].device, tuple(stage_type), l1 is not None))


class FusedAdamW(torch.optim.AdamW):
    """torch.optim.AdamW (train.py:84-91) whose step() is kpf_adamw_step_multi: ~4 launches for the ~300 tensors of the model instead of the
    library's ~30 multi-tensor launches (1.2 ms -> 0.3 ms per iteration at 23 M parameters).  Same hyper-parameters, param_groups and
    per-parameter state keys (`step`, `exp_avg`, `exp_avg_sq`) as the base class — state_dict() / load_state_dict() interchange with
    torch.optim.AdamW(capturable=True); `step` is ONE device scalar shared by all parameters of a group (they step together).  The
    learning rate may be a device tensor (hipGraph replay: the scheduler updates it in place).  Anything this kernel does not cover
    (amsgrad, maximize, non-fp32 / non-CUDA parameters, a closure) goes to the base class."""

    def _kpf_ok(self, group):
        if group["amsgrad"] or group["maximize"] or group.get("differentiable") or group.get("foreach"):
            return False
        return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and (p.grad is None or (p.grad.dtype == torch.float32 and not p.grad.is_sparse))
                   for p in group["params"])

    @torch.no_grad()
    def step(self, closure=None, scaler=None):
        """scaler (LossScaler, fp16 training): the gradients are those of loss * scale — checked for inf / nan in one pass, applied as g / scale or not at all,
        then the scale is updated; all on the device (no .item()), so the call can be captured."""
        if closure is not None or not all(self._kpf_ok(g) for g in self.param_groups):
            if scaler is not None:
                raise ValueError("LossScaler needs the HIP AdamW step (fp32 CUDA parameters, no amsgrad / maximize / closure)")
            return super().step(closure)
        from . import lib as L
        lib = L.load()
        if scaler is not None and len(self.param_groups) != 1:
            raise ValueError("LossScaler: one parameter group (train.py:84-91 has one)")
        for group in self.param_groups:
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                if scaler is not None:  # (a silent `continue` left a latched overflow flag and a stale tracker behind)
                    raise ValueError("FusedAdamW.step(scaler=...): no parameter carries a gradient — every scaled step must follow scaler.scale(loss).backward()")
                continue
            dev = live[0].device
            shared = None
            for p in group["params"]:
                if self.state.get(p):
                    shared = self.state[p]["step"]
                    break
            if shared is None:
                shared = torch.zeros((), dtype=torch.float32, device=dev)
            elif not (torch.is_tensor(shared) and shared.is_cuda and shared.dtype == torch.float32):  # (a loaded host / float64 counter)
                shared = torch.as_tensor(float(shared), dtype=torch.float32, device=dev)
            descs = (L.AdamwDesc * len(live))()
            for i, p in enumerate(live):
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = shared
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                d = descs[i]
                d.p, d.g, d.m, d.v, d.n = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
            lr = group["lr"]
            lr_dev = lr if (torch.is_tensor(lr) and lr.is_cuda) else None
            if lr_dev is not None and lr_dev.dtype != torch.float32:
                lr_dev = lr_dev.float()
            b1, b2 = group["betas"]
            st_ = torch.cuda.current_stream().cuda_stream
            if scaler is None:
                L.check(lib.kpf_adamw_step_multi(descs, len(live), None if lr_dev is None else lr_dev.data_ptr(), 0.0 if lr_dev is not None else float(lr),
                                                 shared.data_ptr(), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), st_), "kpf_adamw_step_multi")
                shared.add_(1.0)
            else:
                scaler._to(dev)
                L.check(lib.kpf_grad_finite_check_multi(descs, len(live), scaler.found.data_ptr(), st_), "kpf_grad_finite_check_multi")
                L.check(lib.kpf_adamw_step_multi_scaled(descs, len(live), None if lr_dev is None else lr_dev.data_ptr(), 0.0 if lr_dev is not None else float(lr),
                                                        shared.data_ptr(), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                        scaler.inv_scale.data_ptr(), scaler.found.data_ptr(), st_), "kpf_adamw_step_multi_scaled")
                scaler.skipped.add_(scaler.found)  # (diagnostic counter; the flag is cleared by the update below)
                L.check(lib.kpf_loss_scale_update(scaler.scale_t.data_ptr(), scaler.inv_scale.data_ptr(), scaler.tracker.data_ptr(), scaler.found.data_ptr(),
                                                  shared.data_ptr(), float(scaler.growth_factor), float(scaler.backoff_factor), int(scaler.growth_interval), st_),
                        "kpf_loss_scale_update")
        return None


class LossScaler:
    """Dynamic loss scaling for fp16 mixed-precision training (`KPFusion.precision = "f16"` in train mode), the device-resident counterpart of
    torch.cuda.amp.GradScaler around the reference's `loss.backward(); optimizer.step()` (train.py:262-264):

        scaler = LossScaler()
        scaler.scale(loss).backward()
        opt.step(scaler=scaler)          # FusedAdamW (make_optimizer(..., capturable=True)): check, unscaled step or skip, scale update

    Every quantity (scale, 1 / scale, the inf / nan flag, the growth tracker, the count of skipped steps) is a device tensor and nothing is read back, so
    the three calls can sit inside a captured hipGraph (GraphedTrainStep(..., scaler=...)).  Data-parallel runs check the gradients AFTER the all-reduce:
    an inf or nan on any rank is one on every rank, so all ranks take the same decision without another collective.  bf16 needs no scaler (fp32 exponent)."""

    def __init__(self, init_scale=2.0 ** 14, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, device=None):
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self._init = float(init_scale)
        self._pending_tracker = 0  # growth tracker restored by load_state_dict before a device is known
        self.scale_t = None
        if device is not None:
            self._to(torch.device(device))

    def _to(self, dev):
        """Allocates the device state ONCE.  A captured GraphedTrainStep bakes these tensors' addresses into its graph (loss * scale, the finite check, the
        scaled AdamW, the scale update), so they are never re-homed: later changes (load_state_dict) are made in place, and asking for another device raises."""
        if self.scale_t is None:
            self.scale_t = torch.tensor(self._init, device=dev, dtype=torch.float32)
            self.inv_scale = torch.tensor(1.0 / self._init, device=dev, dtype=torch.float32)
            self.found = torch.zeros((), device=dev, dtype=torch.int32)
            self.tracker = torch.full((), int(self._pending_tracker), device=dev, dtype=torch.int32)
            self.skipped = torch.zeros((), device=dev, dtype=torch.int32)
            self._pending_tracker = 0
        elif self.scale_t.device != dev:
            raise RuntimeError("LossScaler: state lives on %s (possibly inside a captured graph) and cannot move to %s; build one scaler per device" % (self.scale_t.device, dev))

    @property
    def scale_value(self):
        return self.scale_t

    # (named like GradScaler's)
    @property
    def scale_tensor(self):
        return self.scale_t

    def scale(self, loss):
        self._to(loss.device)
        return loss * self.scale_t

    def get_scale(self):
        """host value (a synchronisation: for logging, not inside a captured region)"""
        return float(self.scale_t) if self.scale_t is not None else self._init

    def state_dict(self):
        return {"scale": self.get_scale(), "growth_tracker": int(self.tracker) if self.scale_t is not None else int(self._pending_tracker)}

    def load_state_dict(self, sd):
        """Restores scale and growth tracker IN PLACE: a GraphedTrainStep captured earlier keeps replaying on the same tensors and sees the restored values
        (re-allocating here orphaned the tensors the graph updates, so a resume on a live step was silently ignored)."""
        self._init = float(sd["scale"])
        tracker = int(sd.get("growth_tracker", 0))
        if self.scale_t is None:
            self._pending_tracker = tracker  # applied when the first use allocates the device state
            return
        self.scale_t.fill_(self._init)
        self.inv_scale.fill_(1.0 / self._init)
        self.tracker.fill_(tracker)
        self.found.zero_()


def make_optimizer(params, lr=8e-4, step_size=10, start_epoch=0, capturable=False):
    """train.py:84-91,120 with config.py's defaults: AdamW(weight_decay 0.01) over all parameters + StepLR(step_size, 0.1).
    capturable (hipGraph replay, GraphedTrainStep): FusedAdamW with the learning rate held in a DEVICE scalar — a Python float would be
    baked into the captured kernel arguments and StepLR's decay would never reach the replays; the scheduler updates the tensor in
    place, so the next replay steps with the new rate."""
    params = list(params)
    fused = bool(capturable and params and params[0].is_cuda)
    lr0 = torch.tensor(float(lr), device=params[0].device, dtype=torch.float32) if fused else lr
    cls = FusedAdamW if fused else torch.optim.AdamW
    opt = cls([{"params": params, "initial_lr": lr}], lr=lr0, weight_decay=0.01, capturable=capturable)
    return opt, torch.optim.lr_scheduler.StepLR(opt, step_size=step_size, gamma=0.1, last_epoch=start_epoch)


# ----------------------------------------------------------------------------------------------------------------
# convolution / Linear with forward and data-gradient on the HIP implicit GEMM
# ----------------------------------------------------------------------------------------------------------------
class DevPack:
    """Kernel-layout view of a convolution weight built ON THE DEVICE (training repacks every step: no host round trip, no BatchNorm
    folding): the attributes engine.conv() reads from a PackedConv.  weight OIHW (or [N][K] for Linear) -> rows [N][Kp], k = (ky,kx,c)."""
    split_allowed = False
    ps = pt = None

    def __init__(self, weight, bias, stride=1, pad=0, patchify=False):
        w = weight.detach()
        if w.dim() == 2:
            w = w[:, :, None, None]
        N, Cin, KH, KW = w.shape
        if patchify:
            assert stride == KH == KW and pad == 0
            self.KH, self.KW, self.Cin, self.sh, self.sw, self.ph, self.pw, self.merge = KH, 1, KW * Cin, KH, 1, 0, 0, KW
        else:
            self.KH, self.KW, self.Cin, self.sh, self.sw, self.ph, self.pw, self.merge = KH, KW, Cin, stride, stride, pad, pad, 1
        K = KH * KW * Cin
        self.N, self.K, self.Kp = N, K, (K + 31) // 32 * 32
        wk = w.permute(0, 2, 3, 1).reshape(N, K)
        self._set_rows(wk)
        self.b = bias.detach().float().contiguous() if bias is not None else None  # NULL bias: the kernel adds nothing
        self.tuned = {}

    def _set_rows(self, rows):
        """rows [N][K]: fp32 -> the fp32 kernel operand (K padded to 32); 16-bit (a shadow copy of the master weight, cast once per
        step for the whole model) -> the 16-bit operand (K padded to 64) with no per-layer cast."""
        N, K = rows.shape
        if rows.dtype == torch.float32:
            self.w = rows.contiguous() if self.Kp == K else F.pad(rows, (0, self.Kp - K)).contiguous()
            self.w16 = None
        else:
            kp = (K + 63) // 64 * 64
            self.w = None
            self.w16 = rows.contiguous() if kp == K else F.pad(rows, (0, kp - K)).contiguous()

    @classmethod
    def packed(cls, weight, bias, mode, prec="f32", stride=1, pad=0, patchify=False, n_pad=None):
        """The operand of one launch of kpf_pack_conv_weight (csrc/kpf_train.hip) instead of flip / permute-clone / pad / cast
        expressions: `weight` [N, Cin, KH, KW] (fp32 master or its 16-bit shadow), mode 0 = forward rows, 1 = data-gradient rows of a
        stride-1 (or dilated) convolution, 2 = data-gradient rows of a patchify convolution; `prec` selects the operand type."""
        from . import lib as L
        w = weight.detach()
        if w.dim() == 2:
            w = w[:, :, None, None]
        w = w.contiguous()
        N, Cin, KH, KW = w.shape
        n_pad = N if n_pad is None else n_pad
        self = cls.__new__(cls)
        gran = 32 if prec == "f32" else 64
        if mode == 0:
            if patchify:
                assert stride == KH == KW and pad == 0
                self.KH, self.KW, self.Cin, self.sh, self.sw, self.ph, self.pw, self.merge = KH, 1, KW * Cin, KH, 1, 0, 0, KW
            else:
                self.KH, self.KW, self.Cin, self.sh, self.sw, self.ph, self.pw, self.merge = KH, KW, Cin, stride, stride, pad, pad, 1
            rows, K = n_pad, KH * KW * Cin
            self.N = n_pad
        elif mode == 1:  # transposed convolution of (dilated) dY: Cin output channels, n_pad input channels, mirrored taps, padding KH-1-pad
            self.KH, self.KW, self.Cin, self.sh, self.sw, self.ph, self.pw, self.merge = KH, KW, n_pad, 1, 1, KH - 1 - pad, KW - 1 - pad, 1
            rows, K = Cin, KH * KW * n_pad
            self.N = Cin
        else:  # mode 2 / 3, patchify: dY rows @ [(ky,kx,c)][n] (Cin = 1: the depthwise tap table [KH*KW][C], mode 3 mirrored)
            self.KH, self.KW, self.Cin, self.sh, self.sw, self.ph, self.pw, self.merge = 1, 1, n_pad, 1, 1, 0, 0, 1
            rows, K = KH * KW * Cin, n_pad
            self.N = rows
        self.K = K
        kp = (K + gran - 1) // gran * gran
        self.Kp = (K + 31) // 32 * 32  # (the fp32 descriptor's row length; the 16-bit operand carries its own, see as16)
        tdt = torch.float32 if prec == "f32" else _TDT[prec]
        buf = torch.empty(rows, kp, device=w.device, dtype=tdt)
        L.check(L.load().kpf_pack_conv_weight(w.data_ptr(), _KDT[w.dtype], buf.data_ptr(), _KDT[tdt], N, Cin, KH, KW, mode, n_pad, kp,
                                              torch.cuda.current_stream().cuda_stream), "kpf_pack_conv_weight")
        self.w, self.w16 = (buf, None) if prec == "f32" else (None, buf)
        self.b = bias.detach().float().contiguous() if bias is not None else None
        self.tuned = {}
        return self

    @classmethod
    def from_rows(cls, rows, KH, KW, Cin, pad):
        """Stride-1 convolution whose weight is already in kernel order: rows [N][(ky,kx,c)] (no bias)."""
        self = cls.__new__(cls)
        N, K = rows.shape
        assert K == KH * KW * Cin
        self.KH, self.KW, self.Cin, self.sh, self.sw, self.ph, self.pw, self.merge = KH, KW, Cin, 1, 1, pad, pad, 1
        self.N, self.K, self.Kp = N, K, (K + 31) // 32 * 32
        self._set_rows(rows)
        self.b = None
        self.tuned = {}
        return self

    def flops(self, M):
        return 2.0 * M * self.N * self.K

    def as16(self, tdt):
        """The object engine16.conv16() takes: 16-bit rows [N][Kp64] of the same weights."""
        kp = (self.K + 63) // 64 * 64
        if self.w16 is not None:
            assert self.w16.dtype == tdt and self.w16.shape[1] == kp
            return type("P16", (), {"pc": self, "Kp": kp, "w": self.w16})()
        w = self.w[:, :self.K]
        w16 = (w if kp == self.K else F.pad(w, (0, kp - self.K))).to(tdt).contiguous()
        return type("P16", (), {"pc": self, "Kp": kp, "w": w16})()


_TDT = {"bf16": torch.bfloat16, "f16": torch.float16}


class PackCache:
    """Persistent kernel-layout operands of the training step's convolutions / Linears, refreshed from the parameters by ONE launch per
    iteration (kpf_pack_conv_weights_multi): the state dict keeps the reference's OIHW layout and fp32 master values, the kernels read
    buffers that live across steps.  An operand is registered under a key (parameter name + role) the first time it is needed — packed
    on the spot by kpf_pack_conv_weight — and from the next `refresh()` on it is rewritten together with all the others at the start of
    the forward.  Only operands whose source is parameter storage (stable address) are registered; derived weights (padded / concatenated
    copies) are packed per use.  Mixed precision: the fp32 master is rounded to the 16-bit operand by the same kernel."""

    def __init__(self):
        self.entries = {}
        self.table = None
        self.total_blocks = 0
        self.dirty = False

    def get(self, key, weight, bias, mode, prec, **kw):
        w = weight.detach()
        ent = self.entries.get(key)
        if ent is not None and ent["src"] == w.data_ptr() and ent["shape"] == tuple(w.shape) and ent["prec"] == prec:
            pc = ent["pc"]
        else:
            assert w.is_contiguous(), "PackCache: a registered source must be a contiguous view of parameter storage"
            pc = DevPack.packed(w, None, mode, prec, **kw)
            w4 = w if w.dim() == 4 else w[:, :, None, None]
            buf = pc.w if pc.w is not None else pc.w16
            self.entries[key] = {"src": w.data_ptr(), "shape": tuple(w.shape), "prec": prec, "pc": pc, "keep": w,
                                 "desc": (w.data_ptr(), buf.data_ptr(), w4.shape[0], w4.shape[1], w4.shape[2], w4.shape[3], mode,
                                          kw.get("n_pad") or w4.shape[0], buf.shape[1], buf.shape[0], _KDT[w.dtype], _KDT[buf.dtype])}
            self.dirty = True
        pc.b = bias.detach().float().contiguous() if bias is not None else None
        return pc

    def get_stacked(self, names, weights, biases):
        """q | k | v: three fp32 Linear weights [C, K] (and biases [C]) of the same shape as ONE forward operand [3C][Kp], ONE data-gradient
        operand [K][3C] and one bias vector [3C] — persistent buffers, each parameter packed into its rows / columns / slot by the refresh launch
        (kpf_pack_desc::reserved, mode 4).  Returns a StackedPack (attributes of a DevPack with N = 3C; .dgrad = the data-gradient operand)."""
        from . import lib as L
        key = ("stack",) + tuple(names)
        srcs = tuple(t.detach().data_ptr() for t in list(weights) + list(biases))
        ent = self.entries.get(key + (0, 0))
        if ent is not None and ent["srcs"] == srcs:
            return ent["sp"]
        assert not torch.cuda.is_current_stream_capturing(), "PackCache.get_stacked: register the operand in an eager iteration, before a capture"
        n, (Cn, K) = len(weights), weights[0].shape
        dev = weights[0].device
        assert all(w.shape == (Cn, K) and w.dtype == torch.float32 and w.is_contiguous() for w in weights) and Cn % 32 == 0 and K % 4 == 0
        kp = (K + 31) // 32 * 32
        buf0 = torch.zeros(n * Cn, kp, device=dev)
        buf1 = torch.zeros(K, n * Cn, device=dev)
        bufb = torch.zeros(n * Cn, device=dev)
        sp = type("StackedPack", (), {})()
        sp.KH = sp.KW = sp.sh = sp.sw = sp.merge = 1
        sp.ph = sp.pw = 0
        sp.Cin, sp.K, sp.Kp, sp.N, sp.w, sp.w16, sp.b, sp.ps, sp.pt, sp.tuned, sp.split_allowed = K, K, kp, n * Cn, buf0, None, bufb, None, None, {}, False
        sp.flops = lambda M: 2.0 * M * n * Cn * K
        dg = type("StackedPack", (), {})()
        dg.KH = dg.KW = dg.sh = dg.sw = dg.merge = 1
        dg.ph = dg.pw = 0
        dg.Cin, dg.K, dg.Kp, dg.N, dg.w, dg.w16, dg.b, dg.ps, dg.pt, dg.tuned, dg.split_allowed = n * Cn, n * Cn, n * Cn, K, buf1, None, None, None, None, {}, False
        dg.flops = lambda M: 2.0 * M * n * Cn * K
        sp.dgrad = dg
        new = []
        for i, (w, b) in enumerate(zip(weights, biases)):
            w, b = w.detach(), b.detach()
            new.append((key + (0, i), {"srcs": srcs, "sp": sp, "keep": w, "desc": (w.data_ptr(), buf0.data_ptr() + i * Cn * kp * 4, Cn, K, 1, 1, 0, Cn, kp, Cn, 0, 0)}))
            new.append((key + (1, i), {"keep": w, "desc": (w.data_ptr(), buf1.data_ptr() + i * Cn * 4, Cn, K, 1, 1, 1, Cn, Cn, K, 0, 0, n * Cn)}))
            new.append((key + (4, i), {"keep": b, "desc": (b.data_ptr(), bufb.data_ptr() + i * Cn * 4, Cn, 1, 1, 1, 4, Cn, Cn, 1, 0, 0)}))
        # fill them now (one launch from a temporary table), then they are part of every refresh
        arr = (L.PackDesc * len(new))()
        blk = 0
        for a, (_, e) in zip(arr, new):
            d = e["desc"]
            a.src, a.dst = d[0], d[1]
            a.N, a.Cin, a.KH, a.KW, a.mode, a.n_pad, a.Kp, a.rows = d[2:10]
            a.src_dtype, a.dst_dtype, a.first_block, a.reserved = d[10], d[11], blk, (d[12] if len(d) > 12 else 0)
            blk += int(L.load().kpf_pack_desc_blocks(C.byref(a)))
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        L.check(L.load().kpf_pack_conv_weights_multi(table.data_ptr(), len(new), blk, torch.cuda.current_stream().cuda_stream), "kpf_pack_conv_weights_multi")
        sp._table = table  # (alive until the launch has run)
        for k, e in new:
            self.entries[k] = e
        self.dirty = True
        return sp

    def build_table(self):
        """(Re)build the device-resident descriptor table when operands were registered since the last one — a host -> device upload, so not
        inside a graph capture (TrainGraph calls this at the end of every eager forward; a capture then starts with a current table)."""
        if not self.dirty or not self.entries or torch.cuda.is_current_stream_capturing():
            return
        from . import lib as L
        arr = (L.PackDesc * len(self.entries))()
        blk = 0
        for i, ent in enumerate(self.entries.values()):
            d = ent["desc"]
            arr[i].src, arr[i].dst = d[0], d[1]
            arr[i].N, arr[i].Cin, arr[i].KH, arr[i].KW, arr[i].mode, arr[i].n_pad, arr[i].Kp, arr[i].rows = d[2:10]
            arr[i].src_dtype, arr[i].dst_dtype, arr[i].first_block = d[10], d[11], blk
            arr[i].reserved = d[12] if len(d) > 12 else 0  # (destination row stride of an operand that is a column range of a stacked matrix)
            blk += int(L.load().kpf_pack_desc_blocks(C.byref(arr[i])))  # (the kernel's own rule: LDS-staged forms per operand geometry, csrc/kpf_train.hip)
        dev = next(iter(self.entries.values()))["keep"].device
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.total_blocks, self.dirty = blk, False

    def refresh(self):
        """Rewrite every registered operand from the current parameter values (call at the start of a forward)."""
        if not self.entries:
            return
        from . import lib as L
        lib = L.load()
        st = torch.cuda.current_stream().cuda_stream
        self.build_table()
        if self.dirty:
            if torch.cuda.is_current_stream_capturing():  # no host -> device table upload inside a capture: one launch per operand
                for ent in self.entries.values():
                    d = ent["desc"]
                    if len(d) > 12 or d[6] == 4:
                        raise RuntimeError("PackCache: a stacked operand was registered during a graph capture (run one eager iteration first)")
                    L.check(lib.kpf_pack_conv_weight(d[0], d[10], d[1], d[11], d[2], d[3], d[4], d[5], d[6], d[7], d[8], st), "kpf_pack_conv_weight")
                return
        L.check(lib.kpf_pack_conv_weights_multi(self.table.data_ptr(), len(self.entries), self.total_blocks, st), "kpf_pack_conv_weights_multi")


# 16-bit products for fp32-storage GEMMs (KPF_MMA_BF16 / _F16 of include/kpf.h): a flag word that _conv_any adds to every fp32 launch while a `head_mma(...)` block
# is open — TrainGraph opens one around DESA's wide Linears in the mixed-precision step (what autocast gives those Linears).  Conv2dNHWC records the word of its
# forward and re-opens it for its data-gradient GEMM.
_HEAD_MMA = [0]


class head_mma:
    def __init__(self, prec):
        from . import lib as L
        self.word = {"bf16": L.KPF_MMA_BF16, "f16": L.KPF_MMA_F16}.get(prec, 0) if not isinstance(prec, int) else int(prec)

    def __enter__(self):
        self.old, _HEAD_MMA[0] = _HEAD_MMA[0], self.word
        return self

    def __exit__(self, *exc):
        _HEAD_MMA[0] = self.old
        return False


def _conv_any(pc, x4, prec, out_ld=None, res=None, flags=0, out2=None):
    """engine.conv (fp32) or engine16.conv16 (bf16 / f16) on an NHWC tensor [B, H, W, C]; returns the NHWC output tensor.  A GroupedPack
    (G convolutions over channel-stacked activations, one launch): x4 is [B, H, W, G*Cin], the result [B, OH, OW, G*N].  out_ld > N: the result
    is [B, OH, OW, out_ld] with only the first N channels written."""
    from .engine import Act, conv
    B, H, W, Cc = x4.shape
    G = getattr(pc, "groups", 1)
    if prec == "f32" and _HEAD_MMA[0]:
        flags |= _HEAD_MMA[0]
    if out_ld is not None and out_ld != pc.N:
        assert G == 1 and pc.merge == 1
        tdt, kdt = (None, None) if prec == "f32" else __import__("keypointfusion_amd.engine16", fromlist=["DTYPES"]).DTYPES[prec]
        xb = (x4 if prec == "f32" else x4.to(tdt)).contiguous().view(-1)
        OH, OW = (H + 2 * pc.ph - pc.KH) // pc.sh + 1, (W + 2 * pc.pw - pc.KW) // pc.sw + 1
        ob = torch.empty(B * OH * OW * out_ld, device=x4.device, dtype=xb.dtype)
        oa = Act(ob, B, OH, OW, pc.N, ld=out_ld)
        if prec == "f32":
            conv(pc, Act(xb, B, H, W, Cc), out=oa)
        else:
            from .engine16 import conv16
            conv16(pc.as16(tdt), Act(xb, B, H, W, Cc), kdt, out=oa)
        return ob.view(B, OH, OW, out_ld)
    tdt = kdt = None
    if prec != "f32":
        from .engine16 import DTYPES, conv16
        tdt, kdt = DTYPES[prec]
    xld = Cc
    if prec == "f32" and G == 1 and H == 1 and W == 1 and x4.dtype == torch.float32 and not x4.is_contiguous() and x4.stride(-1) == 1 and x4.stride(0) % 4 == 0 \
            and x4.stride(0) > Cc and x4.data_ptr() % 16 == 0:
        xb, xld = x4, x4.stride(0)  # rows that are a column slice of a wider matrix (BallGroup's outputs): read in place (in_ld), no copy
    else:
        xb = (x4 if prec == "f32" else x4.to(tdt)).contiguous().view(-1)
    if G > 1:
        assert Cc == G * pc.Cin and pc.merge == 1, (Cc, G, pc.Cin)
        OH, OW = (H + 2 * pc.ph - pc.KH) // pc.sh + 1, (W + 2 * pc.pw - pc.KW) // pc.sw + 1
        xa = Act(xb, B, H, W, pc.Cin, ld=Cc)
        ob = torch.empty(B * OH * OW * G * pc.N, device=x4.device, dtype=xb.dtype)
        oa = Act(ob, B, OH, OW, pc.N, ld=G * pc.N)
        ra = None if res is None else Act(res.to(xb.dtype).contiguous().view(-1), B, OH, OW, pc.N, ld=G * pc.N)  # (out = conv + res: the residual epilogue)
        o2 = None if out2 is None else Act(out2.view(-1), B, OH, OW, pc.N, ld=G * pc.N)  # (second output of a GELU epilogue: the pre-activation)
        conv(pc, xa, out=oa, res=ra, flags=flags, out2=o2) if prec == "f32" else conv16(pc.as16(tdt), xa, kdt, out=oa, res=ra, flags=flags, out2=o2)
        return ob.view(B, OH, OW, G * pc.N)
    ra = None
    if res is not None:
        ra = Act(res.to(xb.dtype).contiguous().view(-1), res.shape[0], res.shape[1], res.shape[2], res.shape[3])
    o2 = None if out2 is None else Act(out2.view(-1), out2.shape[0], out2.shape[1], out2.shape[2], out2.shape[3])
    if prec == "f32":
        out = conv(pc, Act(xb, B, H, W, Cc, ld=xld), res=ra, flags=flags, out2=o2)
    else:
        out = conv16(pc.as16(tdt), Act(xb, B, H, W, Cc), kdt, res=ra, flags=flags, out2=o2)
    return out.buf.view(out.B, out.H, out.W, out.C)


def pad_rows(src, width, dtype=None):
    """src [..., C] (any strides on the leading axes collapse to rows of stride src_ld: a contiguous tensor or a column slice of one) -> dense
    [..., width] with zero columns beyond C (width >= C), optionally in another storage type: kpf_pad_rows, ONE launch (F.pad: fill + copy)."""
    from . import lib as L
    Cc = src.shape[-1]
    s2 = src.reshape(-1, Cc)  # (a view whenever the leading axes collapse to one row stride — contiguous tensors, column slices of them)
    if s2.stride(1) != 1:
        s2 = s2.contiguous()
    rows = s2.shape[0]
    ld = s2.stride(0) if rows > 1 else Cc
    dtype = dtype or src.dtype
    out = torch.empty(tuple(src.shape[:-1]) + (width,), device=src.device, dtype=dtype)
    L.check(L.load().kpf_pad_rows(s2.data_ptr(), _KDT[s2.dtype], out.data_ptr(), _KDT[dtype], rows, Cc, ld, width, torch.cuda.current_stream().cuda_stream), "kpf_pad_rows")
    return out


class PadRowsFn(torch.autograd.Function):
    """pad_rows with a gradient: zero columns appended to the last axis in one launch; backward = the dense copy of the gradient's first C columns (one launch).
    (F.pad is a fill + a copy, its backward a slice + a copy.)  Round 6: the stem weights [N, 3 or 1, 4, 4] read as rows [N, cin * 16] and padded to the
    channel group of the GEMM."""

    @staticmethod
    def forward(ctx, src, width):
        ctx.C = src.shape[-1]
        return pad_rows(src, width)

    @staticmethod
    def backward(ctx, g):
        return pad_rows(g[..., :ctx.C], ctx.C), None


def nchw_to_nhwc_padded(x_nchw, cpad):
    """Dense NCHW fp32 -> NHWC with `cpad` zero channels appended, one launch (kpf_nchw_to_nhwc_f32); no gradient (the images)."""
    from . import lib as L
    B, Cc, H, W = x_nchw.shape
    out = torch.empty(B, H, W, Cc + cpad, device=x_nchw.device, dtype=torch.float32)
    L.check(L.load().kpf_nchw_to_nhwc_f32(x_nchw.data_ptr(), out.data_ptr(), B, Cc, H, W, Cc + cpad, torch.cuda.current_stream().cuda_stream), "kpf_nchw_to_nhwc_f32")
    return out


class _OddPack:
    """A Linear operand whose input width is not a whole channel group, seen by the GEMM at the padded width: the packed rows [N][Kp] are zero
    beyond K anyway (Kp >= the padded width), so only the descriptor changes — the activation rows carry the matching zero channels."""

    def __init__(self, pc, cin_pad):
        self.__dict__.update(pc.__dict__)
        self.Cin = self.K = cin_pad
        self.tuned = {}
        self._base = pc

    split_allowed = False
    ps = pt = None

    def flops(self, M):
        return 2.0 * M * self.N * self.K

    def as16(self, tdt):
        return type("P16", (), {"pc": self, "Kp": self.w16.shape[1], "w": self.w16})()


class GroupedPack:
    """G kernel-layout operands of the same shape as ONE launch descriptor (kpf_conv_desc::groups): the attributes of group 0's DevPack plus
    the element distance to the next group's packed matrix (the G buffers are separate allocations; they only have to be equally spaced)."""

    def __init__(self, pcs, bias):
        p0 = pcs[0]
        self.__dict__.update({k: getattr(p0, k) for k in ("KH", "KW", "Cin", "sh", "sw", "ph", "pw", "merge", "N", "K", "Kp", "w", "w16")})
        self.ps = self.pt = None
        self.split_allowed = False
        self.tuned = {}
        self.groups = len(pcs)
        self.pcs = pcs  # (keeps the G operands alive)
        bufs = [pc.w if pc.w is not None else pc.w16 for pc in pcs]
        es = bufs[0].element_size()
        step = (bufs[1].data_ptr() - bufs[0].data_ptr()) // es
        for a, b in zip(bufs, bufs[1:]):
            assert a.shape == b.shape and a.dtype == b.dtype and (b.data_ptr() - a.data_ptr()) == step * es and step % 8 == 0, "GroupedPack: operands are not equally spaced"
        self.w_gstride = step
        self.b = bias.detach().float().contiguous() if bias is not None else None  # [G*N]

    def flops(self, M):
        return 2.0 * M * self.N * self.K

    def as16(self, tdt):
        assert self.w16 is not None and self.w16.dtype == tdt
        return type("P16", (), {"pc": self, "Kp": self.w16.shape[1], "w": self.w16})()


def _grouped_pack(cache, key, weight, bias, G, mode, prec, **kw):
    """The GroupedPack of a paired weight [G*N, Cin, KH, KW]: one (cached) operand per group slice."""
    n = weight.shape[0] // G
    if cache is not None and key is not None:
        pcs = [cache.get((key, mode, g), weight[g * n:(g + 1) * n], None, mode, prec, **kw) for g in range(G)]
    else:
        pcs = [DevPack.packed(weight[g * n:(g + 1) * n], None, mode, prec, **kw) for g in range(G)]
    if G > 2:  # one launch descriptor carries ONE group stride: more than two operands must be equally spaced — re-home them into one allocation (once)
        attr = "w" if pcs[0].w is not None else "w16"
        bufs = [getattr(pc, attr) for pc in pcs]
        es = bufs[0].element_size()
        step = bufs[1].data_ptr() - bufs[0].data_ptr()
        if any(b.data_ptr() - a.data_ptr() != step for a, b in zip(bufs, bufs[1:])) or step % (8 * es):
            assert not torch.cuda.is_current_stream_capturing(), "grouped operands must be registered in an eager iteration, before a capture"
            big = torch.empty((G,) + tuple(bufs[0].shape), device=bufs[0].device, dtype=bufs[0].dtype)
            for g, (pc, b) in enumerate(zip(pcs, bufs)):
                big[g].copy_(b)
                setattr(pc, attr, big[g])
                if cache is not None and key is not None:
                    ent = cache.entries[(key, mode, g)]
                    d = list(ent["desc"])
                    d[1] = big[g].data_ptr()  # (the refresh launch writes the operand where the kernels now read it)
                    ent["desc"] = tuple(d)
                    cache.dirty = True
    return GroupedPack(pcs, bias)


class DeferredParamGrads:
    """Weight gradients of the small Linear layers of one backward pass in ONE launch per 80 layers after it (kpf_linear_wgrad_grouped)
    instead of a GEMM launch + a reduce launch behind each of ~80 layers of 21 B rows.  While active, Conv2dNHWC.backward of an fp32 1x1 /
    Linear over at most MAX_ROWS rows whose weight IS a parameter (or a stride-preserving view of one: PackCache key without ':')
    keeps (dY, X) alive, returns dW / db tensors that are still UNWRITTEN and registers them; autograd only moves those tensors
    (AccumulateGrad adopts a parameter's first gradient without reading it), `flush()` fills them.  Anything that would read such a
    gradient earlier must not take this path: a second use of the same weight in one forward is refused here, slices / pads /
    concatenations of weights never pass a plain key, gradient hooks are not used by GraphedTrainStep (the only caller).  flush()
    checks on every eager pass that each dW became the parameter's .grad itself (a parameter outside the optimiser whose stale .grad
    made autograd add instead of adopt is how the `live_parameters` name bug of round 3 surfaced)."""
    active = None
    MAX_ROWS = 1024

    def __init__(self, named_params):
        """named_params: {PackCache key (= parameter name): Parameter} of the parameters whose gradients may be deferred — mandatory:
        a gradient is only ever deferred for a parameter this map knows, whose `.grad` is None when its backward node runs (so that
        AccumulateGrad adopts the tensor instead of adding to it), and flush() verifies the adoption of every one of them BEFORE it
        writes through the recorded addresses."""
        if named_params is None:
            raise TypeError("DeferredParamGrads needs the {name: Parameter} map of the parameters it may defer (None defers nothing safely)")
        self.named = dict(named_params)
        self.by_ptr = {p.data_ptr(): p for p in self.named.values()}
        self.items, self.colsums, self.biases, self.seen = [], [], [], set()
        self.reduces, self.reduce_checks, self.stream, self.n_reduces = [], [], None, 0

    def __enter__(self):
        assert DeferredParamGrads.active is None
        DeferredParamGrads.active = self
        self.stream = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None
        return self

    def __exit__(self, et, ev, tb):
        DeferredParamGrads.active = None
        if et is None:
            self.flush()
        else:
            self.items, self.colsums, self.biases, self.seen = [], [], [], set()
            self.reduces, self.reduce_checks = [], []
        return False

    @staticmethod
    def wants_colsum(weight, bias_ptr=None):
        """The d gamma / d beta column sums behind a LayerNorm / layer-scale backward (kpf_colsum_reduce_grouped): deferred when `weight`
        (and the bias at `bias_ptr`, for a LayerNorm: its gradient is the second row of the same deferred tensor) is a whole parameter of
        the model (same storage address and size) that holds no gradient yet."""
        g = DeferredParamGrads.active
        if g is None:
            return None

        if g._whole(weight.data_ptr(), weight.numel()) is None or (bias_ptr is not None and g._whole(bias_ptr, weight.numel()) is None):
            return None
        return g

    def _whole(self, ptr, n):
        """the parameter(s) that make up n fp32 elements from `ptr` on: one parameter, or the adjacent parameters of a paired tensor
        (training.pair_params: the halves of the gradient are adopted as views of the one deferred tensor) — all without a gradient yet"""
        got, ps = 0, []
        while got < n:
            q = self.by_ptr.get(ptr + 4 * got)
            # (a parameter that already holds a gradient would make AccumulateGrad ADD the still-unwritten tensor: not deferred)
            if q is None or q.grad is not None or q.dtype != torch.float32 or got + q.numel() > n:
                return None
            ps.append(q)
            got += q.numel()
        return ps

    # ---- the fixed-order reduces behind the split weight-gradient GEMMs of the backbones: one launch per REDUCE_BATCH calls (kpf_wgrad_reduce_multi) ----
    REDUCE_DEFER = os.environ.get("KPF_REDUCE_DEFER", "1") != "0"  # (tuning aid: 0 = every weight-gradient call launches its own reduce)
    REDUCE_BATCH = int(os.environ.get("KPF_REDUCE_BATCH", "8"))  # (at most KPF_WGRAD_REDUCE_BATCH of include/kpf.h per launch; more = more launches)

    @staticmethod
    def wants_reduce(weight, n_bias=0, bias_ptr=None):
        """The active object when the reduce behind `weight`'s gradient may wait for a batched launch: the weight (and the bias at bias_ptr, n_bias
        elements) is made of whole parameters that hold no gradient yet and receive none twice in this pass — AccumulateGrad then only adopts
        the still-unwritten tensor — and the call runs on the stream the pass was opened on (the batched launch is issued there)."""
        g = DeferredParamGrads.active
        if g is None or not g.REDUCE_DEFER or torch.cuda.current_stream().cuda_stream != g.stream:
            return None
        key = ("reduce", weight.data_ptr())
        if key in g.seen or g._whole(weight.data_ptr(), weight.numel()) is None or (bias_ptr is not None and g._whole(bias_ptr, n_bias) is None):
            return None
        g.seen.add(key)
        return g

    def add_reduce(self, desc, ws, weight, dw, bias_ptr=None, db=None):
        """desc: the kpf_wgrad_reduce_desc a *_deferred call filled; ws stays referenced until the batched launch has been issued."""
        if desc.kind < 0:  # (the call wrote the gradient itself)
            return
        self.reduces.append((desc, ws))
        self.n_reduces += 1
        self.reduce_checks.append(("a convolution weight", weight.data_ptr(), dw.data_ptr(), dw.numel()))
        if db is not None:
            self.reduce_checks.append(("a convolution bias", bias_ptr, db.data_ptr(), db.numel()))
        if len(self.reduces) >= self.REDUCE_BATCH:
            self.flush_reduces()

    def flush_reduces(self):
        """Issue the pending reduces (GraphedTrainStep also calls this before a gradient bucket is packed for its collective)."""
        if not self.reduces:
            return
        from . import lib as L
        arr = (L.WgradReduceDesc * len(self.reduces))()
        for i, (d, _) in enumerate(self.reduces):
            C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(d))
        self.reduces = []
        L.check(L.load().kpf_wgrad_reduce_multi(arr, len(arr), self.stream), "kpf_wgrad_reduce_multi")

    def _parts(self, ptr, n):
        """[(parameter, byte offset)] covering n fp32 elements of parameter storage from ptr on (see wants_colsum)."""
        out, got = [], 0
        while got < n:
            q = self.by_ptr[ptr + 4 * got]
            out.append((q, 4 * got))
            got += q.numel()
        return out

    def add_colsum(self, weight, desc, ws, out, bias_ptr=None, bias_out_ptr=None):
        key = ("colsum", weight.data_ptr())
        if key in self.seen:
            raise RuntimeError("DeferredParamGrads: a normalisation parameter receives a second gradient in one backward pass")
        self.seen.add(key)
        self.colsums.append((desc, ws, weight.data_ptr(), out.data_ptr(), weight.numel()))  # (ws stays referenced; of the output only the address)
        if bias_ptr is not None:
            for q, off in self._parts(bias_ptr, weight.numel()):  # (adoption of the bias row is verified like a Linear's bias; a paired bias: per half)
                self.biases.append(("a LayerNorm", q.data_ptr(), bias_out_ptr + off))

    @staticmethod
    def wants(key, cache, dy, x, kh, kw, stride, pad):
        g = DeferredParamGrads.active
        if g is None or cache is None or not isinstance(key, str) or ":" in key:
            return None
        p = g.named.get(key)
        if p is None or p.grad is not None:  # unknown to the map, or AccumulateGrad would add instead of adopt: the per-layer kernels
            return None
        rows = x.numel() // x.shape[-1]
        ok = (kh == 1 and kw == 1 and stride == 1 and pad == 0 and rows <= g.MAX_ROWS and dy.dtype == torch.float32 and x.dtype == torch.float32
              and x.shape[-1] % 4 == 0 and dy.shape[-1] % 4 == 0)
        return g if ok else None

    def add(self, key, dy, x, dw, db, bias_ptr=None, ldy=0):
        """ldy: floats between the rows of dy when it is a column slice of a wider matrix (0: dense)."""
        if key in self.seen:
            raise RuntimeError("DeferredParamGrads: parameter %r receives a second gradient in one backward pass" % (key,))
        self.seen.add(key)
        # (dY, X) stay referenced until flush; of dW / db only the addresses are kept — a second reference would make AccumulateGrad
        # copy the unwritten tensor instead of adopting it
        self.items.append((key, dy, x, dw.data_ptr(), None if db is None else db.data_ptr(), x.numel() // x.shape[-1], dy.shape[-1], x.shape[-1], int(ldy), dw.data_ptr()))
        if db is not None:
            self.biases.append((key, bias_ptr, db.data_ptr()))

    def add_rows(self, key, parts, dw, db, bias_ptr=None, ldy=0):
        """ONE parameter whose gradient is several row blocks with different X operands (the packed in_proj of nn.MultiheadAttention: rows 0-127 multiply
        query + qpos, rows 128-383 key + kpos): parts = [(first row, dY block, X)], each its own problem of the grouped launch, written into its rows of dw / db."""
        if key in self.seen:
            raise RuntimeError("DeferredParamGrads: parameter %r receives a second gradient in one backward pass" % (key,))
        self.seen.add(key)
        K = dw.shape[-1] if dw.dim() == 2 else dw[0].numel()
        for r0, dy, x in parts:
            self.items.append((key, dy, x, dw.data_ptr() + 4 * r0 * K, None if db is None else db.data_ptr() + 4 * r0, x.numel() // x.shape[-1], dy.shape[-1], x.shape[-1], int(ldy),
                               dw.data_ptr()))
        if db is not None:
            self.biases.append((key, bias_ptr, db.data_ptr()))

    def flush(self):
        from . import lib as L
        self.flush_reduces()
        checks, self.reduce_checks = self.reduce_checks, []
        for what, pptr, optr, n in checks:  # (the reduces have been issued by now; a gradient that autograd copied before that holds garbage: say so)
            for q, off in self._parts(pptr, n):
                if q.grad is None or q.grad.data_ptr() != optr + off:
                    raise RuntimeError("DeferredParamGrads: the gradient of %s %s was copied before its reduce had run (autograd did not adopt the tensor)" % (what, tuple(q.shape)))
        items, colsums, biases, self.items, self.colsums, self.biases, self.seen = self.items, self.colsums, self.biases, [], [], [], set()
        # the parameters whose gradients this pass deferred (GraphedTrainStep keeps them out of the buckets that are reduced DURING backward)
        self.last_deferred = (list({id(self.named[k]): self.named[k] for k, *_ in items if k in self.named}.values()) + [q for c in colsums if c[2] in self.by_ptr for q, _ in self._parts(c[2], c[4])] +
                              [self.by_ptr[b[1]] for b in biases if b[1] in self.by_ptr])
        if not items and not colsums:
            return
        # Adoption is verified BEFORE anything is written: the recorded addresses are only valid while the tensors handed to autograd
        # live on as the parameters' .grad.  If one was copied instead (hook, second reference) its storage may already belong to
        # someone else: raise without launching.
        msg = ("DeferredParamGrads: the gradient of %s was copied before it was written (autograd did not adopt the tensor: is the parameter "
               "hooked, referenced twice, or used twice in one forward?); nothing was written")
        for key, _, _, _, _, _, _, _, _, pbase in items:
            p = self.named.get(key)
            if p is None or p.grad is None or p.grad.data_ptr() != pbase:
                raise RuntimeError(msg % repr(key))
        for _, _, wptr, optr, n in colsums:
            for p, off in (self._parts(wptr, n) if wptr in self.by_ptr else [(None, 0)]):
                if p is None or p.grad is None or p.grad.data_ptr() != optr + off:
                    raise RuntimeError(msg % ("a normalisation parameter" if p is None else "a %s normalisation parameter" % (tuple(p.shape),)))
        for key, bptr, optr in biases:
            p = self.by_ptr.get(bptr)
            if p is None or p.grad is None or p.grad.data_ptr() != optr:
                raise RuntimeError(msg % ("the bias beside %r" % (key,)))
        st = torch.cuda.current_stream().cuda_stream
        if items:
            arr = (L.WgradGroupDesc * len(items))()
            for d, (key, dy, x, pw, pb, M, N, K, ldy, _) in zip(arr, items):
                d.dy, d.x, d.dw, d.db, d.M, d.N, d.K, d.ldy = dy.data_ptr(), x.data_ptr(), pw, pb, M, N, K, ldy
            L.check(L.load().kpf_linear_wgrad_grouped(arr, len(items), st), "kpf_linear_wgrad_grouped")
        if colsums:
            arr = (L.ColsumDesc * len(colsums))(*[c[0] for c in colsums])
            L.check(L.load().kpf_colsum_reduce_grouped(arr, len(colsums), st), "kpf_colsum_reduce_grouped")


GroupedLinearWgrad = DeferredParamGrads  # (the name the first form of this class had)


def _wgrad_groups(lib, dy_ptr, x_ptr, dt, dw, db, ws, nws, groups, dims, st, weight=None, bias_ptr=None):
    """kpf_conv2d_wgrad_groups(dy, x, dt, dw, db, ws, nws, groups, *dims, stream) — or, inside a DeferredParamGrads pass and for a gradient that goes to whole
    parameters (`weight`, and the bias at `bias_ptr`), kpf_conv2d_wgrad_deferred with the reduce joining a batched launch: dw / db are then still UNWRITTEN."""
    from . import lib as L
    dbp = db.data_ptr() if db is not None else None
    grp = DeferredParamGrads.wants_reduce(weight, 0 if db is None else db.numel(), bias_ptr if db is not None else None) if weight is not None else None
    if grp is not None and (db is None or bias_ptr is not None):
        desc = L.WgradReduceDesc()
        L.check(lib.kpf_conv2d_wgrad_deferred(dy_ptr, x_ptr, dt, dw.data_ptr(), dbp, ws.data_ptr(), nws, groups, *dims, C.byref(desc), st), "kpf_conv2d_wgrad_deferred")
        grp.add_reduce(desc, ws, weight, dw, bias_ptr if db is not None else None, db)
    else:
        L.check(lib.kpf_conv2d_wgrad_groups(dy_ptr, x_ptr, dt, dw.data_ptr(), dbp, ws.data_ptr(), nws, groups, *dims, st), "kpf_conv2d_wgrad_groups")


def conv_wgrad_hip(dy, x, wshape, stride, pad, want_db=True, groups=1, weight=None, bias_ptr=None):
    """(dW in OIHW, db or None) of a convolution from NHWC dY [B,OH,OW,N] and X [B,H,W,Cin]: kpf_conv2d_wgrad_f32 / _h16 (f32 MFMA GEMM
    with the pixel index as the reduction dimension, split over workgroups, fixed-order reduce).  weight (+ bias_ptr): the parameter
    tensor the gradient is for — inside a DeferredParamGrads pass its reduce may then join a batched launch (wants_reduce): dW / db are
    returned UNWRITTEN."""
    from . import lib as L
    lib = L.load()
    B, H, W, Cin = x.shape
    _, OH, OW, N = dy.shape
    KH, KW = int(wshape[2]), int(wshape[3])
    ldx, ldy = Cin, N
    Cin, N = Cin // groups, N // groups  # (groups > 1: channel-stacked operands, wshape = [G*N, Cin, KH, KW]: kpf_conv2d_wgrad_groups)
    # both operands in the same 16-bit storage type (mixed-precision step) and whole 8-element granules: kpf_conv2d_wgrad_h16 reads them
    # as they are (fp32 products and sums); anything else is widened to fp32 first
    h16 = dy.dtype == x.dtype and dy.dtype in (torch.bfloat16, torch.float16) and Cin % 8 == 0 and N % 8 == 0
    if not h16:
        dy, x = dy.float(), x.float()
    if groups == 1 and H == 1 and W == 1 and not x.is_contiguous() and x.stride(-1) == 1 and x.stride(0) % 4 == 0 and x.stride(0) > Cin and x.data_ptr() % 16 == 0 and not h16:
        ldx = x.stride(0)  # (rows that are a column slice of a wider matrix: read in place)
    else:
        x = x.contiguous()
    dy = dy.contiguous()
    nws = lib.kpf_conv2d_wgrad_ws_floats(B * OH * OW, N, KH * KW * Cin) * groups
    ws = torch.empty(nws, device=x.device, dtype=torch.float32)
    dw = torch.empty(tuple(wshape), device=x.device, dtype=torch.float32)
    db = torch.empty(N * groups, device=x.device, dtype=torch.float32) if want_db else None
    st = torch.cuda.current_stream().cuda_stream
    dt = (L.KPF_DT_BF16 if dy.dtype == torch.bfloat16 else L.KPF_DT_F16) if h16 else L.KPF_DT_F32
    if not h16 and _HEAD_MMA[0]:  # (inside a head_mma block: the layer's forward multiplied rounded operands, its weight gradient does too)
        dt = L.KPF_DT_F32_MMA_BF16 if _HEAD_MMA[0] == L.KPF_MMA_BF16 else L.KPF_DT_F32_MMA_F16
    grp = DeferredParamGrads.wants_reduce(weight, N * groups, bias_ptr if want_db else None) if weight is not None and tuple(weight.shape) == tuple(wshape) else None
    if grp is not None:
        desc = L.WgradReduceDesc()
        L.check(lib.kpf_conv2d_wgrad_deferred(dy.data_ptr(), x.data_ptr(), dt, dw.data_ptr(), db.data_ptr() if want_db else None, ws.data_ptr(), nws, groups,
                                              B, H, W, Cin, ldx, OH, OW, N, ldy, KH, KW, stride, stride, pad, pad, 0, 0, C.byref(desc), st), "kpf_conv2d_wgrad_deferred")
        grp.add_reduce(desc, ws, weight, dw, bias_ptr if want_db else None, db)
        return dw, db
    if groups > 1:
        L.check(lib.kpf_conv2d_wgrad_groups(dy.data_ptr(), x.data_ptr(), dt, dw.data_ptr(), db.data_ptr() if want_db else None, ws.data_ptr(), nws, groups,
                                            B, H, W, Cin, ldx, OH, OW, N, ldy, KH, KW, stride, stride, pad, pad, 0, 0, st), "kpf_conv2d_wgrad_groups")
        return dw, db
    if h16:
        L.check(lib.kpf_conv2d_wgrad_h16(dy.data_ptr(), x.data_ptr(), L.KPF_DT_BF16 if dy.dtype == torch.bfloat16 else L.KPF_DT_F16, dw.data_ptr(),
                                         db.data_ptr() if want_db else None, ws.data_ptr(), nws, B, H, W, Cin, Cin, OH, OW, N, N, KH, KW,
                                         stride, stride, pad, pad, st), "kpf_conv2d_wgrad_h16")
    else:
        L.check(lib.kpf_conv2d_wgrad_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if want_db else None, ws.data_ptr(), nws,
                                         B, H, W, Cin, ldx, OH, OW, N, N, KH, KW, stride, stride, pad, pad, st), "kpf_conv2d_wgrad_f32")
    return dw, db


class DwConv7NHWC(torch.autograd.Function):
    """Depthwise 7x7 (pad 3) + bias on NHWC fp32 [B,H,W,C] (convNeXT/convnext.py:41), weight in the reference's [C,1,7,7] layout.
    forward kpf_dwconv7_f32; backward: dX = the same kernel on dY with mirrored taps, dW / db = kpf_dwconv7_wgrad_f32."""

    @staticmethod
    def forward(ctx, x, weight, bias, key=None, cache=None, alias=False):
        """alias: also return x itself — the ConvNeXt block adds its input back at the end; routed through the alias, that skip path's
        gradient arrives in THIS backward and kpf_dwconv7_add_f32 folds it into dx (no separate accumulation launch)."""
        from . import lib as L
        lib = L.load()
        x = x.contiguous()
        B, H, W, Cc = x.shape
        assert x.dtype == torch.float32 and Cc % 4 == 0
        ctx.pack = (key, cache) if (cache is not None and key is not None and Cc % 32 == 0) else (None, None)
        wt = _dw_taps(weight, False, ctx.pack)  # [49][C]
        y = torch.empty_like(x)
        L.check(lib.kpf_dwconv7_f32(x.data_ptr(), wt.data_ptr(), bias.detach().contiguous().data_ptr(), y.data_ptr(), B, H, W, Cc,
                                    torch.cuda.current_stream().cuda_stream), "kpf_dwconv7_f32")
        ctx.save_for_backward(x, weight)
        ctx.bias_ptr = bias.data_ptr()
        return (y, x.view(B, H, W, Cc)) if alias else y

    @staticmethod
    def backward(ctx, dy, g_alias=None):
        from . import lib as L
        lib = L.load()
        x, weight = ctx.saved_tensors
        B, H, W, Cc = x.shape
        if dy is None:  # (only the alias was used)
            return (g_alias,) + (None,) * 5
        dy = dy.contiguous()
        st = torch.cuda.current_stream().cuda_stream
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wf = _dw_taps(weight, True, ctx.pack)  # taps mirrored, [49][C]
            dx = torch.empty_like(x)
            zb = _zero_bias(Cc, x.device)
            if g_alias is not None:
                add = g_alias.float().contiguous()
                L.check(lib.kpf_dwconv7_add_f32(dy.data_ptr(), wf.data_ptr(), zb.data_ptr(), add.data_ptr(), dx.data_ptr(), B, H, W, Cc, st), "kpf_dwconv7_add_f32")
            else:
                L.check(lib.kpf_dwconv7_f32(dy.data_ptr(), wf.data_ptr(), zb.data_ptr(), dx.data_ptr(), B, H, W, Cc, st), "kpf_dwconv7_f32")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            nws = lib.kpf_dwconv7_wgrad_ws_floats(B, H, Cc)
            ws = torch.empty(nws, device=x.device, dtype=torch.float32)
            dw = torch.empty(Cc, 1, 7, 7, device=x.device, dtype=torch.float32)
            db = torch.empty(Cc, device=x.device, dtype=torch.float32)
            grp = DeferredParamGrads.wants_reduce(weight, Cc, ctx.bias_ptr)
            if grp is not None:
                desc = L.WgradReduceDesc()
                L.check(lib.kpf_dwconv7_wgrad_deferred(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, B, H, W, Cc, C.byref(desc), st),
                        "kpf_dwconv7_wgrad_deferred")
                grp.add_reduce(desc, ws, weight, dw, ctx.bias_ptr, db)
            else:
                L.check(lib.kpf_dwconv7_wgrad_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, B, H, W, Cc, st),
                        "kpf_dwconv7_wgrad_f32")
        return dx, dw, db, None, None, None


_ZERO_BIAS = {}


def _zero_bias(n, device):
    """A persistent zero vector per (device, length): the data-gradient convolutions add no bias (allocated once, outside any capture
    that replays it; never written)."""
    key = (device.type, device.index, n)
    z = _ZERO_BIAS.get(key)
    if z is None:
        z = _ZERO_BIAS[key] = torch.zeros(n, device=device, dtype=torch.float32)
    return z


def _dw_taps(weight, mirrored, pack=(None, None)):
    """Depthwise weight [C, 1, 7, 7] -> the kernel's tap table [49][C] (mirrored: the data gradient's): a persistent operand of the
    PackCache when the layer is registered there, one launch otherwise."""
    from . import lib as L
    w = weight.detach().contiguous()
    Cc = w.shape[0]
    key, cache = pack
    if cache is not None:
        mode = 3 if mirrored else 2
        return cache.get((key, mode), w, None, mode, "f32", n_pad=Cc).w
    kp = Cc
    out = torch.empty(49, kp, device=w.device, dtype=torch.float32)
    L.check(L.load().kpf_pack_conv_weight(w.data_ptr(), 0, out.data_ptr(), 0, Cc, 1, 7, 7, 3 if mirrored else 2, Cc, kp,
                                          torch.cuda.current_stream().cuda_stream), "kpf_pack_conv_weight")
    return out


def dwconv7_nhwc(x, weight, bias, key=None, cache=None, alias=False):
    return DwConv7NHWC.apply(x, weight, bias, key, cache, alias)


_KDT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}  # KPF_DT_* of include/kpf.h


# ----------------------------------------------------------------------------------------------------------------
# paired parameters: the two backbones as ONE network with grouped convolutions
# ----------------------------------------------------------------------------------------------------------------
class _PairParams(torch.autograd.Function):
    """(a, b) -> the [2, ...] tensor whose halves ARE a's and b's storage (pair_params keeps them adjacent): no copy forward; backward hands
    each parameter its half of the gradient as a view (AccumulateGrad adopts it)."""

    @staticmethod
    def forward(ctx, a, b, both):
        return both.detach().view(both.shape)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return g[0], g[1], None


def pair_storage(registry, name, a, b):
    """The [2, *shape] tensor over the storage of tensors a and b (parameters or buffers of identical shape and type), re-homing them side by
    side the first time (and again if something moved them apart, e.g. module.to()): `a.data` / `b.data` become views of one allocation,
    values preserved.  Optimiser state, state dicts and gradients are per tensor as before; only the addresses change — so this must run
    before anything records them (it does: the first eager forward, ahead of any graph capture)."""
    if a.numel() % 4:  # (the second half would not start 16-byte aligned: not paired; pair_params concatenates)
        return None
    both = registry.get(name)
    es = a.element_size()
    if (both is None or both.data_ptr() != a.data_ptr() or both.data_ptr() + a.numel() * es != b.data_ptr() or both.device != a.device or both.dtype != a.dtype):
        assert a.shape == b.shape and a.dtype == b.dtype and a.device == b.device, "pair_storage: %s: the two tensors differ in shape / type" % name
        assert not (a.is_cuda and torch.cuda.is_current_stream_capturing()), "pair_storage: parameters must be paired before a graph capture"
        both = torch.empty((2,) + tuple(a.shape), device=a.device, dtype=a.dtype)
        with torch.no_grad():
            both[0].copy_(a.detach())
            both[1].copy_(b.detach())
        a.data, b.data = both[0], both[1]
        registry[name] = both
    return both


def pair_params(registry, name, a, b):
    """Two parameters of the paired backbones as one differentiable [2, ...] tensor (group-major: rows of a, then rows of b)."""
    both = pair_storage(registry, name, a, b)
    if both is None:  # (odd sizes: a real concatenation)
        return torch.stack((a, b), 0)
    return _PairParams.apply(a, b, both) if (a.requires_grad or b.requires_grad) else both


class BatchNormReLU(torch.autograd.Function):
    """Train-mode BatchNorm (+ ReLU) on rows [M, C] (NHWC pixels): kpf_bn_train_forward / kpf_bn_train_backward.  x may be fp32 or the
    16-bit storage type of the mixed-precision step, y is written in `out_dtype` (default: x's); statistics and arithmetic are fp32.
    Same arithmetic as F.batch_norm(training=True): biased variance for normalisation, unbiased for the running estimate."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, out_dtype=None, alias=False):
        """alias: also return x itself as a second output.  A Residual block feeds x to this BatchNorm AND to its skip path; routing the
        skip path through the alias hands its gradient to THIS backward, where kpf_bn_train_backward_add folds it into dx — otherwise
        autograd adds the two gradients of x with a separate launch over the whole activation."""
        from . import lib as L
        lib = L.load()
        x = x.contiguous()
        M, Cc = x.shape
        out_dtype = out_dtype or x.dtype
        assert x.dtype in _KDT and out_dtype in _KDT and Cc % 4 == 0
        y = torch.empty(M, Cc, device=x.device, dtype=out_dtype)
        stats = torch.empty(2, Cc, device=x.device, dtype=torch.float32)
        nws = lib.kpf_bn_ws_floats(M, Cc)
        ws = torch.empty(nws, device=x.device, dtype=torch.float32)
        L.check(lib.kpf_bn_train_forward(x.data_ptr(), _KDT[x.dtype], weight.detach().contiguous().data_ptr(), bias.detach().contiguous().data_ptr(),
                                         y.data_ptr(), _KDT[out_dtype], stats[0].data_ptr(), stats[1].data_ptr(),
                                         running_mean.data_ptr() if running_mean is not None else None,
                                         running_var.data_ptr() if running_var is not None else None, float(momentum), float(eps),
                                         int(relu), ws.data_ptr(), nws, M, Cc, torch.cuda.current_stream().cuda_stream),
                "kpf_bn_train_forward")
        ctx.save_for_backward(x, y if relu else None, stats, weight)
        ctx.relu = bool(relu)
        ctx.out_dtype = out_dtype
        return (y, x.view(M, Cc)) if alias else y

    @staticmethod
    def backward(ctx, dy, g_alias=None):
        from . import lib as L
        lib = L.load()
        x, y, stats, weight = ctx.saved_tensors
        M, Cc = x.shape
        if dy is None:  # (only the alias was used)
            return (g_alias,) + (None,) * 9
        dy = dy.to(ctx.out_dtype).contiguous()
        add = None if g_alias is None else g_alias.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dwb = torch.empty(2, Cc, device=x.device, dtype=torch.float32)
        nws = lib.kpf_bn_ws_floats(M, Cc)
        ws = torch.empty(nws, device=x.device, dtype=torch.float32)
        L.check(lib.kpf_bn_train_backward_add(dy.data_ptr(), x.data_ptr(), y.data_ptr() if ctx.relu else None, _KDT[x.dtype], _KDT[ctx.out_dtype],
                                              stats[0].data_ptr(), stats[1].data_ptr(), weight.detach().contiguous().data_ptr(),
                                              None if add is None else add.data_ptr(), dx.data_ptr(), dwb[0].data_ptr(), dwb[1].data_ptr(), int(ctx.relu),
                                              ws.data_ptr(), nws, M, Cc, torch.cuda.current_stream().cuda_stream), "kpf_bn_train_backward_add")
        return dx, dwb[0], dwb[1], None, None, None, None, None, None, None


class BnSlicesSumRelu(torch.autograd.Function):
    """The embedding sums of a fusion block with their BatchNorms inside (round 6): x [rows, n C] = n = n1 + n2 <= 4 sibling Linear outputs side by side (LinearCat) ->
    out [rows, C] = relu(S1) or relu(relu(S1) + S2) over the BatchNorm'd blocks (batch statistics per column), kpf_bn_ssr_forward / _backward: one pass over the
    pre-activations forward and two backward, where BatchNormReLU + SlicesSumRelu wrote the normalised [rows, n C] tensor and read it back (and its gradient)."""

    @staticmethod
    def forward(ctx, x, w, b, rm, rv, momentum, eps, C_, n1, n2):
        from . import lib as L
        lib = L.load()
        x = x.contiguous()
        rows, nC = x.shape
        n = n1 + n2
        assert x.dtype == torch.float32 and nC == n * C_ and C_ % 4 == 0 and C_ <= 256 and n <= 4
        out = torch.empty(rows, C_, device=x.device, dtype=torch.float32)
        stats = torch.empty(2, nC, device=x.device, dtype=torch.float32)
        nws = lib.kpf_bn_ssr_ws_floats(rows, C_, n)
        ws = torch.empty(nws, device=x.device, dtype=torch.float32)
        wc, bc = w.detach().contiguous(), b.detach().contiguous()
        L.check(lib.kpf_bn_ssr_forward(x.data_ptr(), wc.data_ptr(), bc.data_ptr(), out.data_ptr(), stats.data_ptr(), None if rm is None else rm.data_ptr(),
                                       None if rv is None else rv.data_ptr(), float(momentum), float(eps), ws.data_ptr(), nws, rows, C_, n1, n2,
                                       torch.cuda.current_stream().cuda_stream), "kpf_bn_ssr_forward")
        ctx.save_for_backward(x, out, stats, wc, bc)
        ctx.meta = (C_, n1, n2)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import lib as L
        lib = L.load()
        x, out, stats, wc, bc = ctx.saved_tensors
        C_, n1, n2 = ctx.meta
        rows, nC = x.shape
        dout = dout.float().contiguous()
        dx = torch.empty_like(x)
        dwb = torch.empty(2, nC, device=x.device, dtype=torch.float32)
        nws = lib.kpf_bn_ssr_ws_floats(rows, C_, n1 + n2)
        ws = torch.empty(nws, device=x.device, dtype=torch.float32)
        L.check(lib.kpf_bn_ssr_backward(dout.data_ptr(), out.data_ptr(), x.data_ptr(), stats.data_ptr(), wc.data_ptr(), bc.data_ptr(), dx.data_ptr(), dwb[0].data_ptr(),
                                        dwb[1].data_ptr(), ws.data_ptr(), nws, rows, C_, n1, n2, torch.cuda.current_stream().cuda_stream), "kpf_bn_ssr_backward")
        return dx, dwb[0], dwb[1], None, None, None, None, None, None, None


def bn_slices_sum_relu(x, w, b, rm, rv, momentum, eps, C_, n1, n2=0):
    return BnSlicesSumRelu.apply(x, w, b, rm, rv, momentum, eps, C_, n1, n2)


class BnReluGroupMax(torch.autograd.Function):
    """max over `group` consecutive rows of relu(BatchNorm(x)) with batch statistics on fp32 rows [M, C] (DESA's `bn_blocks -> ReLU -> max over the ball`,
    model/model.py:188-192; round 6): kpf_bn_relu_gmax_forward / _backward — the normalised tensor is never written, and the backward's BatchNorm sums run over
    the winners only (BatchNormReLU + GroupMax: eight launches and four more passes over a 43008 x 384 tensor).  The gradient may arrive as a column slice of a
    wider matrix (the concatenation behind the maximum): read in place."""

    @staticmethod
    def forward(ctx, x, w, b, rm, rv, momentum, eps, group):
        from . import lib as L
        lib = L.load()
        x = x.contiguous()
        M, Cc = x.shape
        assert x.dtype == torch.float32 and Cc % 4 == 0 and M % group == 0 and group <= 256
        y = torch.empty(M // group, Cc, device=x.device, dtype=torch.float32)
        arg = torch.empty(M // group, Cc, device=x.device, dtype=torch.uint8)
        stats = torch.empty(2, Cc, device=x.device, dtype=torch.float32)
        nws = lib.kpf_bn_relu_gmax_ws_floats(M, Cc)
        ws = torch.empty(nws, device=x.device, dtype=torch.float32)
        f = lambda t: t.detach().contiguous().data_ptr()
        L.check(lib.kpf_bn_relu_gmax_forward(x.data_ptr(), f(w), f(b), y.data_ptr(), arg.data_ptr(), stats.data_ptr(), None if rm is None else rm.data_ptr(),
                                             None if rv is None else rv.data_ptr(), float(momentum), float(eps), ws.data_ptr(), nws, M, int(group), Cc,
                                             torch.cuda.current_stream().cuda_stream), "kpf_bn_relu_gmax_forward")
        ctx.save_for_backward(x, y, arg, stats, w)
        ctx.group = int(group)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        lib = L.load()
        x, y, arg, stats, w = ctx.saved_tensors
        M, Cc = x.shape
        dy = dy.float()
        if not (dy.dim() == 2 and dy.stride(1) == 1 and dy.stride(0) % 4 == 0 and dy.stride(0) >= Cc and dy.data_ptr() % 16 == 0):
            dy = dy.contiguous()
        dx = torch.empty_like(x)
        dwb = torch.empty(2, Cc, device=x.device, dtype=torch.float32)
        nws = lib.kpf_bn_relu_gmax_ws_floats(M, Cc)
        ws = torch.empty(nws, device=x.device, dtype=torch.float32)
        L.check(lib.kpf_bn_relu_gmax_backward(dy.data_ptr(), dy.stride(0), y.data_ptr(), arg.data_ptr(), x.data_ptr(), stats.data_ptr(), w.detach().contiguous().data_ptr(),
                                              dx.data_ptr(), dwb[0].data_ptr(), dwb[1].data_ptr(), ws.data_ptr(), nws, M, ctx.group, Cc,
                                              torch.cuda.current_stream().cuda_stream), "kpf_bn_relu_gmax_backward")
        return dx, dwb[0], dwb[1], None, None, None, None, None


def bn_relu_group_max(x, w, b, rm, rv, momentum=0.1, eps=1e-5, group=64):
    return BnReluGroupMax.apply(x, w, b, rm, rv, momentum, eps, group)


class Bn2AddRelu(torch.autograd.Function):
    """relu(BatchNorm_a(xa) + BatchNorm_b(xb)) with batch statistics on fp32 rows [M, C] (DESA's local + feature branches, model/model.py:176-190; round 6):
    kpf_bn2_add_relu_forward / _backward — the normalisations, the sum and the ReLU in one pass over the two pre-activations, the backward's masked gradient,
    both branches' sums and both input gradients in four launches (two BatchNormReLU + AddRelu: seven launches each way and two normalised tensors written and
    read back).  Same arithmetic per branch as BatchNormReLU; running statistics updated the same way."""

    @staticmethod
    def forward(ctx, xa, xb, wa, ba, wb, bb, rma, rva, rmb, rvb, momentum, eps):
        from . import lib as L
        lib = L.load()
        xa, xb = xa.contiguous(), xb.contiguous()
        M, Cc = xa.shape
        assert xa.dtype == torch.float32 and xb.dtype == torch.float32 and xb.shape == xa.shape and Cc % 4 == 0
        out = torch.empty_like(xa)
        stats = torch.empty(4, Cc, device=xa.device, dtype=torch.float32)
        nws = lib.kpf_bn2_ws_floats(M, Cc)
        ws = torch.empty(nws, device=xa.device, dtype=torch.float32)
        f = lambda t: t.detach().contiguous().data_ptr()
        o = lambda t: None if t is None else t.data_ptr()
        L.check(lib.kpf_bn2_add_relu_forward(xa.data_ptr(), xb.data_ptr(), f(wa), f(ba), f(wb), f(bb), out.data_ptr(), stats.data_ptr(), o(rma), o(rva), o(rmb), o(rvb),
                                             float(momentum), float(eps), ws.data_ptr(), nws, M, Cc, torch.cuda.current_stream().cuda_stream), "kpf_bn2_add_relu_forward")
        ctx.save_for_backward(xa, xb, out, stats, wa, wb)
        return out

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        lib = L.load()
        xa, xb, out, stats, wa, wb = ctx.saved_tensors
        M, Cc = xa.shape
        dy = dy.float().contiguous()
        dxa, dxb = torch.empty_like(xa), torch.empty_like(xb)
        dwb_a, dwb_b = torch.empty(2, Cc, device=xa.device, dtype=torch.float32), torch.empty(2, Cc, device=xa.device, dtype=torch.float32)
        nws = lib.kpf_bn2_ws_floats(M, Cc)
        ws = torch.empty(nws, device=xa.device, dtype=torch.float32)
        f = lambda t: t.detach().contiguous().data_ptr()
        L.check(lib.kpf_bn2_add_relu_backward(dy.data_ptr(), out.data_ptr(), xa.data_ptr(), xb.data_ptr(), stats.data_ptr(), f(wa), f(wb), dxa.data_ptr(), dxb.data_ptr(),
                                              dwb_a[0].data_ptr(), dwb_a[1].data_ptr(), dwb_b[0].data_ptr(), dwb_b[1].data_ptr(), ws.data_ptr(), nws, M, Cc,
                                              torch.cuda.current_stream().cuda_stream), "kpf_bn2_add_relu_backward")
        return dxa, dxb, dwb_a[0], dwb_a[1], dwb_b[0], dwb_b[1], None, None, None, None, None, None


def bn2_add_relu(xa, xb, wa, ba, wb, bb, rma, rva, rmb, rvb, momentum=0.1, eps=1e-5):
    return Bn2AddRelu.apply(xa, xb, wa, ba, wb, bb, rma, rva, rmb, rvb, momentum, eps)


def batchnorm_relu_rows(x, weight, bias, running_mean, running_var, momentum=0.1, eps=1e-5, relu=True, out_dtype=None, alias=False):
    return BatchNormReLU.apply(x, weight, bias, running_mean, running_var, momentum, eps, relu, out_dtype, alias)


class LayerNormRows(torch.autograd.Function):
    """F.layer_norm over the last axis (convNeXT/convnext.py:43, 199-214; the post-LN layers of the fusion head): kpf_ln_train_forward /
    kpf_ln_train_backward.  x any shape [..., C] fp32; the output may be written directly in the 16-bit operand type of the GEMM that
    follows (`out_dtype`); statistics and gradients are fp32, parameter gradients are added in a fixed order."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype=None, groups=1):
        """groups = G > 1: x [..., G*C] holds G channel groups, each normalised on its own with its own parameter set (weight, bias [G*C]
        group-major: the paired backbones' LayerNorms, pair_params) — kpf_ln_train_forward_g on the [rows*G, C] view."""
        from . import lib as L
        x = x.float().contiguous()
        Cc = x.shape[-1] // groups
        rows = x.numel() // Cc
        out_dtype = out_dtype or torch.float32
        y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
        stats = torch.empty(2, rows, device=x.device, dtype=torch.float32)
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        L.check(L.load().kpf_ln_train_forward_g(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), _KDT[out_dtype], stats[0].data_ptr(), stats[1].data_ptr(),
                                                rows, Cc, groups, float(eps), torch.cuda.current_stream().cuda_stream), "kpf_ln_train_forward_g")
        ctx.save_for_backward(x, stats, w)
        ctx.groups = groups
        ctx.bias_ptr = b.data_ptr()  # (identifies the bias PARAMETER for DeferredParamGrads: its gradient is deferred together with the weight's)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        lib = L.load()
        x, stats, w = ctx.saved_tensors
        G = ctx.groups
        Cc = x.shape[-1] // G
        rows = x.numel() // Cc
        dy = dy.contiguous()
        if dy.dtype not in _KDT:
            dy = dy.float()
        dx = torch.empty_like(x)
        dwb = torch.empty(2, G * Cc, device=x.device, dtype=torch.float32)
        nws = lib.kpf_ln_ws_floats(rows, G * Cc)
        ws = torch.empty(nws, device=x.device, dtype=torch.float32)
        st = torch.cuda.current_stream().cuda_stream
        if G > 1:
            grp = DeferredParamGrads.wants_colsum(w, ctx.bias_ptr)  # (paired parameters: both halves are checked and adopted)
            desc = L.ColsumDesc() if grp is not None else None
            L.check(lib.kpf_ln_train_backward_g(dy.data_ptr(), _KDT[dy.dtype], x.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), w.data_ptr(), dx.data_ptr(),
                                                dwb[0].data_ptr(), dwb[1].data_ptr(), ws.data_ptr(), nws, rows, Cc, G, C.byref(desc) if desc is not None else None, st),
                    "kpf_ln_train_backward_g")
            if grp is not None:
                grp.add_colsum(w, desc, ws, dwb, ctx.bias_ptr, dwb[1].data_ptr())
            return dx, dwb[0], dwb[1], None, None, None
        grp = DeferredParamGrads.wants_colsum(w, ctx.bias_ptr)
        if grp is not None:  # d gamma / d beta: reduced with every other layer's after backward
            desc = L.ColsumDesc()
            L.check(lib.kpf_ln_train_backward_partial(dy.data_ptr(), _KDT[dy.dtype], x.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), w.data_ptr(),
                                                      dx.data_ptr(), dwb[0].data_ptr(), dwb[1].data_ptr(), ws.data_ptr(), nws, rows, Cc, C.byref(desc), st),
                    "kpf_ln_train_backward_partial")
            grp.add_colsum(w, desc, ws, dwb, ctx.bias_ptr, dwb[1].data_ptr())
        else:
            L.check(lib.kpf_ln_train_backward(dy.data_ptr(), _KDT[dy.dtype], x.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), w.data_ptr(), dx.data_ptr(),
                                              dwb[0].data_ptr(), dwb[1].data_ptr(), ws.data_ptr(), nws, rows, Cc, st), "kpf_ln_train_backward")
        return dx, dwb[0], dwb[1], None, None, None


class GeluRows(torch.autograd.Function):
    """GELU(erf) (convNeXT/convnext.py:33; BERT's intermediate activation): kpf_gelu_forward / kpf_gelu_backward on fp32 or 16-bit tensors."""

    @staticmethod
    def forward(ctx, x):
        from . import lib as L
        x = x.contiguous()
        assert x.dtype in _KDT and x.numel() % 4 == 0
        y = torch.empty_like(x)
        L.check(L.load().kpf_gelu_forward(x.data_ptr(), y.data_ptr(), _KDT[x.dtype], x.numel(), torch.cuda.current_stream().cuda_stream), "kpf_gelu_forward")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        (x,) = ctx.saved_tensors
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        L.check(L.load().kpf_gelu_backward(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), _KDT[x.dtype], x.numel(), torch.cuda.current_stream().cuda_stream),
                "kpf_gelu_backward")
        return dx


class Attn21(torch.autograd.Function):
    """ctx = dropout(softmax(scale * Q K^T)) V per (sample, head) for the 21-token stacks: kpf_attn21_forward / _backward.  q, k, v
    [B, 21, H*32] fp32 as the projections produced them; returns ctx in the same layout."""

    @staticmethod
    def forward(ctx_, q, k, v, heads, scale, p_drop, rng, call_id):
        from . import lib as L
        q, k, v = q.float().contiguous(), k.float().contiguous(), v.float().contiguous()
        B, T, Cc = q.shape
        hd = Cc // heads
        out = torch.empty_like(q)
        P = torch.empty(B, heads, T, T, device=q.device, dtype=torch.float32)
        M = torch.empty(B, heads, T, T, device=q.device, dtype=torch.uint8)
        L.check(L.load().kpf_attn21_forward(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), P.data_ptr(), M.data_ptr(), B, T, heads, hd, Cc,
                                            float(scale), float(p_drop), rng.data_ptr() if rng is not None else None, int(call_id),
                                            torch.cuda.current_stream().cuda_stream), "kpf_attn21_forward")
        ctx_.save_for_backward(q, k, v, P, M)
        ctx_.conf = (heads, float(scale), float(p_drop))
        return out

    @staticmethod
    def backward(ctx_, dctx):
        from . import lib as L
        q, k, v, P, M = ctx_.saved_tensors
        heads, scale, p_drop = ctx_.conf
        B, T, Cc = q.shape
        dctx = dctx.float().contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        L.check(L.load().kpf_attn21_backward(dctx.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), P.data_ptr(), M.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                             dv.data_ptr(), B, T, heads, Cc // heads, Cc, scale, p_drop, torch.cuda.current_stream().cuda_stream),
                "kpf_attn21_backward")
        return dq, dk, dv, None, None, None, None, None


def attn21(q, k, v, heads, scale, p_drop=0.0, rng=None, call_id=0):
    return Attn21.apply(q, k, v, heads, scale, p_drop, rng, call_id)


class SelfAttention21(torch.autograd.Function):
    """The self-attention of a BERT layer of the 21-token stacks (model/model.py:30-70) from its input: q | k | v as ONE projection GEMM
    (N = 3C; the three parameters packed into one operand by PackCache.get_stacked), the attention core reading the three column slices
    where that GEMM left them (kpf_attn21_forward_ld), and in the backward ONE data-gradient GEMM over [dq | dk | dv].  The three weight /
    bias gradients join the grouped launch after backward (DeferredParamGrads, dy = a column slice: kpf_wgrad_group_desc::ldy) or take
    kpf_conv2d_wgrad_f32 with ldy = 3C.  7 launches per layer and direction fewer than three Linears + Attn21, and the input's gradient
    arrives as one tensor instead of three to add."""

    @staticmethod
    def forward(ctx, h, wq, bq, wk, bk, wv, bv, names, cache, heads, scale, p_drop, rng, call_id):
        from . import lib as L
        B, T, Cc = h.shape
        M = B * T
        hc = h.float().contiguous()
        sp = cache.get_stacked(names, (wq, wk, wv), (bq, bk, bv))
        qkv = _conv_any(sp, hc.view(M, 1, 1, Cc), "f32").view(M, 3 * Cc)
        out = torch.empty(B, T, Cc, device=h.device, dtype=torch.float32)
        P = torch.empty(B, heads, T, T, device=h.device, dtype=torch.float32)
        Mk = torch.empty(B, heads, T, T, device=h.device, dtype=torch.uint8)
        q = qkv.data_ptr()
        L.check(L.load().kpf_attn21_forward_ld(q, q + 4 * Cc, q + 8 * Cc, out.data_ptr(), P.data_ptr(), Mk.data_ptr(), B, T, heads, Cc // heads, 3 * Cc, Cc,
                                               float(scale), float(p_drop), rng.data_ptr() if rng is not None else None, int(call_id),
                                               torch.cuda.current_stream().cuda_stream), "kpf_attn21_forward_ld")
        ctx.save_for_backward(hc, qkv, P, Mk, wq, wk, wv)
        ctx.conf = (sp, names, cache, heads, float(scale), float(p_drop), tuple(b.data_ptr() for b in (bq, bk, bv)))
        ctx.set_materialize_grads(False)
        # second output: h itself, for the residual path of the layer (LayerNorm(h + ...)) — routed through this alias, the residual's gradient arrives in
        # THIS backward and rides in the data-gradient GEMM's residual epilogue instead of a separate accumulation launch (the BatchNormReLU / DwConv7 trick)
        return out, hc.view(B, T, Cc)

    @staticmethod
    def backward(ctx, dctx, g_alias=None):
        from . import lib as L
        lib = L.load()
        hc, qkv, P, Mk, wq, wk, wv = ctx.saved_tensors
        if dctx is None:  # (only the alias was used)
            return (g_alias,) + (None,) * 13
        sp, names, cache, heads, scale, p_drop, bias_ptrs = ctx.conf
        B, T, Cc = hc.shape
        M = B * T
        st = torch.cuda.current_stream().cuda_stream
        dctx = dctx.float().contiguous()
        dqkv = torch.empty(M, 3 * Cc, device=hc.device, dtype=torch.float32)
        q, dq = qkv.data_ptr(), dqkv.data_ptr()
        L.check(lib.kpf_attn21_backward_ld(dctx.data_ptr(), q, q + 4 * Cc, q + 8 * Cc, P.data_ptr(), Mk.data_ptr(), dq, dq + 4 * Cc, dq + 8 * Cc, B, T, heads, Cc // heads,
                                           3 * Cc, Cc, scale, p_drop, st), "kpf_attn21_backward_ld")
        dh = None
        if ctx.needs_input_grad[0]:
            ga = None if g_alias is None else g_alias.float().contiguous().view(M, 1, 1, Cc)
            dh = _conv_any(sp.dgrad, dqkv.view(M, 1, 1, 3 * Cc), "f32", res=ga).view(B, T, Cc)  # (+ the residual path's gradient, in the epilogue)
        grads = []
        for i, (w, name) in enumerate(zip((wq, wk, wv), names)):
            dyi = dqkv[:, i * Cc:(i + 1) * Cc]
            grp = DeferredParamGrads.wants(name, cache, dyi, hc, 1, 1, 1, 0)
            if grp is not None:
                bp = grp.by_ptr.get(bias_ptrs[i])
                if bp is None or bp.numel() != Cc or bp.grad is not None:
                    grp = None
            dw = torch.empty(tuple(w.shape), device=hc.device, dtype=torch.float32)
            db = torch.empty(Cc, device=hc.device, dtype=torch.float32)
            if grp is not None:
                grp.add(name, dyi, hc, dw, db, bias_ptrs[i], ldy=3 * Cc)
            else:
                nws = lib.kpf_conv2d_wgrad_ws_floats(M, Cc, Cc)
                ws = torch.empty(nws, device=hc.device, dtype=torch.float32)
                L.check(lib.kpf_conv2d_wgrad_f32(dyi.data_ptr(), hc.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, M, 1, 1, Cc, Cc, 1, 1, Cc, 3 * Cc,
                                                 1, 1, 1, 1, 0, 0, st), "kpf_conv2d_wgrad_f32")
            grads += [dw, db]
        return (dh,) + tuple(grads) + (None,) * 7


def self_attention21(h, wq, bq, wk, bk, wv, bv, names, cache, heads, scale, p_drop=0.0, rng=None, call_id=0):
    """-> (context, h_alias): feed h_alias (== h) to the residual path of the layer, see SelfAttention21.forward."""
    return SelfAttention21.apply(h, wq, bq, wk, bk, wv, bv, names, cache, heads, scale, p_drop, rng, call_id)


def _table_prefix_grad(table, dE, B, now_c):
    """The gradient of an embedding table [L, C] of which the stacks used the first T rows, from the per-sample gradients dE [B, T, C] of those rows: their sum
    over the batch, zero behind — ONE descriptor of the grouped column-sum launch (kpf_colsum_desc::reserved) instead of a sum launch and a zero-extending copy.
    Deferred to the launch after backward when the table is a whole parameter without a gradient yet; appended to now_c otherwise."""
    from . import lib as L
    n = dE.numel() // B
    assert n % 2 == 0 and table.numel() >= n and table.dtype == torch.float32
    dt = torch.empty(tuple(table.shape), device=dE.device, dtype=torch.float32)
    desc = L.ColsumDesc()
    desc.part, desc.dw, desc.db, desc.nblk, desc.C, desc.first_block, desc.reserved = dE.data_ptr(), dt.data_ptr(), dt.data_ptr() + 2 * n, B, n // 2, 0, table.numel() - n
    grp = DeferredParamGrads.wants_colsum(table)
    if grp is not None:
        grp.add_colsum(table, desc, dE, dt)
    else:
        now_c.append(desc)
    return dt


class BertStack21(torch.autograd.Function):
    """The four BERT layers of a KP_Interaction_TR stack (model/model.py:30-126: transformers' BertEncoder under .train(), 21 tokens x 128) as ONE launch each way
    (csrc/kpf_trstack.hip: kpf_tr_stack_train_forward / _backward; round 6) — h = layers(dropout(e + pos)) with e the embedding Linear's output.  The unfused
    form (`TrainGraph.bert_layer`, KPF_TR_FUSED=0) is 7 launches per layer forward and ~10 backward on 21 B rows.  The backward writes every Linear's dY beside
    the X the forward kept; the 24 weight / bias gradients join the deferred grouped launch after backward (DeferredParamGrads) exactly like the unfused
    layers' — or one grouped launch right here when no deferral is active — and the 8 LayerNorm parameter gradients leave as per-sample partial sums for the
    grouped column-sum reduce.  Dropout masks: the device-resident (seed, counter) hash of Attn21 / DropAddLN with call ids call0 .. call0 + 12, recomputed in
    the backward (nothing stored).  params: 16 tensors per layer in the order of `BertStack21.ORDER`; names: their parameter names (the deferral's keys)."""
    ORDER = ("attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight", "attention.self.key.bias",
             "attention.self.value.weight", "attention.self.value.bias", "attention.output.dense.weight", "attention.output.dense.bias",
             "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
             "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias")
    # (weight index, X of its weight gradient, dY) per Linear of a layer, as kpf_tr_stack_offset's `which`; (N, K); dY's row stride and column offset
    LINEARS = ((0, 0, 4, 128, 128, 384, 0), (2, 0, 4, 128, 128, 384, 128), (4, 0, 4, 128, 128, 384, 256), (6, 1, 5, 128, 128, 0, 0), (10, 2, 6, 16, 128, 0, 0),
               (12, 3, 7, 128, 16, 0, 0))
    _tables = {}

    @staticmethod
    def param_table(params):
        """DEVICE array of the parameters' addresses (the kernels read the weights where they lie: no packed copies).  Cached per address tuple: the warm-up
        iterations of a GraphedTrainStep build it, the capture finds it."""
        key = tuple(p.data_ptr() for p in params)
        t = BertStack21._tables.get(key)
        if t is None:
            for p in params:
                if p.dtype != torch.float32 or not p.is_contiguous() or p.data_ptr() % 16:
                    raise ValueError("BertStack21: parameters must be contiguous 16-byte-aligned fp32 tensors")
            # (never evicted: a table is 0.5 KB, holds addresses only — valid for as long as those addresses are parameters, whoever owns them — and an eviction
            #  between a GraphedTrainStep's warm-up and its capture would make the capture build one, a host -> device copy a capture cannot contain)
            t = BertStack21._tables[key] = torch.tensor(key, dtype=torch.int64, device=params[0].device)
        return t

    @staticmethod
    def forward(ctx, e, pos, names, cache, p_drop, rng, call0, mma, *params):
        from . import lib as L
        lib = L.load()
        B, T, Cc = e.shape
        # pos: the position table [L >= 21, 128] (its first 21 rows are used; the backward returns the whole table's gradient) or exactly the 21 rows
        assert T == 21 and Cc == 128 and len(params) == 64 and len(names) == 64 and pos.dim() == 2 and pos.shape[0] >= 21 and pos.shape[1] == 128
        ec, pc = e.float().contiguous(), pos.float().contiguous()
        ctx.pos_rows = pos.shape[0]
        table = BertStack21.param_table(params)
        n = lib.kpf_tr_stack_save_floats(B)
        save = torch.empty(n, device=e.device, dtype=torch.float32)
        L.check(lib.kpf_tr_stack_train_forward(ec.data_ptr(), pc.data_ptr(), table.data_ptr(), save.data_ptr(), n, B, float(p_drop),
                                               rng.data_ptr() if (rng is not None and p_drop > 0) else None, int(call0), int(mma), torch.cuda.current_stream().cuda_stream),
                "kpf_tr_stack_train_forward")
        ctx.save_for_backward(save, table, pos, *params)
        ctx.conf = (names, cache, float(p_drop), int(call0), B, int(mma))
        off = lib.kpf_tr_stack_out_offset(B)
        return save[off:off + B * T * Cc].view(B, T, Cc)

    @staticmethod
    def backward(ctx, dh):
        from . import lib as L
        lib = L.load()
        save, table, pos, *params = ctx.saved_tensors
        names, cache, p_drop, call0, B, mma = ctx.conf
        M, dev = B * 21, save.device
        st = torch.cuda.current_stream().cuda_stream
        dh = dh.float().contiguous()
        dE = torch.empty(B, 21, 128, device=dev, dtype=torch.float32)
        dys = torch.empty(lib.kpf_tr_stack_dy_floats(B), device=dev, dtype=torch.float32)
        parts = torch.empty(lib.kpf_tr_stack_part_floats(B), device=dev, dtype=torch.float32)
        L.check(lib.kpf_tr_stack_train_backward(dh.data_ptr(), table.data_ptr(), save.data_ptr(), dE.data_ptr(), dys.data_ptr(), parts.data_ptr(), B, p_drop, call0, mma, st),
                "kpf_tr_stack_train_backward")
        grads = [None] * 64
        now_w, now_c = [], []  # gradients that are not deferred: one grouped launch each, right here
        for l in range(4):
            for wi, xw, yw, N, K, ldy, coff in BertStack21.LINEARS:
                i = 16 * l + wi
                w, bias, name = params[i], params[i + 1], names[i]
                xo, yo = lib.kpf_tr_stack_offset(B, l, xw), lib.kpf_tr_stack_offset(B, l, yw)
                x = save[xo:xo + M * K].view(M, K)
                dyfull = dys[yo:yo + M * (ldy or N)].view(M, ldy or N)
                dy = dyfull[:, coff:coff + N] if ldy else dyfull
                dw = torch.empty(tuple(w.shape), device=dev, dtype=torch.float32)
                db = torch.empty(N, device=dev, dtype=torch.float32)
                grp = DeferredParamGrads.wants(name, cache, dy, x, 1, 1, 1, 0)
                if grp is not None:
                    bp = grp.by_ptr.get(bias.data_ptr())
                    if bp is None or bp.numel() != N or bp.grad is not None:
                        grp = None
                if grp is not None:
                    grp.add(name, dy, x, dw, db, bias.data_ptr(), ldy=ldy)
                else:
                    now_w.append((dy, x, dw, db, M, N, K, ldy))
                grads[i], grads[i + 1] = dw, db
            for ln, wi in ((0, 8), (1, 14)):
                i = 16 * l + wi
                w, bias = params[i], params[i + 1]
                dwb = torch.empty(2, 128, device=dev, dtype=torch.float32)
                po = ((l * 2 + ln) * B) * 256
                desc = L.ColsumDesc()
                desc.part, desc.dw, desc.db, desc.nblk, desc.C, desc.first_block, desc.reserved = parts.data_ptr() + 4 * po, dwb[0].data_ptr(), dwb[1].data_ptr(), B, 128, 0, 0
                grp = DeferredParamGrads.wants_colsum(w, bias.data_ptr())
                if grp is not None:
                    grp.add_colsum(w, desc, parts, dwb, bias.data_ptr(), dwb[1].data_ptr())
                else:
                    now_c.append(desc)
                grads[i], grads[i + 1] = dwb[0], dwb[1]
        if now_w:
            arr = (L.WgradGroupDesc * len(now_w))()
            for d, (dy, x, dw, db, M_, N, K, ldy) in zip(arr, now_w):
                d.dy, d.x, d.dw, d.db, d.M, d.N, d.K, d.ldy = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), M_, N, K, ldy
            L.check(lib.kpf_linear_wgrad_grouped(arr, len(now_w), st), "kpf_linear_wgrad_grouped")
        dpos = _table_prefix_grad(pos, dE, B, now_c) if ctx.needs_input_grad[1] else None
        if now_c:
            arr = (L.ColsumDesc * len(now_c))(*now_c)
            L.check(lib.kpf_colsum_reduce_grouped(arr, len(now_c), st), "kpf_colsum_reduce_grouped")
        return (dE, dpos, None, None, None, None, None, None) + tuple(grads)


MMA_MODE = {"f32": 0, "bf16": 1, "f16": 2}


def bert_stack21(e, pos, names, cache, p_drop, rng, call0, params, prec="f32"):
    """prec: the GEMM arithmetic of the stack — "f32" exact fp32 products, "bf16" / "f16" operands rounded in registers, fp32 accumulation (the mixed-precision step)."""
    return BertStack21.apply(e, pos, tuple(names), cache, p_drop, rng, call0, MMA_MODE[prec], *params)


class XAttnLayer21(torch.autograd.Function):
    """The decoder layer of a fusion block (updatedDecoder layer 3, model/transfusion_head.py:137-173) in train mode as ONE launch each way
    (kpf_xattn_train_forward / _backward, csrc/kpf_trstack.hip; round 6): out = LayerNorm(x + dropout(W2 dropout(relu(W1 x)))), x = LayerNorm(query +
    dropout(attention(query + qpos, key + kpos) Wo^T)).  params (ORDER): in_proj_weight / bias, out_proj.weight / bias, norm2.weight / bias, linear1.weight / bias,
    linear2.weight / bias, norm3.weight / bias; qpos / kpos [21, 128] are the first rows of the position tables (differentiable).  The packed in_proj gradient is
    two problems of one grouped weight-gradient launch here (rows 0-127 from (dq, query + qpos), rows 128-383 from (d(k | v), key + kpos)); out_proj / linear1 /
    linear2 join the deferred grouped launch like any small Linear; the LayerNorm sums leave as per-sample partials."""
    ORDER = ("multihead_attn.in_proj_weight", "multihead_attn.in_proj_bias", "multihead_attn.out_proj.weight", "multihead_attn.out_proj.bias", "norm2.weight", "norm2.bias",
             "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm3.weight", "norm3.bias")

    @staticmethod
    def forward(ctx, query, key, qpos, kpos, names, cache, p_drop, rng, call0, mma, *params):
        from . import lib as L
        lib = L.load()
        B, T, Cc = query.shape
        # qpos / kpos: the position tables [L >= 21, 128] (first 21 rows used; the backward returns whole-table gradients) or exactly the 21 rows
        assert T == 21 and Cc == 128 and len(params) == 12 and all(t.dim() == 2 and t.shape[0] >= 21 and t.shape[1] == 128 for t in (qpos, kpos))
        assert tuple(params[0].shape) == (384, 128) and tuple(params[6].shape) == (128, 128) and tuple(params[8].shape) == (128, 128), "decoder layer: d_model 128, feed-forward 128"
        qc, kc = query.float().contiguous(), key.float().contiguous()
        qp, kp = qpos.float().contiguous(), kpos.float().contiguous()
        table = BertStack21.param_table(tuple(params) + (qp, kp))
        n = lib.kpf_xattn_train_save_floats(B)
        save = torch.empty(n, device=query.device, dtype=torch.float32)
        L.check(lib.kpf_xattn_train_forward(qc.data_ptr(), kc.data_ptr(), table.data_ptr(), save.data_ptr(), n, B, float(p_drop),
                                            rng.data_ptr() if (rng is not None and p_drop > 0) else None, int(call0), int(mma), torch.cuda.current_stream().cuda_stream),
                "kpf_xattn_train_forward")
        ctx.save_for_backward(save, table, qp, kp, *params)
        ctx.conf = (names, cache, float(p_drop), int(call0), B, int(mma))
        off = lib.kpf_xattn_train_offset(B, 5)
        return save[off:off + B * T * Cc].view(B, T, Cc)

    @staticmethod
    def backward(ctx, dout):
        from . import lib as L
        lib = L.load()
        save, table, qp, kp, *params = ctx.saved_tensors
        names, cache, p_drop, call0, B, mma = ctx.conf
        M, dev = B * 21, save.device
        st = torch.cuda.current_stream().cuda_stream
        dout = dout.float().contiguous()
        new = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
        dq, dqe, dke = new(B, 21, 128), new(B, 21, 128), new(B, 21, 128)
        dys, parts = new(lib.kpf_xattn_train_dy_floats(B)), new(2 * B * 256)
        L.check(lib.kpf_xattn_train_backward(dout.data_ptr(), table.data_ptr(), save.data_ptr(), dq.data_ptr(), dqe.data_ptr(), dke.data_ptr(), dys.data_ptr(),
                                             parts.data_ptr(), B, p_drop, call0, mma, st), "kpf_xattn_train_backward")
        X = lambda which, K: (lambda o: save[o:o + M * K].view(M, K))(lib.kpf_xattn_train_offset(B, which))
        dqkv = dys[:M * 384].view(M, 384)
        DY = lambda which: (lambda o: dys[o:o + M * 128].view(M, 128))(lib.kpf_xattn_train_offset(B, which))
        grads = [None] * 12
        now_w, now_c = [], []
        # in_proj: one parameter, two problems (different X): two row blocks of one deferred gradient (add_rows), or two problems of the launch below
        dwin, dbin = new(384, 128), new(384)
        grp = DeferredParamGrads.wants(names[0], cache, dqkv[:, :128], X(0, 128), 1, 1, 1, 0)
        if grp is not None:
            bp = grp.by_ptr.get(params[1].data_ptr())
            if bp is None or bp.numel() != 384 or bp.grad is not None:
                grp = None
        if grp is not None:
            grp.add_rows(names[0], [(0, dqkv[:, :128], X(0, 128)), (128, dqkv[:, 128:], X(1, 128))], dwin, dbin, params[1].data_ptr(), ldy=384)
        else:
            now_w.append((dqkv[:, :128], X(0, 128), dwin.data_ptr(), dbin.data_ptr(), M, 128, 128, 384))
            now_w.append((dqkv[:, 128:], X(1, 128), dwin.data_ptr() + 4 * 128 * 128, dbin.data_ptr() + 4 * 128, M, 256, 128, 384))
        grads[0], grads[1] = dwin, dbin
        for wi, xw, yw in ((2, 2, 7), (6, 3, 8), (8, 4, 9)):
            w, bias, name = params[wi], params[wi + 1], names[wi]
            x, dy = X(xw, 128), DY(yw)
            dw, db = new(128, 128), new(128)
            grp = DeferredParamGrads.wants(name, cache, dy, x, 1, 1, 1, 0)
            if grp is not None:
                bp = grp.by_ptr.get(bias.data_ptr())
                if bp is None or bp.numel() != 128 or bp.grad is not None:
                    grp = None
            if grp is not None:
                grp.add(name, dy, x, dw, db, bias.data_ptr())
            else:
                now_w.append((dy, x, dw.data_ptr(), db.data_ptr(), M, 128, 128, 0))
            grads[wi], grads[wi + 1] = dw, db
        for ln, wi in ((0, 4), (1, 10)):
            w, bias = params[wi], params[wi + 1]
            dwb = new(2, 128)
            desc = L.ColsumDesc()
            desc.part, desc.dw, desc.db, desc.nblk, desc.C, desc.first_block, desc.reserved = parts.data_ptr() + 4 * ln * B * 256, dwb[0].data_ptr(), dwb[1].data_ptr(), B, 128, 0, 0
            grp = DeferredParamGrads.wants_colsum(w, bias.data_ptr())
            if grp is not None:
                grp.add_colsum(w, desc, parts, dwb, bias.data_ptr(), dwb[1].data_ptr())
            else:
                now_c.append(desc)
            grads[wi], grads[wi + 1] = dwb[0], dwb[1]
        if now_w:
            arr = (L.WgradGroupDesc * len(now_w))()
            for d, (dy, x, pw, pb, M_, N, K, ldy) in zip(arr, now_w):
                d.dy, d.x, d.dw, d.db, d.M, d.N, d.K, d.ldy = dy.data_ptr(), x.data_ptr(), pw, pb, M_, N, K, ldy
            L.check(lib.kpf_linear_wgrad_grouped(arr, len(now_w), st), "kpf_linear_wgrad_grouped")
        dqpos = _table_prefix_grad(qp, dqe, B, now_c) if ctx.needs_input_grad[2] else None
        dkpos = _table_prefix_grad(kp, dke, B, now_c) if ctx.needs_input_grad[3] else None
        if now_c:
            arr = (L.ColsumDesc * len(now_c))(*now_c)
            L.check(lib.kpf_colsum_reduce_grouped(arr, len(now_c), st), "kpf_colsum_reduce_grouped")
        return (dq, dke, dqpos, dkpos, None, None, None, None, None, None) + tuple(grads)


def xattn_layer21(query, key, qpos, kpos, names, cache, p_drop, rng, call0, params, prec="f32"):
    return XAttnLayer21.apply(query, key, qpos, kpos, tuple(names), cache, p_drop, rng, call0, MMA_MODE[prec], *params)


class DropAddLN(torch.autograd.Function):
    """y = LayerNorm(h + dropout(o)) over the last axis, fp32, one launch each way (kpf_drop_add_ln_forward / _backward): the `dense -> dropout ->
    + residual -> LayerNorm` tail of both halves of a BERT layer (model/model.py:72-126) — the library path is a dropout, an add and the
    LayerNorm forward, and a dropout backward plus the LayerNorm backward on the way back."""

    @staticmethod
    def forward(ctx, o, h, weight, bias, eps, p_drop, rng, call_id):
        from . import lib as L
        o, h = o.float().contiguous(), h.float().contiguous()
        Cc = h.shape[-1]
        rows = h.numel() // Cc
        y, xs = torch.empty_like(h), torch.empty_like(h)
        stats = torch.empty(2, rows, device=h.device, dtype=torch.float32)
        mask = torch.empty(h.shape, device=h.device, dtype=torch.uint8) if p_drop > 0 else None
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        L.check(L.load().kpf_drop_add_ln_forward(o.data_ptr(), h.data_ptr(), w.data_ptr(), b.data_ptr(), xs.data_ptr(), y.data_ptr(),
                                                 mask.data_ptr() if mask is not None else None, stats[0].data_ptr(), stats[1].data_ptr(), rows, Cc, float(eps),
                                                 float(p_drop), rng.data_ptr() if rng is not None else None, int(call_id), torch.cuda.current_stream().cuda_stream),
                "kpf_drop_add_ln_forward")
        ctx.save_for_backward(xs, stats, w, mask)
        ctx.p_drop = float(p_drop)
        ctx.bias_ptr = b.data_ptr()
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        lib = L.load()
        xs, stats, w, mask = ctx.saved_tensors
        Cc = xs.shape[-1]
        rows = xs.numel() // Cc
        dy = dy.float().contiguous()
        dh, do = torch.empty_like(xs), torch.empty_like(xs)
        dwb = torch.empty(2, Cc, device=xs.device, dtype=torch.float32)
        nws = lib.kpf_ln_ws_floats(rows, Cc)
        ws = torch.empty(nws, device=xs.device, dtype=torch.float32)
        grp = DeferredParamGrads.wants_colsum(w, ctx.bias_ptr)
        desc = L.ColsumDesc() if grp is not None else None
        L.check(lib.kpf_drop_add_ln_backward(dy.data_ptr(), xs.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), w.data_ptr(), mask.data_ptr() if mask is not None else None,
                                             dh.data_ptr(), do.data_ptr(), dwb[0].data_ptr(), dwb[1].data_ptr(), ws.data_ptr(), nws, rows, Cc, ctx.p_drop,
                                             C.byref(desc) if desc is not None else None, torch.cuda.current_stream().cuda_stream), "kpf_drop_add_ln_backward")
        if grp is not None:
            grp.add_colsum(w, desc, ws, dwb, ctx.bias_ptr, dwb[1].data_ptr())
        return do, dh, dwb[0], dwb[1], None, None, None, None


def drop_add_ln(o, h, weight, bias, eps, p_drop=0.0, rng=None, call_id=0):
    return DropAddLN.apply(o, h, weight, bias, eps, p_drop, rng, call_id)


class GateMix(torch.autograd.Function):
    """(sw, gw) of a fusion block's gate from the atten_spatial logits [B*P, J], the geometry map gam [B, J, P], the scalar weight_dis and the pooling
    weight w_fc [1, P] (kpf_gate_mix_forward / _backward; model/model.py:334-341): sw = sigmoid(logits) as [B, J, P], gw = (sig(wd) gam + (1 - sig(wd)) sw) w_fc."""

    @staticmethod
    def forward(ctx, logits, gam, weight_dis, w_fc):
        from . import lib as L
        B, J_, P = gam.shape
        lg, ga = logits.float().contiguous(), gam.float().contiguous()
        wd, wf = weight_dis.detach().float().contiguous(), w_fc.detach().float().contiguous()
        assert lg.numel() == B * P * J_ and wf.numel() == P and wd.numel() == 1
        sw, gw = torch.empty_like(ga), torch.empty_like(ga)
        L.check(L.load().kpf_gate_mix_forward(lg.data_ptr(), ga.data_ptr(), wd.data_ptr(), wf.data_ptr(), sw.data_ptr(), gw.data_ptr(), B, J_, P,
                                              torch.cuda.current_stream().cuda_stream), "kpf_gate_mix_forward")
        ctx.save_for_backward(sw, ga, wd, wf)
        ctx.shapes = (tuple(logits.shape), tuple(weight_dis.shape), tuple(w_fc.shape))
        ctx.set_materialize_grads(False)
        return sw, gw

    @staticmethod
    def backward(ctx, d_sw, d_gw):
        from . import lib as L
        sw, ga, wd, wf = ctx.saved_tensors
        B, J_, P = ga.shape
        if d_gw is None:
            d_gw = torch.zeros_like(ga)
        d_gw = d_gw.float().contiguous()
        d_sw = d_sw.float().contiguous() if d_sw is not None else None
        dgam, dlog = torch.empty_like(ga), torch.empty(B * P, J_, device=ga.device, dtype=torch.float32)
        dwf, dwd = torch.empty(P, device=ga.device, dtype=torch.float32), torch.empty(1, device=ga.device, dtype=torch.float32)
        ws = torch.empty(B * J_, device=ga.device, dtype=torch.float32)
        L.check(L.load().kpf_gate_mix_backward(sw.data_ptr(), ga.data_ptr(), wd.data_ptr(), wf.data_ptr(), d_sw.data_ptr() if d_sw is not None else None, d_gw.data_ptr(),
                                               dgam.data_ptr(), dlog.data_ptr(), dwf.data_ptr(), dwd.data_ptr(), ws.data_ptr(), B, J_, P,
                                               torch.cuda.current_stream().cuda_stream), "kpf_gate_mix_backward")
        ls, ds_, fs = ctx.shapes
        return dlog.view(ls), dgam, dwd.view(ds_), dwf.view(fs)


class SplitRows(torch.autograd.Function):
    """w [n*C, ...] -> n row blocks of C (views); the backward concatenates the n gradients in ONE launch — autograd's own slicing backward is a zero
    fill plus a copy per block and n - 1 adds (the packed in_proj weight / bias of nn.MultiheadAttention, model/transfusion_head.py:437-470)."""

    @staticmethod
    def forward(ctx, w, n):
        c = w.shape[0] // n
        ctx.meta = ((c,) + tuple(w.shape[1:]), w.dtype, w.device)
        return tuple(w[i * c:(i + 1) * c] for i in range(n))

    @staticmethod
    def backward(ctx, *gs):
        shp, dt, dev = ctx.meta
        gs = [g.contiguous() if g is not None else torch.zeros(shp, dtype=dt, device=dev) for g in gs]
        return torch.cat(gs, 0), None


class PrefixRows(torch.autograd.Function):
    """w[:T] of an embedding table [L, C]; backward = the gradient zero-extended to L rows in one launch (kpf_pad_rows) instead of a fill and a copy."""

    @staticmethod
    def forward(ctx, w, T):
        ctx.shape = tuple(w.shape)
        return w[:T]

    @staticmethod
    def backward(ctx, g):
        L_, Cc = ctx.shape[0], 1
        for d in ctx.shape[1:]:
            Cc *= d
        if g.shape[0] == L_:
            return g, None
        return pad_rows(g.contiguous().view(1, -1), L_ * Cc).view(ctx.shape), None


class AddRelu(torch.autograd.Function):
    """relu(scale * (a + b [+ c])) on fp32 tensors of one shape, one launch each way (kpf_add_relu_forward / _backward); every addend receives the
    same gradient tensor."""

    @staticmethod
    def forward(ctx, scale, a, b=None, c=None):
        from . import lib as L
        ts = [t.float().contiguous() for t in (a, b, c) if t is not None]
        assert all(t.shape == ts[0].shape for t in ts) and ts[0].numel() % 4 == 0
        out = torch.empty_like(ts[0])
        L.check(L.load().kpf_add_relu_forward(ts[0].data_ptr(), ts[1].data_ptr() if len(ts) > 1 else None, ts[2].data_ptr() if len(ts) > 2 else None, out.data_ptr(),
                                              out.numel(), float(scale), torch.cuda.current_stream().cuda_stream), "kpf_add_relu_forward")
        ctx.save_for_backward(out)
        ctx.conf = (float(scale), len(ts))
        return out

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        (out,) = ctx.saved_tensors
        scale, n = ctx.conf
        dy = dy.float().contiguous()
        dx = torch.empty_like(out)
        L.check(L.load().kpf_add_relu_backward(dy.data_ptr(), out.data_ptr(), dx.data_ptr(), out.numel(), scale, torch.cuda.current_stream().cuda_stream), "kpf_add_relu_backward")
        return (None,) + (dx,) * n + (None,) * (3 - n)


def add_relu(a, b=None, c=None, scale=1.0):
    return AddRelu.apply(scale, a, b, c)


class BmmSmallK(torch.autograd.Function):
    """out[b] = A[b] (J x P) @ X[b] (P x C) with few rows J (the 21 joints; model/model.py:318-320, 336-341): all three products on the HIP kernels —
    forward kpf_bmm_small_k_fwd, dA = dOut X^T kpf_bmm_small_k_da, dX = A^T dOut kpf_bmm_small_k_dx (a K = J batched GEMM the library takes 443 us for;
    the other two sat on torch.bmm until round 5).  Fixed summation orders: replays are bit-identical."""

    @staticmethod
    def forward(ctx, A, X):
        from . import lib as L
        A, X = A.float().contiguous(), X.float().contiguous()
        B, J, P = A.shape
        Cc = X.shape[-1]
        assert X.shape[:2] == (B, P) and J <= 24 and Cc % 4 == 0, "bmm_small_k: A [B, J <= 24, P] @ X [B, P, C % 4 == 0]"
        ctx.save_for_backward(A, X)
        out = torch.empty(B, J, Cc, device=A.device, dtype=torch.float32)
        L.check(L.load().kpf_bmm_small_k_fwd(A.data_ptr(), X.data_ptr(), out.data_ptr(), B, J, P, Cc, torch.cuda.current_stream().cuda_stream), "kpf_bmm_small_k_fwd")
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import lib as L
        lib = L.load()
        A, X = ctx.saved_tensors
        B, J, P = A.shape
        Cc = X.shape[-1]
        dout = dout.float().contiguous()
        st = torch.cuda.current_stream().cuda_stream
        dA = dX = None
        if ctx.needs_input_grad[0]:
            dA = torch.empty_like(A)
            L.check(lib.kpf_bmm_small_k_da(dout.data_ptr(), X.data_ptr(), dA.data_ptr(), B, J, P, Cc, st), "kpf_bmm_small_k_da")
        if ctx.needs_input_grad[1]:
            dX = torch.empty_like(X)
            L.check(lib.kpf_bmm_small_k_dx(A.data_ptr(), dout.data_ptr(), dX.data_ptr(), B, J, P, Cc, st), "kpf_bmm_small_k_dx")
        return dA, dX


def bmm_small_k(A, X):
    """A [B, J, P] @ X [B, P, C].  The HIP kernels (fixed summation order, bit-identical replays) cover the shapes the model has — J <= 24 rows, C % 4 == 0,
    one sample's dOut tile plus a 32-row A tile J * (C + 32) * 4 <= 64 KB of LDS (C <= 748 at J = 21); anything wider takes torch.bmm, chosen HERE, before autograd records a node, so a
    wider head fails nowhere inside backward."""
    J, Cc = A.shape[1], X.shape[2]
    if A.is_cuda and J <= 24 and Cc % 4 == 0 and J * (Cc + 32) * 4 <= 65536:
        return BmmSmallK.apply(A, X)
    return torch.bmm(A, X.to(A.dtype))


def layer_norm_rows(x, weight, bias, eps, out_dtype=None, groups=1):
    return LayerNormRows.apply(x, weight, bias, eps, out_dtype, groups)


def gelu_rows(x):
    return GeluRows.apply(x)


class Upsample2xNHWC(torch.autograd.Function):
    """Bilinear x2 (align_corners False: nn.Upsample of model/resnetUnet.py:259) on NHWC [B,H,W,C], fp32 or 16-bit storage.
    forward kpf_upsample2x_f32 / _h16, backward kpf_upsample2x_bwd (gather form: run-to-run deterministic, unlike the library's
    atomic scatter)."""

    @staticmethod
    def forward(ctx, x):
        from . import lib as L
        lib = L.load()
        x = x.contiguous()
        B, H, W, Cc = x.shape
        assert x.dtype in _KDT and Cc % 4 == 0
        y = torch.empty(B, 2 * H, 2 * W, Cc, device=x.device, dtype=x.dtype)
        st = torch.cuda.current_stream().cuda_stream
        if x.dtype == torch.float32:
            L.check(lib.kpf_upsample2x_f32(x.data_ptr(), y.data_ptr(), B, H, W, Cc, Cc, 0, st), "kpf_upsample2x_f32")
        else:
            L.check(lib.kpf_upsample2x_h16(x.data_ptr(), y.data_ptr(), B, H, W, Cc, Cc, 0, _KDT[x.dtype], st), "kpf_upsample2x_h16")
        ctx.shape = (B, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        B, H, W, Cc = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, Cc, device=dy.device, dtype=dy.dtype)
        L.check(L.load().kpf_upsample2x_bwd(dy.data_ptr(), dx.data_ptr(), _KDT[dy.dtype], B, H, W, Cc, torch.cuda.current_stream().cuda_stream),
                "kpf_upsample2x_bwd")
        return dx


class MaxPool3x3s2NHWC(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) (model/resnet.py:168) on NHWC; the winning tap of every window is kept (one byte per element) and the
    backward is a gather over the <= 4 windows covering an input element: deterministic, first maximum wins like ATen."""

    @staticmethod
    def forward(ctx, x):
        from . import lib as L
        x = x.contiguous()
        B, H, W, Cc = x.shape
        assert x.dtype in _KDT and Cc % 4 == 0
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(B, OH, OW, Cc, device=x.device, dtype=x.dtype)
        tap = torch.empty(B, OH, OW, Cc, device=x.device, dtype=torch.uint8)
        L.check(L.load().kpf_maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), tap.data_ptr(), _KDT[x.dtype], B, H, W, Cc,
                                              torch.cuda.current_stream().cuda_stream), "kpf_maxpool3x3s2_fwd")
        ctx.save_for_backward(tap)
        ctx.shape = (B, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        (tap,) = ctx.saved_tensors
        B, H, W, Cc = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, Cc, device=dy.device, dtype=dy.dtype)
        L.check(L.load().kpf_maxpool3x3s2_bwd(dy.data_ptr(), tap.data_ptr(), dx.data_ptr(), _KDT[dy.dtype], B, H, W, Cc,
                                              torch.cuda.current_stream().cuda_stream), "kpf_maxpool3x3s2_bwd")
        return dx


class BallGroup(torch.autograd.Function):
    """DESA's grouping at its three radii (model/model.py:166-175: ball query around every joint among points + joints, group_points, centre
    subtraction, offsets / radius) from ONE launch of the inference path's kpf_ball_group_f32.  Returns, per radius, the grouped feature
    differences [B*J*64, 128] and the scaled offsets [B*J*64, 4] (3 + a zero channel: the width their GEMM reads) — column slices of the kernel's
    [.., 132]-wide rows, read in place by the Linears — and the index sets.  Gradient towards the point / joint features only (the reference
    detaches the coordinates): the gather's backward through kpf_row_gather_bwd_f32 (fixed order) minus the group sums for the centres."""

    @staticmethod
    def forward(ctx, pcl_xyz, node_xyz, pcl_feat, node_feat):
        from . import lib as L
        B, N, _ = pcl_xyz.shape
        Jn, Cc = node_feat.shape[1], node_feat.shape[2]
        assert Cc == 128
        dev = pcl_xyz.device
        X, JF = pcl_feat.detach().float().contiguous(), node_feat.detach().float().contiguous()
        G = torch.empty(3, B * Jn * 64, Cc + 4, device=dev)
        idx = torch.empty(3, B * Jn, 64, device=dev, dtype=torch.int32)
        L.check(L.load().kpf_ball_group_f32(pcl_xyz.detach().float().contiguous().data_ptr(), node_xyz.detach().float().contiguous().data_ptr(), X.data_ptr(),
                                            JF.data_ptr(), Cc, G.data_ptr(), idx.data_ptr(), B, N, 0.1, 0.2, 0.4, torch.cuda.current_stream().cuda_stream),
                "kpf_ball_group_f32")
        ctx.save_for_backward(idx)
        ctx.shape = (B, N, Jn, Cc)
        ctx.set_materialize_grads(False)  # (no zero tensors for the outputs nothing differentiates: offsets, indices)
        outs = []
        for i in range(3):
            outs += [G[i][:, :Cc], G[i][:, Cc:]]
        ctx.mark_non_differentiable(outs[1], outs[3], outs[5], idx)
        return tuple(outs) + (idx,)

    @staticmethod
    def backward(ctx, d0, _a, d1, _b, d2, _c, _d):
        from . import lib as L
        lib = L.load()
        (idx,) = ctx.saved_tensors
        B, N, Jn, Cc = ctx.shape
        P, R = N + Jn, Jn * 64
        st = torch.cuda.current_stream().cuda_stream
        nws = lib.kpf_row_gather_ws_ints(3 * B, P, R, 1)
        ws = torch.empty(nws, device=idx.device, dtype=torch.int32)  # the three radii's index sets as 3B images: ONE inversion
        L.check(lib.kpf_row_gather_invert(idx.data_ptr(), ws.data_ptr(), nws, 3 * B, P, R, 1, st), "kpf_row_gather_invert")
        dsrc = centre = None
        for i, d in enumerate((d0, d1, d2)):
            if d is None:
                continue
            d = d.float().contiguous()
            ds = torch.empty(B, P, Cc, device=d.device, dtype=torch.float32)
            L.check(lib.kpf_row_gather_accum_f32(d.data_ptr(), ws.data_ptr() + 4 * i * B * (P + 1), ws.data_ptr() + 4 * (3 * B * (P + 1) + i * B * R), None, ds.data_ptr(),
                                                 B, P, R, 1, Cc, st), "kpf_row_gather_accum_f32")
            dsrc = ds if dsrc is None else dsrc + ds
            centre = d if centre is None else centre + d
        if dsrc is None:
            return None, None, None, None
        dnode = dsrc[:, N:] - centre.view(B, Jn, 64, Cc).sum(2)
        return None, None, dsrc[:, :N], dnode


def ball_group(pcl_xyz, node_xyz, pcl_feat, node_feat):
    return BallGroup.apply(pcl_xyz, node_xyz, pcl_feat, node_feat)


class _GroupParams(torch.autograd.Function):
    """(t_0 .. t_{n-1}) -> the [n, ...] tensor whose slices ARE their storage (group_storage keeps them adjacent): no copy forward; backward hands each
    parameter its slice of the gradient as a view (AccumulateGrad adopts it).  The n-tensor form of _PairParams (round 6: DESA's three radii)."""

    @staticmethod
    def forward(ctx, both, *ts):
        ctx.n = len(ts)
        return both.detach().view(both.shape)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return (None,) + tuple(g[i] for i in range(ctx.n))


def group_storage(registry, name, ts):
    """pair_storage for n tensors of identical shape and type: the [n, *shape] tensor over their (re-homed, adjacent) storage."""
    a = ts[0]
    if a.numel() % 4:
        return None
    both = registry.get(name)
    es = a.element_size()
    ok = both is not None and both.device == a.device and both.dtype == a.dtype and all(t.data_ptr() == both.data_ptr() + i * a.numel() * es for i, t in enumerate(ts))
    if not ok:
        assert all(t.shape == a.shape and t.dtype == a.dtype and t.device == a.device for t in ts), "group_storage: %s: the tensors differ in shape / type" % name
        assert not (a.is_cuda and torch.cuda.is_current_stream_capturing()), "group_storage: parameters must be grouped before a graph capture"
        both = torch.empty((len(ts),) + tuple(a.shape), device=a.device, dtype=a.dtype)
        with torch.no_grad():
            for i, t in enumerate(ts):
                both[i].copy_(t.detach())
        for i, t in enumerate(ts):
            t.data = both[i]
        registry[name] = both
    return both


def group_params(registry, name, ts):
    """n parameters (or buffers) as one [n, ...] tensor, group-major, differentiable towards each of them."""
    both = group_storage(registry, name, ts)
    if both is None:
        return torch.stack(tuple(ts), 0)
    return _GroupParams.apply(both, *ts) if any(t.requires_grad for t in ts) else both


class BallGroup3(torch.autograd.Function):
    """BallGroup with the three radii CHANNEL-STACKED (round 6): returns GF [B*J*64, 3*128] (grouped feature differences, radius i in columns [128 i, 128 i +
    128)), GX [B*J*64, 3*4] (offsets / radius, 3 + a zero channel per radius) and the index sets — what the grouped launches of TrainGraph.desa read — from
    ONE launch (kpf_ball_group_stacked_f32); backward: one index inversion + ONE launch for all three radii, both outputs (kpf_ball_group_bwd_f32: gathered
    rows in radius / entry order, joint rows minus their groups' sums) where BallGroup needed three accumulation launches and ~14 library ops around them."""

    @staticmethod
    def forward(ctx, pcl_xyz, node_xyz, pcl_feat, node_feat):
        from . import lib as L
        B, N, _ = pcl_xyz.shape
        Jn, Cc = node_feat.shape[1], node_feat.shape[2]
        assert Cc == 128
        dev = pcl_xyz.device
        X, JF = pcl_feat.detach().float().contiguous(), node_feat.detach().float().contiguous()
        R = B * Jn * 64
        GF = torch.empty(R, 3 * Cc, device=dev, dtype=torch.float32)
        GX = torch.empty(R, 12, device=dev, dtype=torch.float32)
        idx = torch.empty(3, B * Jn, 64, device=dev, dtype=torch.int32)
        L.check(L.load().kpf_ball_group_stacked_f32(pcl_xyz.detach().float().contiguous().data_ptr(), node_xyz.detach().float().contiguous().data_ptr(), X.data_ptr(),
                                                    JF.data_ptr(), Cc, GF.data_ptr(), GX.data_ptr(), idx.data_ptr(), B, N, 0.1, 0.2, 0.4,
                                                    torch.cuda.current_stream().cuda_stream), "kpf_ball_group_stacked_f32")
        ctx.save_for_backward(idx)
        ctx.shape = (B, N, Jn, Cc)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(GX, idx)
        return GF, GX, idx

    @staticmethod
    def backward(ctx, d3, _a, _b):
        from . import lib as L
        if d3 is None:
            return None, None, None, None
        lib = L.load()
        (idx,) = ctx.saved_tensors
        B, N, Jn, Cc = ctx.shape
        P, R = N + Jn, Jn * 64
        st = torch.cuda.current_stream().cuda_stream
        d3 = d3.float().contiguous()
        nws = lib.kpf_row_gather_ws_ints(3 * B, P, R, 1)
        ws = torch.empty(nws, device=idx.device, dtype=torch.int32)
        L.check(lib.kpf_row_gather_invert(idx.data_ptr(), ws.data_ptr(), nws, 3 * B, P, R, 1, st), "kpf_row_gather_invert")
        dX = torch.empty(B, N, Cc, device=d3.device, dtype=torch.float32)
        dnode = torch.empty(B, Jn, Cc, device=d3.device, dtype=torch.float32)
        L.check(lib.kpf_ball_group_bwd_f32(d3.data_ptr(), 3 * Cc, ws.data_ptr(), ws.data_ptr() + 4 * 3 * B * (P + 1), dX.data_ptr(), dnode.data_ptr(), B, N, Jn, st),
                "kpf_ball_group_bwd_f32")
        return None, None, dX, dnode


def ball_group3(pcl_xyz, node_xyz, pcl_feat, node_feat):
    return BallGroup3.apply(pcl_xyz, node_xyz, pcl_feat, node_feat)


class UnstackRows(torch.autograd.Function):
    """The channel-stacked maps of G paired networks ([B, H, W, ld] in the step's storage type, network g's C channels at column g * gs) as G dense fp32 maps in
    one launch (kpf_unstack_rows): NHWC [B, H, W, C] each, or with nchw=True dense NCHW [B, C, H, W] — the layout the reference returns its offset maps in and the
    decode / the loss read, so that neither needs a transposing copy.  Backward: the G gradients (same layout) back into one stacked tensor in one launch
    (kpf_restack_rows, pad columns zero).  Round 6: replaces a strided cast per map forward and cast + zero fill + strided copy + fan-in add per map backward at
    the seam between the paired backbones and the fusion head."""

    @staticmethod
    def forward(ctx, y, G, C_, gs, nchw=False):
        from . import lib as L
        yc = y.contiguous()
        assert yc.dim() == 4
        B, H, W, ld = yc.shape
        rows, hw = B * H * W, (H * W if nchw else 0)
        out = torch.empty((G, B, C_, H, W) if nchw else (G, B, H, W, C_), device=y.device, dtype=torch.float32)
        L.check(L.load().kpf_unstack_rows(yc.data_ptr(), _KDT[yc.dtype], out.data_ptr(), rows, G, C_, ld, gs, hw, torch.cuda.current_stream().cuda_stream), "kpf_unstack_rows")
        ctx.meta = (G, C_, gs, ld, rows, hw, yc.dtype, tuple(yc.shape))
        return tuple(out[g] for g in range(G))

    @staticmethod
    def backward(ctx, *grads):
        from . import lib as L
        G, C_, gs, ld, rows, hw, dt, shape = ctx.meta
        gs_ = [None if g is None else g.float().contiguous() for g in grads]
        ptrs = (C.c_void_p * G)(*[None if g is None else g.data_ptr() for g in gs_])
        dy = torch.empty(shape, device=next(g for g in gs_ if g is not None).device, dtype=dt)
        L.check(L.load().kpf_restack_rows(ptrs, dy.data_ptr(), _KDT[dt], rows, G, C_, ld, gs, hw, torch.cuda.current_stream().cuda_stream), "kpf_restack_rows")
        return dy, None, None, None, None


def unstack_rows(y, G, C_, gs, nchw=False):
    return UnstackRows.apply(y, G, C_, gs, nchw)


class GroupMax(torch.autograd.Function):
    """y = x.view(rows, group, C).max(1)[0] on fp32 rows (model/model.py:192: the maximum over a ball's 64 members) with the winner's member index kept: ONE launch
    each way (kpf_group_max_train_forward / _backward; the library's max + its backward are a reduction, a zero fill and a scatter)."""

    @staticmethod
    def forward(ctx, x, group):
        from . import lib as L
        xc = x.float().contiguous()
        Cc = xc.shape[-1]
        rows = xc.numel() // (Cc * group)
        y = torch.empty(rows, Cc, device=x.device, dtype=torch.float32)
        arg = torch.empty(rows, Cc, device=x.device, dtype=torch.uint8)
        L.check(L.load().kpf_group_max_train_forward(xc.data_ptr(), y.data_ptr(), arg.data_ptr(), rows, int(group), Cc, torch.cuda.current_stream().cuda_stream),
                "kpf_group_max_train_forward")
        ctx.save_for_backward(arg)
        ctx.meta = (tuple(x.shape), int(group), x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        (arg,) = ctx.saved_tensors
        shape, group, dt = ctx.meta
        rows, Cc = arg.shape
        dy = dy.float().reshape(rows, Cc)
        if not (dy.stride(1) == 1 and dy.stride(0) % 4 == 0 and dy.stride(0) >= Cc and dy.data_ptr() % 16 == 0):
            dy = dy.contiguous()  # (otherwise read in place: the gradient usually arrives as a column slice of the concatenation's)
        dx = torch.empty(rows * group, Cc, device=dy.device, dtype=torch.float32)
        L.check(L.load().kpf_group_max_train_backward(dy.data_ptr(), dy.stride(0) if rows > 1 else Cc, arg.data_ptr(), dx.data_ptr(), rows, group, Cc,
                                                      torch.cuda.current_stream().cuda_stream), "kpf_group_max_train_backward")
        return dx.view(shape).to(dt), None


def group_max(x, group):
    return GroupMax.apply(x, group)


class LinearSlices(torch.autograd.Function):
    """n independent Linear layers with an ODD input width (DESA's Conv2d(3 -> 128) on the offsets of each radius, model/model.py:176-179) over the column blocks
    of one channel-stacked input: y[:, N i : N i + N] = x[:, Kp i : Kp i + K] W_i^T + b_i, written by n launches into ONE [rows, n N] tensor (so that the
    BatchNorm behind it is one 3-launch pass over n N channels instead of n passes).  x [rows, n * Kp] with Kp = K rounded up to 4 (the pad columns zero) and no
    gradient (the offsets are not differentiated); backward = the n weight gradients from dY's column blocks in place (kpf_conv2d_wgrad_groups: ldx, ldy, trim)."""

    @staticmethod
    def forward(ctx, x, keys, cache, *wb):
        from .engine import Act, conv
        n = len(wb) // 2
        ws, bs = wb[0::2], wb[1::2]
        N, K = ws[0].shape[0], ws[0][0].numel()
        Kp = (K + 3) // 4 * 4
        xc = x.detach().float().contiguous()
        rows = xc.shape[0]
        assert xc.shape[1] == n * Kp and N % 4 == 0
        y = torch.empty(rows, n * N, device=x.device, dtype=torch.float32)
        for i in range(n):
            w2 = ws[i].reshape(N, K, 1, 1)
            pc = cache.get((keys[i], 0), w2, bs[i], 0, "f32", stride=1, pad=0, patchify=False) if cache is not None else DevPack.packed(w2, bs[i], 0, "f32", stride=1, pad=0, patchify=False)
            conv(_OddPack(pc, Kp), Act(xc.view(-1), 1, 1, rows, Kp, n * Kp, Kp * i), out=Act(y.view(-1), 1, 1, rows, N, n * N, N * i))
        ctx.save_for_backward(xc, *ws)
        ctx.meta = (n, N, K, Kp, [b is not None for b in bs])
        ctx.bias_ptrs = [None if b is None else b.data_ptr() for b in bs]
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        lib = L.load()
        xc, *ws = ctx.saved_tensors
        n, N, K, Kp, has_b = ctx.meta
        rows = xc.shape[0]
        dy = dy.float().contiguous()
        st = torch.cuda.current_stream().cuda_stream
        grads = []
        nws = lib.kpf_conv2d_wgrad_ws_floats(rows, N, Kp)
        for i in range(n):
            wsb = torch.empty(nws, device=xc.device, dtype=torch.float32)
            dw = torch.empty(tuple(ws[i].shape), device=xc.device, dtype=torch.float32)
            db = torch.empty(N, device=xc.device, dtype=torch.float32) if has_b[i] else None
            _wgrad_groups(lib, dy.data_ptr() + 4 * N * i, xc.data_ptr() + 4 * Kp * i, L.KPF_DT_F32, dw, db, wsb, nws, 1,
                          (1, 1, rows, Kp, n * Kp, 1, rows, N, n * N, 1, 1, 1, 1, 0, 0, K, N), st, weight=ws[i], bias_ptr=ctx.bias_ptrs[i])
            grads += [dw, db]
        return (None, None, None) + tuple(grads)


def linear_slices(x, keys, cache, wb):
    return LinearSlices.apply(x, tuple(keys), cache, *wb)


class LinearCat(torch.autograd.Function):
    """n independent Linear layers of the same output width N over n DIFFERENT row tensors, written side by side into ONE [rows, n N] tensor (round 6: the
    point / joint embeddings of a fusion block, model/model.py:254-259, 417-422 — so that the n BatchNorms behind them are one pass over n N channels and the sum +
    ReLU one launch, training.slices_sum_relu).  x_i [rows, K_i] fp32 (K_i a multiple of 4: odd widths arrive padded with zero columns, the weight keeps its own
    width); backward: per layer the weight / bias gradient from dY's column block in place (ldy = n N) and, where x_i requires a gradient, dX_i = dY_i W_i from the
    data-gradient GEMM reading the same block in place."""

    @staticmethod
    def forward(ctx, keys, cache, n, *args):
        from .engine import Act, conv
        xs, wb = args[:n], args[n:]
        ws, bs = wb[0::2], wb[1::2]
        N = ws[0].shape[0]
        rows = xs[0].shape[0]
        y = torch.empty(rows, n * N, device=xs[0].device, dtype=torch.float32)
        xcs, kps = [], []
        for i in range(n):
            xc = xs[i].detach().float().contiguous()
            K = ws[i][0].numel()
            Kp = xc.shape[1]
            assert xc.shape[0] == rows and Kp % 4 == 0 and Kp >= K and Kp - K < 4 and ws[i].shape[0] == N and N % 4 == 0
            w2 = ws[i].reshape(N, K, 1, 1)
            pc = cache.get((keys[i], 0), w2, bs[i], 0, "f32", stride=1, pad=0, patchify=False) if cache is not None else DevPack.packed(w2, bs[i], 0, "f32", stride=1, pad=0, patchify=False)
            conv(_OddPack(pc, Kp) if Kp != K else pc, Act(xc.view(-1), 1, 1, rows, Kp), out=Act(y.view(-1), 1, 1, rows, N, n * N, N * i))
            xcs.append(xc)
            kps.append((K, Kp))
        ctx.save_for_backward(*xcs, *ws)
        ctx.meta = (n, N, kps, [b is not None for b in bs], keys, cache)
        ctx.bias_ptrs = [None if b is None else b.data_ptr() for b in bs]
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        from .engine import Act, conv
        lib = L.load()
        n, N, kps, has_b, keys, cache = ctx.meta
        xcs, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        rows = xcs[0].shape[0]
        dy = dy.float().contiguous()
        st = torch.cuda.current_stream().cuda_stream
        dxs, grads = [], []
        for i in range(n):
            K, Kp = kps[i]
            dx = None
            if ctx.needs_input_grad[3 + i]:
                assert K == Kp, "LinearCat: data gradients for whole channel groups only"
                w2 = ws[i].detach().reshape(N, K, 1, 1)
                pc = cache.get((keys[i], 1), w2, None, 1, "f32", pad=0, n_pad=N) if cache is not None else DevPack.packed(w2, None, 1, "f32", pad=0, n_pad=N)
                dx = torch.empty(rows, K, device=dy.device, dtype=torch.float32)
                conv(pc, Act(dy.view(-1), 1, 1, rows, N, n * N, N * i), out=Act(dx.view(-1), 1, 1, rows, K))
            dxs.append(dx)
            dw = db = None
            if ctx.needs_input_grad[3 + n + 2 * i]:
                nws = lib.kpf_conv2d_wgrad_ws_floats(rows, N, Kp)
                wsb = torch.empty(nws, device=dy.device, dtype=torch.float32)
                dw = torch.empty(tuple(ws[i].shape), device=dy.device, dtype=torch.float32)
                db = torch.empty(N, device=dy.device, dtype=torch.float32) if has_b[i] else None
                _wgrad_groups(lib, dy.data_ptr() + 4 * N * i, xcs[i].data_ptr(), L.KPF_DT_F32, dw, db, wsb, nws, 1,
                              (1, 1, rows, Kp, Kp, 1, rows, N, n * N, 1, 1, 1, 1, 0, 0, K, N), st, weight=ws[i], bias_ptr=ctx.bias_ptrs[i])
            grads += [dw, db]
        return (None, None, None) + tuple(dxs) + tuple(grads)


def linear_cat(xs, keys, cache, wb):
    return LinearCat.apply(tuple(keys), cache, len(xs), *xs, *wb)


class SlicesSumRelu(torch.autograd.Function):
    """out = relu(S1) or relu(relu(S1) + S2) over the column blocks of y [rows, (n1 + n2) C] (kpf_slices_sum_relu_forward / _backward: one launch each way)."""

    @staticmethod
    def forward(ctx, y, C_, n1, n2):
        from . import lib as L
        yc = y.float().contiguous()
        rows = yc.shape[0]
        assert yc.shape[1] == (n1 + n2) * C_
        out = torch.empty(rows, C_, device=y.device, dtype=torch.float32)
        L.check(L.load().kpf_slices_sum_relu_forward(yc.data_ptr(), out.data_ptr(), rows, C_, n1, n2, torch.cuda.current_stream().cuda_stream), "kpf_slices_sum_relu_forward")
        ctx.save_for_backward(yc, out)
        ctx.meta = (C_, n1, n2)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import lib as L
        yc, out = ctx.saved_tensors
        C_, n1, n2 = ctx.meta
        dout = dout.float().contiguous()
        dy = torch.empty_like(yc)
        L.check(L.load().kpf_slices_sum_relu_backward(dout.data_ptr(), out.data_ptr(), yc.data_ptr(), dy.data_ptr(), yc.shape[0], C_, n1, n2,
                                                      torch.cuda.current_stream().cuda_stream), "kpf_slices_sum_relu_backward")
        return dy, None, None, None


def slices_sum_relu(y, C_, n1, n2=0):
    return SlicesSumRelu.apply(y, C_, n1, n2)


class RowGather(torch.autograd.Function):
    """out[b, r] = sum_{g < G} w[b, r, g] * src[b, idx[b, r, g]] over feature rows (fp32): the 4-nearest-pixel sampling of the point
    features (model/model.py:297-306; w = closeness) and DESA's ball-query grouping (model/model.py:174; G = 1, no weights).
    src [B, P, C], idx int32 [B, R, G], w [B, R, G] or None.  Gradient w.r.t. src only (indices / closeness carry none in the
    reference either): kpf_row_gather_bwd_f32 inverts the index list per image and adds in entry order — no atomics."""

    @staticmethod
    def forward(ctx, src, idx, w, inv=None):
        """inv: the index tensor's inversion (row_gather_invert(idx, P)) when several gathers share it — the backward then skips its own."""
        from . import lib as L
        src = src.contiguous()
        idx = idx.contiguous()
        B, P, Cc = src.shape
        _, R, G = idx.shape
        assert src.dtype == torch.float32 and idx.dtype == torch.int32 and Cc % 4 == 0
        w = w.detach().float().contiguous() if w is not None else None
        out = torch.empty(B, R, Cc, device=src.device, dtype=torch.float32)
        L.check(L.load().kpf_row_gather_fwd_f32(src.data_ptr(), idx.data_ptr(), w.data_ptr() if w is not None else None, out.data_ptr(), B, P, R, G, Cc,
                                                torch.cuda.current_stream().cuda_stream), "kpf_row_gather_fwd_f32")
        ctx.save_for_backward(idx, w, inv)
        ctx.shape = (B, P, R, G, Cc)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import lib as L
        idx, w, inv = ctx.saved_tensors
        B, P, R, G, Cc = ctx.shape
        dout = dout.float().contiguous()
        dsrc = torch.empty(B, P, Cc, device=dout.device, dtype=torch.float32)
        lib = L.load()
        st = torch.cuda.current_stream().cuda_stream
        if inv is not None:
            L.check(lib.kpf_row_gather_accum_f32(dout.data_ptr(), inv.data_ptr(), inv.data_ptr() + 4 * B * (P + 1), w.data_ptr() if w is not None else None, dsrc.data_ptr(),
                                                 B, P, R, G, Cc, st), "kpf_row_gather_accum_f32")
            return dsrc, None, None, None
        nws = lib.kpf_row_gather_ws_ints(B, P, R, G)
        ws = torch.empty(nws, device=dout.device, dtype=torch.int32)
        L.check(lib.kpf_row_gather_bwd_f32(dout.data_ptr(), idx.data_ptr(), w.data_ptr() if w is not None else None, dsrc.data_ptr(), ws.data_ptr(), nws,
                                           B, P, R, G, Cc, st), "kpf_row_gather_bwd_f32")
        return dsrc, None, None, None


def upsample2x_nhwc(x):
    return Upsample2xNHWC.apply(x)


def maxpool3x3s2_nhwc(x):
    return MaxPool3x3s2NHWC.apply(x)


def row_gather_invert(idx, P):
    """The inversion of an index tensor [B, R, G] over P source rows (kpf_row_gather_invert) for RowGather(inv=...): per image, the list of
    gathered entries per source row in ascending entry order."""
    from . import lib as L
    lib = L.load()
    idx = idx.contiguous()
    B, R, G = idx.shape
    assert idx.dtype == torch.int32
    nws = lib.kpf_row_gather_ws_ints(B, P, R, G)
    ws = torch.empty(nws, device=idx.device, dtype=torch.int32)
    L.check(lib.kpf_row_gather_invert(idx.data_ptr(), ws.data_ptr(), nws, B, P, R, G, torch.cuda.current_stream().cuda_stream), "kpf_row_gather_invert")
    return ws


def row_gather(src, idx, w=None, inv=None):
    """Weighted row gather (RowGather).  Shapes outside the backward kernel's limits (more than 8192 gathered entries or 4096 source rows per
    image: crops beyond 256 x 256, > 2048 points) are decided HERE, before autograd records a node, and take torch.gather — whose backward
    (index_add) is correct but adds with atomics, i.e. is not run-to-run bit-reproducible; the reference's sizes never get there."""
    B, P, Cc = src.shape
    _, R, G = idx.shape
    if not src.requires_grad and (w is None or not w.requires_grad) and src.is_cuda and src.dtype == torch.float32 and idx.dtype == torch.int32 and (Cc % 4 or not src.is_contiguous()):
        # no gradient and a width / layout the quad kernel does not take (the 21 weight-logit channels: channel planes of the NCHW offset map, or a column slice
        # of its NHWC form): one forward launch on the tensor where it lies (any strides)
        from . import lib as L
        out = torch.empty(B, R, Cc, device=src.device, dtype=torch.float32)
        wc = None if w is None else w.float().contiguous()
        L.check(L.load().kpf_row_gather_cols_f32(src.data_ptr(), src.stride(0), src.stride(1), src.stride(2), idx.contiguous().data_ptr(), None if wc is None else wc.data_ptr(), out.data_ptr(), B, P, R, G, Cc,
                                                 torch.cuda.current_stream().cuda_stream), "kpf_row_gather_cols_f32")
        return out
    if R * G > ROW_GATHER_MAX_E or P > ROW_GATHER_MAX_P or Cc % 4:
        g = torch.gather(src, 1, idx.long().reshape(B, R * G, 1).expand(-1, -1, Cc)).view(B, R, G, Cc)
        return (g * w.unsqueeze(-1)).sum(2) if w is not None else g.sum(2)
    return RowGather.apply(src, idx, w, inv)


class Conv2dNHWC(torch.autograd.Function):
    """y = conv2d(x, w) + b on NHWC activations [B, H, W, Cin] (fp32, HIP device), weight in the reference's OIHW layout.
    forward : kpf_conv2d_f32 (f32-input MFMA implicit GEMM).
    backward: dX  = kpf_conv2d_f32 of dY with the spatially flipped, channel-transposed weight (stride 1, any padding); for patchify
                    convolutions (kernel == stride, pad 0) a 1x1 GEMM dY @ W[N][(ky,kx,c)] followed by the pixel un-shuffle;
              dW  = dY^T X as a library GEMM for 1x1, torch.nn.grad.conv2d_weight otherwise (weight gradients are plain reductions
                    over pixels: not on the hand-written path yet);
              db  = sum of dY over pixels.
    Linear layers are the 1x1 case on a [rows, 1, 1, K] view."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, prec="f32", w16=None, key=None, cache=None, groups=1, res=None, gelu_in=False, alias=False, gelu_out=False,
                g_pre=None):
        """prec "bf16" / "f16": operands rounded to 16 bits, fp32 accumulation on the 16-bit MFMA (kpf_conv2d_h16), 16-bit output;
        the weight stays the fp32 master copy and receives an fp32 gradient.  w16: the weight already rounded to the compute type
        (same shape; TrainGraph casts all of them once per step) — the packs are then built from it without per-layer casts.
        groups = G > 1: G convolutions in one launch each way — x [B, H, W, G*Cin] channel-stacked, weight [G*N, Cin, KH, KW] and bias
        [G*N] group-major (the paired backbones' parameters: pair_params), output [B, OH, OW, G*N].
        res: a tensor of the output's shape added in the GEMM's epilogue (y = conv(x) + b + res: the skip path of a Residual block,
        model/hourglass.py:106-119) — one launch less than a separate add; its gradient is dY itself.
        gelu_in: the layer is Linear(gelu(x)) (pwconv2 of a ConvNeXt block, output.dense of a BERT layer: convNeXT/convnext.py:45-46, model/model.py:
        97-104) with x the PRE-activation: gelu runs here (kpf_gelu_forward; its output is also what the weight gradient multiplies) and the backward
        returns d x = (dY W) * gelu'(x) from the data-gradient GEMM's own epilogue (KPF_RES_GELU_GRAD) — no separate GELU-backward pass.
        alias: also return x itself; a second consumer of x (the residual path around a feed-forward) that reads the alias hands its gradient to THIS
        backward, where it rides in the data-gradient GEMM's residual epilogue (dense stride-1 layers without gelu_in).
        gelu_out: returns (z, gelu(z)) with z = x W^T + b from ONE launch (KPF_ACT_GELU_SAVE: the GEMM's epilogue stores both); gelu(z) carries no
        gradient — hand it to the following layer as g_pre together with gelu_in=True (that layer then skips its own GELU pass and returns d z)."""
        assert x.is_cuda and x.dim() == 4
        ctx.mma16 = _HEAD_MMA[0]  # (the product type of this forward: the data-gradient GEMM of the backward takes the same)
        B, H, W, Cin = x.shape
        N, Cw, KH, KW = weight.shape
        cm = 4 if prec == "f32" else 8
        patch = stride == KH == KW and pad == 0 and stride > 1
        use16 = prec != "f32" and w16 is not None
        ctx.groups = groups
        ctx.odd = None
        if groups == 1 and KH == 1 and KW == 1 and stride == 1 and pad == 0 and w16 is None and (Cw % cm or N % cm):
            # A Linear whose widths are not whole channel groups (the 3-d coordinates, 105 pose channels, 131- / 149-wide inputs and 3- / 21-wide
            # outputs of the fusion head: model/model.py:99-104, 254-262, 336).  The GEMM sees the input at the padded width cpad: x either arrives
            # padded (a producer that writes the zero channels itself, e.g. kpf_pose_tokens_f32) or is padded here in one launch; the packed
            # weight rows are zero beyond K already; the output is dense [.., N].  backward pads dY likewise and trims dW / db in the reduce.
            cpad, npad = (Cw + cm - 1) // cm * cm, (N + cm - 1) // cm * cm
            assert Cin in (Cw, cpad), "Conv2dNHWC: input width %d matches neither the weight's %d nor its padded width %d" % (Cin, Cw, cpad)
            tdt = torch.float32 if prec == "f32" else _TDT[prec]
            xc = pad_rows(x, cpad, tdt) if Cin != cpad else (x.to(tdt).contiguous())
            if cache is not None and key is not None:
                pc = cache.get((key, 0), weight, bias, 0, prec, stride=1, pad=0, patchify=False)
            else:
                pc = DevPack.packed(weight, bias, 0, prec, stride=1, pad=0, patchify=False)
            y = _conv_any(_OddPack(pc, cpad), xc, prec)
            ctx.pack = (key, cache)
            assert res is None and not gelu_in and not alias and not gelu_out
            ctx.odd = (Cin, cpad, npad)
            ctx.save_for_backward(xc, weight, None)
            ctx.alias = ctx.gelu_out = False
            ctx.w16, ctx.x_dtype = None, x.dtype
            ctx.conf = (stride, pad, False, bias is not None, prec)
            ctx.bias_ptr = bias.data_ptr() if bias is not None else None
            return y
        assert Cw * groups == Cin and Cw % cm == 0, "Conv2dNHWC: input channels must match and be a multiple of 4 (8 for 16-bit)"
        if groups > 1:
            assert w16 is None and (N // groups) % cm == 0, "grouped Conv2dNHWC: whole channel groups per group"
            patch = False  # (the kx-merging GEMM view of a patchify convolution does not survive channel stacking: general strided form)
            pc = _grouped_pack(cache, key, weight, bias, groups, 0, prec, stride=stride, pad=pad, patchify=False)
        elif cache is not None and key is not None:  # persistent operand, refreshed once per iteration for all layers (PackCache)
            pc = cache.get((key, 0), weight, bias, 0, prec, stride=stride, pad=pad, patchify=patch)
        else:
            pc = DevPack.packed(w16 if use16 else weight, bias, 0, prec, stride=stride, pad=pad, patchify=patch)
        ctx.pack = (key, cache)
        xc = x.float() if prec == "f32" else x.to(_TDT[prec])  # the operand as the GEMM sees it — also what the weight gradient multiplies
        z = None
        if gelu_in:
            from . import lib as L
            assert KH == 1 and KW == 1 and stride == 1 and pad == 0 and xc.numel() % 4 == 0, "gelu_in: Linear layers"
            z = xc.contiguous()
            if g_pre is not None:  # gelu(z) came out of the producing GEMM's epilogue (gelu_out there)
                assert g_pre.shape == z.shape and g_pre.dtype == z.dtype
                xc = g_pre.detach().contiguous()
            else:
                xc = torch.empty_like(z)
                L.check(L.load().kpf_gelu_forward(z.data_ptr(), xc.data_ptr(), _KDT[z.dtype], z.numel(), torch.cuda.current_stream().cuda_stream), "kpf_gelu_forward")
        if gelu_out:
            from . import lib as L
            assert KH == 1 and KW == 1 and stride == 1 and pad == 0 and res is None and groups * 0 == 0, "gelu_out: Linear layers"
            zo = torch.empty(B, H, W, N, device=x.device, dtype=xc.dtype)
            y = _conv_any(pc, xc, prec, flags=L.KPF_ACT_GELU, out2=zo)
            ctx.res_dtype, ctx.alias, ctx.gelu_out = None, bool(alias), True
            ctx.save_for_backward(xc, weight, z)
            ctx.w16 = w16 if use16 else None
            ctx.x_dtype = x.dtype
            ctx.conf = (stride, pad, patch, bias is not None, prec)
            ctx.bias_ptr = bias.data_ptr() if bias is not None else None
            ctx.mark_non_differentiable(y)
            ctx.set_materialize_grads(False)
            return (zo, y, x.view(x.shape)) if alias else (zo, y)
        ctx.gelu_out = False
        y = _conv_any(pc, xc, prec, res=res)
        ctx.res_dtype = None if res is None else res.dtype
        ctx.alias = bool(alias)
        if alias:
            assert not gelu_in and stride == 1 and not patch and groups == 1
            ctx.set_materialize_grads(False)
        ctx.save_for_backward(xc, weight, z)
        ctx.w16 = w16 if use16 else None
        ctx.x_dtype = x.dtype
        ctx.conf = (stride, pad, patch, bias is not None, prec)
        ctx.bias_ptr = bias.data_ptr() if bias is not None else None  # (identifies the bias PARAMETER for DeferredParamGrads' adoption check)
        return (y, x.view(x.shape)) if alias else y

    @staticmethod
    def backward(ctx, dy, g2=None, g3=None):
        with head_mma(int(getattr(ctx, "mma16", 0))):
            return Conv2dNHWC._backward(ctx, dy, g2, g3)

    @staticmethod
    def _backward(ctx, dy, g2=None, g3=None):
        g_alias = (g3 if getattr(ctx, "gelu_out", False) else g2) if ctx.alias else None  # (outputs: y [, gelu(y) without gradient] [, the alias of x])
        x, weight, z = ctx.saved_tensors
        if dy is None:  # (only the alias was used)
            return (g_alias,) + (None,) * 14
        if g_alias is not None:
            assert z is None
            gg_alias = dict(res=g_alias.to(x.dtype).contiguous().view(x.shape))
        else:
            gg_alias = {}
        stride, pad, patch, has_bias, prec = ctx.conf
        from . import lib as L
        gg = dict(res=z, flags=L.KPF_RES_GELU_GRAD) if z is not None else {}  # (gelu_in: the data gradient's epilogue multiplies by gelu'(z))
        B, H, W, Cin = x.shape
        N, _, KH, KW = weight.shape
        dy = dy.contiguous()
        OH, OW = dy.shape[1], dy.shape[2]
        dx = dw = db = None
        cmul = 4 if prec == "f32" else 8  # channel granularity of the GEMM's activation operand
        G = ctx.groups
        dres = dy.to(ctx.res_dtype) if (getattr(ctx, "res_dtype", None) is not None and ctx.needs_input_grad[10]) else None
        if ctx.odd is not None:  # odd-width Linear (see forward): x is the padded operand [M, 1, 1, cpad]
            from . import lib as L
            cin_given, cpad, npad = ctx.odd
            Cw = weight.shape[1]
            tdt = torch.float32 if prec == "f32" else _TDT[prec]
            dyp = pad_rows(dy, npad, tdt) if npad != N else dy.to(tdt).contiguous()
            if ctx.needs_input_grad[0]:
                # dX = dY W: rows of the transposed weight; written at the width the caller's x had (its zero channels receive nothing: a producer
                # that padded x itself never reads them back — concatenations slice their own columns out)
                dx = _conv_any(_dgrad_pack(ctx, weight.detach(), 1, prec, pad=0, n_pad=npad), dyp, prec, out_ld=cin_given).view(B, H, W, cin_given).to(ctx.x_dtype)
            if ctx.needs_input_grad[1]:
                lib = L.load()
                want_db = has_bias and ctx.needs_input_grad[2]
                M = B * H * W
                nws = lib.kpf_conv2d_wgrad_ws_floats(M, npad, cpad)
                ws = torch.empty(nws, device=x.device, dtype=torch.float32)
                dw = torch.empty(tuple(weight.shape), device=x.device, dtype=torch.float32)
                db = torch.empty(N, device=x.device, dtype=torch.float32) if want_db else None
                _wgrad_groups(lib, dyp.data_ptr(), x.data_ptr(), _KDT[tdt], dw, db, ws, nws, 1, (B, H, W, cpad, cpad, H, W, npad, npad, 1, 1, 1, 1, 0, 0, Cw, N),
                              torch.cuda.current_stream().cuda_stream, weight=weight, bias_ptr=ctx.bias_ptr)
            return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None, None
        if G > 1:  # channel-stacked groups: the same three GEMMs, one launch each for all groups
            key, cache = ctx.pack
            n, wd = N // G, weight.detach()
            if ctx.needs_input_grad[0]:
                dyc = dy if prec == "f32" else dy.to(_TDT[prec])
                if stride == KH == KW and pad == 0 and stride > 1:  # patchify: rows of dY @ W[n][(ky,kx,c)] per group, then the pixel un-shuffle
                    g = _conv_any(_grouped_pack(cache, key, wd, None, G, 2, prec, n_pad=n), dyc, prec).view(B, OH, OW, G, KH, KW, Cin // G)
                    # (the un-shuffle and the change to x's type in ONE strided copy)
                    dx = torch.empty(B, OH * KH, OW * KW, Cin, device=g.device, dtype=ctx.x_dtype)
                    dx.view(B, OH, KH, OW, KW, G, Cin // G).copy_(g.permute(0, 1, 4, 2, 5, 3, 6))
                    if dx.shape[1] != H or dx.shape[2] != W:
                        dx = F.pad(dx, (0, 0, 0, W - dx.shape[2], 0, H - dx.shape[1]))
                else:
                    assert stride == 1, "grouped Conv2dNHWC: stride 1 or patchify"
                    dx = _conv_any(_grouped_pack(cache, key, wd, None, G, 1, prec, pad=pad, n_pad=n), dyc, prec, **gg).view(B, H, W, Cin)
                dx = dx.to(ctx.x_dtype)
            if ctx.needs_input_grad[1]:
                dw, db = conv_wgrad_hip(dy, x, weight.shape, stride, pad, has_bias and ctx.needs_input_grad[2], groups=G, weight=weight, bias_ptr=ctx.bias_ptr)
            return dx, dw, db, None, None, None, None, None, None, None, dres, None, None, None, None
        if ctx.needs_input_grad[0]:
            wsrc = ctx.w16 if ctx.w16 is not None else weight.detach()
            npad = (N + cmul - 1) // cmul * cmul
            dy_in = dy if npad == N else F.pad(dy, (0, npad - N))  # the kernel needs whole channel groups: zero channels on dY (and zero weight rows)
            if patch:  # dX[b, oy*s+ky, ox*s+kx, c] = sum_n dY[b,oy,ox,n] W[n,c,ky,kx]: rows of a GEMM, then un-shuffle
                g = _conv_any(_dgrad_pack(ctx, wsrc, 2, prec, n_pad=npad), dy_in, prec).view(B, OH, OW, KH, KW, Cin)
                dx = torch.empty(B, OH * KH, OW * KW, Cin, device=g.device, dtype=ctx.x_dtype)  # (un-shuffle + x's type in one strided copy)
                dx.view(B, OH, KH, OW, KW, Cin).copy_(g.permute(0, 1, 3, 2, 4, 5))
                if dx.shape[1] != H or dx.shape[2] != W:  # rows / columns the strided convolution never read
                    dx = F.pad(dx, (0, 0, 0, W - dx.shape[2], 0, H - dx.shape[1]))
            else:
                if stride != 1:
                    # strided (non-patchify) convolution — ResNet's 3x3/s2 and 1x1/s2 (model/resnet.py:52-55,190-194): the data gradient
                    # is the stride-1 transposed convolution of dY dilated by the stride (zeros between its pixels), laid out so that
                    # the symmetric padding KH-1-pad yields exactly H x W rows (rows the strided convolution never read get zeros)
                    hz, wz = H - KH + 1 + 2 * pad, W - KW + 1 + 2 * pad
                    dil = dy_in.new_zeros(B, hz, wz, npad)
                    dil[:, :(OH - 1) * stride + 1:stride, :(OW - 1) * stride + 1:stride] = dy_in
                    dy_in = dil
                dx = _conv_any(_dgrad_pack(ctx, wsrc, 1, prec, pad=pad, n_pad=npad), dy_in, prec, **gg, **gg_alias).view(B, H, W, Cin)
            dx = dx.to(ctx.x_dtype)
        if ctx.needs_input_grad[1] and Cin % 4 == 0 and N % 4 == 0:
            # hand-written split-K weight gradient (fp32 products and accumulation in every precision mode: the master weight's
            # gradient is not rounded to 16 bits; 16-bit dY / X are read as stored), bias gradient from the same pass
            want_db = has_bias and ctx.needs_input_grad[2]
            grp = DeferredParamGrads.wants(ctx.pack[0], ctx.pack[1], dy, x, KH, KW, stride, pad)
            if grp is not None and want_db:  # the bias gradient is handed over unwritten too: same conditions as for the weight
                bp = grp.by_ptr.get(ctx.bias_ptr)
                if bp is None or bp.numel() != N or bp.grad is not None:
                    grp = None
            if grp is not None:  # small Linear layer: its weight gradient joins the grouped launch after backward
                dyc, xc = dy.contiguous(), x.contiguous()
                dw = torch.empty(tuple(weight.shape), device=x.device, dtype=torch.float32)
                db = torch.empty(N, device=x.device, dtype=torch.float32) if want_db else None
                grp.add(ctx.pack[0], dyc, xc, dw, db, ctx.bias_ptr if want_db else None)
            else:
                dw, db = conv_wgrad_hip(dy, x, weight.shape, stride, pad, want_db, weight=weight, bias_ptr=ctx.bias_ptr)
            return dx, dw, db, None, None, None, None, None, None, None, dres, None, None, None, None
        if ctx.needs_input_grad[1]:
            xw = x if prec == "f32" else x.to(_TDT[prec])  # weight gradient in the compute precision, handed to the fp32 master weight
            dyw = dy if prec == "f32" else dy.to(_TDT[prec])
            if KH == 1 and KW == 1 and stride == 1:
                dw = (dyw.view(-1, N).t() @ xw.reshape(-1, Cin)).view(N, Cin, 1, 1).float()
            else:
                dw = torch.nn.grad.conv2d_weight(xw.permute(0, 3, 1, 2), weight.shape, dyw.permute(0, 3, 1, 2), stride=stride, padding=pad).float()
        if has_bias and ctx.needs_input_grad[2]:
            db = dy.float().view(-1, N).sum(0)
        return dx, dw, db, None, None, None, None, None, None, None, dres, None, None, None, None


def _dgrad_pack(ctx, wsrc, mode, prec, **kw):
    key, cache = ctx.pack
    if cache is not None and key is not None:
        return cache.get((key, mode), wsrc, None, mode, prec, **kw)
    return DevPack.packed(wsrc, None, mode, prec, **kw)


def conv2d_nhwc(x, weight, bias=None, stride=1, pad=0, prec="f32", w16=None, key=None, cache=None, groups=1, res=None):
    return Conv2dNHWC.apply(x, weight, bias, stride, pad, prec, w16, key, cache, groups, res, False, False, False, None)


def linear_hip(x, weight, bias=None, prec="f32", w16=None, key=None, cache=None, groups=1, gelu_in=False, alias=False, gelu_out=False, g_pre=None):
    """nn.Linear on rows [..., K] through the same Function (a 1x1 convolution over a [rows, 1, 1, K] view); groups: see Conv2dNHWC
    (x [..., G*K], weight [G*N, K])."""
    K = x.shape[-1]
    Kw = weight.shape[-1]
    y = Conv2dNHWC.apply(x.reshape(-1, 1, 1, K), weight.reshape(weight.shape[0], Kw, 1, 1), bias, 1, 0, prec,
                         w16.reshape(weight.shape[0], Kw, 1, 1) if w16 is not None else None, key, cache, groups, None, gelu_in, alias, gelu_out,
                         g_pre.reshape(-1, 1, 1, K) if g_pre is not None else None)
    if gelu_out:
        outs = (y[0].view(*x.shape[:-1], weight.shape[0]), y[1].view(*x.shape[:-1], weight.shape[0]))
        return outs + (y[2].view(x.shape),) if alias else outs
    if alias:
        return y[0].view(*x.shape[:-1], weight.shape[0]), y[1].view(x.shape)
    return y.view(*x.shape[:-1], weight.shape[0])


class GraphedTrainStep:
    """One training iteration (train-mode forward, loss, backward, optimiser step) replayed from captured hipGraphs.

    The eager iteration is host-bound (several thousand small launches: ~127 ms at B = 32 for ~40 ms of device work); the integer
    decisions come from device kernels, BatchNorm statistics and AdamW state are device tensors (`capturable=True`), so after the
    eager warm-up iterations on a side stream the iteration is captured once and replayed with the batch copied into static buffers.
    What the host still decides is frozen at capture time unless it is handed over as device data: build the optimiser with
    `make_optimizer(..., capturable=True)` (learning rate in a device scalar that StepLR updates in place) and pass the epoch of the
    loss schedule as a 0-d device tensor inside the batch (`kpfusion_loss(epoch=batch["epoch"])` gates the spatial terms on the device).

    One process (dist_mod None): a single graph [zero grads, forward, loss, backward, optimiser step].
    Data parallel (dist_mod = torch.distributed, one process per GPU): graph A [zero grads, forward, loss, backward, pack the
    gradients into flat buckets] -> eager all-reduce of the buckets over RCCL (SUM; a handful of large collectives, reverse
    registration order like `parallel.GradBucketReducer`) -> graph B [average, unpack into .grad, optimiser step].  The reduction is
    not overlapped with backward here (backward is inside a graph); it moves the payload once over xGMI (268 MB for ConvNeXt-T) where
    the eager iteration would spend 3x the whole step on launch overhead.
    usage:  step = GraphedTrainStep(model, optimizer, loss_fn, example_batch[, dist_mod=dist, params=live]);  loss = step(batch)"""

    def __init__(self, model, optimizer, loss_fn, batch, warmup=3, dist_mod=None, params=None, bucket_mb=64.0, group=None, dp_mode=None,
                 grad_payload="f32", collective="allreduce", scaler=None):
        """dp_mode (data parallel only): "overlap" — ONE graph in which each bucket's collective is a node on RCCL's stream, launched by a
        post-accumulate-grad hook the moment the bucket's last gradient exists, so the reduction runs under the rest of backward (what
        DistributedDataParallel does under CUDA graphs; the replacement of train.py:263-265's DataParallel reduce); "split" — graph A,
        eager collectives, graph B (the only form a backend that cannot be captured allows: gloo).  None: "overlap" for the nccl backend
        unless KPF_DP_GRAPH=split, and a fallback to "split" if the capture with collectives fails.
        grad_payload "f32" | "bf16": the type the gradients travel in (bf16 halves the bytes on xGMI: 134 MB for ConvNeXt-T; the mean is then
        the mean of bf16-rounded gradients — not bit-equal to the fp32 form).  collective "allreduce" | "rs_ag": one all-reduce per bucket,
        or reduce-scatter + all-gather of the (padded) bucket — the same bytes per link on the xGMI mesh, two schedulable halves."""
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.scaler = scaler  # LossScaler (fp16 training): loss * scale before backward; check, unscaled-or-skipped step and scale update inside the graph
        if scaler is not None and grad_payload != "f32":
            raise ValueError("GraphedTrainStep: a LossScaler goes with fp32 gradient payloads")
        self.dist, self.group = dist_mod, group
        bucket_mb = float(os.environ.get("KPF_DP_BUCKET_MB", bucket_mb))  # (tuning aid)
        assert grad_payload in ("f32", "bf16") and collective in ("allreduce", "rs_ag")
        self.grad_payload, self.collective = grad_payload, collective
        if dist_mod is not None and dp_mode is None:
            be = dist_mod.get_backend(group)
            dp_mode = "overlap" if (be == "nccl" and os.environ.get("KPF_DP_GRAPH", "overlap") != "split") else "split"
        self.dp_mode = dp_mode if dist_mod is not None else None
        self._deferred_ids = set()
        from .graphs import assert_replay_is_sound, prepare_training_graphs
        prepare_training_graphs()
        assert_replay_is_sound(next(iter(batch.values())).device)  # (once per process: refuses a runtime that mis-replays reductions)
        self.static = {k: v.detach().clone() for k, v in batch.items()}
        self.params = [p for p in (params if params is not None else model.parameters()) if p.requires_grad]
        self._named = dict(model.named_parameters())
        mine = {id(p) for g in optimizer.param_groups for p in g["params"]}
        self._outside = [p for p in model.parameters() if id(p) not in mine]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        self._acc_counts = {}
        with torch.cuda.stream(side):
            for it in range(warmup):
                count_hooks = []
                if it == warmup - 1 and self.dp_mode == "overlap":  # how many times each parameter's gradient is accumulated in one backward pass
                    self._acc_counts = {}
                    bump = lambda q: self._acc_counts.__setitem__(id(q), self._acc_counts.get(id(q), 0) + 1)
                    count_hooks = [q.register_post_accumulate_grad_hook(bump) for q in self.params]
                self._forward_backward()
                for h in count_hooks:
                    h.remove()
                self._reduce_eager()
                self._opt_step()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        # what the warm-up iterations established (the one-graph data-parallel form fixes its buckets before backward is captured): the
        # parameters that receive a gradient, in the order backward produces them, and those whose gradient is deferred past backward
        self._warm_live = [p for p in reversed(self.params) if p.grad is not None]
        self._warm_deferred = set(self._deferred_ids)
        self.opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        self.graph_b = None
        import gc
        gc.collect()  # nothing may be finalised while the stream is capturing: an older CUDAGraph (or any tensor whose deleter touches
        gc.disable()  # the device) collected in the middle of a capture aborts the process
        try:
            self._capture(bucket_mb, group)
        finally:
            gc.enable()

    def _capture(self, bucket_mb, group):
        for cache in getattr(self.model, "__dict__", {}).get("_pack_cache", {}).values():
            cache.build_table()  # (operands the warm-up iterations' backward registered: the capture's refresh is then the one-launch form)
        if self.dist is None:
            with torch.cuda.graph(self.graph):
                self.loss = self._forward_backward()
                self._opt_step()
            return
        self.world = self.dist.get_world_size(group)
        if self.dp_mode == "overlap":
            err = None
            try:
                self._capture_overlap(bucket_mb)
            except Exception as e:  # noqa: BLE001 — a runtime / RCCL build that cannot capture collectives: the two-graph form still works
                err = e
            # The form is agreed on by the whole group (a capture only records, so nothing has been exchanged yet): if ANY rank failed,
            # every rank takes the two-graph form — ranks replaying collectives as graph nodes beside ranks issuing eager all-reduces
            # between two graphs would be a mismatched collective sequence (a hang, or silently mixed gradients).
            ok = torch.tensor([0 if err is not None else 1], device=next(iter(self.params)).device, dtype=torch.int32)
            if self.world > 1:
                self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 1:
                return
            import warnings
            warnings.warn("GraphedTrainStep: capturing the bucket collectives failed on %s (%s); every rank uses the two-graph form" % (
                "this rank" if err is not None else "another rank", "%s: %s" % (type(err).__name__, err) if err is not None else "see its log"))
            self._remove_hooks()
            self.dp_mode = "split"
            torch.cuda.synchronize()
            self.opt.zero_grad(set_to_none=True)
            self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._forward_backward()
            self._make_buckets(bucket_mb)  # (the gradients now exist: static tensors of the graph's pool)
            for flat, views, grads in self.buckets:
                torch._foreach_copy_(views, grads)
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, pool=self.graph.pool()):
            for flat, views, grads in self.buckets:
                if self.world > 1:
                    flat.div_(self.world)
                torch._foreach_copy_(grads, views)
            self._opt_step()

    def _opt_step(self):
        if self.scaler is None:
            self.opt.step()
        else:
            self.opt.step(scaler=self.scaler)

    def _forward_backward(self):
        self.opt.zero_grad(set_to_none=True)
        for p in self._outside:  # (parameters that get a gradient but are not the optimiser's: their stale .grad would be added to, not replaced)
            p.grad = None
        dpg = DeferredParamGrads(self._named)
        with dpg:  # small Linear layers' weight gradients, LayerNorm / layer-scale parameter sums: grouped launches after backward
            loss = self.loss_fn(self.model, self.static)
            (loss if self.scaler is None else self.scaler.scale(loss)).backward()
        self._deferred_ids = {id(p) for p in getattr(dpg, "last_deferred", [])}
        return loss.detach()

    # ---- data parallel, one graph: bucket collectives as graph nodes, launched from gradient hooks during backward ----
    def _flat_bucket(self, plist):
        pdt = torch.bfloat16 if self.grad_payload == "bf16" else plist[0].dtype  # (gradients have their parameter's type)
        n = sum(p.numel() for p in plist)
        pad = (-n) % self.world if self.collective == "rs_ag" else 0  # (reduce-scatter wants equal shards)
        flat = torch.zeros(n + pad, dtype=pdt, device=plist[0].device)
        offs, off = [], 0
        for q in plist:
            offs.append(off)
            off += q.numel()
        shard = torch.empty((n + pad) // self.world, dtype=pdt, device=flat.device) if self.collective == "rs_ag" else None
        # pending: accumulation events still to come before the bucket is complete (a parameter used twice in the forward is accumulated twice: its
        # gradient is only final after the second event — counted in the last warm-up iteration)
        pending = sum(max(1, getattr(self, "_acc_counts", {}).get(id(q), 1)) for q in plist)
        return {"flat": flat, "params": plist, "offs": offs, "shard": shard, "pending": pending, "work": None}

    def _launch_bucket(self, b):
        if self.collective == "rs_ag":
            self.dist.reduce_scatter_tensor(b["shard"], b["flat"], op=self.dist.ReduceOp.SUM, group=self.group)
            b["work"] = self.dist.all_gather_into_tensor(b["flat"], b["shard"], group=self.group, async_op=True)
        else:
            b["work"] = self.dist.all_reduce(b["flat"], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        slot = self._slot.get(id(p))
        if slot is None or not self._hooks_armed:
            return
        b, i = slot
        b["pending"] -= 1
        cur = torch.cuda.current_stream(p.device)
        b.setdefault("streams", {})[cur.cuda_stream] = cur  # (backward nodes run on their forward's stream: the unpaired backbones use two)
        if b["pending"] == 0:  # the bucket's last gradient exists: pack it (one multi-tensor copy, not one copy per parameter) and start its collective
            if DeferredParamGrads.active is not None:  # (weight gradients whose batched reduce is still pending: issue it, on the pass's own stream)
                DeferredParamGrads.active.flush_reduces()
            for st in b["streams"].values():  # gradients enqueued on other streams must be complete before this stream reads them
                if st.cuda_stream != cur.cuda_stream:
                    cur.wait_stream(st)
            self._pack_bucket(b)
            self._launch_bucket(b)

    @staticmethod
    def _pack_bucket(b):
        have = [(o, q) for o, q in zip(b["offs"], b["params"]) if q.grad is not None]
        if have:
            torch._foreach_copy_([b["flat"][o:o + q.numel()].view_as(q.grad) for o, q in have], [q.grad for _, q in have])  # (casts when the payload is bf16)

    def _remove_hooks(self):
        for h in getattr(self, "_hooks", []):
            h.remove()
        self._hooks = []
        self._hooks_armed = False

    def _capture_overlap(self, bucket_mb):
        """Bucket composition is taken from the warm-up iterations: parameters that received a gradient, in reverse registration order
        (the order backward produces them); those whose gradient DeferredParamGrads fills only after backward (the small Linear layers, the
        LayerNorm / layer-scale sums) go to late buckets that are reduced after the flush — they are a few MB of the payload."""
        cap = int(bucket_mb * 1024 * 1024)
        live = self._warm_live
        if not live:
            raise RuntimeError("no parameter received a gradient in the warm-up iterations")
        early = [p for p in live if id(p) not in self._warm_deferred]
        late = [p for p in live if id(p) in self._warm_deferred]

        def split(plist):
            out, cur, nb = [], [], 0
            for q in plist:
                b = q.numel() * (2 if self.grad_payload == "bf16" else q.element_size())
                if cur and (nb + b > cap or cur[0].dtype != q.dtype):
                    out.append(cur)
                    cur, nb = [], 0
                cur.append(q)
                nb += b
            if cur:
                out.append(cur)
            return out

        self._early = [self._flat_bucket(pl) for pl in split(early)]
        self._late = [self._flat_bucket(pl) for pl in split(late)]
        self.buckets = [(b["flat"], None, None) for b in self._early + self._late]  # (payload_bytes() / introspection)
        self._slot = {id(q): (b, i) for b in self._early for i, q in enumerate(b["params"])}
        self._hooks = [q.register_post_accumulate_grad_hook(self._on_grad) for b in self._early for q in b["params"]]
        self._hooks_armed = True
        inv = 1.0 / self.world
        self.opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._forward_backward()  # hooks pack and launch the early buckets while backward runs
            self._hooks_armed = False
            for b in self._late:  # written by the grouped launches that closed backward
                torch._foreach_copy_([b["flat"][o:o + q.numel()].view_as(q.grad) for o, q in zip(b["offs"], b["params"])], [q.grad for q in b["params"]])
                self._launch_bucket(b)
            for b in self._early:
                if b["pending"] > 0:  # a parameter of the bucket got no gradient in this pass although the warm-up saw one: reduce what is there
                    for st in b.get("streams", {}).values():
                        torch.cuda.current_stream().wait_stream(st)
                    self._pack_bucket(b)
                    self._launch_bucket(b)
            for b in self._early + self._late:
                if b["work"] is not None:
                    b["work"].wait()
                if self.world > 1:
                    b["flat"].mul_(inv)
                if b["flat"].dtype == b["params"][0].dtype:
                    # the reduced gradients stay where the collective left them: `.grad` becomes a view of the bucket (what the optimiser nodes of the graph
                    # read from here on; backward keeps writing the tensors it was captured with, the pack above moves them) — no unpack copy
                    for o, q in zip(b["offs"], b["params"]):
                        if q.grad is not None:
                            q.grad = b["flat"][o:o + q.numel()].view_as(q.grad)
                else:  # bf16 payload: back to the parameters' type
                    torch._foreach_copy_([q.grad for q in b["params"]], [b["flat"][o:o + q.numel()].view_as(q.grad) for o, q in zip(b["offs"], b["params"])])
            self._opt_step()
        self._remove_hooks()  # the captured graph no longer needs them (replays do not run Python)

    def _reduce_eager(self):
        """Warm-up iterations only: per-parameter all-reduce (keeps the replicas in step before the capture)."""
        if self.dist is None or self.dist.get_world_size(self.group) == 1:
            return
        w = self.dist.get_world_size(self.group)
        for p in self.params:
            if p.grad is not None:
                self.dist.all_reduce(p.grad, group=self.group)
                p.grad.div_(w)

    def _make_buckets(self, bucket_mb):
        cap = int(bucket_mb * 1024 * 1024)
        self.buckets, cur, cur_bytes = [], [], 0
        live = [p for p in reversed(self.params) if p.grad is not None]

        def close(plist):
            flat = torch.zeros(sum(p.numel() for p in plist), dtype=plist[0].grad.dtype, device=plist[0].device)
            views, off = [], 0
            for q in plist:
                views.append(flat[off:off + q.numel()].view_as(q.grad))
                off += q.numel()
            self.buckets.append((flat, views, [q.grad for q in plist]))

        for p in live:
            nb = p.numel() * p.grad.element_size()
            if cur and (cur_bytes + nb > cap or cur[0].grad.dtype != p.grad.dtype):
                close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            close(cur)

    def time_collectives(self, time_one):
        """Diagnostics (bench.py's `collectives` record; EVERY rank must call it): each gradient bucket's collective issued eagerly on the step's own static
        buffers — the same call the captured graph holds as a node — timed by `time_one(fn, nbytes)`.  The buffers are zeroed first and are rewritten by the
        next replay's pack, so the step is not disturbed."""
        if self.dist is None:
            return []
        out = []
        for flat, _, _ in self.buckets:
            flat.zero_()
            if self.dp_mode == "overlap" and self.collective == "rs_ag":
                shard = torch.empty(flat.numel() // self.world, dtype=flat.dtype, device=flat.device)

                def fn(flat=flat, shard=shard):
                    self.dist.reduce_scatter_tensor(shard, flat, op=self.dist.ReduceOp.SUM, group=self.group)
                    self.dist.all_gather_into_tensor(flat, shard, group=self.group)
            else:
                def fn(flat=flat):
                    self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
            rec = time_one(fn, flat.numel() * flat.element_size())
            rec["dtype"] = str(flat.dtype).replace("torch.", "")
            out.append(rec)
        return out

    def payload_bytes(self):
        return sum(f.numel() * f.element_size() for f, _, _ in self.buckets) if self.dist is not None else 0  # (per iteration and rank, either form)

    def __call__(self, batch):
        # the batch into the graph's static inputs: ONE multi-tensor copy per dtype instead of a copy node per tensor (nine 5-us launches per iteration); a tensor
        # that already IS the static input (the caller filled `self.static[k]` in place) is skipped
        todo = [(self.static[k], v) for k, v in batch.items() if v is not self.static[k]]
        same = [(d, v) for d, v in todo if torch.is_tensor(v) and v.device == d.device and v.shape == d.shape]
        if same:
            torch._foreach_copy_([d for d, _ in same], [v for _, v in same])
        for d, v in todo:
            if not (torch.is_tensor(v) and v.device == d.device and v.shape == d.shape):
                d.copy_(v)
        self.graph.replay()  # (dp_mode "overlap": the collectives are nodes of this graph)
        if self.graph_b is not None:
            works = [self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True) for flat, _, _ in self.buckets]
            for w in works:
                w.wait()
            self.graph_b.replay()
        return self.loss
