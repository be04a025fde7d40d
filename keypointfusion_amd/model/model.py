"""Drop-in counterpart of the reference's `model.model.KPFusion` (model/model.py:354-426) for MI355X.

Same constructor, same forward signature and return structure, same state-dict keys (keypointfusion_amd/spec.py),
so `train.py` / `demo_RGBD.py`-style callers can switch `from model.model import KPFusion` to
`from keypointfusion_amd.model.model import KPFusion`.  The module owns its parameters as ordinary
torch Parameters/buffers in the reference (NCHW/OIHW) layout; at the first forward after construction or after a
load_state_dict they are repacked into kernel layouts (keypointfusion_amd.engine) and all compute runs in
libkpf_hip.so on the calling thread's current device/stream.  There is no CPU or PyTorch-op fallback.

Differences that are deliberate: nothing is downloaded (`pretrain` is accepted and ignored — the
reference pulls torchvision / fbaipublicfiles weights), no hard-coded `.cuda()`, no host synchronisation inside
forward, dead sub-modules (decoder layers 0-2, BERT embeddings/poolers, ConvNeXt head) own their keys but cost nothing.
"""
import threading

import torch
import torch.nn as nn

from ..spec import kpfusion_spec, parse_net
from ..weights import _draw  # noqa: F401  (init scales shared with the synthetic generator)
from ._base import _Node, _attach  # noqa: F401  (shared with the stand-alone heads)


class KPFusion(nn.Module):
    def __init__(self, net, pretrain, joint_num, dataset, mano_dir, kernel_size=1, seed=0, crop_size=128):
        """Positional arguments as the reference's (model/model.py:355).  Two keyword-only-in-practice extensions: `seed` of the initial weights, and
        `crop_size` — 128 is the reference, whose fusion block hard-codes a 32 x 32 feature map (nn.Linear(32 * 32, 1), model/model.py:264); another
        multiple of 32 (256: the crop size BASELINE's metric is quoted on) builds the WIDE extension SURVEY section 0 allows: identical architecture and
        kernels, `block{1,2}.fc_spatial2joint_feature.weight` sized [1, (crop_size / 4)^2] — a checkpoint of the reference loads into everything else."""
        super().__init__()
        if joint_num != 21:
            raise ValueError("KPFusion is wired for 21 joints (Block_KPFusion, model/model.py:209)")
        self.crop_size = int(crop_size)
        self.net = net
        self.joint_num = joint_num
        self.kernel_size = kernel_size
        self.dim = 128
        self.num_stages = 2
        self.family, self.size = parse_net(net)
        import numpy as np
        import zlib
        for name, shape, dtype, init in kpfusion_spec(net, self.crop_size):
            rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
            val = torch.from_numpy(np.asarray(_draw(rng, shape, init)).copy()).reshape(shape)
            is_buffer = name.endswith(("running_mean", "running_var", "num_batches_tracked"))
            _attach(self, name, val, is_buffer)
        self._plans = {}
        self._plan_lock = threading.Lock()
        # storage precision of the backbones: "f32" (the reference's arithmetic, default), "bf16" or "f16" (16-bit activations and
        # weights, fp32 accumulation; the fusion head stays fp32) — set before the first forward or at any time
        self.precision = "f32"
        self.train_dropout = 0.1  # dropout probability of the transformer layers in train mode (config/config.json; transfusion_head.py:95)
        self.use_graphs = False  # opt-in: replay each forward from a captured hipGraph (eval / no_grad, fixed shapes)

    # -- weight repacking ------------------------------------------------------------------------------------
    def _state_version(self):
        """Sum of the in-place version counters of every parameter and buffer (detects optimiser steps and weight surgery).  The module
        tree is walked once (3 ms for 1580 tensors); later calls read the cached tensor list (0.17 ms)."""
        ts = self.__dict__.get("_tensor_list")
        if ts is None:
            ts = self.__dict__["_tensor_list"] = list(self.parameters()) + list(self.buffers())
        return sum(t._version for t in ts)

    # -- copy.deepcopy / pickle: the per-device caches (packed weights, captured graphs, streams, locks) are rebuilt on demand ----
    _CACHE_ATTRS = ("_plans", "_plan_lock", "_tensor_list", "_train_side_stream", "_w16_shadow", "_pack_cache", "_drop_rng")

    def __getstate__(self):
        st = self.__dict__.copy()
        for k in self._CACHE_ATTRS:
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self.__dict__["_plans"] = {}
        self.__dict__["_plan_lock"] = threading.Lock()
        self.__dict__["_tensor_list"] = None

    def _apply(self, fn, *a, **k):  # .to() / .cuda() / .float(): tensors may be replaced
        self.__dict__["_tensor_list"] = None
        self.__dict__.pop("_pack_cache", None)
        self._plans.clear()
        return super()._apply(fn, *a, **k)

    def _plan(self, device):
        """Kernel-layout weights for `device`, rebuilt when any parameter changed (load_state_dict, optimiser step)."""
        if self.precision not in ("f32", "bf16", "f16"):
            raise ValueError("KPFusion.precision must be 'f32', 'bf16' or 'f16' (got %r)" % (self.precision,))
        key = (device.type, device.index, self.precision)
        ver = self._state_version()
        with self._plan_lock:
            ent = self._plans.get(key)
            if ent is None or ent[0] != ver:
                from ..engine import ModelPlan
                sd = {k: v.detach() for k, v in self.state_dict().items()}
                ent = (ver, ModelPlan(sd, self.net, device, self.precision))
                self._plans[key] = ent
            return ent[1]

    def _load_from_state_dict(self, *a, **k):
        self._plans.clear()
        self.__dict__["_tensor_list"] = None
        return super()._load_from_state_dict(*a, **k)

    # -- forward ---------------------------------------------------------------------------------------------
    @staticmethod
    def _require_gpu(t):
        if not t.is_cuda:
            raise RuntimeError("keypointfusion_amd.KPFusion runs on MI355X only: inputs must be on a HIP device "
                               "(there is no CPU fallback; the CPU oracle lives under oracle/ for tests)")

    def forward_backbones(self, img_rgb, img):
        """The two UNet streams only (BASELINE.json configs[1]): returns (img_offset, img_feat, img_offset_rgb,
        img_feat_rgb) as NCHW tensors like the reference's backbone_d / backbone_rgb calls (model/model.py:397-398)."""
        self._require_gpu(img)
        from ..engine import nhwc_to_nchw
        plan = self._plan(img.device)
        with torch.cuda.device(img.device):
            (od, fd), (orgb, frgb) = plan.backbones(img.detach().float().contiguous(), img_rgb.detach().float().contiguous())
            return od, nhwc_to_nchw(fd), orgb, nhwc_to_nchw(frgb)

    def forward(self, img_rgb, img, pcl, loader, center, M, cube, cam_para, kernel=0.8, writer=None, ii=0):
        self._require_gpu(img)
        if img.shape[-1] != self.crop_size:
            # the reference hard-codes nn.Linear(32*32, 1) (model/model.py:264): the full model exists at the size it was built for only
            raise RuntimeError("mat1 and mat2 shapes cannot be multiplied: the fusion block needs %dx%d crops (got %d); use forward_backbones() for "
                               "other sizes, or build KPFusion(..., crop_size=%d)" % (self.crop_size, self.crop_size, img.shape[-1], img.shape[-1]))
        img_size = int(getattr(loader, "img_size", self.crop_size))
        flip = int(getattr(loader, "flip", 1))
        if self.training:
            # train mode: batch-statistics BatchNorm, dropout, autograd-connected outputs on the module's own
            # Parameters — keypointfusion_amd/train_graph.py (convolutions / Linears forward + data-gradient on the HIP GEMM)
            # precision "f16" in train mode: fp16 GEMM operands / activations with fp32 master weights like "bf16", but gradients underflow below 6e-8 —
            # scale the loss (training.LossScaler: loss * scale before backward, checked / unscaled / skipped step on the device)
            from ..train_graph import TrainGraph
            with torch.cuda.device(img.device):
                return TrainGraph(self).forward(img_rgb, img, pcl, center, M, cube, cam_para, float(kernel), img_size, flip)
        plan = self._plan(img.device)
        with torch.cuda.device(img.device):
            if self.use_graphs and not torch.is_grad_enabled():
                return plan.forward_graphed(img_rgb, img, pcl, center, M, cube, cam_para, float(kernel), img_size, flip)
            return plan.forward(img_rgb, img, pcl, center, M, cube, cam_para, float(kernel), img_size, flip)
