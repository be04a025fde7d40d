"""Import-time shims that let the *reference* KeypointFusion package (read-only at
/root/reference) be imported on this CPU-only build container, so that golden vectors can be
generated from the reference's own forward.  Used ONLY by tests/golden/gen_golden.py (and by the
optional oracle-vs-reference cross-check test, which skips when /root/reference is absent, as on the GPU box).

Nothing here is product code and nothing here copies reference source: every shim is a stand-in
for a *third-party* package the reference imports but this image lacks:

  cv2, torchvision, timm, pycocotools   -> empty stubs (only imported, never used on the hot path)
  pointnet2_ops==3.0.0 (requirements.txt:15; call sites model/model.py:16,158,174)
      -> pure-torch restatement of QueryAndGroup's published semantics (ball_query = first
         `nsample` indices in index order with d^2 < r^2, unfilled slots repeat the first hit;
         group = gather; output cat[(xyz[idx]-new_xyz), features[idx]]).  The reference has no
         test for this op, so parity at this boundary is defined by this restatement
         ("parity unpinned" for the ball-query op — see DESIGN.md).
  transformers 4.25.1 -> 5.x drift: torch_int_div re-added, PreTrainedModel.init_weights patched.
  Tensor.cuda / Module.cuda -> identity (hard-coded .cuda() at model/model.py:50, transfusion_head.py:692).
"""
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("KPF_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "model"))


class _QueryAndGroup(nn.Module):
    """pointnet2_ops.pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=True) restated in torch."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    @staticmethod
    def ball_query(radius, nsample, xyz, new_xyz):
        # xyz B x N x 3, new_xyz B x S x 3 -> idx B x S x nsample (int64)
        B, N, _ = xyz.shape
        S = new_xyz.shape[1]
        # d2 accumulated in the CUDA kernel's order: dx*dx + dy*dy + dz*dz
        d = new_xyz.unsqueeze(2) - xyz.unsqueeze(1)
        d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
        r32 = torch.tensor(radius, dtype=torch.float32)
        within = d2 < (r32 * r32)  # fp32 radius2, as in the CUDA kernel
        idx = torch.zeros(B, S, nsample, dtype=torch.long)
        ar = torch.arange(N)
        for b in range(B):
            for s in range(S):
                hits = ar[within[b, s]]
                if hits.numel() == 0:
                    continue
                k = min(nsample, hits.numel())
                idx[b, s, :] = hits[0]
                idx[b, s, :k] = hits[:k]
        return idx

    def forward(self, xyz, new_xyz, features=None):
        idx = self.ball_query(self.radius, self.nsample, xyz, new_xyz)
        B, S, ns = idx.shape
        flat = idx.reshape(B, S * ns)
        g_xyz = torch.gather(xyz, 1, flat.unsqueeze(-1).expand(-1, -1, 3)).reshape(B, S, ns, 3)
        g_xyz = (g_xyz - new_xyz.unsqueeze(2)).permute(0, 3, 1, 2)  # B 3 S ns
        if features is None:
            return g_xyz
        C = features.shape[1]
        g_f = torch.gather(features, 2, flat.unsqueeze(1).expand(-1, C, -1)).reshape(B, C, S, ns)
        return torch.cat([g_xyz, g_f], dim=1) if self.use_xyz else g_f


def install_shims():
    if getattr(install_shims, "_done", False):
        return
    import transformers  # noqa: F401  (must be imported before torchvision is stubbed)
    import transformers.pytorch_utils as pu
    from transformers.modeling_utils import PreTrainedModel
    import transformers.models.bert.modeling_bert  # noqa: F401

    if not hasattr(pu, "torch_int_div"):
        pu.torch_int_div = lambda a, b: torch.div(a, b, rounding_mode="floor")
    PreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)

    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    _stub("cv2")
    tvm = _stub("torchvision.models")
    for n in ("resnet18", "resnet34", "resnet50", "resnet101"):
        setattr(tvm, n, lambda pretrained=False, **kw: nn.Module())
    tvt = _stub("torchvision.transforms")
    _stub("torchvision", models=tvm, transforms=tvt)

    def trunc_normal_(t, std=1.0, **kw):
        return nn.init.trunc_normal_(t, std=std)

    class DropPath(nn.Identity):
        def __init__(self, p=0.0):
            super().__init__()

    tl = _stub("timm.models.layers", trunc_normal_=trunc_normal_, DropPath=DropPath)
    tr = _stub("timm.models.registry", register_model=lambda f: f)
    tm = _stub("timm.models", layers=tl, registry=tr)
    _stub("timm", models=tm)
    pc = _stub("pycocotools.coco", COCO=object)
    _stub("pycocotools", coco=pc)
    p2u = _stub("pointnet2_ops.pointnet2_utils", QueryAndGroup=_QueryAndGroup)
    _stub("pointnet2_ops", pointnet2_utils=p2u)

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    install_shims._done = True


def load_reference():
    """Returns (KPFusion class, loader instance with img_size=128, flip=1)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    install_shims()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    cwd = os.getcwd()
    os.chdir(REF_ROOT)  # BertConfig.from_pretrained("./config/") is cwd-relative (model/model.py:222)
    try:
        from model.model import KPFusion  # noqa
        from dataloader.loader import loader as LoaderBase
    finally:
        os.chdir(cwd)
    ld = LoaderBase("/nonexistent", "test", 128, "joint_mean", "DexYCB")
    ld.flip = 1
    return KPFusion, ld


def build_reference_model(net):
    KPFusion, ld = load_reference()
    cwd = os.getcwd()
    os.chdir(REF_ROOT)
    try:
        torch.manual_seed(0)
        m = KPFusion(net, "", 21, "dexycb", "")
    finally:
        os.chdir(cwd)
    m.eval()
    return m, ld


# ----------------------------------------------------------------------------------------------------------------
# Stand-alone heads
# ----------------------------------------------------------------------------------------------------------------
def load_reference_aux():
    """Returns the reference modules model.cbam and model.hourglass (torch-only, import as they are)."""
    install_shims()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from model import cbam, hourglass
    return cbam, hourglass


class _R:
    """Stand-in for a chumpy array: the reference only reads `.r` (manolayer.py:69-88)."""

    def __init__(self, a):
        self.r = a


def load_reference_mano_head(mano_model):
    """The reference's mano_regHead (model/mano_head.py:177-231) built over `mano_model` (dict of arrays with MANO_RIGHT.pkl's
    field names).  chumpy — needed only to unpickle the licence-restricted MANO file — is absent from this image, so the one
    function that touches it, `ready_arguments` (util/manopth/mano/webuser/smpl_handpca_wrapper_HAND_only.py:22-68, a file
    loader, no arithmetic on the path), is replaced by a stub handing the arrays over; ManoLayer.forward and mano_head's
    rotation conversions are the reference's own code."""
    import numpy as np
    import scipy.sparse as sp
    install_shims()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    def ready_arguments(fname_or_dict, posekey4vposed="pose"):
        m = mano_model
        return {"hands_components": np.asarray(m["hands_components"]), "hands_mean": np.asarray(m["hands_mean"]),
                "betas": _R(np.zeros(10)), "shapedirs": _R(np.asarray(m["shapedirs"])), "posedirs": _R(np.asarray(m["posedirs"])),
                "v_template": _R(np.asarray(m["v_template"])), "J_regressor": sp.csc_matrix(np.asarray(m["J_regressor"])),
                "weights": _R(np.asarray(m["weights"])), "f": np.asarray(m["f"]), "kintree_table": np.asarray(m["kintree_table"])}

    for name in ("util.manopth.mano", "util.manopth.mano.webuser"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = []
            sys.modules[name] = mod
    w = types.ModuleType("util.manopth.mano.webuser.smpl_handpca_wrapper_HAND_only")
    w.ready_arguments = ready_arguments
    sys.modules[w.__name__] = w
    import util.manopth.manopth as real_pkg
    import util.manopth.manopth.manolayer as real_layer
    real_layer.ready_arguments = ready_arguments
    sys.modules["manopth"] = real_pkg  # model/mano_head.py:5 imports the top-level name
    sys.modules["manopth.manolayer"] = real_layer
    from model import mano_head
    return mano_head
