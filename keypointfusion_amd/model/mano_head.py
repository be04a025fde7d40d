"""Drop-in counterpart of the reference's `model.mano_head.mano_regHead` (model/mano_head.py:177-225): MLP -> 16 x 6D rotations ->
rotation matrices -> axis-angle -> MANO layer (util/manopth/manopth/manolayer.py) -> 778 vertices + 21 joints (mm).

The MANO hand model is licence-restricted data the user supplies: `mano_root` must hold `MANO_RIGHT.pkl` (read without chumpy by
`load_mano_pkl`), or pass `mano_model` = dict of arrays with the pickle's field names.  Without either, a *synthetic* hand of the same
structure is used (keypointfusion_amd.weights.synthetic_mano_model) so that the head is runnable and testable here.  The reference
hard-codes its author's home directory for mano_root (model/mano_head.py:180); the argument replaces that."""
import os
import pickle

import numpy as np
import torch

from ..spec import mano_head_spec
from ..weights import mano_layer_buffers, synthetic_mano_model
from ._base import SpecModule


class _ChStub:
    """Receives the state of a pickled chumpy object (chumpy itself is not needed to read the arrays)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, st):
        self.__dict__.update(st if isinstance(st, dict) else {"state": st})


class _Unpickler(pickle.Unpickler):
    def find_class(self, mod, name):
        if mod.split(".")[0] == "chumpy":
            return _ChStub
        return super().find_class(mod, name)


def _ch_value(v):
    """numpy value of a field that may be a plain array or a (possibly `Select`-wrapped) chumpy array."""
    if isinstance(v, _ChStub):
        d = v.__dict__
        if "x" in d:
            return np.asarray(d["x"])
        if "a" in d and "idxs" in d:  # chumpy.reordering.Select: a.ravel()[idxs].reshape(preferred_shape)
            return _ch_value(d["a"]).ravel()[np.asarray(d["idxs"])].reshape(d["preferred_shape"])
        raise ValueError("unsupported chumpy object in the MANO pickle: %s" % sorted(d))
    return v


def load_mano_pkl(path):
    """MANO_{LEFT,RIGHT}.pkl -> dict of numpy arrays (the fields ManoLayer reads, manolayer.py:64-104)."""
    with open(path, "rb") as f:
        dd = _Unpickler(f, encoding="latin1").load()
    out = {}
    for k in ("v_template", "shapedirs", "posedirs", "weights", "hands_components", "hands_mean", "kintree_table", "f"):
        out[k] = np.asarray(_ch_value(dd[k]))
    jr = dd["J_regressor"]
    out["J_regressor"] = np.asarray(jr.toarray() if hasattr(jr, "toarray") else jr)
    return out


class mano_regHead(SpecModule):
    def __init__(self, feature_size=1024, mano_neurons=(1024, 512), mano_root=None, mano_model=None, seed=0):
        super().__init__()
        if mano_model is None:
            path = os.path.join(mano_root, "MANO_RIGHT.pkl") if mano_root else None
            mano_model = load_mano_pkl(path) if path and os.path.exists(path) else synthetic_mano_model(seed)
        parents = [(-1 if int(p) > 1000 else int(p)) for p in np.asarray(mano_model["kintree_table"])[0]]
        if parents != [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]:
            raise ValueError("unexpected MANO kinematic tree %s" % parents)
        self.fingertip_vertex_idx = [728, 353, 442, 576, 694]  # model/mano_head.py:182 (attribute kept; unused by forward)
        self.pose6d_size, self.mano_pose_size = 16 * 6, 16 * 3
        self._materialise(mano_head_spec(feature_size, tuple(mano_neurons)), seed, prefix="mano_head.", values=mano_layer_buffers(mano_model),
                          buffers=("mano_layer.",))

    def forward(self, features):
        self._require_gpu(features)
        if self.training:  # autograd-connected outputs on this module's Parameters (keypointfusion_amd/heads_train.py: convolutions / Linears on the HIP GEMM)
            from ..heads_train import mano_head_train_forward
            with torch.cuda.device(features.device):
                return mano_head_train_forward(self, features)
        from ..heads import ManoHeadPlan
        plan = self._plan(features.device, lambda sd, dev: ManoHeadPlan(sd, dev))
        with torch.cuda.device(features.device):
            return plan(features)
