"""CPU checks of the stand-alone heads: the oracle restatements against the golden vectors generated from the
imported reference (tests/golden/gen_golden_aux.py), the state-dict contracts, and host-side MANO plumbing."""
import json
import os

import numpy as np
import pytest
import torch

from keypointfusion_amd import spec as S
from keypointfusion_amd.weights import synthetic_from_spec, synthetic_mano_head_state, synthetic_mano_model, synthetic_tensor
from oracle import aux_oracle as A
from oracle.kpf_oracle import to_torch_sd

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CBAM_CASES = ((128, False, 2, 16, 16), (64, True, 2, 8, 12), (256, False, 1, 8, 8))
POSENET_CASES = ((2, 128, 1, 128), (1, 256, 2, 64))


def _close(a, b, atol=2e-5, rtol=1e-4):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape
    err = (a - b).abs()
    assert bool((err <= atol + rtol * b.abs()).all()), "max abs err %.3e" % float(err.max())


def cbam_case(C, nosp, B, H, W):
    tag = "cbam_C%d_%d" % (C, int(nosp))
    sd = to_torch_sd(synthetic_from_spec(S.cbam_spec(C, no_spatial=nosp), 0, prefix=tag + "."))
    return tag, sd, torch.from_numpy(synthetic_tensor((B, C, H, W), 3, tag))


def posenet_case(nstack, dim, B, Sz):
    tag = "posenet_n%d_d%d" % (nstack, dim)
    sd = to_torch_sd(synthetic_from_spec(S.posenet_spec(nstack, 21, dim), 0, prefix=tag + "."))
    return tag, sd, torch.from_numpy(synthetic_tensor((B, 1, Sz, Sz), 4, tag, -1.0, 1.0))


def mano_case(B=4):
    return to_torch_sd(synthetic_mano_head_state(0)), torch.from_numpy(synthetic_tensor((B, 1024), 5, "mano_features"))


def test_cbam_oracle_matches_reference_vectors():
    g = np.load(os.path.join(GOLD, "aux_cbam.npz"))
    for case in CBAM_CASES:
        tag, sd, x = cbam_case(*case)
        _close(A.cbam_channel_scale(sd, x), g[tag + "_scale"])
        out = A.cbam_forward(sd, x, case[1])
        if case[1]:
            _close(out[:, ::4], g[tag + "_out"])
        else:
            _close(out[0][:, ::4], g[tag + "_out0"])
            _close(out[1][:, ::4], g[tag + "_out1"])
            _close(out[0] + out[1], x * A.cbam_channel_scale(sd, x)[:, :, None, None], atol=1e-5)  # the tuple partitions x_out


def test_posenet_oracle_matches_reference_vectors():
    g = np.load(os.path.join(GOLD, "aux_posenet.npz"))
    for case in POSENET_CASES:
        tag, sd, x = posenet_case(*case)
        preds, feat = A.posenet_forward(sd, x, case[0])
        _close(preds[:, :, ::2, ::2], g[tag + "_preds_sub"], atol=5e-5)
        _close(feat[:, ::4], g[tag + "_feat_sub"], atol=5e-5)


def test_mano_oracle_matches_reference_vectors():
    g = np.load(os.path.join(GOLD, "aux_mano.npz"))
    sd, feats = mano_case()
    out = A.mano_head_forward(sd, feats)
    for k in ("mano_shape", "mano_pose", "mano_pose_aa"):
        _close(out[k], g[k])
    for k in ("verts3d", "joints3d"):
        _close(out[k], g[k], atol=2e-3)  # millimetres
    # structure: rotations are orthonormal, the axis-angle round trip reproduces them, wrist joint = root of the chain
    R = out["mano_pose"].reshape(-1, 3, 3)
    _close(R @ R.transpose(1, 2), torch.eye(3).expand_as(R), atol=1e-5)
    _close(A.rodrigues(out["mano_pose_aa"].reshape(-1, 3)), R, atol=1e-5)


def test_mano_zero_pose_is_the_shaped_template():
    """Identity rotations and zero betas leave the template untouched (LBS weights are row-stochastic)."""
    sd, _ = mano_case()
    verts, joints = A.mano_layer(sd, torch.zeros(1, 48), torch.zeros(1, 10))
    _close(verts[0], sd["mano_layer.th_v_template"][0] * 1000, atol=1e-3)
    _close(joints[0, 0], (sd["mano_layer.th_J_regressor"] @ sd["mano_layer.th_v_template"][0])[0] * 1000, atol=1e-3)


def test_head_state_dict_contracts():
    with open(os.path.join(GOLD, "state_keys_aux.json")) as f:
        ref = json.load(f)
    mine = {"cbam_128": S.cbam_spec(128), "cbam_64_nospatial": S.cbam_spec(64, no_spatial=True), "posenet_2_21_128": S.posenet_spec(2, 21, 128),
            "posenet_1_21_256": S.posenet_spec(1, 21, 256), "mano_regHead": S.mano_head_spec()}
    for k, sp in mine.items():
        assert [[n, list(s), d] for n, s, d, _ in sp] == ref[k], k


def test_head_modules_own_the_reference_keys_and_refuse_cpu():
    from keypointfusion_amd.model.cbam import CBAM
    from keypointfusion_amd.model.hourglass import PoseNet
    from keypointfusion_amd.model.mano_head import mano_regHead
    with open(os.path.join(GOLD, "state_keys_aux.json")) as f:
        ref = json.load(f)
    for k, m, x in (("cbam_128", CBAM(128), torch.zeros(1, 128, 8, 8)), ("posenet_1_21_256", PoseNet(1, 21, 256), torch.zeros(1, 1, 64, 64)),
                    ("mano_regHead", mano_regHead(), torch.zeros(1, 1024))):
        assert [[n, list(v.shape), str(v.dtype).replace("torch.", "")] for n, v in m.state_dict().items()] == ref[k]
        m.eval()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m(x)
    # buffers vs parameters as in the reference: the MANO arrays are buffers, the MLP is trainable
    names = {n for n, _ in mano_regHead().named_parameters()}
    assert names == {"mano_base_layer.0.weight", "mano_base_layer.0.bias", "mano_base_layer.2.weight", "mano_base_layer.2.bias",
                     "pose_reg.weight", "pose_reg.bias", "shape_reg.weight", "shape_reg.bias"}


def test_mano_pickle_reader_handles_chumpy_objects(tmp_path):
    """load_mano_pkl reads a MANO-format pickle whose arrays are chumpy objects without importing chumpy."""
    import pickle
    import sys
    import types
    import scipy.sparse as sp
    from keypointfusion_amd.model.mano_head import load_mano_pkl
    m = synthetic_mano_model(1)
    ch = types.ModuleType("chumpy")
    chch = types.ModuleType("chumpy.ch")
    chre = types.ModuleType("chumpy.reordering")

    class Ch:
        def __init__(self, x=None):
            self.x = x

    class Select:
        def __init__(self, a, idxs, shape):
            self.a, self.idxs, self.preferred_shape = a, idxs, shape

    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    Select.__module__, Select.__qualname__ = "chumpy.reordering", "Select"
    chch.Ch, chre.Select = Ch, Select
    sys.modules.update({"chumpy": ch, "chumpy.ch": chch, "chumpy.reordering": chre})
    try:
        full = np.concatenate([m["shapedirs"].ravel(), np.zeros(7)])
        dd = dict(m, v_template=np.asarray(m["v_template"]), posedirs=Ch(m["posedirs"]),
                  shapedirs=Select(Ch(full), np.arange(m["shapedirs"].size), m["shapedirs"].shape), J_regressor=sp.csc_matrix(m["J_regressor"]))
        path = tmp_path / "MANO_RIGHT.pkl"
        with open(path, "wb") as f:
            pickle.dump(dd, f, protocol=2)
    finally:
        for k in ("chumpy", "chumpy.ch", "chumpy.reordering"):
            sys.modules.pop(k, None)
    got = load_mano_pkl(str(path))
    for k in ("v_template", "shapedirs", "posedirs", "weights", "J_regressor", "hands_components"):
        np.testing.assert_array_equal(got[k], m[k])
