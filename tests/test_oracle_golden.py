"""CPU: the oracle (oracle/kpf_oracle.py) against the golden vectors generated from the imported reference
(tests/golden/gen_golden.py).  This is what pins the oracle; the GPU tests then compare the HIP path to the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, synthetic_sd
from keypointfusion_amd.weights import synthetic_batch, synthetic_state_dict, state_dict_digest
from keypointfusion_amd.spec import kpfusion_spec
from oracle import kpf_oracle as O

ATOL, RTOL = 2e-5, 1e-4  # fp32 CPU restatement vs reference; integer tensors exact
NETS = ["convnext-tiny", "resnet-18", "resnet-50", "convnext-base"]  # full-model fixtures (convnext-base: the network of BASELINE configs[4])
BACKBONE_NETS = ["convnext-tiny", "convnext-small", "convnext-base", "convnext-large", "resnet-18", "resnet-50", "resnet-101"]


def close(a, b, atol=ATOL, rtol=RTOL):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    err = np.abs(a - b)
    assert a.shape == b.shape
    assert (err <= atol + rtol * np.abs(b)).all(), "max abs err %.3e" % err.max()


@pytest.mark.parametrize("net", ["convnext-tiny", "convnext-base", "resnet-18", "resnet-50", "resnet-101"])
def test_state_dict_contract(net):
    """Key names, shapes, dtypes and order equal the reference module tree's state_dict."""
    ref = json.load(open(os.path.join(GOLDEN, "state_keys_%s.json" % net)))
    mine = kpfusion_spec("KPFusion-" + net)
    assert [(k, tuple(s), d) for k, s, d in ref] == [(k, s, d) for k, s, d, _ in mine]


@pytest.mark.parametrize("net", NETS)
def test_full_forward_matches_reference_fixture(net):
    z = np.load(os.path.join(GOLDEN, "kpf_%s_B2_S128.npz" % net))
    np_sd = synthetic_state_dict("KPFusion-" + net, 0)
    assert state_dict_digest(np_sd) == str(z["weights_sha256"]), "synthetic weight generator drifted"
    sd = synthetic_sd("KPFusion-" + net)
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(2, 128, seed=1).items()}
    aux = {}
    res, sws = O.kpfusion_forward(sd, b["img_rgb"], b["img"], b["pcl"], b["center"], b["M"], b["cube"], b["cam_para"], 0.8, aux=aux)
    close(res[0][:, :, ::2, ::2], z["img_offset_sub"])
    close(res[1][:, :, ::2, ::2], z["img_offset_rgb_sub"])
    for i, k in enumerate(("r3d1", "r2d1", "r3d2", "r2d2")):
        close(res[2 + i], z[k])
    close(sws[0], z["sw1"])
    close(sws[1], z["sw2"])
    close(aux["img_feat"][:, ::8, ::4, ::4], z["img_feat_sub"])
    close(aux["img_feat_rgb"][:, ::8, ::4, ::4], z["img_feat_rgb_sub"])
    close(aux["joint_xyz0"], z["joint_xyz0"])
    close(aux["pcl_closeness"], z["pcl_closeness"])
    # integer outputs: bit-exact
    assert np.array_equal(aux["pcl_index"].numpy(), z["pcl_index"])
    for i in (1, 2):
        a = aux["block%d" % i]
        close(a["joint_feat_desa"], z["b%d_FA" % i])
        close(a["h_init"], z["b%d_h_init" % i])
        close(a["dec"], z["b%d_dec" % i])
        for g in range(3):
            assert np.array_equal(a["ball_idx"][g].numpy(), z["b%d_ball%d" % (i, g)])
    # "argmax indices" 
    d = torch.nn.functional.interpolate(b["img"], [32, 32])
    for off, key in ((res[0], "argmax_w_d"), (res[1], "argmax_w_rgb")):
        w = off[:, 84:].masked_fill(d > 0.99, -1e8).reshape(2, 21, -1)
        assert np.array_equal(w.argmax(-1).numpy(), z[key])
    assert np.array_equal(sws[0].reshape(2, 21, -1).argmax(-1).numpy(), z["argmax_sw1"])
    assert np.array_equal(sws[1].reshape(2, 21, -1).argmax(-1).numpy(), z["argmax_sw2"])


@pytest.mark.parametrize("net", BACKBONE_NETS)
def test_backbones_other_size(net):
    z = np.load(os.path.join(GOLDEN, "backbone_%s_B1_S64.npz" % net))
    sd = synthetic_sd("KPFusion-" + net)
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(1, 64, seed=1).items()}
    od, fd, orgb, frgb = O.backbones_forward(sd, b["img_rgb"], b["img"])
    close(od, z["img_offset"])
    close(orgb, z["img_offset_rgb"])
    close(fd[:, ::4], z["img_feat_sub"])
    close(frgb[:, ::4], z["img_feat_rgb_sub"])


def test_ball_query_edge_cases():
    """Empty neighbourhood -> zeros; fewer than nsample hits -> padded with the first hit; index order kept."""
    xyz = torch.tensor([[[0.0, 0, 0], [0.05, 0, 0], [0.3, 0, 0], [0.06, 0, 0]]])
    q = torch.tensor([[[0.0, 0, 0], [5.0, 5, 5]]])
    idx = O.ball_query(0.1, 4, xyz, q)
    assert idx[0, 0].tolist() == [0, 1, 3, 0]
    assert idx[0, 1].tolist() == [0, 0, 0, 0]
    idx = O.ball_query(0.1, 2, xyz, q)
    assert idx[0, 0].tolist() == [0, 1]


def test_all_background_depth_gives_uniform_softmax():
    """model/model.py:488 fills with -1e8, not -inf: an all-background crop decodes to the grid centroid, not NaN."""
    off = torch.randn(1, 105, 32, 32)
    depth = torch.ones(1, 1, 128, 128)
    j = O.offset2joint_weight(off, depth, 0.8)
    assert torch.isfinite(j).all()
    assert torch.allclose(j[..., :2], torch.zeros(1, 21, 2), atol=1e-5)
    assert torch.allclose(j[..., 2], torch.ones(1, 21), atol=1e-5)
