"""The torch-free C restatement of the integer sub-ops (oracle/kpf_index_oracle.c) against the torch oracle (pinned to the imported
reference by tests/test_oracle_golden.py) and against the reference-generated index fixtures; on the GPU box it is the second checker
of the HIP kernels' index outputs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, synthetic_sd
from keypointfusion_amd.weights import synthetic_batch
from oracle import kpf_oracle as O

_SO = os.path.join(ROOT, "oracle", "_build", "libkpf_index_oracle.so")


def clib():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(_SO)
    P = C.c_void_p
    lib.kpf_oracle_ball_query.argtypes = [P, C.c_int, P, C.c_int, C.c_float, C.c_int, P]
    lib.kpf_oracle_ball_query.restype = None
    lib.kpf_oracle_top4.argtypes = [P, C.c_int, P, C.c_int, P, P]
    lib.kpf_oracle_top4.restype = None
    return lib


def c_ball_query(radius, nsample, xyz, new_xyz):
    lib = clib()
    xyz, new_xyz = np.ascontiguousarray(xyz, np.float32), np.ascontiguousarray(new_xyz, np.float32)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.zeros((B, S, nsample), np.int32)
    for b in range(B):
        lib.kpf_oracle_ball_query(xyz[b].ctypes.data, N, new_xyz[b].ctypes.data, S, radius, nsample, out[b].ctypes.data)
    return out


def c_top4(pcl, img_xyz):
    lib = clib()
    pcl, img_xyz = np.ascontiguousarray(pcl, np.float32), np.ascontiguousarray(img_xyz, np.float32)
    B, N, _ = pcl.shape
    P = img_xyz.shape[1]
    idx = np.zeros((B, N, 4), np.int32)
    d2 = np.zeros((B, N, 4), np.float32)
    for b in range(B):
        lib.kpf_oracle_top4(pcl[b].ctypes.data, N, img_xyz[b].ctypes.data, P, idx[b].ctypes.data, d2[b].ctypes.data)
    return idx, d2


def test_c_ball_query_equals_torch_oracle_and_edge_cases():
    rng = np.random.default_rng(0)
    xyz = (rng.random((3, 1045, 3)).astype(np.float32) - 0.5) * 1.2
    new = xyz[:, -21:].copy()
    for r in (0.1, 0.2, 0.4):
        ref = O.ball_query(r, 64, torch.from_numpy(xyz), torch.from_numpy(new)).numpy()
        np.testing.assert_array_equal(c_ball_query(r, 64, xyz, new), ref)
    # no hit at all -> zeros; fewer hits than slots -> first hit repeated; a point exactly on the radius is outside (strict <)
    pts = np.array([[[0, 0, 0], [0.5, 0, 0], [0.05, 0, 0], [10, 10, 10]]], np.float32)
    q = np.array([[[0, 0, 0], [5, 5, 5], [0.25, 0, 0]]], np.float32)
    got = c_ball_query(0.25, 4, pts, q)
    np.testing.assert_array_equal(got[0, 0], [0, 2, 0, 0])
    np.testing.assert_array_equal(got[0, 1], [0, 0, 0, 0])
    np.testing.assert_array_equal(got[0, 2], [2, 2, 2, 2])  # |0.25-0| == r and |0.5-0.25| == r are not inside
    np.testing.assert_array_equal(got, O.ball_query(0.25, 4, torch.from_numpy(pts), torch.from_numpy(q)).numpy())


@pytest.mark.parametrize("net", ["convnext-tiny", "resnet-18"])
def test_c_index_ops_equal_reference_fixture(net):
    """On the golden batch: the C top-4 indices equal the reference's img2pcl_index output and the C ball query equals the index
    tensors captured from the reference's three groupers (through the oracle's intermediate joints)."""
    z = np.load(os.path.join(GOLDEN, "kpf_%s_B2_S128.npz" % net))
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(2, 128, seed=1).items()}
    d = torch.nn.functional.interpolate(b["img"], [32, 32])
    img_xyz = O.img_xyz_grid(d, b["center"], b["M"], b["cube"], b["cam_para"], 128, 1)
    idx, d2 = c_top4(b["pcl"].numpy(), img_xyz.numpy())
    ref_idx = z["pcl_index"]
    same = idx == ref_idx
    if not same.all():  # ATen's topk may order exact ties differently: a mismatch must be a tie in d^2
        n_bad = np.argwhere(~same)
        for bb, n, k in n_bad:
            assert sorted(idx[bb, n]) == sorted(ref_idx[bb, n]) or np.isclose(d2[bb, n, k], d2[bb, n, 3], rtol=0, atol=0)
    c = 1.0 / (d2 + 1e-8)
    np.testing.assert_allclose(c / (c.sum(-1, keepdims=True) + 1e-8), z["pcl_closeness"], rtol=1e-5, atol=1e-7)
    sd = synthetic_sd("KPFusion-" + net)
    aux = {}
    res, _ = O.kpfusion_forward(sd, b["img_rgb"], b["img"], b["pcl"], b["center"], b["M"], b["cube"], b["cam_para"], 0.8, aux=aux)
    for blk, joints in ((1, aux["joint_xyz0"]), (2, res[3])):  # block 2 starts from block 1's final_TR joints (model/model.py:413-418)
        xyz = torch.cat([b["pcl"], joints], 1).numpy()
        for gi, r in enumerate((0.1, 0.2, 0.4)):
            np.testing.assert_array_equal(c_ball_query(r, 64, xyz, joints.numpy()), z["b%d_ball%d" % (blk, gi)].astype(np.int32))


@pytest.mark.gpu
def test_hip_index_kernels_equal_c_oracle():
    """kpf_img2pcl_top4_f32 and kpf_ball_group_f32 against the C checker on identical inputs (bit-exact index tensors)."""
    from keypointfusion_amd import lib as L
    from keypointfusion_amd.engine import _ptr, _stream, crop_inverse
    assert torch.cuda.is_available()
    lib = L.load()
    dev = torch.device("cuda:0")
    B, N = 3, 1024
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, 128, seed=4).items()}
    g = {k: v.to(dev) for k, v in b.items()}
    clo = torch.empty(B, N, 4, device=dev)
    idx = torch.empty(B, N, 4, device=dev, dtype=torch.int32)
    ixyz = torch.empty(B, 1024, 3, device=dev)
    L.check(lib.kpf_img2pcl_top4_f32(_ptr(g["pcl"]), _ptr(g["img"]), _ptr(g["center"]), _ptr(crop_inverse(g["M"])), _ptr(g["cube"]), _ptr(g["cam_para"]), _ptr(clo),
                                     _ptr(idx), _ptr(ixyz), B, N, 128, 32, 128, 1, _stream()), "top4")
    torch.cuda.synchronize()
    cidx, _ = c_top4(b["pcl"].numpy(), ixyz.cpu().numpy())  # same pixel positions -> the search itself must agree exactly
    np.testing.assert_array_equal(idx.cpu().numpy(), cidx)
    rng = np.random.default_rng(2)
    joints = torch.from_numpy((rng.random((B, 21, 3)).astype(np.float32) - 0.5) * 0.8)
    X = torch.zeros(B * N, 128, device=dev)
    JF = torch.zeros(B * 21, 128, device=dev)
    G = torch.empty(3, B * 21 * 64, 132, device=dev)
    bi = torch.empty(3, B * 21, 64, device=dev, dtype=torch.int32)
    jd = joints.to(dev)
    L.check(lib.kpf_ball_group_f32(_ptr(g["pcl"]), _ptr(jd), _ptr(X), _ptr(JF), 128, _ptr(G), _ptr(bi), B, N, 0.1, 0.2, 0.4, _stream()), "ball")
    torch.cuda.synchronize()
    xyz = torch.cat([b["pcl"], joints], 1).numpy()
    for gi, r in enumerate((0.1, 0.2, 0.4)):
        np.testing.assert_array_equal(bi[gi].cpu().numpy().reshape(B, 21, 64), c_ball_query(r, 64, xyz, joints.numpy()))
