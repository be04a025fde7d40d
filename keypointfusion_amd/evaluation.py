"""Per-stage decode and error metrics of the reference's test loop (train.py:326-396, 470-488;
util/generateFeature.py:166-195 (feature2joint -> offset2joint_weight), :676-703 (rigid_align); util/eval_utils.py:38-81).

The decode of the two dense stages (masked soft-argmax + un-crop/back-projection) runs in the HIP library (the same kernel the
forward uses for its own initial joints); the metrics are host-side numpy like the reference's (they consume B x 21 x 3 arrays).
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L

STAGE_TYPE = (1, 1, 2, 3, 2, 3)  # config.py:74 — depth backbone, RGB backbone, (3D, 2D) x 2 fusion blocks


def decode_stage(result, stage_type, img, center, M, cube, cam_para, kernel=0.8, img_size=128, flip=1):
    """Normalised xyz joints (B x 21 x 3, on the device) of one forward result, as train.py:330-362 / demo_RGBD.py:115-124 derive
    them.  stage_type 1: dense offset maps -> GFM.feature2joint('weight_offset') -> loader.uvd_nl2xyznl_tensor, always with the
    DEPTH image as mask/depth channel (also for the RGB stream: train.py:339).  stage_type 2/3: the result already is xyz."""
    if stage_type in (2, 3):
        return result
    if stage_type != 1:
        raise ValueError("stage_type %r is not produced by KPFusion (config.stage_type = [1,1,2,3,2,3])" % (stage_type,))
    lib = L.load()
    B, ch, F, _ = result.shape
    assert ch == 105
    prep = lambda t: t.detach().float().contiguous()
    result, img, center, M, cube, cam_para = map(prep, (result, img, center, M, cube, cam_para))
    uvd = torch.empty(B, 21, 3, device=result.device, dtype=torch.float32)
    xyz = torch.empty_like(uvd)
    p = lambda t: C.c_void_p(t.data_ptr())
    from .engine import crop_inverse
    M = crop_inverse(M)  # kpf_offset2joint_f32 takes M^-1 in the reference's torch.linalg.inv rounding order
    L.check(lib.kpf_offset2joint_f32(p(result), p(img), p(center), p(M), p(cube), p(cam_para), p(uvd), p(xyz), B, img.shape[-1], F,
                                     float(kernel), int(img_size), int(flip), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "kpf_offset2joint_f32")
    return xyz


def xyz2error(pred, gt, center, cube):
    """train.py:470-488 — per-joint Euclidean error in mm: |(pred - gt) * cube/2|, B x J."""
    pred, gt, center, cube = (np.asarray(t.detach().cpu() if hasattr(t, "detach") else t, dtype=np.float32) for t in (pred, gt, center, cube))
    B, J, _ = pred.shape
    c = np.tile(center.reshape(B, 1, -1), [1, J, 1])
    s = np.tile(cube.reshape(B, 1, -1), [1, J, 1])
    a = pred * s / 2 + c
    b = gt * s / 2 + c
    e = (a - b) * (a - b)
    if J == 23:  # reference's NYU subset selection
        e = e[:, [0, 2, 4, 6, 8, 10, 12, 14, 16, 17, 18, 21, 22, 20], :]
    return np.sqrt(np.sum(e, axis=2))


def rigid_transform_3d(A, B):
    """util/generateFeature.py:676-696 — similarity transform (scale, rotation, translation) of A onto B (Umeyama)."""
    n, _ = A.shape
    ca, cb = np.mean(A, axis=0), np.mean(B, axis=0)
    H = np.dot(np.transpose(A - ca), B - cb) / n
    U, s, V = np.linalg.svd(H)
    R = np.dot(np.transpose(V), np.transpose(U))
    if np.linalg.det(R) < 0:
        s[-1] = -s[-1]
        V[2] = -V[2]
        R = np.dot(np.transpose(V), np.transpose(U))
    var = np.var(A, axis=0).sum()
    c = 1 / var * np.sum(s)
    t = -np.dot(c * R, np.transpose(ca)) + np.transpose(cb)
    return c, R, t


def rigid_align(A, B):
    """util/generateFeature.py:698-703."""
    c, R, t = rigid_transform_3d(A, B)
    return np.transpose(np.dot(c * R, np.transpose(A))) + t


def evaluate_batch(results, img, xyz_gt, center, M, cube, cam_para, stage_type=STAGE_TYPE, kernel=0.8, img_size=128, flip=1):
    """One iteration of Trainer.test (train.py:326-383): per stage the B x 21 joint errors (mm), their batch mean, and the
    Procrustes-aligned mean error."""
    out = []
    gt = xyz_gt.detach().cpu().numpy()
    for i, st in enumerate(stage_type):
        xyz = decode_stage(results[i], st, img, center, M, cube, cam_para, kernel, img_size, flip)
        err = xyz2error(xyz, xyz_gt, center, cube)
        pa = 0.0
        xn = xyz.detach().cpu().numpy()
        for b in range(xn.shape[0]):
            al = rigid_align(xn[b], gt[b])
            pa = pa + xyz2error(al[None], gt[b][None], center[b:b + 1], cube[b:b + 1])
        pa = pa / xn.shape[0]
        out.append({"joint_errors": err, "mean_error": float(np.mean(np.mean(err, axis=-1))), "pa_mean_error": float(np.mean(pa))})
    return out


def _trapz(y, x):
    y, x = np.asarray(y, dtype=np.float64), np.asarray(x, dtype=np.float64)
    return float(np.sum((y[1:] + y[:-1]) * (x[1:] - x[:-1]) / 2.0))


def pck_auc(per_joint_errors, val_min=0.0, val_max=50.0, steps=20):
    """util/eval_utils.py:38-81 (get_measures / _get_pck / calc_auc), as eval_auc calls it (thresholds 0..50 in 20 steps):
    per_joint_errors = list of 21 sequences of errors.  Returns (auc, pck_curve, thresholds, auc_20_50) where auc_20_50 is the
    reference's "Area under curve between 20mm - 50mm" (curve from threshold index 8 on, thresholds * 1000)."""
    th = np.linspace(val_min, val_max, steps)
    norm = _trapz(np.ones_like(th), th)
    aucs, curves = [], []
    for j in range(len(per_joint_errors)):
        d = np.asarray(per_joint_errors[j], dtype=np.float64)
        curve = np.array([np.mean((d <= t).astype("float")) for t in th])
        curves.append(curve)
        aucs.append(_trapz(curve, th) / norm)
    curve = np.mean(np.array(curves), 0)
    x = th[8:] * 1000.0
    sub = _trapz(curve[8:], x) / _trapz(np.ones_like(x), x)
    return float(np.mean(aucs)), curve, th, float(sub)
