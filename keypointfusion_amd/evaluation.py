"""Per-stage decode and error metrics of the reference's test loop (train.py:326-396, 470-488;
util/generateFeature.py:166-195 (feature2joint -> offset2joint_weight), :676-703 (rigid_align); util/eval_utils.py:38-81).

The decode of the two dense stages (masked soft-argmax + un-crop/back-projection) runs in the HIP library (the same kernel the
forward uses for its own initial joints); the per-joint errors and the Procrustes alignment are batched tensor expressions on the
device (one 3 x 3 SVD batch per stage, no per-sample host loop); only PCK / AUC, which consume the accumulated error lists of a whole
evaluation, are numpy.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L

STAGE_TYPE = (1, 1, 2, 3, 2, 3)  # config.py:74 — depth backbone, RGB backbone, (3D, 2D) x 2 fusion blocks
NYU_SCORED_JOINTS = (0, 2, 4, 6, 8, 10, 12, 14, 16, 17, 18, 21, 22, 20)  # train.py:484-486: of NYU's 23 predicted joints these 14 are scored, in this order


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


def _as_tensor(t, like=None):
    if torch.is_tensor(t):
        return t.detach()
    return torch.as_tensor(np.asarray(t), device=like.device if like is not None else None)


def xyz2error(pred, gt, center, cube):
    """Per-joint Euclidean error in mm (what train.py:470-488 computes): B x J, or B x 14 for NYU's 23-joint predictions, of which the
    reference scores the 14 joints of `NYU_SCORED_JOINTS` (train.py:484-486).  Normalised joints are scaled by cube / 2; the crop
    centre the reference adds to both sides cancels in the difference, so it is not added here (no 700 mm + 0.01 mm cancellation in
    fp32).  Tensors stay on their device; numpy in -> numpy out."""
    was_np = not torch.is_tensor(pred)
    p = _as_tensor(pred).float()
    g = _as_tensor(gt, p).to(p).float()
    half = _as_tensor(cube, p).to(p).float().reshape(p.shape[0], 1, -1) / 2
    e = torch.linalg.vector_norm((p - g) * half, dim=-1)
    if p.shape[1] == 23:
        e = e[:, list(NYU_SCORED_JOINTS)]
    return e.cpu().numpy() if was_np else e


def similarity_align(A, B):
    """Batched similarity Procrustes (Umeyama 1991; the result of util/generateFeature.py:676-703 `rigid_align` for every sample at once):
    scale c, rotation R, translation t minimising |c R a_j + t - b_j|^2 over the J points of each sample, applied to A.
    A, B: [N, J, 3] tensors on any device (fp32 or fp64); one batched 3 x 3 SVD instead of a host loop over samples.
    Reflections: when det(V U^T) < 0 the smallest singular direction is flipped (and its singular value counted negative in the scale)."""
    J = A.shape[1]
    ca, cb = A.mean(1, keepdim=True), B.mean(1, keepdim=True)
    A0, B0 = A - ca, B - cb
    H = A0.transpose(1, 2) @ B0 / J                                  # cross-covariance, [N, 3, 3]
    U, S, Vh = torch.linalg.svd(H)
    d = torch.sign(torch.linalg.det(Vh.transpose(1, 2) @ U.transpose(1, 2)))
    flip = torch.ones_like(S)
    flip[:, 2] = d
    R = (Vh.transpose(1, 2) * flip.unsqueeze(1)) @ U.transpose(1, 2)  # V diag(1, 1, d) U^T
    scale = (S * flip).sum(-1) / (A0.pow(2).sum((1, 2)) / J)          # trace(D S) / total variance of A
    return scale.view(-1, 1, 1) * (A0 @ R.transpose(1, 2)) + cb


def rigid_align(A, B):
    """One sample (J x 3) or a batch (N x J x 3); numpy in -> numpy out (float64 kept), tensor in -> tensor out."""
    was_np = not torch.is_tensor(A)
    a = _as_tensor(A)
    b = _as_tensor(B, a).to(a)
    single = a.dim() == 2
    out = similarity_align(a[None] if single else a, b[None] if single else b)
    out = out[0] if single else out
    return out.cpu().numpy() if was_np else out


def evaluate_batch(results, img, xyz_gt, center, M, cube, cam_para, stage_type=STAGE_TYPE, kernel=0.8, img_size=128, flip=1):
    """One iteration of Trainer.test (train.py:326-383): per stage the B x 21 joint errors (mm), their batch mean, and the
    Procrustes-aligned mean error.  Everything up to the three scalars per stage is computed on the device for the whole batch (the
    reference aligns sample by sample in numpy on the host, train.py:346-357: a synchronisation and B small SVDs per stage); one
    device -> host transfer per stage at the end."""
    out = []
    gt = xyz_gt.detach().float()
    for i, st in enumerate(stage_type):
        xyz = decode_stage(results[i], st, img, center, M, cube, cam_para, kernel, img_size, flip).float()
        err = xyz2error(xyz, gt, center, cube)
        pa = xyz2error(similarity_align(xyz.double(), gt.double()).float(), gt, center, cube)
        packed = torch.cat((err.reshape(-1), err.mean().reshape(1), pa.mean().reshape(1))).cpu().numpy()
        out.append({"joint_errors": packed[:-2].reshape(tuple(err.shape)), "mean_error": float(packed[-2]), "pa_mean_error": float(packed[-1])})
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
