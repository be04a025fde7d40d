#!/usr/bin/env python3
"""Generates tests/golden/train_step_<net>.npz from the IMPORTED reference (/root/reference, this container only): one training
iteration of train.py:209-265 on the synthetic batch — the model in .train() mode (batch-statistics BatchNorm), every dropout
probability set to 0 (the reference's random stream cannot be reproduced), the loss of train.py:211-261 at epoch 0, loss.backward()
— and stores the loss, its parts, the gradient norm of every parameter that received one, and a few updated BatchNorm running
statistics.  Inputs and weights are regenerated from their seeds.  Run from the repo root:
    python tests/golden/gen_golden_trainstep.py [net ...]"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from keypointfusion_amd.weights import synthetic_batch, synthetic_state_dict  # noqa: E402


def run(net, B=3):
    torch.set_num_threads(8)
    model, ld = ref_import.build_reference_model(net)
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(net, seed=0).items()}
    model.load_state_dict(sd, strict=True)
    model.train()
    for m in model.modules():  # no dropout: nn.Dropout modules and the functional dropout_p of the decoder's MultiheadAttention
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0
    sys.path.insert(0, ref_import.REF_ROOT)
    from util.generateFeature import GFM
    from model.loss import SmoothL1Loss
    gfm, L1 = GFM(), SmoothL1Loss()
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, 128, seed=11).items()}
    g = torch.Generator().manual_seed(33)
    uvd_gt = torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6
    xyz_gt = torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6
    balls = []  # the 6 ball-query index tensors (2 blocks x 3 radii), in call order: integer decisions taken around network outputs
    orig_bq = ref_import._QueryAndGroup.ball_query

    def rec_bq(radius, nsample, xyz, new_xyz):
        idx = orig_bq(radius, nsample, xyz, new_xyz)
        balls.append(idx.clone())
        return idx

    ref_import._QueryAndGroup.ball_query = staticmethod(rec_bq)
    results, sws, _ = model(b["img_rgb"], b["img"], b["pcl"], ld, b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
    ref_import._QueryAndGroup.ball_query = staticmethod(orig_bq)
    assert len(balls) == 6
    img = b["img"]
    loss = 0
    parts = {}
    fs = None
    for index, st in enumerate([1, 1, 2, 3, 2, 3]):  # train.py:211-246
        if st == 1:
            pd = results[index]
            fs = pd.size(-1)
            pixel_gt = gfm.joint2feature(uvd_gt, img, [0.8], fs, ["weight_offset"])
            ju = gfm.feature2joint(img, pd, ["weight_offset"], [0.8])
            lp = L1(pd[:, :pixel_gt.size(1)], pixel_gt) * 1
            lc = L1(ju, uvd_gt) * 100
            loss = loss + (lp + lc)
            parts["loss_pixel_%d" % index], parts["loss_coord_%d" % index] = float(lp.detach()), float(lc.detach())
        else:
            lc = L1(results[index], xyz_gt) * 100
            loss = loss + lc
            parts["loss_coord_%d" % index] = float(lc.detach())
    for index, sw in enumerate(sws):  # train.py:249-259
        hm = gfm.joint2heatmap(uvd_gt[:, :, :2], 0.8, fs, sigma=3 if index == 0 else 2)
        ls = L1(sw, hm / hm.max()) * 10
        loss = loss + ls
        parts["loss_spatial_%d" % index] = float(ls.detach())
    loss.backward()
    names, norms = [], []
    for n, p in model.named_parameters():
        if p.grad is not None:
            names.append(n)
            norms.append(float(p.grad.double().norm()))
    out = {"loss": np.float64(float(loss.detach())), "grad_names": np.array(names), "grad_norms": np.array(norms, dtype=np.float64),
           "uvd_gt": uvd_gt.numpy(), "xyz_gt": xyz_gt.numpy(), "ball_idx": torch.stack(balls).numpy().astype(np.int16), "r2d2": results[5].detach().numpy(), "r3d1": results[2].detach().numpy()}
    for k, v in parts.items():
        out[k] = np.float64(v)
    msd = model.state_dict()
    for k in ("backbone_d.up4.0.bn1.running_mean", "backbone_rgb.fusion_layer2.bn3.running_var", "block1.FA.bn_f0_blocks.1.running_mean",
              "block2.pcl_feat_emb.1.running_var"):
        out["bn::" + k] = msd[k].numpy()
    np.savez_compressed(os.path.join(HERE, "train_step_%s.npz" % net.replace("KPFusion-", "")), **out)
    print(net, "loss %.6f" % float(loss), "params with grad:", len(names), "of", sum(1 for _ in model.parameters()))


if __name__ == "__main__":
    for net in (sys.argv[1:] or ["KPFusion-convnext-tiny", "KPFusion-resnet-18"]):
        run(net)
