#!/usr/bin/env python3
"""Generates tests/golden/train_loss.npz from the IMPORTED reference (/root/reference, this container only): the loss codec, the
custom SmoothL1Loss and the loss schedule of one training iteration (train.py:211-261) on seeded inputs.  Run from the repo root:
    python tests/golden/gen_golden_train.py
Nothing of the reference is stored: only inputs and the numbers its own functions return."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install_shims()
sys.path.insert(0, ref_import.REF_ROOT)
cwd = os.getcwd()
os.chdir(ref_import.REF_ROOT)
from util.generateFeature import GFM  # noqa: E402
from model.loss import SmoothL1Loss  # noqa: E402
os.chdir(cwd)

from keypointfusion_amd.weights import synthetic_batch  # noqa: E402

g = torch.Generator().manual_seed(21)
B, J, Fs = 3, 21, 32
b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, 128, seed=6).items()}
img = b["img"]
uvd_gt = torch.rand(B, J, 3, generator=g) * 1.4 - 0.7
xyz_gt = torch.rand(B, J, 3, generator=g) * 1.4 - 0.7
results = [torch.randn(B, 5 * J, Fs, Fs, generator=g) * 0.5, torch.randn(B, 5 * J, Fs, Fs, generator=g) * 0.5]
results += [xyz_gt + torch.randn(B, J, 3, generator=g) * s for s in (0.05, 0.02, 0.004, 0.02)]
sws = [torch.rand(B, J, Fs, Fs, generator=g), torch.rand(B, J, Fs, Fs, generator=g)]
for t in results + sws:
    t.requires_grad_(True)

gfm = GFM()
L1 = SmoothL1Loss()
# the inputs are NOT stored (random maps do not compress): tests/test_training.py re-draws them from the same seeds and checks this
# checksum first
out = {"uvd_gt": uvd_gt.numpy(), "xyz_gt": xyz_gt.numpy(),
       "input_checksum": np.float64(sum(float(t.detach().double().abs().sum()) for t in results + sws) + float(img.double().abs().sum()))}
pixel_gt = gfm.joint2feature(uvd_gt, img, [0.8], Fs, ["weight_offset"])
out["pixel_gt"] = pixel_gt.numpy()
out["decode0"] = gfm.feature2joint(img, results[0], ["weight_offset"], [0.8]).detach().numpy()
out["hm_sigma3"] = gfm.joint2heatmap(uvd_gt[:, :, :2], 0.8, Fs, sigma=3).numpy()[:, ::5]
out["hm_sigma2"] = gfm.joint2heatmap(uvd_gt[:, :, :2], 0.8, Fs, sigma=2).numpy()[:, ::5]
# the schedule of train.py:211-261 (epoch 0), written with the reference's own functions
loss = 0
parts = {}
for index, st in enumerate([1, 1, 2, 3, 2, 3]):
    if st == 1:
        pd = results[index]
        ju = gfm.feature2joint(img, pd, ["weight_offset"], [0.8])
        lp = L1(pd[:, :pixel_gt.size(1)], pixel_gt) * 1
        lc = L1(ju, uvd_gt) * 100
        loss = loss + (lp + lc)
        parts["loss_pixel_%d" % index], parts["loss_coord_%d" % index] = float(lp), float(lc)
    else:
        lc = L1(results[index], xyz_gt) * 100
        loss = loss + lc
        parts["loss_coord_%d" % index] = float(lc)
for index, sw in enumerate(sws):
    hm = gfm.joint2heatmap(uvd_gt[:, :, :2], 0.8, Fs, sigma=3 if index == 0 else 2)
    ls = L1(sw, hm / hm.max()) * 10
    loss = loss + ls
    parts["loss_spatial_%d" % index] = float(ls)
loss.backward()
out["loss"] = np.float64(float(loss))
for k, v in parts.items():
    out[k] = np.float64(v)
for name, t in [("result%d" % i, t) for i, t in enumerate(results)] + [("sw%d" % i, t) for i, t in enumerate(sws)]:
    out["gradnorm_" + name] = np.float64(float(t.grad.double().norm()))   # first-step gradients: norm + a strided sample
    out["gradsample_" + name] = t.grad.reshape(-1)[::97].numpy().copy()
# SmoothL1Loss alone around its 0.01 knee
z = torch.tensor([[0.0, 0.005, -0.0099, 0.01, -0.01, 0.0101, 0.5, -2.0]])
out["sl1_x"] = z.numpy()
out["sl1_mean"] = np.float64(float(L1(z, torch.zeros_like(z))))
out["sl1_sum"] = np.float64(float(SmoothL1Loss(size_average=False)(z, torch.zeros_like(z))))
np.savez_compressed(os.path.join(HERE, "train_loss.npz"), **out)
print("wrote train_loss.npz: loss %.6f" % float(loss), parts)
