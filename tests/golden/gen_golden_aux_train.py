"""Train-mode golden vectors for the stand-alone heads, from the *imported reference* in the build container:

    python tests/golden/gen_golden_aux_train.py   ->   tests/golden/aux_train.npz

For CBAM, PoseNet and the MANO head: the reference module in .train() mode on our seeded synthetic weights and inputs — outputs (strided), a scalar loss
= sum of outputs x fixed seeded projections, the gradient NORM of every parameter after loss.backward(), the gradient of the input, and the BatchNorm running
statistics after the step.  The product's train-mode forwards (keypointfusion_amd/heads_train.py) are compared with these in tests/test_heads_gpu.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from keypointfusion_amd import spec as S  # noqa: E402
from keypointfusion_amd.weights import synthetic_from_spec, synthetic_mano_model, synthetic_mano_head_state, synthetic_tensor  # noqa: E402
from oracle.kpf_oracle import to_torch_sd  # noqa: E402

CBAM_CASES = ((128, False, 3, 16, 16), (64, True, 2, 8, 12))  # (C, no_spatial, B, H, W)
POSENET_CASE = (1, 128, 4, 128)                               # (nstack, inp_dim, B, S)
MANO_B = 4


def proj(shape, tag):
    return torch.from_numpy(synthetic_tensor(tuple(shape), 9, "proj_" + tag, -1.0, 1.0)).double()


def step(m, outs, x, tag, out):
    loss = sum((o.double() * proj(o.shape, "%s_%d" % (tag, i))).sum() for i, o in enumerate(outs))
    loss.backward()
    out[tag + "_loss"] = np.float64(loss.item())
    out[tag + "_dx"] = x.grad.float().numpy()
    for n, p in m.named_parameters():
        if p.grad is not None:
            out[tag + "_gnorm::" + n] = np.float64(p.grad.double().norm().item())
    for n, b in m.named_buffers():
        if n.endswith(("running_mean", "running_var")):
            out[tag + "_bn::" + n] = b.detach().float().numpy().copy()


def main():
    if not ref_import.reference_available():
        sys.exit("reference tree not found; golden vectors can only be generated in the build container")
    torch.set_num_threads(8)
    cbam, hourglass = ref_import.load_reference_aux()
    out = {}
    for C, nosp, B, H, W in CBAM_CASES:
        tag = "cbam_C%d_%d" % (C, int(nosp))
        m = cbam.CBAM(C, no_spatial=nosp)
        m.load_state_dict(to_torch_sd(synthetic_from_spec(S.cbam_spec(C, no_spatial=nosp), 0, prefix=tag + ".")), strict=True)
        m.train()
        x = torch.from_numpy(synthetic_tensor((B, C, H, W), 3, tag + "_train")).requires_grad_(True)
        r = m(x)
        outs = [r] if nosp else list(r)
        for i, o in enumerate(outs):
            out["%s_out%d" % (tag, i)] = o.detach()[:, ::4].numpy()
        step(m, outs, x, tag, out)
        print(tag, "loss", out[tag + "_loss"])
    nstack, dim, B, Sz = POSENET_CASE
    tag = "posenet_n%d_d%d" % (nstack, dim)
    # The deepest hourglass level normalises over B x 4 x 4 samples per channel with batch statistics, which amplifies rounding both ways: the reference's own
    # fp32 step is percent-level away from its fp64 step at small B.  The fixture therefore holds the float64 step (the function the reference defines) and,
    # beside it, how far the reference's fp32 step is from that (ref32_*): the product's fp32 step is held to a small multiple of the reference's own distance.
    res = {}
    for dt in (torch.float64, torch.float32):
        m = hourglass.PoseNet(nstack, 21, dim)
        m.load_state_dict(to_torch_sd(synthetic_from_spec(S.posenet_spec(nstack, 21, dim), 0, prefix=tag + ".")), strict=True)
        m = m.to(dt).train()
        x = torch.from_numpy(synthetic_tensor((B, 1, Sz, Sz), 4, tag + "_train", -1.0, 1.0)).to(dt).requires_grad_(True)
        preds, feat = m(x)
        o = {}
        o[tag + "_preds_sub"], o[tag + "_feat_sub"] = preds.detach()[:, :, ::4, ::4].float().numpy(), feat.detach()[:, ::8, ::2, ::2].float().numpy()
        step(m, [preds, feat], x, tag, o)
        res[dt] = o
    r64, r32 = res[torch.float64], res[torch.float32]
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / np.abs(np.asarray(b, np.float64)).max())
    gmax = max(float(v) for k, v in r64.items() if "_gnorm::" in k)
    out.update({k: (np.asarray(v, np.float32) if np.asarray(v).ndim else v) for k, v in r64.items()})
    out[tag + "_ref32_out_rel"] = np.float64(max(rel(r32[tag + "_preds_sub"], r64[tag + "_preds_sub"]), rel(r32[tag + "_feat_sub"], r64[tag + "_feat_sub"])))
    out[tag + "_ref32_dx_rel"] = np.float64(rel(r32[tag + "_dx"], r64[tag + "_dx"]))
    out[tag + "_ref32_gnorm_rel"] = np.float64(max(abs(float(r32[k]) - float(v)) / float(v) for k, v in r64.items() if "_gnorm::" in k and float(v) > 1e-5 * gmax))
    out[tag + "_ref32_bn_rel"] = np.float64(max(rel(r32[k], v) for k, v in r64.items() if "_bn::" in k))
    print(tag, "loss", out[tag + "_loss"], "reference fp32 vs fp64: out %.2e dx %.2e gnorm %.2e bn %.2e" % tuple(float(out[tag + k]) for k in ("_ref32_out_rel", "_ref32_dx_rel", "_ref32_gnorm_rel", "_ref32_bn_rel")))
    mh = ref_import.load_reference_mano_head(synthetic_mano_model(0))
    head = mh.mano_regHead()
    head.load_state_dict(to_torch_sd(synthetic_mano_head_state(0)), strict=True)
    head.train()
    feats = torch.from_numpy(synthetic_tensor((MANO_B, 1024), 5, "mano_features_train")).requires_grad_(True)
    r = head(feats)
    keys = ("verts3d", "joints3d", "mano_shape", "mano_pose", "mano_pose_aa")
    for k in keys:
        out["mano_" + k] = r[k].detach().numpy()
    step(head, [r[k] for k in keys], feats, "mano", out)
    print("mano loss", out["mano_loss"])
    path = os.path.join(HERE, "aux_train.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
