"""Generate the golden vectors under tests/golden/ from the *imported reference* (run in the build container only).

    python tests/golden/gen_golden.py

For each network family it (1) builds the reference KPFusion with the import shims of ref_import.py, (2) loads the
synthetic state dict of keypointfusion_amd.weights (strict=True, so the key contract is checked too), (3) runs the
reference's own forward on the synthetic batch, capturing intermediates with forward hooks, (4) asserts that
oracle/kpf_oracle.py reproduces every captured tensor (fp32 tolerance below; integer index tensors exactly), and only
then (5) writes `kpf_<net>_B2_S128.npz`.  Inputs and weights are NOT stored — they are regenerated from their seeds
(a sha256 of the weights is stored to detect generator drift).  Large tensors are stored strided (see SUB below).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from keypointfusion_amd.weights import synthetic_state_dict, synthetic_batch, state_dict_digest  # noqa: E402
from oracle import kpf_oracle as O  # noqa: E402

ATOL, RTOL = 2e-5, 1e-4


def sub_offset(a):  # B x 105 x F x F -> every 2nd pixel
    return a[:, :, ::2, ::2]


def sub_feat(a):  # B x 128 x F x F
    return a[:, ::8, ::4, ::4]


def close(name, a, b, atol=ATOL, rtol=RTOL):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if a.dtype in (torch.int64, torch.int32):
        assert torch.equal(a, b), "%s: integer mismatch (%d of %d)" % (name, (a != b).sum(), a.numel())
        return 0.0
    err = (a - b).abs()
    ok = err <= atol + rtol * b.abs()
    assert bool(ok.all()), "%s: max abs err %.3e (ref max %.3e)" % (name, err.max(), b.abs().max())
    return float(err.max())


def run(net, B=2, S=128):
    torch.set_num_threads(8)
    model, ld = ref_import.build_reference_model(net)
    sd_np = synthetic_state_dict(net, seed=0)
    sd = O.to_torch_sd(sd_np)
    model.load_state_dict(sd, strict=True)
    model.eval()
    batch = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, S, seed=1).items()}

    cap = {}

    def hook(name):
        def f(mod, inp, out):
            cap[name] = out
        return f

    hs = [model.backbone_d.register_forward_hook(hook("backbone_d")),
          model.backbone_rgb.register_forward_hook(hook("backbone_rgb"))]
    for i in (1, 2):
        blk = getattr(model, "block%d" % i)
        hs += [blk.FA.register_forward_hook(hook("b%d.FA" % i)),
               blk.init_TR.register_forward_hook(hook("b%d.init_TR" % i)),
               blk.crossTR.register_forward_hook(hook("b%d.crossTR" % i)),
               blk.final_TR.register_forward_hook(hook("b%d.final_TR" % i))]
        for gi, g in enumerate(blk.FA.groupers):
            orig = g.ball_query

            def bq(radius, nsample, xyz, new_xyz, _o=orig, _n="b%d.ball%d" % (i, gi)):
                r = _o(radius, nsample, xyz, new_xyz)
                cap[_n] = r
                return r
            g.ball_query = bq
    o_idx = ld.img2pcl_index

    def idx_hook(*a, **k):
        r = o_idx(*a, **k)
        cap["img2pcl"] = r
        return r
    ld.img2pcl_index = idx_hook
    o_uvd = ld.uvd_nl2xyznl_tensor
    first = {}

    def uvd_hook(*a, **k):
        r = o_uvd(*a, **k)
        first.setdefault("joint_xyz0", r)
        return r
    ld.uvd_nl2xyznl_tensor = uvd_hook

    with torch.no_grad():
        res, sws, _ = model(batch["img_rgb"], batch["img"], batch["pcl"], ld, batch["center"], batch["M"], batch["cube"],
                            batch["cam_para"], 0.8)
    for h in hs:
        h.remove()

    aux = {}
    ores, osws = O.kpfusion_forward(sd, batch["img_rgb"], batch["img"], batch["pcl"], batch["center"], batch["M"],
                                    batch["cube"], batch["cam_para"], 0.8, aux=aux)
    report = {}
    for i in range(6):
        report["result%d" % i] = close("result%d" % i, ores[i], res[i])
    for i in range(2):
        report["spatial%d" % i] = close("spatial%d" % i, osws[i], sws[i])
    report["img_feat"] = close("img_feat", aux["img_feat"], cap["backbone_d"][1])
    report["img_feat_rgb"] = close("img_feat_rgb", aux["img_feat_rgb"], cap["backbone_rgb"][1])
    report["joint_xyz0"] = close("joint_xyz0", aux["joint_xyz0"], first["joint_xyz0"])
    report["pcl_closeness"] = close("pcl_closeness", aux["pcl_closeness"], cap["img2pcl"][0])
    close("pcl_index", aux["pcl_index"], cap["img2pcl"][1])
    for i in (1, 2):
        a = aux["block%d" % i]
        report["b%d.FA" % i] = close("b%d.FA" % i, a["joint_feat_desa"], cap["b%d.FA" % i])
        report["b%d.h_init" % i] = close("b%d.h_init" % i, a["h_init"], cap["b%d.init_TR" % i][0])
        report["b%d.dec" % i] = close("b%d.dec" % i, a["dec"], cap["b%d.crossTR" % i].permute(0, 2, 1))
        for gi in range(3):
            close("b%d.ball%d" % (i, gi), a["ball_idx"][gi], cap["b%d.ball%d" % (i, gi)])
    print(net, "oracle == reference:", json.dumps({k: "%.2e" % v for k, v in report.items()}))

    # "argmax indices" (SURVEY D7): per-joint argmax of the masked weight logits of each stream, and of each spatial weight
    def argmax_logits(off, img):
        Fs = off.shape[-1]
        d = torch.nn.functional.interpolate(img, [Fs, Fs])
        w = off[:, 84:].masked_fill(d > 0.99, -1e8)
        return w.reshape(w.shape[0], 21, -1).argmax(-1)

    out = dict(
        weights_sha256=np.array(state_dict_digest(sd_np)),
        img_offset_sub=sub_offset(res[0]).numpy(), img_offset_rgb_sub=sub_offset(res[1]).numpy(),
        r3d1=res[2].numpy(), r2d1=res[3].numpy(), r3d2=res[4].numpy(), r2d2=res[5].numpy(),
        sw1=sws[0].numpy(), sw2=sws[1].numpy(),
        img_feat_sub=sub_feat(cap["backbone_d"][1]).numpy(), img_feat_rgb_sub=sub_feat(cap["backbone_rgb"][1]).numpy(),
        joint_xyz0=first["joint_xyz0"].numpy(),
        pcl_closeness=cap["img2pcl"][0].numpy(), pcl_index=cap["img2pcl"][1].numpy().astype(np.int32),
        argmax_w_d=argmax_logits(res[0], batch["img"]).numpy().astype(np.int32),
        argmax_w_rgb=argmax_logits(res[1], batch["img"]).numpy().astype(np.int32),
        argmax_sw1=sws[0].reshape(B, 21, -1).argmax(-1).numpy().astype(np.int32),
        argmax_sw2=sws[1].reshape(B, 21, -1).argmax(-1).numpy().astype(np.int32),
    )
    for i in (1, 2):
        out["b%d_FA" % i] = cap["b%d.FA" % i].numpy()
        out["b%d_h_init" % i] = cap["b%d.init_TR" % i][0].numpy()
        out["b%d_dec" % i] = cap["b%d.crossTR" % i].permute(0, 2, 1).contiguous().numpy()
        for gi in range(3):
            out["b%d_ball%d" % (i, gi)] = cap["b%d.ball%d" % (i, gi)].numpy().astype(np.int16)
    path = os.path.join(HERE, "kpf_%s_B%d_S%d.npz" % (net.split("KPFusion-")[1], B, S))
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def run_backbone(net, B=1, S=64):
    """Backbones only at a second input size (fully convolutional part; BASELINE.json configs[1] uses S=256)."""
    model, ld = ref_import.build_reference_model(net)
    sd_np = synthetic_state_dict(net, seed=0)
    sd = O.to_torch_sd(sd_np)
    model.load_state_dict(sd, strict=True)
    batch = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, S, seed=1).items()}
    with torch.no_grad():
        od, fd = model.backbone_d(batch["img"])
        orgb, frgb = model.backbone_rgb(batch["img_rgb"])
    ood, ofd, oorgb, ofrgb = O.backbones_forward(sd, batch["img_rgb"], batch["img"])
    e = [close("od", ood, od), close("fd", ofd, fd), close("orgb", oorgb, orgb), close("frgb", ofrgb, frgb)]
    print(net, "backbones S=%d oracle == reference" % S, ["%.2e" % x for x in e])
    path = os.path.join(HERE, "backbone_%s_B%d_S%d.npz" % (net.split("KPFusion-")[1], B, S))
    np.savez_compressed(path, img_offset=od.numpy(), img_offset_rgb=orgb.numpy(), img_feat_sub=fd[:, ::4].numpy(),
                        img_feat_rgb_sub=frgb[:, ::4].numpy())
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def run_metrics():
    """Metric known-answers from the reference's own numpy code (util/generateFeature.GFM.rigid_align, util/eval_utils)."""
    ref_import.install_shims()
    if ref_import.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REF_ROOT)
    from util.generateFeature import GFM
    from util import eval_utils
    rng = np.random.default_rng(5)
    A = rng.normal(size=(4, 21, 3))
    Bm = A * 1.3 + rng.normal(size=(4, 21, 3)) * 0.05 + np.array([0.2, -0.1, 0.4])
    Bm[1] = Bm[1] * np.array([1, 1, -1])  # forces the reflection branch (det < 0)
    g = GFM()
    aligned = np.stack([g.rigid_align(A[i], Bm[i]) for i in range(4)])
    errs = [list(rng.gamma(2.0, 6.0, size=200)) for _ in range(21)]
    auc, curve, th = eval_utils.get_measures(errs, 0, 50, 20)
    sub = eval_utils.calc_auc(th[8:] * 1000.0, curve[8:])
    path = os.path.join(HERE, "metrics.npz")
    np.savez_compressed(path, A=A, B=Bm, aligned=aligned, errs=np.array(errs), auc=auc, curve=curve, th=th, sub=sub)
    print("wrote", path)


def dump_keys(net):
    """state_keys_<net>.json: [key, shape, dtype] of the reference module's state_dict, in order (the state-dict contract)."""
    model, _ = ref_import.build_reference_model(net)
    rows = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()]
    with open(os.path.join(HERE, "state_keys_%s.json" % net.split("KPFusion-")[1]), "w") as f:
        json.dump(rows, f)


if __name__ == "__main__":
    if not ref_import.reference_available():
        sys.exit("reference tree not found; golden vectors can only be generated in the build container")
    only = sys.argv[1:]
    for net in ("KPFusion-convnext-tiny", "KPFusion-resnet-18", "KPFusion-resnet-50"):
        if only and net not in only:
            continue
        run(net)
        run_backbone(net)
    # round 4: the remaining families the model constructor accepts (convNeXT/resnetUnet.py:40-45 depths [3,3,27,3] for base — the
    # network of BASELINE configs[4] —, model/resnetUnet.py for resnet-101): backbones at S=64, and the full model for convnext-base
    for net in ("KPFusion-convnext-base", "KPFusion-convnext-small", "KPFusion-convnext-large", "KPFusion-resnet-101"):
        if only and net not in only:
            continue
        run_backbone(net)
    if not only or "KPFusion-convnext-base" in only:
        run("KPFusion-convnext-base")
    for net in ("KPFusion-convnext-tiny", "KPFusion-convnext-base", "KPFusion-resnet-18", "KPFusion-resnet-50", "KPFusion-resnet-101"):
        if only and net not in only:
            continue
        dump_keys(net)
    if not only:
        run_metrics()
