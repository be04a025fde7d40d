"""Golden vectors for the stand-alone heads (SURVEY.md §8 a17 CBAM, a18 PoseNet/Hourglass, a19 MANO head), generated from the
*imported reference* in the build container:

    python tests/golden/gen_golden_aux.py

Same procedure as gen_golden.py: build the reference module, load our seeded synthetic weights with strict=True (checks the key
contract), run the reference's forward, assert oracle/aux_oracle.py reproduces it, then write tests/golden/aux_*.npz.  Inputs and
weights are regenerated from seeds, not stored; large outputs are stored strided (every 4th channel / every 2nd pixel).  The MANO head runs over a synthetic hand model (see ref_import.py).
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
from gen_golden import close  # noqa: E402
from keypointfusion_amd import spec as S  # noqa: E402
from keypointfusion_amd.weights import synthetic_from_spec, synthetic_mano_model, synthetic_mano_head_state, synthetic_tensor  # noqa: E402
from oracle import aux_oracle as A  # noqa: E402
from oracle.kpf_oracle import to_torch_sd  # noqa: E402

CBAM_CASES = ((128, False, 2, 16, 16), (64, True, 2, 8, 12), (256, False, 1, 8, 8))  # (C, no_spatial, B, H, W)
POSENET_CASES = ((2, 128, 1, 128), (1, 256, 2, 64))  # (nstack, inp_dim, B, S)
MANO_B = 4


def run_cbam():
    cbam, _ = ref_import.load_reference_aux()
    out = {}
    for C, nosp, B, H, W in CBAM_CASES:
        tag = "cbam_C%d_%d" % (C, int(nosp))
        sd = to_torch_sd(synthetic_from_spec(S.cbam_spec(C, no_spatial=nosp), 0, prefix=tag + "."))
        m = cbam.CBAM(C, no_spatial=nosp)
        m.load_state_dict(sd, strict=True)
        m.eval()
        x = torch.from_numpy(synthetic_tensor((B, C, H, W), 3, tag))
        with torch.no_grad():
            ref = m(x)
            scale = torch.sigmoid(m.ChannelGate.mlp(x.mean((2, 3))) + m.ChannelGate.mlp(x.amax((2, 3))))
        mine = A.cbam_forward(sd, x, nosp)
        close(tag + ".scale", A.cbam_channel_scale(sd, x), scale)
        out[tag + "_scale"] = scale.numpy()
        if nosp:
            close(tag, mine, ref)
            out[tag + "_out"] = ref[:, ::4].numpy()
        else:
            close(tag + ".0", mine[0], ref[0])
            close(tag + ".1", mine[1], ref[1])
            out[tag + "_out0"] = ref[0][:, ::4].numpy()
            out[tag + "_out1"] = ref[1][:, ::4].numpy()
        print(tag, "oracle == reference")
    np.savez_compressed(os.path.join(HERE, "aux_cbam.npz"), **out)


def run_posenet():
    _, hourglass = ref_import.load_reference_aux()
    out = {}
    for nstack, dim, B, Sz in POSENET_CASES:
        tag = "posenet_n%d_d%d" % (nstack, dim)
        sd = to_torch_sd(synthetic_from_spec(S.posenet_spec(nstack, 21, dim), 0, prefix=tag + "."))
        m = hourglass.PoseNet(nstack, 21, dim)
        m.load_state_dict(sd, strict=True)
        m.eval()
        x = torch.from_numpy(synthetic_tensor((B, 1, Sz, Sz), 4, tag, -1.0, 1.0))
        with torch.no_grad():
            preds, feat = m(x)
        op, of = A.posenet_forward(sd, x, nstack)
        e = (close(tag + ".preds", op, preds, atol=5e-5), close(tag + ".feat", of, feat, atol=5e-5))
        print(tag, "oracle == reference", ["%.2e" % v for v in e], "out absmax %.2f" % float(preds.abs().max()))
        out[tag + "_preds_sub"] = preds[:, :, ::2, ::2].numpy()
        out[tag + "_feat_sub"] = feat[:, ::4].numpy()
    np.savez_compressed(os.path.join(HERE, "aux_posenet.npz"), **out)


def run_mano():
    model = synthetic_mano_model(0)
    mh = ref_import.load_reference_mano_head(model)
    head = mh.mano_regHead()
    sd = to_torch_sd(synthetic_mano_head_state(0))
    head.load_state_dict(sd, strict=True)
    head.eval()
    feats = torch.from_numpy(synthetic_tensor((MANO_B, 1024), 5, "mano_features"))
    with torch.no_grad():
        ref = head(feats)
    mine = A.mano_head_forward(sd, feats)
    out = {}
    for k in ("verts3d", "joints3d", "mano_shape", "mano_pose", "mano_pose_aa"):
        e = close("mano." + k, mine[k], ref[k], atol=2e-3 if k in ("verts3d", "joints3d") else 2e-5)
        print("mano", k, "max err %.2e (absmax %.2f)" % (e, float(ref[k].abs().max())))
        out[k] = ref[k].numpy()
    np.savez_compressed(os.path.join(HERE, "aux_mano.npz"), **out)


def dump_keys():
    """Key/shape lists of the reference modules (state-dict contract of the drop-in heads)."""
    import json
    cbam, hourglass = ref_import.load_reference_aux()
    mh = ref_import.load_reference_mano_head(synthetic_mano_model(0))
    mods = {"cbam_128": cbam.CBAM(128), "cbam_64_nospatial": cbam.CBAM(64, no_spatial=True), "posenet_2_21_128": hourglass.PoseNet(2, 21, 128),
            "posenet_1_21_256": hourglass.PoseNet(1, 21, 256), "mano_regHead": mh.mano_regHead()}
    out = {k: [[n, list(v.shape), str(v.dtype).replace("torch.", "")] for n, v in m.state_dict().items()] for k, m in mods.items()}
    with open(os.path.join(HERE, "state_keys_aux.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    if not ref_import.reference_available():
        sys.exit("reference tree not found; golden vectors can only be generated in the build container")
    torch.set_num_threads(8)
    run_cbam()
    run_posenet()
    run_mano()
    dump_keys()
