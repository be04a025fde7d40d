"""GPU parity tests (run with -m gpu on the MI355X box): every C-ABI kernel against a plain fp32 torch-CPU
restatement of the same op, and the backbone / full model against the oracle (oracle/kpf_oracle.py), which is itself
pinned to the imported reference by tests/test_oracle_golden.py.  Nothing here reads /root/reference.

Tolerances: per-op 1e-4 relative to the tensor's max magnitude (fp32 GEMMs with K up to 3072 in a different summation
order); end-to-end 1e-3 relative as stated by BASELINE.json north_star, and |d joint| * cube/2 <= 0.05 mm.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, synthetic_sd
from keypointfusion_amd.weights import synthetic_batch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from keypointfusion_amd import lib
    lib.load()  # fail loudly if the HIP library is not built
    return torch.device("cuda:0")


def rel_err(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def nhwc(t):  # NCHW cpu tensor -> Act on device
    from keypointfusion_amd.engine import Act
    B, C, H, W = t.shape
    return Act(t.permute(0, 2, 3, 1).contiguous().view(-1).to(_dev()), B, H, W, C)


def to_nchw(act):
    return act.buf.view(act.B, act.H, act.W, act.ld)[..., act.coff:act.coff + act.C].permute(0, 3, 1, 2).cpu()


CONV_CASES = [
    # B, Cin, H, W, N, k, stride, pad
    (2, 96, 16, 16, 384, 1, 1, 0),
    (2, 384, 16, 16, 96, 1, 1, 0),
    (1, 48, 32, 32, 48, 3, 1, 1),
    (2, 64, 9, 7, 64, 3, 1, 1),      # ragged M (126 pixels)
    (1, 128, 32, 32, 105, 1, 1, 0),  # N tail (105 = 4*26+1)
    (2, 64, 16, 16, 128, 3, 2, 1),   # strided 3x3
    (2, 64, 16, 16, 128, 1, 2, 0),   # strided 1x1 (ResNet downsample)
    (1, 4, 32, 32, 64, 7, 2, 3),     # ResNet stem, padded input channels, K=196 -> Kp=224
    (3, 768, 4, 4, 3072, 1, 1, 0),   # M=48: small-tile config
    (1, 1152, 8, 8, 192, 1, 1, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_torch(case):
    from keypointfusion_amd import engine as E
    B, Cin, H, W, N, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(N, generator=g)
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    pc = E.PackedConv(w, b, _dev(), stride=s, pad=p)
    out = E.conv(pc, nhwc(x))
    assert rel_err(to_nchw(out), ref) < 1e-5


def test_conv2d_prologue_epilogues_and_slices():
    """BN+ReLU operand prologue, folded BN + ReLU epilogue, residual (+gamma), GELU, channel-slice in/out, NCHW out."""
    from keypointfusion_amd import engine as E, lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    B, H, W = 2, 12, 12
    xcat = torch.randn(B, 96 + 32, H, W, generator=g)
    x = xcat[:, 32:]  # consume channels [32,128) of a wider buffer
    w = torch.randn(64, 96, 1, 1, generator=g) / 96 ** 0.5
    b = torch.randn(64, generator=g)
    s1, t1 = torch.rand(96, generator=g) + 0.5, torch.randn(96, generator=g)
    s2, t2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    ref = F.relu(F.conv2d(F.relu(x * s1[None, :, None, None] + t1[None, :, None, None]), w, b) * s2[None, :, None, None] + t2[None, :, None, None])
    pc = E.PackedConv(w, b, dev, fold_bn=(s2.double(), t2.double()), prologue=(s1.double(), t1.double()))
    xin = nhwc(xcat).slice(32, 96)
    big = E.Act.empty(B, H, W, 64 + 64, dev)
    big.buf.zero_()
    out = E.conv(pc, xin, out=big.slice(64, 64), flags=L.KPF_ACT_RELU)
    assert rel_err(to_nchw(out), ref) < 1e-5
    assert float(big.buf.view(B, H, W, 128)[..., :64].abs().max()) == 0.0  # neighbouring slice untouched

    # gamma * gelu-less linear + residual, in place (ConvNeXt pw2) and GELU (pw1)
    xr = torch.randn(B, 96, H, W, generator=g)
    hmid = torch.randn(B, 384, H, W, generator=g)
    w2 = torch.randn(96, 384, generator=g) / 384 ** 0.5
    b2 = torch.randn(96, generator=g)
    gam = torch.rand(96, generator=g)
    ref2 = xr + gam[None, :, None, None] * F.conv2d(hmid, w2[:, :, None, None], b2)
    pc2 = E.PackedConv(w2, b2, dev)
    xa = nhwc(xr)
    gamd = gam.to(dev)
    E.conv(pc2, nhwc(hmid), out=xa, gamma=gamd, res=xa)
    assert rel_err(to_nchw(xa), ref2) < 1e-5
    w1 = torch.randn(384, 96, generator=g) / 96 ** 0.5
    b1 = torch.randn(384, generator=g)
    ref3 = F.gelu(F.conv2d(xr, w1[:, :, None, None], b1))
    o3 = E.conv(E.PackedConv(w1, b1, dev), nhwc(xr), flags=L.KPF_ACT_GELU)
    assert rel_err(to_nchw(o3), ref3) < 1e-5

    # residual then ReLU (BasicBlock tail) and NCHW store with a ragged channel count
    w3 = torch.randn(105, 96, 1, 1, generator=g) / 96 ** 0.5
    b3 = torch.randn(105, generator=g)
    ref4 = F.conv2d(xr, w3, b3)
    o4 = torch.empty(B, 105, H, W, device=dev)
    E.conv(E.PackedConv(w3, b3, dev), nhwc(xr), out_nchw=o4)
    assert rel_err(o4, ref4) < 1e-5
    w4 = torch.randn(96, 96, 3, 3, generator=g) / (96 * 9) ** 0.5
    ref5 = F.relu(F.conv2d(xr, w4, None, padding=1) + xr)
    o5 = E.conv(E.PackedConv(w4, None, dev, pad=1), nhwc(xr), res=nhwc(xr), flags=L.KPF_RELU_AFTER_RES)
    assert rel_err(to_nchw(o5), ref5) < 1e-5


@pytest.mark.parametrize("cin,k", [(1, 4), (3, 4), (96, 2), (192, 2)])
def test_patchify_conv(cin, k):
    from keypointfusion_amd import engine as E
    g = torch.Generator().manual_seed(cin * 10 + k)
    x = torch.randn(2, cin, 32, 32, generator=g)
    w = torch.randn(96, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(96, generator=g)
    ref = F.conv2d(x, w, b, stride=k)
    out = E.conv(E.PackedConv(w, b, _dev(), stride=k, patchify=True), nhwc(x))
    assert rel_err(to_nchw(out), ref) < 1e-5


@pytest.mark.parametrize("C,H,W", [(96, 16, 16), (192, 8, 8), (384, 5, 7), (768, 4, 4), (128, 32, 32), (1024, 2, 2), (384, 16, 16), (768, 8, 8),
                                   (512, 32, 32), (1024, 16, 16), (320, 9, 13), (1024, 3, 3)])
def test_dwconv7_ln(C, H, W):
    import ctypes
    from keypointfusion_amd import engine as E, lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(2, C, H, W, generator=g)
    wd = torch.randn(C, 1, 7, 7, generator=g) / 7
    bd = torch.randn(C, generator=g)
    lw, lb = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ref = F.layer_norm(F.conv2d(x, wd, bd, padding=3, groups=C).permute(0, 2, 3, 1), (C,), lw, lb, 1e-6).permute(0, 3, 1, 2)
    xa = nhwc(x)
    ya = E.Act.empty(2, H, W, C, dev)
    # keep the device tensors alive across the launch (raw pointers cross the C ABI)
    wdd, bdd, lwd, lbd = wd.reshape(C, 49).t().contiguous().to(dev), bd.to(dev), lw.to(dev), lb.to(dev)
    L.check(L.load().kpf_dwconv7_ln_f32(E._ptr(xa.buf), E._ptr(wdd), E._ptr(bdd), E._ptr(lwd), E._ptr(lbd), E._ptr(ya.buf),
                                        2, H, W, C, 1e-6, E._stream()))
    torch.cuda.synchronize()
    assert rel_err(to_nchw(ya), ref) < 2e-5


@pytest.mark.parametrize("C,H,W", [(384, 8, 8), (768, 4, 4), (512, 9, 11)])
def test_dwconv7_ln_wide_forms_agree_bitwise(C, H, W):
    """The wide depthwise + LayerNorm kernel takes one-row strips while two-row strips would leave CUs without a workgroup (small batches) and two-row strips
    otherwise: a sample's result must not depend on the batch it arrives in — image 0 of a 2-image launch (one-row form) and of a 72-image launch (two-row
    form) are the same bits, in fp32 and in 16-bit storage."""
    import ctypes
    from keypointfusion_amd import engine as E, lib as L
    dev = _dev()
    lib = L.load()
    g = torch.Generator().manual_seed(C + H)
    xb = torch.randn(72, H, W, C, generator=g)
    wdd = (torch.randn(49, C, generator=g) / 7).to(dev)
    bdd, lwd, lbd = torch.randn(C, generator=g).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    for tdt, kdt in ((torch.float32, None), (torch.bfloat16, L.KPF_DT_BF16), (torch.float16, L.KPF_DT_F16)):
        outs = []
        for B in (2, 72):
            x = xb[:B].to(tdt).contiguous().to(dev)
            y = torch.full_like(x, float("nan"))
            if kdt is None:
                L.check(lib.kpf_dwconv7_ln_f32(E._ptr(x), E._ptr(wdd), E._ptr(bdd), E._ptr(lwd), E._ptr(lbd), E._ptr(y), B, H, W, C, 1e-6, E._stream()))
            else:
                L.check(lib.kpf_dwconv7_ln_h16(E._ptr(x), E._ptr(wdd), E._ptr(bdd), E._ptr(lwd), E._ptr(lbd), E._ptr(y), B, H, W, C, 1e-6, kdt, E._stream()))
            outs.append(y[:2].float().cpu())
        assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1]), (C, tdt)
        ref = F.layer_norm(F.conv2d(xb[:2].to(tdt).float().permute(0, 3, 1, 2), wdd.cpu().t().reshape(C, 1, 7, 7), bdd.cpu(), padding=3, groups=C).permute(0, 2, 3, 1),
                           (C,), lwd.cpu(), lbd.cpu(), 1e-6)
        assert rel_err(outs[0], ref) < (2e-5 if kdt is None else (1.2e-2 if tdt == torch.bfloat16 else 2e-3))


def test_layernorm_upsample_maxpool_repack():
    from keypointfusion_amd import engine as E
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 192, 9, 5, generator=g) * 3 + 1
    lw, lb = torch.rand(192, generator=g) + 0.5, torch.randn(192, generator=g)
    ref = F.layer_norm(x.permute(0, 2, 3, 1), (192,), lw, lb, 1e-6).permute(0, 3, 1, 2)
    xa = nhwc(x)
    lwd, lbd = lw.to(dev), lb.to(dev)
    out = E.layernorm(xa, lwd, lbd, 1e-6, out=E.Act.empty(2, 9, 5, 192, dev))
    assert rel_err(to_nchw(out), ref) < 1e-5
    up_ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    cat = E.Act.empty(2, 18, 10, 192 + 64, dev)
    E.upsample2x(xa, cat.slice(64, 192))
    assert rel_err(to_nchw(cat.slice(64, 192)), up_ref) < 1e-6
    mp = E.maxpool3x3s2(xa)
    assert rel_err(to_nchw(mp), F.max_pool2d(x, 3, 2, 1)) == 0.0
    rgb = torch.rand(2, 3, 8, 8, generator=g)
    a = E.nchw_to_nhwc(rgb.to(dev), cpad=4)
    back = a.buf.view(2, 8, 8, 4).cpu()
    assert torch.equal(back[..., :3].permute(0, 3, 1, 2), rgb) and float(back[..., 3].abs().max()) == 0.0
    assert torch.equal(E.nhwc_to_nchw(xa).cpu(), x)


def _model(net):
    from keypointfusion_amd.model.model import KPFusion
    m = KPFusion("KPFusion-" + net, "", 21, "dexycb", "")
    m.load_state_dict(synthetic_sd("KPFusion-" + net), strict=True)
    return m.to(_dev()).eval()


@pytest.mark.parametrize("net,B,S", [("convnext-tiny", 2, 128), ("convnext-tiny", 1, 64), ("resnet-18", 2, 128), ("resnet-18", 1, 64),
                                     ("convnext-base", 1, 64), ("convnext-small", 1, 64), ("convnext-tiny", 3, 96), ("resnet-50", 2, 128),
                                     ("resnet-50", 1, 64), ("resnet-101", 1, 64), ("convnext-large", 1, 64)])
def test_backbones_match_oracle(net, B, S):
    from oracle import kpf_oracle as O
    sd = synthetic_sd("KPFusion-" + net)
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, S, seed=1).items()}
    ref = O.backbones_forward(sd, b["img_rgb"], b["img"])
    m = _model(net)
    with torch.no_grad():
        out = m.forward_backbones(b["img_rgb"].to(_dev()), b["img"].to(_dev()))
    for o, r, name in zip(out, ref, ("img_offset", "img_feat", "img_offset_rgb", "img_feat_rgb")):
        e = rel_err(o, r)
        assert e < 1e-3, "%s: rel err %.2e" % (name, e)  # north_star tolerance
        assert e < 2e-4, "%s: rel err %.2e (regression guard)" % (name, e)
    # against the committed reference-generated fixture as well (S=64 fixture holds full img_offset tensors)
    if S == 64:  # (round 4: every family has a fixture — convnext tiny/small/base/large, resnet-18/50/101)
        z = np.load(os.path.join(GOLDEN, "backbone_%s_B1_S64.npz" % net))
        assert rel_err(out[0], torch.from_numpy(z["img_offset"])) < 1e-3
        assert rel_err(out[2], torch.from_numpy(z["img_offset_rgb"])) < 1e-3


# ----------------------------------------------------------------------------------------------------------------
# fusion head
# ----------------------------------------------------------------------------------------------------------------
def _run_full(net, B, seed=1, kernel=0.8):
    from oracle.compare import oracle_with_device_decisions
    sd = synthetic_sd("KPFusion-" + net)
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, 128, seed=seed).items()}
    m = _model(net)
    dev = _dev()
    plan = m._plan(dev)
    with torch.no_grad():
        out, sws, ctx = plan.forward(b["img_rgb"].to(dev), b["img"].to(dev), b["pcl"].to(dev), b["center"].to(dev), b["M"].to(dev),
                                     b["cube"].to(dev), b["cam_para"].to(dev), kernel, 128, 1, want_aux=True)
    torch.cuda.synchronize()
    ref, rsw, aux, report = oracle_with_device_decisions(sd, b, ctx, kernel=kernel)
    return b, ref, rsw, aux, out, sws, ctx, report


def test_wide_model_at_256x256_matches_the_oracle():
    """KPFusion(..., crop_size=256): the labelled WIDE extension of SURVEY section 0 — the reference's architecture with `fc_spatial2joint_feature` sized for
    the 64 x 64 feature map of a 256 x 256 crop (the reference hard-codes nn.Linear(32 * 32, 1), model/model.py:264, so it cannot run BASELINE's crop size at
    all).  Same kernels, F = 64 everywhere (4096-pixel gates, top-4 search over 4096 pixels): every output against the oracle at that size, same bars as at 128."""
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.weights import synthetic_state_dict
    from oracle.compare import oracle_with_device_decisions
    net, dev = "KPFusion-convnext-tiny", _dev()
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(net, 0, crop_size=256).items()}
    assert tuple(sd["block1.fc_spatial2joint_feature.weight"].shape) == (1, 4096)
    ref_sd = synthetic_sd(net)
    assert list(sd) == list(ref_sd) and [k for k in sd if sd[k].shape != ref_sd[k].shape] == ["block1.fc_spatial2joint_feature.weight", "block2.fc_spatial2joint_feature.weight"]
    m = KPFusion(net, "", 21, "dexycb", "", crop_size=256)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(2, 256, seed=4).items()}
    d = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad():
        out, sws, ctx = m._plan(dev).forward(d["img_rgb"], d["img"], d["pcl"], d["center"], d["M"], d["cube"], d["cam_para"], 0.8, 256, 1, want_aux=True)
    torch.cuda.synchronize()
    ref, rsw, aux, report = oracle_with_device_decisions(sd, b, ctx, img_size=256)
    print("index report wide model: %s" % report)
    assert report["top4_flips"] == 0 and report["ball_flips"] == 0, report  # frozen to what this seed gives, like the six committed 128 x 128 cases
    assert tuple(out[0].shape) == (2, 105, 64, 64) and tuple(sws[0].shape) == (2, 21, 64, 64)
    for o, r in list(zip(out, ref)) + list(zip(sws, rsw)):
        assert rel_err(o, r) < 1e-3
    for k in range(2, 6):
        assert float((out[k].cpu() - ref[k]).abs().max()) * 125.0 < 0.05

    class Loader:
        img_size, flip = 256, 1

    with torch.no_grad():
        res, sw2, _ = m(d["img_rgb"], d["img"], d["pcl"], Loader(), d["center"], d["M"], d["cube"], d["cam_para"], 0.8)
        assert all(torch.equal(a, c) for a, c in zip(res + sw2, out + sws))
        m.use_graphs = True  # hipGraph replay of the wide forward
        res_g = m(d["img_rgb"], d["img"], d["pcl"], Loader(), d["center"], d["M"], d["cube"], d["cam_para"], 0.8)[0]
        assert all(torch.equal(a, c) for a, c in zip(res_g, out))
        with pytest.raises(RuntimeError, match="needs 256x256 crops"):
            m(d["img_rgb"][..., :128, :128], d["img"][..., :128, :128], d["pcl"], Loader(), d["center"], d["M"], d["cube"], d["cam_para"], 0.8)
        with pytest.raises(RuntimeError, match="needs 128x128 crops"):  # the reference-sized model keeps refusing other sizes, like the reference's shape error
            _model("convnext-tiny")(d["img_rgb"], d["img"], d["pcl"], Loader(), d["center"], d["M"], d["cube"], d["cam_para"], 0.8)


def test_forward_kernel_argument_reaches_only_the_decode():
    """forward(..., kernel=0.6): the reference applies `kernel` in offset2joint_weight only; Block_KPFusion hard-codes
    pcl_joint2offset(joint_xyz, pcl, 0.8) (model/model.py:294).  A caller passing another kernel must still match the oracle."""
    b, ref, rsw, aux, out, sws, ctx, report = _run_full("convnext-tiny", 2, seed=3, kernel=0.6)
    assert report["top4_flips"] == 0
    assert rel_err(ctx["joint_uvd"], aux["joint_uvd"]) < 1e-4
    for i in (0, 1):
        assert rel_err(ctx["aux"][i]["X"], aux["block%d" % (i + 1)]["pcl_feat"]) < 1e-3
    for o, r in list(zip(out, ref)) + list(zip(sws, rsw)):
        assert rel_err(o, r) < 1e-3
    b8 = _run_full("convnext-tiny", 2, seed=3, kernel=0.8)
    assert rel_err(out[2], b8[4][2]) > 1e-4, "kernel must change the decoded joints"


def test_ball_query_decisions_over_a_seed_sweep():
    """How often does a ball-query set taken on the device differ from the oracle's, and is every difference a point ON the radius?  (VERDICT r05 item 4;
    model/model.py:129-204.)  The sets are built around NETWORK OUTPUTS (joints that agree with the oracle's to ~1e-6), so a point whose distance equals the
    radius to the last bits may change sides; pointnet2_ops' semantics are restated, not pinned.  16 input seeds x B = 4 = 16 x 4 x 21 x 3 x 2 =
    8064 sets: every differing set is proved to sit on the radius boundary inside `oracle_with_device_decisions` (`_check_ball_flips`: nothing clearly
    outside, at least one differing point within 1e-3 of r^2) and the outputs still match the oracle with the device's decisions injected.  The count is
    printed and bounded: a systematic disagreement (another scan order, `<=` for `<`, another slot fill) would flip hundreds of sets, not a handful."""
    flips, sets, worst = 0, 0, 0.0
    for seed in range(40, 56):
        b, ref, rsw, aux, out, sws, ctx, report = _run_full("convnext-tiny", 4, seed)
        assert report["top4_flips"] == 0, (seed, report)
        flips += report["ball_flips"]
        sets += 4 * 21 * 3 * 2
        for o, r in list(zip(out, ref)) + list(zip(sws, rsw)):
            assert rel_err(o, r) < 1e-3, (seed, report)
        worst = max(worst, max(float((out[k].cpu() - ref[k]).abs().max()) * 125.0 for k in range(2, 6)))
        assert worst < 0.05, (seed, worst)
    print("ball-query seed sweep: %d of %d sets differ from the oracle's own (all on the radius); max joint deviation %.5f mm" % (flips, sets, worst))
    assert flips <= sets // 500, (flips, sets)


@pytest.mark.parametrize("net,B,seed", [("convnext-tiny", 2, 1), ("resnet-18", 2, 1), ("convnext-tiny", 1, 1), ("convnext-tiny", 3, 7),
                                        ("resnet-50", 2, 1), ("convnext-base", 2, 1)])
def test_full_forward_matches_oracle(net, B, seed):
    """End-to-end: all 6 results and both spatial weights within 1e-3 relative of the oracle, joints within 0.05 mm.
    The top-4 pixel index tensor must EQUAL the oracle's (asserted inside oracle_with_device_decisions); ball-query sets are taken
    around network outputs and may differ by a point sitting on the radius (checked there too, then the rest of the pipeline is
    compared on equal decisions)."""
    b, ref, rsw, aux, out, sws, ctx, report = _run_full(net, B, seed)
    print("index report %s B=%d seed=%d: %s" % (net, B, seed, report))
    # frozen to what these seeds give: the device's ball-query sets equal the oracle's on every committed case
    assert report["top4_flips"] == 0 and report["ball_flips"] == 0, report
    assert torch.equal(ctx["img_xyz"].cpu(), O_img_xyz(aux, b)), "pixel positions must be bit-identical to the oracle's"
    assert rel_err(ctx["joint_uvd"], aux["joint_uvd"]) < 1e-4
    assert rel_err(ctx["joint_xyz0"], aux["joint_xyz0"]) < 1e-4
    assert rel_err(ctx["closeness"], aux["pcl_closeness"]) < 1e-3
    for i in (0, 1):
        a, g = aux["block%d" % (i + 1)], ctx["aux"][i]
        assert rel_err(g["X"], a["pcl_feat"]) < 1e-3, "point features block %d" % (i + 1)
        assert rel_err(g["D"], a["joint_feat_desa"]) < 2e-3, "DESA block %d" % (i + 1)
        assert rel_err(g["h_init"], a["h_init"]) < 2e-3
        assert rel_err(g["dec"], a["dec"]) < 2e-3
    names = ["img_offset", "img_offset_rgb", "r3d1", "r2d1", "r3d2", "r2d2"]
    for o, r, n in zip(out, ref, names):
        e = rel_err(o, r)
        assert e < 1e-3, "%s rel err %.2e" % (n, e)
    for o, r in zip(sws, rsw):
        assert rel_err(o, r) < 1e-3
    # north_star accuracy bound: joints within 0.05 mm of the reference path (cube 250 mm => x * 125 mm)
    for k in range(2, 6):
        mm = float((out[k].cpu() - ref[k]).abs().max()) * 125.0
        assert mm < 0.05, "%s deviates %.4f mm" % (names[k], mm)
    # "argmax indices" (SURVEY D7): bit-exact argmax of masked weight logits and of the spatial weights
    d = F.interpolate(b["img"], [32, 32])
    for o, r in ((out[0], ref[0]), (out[1], ref[1])):
        wg = o.cpu()[:, 84:].masked_fill(d > 0.99, -1e8).reshape(B, 21, -1)
        wr = r[:, 84:].masked_fill(d > 0.99, -1e8).reshape(B, 21, -1)
        top2 = torch.topk(wr, 2, -1)[0]
        decided = (top2[..., 0] - top2[..., 1]) > 1e-3
        assert bool((wg.argmax(-1) == wr.argmax(-1))[decided].all())
    for o, r in zip(sws, rsw):
        og, rr = o.cpu().reshape(B, 21, -1), r.reshape(B, 21, -1)
        top2 = torch.topk(rr, 2, -1)[0]
        decided = (top2[..., 0] - top2[..., 1]) > 1e-4
        assert bool((og.argmax(-1) == rr.argmax(-1))[decided].all())


def O_img_xyz(aux, b):
    from oracle import kpf_oracle as O
    return O.img_xyz_grid(aux["img_down"], b["center"], b["M"], b["cube"], b["cam_para"])


def test_ball_group_bit_exact_on_identical_inputs():
    """Ball query indices are integer work: with identical fp32 inputs the HIP kernel must reproduce the oracle exactly
    (empty, partially filled and saturated neighbourhoods all occur at these radii), and the grouped operand rows too."""
    import ctypes as C
    from keypointfusion_amd import engine as E, lib as L
    from oracle import kpf_oracle as O
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    B, N = 3, 1024
    pcl = (torch.rand(B, N, 3, generator=g) * 1.6 - 0.8)
    joints = (torch.rand(B, 21, 3, generator=g) * 1.6 - 0.8)
    joints[0, 0] = 5.0  # isolated query: finds only itself
    X = torch.randn(B, N, 128, generator=g)
    JF = torch.randn(B, 21, 128, generator=g)
    xyz = torch.cat((pcl, joints), 1)
    feat = torch.cat((X, JF), 1)
    G = torch.empty(3, B * 21 * 64, 132, device=dev)
    idx = torch.empty(3, B * 21, 64, device=dev, dtype=torch.int32)
    d = [t.to(dev) for t in (pcl, joints, X, JF)]
    L.check(L.load().kpf_ball_group_f32(E._ptr(d[0]), E._ptr(d[1]), E._ptr(d[2]), E._ptr(d[3]), 128, E._ptr(G), E._ptr(idx), B, N,
                                        0.1, 0.2, 0.4, E._stream()))
    torch.cuda.synchronize()
    fills = []
    for ri, r in enumerate((0.1, 0.2, 0.4)):
        ref = O.ball_query(r, 64, xyz, joints)
        got = idx[ri].cpu().long().view(B, 21, 64)
        assert torch.equal(got, ref), "radius %g: %d index mismatches" % (r, int((got != ref).sum()))
        fills.append(int((ref != ref[..., :1]).any(-1).sum()))
        flat = ref.reshape(B, -1)
        gx = (torch.gather(xyz, 1, flat.unsqueeze(-1).expand(-1, -1, 3)).view(B, 21, 64, 3) - joints.unsqueeze(2)) / r
        gf = torch.gather(feat, 1, flat.unsqueeze(-1).expand(-1, -1, 128)).view(B, 21, 64, 128) - JF.unsqueeze(2)
        rows = G[ri].cpu().view(B, 21, 64, 132)
        assert torch.equal(rows[..., :128], gf)
        assert rel_err(rows[..., 128:131], gx) < 1e-6
        assert float(rows[..., 131].abs().max()) == 0.0
    assert int(idx[0].cpu().view(B, 21, 64)[0, 0].unique().numel()) == 1  # the isolated query holds its own index 64 times


def test_top4_bit_exact_on_identical_inputs():
    """img2pcl_index: same pixel positions in, same integer indices out (ties aside) — checked with the positions the kernel
    itself produced, so only the search is under test."""
    from keypointfusion_amd import engine as E, lib as L
    dev = _dev()
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(2, 128, seed=5).items()}
    B, N = 2, 1024
    d = {k: v.to(dev) for k, v in b.items()}
    clos = torch.empty(B, N, 4, device=dev)
    idx = torch.empty(B, N, 4, device=dev, dtype=torch.int32)
    ixyz = torch.empty(B, 1024, 3, device=dev)
    L.check(L.load().kpf_img2pcl_top4_f32(E._ptr(d["pcl"]), E._ptr(d["img"]), E._ptr(d["center"]), E._ptr(E.crop_inverse(d["M"])), E._ptr(d["cube"]),
                                          E._ptr(d["cam_para"]), E._ptr(clos), E._ptr(idx), E._ptr(ixyz), B, N, 128, 32, 128, 1, E._stream()))
    torch.cuda.synchronize()
    px = ixyz.cpu()
    dist = torch.sum(torch.pow(b["pcl"].unsqueeze(2) - px.unsqueeze(1), 2), dim=-1)
    val, ref = torch.topk(dist, 4, largest=False)
    got = idx.cpu().long()
    # distances of the chosen pixels must be exactly the 4 smallest, in ascending order (robust to exact ties in index)
    assert torch.equal(torch.gather(dist, 2, got), val)
    c = 1 / (val + 1e-8)
    assert rel_err(clos, c / (c.sum(-1, keepdim=True) + 1e-8)) < 1e-6


def test_module_boundary_forward_signature_and_errors():
    """The nn.Module boundary: reference call signature (with a `loader` object), return structure, state-dict reload,
    the reference's own failure at S != 128, and a loud error instead of a CPU fallback."""
    from oracle import kpf_oracle as O
    net = "convnext-tiny"
    m = _model(net)
    dev = _dev()
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(1, 128, seed=2).items()}

    class Loader:
        img_size, flip = 128, 1

    args = [b[k].to(dev) for k in ("img_rgb", "img", "pcl")] + [Loader()] + [b[k].to(dev) for k in ("center", "M", "cube", "cam_para")]
    with torch.no_grad():
        res, sws, none = m(*args, 0.8)
    assert none is None and len(res) == 6 and len(sws) == 2
    assert [tuple(t.shape) for t in res] == [(1, 105, 32, 32)] * 2 + [(1, 21, 3)] * 4
    assert [tuple(t.shape) for t in sws] == [(1, 21, 32, 32)] * 2
    assert all(t.is_cuda and t.dtype == torch.float32 for t in res + sws)
    sd = synthetic_sd("KPFusion-" + net)
    ref, _ = O.kpfusion_forward(sd, b["img_rgb"], b["img"], b["pcl"], b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
    assert rel_err(res[5], ref[5]) < 1e-3
    # a checkpoint in the reference's format ({"model": sd} with "module." keys, loaded by intersection: train.py:100-107)
    sd2 = {k: (v * 0.5 if k == "backbone_d.finals.0.weight" else v) for k, v in sd.items()}
    ck = {"module." + k: v for k, v in sd2.items()}
    own = m.state_dict()
    inter = {k[len("module."):]: v for k, v in ck.items() if k[len("module."):] in own}
    own.update(inter)
    m.load_state_dict(own)
    with torch.no_grad():
        res2, _, _ = m(*args, 0.8)
    with torch.no_grad():
        ref2 = O.unet(sd2, "backbone_d", b["img"])[0]
    assert rel_err(res2[0], ref2) < 1e-3 and rel_err(res2[0], ref[0]) > 1e-2  # weights were repacked after load_state_dict
    with pytest.raises(RuntimeError):
        m(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)  # CPU tensors
    big = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(1, 256, seed=2).items()}
    with pytest.raises(RuntimeError):
        m(big["img_rgb"], big["img"], big["pcl"], Loader(), big["center"], big["M"], big["cube"], big["cam_para"], 0.8)


def test_graph_replay_equals_eager():
    """hipGraph replay of the whole forward (opt-in, small-batch latency path) returns exactly what the eager launches return,
    also after the inputs change (static-buffer copy) — the graph holds no stale pointers."""
    m = _model("convnext-tiny")
    dev = _dev()

    class Loader:
        img_size, flip = 128, 1

    outs = []
    for seed in (3, 4):
        b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(2, 128, seed=seed).items()}
        args = [b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8]
        with torch.no_grad():
            m.use_graphs = False
            eager = m(*args)
            m.use_graphs = True
            graphed = m(*args)
        m.use_graphs = False
        for a, g in zip(eager[0] + eager[1], graphed[0] + graphed[1]):
            assert torch.equal(a, g)
        outs.append(graphed[0][5].clone())
    assert not torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("C,M", [(96, 128 * 3 + 37), (128, 64 * 5), (192, 64 * 2 + 5)])
def test_fused_convnext_mlp(C, M):
    """kpf_convnext_mlp_f32 == x + gamma * pwconv2(gelu(pwconv1(y))) (convNeXT/convnext.py:44-51), ragged row count, in place."""
    from keypointfusion_amd import engine as E, lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(C + M)
    y = torch.randn(M, C, generator=g)
    x = torch.randn(M, C, generator=g)
    w1 = torch.randn(4 * C, C, generator=g) / C ** 0.5
    b1 = torch.randn(4 * C, generator=g) * 0.5
    w2 = torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5
    b2 = torch.randn(C, generator=g)
    gam = torch.rand(C, generator=g)
    ref = x + gam * F.linear(F.gelu(F.linear(y, w1, b1)), w2, b2)
    d = [t.to(dev) for t in (y, x, w1, b1, w2, b2, gam)]
    L.check(L.load().kpf_convnext_mlp_f32(*[E._ptr(t) for t in d], E._ptr(d[1]), M, C, E._stream()))
    torch.cuda.synchronize()
    assert rel_err(d[1], ref) < 1e-5


# ----------------------------------------------------------------------------------------------------------------
# split (3 x f16 MFMA) GEMM arithmetic
# ----------------------------------------------------------------------------------------------------------------
def to_split(t):
    """fp32 [..., C] -> same-shape fp32 tensor whose bytes are [32 x f16 hi | 32 x f16 lo] per 32-channel block (include/kpf.h)."""
    C = t.shape[-1]
    hi = t.half()
    lo = (t - hi.float()).half()
    blk = torch.stack([hi.reshape(-1, C // 32, 32), lo.reshape(-1, C // 32, 32)], 2).contiguous()
    return blk.view(torch.float32).reshape(t.shape)


def from_split(t):
    C = t.shape[-1]
    blk = t.contiguous().view(torch.float16).reshape(-1, C // 32, 2, 32).float()
    return (blk[:, :, 0] + blk[:, :, 1]).reshape(t.shape)


@pytest.mark.parametrize("M,N,K,wscale", [(1024, 384, 96, 0.1), (16384, 1536, 384, 0.05), (4096, 768, 3072, 0.02), (300, 96, 384, 1e-3),
                                          (777, 160, 64, 0.3)])
def test_split_gemm_is_as_accurate_as_fp32(M, N, K, wscale):
    """kpf_conv2d_f32 with KPF_IN_SPLIT against an fp64 product: the 3 x f16 split must be at least as close to the exact result as
    the f32-input MFMA path and the CPU's fp32 GEMM (per-product error ~2^-21, fp32 accumulation in all three)."""
    from keypointfusion_amd.engine import Act, PackedConv, conv
    from keypointfusion_amd import lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * wscale
    bias = torch.randn(N, generator=g) * 0.1
    ref = a.double() @ w.double().t() + bias.double()
    pc = PackedConv(w, bias, dev)
    xa = Act(a.to(dev).view(-1), M, 1, 1, K)
    o32 = conv(pc, xa).buf.view(M, N).cpu().double()
    xs = Act(to_split(a).to(dev).view(-1), M, 1, 1, K, split=True)
    osp = conv(pc, xs).buf.view(M, N).cpu().double()
    cpu32 = (a @ w.t() + bias).double()
    den = ref.abs().max()
    e_split, e_mfma, e_cpu = [float((o - ref).abs().max() / den) for o in (osp, o32, cpu32)]
    assert e_split < 2e-6, (e_split, e_mfma, e_cpu)
    assert e_split < 2.0 * max(e_mfma, e_cpu) + 1e-7, (e_split, e_mfma, e_cpu)
    # split output of the same product (GELU epilogue): decode and compare
    if N % 32 == 0:
        out = Act(torch.empty(M * N, device=dev), M, 1, 1, N)
        conv(pc, xs, out=out, flags=L.KPF_ACT_GELU, out_split=True)
        got = from_split(out.buf.view(M, N).cpu()).double()
        want = F.gelu(ref)
        assert float((got - want).abs().max() / want.abs().max()) < 2e-6


def test_split_operand_precision_floor_is_what_the_docs_say():
    """split operands are stored unscaled, so an element of magnitude |x| keeps about 25 + log2|x| bits (22 from |x| = 1/8
    up).  With every activation ~1e-4 the split GEMM is therefore only ~12-bit accurate ELEMENTWISE — which is why the engine uses it
    solely behind LayerNorm / GELU outputs with a pack-time bound — while the f32-input MFMA (the default) stays at fp32 accuracy."""
    from keypointfusion_amd.engine import Act, PackedConv, conv
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    M, N, K = 512, 128, 256
    w = torch.randn(N, K, generator=g) / K ** 0.5
    pc = PackedConv(w, None, dev)
    for scale, lo_bits, hi_bits in ((1.0, 19, 30), (1e-4, 9, 15)):
        a = torch.randn(M, K, generator=g).abs() * scale + scale  # one sign: no cancellation, so the elementwise relative error is meaningful
        wp = w.abs()
        pcp = PackedConv(wp, None, dev)
        ref = a.double() @ wp.double().t()
        o32 = conv(pcp, Act(a.to(dev).view(-1), M, 1, 1, K)).buf.view(M, N).cpu().double()
        osp = conv(pcp, Act(to_split(a).to(dev).view(-1), M, 1, 1, K, split=True)).buf.view(M, N).cpu().double()
        e32 = float(((o32 - ref).abs() / ref).max())
        esp = float(((osp - ref).abs() / ref).max())
        assert e32 < 2e-6, (scale, e32)                                  # IEEE fp32 at any magnitude
        assert 2.0 ** -hi_bits < esp < 2.0 ** -lo_bits, (scale, esp)     # the documented floor, neither better nor worse
    del pc


def test_split_layernorm_producers():
    from keypointfusion_amd import lib as L
    from keypointfusion_amd.engine import _ptr, _stream
    dev = _dev()
    lib = L.load()
    torch.manual_seed(3)
    for B, H, W, C in ((2, 16, 16, 192), (1, 8, 8, 384), (2, 12, 20, 96), (1, 4, 4, 768)):
        x = torch.randn(B, H, W, C) * 3 + 0.5
        wdw, bdw = torch.randn(49, C) / 7, torch.randn(C) * 0.1
        lw, lb = torch.rand(C) + 0.5, torch.randn(C) * 0.1
        dx, dw, db, dlw, dlb = (t.to(dev).contiguous() for t in (x, wdw, bdw, lw, lb))
        y32 = torch.empty_like(dx)
        ysp = torch.empty_like(dx)
        L.check(lib.kpf_dwconv7_ln_f32(_ptr(dx), _ptr(dw), _ptr(db), _ptr(dlw), _ptr(dlb), _ptr(y32), B, H, W, C, 1e-6, _stream()), "dw")
        L.check(lib.kpf_dwconv7_ln_split_f32(_ptr(dx), _ptr(dw), _ptr(db), _ptr(dlw), _ptr(dlb), _ptr(ysp), B, H, W, C, 1e-6, _stream()), "dws")
        assert rel_err(from_split(ysp.cpu()), y32.cpu()) < 1e-6
        z32 = torch.empty_like(dx)
        zsp = torch.empty_like(dx)
        L.check(lib.kpf_layernorm_f32(_ptr(dx), _ptr(dlw), _ptr(dlb), _ptr(z32), B * H * W, C, 1e-6, _stream()), "ln")
        L.check(lib.kpf_layernorm_split_f32(_ptr(dx), _ptr(dlw), _ptr(dlb), _ptr(zsp), B * H * W, C, 1e-6, _stream()), "lns")
        assert rel_err(from_split(zsp.cpu()), z32.cpu()) < 1e-6
        inpl = dx.clone()  # in place, as the two-kernel depthwise path uses it
        L.check(lib.kpf_layernorm_split_f32(_ptr(inpl), _ptr(dlw), _ptr(dlb), _ptr(inpl), B * H * W, C, 1e-6, _stream()), "lns")
        assert torch.equal(inpl.cpu().view(torch.int32), zsp.cpu().view(torch.int32))


@pytest.mark.parametrize("net,B,S", [("convnext-tiny", 2, 128), ("convnext-base", 1, 64), ("convnext-tiny", 3, 96), ("resnet-18", 2, 128),
                                     ("resnet-50", 1, 64)])
@pytest.mark.parametrize("unfused", [False, True])
def test_backbones_match_oracle_in_split_mode(net, B, S, unfused, monkeypatch):
    """KPF_GEMM=split: split (3 x f16) arithmetic runs ONLY where a pack-time bound puts the operands inside the f16 range — the
    ConvNeXt blocks' pointwise GEMMs and the downsample convolutions behind a LayerNorm; pre-activation Residuals, ResNet stages and
    heads have unbounded operands and stay on the f32-input MFMA.  Same parity bars as the f32 path."""
    from keypointfusion_amd import engine as E
    from keypointfusion_amd.spec import CONVNEXT
    from oracle import kpf_oracle as O
    monkeypatch.setattr(E, "GEMM_MODE", "split")
    monkeypatch.setattr(E, "FORCE_UNFUSED_MLP", unfused)
    sd = synthetic_sd("KPFusion-" + net)
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, S, seed=1).items()}
    ref = O.backbones_forward(sd, b["img_rgb"], b["img"])
    m = _model(net)
    launched = []
    monkeypatch.setattr(E, "PROFILE", launched)
    with torch.no_grad():
        out = m.forward_backbones(b["img_rgb"].to(_dev()), b["img"].to(_dev()))
    n_split = sum(1 for r in launched if r[0] == "igemm_split_kernel")
    n_fused = sum(1 for r in launched if r[0] == "convnext_mlp_split_kernel")
    if "convnext" in net:
        depths = CONVNEXT[net.split("-")[-1]][0]
        assert n_split + 2 * n_fused == 2 * (2 * sum(depths) + 3), "split GEMMs: %d plain + %d fused blocks" % (n_split, n_fused)
        if not unfused:
            assert n_fused >= 2 * 3, "fused split MLP did not run"
    else:
        assert n_split == 0 and n_fused == 0, "no operand of a ResNet backbone has a pack-time range proof: nothing may run split"
    for o, r, name in zip(out, ref, ("img_offset", "img_feat", "img_offset_rgb", "img_feat_rgb")):
        e = rel_err(o, r)
        assert e < 2e-4, "%s: rel err %.2e" % (name, e)


def test_split_mode_falls_back_to_f32_when_the_range_proof_fails(monkeypatch):
    """A checkpoint whose LayerNorm scale makes the pack-time bound exceed the f16 range (|LN out| <= sqrt(C) max|w| + max|b|, times
    the row-L1 norm of pwconv1 for the hidden tensor) must not run split: the block falls back to the f32 MFMA and the output stays
    right — with activations far beyond 65504 that a split store would have clamped silently."""
    from keypointfusion_amd import engine as E
    from oracle import kpf_oracle as O
    monkeypatch.setattr(E, "GEMM_MODE", "split")
    net = "convnext-tiny"
    sd = dict(synthetic_sd("KPFusion-" + net))
    hot = "backbone_d.backbone.stages.1.0"  # one block: LayerNorm scale x 3e4 -> hidden activations ~1e6; layer scale compensates
    sd[hot + ".norm.weight"] = sd[hot + ".norm.weight"] * 3.0e4
    sd[hot + ".gamma"] = sd[hot + ".gamma"] / 3.0e4
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(1, 64, seed=1).items()}
    ref = O.backbones_forward(sd, b["img_rgb"], b["img"])
    m = _model(net)
    m.load_state_dict(sd)
    plan = m._plan(_dev())
    blk = plan.backbone_d.stages[1][0]
    assert not blk.split_ok and not blk.pw1.split_allowed and plan.backbone_d.stages[1][1].split_ok
    launched = []
    monkeypatch.setattr(E, "PROFILE", launched)
    with torch.no_grad():
        out = m.forward_backbones(b["img_rgb"].to(_dev()), b["img"].to(_dev()))
    assert any(r[0] == "igemm_f32_kernel" and r[5][:3] == (8 * 8, 768, 192) for r in launched), "the unproven block must run on the f32 MFMA"
    for o, r in zip(out, ref):
        assert rel_err(o, r) < 2e-4


@pytest.mark.parametrize("net,B,S", [("convnext-tiny", 2, 128), ("resnet-18", 1, 64), ("convnext-base", 1, 64)])
def test_backbones_match_oracle_in_f32_mfma_mode(net, B, S, monkeypatch):
    """KPF_GEMM=f32: every GEMM on the f32-input MFMA (the strict-fp32 arithmetic path)."""
    from keypointfusion_amd import engine as E
    from oracle import kpf_oracle as O
    monkeypatch.setattr(E, "GEMM_MODE", "f32")
    sd = synthetic_sd("KPFusion-" + net)
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, S, seed=1).items()}
    ref = O.backbones_forward(sd, b["img_rgb"], b["img"])
    m = _model(net)
    launched = []
    monkeypatch.setattr(E, "PROFILE", launched)
    with torch.no_grad():
        out = m.forward_backbones(b["img_rgb"].to(_dev()), b["img"].to(_dev()))
    assert not any(r[0].endswith("split_kernel") for r in launched)
    for o, r in zip(out, ref):
        assert rel_err(o, r) < 2e-4


@pytest.mark.parametrize("mode", ["split", "f32"])
def test_full_forward_matches_oracle_in_both_gemm_modes(mode, monkeypatch):
    from keypointfusion_amd import engine as E
    monkeypatch.setattr(E, "GEMM_MODE", mode)
    b, ref, rsw, aux, out, sws, ctx, report = _run_full("convnext-tiny", 2, 1)
    names = ["img_offset", "img_offset_rgb", "r3d1", "r2d1", "r3d2", "r2d2"]
    for o, r, n in zip(out, ref, names):
        assert rel_err(o, r) < 1e-3, "%s rel err %.2e" % (n, rel_err(o, r))
    for k in range(2, 6):
        assert float((out[k].cpu() - ref[k]).abs().max()) * 125.0 < 0.05


@pytest.mark.parametrize("C,M", [(96, 4096 + 37), (128, 1000), (192, 2048 + 5), (256, 777)])
def test_fused_split_mlp_matches_fp64(C, M):
    """kpf_convnext_mlp_split_f32 (hidden tensor in registers, f16 matrix cores) against an fp64 evaluation of the block's MLP."""
    from keypointfusion_amd import lib as L
    from keypointfusion_amd.engine import MLP_HIDDEN_PERM, _ptr, _stream, split_pack
    dev = _dev()
    lib = L.load()
    g = torch.Generator().manual_seed(C)
    y = torch.randn(M, C, generator=g)
    x = torch.randn(M, C, generator=g)
    w1 = torch.randn(4 * C, C, generator=g) / C ** 0.5
    b1 = torch.randn(4 * C, generator=g) * 0.1
    w2 = torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5
    b2 = torch.randn(C, generator=g) * 0.1
    gamma = torch.rand(C, generator=g) * 0.2 + 0.05
    ref = x.double() + gamma.double() * (F.gelu(y.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double())
    w1s, us1 = split_pack(w1)
    perm = torch.tensor([32 * q + k for q in range(4 * C // 32) for k in MLP_HIDDEN_PERM])
    w2s, us2 = split_pack(w2[:, perm])
    d = lambda t: t.to(dev).contiguous()  # noqa: E731
    out = torch.empty(M, C, device=dev)
    ys, xd, w1d, w2d, b1d, b2d, gd = d(to_split(y)), d(x), d(w1s), d(w2s), d(b1), d(b2), d(gamma)
    L.check(lib.kpf_convnext_mlp_split_f32(_ptr(ys), _ptr(xd), _ptr(w1d), _ptr(b1d), us1, _ptr(w2d), _ptr(b2d), us2, _ptr(gd), _ptr(out), M, C,
                                           _stream()), "mlp_split")
    e = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert e < 2e-6, e


def test_split_arithmetic_is_no_less_accurate_than_fp32_end_to_end(monkeypatch):
    """Both ConvNeXt-T backbones at 64x64 against an fp64 evaluation of the oracle: the default split arithmetic (3 x f16 MFMA per
    product) must be as close to the exact forward as fp32 implementations are — the f32-input MFMA path and the CPU fp32 oracle."""
    from keypointfusion_amd import engine as E
    from oracle import kpf_oracle as O
    net = "convnext-tiny"
    sd = synthetic_sd("KPFusion-" + net)
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(1, 64, seed=1).items()}
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    exact = O.backbones_forward(sd64, b["img_rgb"].double(), b["img"].double())
    cpu32 = O.backbones_forward(sd, b["img_rgb"], b["img"])
    outs = {}
    for mode in ("split", "f32"):
        monkeypatch.setattr(E, "GEMM_MODE", mode)
        m = _model(net)
        with torch.no_grad():
            outs[mode] = [t.cpu().double() for t in m.forward_backbones(b["img_rgb"].to(_dev()), b["img"].to(_dev()))]
    for i, name in enumerate(("img_offset", "img_feat", "img_offset_rgb", "img_feat_rgb")):
        den = float(exact[i].abs().max())
        e_split = float((outs["split"][i] - exact[i]).abs().max()) / den
        e_mfma = float((outs["f32"][i] - exact[i]).abs().max()) / den
        e_cpu = float((cpu32[i].double() - exact[i]).abs().max()) / den
        print("%s: |err| / max|exact|  split %.2e  f32-MFMA %.2e  CPU fp32 %.2e" % (name, e_split, e_mfma, e_cpu))
        assert e_split < 1e-4, (name, e_split)
        assert e_split <= 2.0 * max(e_mfma, e_cpu), "%s: split %.2e vs f32-MFMA %.2e, CPU fp32 %.2e" % (name, e_split, e_mfma, e_cpu)


def test_full_size_bench_config_properties():
    """BASELINE.json configs[1] at its full size (B=64, 256x256, ConvNeXt-T, both backbones): size-independent properties of the
    forward — every sample is independent of its batch mates and of its position in the batch, the graph replay the benchmark
    times equals the eager launches — plus the CPU oracle on two of the 64 samples."""
    from oracle import kpf_oracle as O
    net, B, S = "convnext-tiny", 64, 256
    sd = synthetic_sd("KPFusion-" + net)
    hb = synthetic_batch(B, S, seed=1)
    img, rgb = torch.from_numpy(hb["img"]).to(_dev()), torch.from_numpy(hb["img_rgb"]).to(_dev())
    m = _model(net)
    plan = m._plan(_dev())
    with torch.no_grad():
        full = [t.clone() for t in m.forward_backbones(rgb, img)]
        # graph replay (what bench.py times) == eager
        (od, fd), (orgb, frgb) = plan.backbones_graphed(img, rgb)
        from keypointfusion_amd.engine import nhwc_to_nchw
        for a, b in zip(full, (od, nhwc_to_nchw(fd), orgb, nhwc_to_nchw(frgb))):
            assert torch.equal(a, b)
        # sample independence: a sample run alone gives the same result as inside the batch
        for s in (0, 31, 63):
            one = m.forward_backbones(rgb[s:s + 1], img[s:s + 1])
            for a, b in zip(full, one):
                assert rel_err(b, a[s:s + 1].cpu()) < 1e-5
        # position independence: a permuted batch gives the permuted outputs
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(_dev())
        pout = m.forward_backbones(rgb[perm], img[perm])
        for a, b in zip(full, pout):
            assert rel_err(b, a[perm].cpu()) < 1e-6
    # the CPU oracle on two samples of the batch
    for s in (5, 40):
        ref = O.backbones_forward(sd, torch.from_numpy(hb["img_rgb"][s:s + 1]), torch.from_numpy(hb["img"][s:s + 1]))
        for a, r in zip(full, ref):
            assert rel_err(a[s:s + 1], r) < 2e-4


def test_module_under_dataparallel_and_threads():
    """The reference wraps the model in torch.nn.DataParallel (train.py:81, demo_RGBD.py:49) and calls it from worker threads: the
    drop-in must survive replicate() and concurrent forwards from several Python threads on their own streams."""
    import threading
    net = "resnet-18"
    m = _model(net)
    dev = _dev()
    b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(2, 128, seed=3).items()}

    class Loader:
        img_size, flip = 128, 1

    args = [b[k] for k in ("img_rgb", "img", "pcl")] + [Loader()] + [b[k] for k in ("center", "M", "cube", "cam_para")]
    with torch.no_grad():
        want, want_sw, _ = m(*args, 0.8)
        dp = torch.nn.DataParallel(m, device_ids=[0])
        got, got_sw, _ = dp(*args, 0.8)
    for a, c in zip(want + want_sw, got + got_sw):
        assert torch.equal(a, c)
    outs, errs = {}, []

    def worker(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=dev)), torch.no_grad():
                outs[i] = m(*args, 0.8)[0]
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    torch.cuda.synchronize()
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(3):
        for a, c in zip(want, outs[i]):
            assert torch.equal(a, c)


def test_offset2joint_edge_cases():
    """Masked soft-argmax decode (model/model.py:466-500) on the inputs where the reference's masks are delicate: an all-background
    crop (every logit masked: softmax over a constant), pixels at exactly 0.99 (`< 0.99` and `> 0.99` are not complements there:
    unmasked logit, zero offset term), a single foreground pixel, and a large-logit crop (softmax overflow guard)."""
    from keypointfusion_amd import lib as L
    from keypointfusion_amd.engine import _ptr, _stream, crop_inverse
    from oracle import kpf_oracle as O
    dev = _dev()
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    B, S, Fs = 5, 128, 32
    off = torch.randn(B, 105, Fs, Fs, generator=g)
    off[4, 84:] *= 60.0  # large logits
    depth = torch.ones(B, 1, S, S)
    depth[1] = torch.rand(1, S, S, generator=g) * 1.2 - 0.6
    depth[1, 0][torch.rand(S, S, generator=g) < 0.3] = 0.99      # exactly on the threshold
    depth[1, 0][torch.rand(S, S, generator=g) < 0.3] = 1.0
    depth[2, 0, 66, 70] = 0.25                                     # one foreground pixel (it is sampled by the nearest downsample)
    depth[2, 0, 64, 68] = 0.25
    depth[3] = torch.rand(1, S, S, generator=g) * 1.2 - 0.6
    depth[4] = torch.rand(1, S, S, generator=g) * 1.2 - 0.6
    sb = synthetic_batch(B, S, seed=3)
    center, M, cube, cam = (torch.from_numpy(sb[k]) for k in ("center", "M", "cube", "cam_para"))
    uvd_ref = O.offset2joint_weight(off, depth, 0.8)
    xyz_ref = O.uvd2xyz(uvd_ref, center, M, cube, cam, 128, 1)
    d = [t.to(dev).contiguous() for t in (off, depth, center, M, cube, cam)]
    d[3] = crop_inverse(d[3])  # the geometry kernels take M^-1 (ABI v4)
    uvd = torch.empty(B, 21, 3, device=dev)
    xyz = torch.empty(B, 21, 3, device=dev)
    L.check(lib.kpf_offset2joint_f32(_ptr(d[0]), _ptr(d[1]), _ptr(d[2]), _ptr(d[3]), _ptr(d[4]), _ptr(d[5]), _ptr(uvd), _ptr(xyz), B, S, Fs, 0.8,
                                     128, 1, _stream()), "offset2joint")
    assert torch.isfinite(uvd).all() and torch.isfinite(xyz).all()
    for b in range(B):
        assert float((uvd[b].cpu() - uvd_ref[b]).abs().max()) < 2e-5, (b, float((uvd[b].cpu() - uvd_ref[b]).abs().max()))
    assert rel_err(xyz, xyz_ref) < 1e-4


def test_two_devices_in_one_process_opt_in_per_device():
    """The > 64 KiB dynamic-LDS opt-in is a per-device attribute: a process that drives two GPUs — torch.nn.DataParallel, the
    reference's own wrapper — must be able to run the large-tile GEMMs and the fused MLP on the second device too.  Needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs in one process (the 1-GPU test boxes cannot exercise this)")
    from oracle import kpf_oracle as O
    sd = synthetic_sd("KPFusion-convnext-tiny")
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(2, 128, seed=1).items()}
    ref = O.backbones_forward(sd, b["img_rgb"], b["img"])
    m = _model("convnext-tiny")
    outs = []
    for d in (0, 1):
        dev = torch.device("cuda", d)
        with torch.no_grad():
            outs.append([t.cpu() for t in m.to(dev).forward_backbones(b["img_rgb"].to(dev), b["img"].to(dev))])
    for o0, o1, r in zip(outs[0], outs[1], ref):
        assert rel_err(o0, r) < 2e-4 and rel_err(o1, r) < 2e-4


@pytest.mark.parametrize("stages", [True, False])
def test_pipelined_eval_returns_what_forward_returns(stages):
    """serving.PipelinedEval: several batches in flight on independent graph slots / streams must give, batch by batch, exactly the
    tensors the module's own forward gives (same kernels, same order of arithmetic)."""
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.serving import PipelinedEval
    from keypointfusion_amd.weights import synthetic_batch, synthetic_state_dict
    dev = _dev()
    net = "KPFusion-resnet-18"
    m = KPFusion(net, "", 21, "dexycb", "")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(net, 0).items()}, strict=True)
    m = m.to(dev).eval()

    class Loader:
        img_size, flip = 128, 1

    batches = [{k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(3, 128, seed=20 + i).items()} for i in range(5)]
    args = lambda b: (b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
    with torch.no_grad():
        ref = [m(*args(b)) for b in batches]
    pe = PipelinedEval(m, depth=2, stages=stages)
    for feed in (torch.cuda.current_stream(dev), pe.feed_stream(dev)):  # from the default stream (serialised, correct) and from a loop stream (overlapped)
        feed.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(feed):
            tickets = [pe.submit(*args(b)) for b in batches]
            for t, (rres, rsw, _) in zip(tickets, ref):
                res, sws, _ = pe.collect(t)
                assert all(torch.equal(a, b) for a, b in zip(res + sws, rres + rsw))
        torch.cuda.current_stream(dev).wait_stream(feed)
    with pytest.raises(RuntimeError):
        m.train()
        pe.submit(*args(batches[0]))
