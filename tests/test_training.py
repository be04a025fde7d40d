"""Training-step slice: loss codec, SmoothL1Loss, the loss schedule of train.py:211-261 and its first-step gradients
against fixtures generated from the imported reference (tests/golden/gen_golden_train.py); the bucketed gradient all-reduce over gloo
(world size 2) driven by the product's own hooks; the HIP convolution autograd Function against torch autograd (GPU)."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from keypointfusion_amd import training as T
from oracle import train_oracle as TO
from keypointfusion_amd.weights import synthetic_batch

Z = np.load(os.path.join(GOLDEN, "train_loss.npz"))


def _inputs():
    g = torch.Generator().manual_seed(21)
    B, J, Fs = 3, 21, 32
    img = torch.from_numpy(synthetic_batch(B, 128, seed=6)["img"])
    uvd_gt = torch.rand(B, J, 3, generator=g) * 1.4 - 0.7
    xyz_gt = torch.rand(B, J, 3, generator=g) * 1.4 - 0.7
    results = [torch.randn(B, 5 * J, Fs, Fs, generator=g) * 0.5, torch.randn(B, 5 * J, Fs, Fs, generator=g) * 0.5]
    results += [xyz_gt + torch.randn(B, J, 3, generator=g) * s for s in (0.05, 0.02, 0.004, 0.02)]
    sws = [torch.rand(B, J, Fs, Fs, generator=g), torch.rand(B, J, Fs, Fs, generator=g)]
    chk = sum(float(t.double().abs().sum()) for t in results + sws) + float(img.double().abs().sum())
    assert abs(chk - float(Z["input_checksum"])) < 1e-6 * chk, "the seeded inputs differ from the ones the fixture was generated on"
    assert np.array_equal(uvd_gt.numpy(), Z["uvd_gt"]) and np.array_equal(xyz_gt.numpy(), Z["xyz_gt"])
    return img, uvd_gt, xyz_gt, results, sws


def test_loss_codec_matches_reference_fixtures():
    img, uvd_gt, xyz_gt, results, sws = _inputs()
    pg = TO.joint2offset(uvd_gt, img, 0.8, 32)
    assert np.array_equal(pg.numpy(), Z["pixel_gt"]), "GFM.joint2offset target maps"  # same torch ops in the same order: bit-exact
    assert np.abs(TO.offset2joint_weight(results[0], img, 0.8).numpy() - Z["decode0"]).max() < 1e-6
    assert np.array_equal(TO.joint2heatmap(uvd_gt[:, :, :2], 0.8, 32, sigma=3).numpy()[:, ::5], Z["hm_sigma3"])
    assert np.array_equal(TO.joint2heatmap(uvd_gt[:, :, :2], 0.8, 32, sigma=2).numpy()[:, ::5], Z["hm_sigma2"])
    z = torch.from_numpy(Z["sl1_x"])
    assert abs(float(T.SmoothL1Loss()(z, torch.zeros_like(z))) - float(Z["sl1_mean"])) < 1e-9
    assert abs(float(T.SmoothL1Loss(size_average=False)(z, torch.zeros_like(z))) - float(Z["sl1_sum"])) < 1e-9
    # the knee: quadratic strictly below 0.01, linear from 0.01 on (continuous there)
    one = lambda v: float(T.SmoothL1Loss()(torch.tensor([[v]]), torch.zeros(1, 1)))
    assert abs(one(0.005) - 0.5 * 0.005 ** 2) < 1e-10 and abs(one(0.01) - 0.01 * (0.01 - 0.005)) < 1e-10 and abs(one(-2.0) - 0.01 * 1.995) < 1e-8


def test_loss_schedule_and_first_step_gradients_match_reference():
    img, uvd_gt, xyz_gt, results, sws = _inputs()
    for t in results + sws:
        t.requires_grad_(True)
    loss, parts = TO.kpfusion_loss(results, sws, img, uvd_gt, xyz_gt, epoch=0)
    assert abs(float(loss) - float(Z["loss"])) < 1e-6 * abs(float(Z["loss"]))
    for k, v in parts.items():
        assert abs(float(v) - float(Z[k])) < 1e-6 * max(abs(float(Z[k])), 1e-6), k
    loss.backward()
    for name, t in [("result%d" % i, t) for i, t in enumerate(results)] + [("sw%d" % i, t) for i, t in enumerate(sws)]:
        gn = float(t.grad.double().norm())
        assert abs(gn - float(Z["gradnorm_" + name])) < 1e-5 * gn, name
        assert np.abs(t.grad.reshape(-1)[::97].numpy() - Z["gradsample_" + name]).max() < 1e-7 + 1e-5 * np.abs(Z["gradsample_" + name]).max(), name
    # past the spatial epochs the heat-map terms drop out (train.py:249)
    l2, p2 = TO.kpfusion_loss([r.detach() for r in results], [s.detach() for s in sws], img, uvd_gt, xyz_gt, epoch=25)
    assert "loss_spatial_0" not in p2 and float(l2) < float(loss)


def test_optimizer_setup():
    p = [torch.nn.Parameter(torch.zeros(3))]
    opt, sched = T.make_optimizer(p)
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults["weight_decay"] == 0.01 and opt.defaults["lr"] == 8e-4
    assert sched.step_size == 10 and sched.gamma == 0.1


def test_host_side_of_the_fused_optimizer_and_deferred_gradients_without_a_gpu():
    """FusedAdamW hands anything its kernel does not cover to torch.optim.AdamW (here: CPU parameters — identical steps), and
    DeferredParamGrads declines every tensor when it is not active or the weight is not a whole parameter of the model."""
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ours, ref = T.FusedAdamW(a, lr=8e-4, weight_decay=0.01), torch.optim.AdamW(b, lr=8e-4, weight_decay=0.01)
    for _ in range(3):
        for p, q in zip(a, b):
            g = torch.randn(p.shape)
            p.grad, q.grad = g.clone(), g.clone()
        ours.step(), ref.step()
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    w = torch.nn.Parameter(torch.ones(8))
    assert T.DeferredParamGrads.wants_colsum(w) is None  # (not active)
    x = torch.zeros(4, 8)
    assert T.DeferredParamGrads.wants("k", object(), x, x, 1, 1, 1, 0) is None
    with pytest.raises(TypeError):
        T.DeferredParamGrads(None)  # (the parameter map is mandatory: without it nothing could be verified)
    k = torch.nn.Parameter(torch.ones(8, 8))
    with T.DeferredParamGrads({"w": w, "k": k}) as d:
        assert T.DeferredParamGrads.wants_colsum(w) is d
        assert T.DeferredParamGrads.wants("other", object(), x, x, 1, 1, 1, 0) is None  # a key the map does not know
        k.grad = torch.zeros(8, 8)  # a parameter that already holds a gradient: autograd would ADD the unwritten tensor -> not deferred
        assert T.DeferredParamGrads.wants("k", object(), x, x, 1, 1, 1, 0) is None
        k.grad = None
        w.grad = torch.zeros(8)
        assert T.DeferredParamGrads.wants_colsum(w) is None
        w.grad = None
        assert T.DeferredParamGrads.wants_colsum(torch.ones(8)) is None and T.DeferredParamGrads.wants_colsum(w[:4]) is None  # not / not all of a parameter
        assert T.DeferredParamGrads.wants("k:q", object(), x, x, 1, 1, 1, 0) is None and T.DeferredParamGrads.wants("k", None, x, x, 1, 1, 1, 0) is None
        assert T.DeferredParamGrads.wants("k", object(), x, x, 3, 3, 1, 1) is None and T.DeferredParamGrads.wants("k", object(), x.half(), x.half(), 1, 1, 1, 0) is None
        assert T.DeferredParamGrads.wants("k", object(), x, x, 1, 1, 1, 0) is d
    assert T.DeferredParamGrads.active is None  # (nothing was registered: flush() had nothing to launch)


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from keypointfusion_amd.parallel import GradBucketReducer, shard_batch
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    dead = torch.nn.Parameter(torch.ones(5))  # a parameter that never receives a gradient (the reference has 20 % of those)
    params = list(net.parameters()) + [dead]
    red = GradBucketReducer(params, dist, bucket_mb=0.001)  # ~1 KB buckets: several buckets, reduced while backward still runs
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(12, 16, generator=g), torch.randn(12, 4, generator=g)
    sh = shard_batch({"x": x, "y": y}, rank, world)
    red.reset()
    loss = F.mse_loss(net(sh["x"]), sh["y"], reduction="sum") / x.shape[0]  # global-mean loss: shard sums / global count
    loss.backward()
    red.finish()
    # single-process reference on the full batch
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    F.mse_loss(ref(x), y, reduction="sum").div(x.shape[0]).backward()
    # averaged shard gradients x world == full-batch gradient (each rank's loss is its shard's share of the global mean)
    err = max(float((p.grad * world - q_.grad).abs().max()) for p, q_ in zip(net.parameters(), ref.parameters()))
    q.put((rank, err, len(red.buckets), -1.0 if dead.grad is None else float(dead.grad.abs().max()), red.payload_bytes()))
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, nb, deadg, payload in res:
        assert err < 1e-6, (rank, err)
        # the gradient-less parameter keeps .grad = None (the optimiser then skips it, as behind the reference's DataParallel)
        assert nb >= 3 and deadg == -1.0 and payload == (16 * 32 + 32 + 32 * 8 + 8 + 8 * 4 + 4 + 5) * 4


def test_live_parameters_excludes_the_dead_modules():
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.parallel import live_parameters
    m = KPFusion("KPFusion-resnet-18", "", 21, "dexycb", "")
    total = sum(p.numel() for p in m.parameters())
    live = sum(p.numel() for p in live_parameters(m))
    assert 0.55 * total < live < 0.9 * total, (live, total)  # 28.7 M live of 46.3 M (R18)


@pytest.mark.parametrize("net", ["convnext-tiny", "resnet-18"])
def test_live_parameters_cover_every_parameter_the_reference_gives_a_gradient(net):
    """live_parameters() (what bench.py / GraphedTrainStep hand to the optimiser and to the gradient all-reduce) against the set of
    parameters that received a gradient in the imported reference's own iteration (train_step_*.npz `grad_names`): a parameter missing
    here would silently never train (round 3 found block*.joint_feat_emb / pcl_feat_emb* excluded by a substring match on 'feat_emb')."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.parallel import live_parameters
    m = KPFusion("KPFusion-" + net, "", 21, "dexycb", "")
    live = {id(p) for p in live_parameters(m)}
    names = {n for n, p in m.named_parameters() if id(p) in live}
    ref = set(np.load(os.path.join(GOLDEN, "train_step_%s.npz" % net), allow_pickle=True)["grad_names"].tolist())
    assert ref <= names, sorted(ref - names)[:10]
    # (a superset is harmless — a parameter without a gradient keeps .grad = None and the optimiser / the buckets skip it: the identity
    # skips of equal-width Residual blocks, whose 1x1 convolution exists in the state dict but is never called, model/hourglass.py:106-116)
    assert all(".skip_layer.conv." in n for n in names - ref), sorted(names - ref)[:10]


@pytest.mark.gpu
@pytest.mark.parametrize("net", ["convnext-tiny", "resnet-18"])
def test_train_step_matches_the_reference_loss_and_gradients(net):
    """One iteration of train.py:209-265 through the product: KPFusion in .train() mode (batch-statistics BatchNorm; dropout 0 like the
    fixture), the loss schedule, loss.backward() — against what the imported reference produced on the same seeded weights and batch
    (tests/golden/gen_golden_trainstep.py): the loss and its parts, which parameters receive a gradient, every per-parameter
    gradient norm, updated BatchNorm running statistics; then an AdamW step must lower the loss."""
    from conftest import synthetic_sd
    from keypointfusion_amd.model.model import KPFusion
    Zs = np.load(os.path.join(GOLDEN, "train_step_%s.npz" % net))
    dev = torch.device("cuda:0")
    m = KPFusion("KPFusion-" + net, "", 21, "dexycb", "")
    m.load_state_dict(synthetic_sd("KPFusion-" + net), strict=True)
    m = m.to(dev).train()
    m.train_dropout = 0.0
    B = 3
    b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=11).items()}
    uvd_gt, xyz_gt = torch.from_numpy(Zs["uvd_gt"]).to(dev), torch.from_numpy(Zs["xyz_gt"]).to(dev)

    class Loader:
        img_size, flip = 128, 1

    def step_loss(ball=None):
        m._debug_ball_override = ball
        results, sws, _ = m(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
        m._debug_ball_override = None
        return results, T.kpfusion_loss(results, sws, b["img"], uvd_gt, xyz_gt, epoch=0)

    # ball-query sets are integer decisions taken around network outputs (joints equal to the reference's only to ~1e-6): the comparison
    # runs on the reference's sets, and the product's own sets may differ from them only by a few boundary points
    ball = [torch.from_numpy(Zs["ball_idx"][i].astype(np.int64)) for i in range(6)]
    results, (loss, parts) = step_loss(ball)
    print("train-mode ball-query sets that differ from the reference's: %d" % int(m._debug_ball_flips))
    # observed on the committed fixtures: 0 (ResNet-18) and 0 or 1 (ConvNeXt-T: one point sits on a radius to within the last bits of the
    # train-mode LayerNorm / GELU arithmetic — 1 with the library kernels, 0 with the HIP ones)
    assert int(m._debug_ball_flips) <= {"convnext-tiny": 1, "resnet-18": 0}[net], int(m._debug_ball_flips)
    assert all(r.requires_grad for r in results)
    assert float((results[2].detach().cpu() - torch.from_numpy(Zs["r3d1"])).abs().max()) < 1e-3
    assert float((results[5].detach().cpu() - torch.from_numpy(Zs["r2d2"])).abs().max()) < 1e-3
    # (measured: loss 4e-7 relative, joints 2e-5, gradient norms 6e-5 median; the bounds leave room for the library convolutions'
    # algorithm choice on the strided / 1- and 3-channel layers, which differs between boxes)
    assert abs(float(loss.detach()) - float(Zs["loss"])) < 1e-4 * float(Zs["loss"]), (float(loss.detach()), float(Zs["loss"]))
    for k, v in parts.items():
        assert abs(float(v.detach()) - float(Zs[k])) < 1e-3 * max(float(Zs[k]), 1e-3), (k, float(v.detach()), float(Zs[k]))
    loss.backward()
    ref_norm = dict(zip([str(n) for n in Zs["grad_names"]], Zs["grad_norms"]))
    got = {n: float(p.grad.double().norm()) for n, p in m.named_parameters() if p.grad is not None}
    assert set(got) == set(ref_norm), (sorted(set(got) ^ set(ref_norm))[:10])  # exactly the reference's live parameters receive gradients
    scale = max(ref_norm.values())
    # (biases in front of a BatchNorm have a mathematically zero gradient: both sides hold rounding noise there, hence the floor)
    bad = [(n, got[n], ref_norm[n]) for n in ref_norm if abs(got[n] - ref_norm[n]) > 1e-2 * ref_norm[n] + 5e-6 * scale]
    assert not bad, "%d of %d gradient norms off: %s" % (len(bad), len(ref_norm), bad[:5])
    # per top-level module, tighter
    for top in ("backbone_d", "backbone_rgb", "block1", "block2"):
        a = sum(v * v for n, v in got.items() if n.startswith(top)) ** 0.5
        r = sum(v * v for n, v in ref_norm.items() if n.startswith(top)) ** 0.5
        assert abs(a - r) < 2e-3 * r, (top, a, r)
    sd = m.state_dict()
    for k in [k for k in Zs.files if k.startswith("bn::")]:
        assert float((sd[k[4:]].cpu() - torch.from_numpy(Zs[k])).abs().max()) < 1e-4 * (float(np.abs(Zs[k]).max()) + 1e-3), k
    # an optimiser step on these gradients lowers the loss (same batch)
    from keypointfusion_amd.parallel import live_parameters
    opt, _ = T.make_optimizer(live_parameters(m), lr=2e-4)
    opt.step()
    with torch.no_grad():
        _, (loss2, _) = step_loss()
    assert float(loss2) < float(loss)
    m.eval()  # and the module still serves inference afterwards (weights repacked from the updated parameters)
    with torch.no_grad():
        res_eval, _, _ = m(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
    assert all(bool(torch.isfinite(t).all()) and not t.requires_grad for t in res_eval)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,loss_tol,grad_tol", [("bf16", 3e-2, 0.15)])  # (fp16 training would need loss scaling: refused by the module)
def test_mixed_precision_train_step_tracks_the_fp32_reference(prec, loss_tol, grad_tol):
    """configs[3] trains in bf16: fp32 master weights, GEMM operands rounded to 16 bits (HIP convolutions on the 16-bit MFMA, library ops
    under torch.autocast), fp32 statistics / geometry / loss.  Against the fp32 reference fixture of the same iteration (reference ball-
    query sets injected): loss, the set of gradient-receiving parameters, per top-level-module gradient norms within the stated
    tolerance; gradients are fp32 tensors on the fp32 parameters; an AdamW step lowers the loss."""
    from conftest import synthetic_sd
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.parallel import live_parameters
    net = "convnext-tiny"
    Zs = np.load(os.path.join(GOLDEN, "train_step_%s.npz" % net))
    dev = torch.device("cuda:0")
    m = KPFusion("KPFusion-" + net, "", 21, "dexycb", "")
    m.load_state_dict(synthetic_sd("KPFusion-" + net), strict=True)
    m = m.to(dev).train()
    m.train_dropout, m.precision = 0.0, prec
    b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(3, 128, seed=11).items()}
    uvd_gt, xyz_gt = torch.from_numpy(Zs["uvd_gt"]).to(dev), torch.from_numpy(Zs["xyz_gt"]).to(dev)

    class Loader:
        img_size, flip = 128, 1

    def step_loss(ball=None):
        m._debug_ball_override = ball
        results, sws, _ = m(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
        m._debug_ball_override = None
        assert all(t.dtype == torch.float32 for t in results + sws)
        return T.kpfusion_loss(results, sws, b["img"], uvd_gt, xyz_gt, epoch=0)[0]

    loss = step_loss([torch.from_numpy(Zs["ball_idx"][i].astype(np.int64)) for i in range(6)])
    assert abs(float(loss.detach()) - float(Zs["loss"])) < loss_tol * float(Zs["loss"]), (float(loss.detach()), float(Zs["loss"]))
    loss.backward()
    ref_norm = dict(zip([str(n) for n in Zs["grad_names"]], Zs["grad_norms"]))
    got = {n: float(p.grad.double().norm()) for n, p in m.named_parameters() if p.grad is not None}
    assert set(got) == set(ref_norm)
    assert all(p.grad.dtype == torch.float32 for p in m.parameters() if p.grad is not None)
    for top in ("backbone_d", "backbone_rgb", "block1", "block2"):
        a = sum(v * v for n, v in got.items() if n.startswith(top)) ** 0.5
        r = sum(v * v for n, v in ref_norm.items() if n.startswith(top)) ** 0.5
        print("mixed precision %s: gradient norm of %s %.4f (fp32 reference %.4f)" % (prec, top, a, r))
        assert abs(a - r) < grad_tol * r, (prec, top, a, r)
    opt, _ = T.make_optimizer(live_parameters(m), lr=2e-4)
    opt.step()
    with torch.no_grad():
        assert float(step_loss()) < float(loss.detach())


def _fresh(net, sd):
    from keypointfusion_amd.model.model import KPFusion
    m = KPFusion(net, "", 21, "dexycb", "")
    m.load_state_dict(sd, strict=True)
    return m


def test_loss_schedule_gate_works_on_a_device_scalar_epoch():
    """(the oracle's statement of the schedule; the fused HIP loss is checked against it in test_kernels_train_gpu.py, both gates)
    kpfusion_loss(epoch=<0-d tensor>): the spatial terms are gated by (epoch <= 24) as data, not by a host `if` (which a captured
    hipGraph would freeze) — same numbers as the host-side schedule on both sides of the boundary (train.py:250-261)."""
    g = torch.Generator().manual_seed(3)
    B = 2
    res = [torch.randn(B, 105, 32, 32, generator=g) * 0.1 for _ in range(2)] + [torch.randn(B, 21, 3, generator=g) * 0.3 for _ in range(4)]
    sws = [torch.rand(B, 21, 32, 32, generator=g) for _ in range(2)]
    img = torch.rand(B, 1, 128, 128, generator=g) * 2 - 1
    uvd, xyz = torch.rand(B, 21, 3, generator=g) - 0.5, torch.rand(B, 21, 3, generator=g) - 0.5
    for ep in (0, 24, 25, 40):
        host, _ = TO.kpfusion_loss(res, sws, img, uvd, xyz, epoch=ep)
        devs, _ = TO.kpfusion_loss(res, sws, img, uvd, xyz, epoch=torch.tensor(ep))
        assert torch.allclose(host, devs, rtol=1e-6), (ep, float(host), float(devs))
    assert float(TO.kpfusion_loss(res, sws, img, uvd, xyz, epoch=torch.tensor(25))[0]) < float(TO.kpfusion_loss(res, sws, img, uvd, xyz, epoch=torch.tensor(24))[0])


def test_product_loss_has_no_library_fallback():
    """The product's kpfusion_loss is the fused HIP node or an error: CPU tensors / other schedules do not silently take torch ops."""
    g = torch.Generator().manual_seed(3)
    res = [torch.randn(1, 105, 32, 32, generator=g) for _ in range(2)] + [torch.randn(1, 21, 3, generator=g) for _ in range(4)]
    sws = [torch.rand(1, 21, 32, 32, generator=g) for _ in range(2)]
    with pytest.raises(ValueError, match="fused HIP loss"):
        T.kpfusion_loss(res, sws, torch.zeros(1, 1, 128, 128), torch.zeros(1, 21, 3), torch.zeros(1, 21, 3))


@pytest.mark.gpu
def test_graphed_step_follows_the_lr_schedule():
    """under hipGraph replay a Python-float learning rate is baked into the captured AdamW kernel.  make_optimizer(capturable=True)
    keeps it in a device scalar that StepLR updates in place: after the scheduler crosses step_size the replayed step must shrink 10x."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 4).to(dev)
    opt, sched = T.make_optimizer(lin.parameters(), lr=1e-2, step_size=2, capturable=True)
    assert torch.is_tensor(opt.param_groups[0]["lr"]) and opt.param_groups[0]["lr"].is_cuda
    c = torch.randn(4, 8, device=dev)
    batch = {"x": torch.ones(1, device=dev)}
    step = T.GraphedTrainStep(lin, opt, lambda m, bt: (m.weight * c).sum() * bt["x"].sum() + m.bias.sum(), batch, warmup=1)

    def delta():
        w0 = lin.weight.detach().clone()
        step(batch)
        torch.cuda.synchronize()
        return float((lin.weight.detach() - w0).abs().mean())

    d1 = delta()  # constant gradient: an AdamW step moves every weight by ~lr
    sched.step()
    sched.step()
    assert abs(float(opt.param_groups[0]["lr"]) - 1e-3) < 1e-9
    d2 = delta()
    assert 0.8e-2 < d1 < 1.2e-2 and 0.08 < d2 / d1 < 0.12, (d1, d2)


def _train_fixture(net, B, dev, seed=5):
    from conftest import synthetic_sd
    from keypointfusion_amd.weights import synthetic_batch
    sd = synthetic_sd(net)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=seed).items()}
    g = torch.Generator().manual_seed(1)
    batch["uvd_gt"] = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)
    batch["xyz_gt"] = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)

    class Loader:
        img_size, flip = 128, 1

    def loss_fn(mdl, bt):
        results, sws, _ = mdl(bt["img_rgb"], bt["img"], bt["pcl"], Loader(), bt["center"], bt["M"], bt["cube"], bt["cam_para"], 0.8)
        return T.kpfusion_loss(results, sws, bt["img"], bt["uvd_gt"], bt["xyz_gt"], epoch=0)[0]

    return sd, batch, loss_fn


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_graphed_training_with_the_fused_optimizer_fits_a_fixed_batch(prec):
    """The captured iteration as bench.py runs it (FusedLoss, FusedAdamW, grouped / deferred parameter gradients, two streams) driven for 30
    replays on one batch: the loss must fall steadily and stay finite — a parameter whose deferred gradient were never written, or an
    optimiser state that did not advance, shows up here — and eager iterations (per-layer kernels, no deferral) with the same optimiser
    class from the same start must produce the SAME losses bit for bit: every kernel adds in a fixed order and the grouped forms
    reproduce the per-layer summation order."""
    from keypointfusion_amd.parallel import live_parameters
    dev = torch.device("cuda:0")
    net = "KPFusion-convnext-tiny"
    sd, batch, loss_fn = _train_fixture(net, 4, dev)
    traj = {}
    for mode in ("graphed", "eager"):
        torch.manual_seed(0)
        m = _fresh(net, sd).to(dev).train()
        m.train_dropout = 0.0
        m.precision = prec
        live = live_parameters(m)
        if mode == "graphed":
            opt, _ = T.make_optimizer(live, lr=2e-4, capturable=True)
            assert isinstance(opt, T.FusedAdamW)
            step = T.GraphedTrainStep(m, opt, loss_fn, batch, warmup=2, params=live)  # (the warm-up iterations already step the optimiser)
            traj[mode] = [float(step(batch)) for _ in range(30)]
        else:
            opt, _ = T.make_optimizer(live, lr=2e-4, capturable=True)
            out = []
            for _ in range(6):
                opt.zero_grad(set_to_none=True)
                loss = loss_fn(m, batch)
                loss.backward()
                opt.step()
                out.append(float(loss))
            traj[mode] = out
        del m, opt
    g, e = traj["graphed"], traj["eager"]
    assert all(math.isfinite(v) for v in g), g
    assert g[-1] < 0.6 * g[0] and sum(b < a for a, b in zip(g, g[1:])) >= 24, g
    # graphed replay k is optimiser step k + 2 (two warm-up steps): the overlapping part of the two trajectories
    assert g[:4] == e[2:6], (g[:6], e)


@pytest.mark.gpu
@pytest.mark.parametrize("net,prec", [("KPFusion-resnet-18", "f32"), ("KPFusion-convnext-tiny", "f32"), ("KPFusion-convnext-tiny", "bf16")])
def test_graphed_train_step_replays_are_bit_identical(net, prec):
    """Round-2 defect: with frozen parameters (SGD lr 0) and a fixed batch, two replays of the captured iteration
    returned different losses.  Cause: ResNet's strided convolutions went through the vendor library, whose forward is not run-to-run
    reproducible (1e-5), and the noise flipped a ball-query membership.  Every convolution now runs on the HIP kernels (fixed
    summation order), so N replays must return the SAME BITS: loss and every gradient."""
    from keypointfusion_amd.parallel import live_parameters
    dev = torch.device("cuda:0")
    sd, batch, loss_fn = _train_fixture(net, 4, dev)
    torch.manual_seed(0)
    m = _fresh(net, sd).to(dev).train()
    m.train_dropout = 0.0
    m.precision = prec
    live = live_parameters(m)
    opt = torch.optim.SGD(live, lr=0.0)
    step = T.GraphedTrainStep(m, opt, loss_fn, batch, warmup=1, params=live)
    losses, grads = [], []
    for _ in range(5):
        losses.append(float(step(batch)))
        torch.cuda.synchronize()
        grads.append([None if p.grad is None else p.grad.detach().clone() for p in live])
    assert all(l == losses[0] for l in losses), losses
    names = [n for n, p in m.named_parameters() if any(p is q for q in live)]
    for i in range(1, 5):
        bad = [n for n, a, b in zip(names, grads[0], grads[i]) if a is not None and not torch.equal(a, b)]
        assert not bad, "replay %d: %d gradient tensors differ from replay 0, e.g. %s" % (i, len(bad), bad[:5])


@pytest.mark.gpu
def test_bf16_training_iteration_at_the_stated_batch_of_configs3():
    """configs[3] at its stated batch: one B = 32 bf16 training iteration (tile selection depends on M = B H W, so these are not the B = 3 / 4
    launches the parity fixtures run).  Properties that hold at any size: three replays of the captured iteration give the same bits (loss and
    every gradient), everything is finite, and an EAGER iteration on the same batch and weights (per-layer kernels, no deferral) gives the same
    loss bit for bit.  (Per-sample equality with a smaller batch is not a property of a train-mode forward: BatchNorm uses the batch's
    statistics — model/hourglass.py:106-119 under .train().)"""
    from keypointfusion_amd.parallel import live_parameters
    dev = torch.device("cuda:0")
    net = "KPFusion-convnext-tiny"
    sd, batch, loss_fn = _train_fixture(net, 32, dev, seed=9)
    torch.manual_seed(0)
    m = _fresh(net, sd).to(dev).train()
    m.train_dropout = 0.0
    m.precision = "bf16"
    live = live_parameters(m)
    opt = torch.optim.SGD(live, lr=0.0)  # frozen weights: every iteration sees the same parameters
    for p in m.parameters():
        p.grad = None
    eager = float(loss_fn(m, batch))
    step = T.GraphedTrainStep(m, opt, loss_fn, batch, warmup=1, params=live)
    losses, grads = [], []
    for _ in range(3):
        losses.append(float(step(batch)))
        torch.cuda.synchronize()
        grads.append([None if p.grad is None else p.grad.detach().clone() for p in live])
    assert math.isfinite(losses[0]) and losses[0] == losses[1] == losses[2] == eager, (losses, eager)
    assert sum(g is not None for g in grads[0]) > 300
    for g0, g1, g2 in zip(*grads):
        if g0 is not None:
            assert bool(torch.isfinite(g0).all()) and torch.equal(g0, g1) and torch.equal(g0, g2)


@pytest.mark.gpu
def test_graphed_train_step_with_bucketed_allreduce_equals_the_single_graph():
    """GraphedTrainStep in its data-parallel form (graph A: forward + backward + pack, all-reduce of the flat buckets over RCCL,
    graph B: average + unpack + optimiser) on a one-rank RCCL group must compute exactly what the single-graph form computes: the
    pack / reduce / unpack path moves values, it does not change them — and since every kernel of the iteration adds in a fixed order,
    "exactly" means bit for bit: losses and every gradient tensor.  (world_size 2: the gloo tests; a 1-GPU box cannot host two RCCL ranks.)"""
    import socket
    import torch.distributed as dist
    from keypointfusion_amd.parallel import live_parameters
    dev = torch.device("cuda:0")
    net = "KPFusion-resnet-18"
    sd, batch, loss_fn = _train_fixture(net, 4, dev)

    def run(dist_mod):
        torch.manual_seed(0)
        m = _fresh(net, sd).to(dev).train()
        m.train_dropout = 0.0
        live = live_parameters(m)
        opt = torch.optim.SGD(live, lr=0.0)  # frozen parameters: every replay of either form sees the same weights
        step = T.GraphedTrainStep(m, opt, loss_fn, batch, warmup=1, dist_mod=dist_mod, params=live, **kw)
        losses = [float(step(batch)) for _ in range(3)]
        torch.cuda.synchronize()
        return losses, [None if p.grad is None else p.grad.detach().clone() for p in live], step

    kw = {}
    l1, p1, _ = run(None)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        kw = {"dp_mode": "split"}
        l2, p2, step = run(dist)
        assert step.graph_b is not None and step.payload_bytes() > 40e6 and len(step.buckets) >= 1
        # round 4: ONE graph with the bucket collectives as nodes, launched from gradient hooks during backward (several buckets, so that the
        # hooks really fire mid-backward) — also as reduce-scatter + all-gather
        kw = {"dp_mode": "overlap", "bucket_mb": 8.0}
        l3, p3, step3 = run(dist)
        assert step3.dp_mode == "overlap" and step3.graph_b is None, "the capture with RCCL collectives fell back to the two-graph form"
        assert len(step3._early) >= 4 and len(step3._late) >= 1 and step3.payload_bytes() > 40e6
        kw = {"dp_mode": "overlap", "bucket_mb": 8.0, "collective": "rs_ag"}
        l4, p4, step4 = run(dist)
        assert step4.dp_mode == "overlap"
        kw = {"dp_mode": "overlap", "grad_payload": "bf16"}
        l5, p5, step5 = run(dist)
        assert step5.dp_mode == "overlap" and step5.payload_bytes() < 0.6 * step3.payload_bytes()
    finally:
        dist.destroy_process_group()
    assert l1 == l2 == l3 == l4 and len(set(l1)) == 1, (l1, l2, l3, l4)
    assert sum(g is not None for g in p1) > 100
    for name, pp in (("two-graph", p2), ("overlapped", p3), ("overlapped rs+ag", p4)):
        assert [g is None for g in p1] == [g is None for g in pp]
        bad = [i for i, (a, b) in enumerate(zip(p1, pp)) if a is not None and not torch.equal(a, b)]
        assert not bad, "%d gradient tensors differ between the single-graph and the %s form" % (len(bad), name)
    # bf16 payload: the mean of bf16-rounded gradients (one rank: the rounding itself)
    for a, b in zip(p1, p5):
        if a is not None:
            assert torch.equal(b, a.to(torch.bfloat16).float())


def _graphed_dp_worker(rank, world, port, q, ref_path):
    """One of two processes sharing cuda:0: GraphedTrainStep in its data-parallel form over the gloo backend (RCCL cannot host two
    ranks on one device; gloo reduces CUDA tensors through host memory) on this rank's shard of the batch."""
    try:
        import torch.distributed as dist
        from keypointfusion_amd.parallel import live_parameters, shard_batch
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        net = "KPFusion-resnet-18"
        sd, batch, loss_fn = _train_fixture(net, 4, dev)
        shard = {k: v.contiguous() for k, v in shard_batch(batch, rank, world).items()}
        torch.manual_seed(0)
        m = _fresh(net, sd).to(dev).train()
        m.train_dropout = 0.0
        live = live_parameters(m)
        names = [n for n, p in m.named_parameters() if any(p is x for x in live)]
        opt = torch.optim.SGD(live, lr=0.0)
        step = T.GraphedTrainStep(m, opt, loss_fn, shard, warmup=1, dist_mod=dist, params=live, bucket_mb=16.0)
        losses = [float(step(shard)) for _ in range(2)]
        torch.cuda.synchronize()
        ref = torch.load(ref_path)
        got = {n: p.grad.detach().cpu() for n, p in zip(names, live) if p.grad is not None}
        bad = [n for n in ref["mean"] if n not in got or not torch.equal(got[n], ref["mean"][n])]
        q.put((rank, None, losses, len(step.buckets), len(got), len(bad), bad[:4]))
        dist.destroy_process_group()
    except Exception as e:  # report instead of dying silently: the parent would only see a queue time-out
        import traceback
        q.put((rank, traceback.format_exc()[-3000:], None, 0, 0, 0, []))


@pytest.mark.gpu
def test_graphed_train_step_data_parallel_two_ranks_on_one_gpu(tmp_path):
    """VERDICT r02 missing #2: the product's graphed data-parallel step (graph A -> bucket all-reduce -> graph B) with world_size 2.
    Two processes share cuda:0 and reduce over gloo; each takes half of a 4-image batch.  Every rank must end with the MEAN of the two
    shards' gradients, where the per-shard gradients are what a single process computes eagerly on that shard (per-replica BatchNorm
    statistics, like the reference's DataParallel) — bit for bit, since every kernel of the iteration adds in a fixed order."""
    import torch.multiprocessing as mp
    from keypointfusion_amd.parallel import live_parameters, shard_batch
    dev = torch.device("cuda:0")
    net = "KPFusion-resnet-18"
    sd, batch, loss_fn = _train_fixture(net, 4, dev)
    per_shard, shard_loss = [], []
    for r in range(2):
        torch.manual_seed(0)
        m = _fresh(net, sd).to(dev).train()
        m.train_dropout = 0.0
        live = live_parameters(m)
        names = [n for n, p in m.named_parameters() if any(p is x for x in live)]
        loss = loss_fn(m, {k: v.contiguous() for k, v in shard_batch(batch, r, 2).items()})
        loss.backward()
        shard_loss.append(float(loss))
        per_shard.append({n: p.grad.detach().cpu() for n, p in zip(names, live) if p.grad is not None})
        del m, live, loss
    ref_path = str(tmp_path / "ref.pt")
    torch.save({"mean": {n: (per_shard[0][n] + per_shard[1][n]) / 2 for n in per_shard[0]}}, ref_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_graphed_dp_worker, args=(r, 2, port, q, ref_path)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    for rank, err, losses, nb, ngot, nbad, bad in res:
        assert err is None, "rank %d failed:\n%s" % (rank, err)
        assert nb >= 2, "several buckets expected at 16 MB"
        assert losses == [shard_loss[rank]] * 2, (rank, losses, shard_loss)
        assert ngot == len(per_shard[0]) > 100
        assert nbad == 0, "rank %d: %d gradient tensors are not the mean of the two shards' gradients, e.g. %s" % (rank, nbad, bad)


@pytest.mark.gpu
def test_wide_model_trains_at_256x256():
    """The wide extension (KPFusion(..., crop_size=256)) in .train() mode: one iteration at 256 x 256 — the fused HIP loss agrees with the float64 torch
    restatement of the loss (oracle/train_oracle.py, pinned to the reference's codec at 128) evaluated on the SAME outputs at F = 64, every live parameter
    receives a finite gradient, two iterations on the same batch are bit-identical, and an AdamW step lowers the loss."""
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.parallel import live_parameters
    from keypointfusion_amd.weights import synthetic_state_dict
    net, dev, B = "KPFusion-convnext-tiny", torch.device("cuda:0"), 2
    m = KPFusion(net, "", 21, "dexycb", "", crop_size=256)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(net, 0, crop_size=256).items()}, strict=True)
    m = m.to(dev).train()
    m.train_dropout = 0.0
    b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 256, seed=12).items()}
    g = torch.Generator().manual_seed(2)
    uvd, xyz = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev), (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)

    class Loader:
        img_size, flip = 256, 1

    def iteration():
        for p in m.parameters():
            p.grad = None
        res, sws, _ = m(b["img_rgb"], b["img"], b["pcl"], Loader(), b["center"], b["M"], b["cube"], b["cam_para"], 0.8)
        loss, parts = T.kpfusion_loss(res, sws, b["img"], uvd, xyz, epoch=0)
        loss.backward()
        return res, sws, loss

    live = live_parameters(m)
    stats = {k: v.clone() for k, v in m.named_buffers()}  # (BatchNorm running statistics move in train mode: the second iteration starts from the same)
    res, sws, loss = iteration()
    assert tuple(res[0].shape) == (B, 105, 64, 64) and tuple(sws[0].shape) == (B, 21, 64, 64)
    ref, _ = TO.kpfusion_loss([r.detach().double().cpu() for r in res], [s.detach().double().cpu() for s in sws], b["img"].double().cpu(), uvd.double().cpu(),
                              xyz.double().cpu(), epoch=0)
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref)), (float(loss), float(ref))
    names = {id(p): n for n, p in m.named_parameters()}
    none = [names[id(p)] for p in live if p.grad is None]
    # (the identity skips of Residual blocks with equal widths own a convolution the forward never calls — model/hourglass.py:99-104 — exactly as at 128 x 128)
    assert all(".skip_layer.conv." in n for n in none), [n for n in none if ".skip_layer.conv." not in n][:10]
    live = [p for p in live if p.grad is not None]
    g1 = [p.grad.clone() for p in live]
    assert all(bool(torch.isfinite(x).all()) for x in g1)
    assert sum(float(x.abs().sum()) > 0 for x in g1) > 0.95 * len(g1)
    with torch.no_grad():
        for k, v in m.named_buffers():
            v.copy_(stats[k])
    _, _, loss2 = iteration()
    assert float(loss2) == float(loss) and all(torch.equal(a, p.grad) for a, p in zip(g1, live)), "two iterations on the same batch must give the same bits"
    # (lr: AdamW's first step moves EVERY weight by lr whatever its gradient; from 2e-5 on that is enough to change a ball-query set of block 2 on this batch and
    #  the loss then jumps by +-5 % in either direction — 5e-5: -0.03 with the op-by-op decoder layer, +0.12 with the fused one; 2e-5: +0.07 / +0.10 — while up to
    #  1e-5 every variant of the graph falls by the same 0.08-0.12: round 6, NOTES_r06.md section 1)
    opt, _ = T.make_optimizer(live, lr=3e-6, capturable=False)
    opt.step()
    _, _, loss3 = iteration()
    assert float(loss3) < float(loss) - 0.02, (float(loss3), float(loss))


@pytest.mark.gpu
def test_loss_scaler_checks_unscales_skips_and_adapts_on_the_device():
    """training.LossScaler + FusedAdamW.step(scaler=...) (kpf_grad_finite_check_multi, kpf_adamw_step_multi_scaled, kpf_loss_scale_update): a step on
    gradients of loss * 2^k equals the step on the unscaled gradients bit for bit; one inf anywhere cancels the WHOLE step (parameters, moments and step
    count untouched), halves the scale and is counted; `growth_interval` clean steps double it; nothing is read back to the host in between."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    shapes = [(96, 3, 4, 4), (96,), (384, 96), (5,), (3, 128), (1000001,)]
    mk = lambda: [torch.randn(*sh, generator=g).to(dev).requires_grad_(True) for sh in shapes]
    g = torch.Generator().manual_seed(0)
    pa = mk()
    g = torch.Generator().manual_seed(0)
    pb = mk()
    grads = [torch.randn(*sh, generator=g).to(dev) * 1e-3 for sh in shapes]
    oa, _ = T.make_optimizer(pa, lr=1e-3, capturable=True)
    ob, _ = T.make_optimizer(pb, lr=1e-3, capturable=True)
    sc = T.LossScaler(init_scale=2.0 ** 10, growth_interval=2)
    for it in range(2):
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr * 2.0 ** 10  # what backward of loss * scale leaves
        oa.step()
        ob.step(scaler=sc)
        assert all(torch.equal(p, q) for p, q in zip(pa, pb)), "scaled step differs from the plain one (iteration %d)" % it
    assert sc.get_scale() == 2.0 ** 11 and int(sc.skipped) == 0  # two clean steps at growth_interval 2
    assert float(ob.state[pb[0]]["step"]) == 2.0
    before = [q.detach().clone() for q in pb]
    m_before = [ob.state[q]["exp_avg"].clone() for q in pb]
    for q, gr in zip(pb, grads):
        q.grad = gr * 2.0 ** 11
    pb[4].grad[1, 77] = float("inf")
    ob.step(scaler=sc)
    assert all(torch.equal(a, q) for a, q in zip(before, pb)) and all(torch.equal(a, ob.state[q]["exp_avg"]) for a, q in zip(m_before, pb))
    assert sc.get_scale() == 2.0 ** 10 and int(sc.skipped) == 1 and float(ob.state[pb[0]]["step"]) == 2.0 and int(sc.found) == 0
    pb[4].grad[1, 77] = float("nan")
    ob.step(scaler=sc)
    assert sc.get_scale() == 2.0 ** 9 and int(sc.skipped) == 2
    sd = sc.state_dict()
    sc2 = T.LossScaler(device=dev)
    addr = (sc2.scale_t.data_ptr(), sc2.inv_scale.data_ptr(), sc2.tracker.data_ptr(), sc2.found.data_ptr(), sc2.skipped.data_ptr())
    sc2.load_state_dict(dict(sd, growth_tracker=1))
    assert sc2.get_scale() == 2.0 ** 9 and float(sc2.inv_scale) == 2.0 ** -9 and int(sc2.tracker) == 1
    # restored IN PLACE: a graph captured on these tensors keeps seeing them
    assert addr == (sc2.scale_t.data_ptr(), sc2.inv_scale.data_ptr(), sc2.tracker.data_ptr(), sc2.found.data_ptr(), sc2.skipped.data_ptr())
    sc3 = T.LossScaler()  # no device yet: the tracker waits for the first use
    sc3.load_state_dict({"scale": 2.0 ** 7, "growth_tracker": 3})
    assert sc3.state_dict() == {"scale": 2.0 ** 7, "growth_tracker": 3}
    sc3.scale(torch.ones((), device=dev))
    assert sc3.get_scale() == 2.0 ** 7 and int(sc3.tracker) == 3
    with pytest.raises(RuntimeError, match="cannot move"):
        sc3._to(torch.device("cpu"))


@pytest.mark.gpu
def test_fp16_training_with_loss_scaling_follows_the_fp32_gradients_and_replays_from_a_graph():
    """precision "f16" in train mode (VERDICT r04 missing #4): fp16 GEMM operands / activations, fp32 master weights, the loss scaled by training.LossScaler.
    (a) the unscaled gradients are as close to the fp32 step's as the bf16 step's are (a structural error would not be); (b) the captured iteration
    (GraphedTrainStep(scaler=...)) lowers the loss over a few replays without a skipped step and keeps its scale on the device."""
    from conftest import synthetic_sd
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.parallel import live_parameters
    net, dev, B = "KPFusion-convnext-tiny", torch.device("cuda:0"), 4
    b = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=13).items()}
    g = torch.Generator().manual_seed(3)
    b["uvd_gt"], b["xyz_gt"] = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev), (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)

    class Loader:
        img_size, flip = 128, 1

    def loss_of(m, bt):
        r, s, _ = m(bt["img_rgb"], bt["img"], bt["pcl"], Loader(), bt["center"], bt["M"], bt["cube"], bt["cam_para"], 0.8)
        return T.kpfusion_loss(r, s, bt["img"], bt["uvd_gt"], bt["xyz_gt"], epoch=0)[0]

    def grads(prec, scale):
        m = KPFusion(net, "", 21, "dexycb", "")
        m.load_state_dict(synthetic_sd(net), strict=True)
        m = m.to(dev).train()
        m.train_dropout, m.precision = 0.0, prec
        loss = loss_of(m, b)
        (loss * scale).backward()
        return float(loss), {n: p.grad.detach().float() / scale for n, p in m.named_parameters() if p.grad is not None}

    l32, g32 = grads("f32", 1.0)
    lb, gb = grads("bf16", 1.0)
    lh, gh = grads("f16", 2.0 ** 12)
    assert set(gh) == set(g32) and all(bool(torch.isfinite(v).all()) for v in gh.values())
    nrm = lambda d: torch.stack([v.norm() for v in d.values()])
    dist = lambda d: float(torch.stack([(d[k] - g32[k]).norm() for k in g32]).norm() / nrm(g32).norm())
    print("gradient distance from the fp32 step: bf16 %.3e, f16 (loss x 2^12) %.3e; losses %.6f / %.6f / %.6f" % (dist(gb), dist(gh), l32, lb, lh))
    assert abs(lh - l32) < 2e-2 * abs(l32) and dist(gh) < max(1.5 * dist(gb), 0.05)

    m = KPFusion(net, "", 21, "dexycb", "")
    m.load_state_dict(synthetic_sd(net), strict=True)
    m = m.to(dev).train()
    m.precision = "f16"
    live = live_parameters(m)
    opt, _ = T.make_optimizer(live, lr=1e-4, capturable=True)
    sc = T.LossScaler(init_scale=2.0 ** 12, growth_interval=4)
    step = T.GraphedTrainStep(m, opt, loss_of, b, scaler=sc)
    losses = [float(step(b)) for _ in range(8)]
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
    assert int(sc.skipped) == 0 and sc.get_scale() >= 2.0 ** 12  # (clean steps: the scale only grew)
    # resume on a LIVE captured step: the restored scale and tracker are the ones the next replays run on and update
    sc.load_state_dict({"scale": 2.0 ** 8, "growth_tracker": 3})
    float(step(b))
    assert sc.get_scale() == 2.0 ** 9 and int(sc.tracker) == 0, (sc.get_scale(), int(sc.tracker))  # tracker 3 + one clean step = growth_interval 4: doubled
