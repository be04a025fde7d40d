"""CPU tests of the host side: C-ABI library exports, data-parallel sharding over gloo (world_size 2), synthetic data."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, synthetic_sd
from keypointfusion_amd import lib as L
from keypointfusion_amd.parallel import shard_bounds, shard_batch, gather_outputs, max_over_ranks
from keypointfusion_amd.weights import synthetic_batch


def test_library_loads_and_exports_every_declared_symbol():
    """include/kpf.h is the contract: every function it declares must be exported by libkpf_hip.so (no compute calls)."""
    hdr = open(os.path.join(ROOT, "include", "kpf.h")).read()
    declared = set(re.findall(r"\b(kpf_[a-z0-9_]+)\s*\(", hdr)) - {"kpf_conv_desc", "kpf_pack_desc", "kpf_adamw_desc", "kpf_wgrad_group_desc", "kpf_colsum_desc"}
    assert len(declared) >= 19
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libkpf_hip.so does not export %s" % name
    assert set(L.EXPORTS) == declared, (sorted(declared - set(L.EXPORTS)), sorted(set(L.EXPORTS) - declared))
    l = L.load()
    m = re.search(r"#define KPF_ABI_VERSION (\d+)", hdr)
    assert m and l.kpf_abi_version() == int(m.group(1)) == L.ABI_VERSION  # header, library and binding describe the same interface
    assert l.kpf_tr_encoder_weight_floats(128) == 128 * 128 + 128 + 21 * 128 + 4 * (128 * 384 + 384 + 128 * 128 + 128 * 6 + 128 * 16 + 16 + 16 * 128) + 128 * 3 + 3 + 128 * 3 + 3


def test_library_of_another_abi_version_is_refused(monkeypatch):
    """A stale libkpf_hip.so (entry points whose arguments changed meaning under the same names) must not load silently."""
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "ABI_VERSION", L.ABI_VERSION + 1)
    with pytest.raises(L.KpfError, match="ABI version"):
        L.load()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.KpfError):
        L.load()


def test_model_refuses_cpu_tensors():
    from keypointfusion_amd.model.model import KPFusion
    m = KPFusion("KPFusion-resnet-18", "", 21, "dexycb", "")
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(1, 128).items()}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(b["img_rgb"], b["img"], b["pcl"], None, b["center"], b["M"], b["cube"], b["cam_para"])
    with pytest.raises(KeyError):
        KPFusion("KPFusion-convnext-T", "", 21, "dexycb", "")  # the reference's own failure for this spelling


def test_shard_bounds_partition():
    for n in (1, 2, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import kpf_oracle as O
        sd = synthetic_sd("KPFusion-resnet-18")
        full = {k: torch.from_numpy(v) for k, v in synthetic_batch(n, 64, seed=3).items()}
        mine = shard_batch(full, rank, world)
        # the per-rank "model" here is the CPU oracle's backbone: the sharding/gather logic is what is under test
        od, fd, orgb, frgb = O.backbones_forward(sd, mine["img_rgb"], mine["img"])
        g = gather_outputs([od, orgb], n, dist)
        t = max_over_ranks(1.0 + rank, torch.device("cpu"), dist)
        if rank == 0:
            ref = O.backbones_forward(sd, full["img_rgb"], full["img"])
            q.put((float((g[0] - ref[0]).abs().max()), float((g[1] - ref[2]).abs().max()), tuple(g[0].shape), t))
    finally:
        dist.destroy_process_group()


def test_data_parallel_shards_and_gather_gloo_world2():
    """N>1 path on CPU: 2 ranks, ragged split of 3 samples, gathered outputs equal the unsharded forward."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    e0, e1, shape, t = q.get(timeout=10)
    assert shape == (3, 105, 16, 16)
    assert e0 < 1e-5 and e1 < 1e-5  # batch-size dependent conv algorithms may differ in the last bits
    assert t == 2.0


def test_synthetic_batch_is_geometrically_consistent():
    b = synthetic_batch(2, 128, seed=1)
    assert b["img"].shape == (2, 1, 128, 128) and b["pcl"].shape == (2, 1024, 3)
    assert float(np.abs(b["pcl"]).max()) < 1.0  # back-projected foreground pixels fall inside the cube
    fg = b["img"] < 0.99
    assert 0.35 < fg.mean() < 0.55 and float(b["img"][~fg].min()) == 1.0


def test_modules_survive_deepcopy_and_pickle():
    """The modules hold per-device caches (locks, packed weights, captured graphs): copy.deepcopy / pickle (EMA copies, torch.save of
    a whole module) must drop them and keep parameters, buffers and settings."""
    import copy
    import io
    import torch
    from keypointfusion_amd.model.model import KPFusion
    m = KPFusion("KPFusion-resnet-18", "", 21, "dexycb", "")
    m.precision = "bf16"
    for clone in (copy.deepcopy(m), torch.load(io.BytesIO((lambda b: (torch.save(m, b), b.getvalue())[1])(io.BytesIO())), weights_only=False)):
        assert clone.precision == "bf16" and clone._plans == {} and clone._plan_lock is not m._plan_lock
        sd, sc = m.state_dict(), clone.state_dict()
        assert list(sd) == list(sc) and all(torch.equal(sd[k], sc[k]) for k in sd)


def test_dropin_model_package_resolves_the_reference_import_lines(tmp_path):
    """dropin/model replaces the reference's model/ directory: with it first on the path, the reference's own import statements
    (train.py:5-6 `from model.model import KPFusion`; cbam / hourglass / mano_head likewise) give the HIP classes."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from model.model import KPFusion; from model.cbam import CBAM; from model.hourglass import PoseNet; from model.mano_head import mano_regHead;"
            "import keypointfusion_amd.model.model as M; assert KPFusion is M.KPFusion; print('ok')") % (ROOT, os.path.join(ROOT, "dropin"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


def test_importing_the_inference_path_leaves_the_environment_alone(tmp_path):
    """a drop-in library must not change the HIP runtime's behaviour for its host application.  Importing the package and the whole
    inference path (model, engine, engine16, serving, heads) leaves os.environ as it was; only the TRAINING module makes the one documented process-wide
    setting (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, graphs.prepare_training_graphs), keeps a value the host exported, and can be told to keep its hands off."""
    import subprocess
    import sys
    var = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
    env = {k: v for k, v in os.environ.items() if k not in (var, "KPF_KEEP_HIP_ENV")}
    code = ("import os, sys; sys.path.insert(0, %r); before = dict(os.environ);"
            "import keypointfusion_amd; from keypointfusion_amd.model.model import KPFusion; from keypointfusion_amd import engine, engine16, serving, heads, lib;"
            "from keypointfusion_amd.model import cbam, hourglass, mano_head;"
            "assert dict(os.environ) == before, set(os.environ) ^ set(before);"
            "from keypointfusion_amd import training;"
            "print(os.environ.get(%r))") % (ROOT, var)
    run = lambda e: subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), env=e)
    out = run(env)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stderr[-2000:]
    out = run(dict(env, **{var: "1"}))  # the host application's own choice is kept
    assert out.returncode == 0 and out.stdout.strip() == "1", out.stderr[-2000:]
    out = run(dict(env, KPF_KEEP_HIP_ENV="1"))
    assert out.returncode == 0 and out.stdout.strip() == "None", out.stderr[-2000:]


def test_wide_extension_changes_two_tensors_and_nothing_else():
    """crop_size=256 (the labelled wide extension): same keys in the same order as the reference's state dict, only `fc_spatial2joint_feature.weight` of the
    two blocks resized to (crop_size / 4)^2 inputs; crop_size=128 IS the reference's spec; sizes the architecture cannot take are refused."""
    import pytest
    import torch
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.spec import kpfusion_spec
    ref, wide = kpfusion_spec("KPFusion-resnet-18"), kpfusion_spec("KPFusion-resnet-18", 256)
    assert kpfusion_spec("KPFusion-resnet-18", 128) == ref and [r[0] for r in ref] == [w[0] for w in wide]
    diff = [(r[0], r[1], w[1]) for r, w in zip(ref, wide) if r != w]
    assert diff == [("block1.fc_spatial2joint_feature.weight", (1, 1024), (1, 4096)), ("block2.fc_spatial2joint_feature.weight", (1, 1024), (1, 4096))]
    for bad in (100, 32, 250):
        with pytest.raises(ValueError):
            kpfusion_spec("KPFusion-resnet-18", bad)
    m = KPFusion("KPFusion-resnet-18", "", 21, "dexycb", "", crop_size=256)
    assert m.crop_size == 256 and tuple(m.state_dict()["block2.fc_spatial2joint_feature.weight"].shape) == (1, 4096)
    # a reference-sized checkpoint loads into everything but those two tensors (train.py:100-107 loads by key intersection and shape errors are the caller's:
    # here the mismatch is reported by load_state_dict itself)
    sd = KPFusion("KPFusion-resnet-18", "", 21, "dexycb", "").state_dict()
    with pytest.raises(RuntimeError, match="fc_spatial2joint_feature"):
        m.load_state_dict(sd, strict=True)
    keep = {k: v for k, v in sd.items() if "fc_spatial2joint_feature.weight" not in k}
    missing, unexpected = m.load_state_dict(keep, strict=False)
    assert sorted(missing) == ["block1.fc_spatial2joint_feature.weight", "block2.fc_spatial2joint_feature.weight"] and not unexpected


def test_layernorm_fold_rule_depends_on_the_layer_not_on_the_batch():
    """kpf_conv2d_h16_ln_fold_supported (host-side rule, no GPU): pwconv1 of ConvNeXt-B / -T stages 3-4 takes the folded LayerNorm at every batch size — which
    arithmetic a sample gets must not depend on how many samples share its launch — and layers the eight-phase GEMM does not cover never do."""
    import ctypes as C
    from keypointfusion_amd import lib as L
    lib = L.load()

    def desc(B, HW, Cin, N, flags, Kp=None, k=1):
        d = L.ConvDesc()
        d.B, d.IH, d.IW, d.Cin, d.in_ld, d.in_coff = B, HW, HW, Cin, Cin, 0
        d.OH, d.OW, d.N = HW, HW, N
        d.KH = d.KW = k
        d.sh = d.sw = 1
        d.ph = d.pw = k // 2
        d.Kp = Kp or k * k * Cin
        d.out_ld, d.out_coff, d.flags = N, 0, flags
        return d

    for Cin in (512, 1024, 384, 768):  # ConvNeXt-B stages 3-4, ConvNeXt-T stages 3-4
        got = {B: lib.kpf_conv2d_h16_ln_fold_supported(C.byref(desc(B, 32, Cin, 4 * Cin, L.KPF_ACT_GELU))) for B in (1, 2, 7, 64)}
        assert set(got.values()) == {1}, (Cin, got)
    assert lib.kpf_conv2d_h16_ln_fold_supported(C.byref(desc(8, 32, 512, 2048, 0))) == 0                       # no GELU: not pwconv1
    assert lib.kpf_conv2d_h16_ln_fold_supported(C.byref(desc(8, 32, 512, 2048, L.KPF_ACT_GELU | L.KPF_RES_ADD))) == 0
    assert lib.kpf_conv2d_h16_ln_fold_supported(C.byref(desc(8, 32, 96, 384, L.KPF_ACT_GELU, Kp=128))) == 0     # Cin % 64 != 0 / N % 256 != 0 (ConvNeXt-T stage 1)
    assert lib.kpf_conv2d_h16_ln_fold_supported(C.byref(desc(8, 32, 192, 768, L.KPF_ACT_GELU))) == 0            # Kp % 128 != 0 (stage 2)
    assert lib.kpf_conv2d_h16_ln_fold_supported(C.byref(desc(8, 32, 512, 2048, L.KPF_ACT_GELU, k=3))) == 0      # not a dense 1x1
