"""Bit-exact 3x3 inverse: the reference inverts the crop matrix with torch.linalg.inv on the CPU
(dataloader/loader.py:781); pixel positions, and through them the integer top-4 indices, depend on its last bit."""
import numpy as np
import pytest
import torch

from keypointfusion_amd.weights import synthetic_batch
from keypointfusion_amd import inv3x3 as I


def test_restated_order_equals_torch_linalg_inv_bitwise():
    """The host library takes one of the two pinned rounding orders (fused on Intel, separately rounded on AMD), and the restatement
    of that order reproduces torch.linalg.inv bit for bit — crop matrices, general matrices (every pivot pattern), synthetic crops."""
    mode = I.probe_host()
    assert mode in (0, 1), "this host's torch.linalg.inv follows neither pinned order (the device then uses a fixed one: no bit parity here)"
    assert I.host_mode() == mode
    Ms = I.crop_matrices(1500, seed=3)
    ref = torch.linalg.inv(torch.from_numpy(Ms).view(-1, 1, 3, 3)).view(-1, 3, 3).numpy()
    for m, r in zip(Ms, ref):
        assert np.array_equal(I.inv3x3(m, mode), r), (m, I.inv3x3(m, mode), r)
    rng = np.random.default_rng(5)
    G = rng.normal(size=(300, 3, 3)).astype(np.float32)
    ref = torch.linalg.inv(torch.from_numpy(G)).numpy()
    assert all(np.array_equal(I.inv3x3(m, mode), r) for m, r in zip(G, ref))
    b = synthetic_batch(8, 128, seed=2)
    ref = torch.linalg.inv(torch.from_numpy(b["M"])).numpy()
    assert all(np.array_equal(I.inv3x3(m, mode), r) for m, r in zip(b["M"], ref))
    assert not all(np.array_equal(I.inv3x3(m, 1 - mode), r) for m, r in zip(Ms, torch.linalg.inv(torch.from_numpy(Ms)).numpy())), \
        "the two orders must be distinguishable on crop matrices"


@pytest.mark.gpu
def test_hip_inverse_equals_torch_linalg_inv_bitwise():
    from keypointfusion_amd import engine as E, lib as L
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    Ms = np.concatenate([I.crop_matrices(10000, seed=4), rng.normal(size=(2000, 3, 3)).astype(np.float32)])
    M = torch.from_numpy(Ms)
    ref = torch.linalg.inv(M.view(-1, 1, 3, 3)).view(-1, 3, 3)  # CPU, the reference's call (a B x 1 x 3 x 3 batch)
    got = E.crop_inverse(M.to(dev)).cpu()  # what every forward uses: the device routine in the host library's rounding order
    bad = (got != ref).view(len(Ms), -1).any(1)
    assert not bool(bad.any()), "%d of %d inverses differ from torch.linalg.inv in some bit" % (int(bad.sum()), len(Ms))
    for fused in (0, 1):  # both device variants against their scalar restatements (whatever the host is)
        out = torch.empty(64, 3, 3, device=dev)
        L.check(L.load().kpf_inv3x3_f32(E._ptr(M[:64].to(dev)), E._ptr(out), 64, fused, E._stream()), "kpf_inv3x3_f32")
        want = np.stack([I.inv3x3(m, fused) for m in Ms[:64]])
        assert np.array_equal(out.cpu().numpy(), want)


def test_unknown_host_library_is_reported_and_the_environment_pins_an_order(monkeypatch):
    """Third branch: a host whose torch.linalg.inv follows neither pinned order.  host_mode() says so (-1, one
    warning): eager forwards then invert on the host (exact), captured ones use the fixed device order (engine.crop_inverse)."""
    monkeypatch.setattr(I, "_mode", None)
    monkeypatch.setattr(I, "probe_host", lambda: -1)
    monkeypatch.delenv("KPF_INV3X3_MODE", raising=False)
    with pytest.warns(UserWarning, match="neither known 3x3 rounding order"):
        assert I.host_mode() == -1
    monkeypatch.setattr(I, "_mode", None)
    monkeypatch.setenv("KPF_INV3X3_MODE", "1")
    assert I.host_mode() == 1  # pinned by the environment whatever the host
    monkeypatch.setattr(I, "_mode", None)
    monkeypatch.setenv("KPF_INV3X3_MODE", "fused")
    with pytest.raises(ValueError, match="KPF_INV3X3_MODE"):  # (a raised error, not an assert that `python -O` strips)
        I.host_mode()


@pytest.mark.gpu
def test_crop_inverse_is_capturable_on_an_unknown_host(monkeypatch):
    """The fallback order runs on the device like the pinned ones: crop_inverse inside a hipGraph capture, results = the scalar
    restatement of that order."""
    from keypointfusion_amd import engine as E
    monkeypatch.setattr(I, "_mode", None)
    monkeypatch.setattr(I, "probe_host", lambda: -1)
    monkeypatch.delenv("KPF_INV3X3_MODE", raising=False)
    dev = torch.device("cuda:0")
    Ms = I.crop_matrices(32, seed=8)
    M = torch.from_numpy(Ms).to(dev)
    with pytest.warns(UserWarning):
        eager = E.crop_inverse(M)
    # eager on such a host: the reference's own call on the host, bit for bit
    assert np.array_equal(eager.cpu().numpy(), torch.linalg.inv(torch.from_numpy(Ms).view(-1, 1, 3, 3)).view(-1, 3, 3).numpy())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = E.crop_inverse(M)
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), np.stack([I.inv3x3(m, 0) for m in Ms]))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 5, 7, 11])
def test_pixel_positions_and_top4_indices_equal_the_oracle_bitwise(seed):
    """Identical raw inputs in, identical integers out: pixel positions bit for bit, top-4 index tensor exactly (a9)."""
    from keypointfusion_amd import engine as E, lib as L
    from oracle import kpf_oracle as O
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    B, N = 4, 1024
    b = {k: torch.from_numpy(v) for k, v in synthetic_batch(B, 128, seed=seed).items()}
    d = {k: v.to(dev) for k, v in b.items()}
    clos = torch.empty(B, N, 4, device=dev)
    idx = torch.empty(B, N, 4, device=dev, dtype=torch.int32)
    ixyz = torch.empty(B, 1024, 3, device=dev)
    L.check(L.load().kpf_img2pcl_top4_f32(E._ptr(d["pcl"]), E._ptr(d["img"]), E._ptr(d["center"]), E._ptr(E.crop_inverse(d["M"])), E._ptr(d["cube"]),
                                          E._ptr(d["cam_para"]), E._ptr(clos), E._ptr(idx), E._ptr(ixyz), B, N, 128, 32, 128, 1, E._stream()))
    torch.cuda.synchronize()
    img_down = F.interpolate(b["img"], size=[32, 32])
    ref_xyz = O.img_xyz_grid(img_down, b["center"], b["M"], b["cube"], b["cam_para"])
    assert torch.equal(ixyz.cpu(), ref_xyz), "pixel positions differ from the oracle in %d values" % int((ixyz.cpu() != ref_xyz).sum())
    ref_c, ref_i = O.img2pcl_index(b["pcl"], img_down, b["center"], b["M"], b["cube"], b["cam_para"])
    got = idx.cpu().long()
    if not torch.equal(got, ref_i):  # exact distance ties may be ordered differently: then the distances must be identical
        dist = torch.sum(torch.pow(b["pcl"].unsqueeze(2) - ref_xyz.unsqueeze(1), 2), dim=-1)
        assert torch.equal(torch.gather(dist, 2, got), torch.gather(dist, 2, ref_i))
        assert int((got != ref_i).any(-1).sum()) <= 2, "more than exact-tie reorderings"
    assert float((clos.cpu() - ref_c).abs().max()) < 1e-6
