"""Deterministic synthetic weights and synthetic RGB-D batches.

There is no checkpoint and no dataset in this environment, so every parity test and the
benchmark use weights drawn here: one numpy PCG64 stream per state-dict key (seeded by (seed, crc32(key)),
so a tensor does not depend on enumeration order), scaled so activations stay O(1) through ~60 layers, with
non-trivial BatchNorm running statistics and a ConvNeXt layer-scale large enough that the MLP branch is
visible in the outputs.  The same generator feeds the imported reference when golden vectors are made
(tests/golden/gen_golden.py) and the HIP model on the GPU box.

Synthetic inputs follow SURVEY.md §8d: rgb ~ U[0,1); depth = background 1.0 with a centred disc of
U[-0.6,0.6]; pcl = 1024 foreground pixels back-projected like dataloader/loader.py:775-789 does;
center=(0,0,600)+U[-30,30]^3 mm; cube=250 mm; M = [[s,0,tx],[0,s,ty],[0,0,1]] chosen so that the cube's
projection fills the crop (geometrically consistent, unlike §8d's free s/tx/ty); DexYCB-like intrinsics.
"""
import hashlib
import math
import zlib

import numpy as np

from .spec import kpfusion_spec


def _draw(rng, shape, init):
    n = int(np.prod(shape)) if len(shape) else 1
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1

    def normal(std):
        return (rng.standard_normal(n) * std).astype(np.float32).reshape(shape)

    def uniform(lo, hi):
        return (rng.random(n) * (hi - lo) + lo).astype(np.float32).reshape(shape)

    if init in ("conv", "pw", "tr"):
        return normal(1.0 / math.sqrt(fan_in))
    if init == "final":  # small heads keep the decoded joints inside the point cloud (so ball queries are populated)
        return normal(0.3 / math.sqrt(fan_in))
    if init == "dwconv":
        return normal(1.0 / 7.0)
    if init == "linear":
        return normal(1.0 / math.sqrt(fan_in))
    if init == "bias":
        return normal(0.1)
    if init == "norm_w":
        return uniform(0.8, 1.2)
    if init == "norm_b":
        return normal(0.1)
    if init == "bn_mean":
        return normal(0.2)
    if init == "bn_var":
        return uniform(0.6, 1.4)
    if init == "gamma":
        return uniform(0.05, 0.2)
    if init == "emb":
        return normal(0.5)
    if init == "head3":
        return normal(0.3 / math.sqrt(fan_in))
    if init == "fc_spatial":
        return normal(1.0 / 32.0)
    if init == "weight_dis":
        return np.full(shape, 0.3, np.float32)
    if init == "dead":
        return normal(0.02)
    if init == "shape_reg":
        return normal(1.0 / math.sqrt(fan_in))
    if init == "zero_i64":
        return np.zeros(shape, np.int64)
    raise ValueError(init)


def synthetic_state_dict(net, seed=0, crop_size=128):
    """dict key -> numpy array, for every key of the reference's state dict (crop_size != 128: the wide extension, spec.kpfusion_spec)."""
    out = {}
    for name, shape, dtype, init in kpfusion_spec(net, crop_size):
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        out[name] = _draw(rng, shape, init)
    return out


def synthetic_from_spec(spec, seed=0, prefix=""):
    """Same per-key generator for any (name, shape, dtype, init) list (CBAM / PoseNet / MANO-head specs of spec.py);
    `prefix` namespaces the crc32 seed so equal key names of different modules draw different values."""
    out = {}
    for name, shape, dtype, init in spec:
        if init == "mano":
            continue
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32((prefix + name).encode())]))
        out[name] = _draw(rng, shape, init)
    return out


MANO_PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)  # kintree_table[0] of MANO_{LEFT,RIGHT}.pkl


def synthetic_mano_model(seed=0):
    """A random hand model with the *shapes and structure* of MANO_RIGHT.pkl (778 vertices, 16 joints, 10 shape and 135 pose
    blend shapes, row-stochastic joint regressor and skinning weights, the MANO kinematic tree).  The MANO data itself is
    licence-restricted and is neither shipped nor needed for parity: the MANO layer is linear-blend skinning over whatever
    arrays it is given (util/manopth/manopth/manolayer.py:69-104)."""
    rng = np.random.Generator(np.random.PCG64([seed, 778]))
    V, Jn = 778, 16
    v = (rng.random((V, 3)) - 0.5) * np.array([0.10, 0.18, 0.03])  # metres, hand-sized slab
    shapedirs = rng.standard_normal((V, 3, 10)) * 0.004
    posedirs = rng.standard_normal((V, 3, 135)) * 0.0015
    jreg = np.zeros((Jn, V))
    for j in range(Jn):
        idx = rng.choice(V, 96, replace=False)
        w = rng.random(96)
        jreg[j, idx] = w / w.sum()
    weights = np.zeros((V, Jn))
    for i in range(V):
        idx = rng.choice(Jn, 4, replace=False)
        w = rng.random(4) ** 2
        weights[i, idx] = w / w.sum()
    faces = rng.integers(0, V, (1538, 3))
    comps = rng.standard_normal((45, 45)) * 0.3
    mean = rng.standard_normal(45) * 0.2
    kin = np.stack([np.array([4294967295 if p < 0 else p for p in MANO_PARENTS], np.int64), np.arange(16, dtype=np.int64)])
    return dict(v_template=v, shapedirs=shapedirs, posedirs=posedirs, J_regressor=jreg, weights=weights, f=faces.astype(np.uint32),
                hands_components=comps, hands_mean=mean, kintree_table=kin)


def mano_layer_buffers(model):
    """MANO arrays -> the ManoLayer buffers of the reference (manolayer.py:69-104; use_pca=False, flat_hand_mean=True), fp32."""
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32))  # noqa: E731
    return {
        "mano_layer.th_betas": np.zeros((1, 10), np.float32),
        "mano_layer.th_shapedirs": f32(model["shapedirs"]),
        "mano_layer.th_posedirs": f32(model["posedirs"]),
        "mano_layer.th_v_template": f32(model["v_template"])[None],
        "mano_layer.th_J_regressor": f32(model["J_regressor"]),
        "mano_layer.th_weights": f32(model["weights"]),
        "mano_layer.th_faces": np.asarray(model["f"]).astype(np.int32).astype(np.int64),
        "mano_layer.th_hands_mean": np.zeros((1, 45), np.float32),
        "mano_layer.th_selected_comps": f32(np.asarray(model["hands_components"])[:6]),
    }


def synthetic_mano_head_state(seed=0, feature_size=1024, mano_neurons=(1024, 512)):
    from .spec import mano_head_spec
    spec = mano_head_spec(feature_size, mano_neurons)
    sd = mano_layer_buffers(synthetic_mano_model(seed))
    sd.update(synthetic_from_spec(spec, seed, prefix="mano_head."))
    return {name: sd[name] for name, _, _, _ in spec}


def state_dict_digest(sd, keys=None):
    """sha256 over the float tensors (sorted by key) — recorded in fixtures to detect generator drift."""
    h = hashlib.sha256()
    for k in sorted(sd if keys is None else keys):
        a = sd[k]
        a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        h.update(k.encode())
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _inv3(M):
    return np.linalg.inv(M.astype(np.float64))


def synthetic_batch(B, S=128, seed=1, n_points=1024, img_size=None):
    """Returns dict of numpy arrays: img_rgb B3SS, img B1SS, pcl BN3, center B3, M B33, cube B3, cam_para B4."""
    rng = np.random.Generator(np.random.PCG64([seed, B, S]))
    img_size = S if img_size is None else img_size
    rgb = rng.random((B, 3, S, S)).astype(np.float32)
    yy, xx = np.mgrid[0:S, 0:S]
    r2 = (xx + 0.5 - S / 2) ** 2 + (yy + 0.5 - S / 2) ** 2
    fg = r2 < (0.45 * S * S / math.pi)  # disc covering ~45 % of the crop
    depth = np.ones((B, 1, S, S), np.float32)
    vals = (rng.random((B, S, S)) * 1.2 - 0.6).astype(np.float32)
    depth[:, 0][:, fg] = vals[:, fg]
    center = (np.array([0, 0, 600.0]) + (rng.random((B, 3)) * 60 - 30)).astype(np.float32)
    cube = np.full((B, 3), 250.0, np.float32)
    cam = np.tile(np.array([906.96, 906.79, 956.75, 547.23], np.float32), (B, 1))
    # crop affine consistent with the cube: the cube's projection fills the crop (dataloader/loader.py:604-750 behaviour)
    s = img_size / (cube[:, 0] * cam[:, 0] / center[:, 2]) * (rng.random(B) * 0.2 + 0.9)
    uc = cam[:, 2] + center[:, 0] * cam[:, 0] / center[:, 2]
    vc = cam[:, 3] + center[:, 1] * cam[:, 1] / center[:, 2]
    tx = img_size / 2 - s * uc + (rng.random(B) * 8 - 4)
    ty = img_size / 2 - s * vc + (rng.random(B) * 8 - 4)
    M = np.zeros((B, 3, 3), np.float32)
    M[:, 0, 0] = s
    M[:, 1, 1] = s
    M[:, 0, 2] = tx
    M[:, 1, 2] = ty
    M[:, 2, 2] = 1
    # point cloud: sample foreground pixels, back-project (uvd normalised -> xyz normalised)
    fy, fx = np.nonzero(fg)
    pcl = np.zeros((B, n_points, 3), np.float32)
    for b in range(B):
        sel = rng.integers(0, fy.size, n_points)
        py, px = fy[sel], fx[sel]
        u = (px + 0.5) / S * 2 - 1
        v = (py + 0.5) / S * 2 - 1
        d = depth[b, 0, py, px]
        uvp = np.stack([(u + 1) * (img_size / 2), (v + 1) * (img_size / 2), np.ones_like(u)], 0)
        Mi = _inv3(M[b])
        uw = Mi[0] @ uvp
        vw = Mi[1] @ uvp
        dmm = d * cube[b, 2] / 2 + center[b, 2]
        x = (uw - cam[b, 2]) * dmm / cam[b, 0]
        y = (vw - cam[b, 3]) * dmm / cam[b, 1]
        xyz = np.stack([x, y, dmm], 1)
        pcl[b] = np.clip((xyz - center[b]) / (cube[b] / 2), -1, 1)
    return dict(img_rgb=rgb, img=depth, pcl=pcl, center=center, M=M, cube=cube, cam_para=cam)


def synthetic_tensor(shape, seed, tag, lo=None, hi=None):
    """Seeded fp32 test tensor (standard normal, or uniform [lo, hi)) — inputs of the stand-alone head fixtures."""
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(tag.encode())]))
    n = int(np.prod(shape))
    a = rng.standard_normal(n) if lo is None else rng.random(n) * (hi - lo) + lo
    return a.astype(np.float32).reshape(shape)
