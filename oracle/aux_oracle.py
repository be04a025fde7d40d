"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatements (fp32, torch-CPU ATen ops) of the three stand-alone heads the north star names but KPFusion.forward never
calls: CBAM, the stacked-hourglass PoseNet and the MANO regression head.  Same rules as
oracle/kpf_oracle.py: pure functions of a state dict with the reference's key names, every function cites the reference lines
it follows, only tests/ and __graft_entry__.smoke() may import this module.

Pinning: tests/golden/gen_golden_aux.py imports the reference's own modules in the build container (model/cbam.py and
model/hourglass.py import as they are; model/mano_head.py + util/manopth/manopth/manolayer.py run over a *synthetic* hand model
because the MANO pickle needs chumpy and is licence-restricted — only the file loader is stubbed, see
tests/golden/ref_import.py::load_reference_mano_head), asserts these functions reproduce the reference's outputs, then writes
tests/golden/aux_*.npz which `pytest -m "not gpu"` re-checks.
"""
import torch
import torch.nn.functional as F

from .kpf_oracle import _bn, residual

OBMAN2MANO = (0, 5, 6, 7, 9, 10, 11, 17, 18, 19, 13, 14, 15, 1, 2, 3, 8, 12, 20, 16, 4)  # model/mano_head.py:7-16
MANO_PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)  # manolayer.py:197-202 level lists == kintree_table[0]
MANO_TIPS_RIGHT = (745, 317, 444, 556, 673)  # manolayer.py:251-252
MANO_JOINT_ORDER = (0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20)  # manolayer.py:261


# --------------------------------------------------------------------------------------------------------------
# a17: CBAM
# --------------------------------------------------------------------------------------------------------------
def cbam_channel_scale(sd, x, p="ChannelGate"):
    """model/cbam.py:26-57 — sigmoid(mlp(avgpool) + mlp(maxpool)) as a B x C gate."""
    def mlp(v):
        h = F.relu(F.linear(v, sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"]))
        return F.linear(h, sd[p + ".mlp.3.weight"], sd[p + ".mlp.3.bias"])
    avg = x.mean(dim=(2, 3))
    mx = x.amax(dim=(2, 3))
    return torch.sigmoid(mlp(avg) + mlp(mx))


def cbam_forward(sd, x, no_spatial=False):
    """model/cbam.py:84-94.  Returns x*scale when no_spatial, else the SpatialGate *tuple* (x_out*s, x_out*(1-s)) (:82)."""
    scale = cbam_channel_scale(sd, x)
    x_out = x * scale[:, :, None, None]
    if no_spatial:
        return x_out
    comp = torch.cat([x_out.amax(1, keepdim=True), x_out.mean(1, keepdim=True)], 1)  # ChannelPool, :65-67
    s = F.conv2d(comp, sd["SpatialGate.spatial.conv.weight"], None, padding=3)
    s = torch.sigmoid(_bn(sd, "SpatialGate.spatial.bn", s))
    return x_out * s, x_out * (1 - s)


# --------------------------------------------------------------------------------------------------------------
# a18: Hourglass / PoseNet
# --------------------------------------------------------------------------------------------------------------
def hourglass(sd, p, x, n):
    """model/hourglass.py:122-149."""
    up1 = residual(sd, p + ".up1", x)
    low = residual(sd, p + ".low1", F.max_pool2d(x, 2, 2))
    low = hourglass(sd, p + ".low2", low, n - 1) if n > 1 else residual(sd, p + ".low2", low)
    low = residual(sd, p + ".low3", low)
    return up1 + F.interpolate(low, scale_factor=2, mode="nearest")


def posenet_forward(sd, img, nstack):
    """model/hourglass.py:211-229 — returns (preds B x 5J x S/4 x S/4, feature B x inp_dim x S/4 x S/4) of the last stack."""
    x = F.conv2d(img, sd["pre.0.conv.weight"], sd["pre.0.conv.bias"], stride=2, padding=3)
    x = F.relu(_bn(sd, "pre.0.bn", x))
    x = residual(sd, "pre.1", x)
    x = F.max_pool2d(x, 2, 2)
    x = residual(sd, "pre.3", x)
    x = residual(sd, "pre.4", x)
    preds = feat = None
    for i in range(nstack):
        hg = hourglass(sd, "hgs.%d" % i, x, 4)
        feat = residual(sd, "features.%d.0" % i, hg)
        feat = F.conv2d(feat, sd["features.%d.1.conv.weight" % i], sd["features.%d.1.conv.bias" % i])
        feat = F.relu(_bn(sd, "features.%d.1.bn" % i, feat))
        preds = torch.cat([F.conv2d(feat, sd["outs_%d.%d.weight" % (k, i)], sd["outs_%d.%d.bias" % (k, i)]) for k in (1, 2, 3)], 1)
        if i < nstack - 1:
            x = x + F.conv2d(preds, sd["merge_preds.%d.conv.conv.weight" % i], sd["merge_preds.%d.conv.conv.bias" % i]) \
                  + F.conv2d(feat, sd["merge_features.%d.conv.conv.weight" % i], sd["merge_features.%d.conv.conv.bias" % i])
    return preds, feat


# --------------------------------------------------------------------------------------------------------------
# a19: MANO regression head
# --------------------------------------------------------------------------------------------------------------
def rot6d_to_mat(x6):
    """model/mano_head.py:144-153 — Gram-Schmidt on two 3-vectors; b1,b2,b3 are the *columns* of the result."""
    a1, a2 = x6[:, 0:3], x6[:, 3:6]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def mat_to_quat(R, eps=1e-6):
    """model/mano_head.py:84-141 (kornia's branchy rotation-matrix -> quaternion, on the transposed matrix)."""
    t = R.transpose(1, 2)
    m00, m11, m22 = t[:, 0, 0], t[:, 1, 1], t[:, 2, 2]
    d2 = m22 < eps
    d01 = m00 > m11
    d0n1 = m00 < -m11
    t0 = 1 + m00 - m11 - m22
    q0 = torch.stack([t[:, 1, 2] - t[:, 2, 1], t0, t[:, 0, 1] + t[:, 1, 0], t[:, 2, 0] + t[:, 0, 2]], -1)
    t1 = 1 - m00 + m11 - m22
    q1 = torch.stack([t[:, 2, 0] - t[:, 0, 2], t[:, 0, 1] + t[:, 1, 0], t1, t[:, 1, 2] + t[:, 2, 1]], -1)
    t2 = 1 - m00 - m11 + m22
    q2 = torch.stack([t[:, 0, 1] - t[:, 1, 0], t[:, 2, 0] + t[:, 0, 2], t[:, 1, 2] + t[:, 2, 1], t2], -1)
    t3 = 1 + m00 + m11 + m22
    q3 = torch.stack([t3, t[:, 1, 2] - t[:, 2, 1], t[:, 2, 0] - t[:, 0, 2], t[:, 0, 1] - t[:, 1, 0]], -1)
    c0 = (d2 & d01).float()[:, None]
    c1 = (d2 & ~d01).float()[:, None]
    c2 = (~d2 & d0n1).float()[:, None]
    c3 = (~d2 & ~d0n1).float()[:, None]
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0[:, None] * c0 + t1[:, None] * c1 + t2[:, None] * c2 + t3[:, None] * c3)
    return q * 0.5


def quat_to_aa(q):
    """model/mano_head.py:49-81."""
    q1, q2, q3 = q[:, 1], q[:, 2], q[:, 3]
    s2 = q1 * q1 + q2 * q2 + q3 * q3
    s = torch.sqrt(s2)
    c = q[:, 0]
    two_theta = 2.0 * torch.where(c < 0.0, torch.atan2(-s, -c), torch.atan2(s, c))
    k = torch.where(s2 > 0.0, two_theta / s, torch.full_like(s, 2.0))
    aa = torch.stack([q1 * k, q2 * k, q3 * k], -1)
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)  # mat2aa, :171-173


def rodrigues(aa):
    """util/manopth/manopth/rodrigues_layer.py:16-57 — axis-angle -> rotation matrix through a normalised half-angle quaternion."""
    ang = torch.norm(aa + 1e-8, p=2, dim=1, keepdim=True)
    n = aa / ang
    h = ang * 0.5
    q = torch.cat([torch.cos(h), torch.sin(h) * n], 1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], 1).view(-1, 3, 3)


def mano_layer(sd, pose_aa, betas, p="mano_layer"):
    """util/manopth/manopth/manolayer.py:106-273 with use_pca=False, joint_rot_mode='axisang', flat_hand_mean=True, side='right',
    center_idx=None, th_trans absent: pose_aa B x 48, betas B x 10 -> (verts B x 778 x 3, joints B x 21 x 3) in millimetres."""
    B = pose_aa.shape[0]
    full = torch.cat([pose_aa[:, :3], sd[p + ".th_hands_mean"] + pose_aa[:, 3:48]], 1)
    R = rodrigues(full.reshape(-1, 3)).view(B, 16, 3, 3)
    pose_map = (R[:, 1:] - torch.eye(3)).reshape(B, 135)
    v_shaped = torch.matmul(sd[p + ".th_shapedirs"], betas.t()).permute(2, 0, 1) + sd[p + ".th_v_template"]
    Jr = torch.matmul(sd[p + ".th_J_regressor"], v_shaped)  # B x 16 x 3
    v_posed = v_shaped + torch.matmul(sd[p + ".th_posedirs"], pose_map.t()).permute(2, 0, 1)
    G = [None] * 16
    for j in range(16):
        par = MANO_PARENTS[j]
        rel = Jr[:, j] if par < 0 else Jr[:, j] - Jr[:, par]
        loc = torch.cat([torch.cat([R[:, j], rel[:, :, None]], 2), torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(B, 1, 4)], 1)
        G[j] = loc if par < 0 else torch.matmul(G[par], loc)
    G = torch.stack(G, 1)  # B x 16 x 4 x 4 (global joint transforms)
    # remove the rest pose: translation -= R @ J  (the homogeneous coordinate of J is 0 at :231-233)
    Jh = torch.cat([Jr, torch.zeros(B, 16, 1)], 2)
    corr = torch.matmul(G, Jh[..., None])
    G2 = G - torch.cat([torch.zeros(B, 16, 4, 3), corr], 3)
    T = torch.matmul(G2.permute(0, 2, 3, 1), sd[p + ".th_weights"].t())  # B x 4 x 4 x 778
    rest = torch.cat([v_posed.transpose(2, 1), torch.ones(B, 1, v_posed.shape[1])], 1)
    verts = (T * rest[:, None]).sum(2).transpose(2, 1)[:, :, :3]
    jtr = torch.cat([G[:, :, :3, 3], verts[:, list(MANO_TIPS_RIGHT)]], 1)[:, list(MANO_JOINT_ORDER)]
    return verts * 1000, jtr * 1000


def mano_head_forward(sd, features):
    """model/mano_head.py:208-225 — dict with verts3d, joints3d (OBMAN2MANO order), mano_shape, mano_pose (rotmat), mano_pose_aa."""
    h = features
    i = 0
    while "mano_base_layer.%d.weight" % i in sd:
        h = F.leaky_relu(F.linear(h, sd["mano_base_layer.%d.weight" % i], sd["mano_base_layer.%d.bias" % i]))
        i += 2
    pose6d = F.linear(h, sd["pose_reg.weight"], sd["pose_reg.bias"])
    shape = F.linear(h, sd["shape_reg.weight"], sd["shape_reg.bias"])
    R = rot6d_to_mat(pose6d.reshape(-1, 6))
    aa = quat_to_aa(mat_to_quat(R)).reshape(-1, 48)
    verts, joints = mano_layer(sd, aa, shape)
    return {"verts3d": verts, "joints3d": joints[:, list(OBMAN2MANO)], "mano_shape": shape, "mano_pose": R.view(-1, 16, 3, 3),
            "mano_pose_aa": aa}
