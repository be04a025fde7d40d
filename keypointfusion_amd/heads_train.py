"""Train-mode forward of the stand-alone heads (CBAM model/cbam.py:26-94, stacked-hourglass PoseNet model/hourglass.py:64-229,
MANO regression head model/mano_head.py:177-225) — autograd-connected outputs on the module's own Parameters, like KPFusion's train graph: every convolution /
Linear runs forward, data- and weight-gradient on the HIP GEMM kernels (training.conv2d_nhwc / linear_hip), BatchNorm uses batch statistics and updates its
running estimates on the HIP kernels (training.batchnorm_relu_rows; F.batch_norm for CBAM's one-channel norm, which the row kernels' quad width does not
cover), activations stay NHWC between them; pooling, nearest up-sampling, gates and the MANO layer's rigid-body algebra (Rodrigues, blend shapes, linear-blend
skinning: a few hundred small batched products) are torch ops under autograd.  The reference's heads are ordinary autograd modules (model/cbam.py:84-94) that
KPFusion.forward never calls; these forwards exist so that a caller who wires them into a training loop gets gradients instead of a
NotImplementedError.  Eval mode keeps the inference plans of heads.py."""
import torch
import torch.nn.functional as F

from .training import batchnorm_relu_rows, conv2d_nhwc, linear_hip


def _params(module):
    t = dict(module.named_parameters())
    t.update(dict(module.named_buffers()))
    return t


def _bn(t, p, x, relu, momentum=0.1):
    """BatchNorm2d with batch statistics (+ ReLU) on NHWC; running estimates and the batch counter updated like nn.BatchNorm2d.train()."""
    C = x.shape[-1]
    t[p + ".num_batches_tracked"].add_(1)
    if C % 4:
        y = F.batch_norm(x.permute(0, 3, 1, 2), t[p + ".running_mean"], t[p + ".running_var"], t[p + ".weight"], t[p + ".bias"], True, momentum, 1e-5).permute(0, 2, 3, 1)
        return F.relu(y) if relu else y
    return batchnorm_relu_rows(x.reshape(-1, C).contiguous(), t[p + ".weight"], t[p + ".bias"], t[p + ".running_mean"], t[p + ".running_var"], momentum, 1e-5, relu).view(x.shape)


def _conv(t, p, x, stride=1, pad=0, res=None, bias=True):
    """nn.Conv2d on NHWC through the HIP implicit GEMM (forward, data gradient, weight gradient); 1- / 2- / 3-channel inputs get zero channels up to 4."""
    w = t[p + ".weight"]
    cpad = (-w.shape[1]) % 4
    if cpad:
        x, w = F.pad(x, (0, cpad)), F.pad(w, (0, 0, 0, 0, 0, cpad))
    return conv2d_nhwc(x.contiguous(), w, t[p + ".bias"] if bias and (p + ".bias") in t else None, stride, pad, "f32", None, None, None, 1, res)


def _residual(t, p, x):
    """model/hourglass.py:84-119: pre-activation bottleneck, conv3 + skip in the GEMM's residual epilogue."""
    out = _conv(t, p + ".conv1.conv", _bn(t, p + ".bn1", x, True))
    out = _conv(t, p + ".conv2.conv", _bn(t, p + ".bn2", out, True), pad=1)
    out = _bn(t, p + ".bn3", out, True)
    skip = _conv(t, p + ".skip_layer.conv", x) if x.shape[-1] != t[p + ".conv3.conv.weight"].shape[0] else x
    return _conv(t, p + ".conv3.conv", out, res=skip.contiguous())


def _pool2(x):  # nn.MaxPool2d(2, 2) on NHWC (channels-last view: no layout copy)
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def _hourglass(t, p, x, n):
    """model/hourglass.py:122-149."""
    up1 = _residual(t, p + ".up1", x)
    low = _residual(t, p + ".low1", _pool2(x))
    low = _hourglass(t, p + ".low2", low, n - 1) if n > 1 else _residual(t, p + ".low2", low)
    low = _residual(t, p + ".low3", low)
    return up1 + F.interpolate(low.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)


def posenet_train_forward(module, img):
    """model/hourglass.py:211-229 in train mode: (preds [B, 5J, S/4, S/4], feature [B, inp_dim, S/4, S/4]) of the last stack, NCHW like the reference."""
    t = _params(module)
    x = img.float().permute(0, 2, 3, 1)
    x = _bn(t, "pre.0.bn", _conv(t, "pre.0.conv", x, stride=2, pad=3), True)
    x = _residual(t, "pre.1", x)
    x = _residual(t, "pre.3", _pool2(x))
    x = _residual(t, "pre.4", x)
    preds = feat = None
    for i in range(module.nstack):
        hg = _hourglass(t, "hgs.%d" % i, x, 4)
        feat = _residual(t, "features.%d.0" % i, hg)
        feat = _bn(t, "features.%d.1.bn" % i, _conv(t, "features.%d.1.conv" % i, feat), True)
        preds = torch.cat([_conv(t, "outs_%d.%d" % (k, i), feat) for k in (1, 2, 3)], -1)
        if i < module.nstack - 1:
            x = x + _conv(t, "merge_preds.%d.conv.conv" % i, preds) + _conv(t, "merge_features.%d.conv.conv" % i, feat)
    return preds.permute(0, 3, 1, 2), feat.permute(0, 3, 1, 2)


def cbam_train_forward(module, x):
    """model/cbam.py:84-94 in train mode: x * scale for no_spatial, else the SpatialGate tuple (x_out * s, x_out * (1 - s)) (model/cbam.py:82); NCHW in / out."""
    t = _params(module)
    xn = x.float().permute(0, 2, 3, 1)  # NHWC view
    mlp = lambda v: linear_hip(F.relu(linear_hip(v.contiguous(), t["ChannelGate.mlp.1.weight"], t["ChannelGate.mlp.1.bias"])), t["ChannelGate.mlp.3.weight"],
                               t["ChannelGate.mlp.3.bias"])
    scale = torch.sigmoid(mlp(xn.mean(dim=(1, 2))) + mlp(xn.amax(dim=(1, 2))))
    x_out = xn * scale[:, None, None, :]
    if module.no_spatial:
        return x_out.permute(0, 3, 1, 2)
    comp = torch.stack([x_out.amax(-1), x_out.mean(-1)], -1)  # ChannelPool (model/cbam.py:65-67): [B, H, W, 2]
    s = torch.sigmoid(_bn(t, "SpatialGate.spatial.bn", _conv(t, "SpatialGate.spatial.conv", comp, pad=3, bias=False), False, momentum=0.01))  # model/cbam.py:11: momentum 0.01
    return (x_out * s).permute(0, 3, 1, 2), (x_out * (1 - s)).permute(0, 3, 1, 2)


# ---- MANO head: MLP on the HIP GEMM, the hand model's algebra under autograd (model/mano_head.py:49-173, util/manopth/manopth/manolayer.py:106-273) ----
_OBMAN2MANO = (0, 5, 6, 7, 9, 10, 11, 17, 18, 19, 13, 14, 15, 1, 2, 3, 8, 12, 20, 16, 4)
_PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)
_TIPS = (745, 317, 444, 556, 673)
_JOINT_ORDER = (0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20)


def _rot6d_to_mat(x6):
    a1, a2 = x6[:, 0:3], x6[:, 3:6]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=1)), dim=-1)


def _mat_to_aa(R, eps=1e-6):
    """rotation matrix -> quaternion (the branch selection of model/mano_head.py:84-141, on the transposed matrix) -> axis-angle (:49-81)."""
    m = R.transpose(1, 2)
    m00, m11, m22 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    d2, d01, d0n1 = m22 < eps, m00 > m11, m00 < -m11
    t0, t1, t2, t3 = 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22, 1 + m00 + m11 + m22
    q0 = torch.stack([m[:, 1, 2] - m[:, 2, 1], t0, m[:, 0, 1] + m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2]], -1)
    q1 = torch.stack([m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] + m[:, 1, 0], t1, m[:, 1, 2] + m[:, 2, 1]], -1)
    q2 = torch.stack([m[:, 0, 1] - m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2], m[:, 1, 2] + m[:, 2, 1], t2], -1)
    q3 = torch.stack([t3, m[:, 1, 2] - m[:, 2, 1], m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] - m[:, 1, 0]], -1)
    c0, c1, c2, c3 = [c.float()[:, None] for c in (d2 & d01, d2 & ~d01, ~d2 & d0n1, ~d2 & ~d0n1)]
    q = (q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3) / torch.sqrt(t0[:, None] * c0 + t1[:, None] * c1 + t2[:, None] * c2 + t3[:, None] * c3) * 0.5
    s2 = (q[:, 1:] * q[:, 1:]).sum(-1)
    s, c = torch.sqrt(s2), q[:, 0]
    two_theta = 2.0 * torch.where(c < 0.0, torch.atan2(-s, -c), torch.atan2(s, c))
    k = torch.where(s2 > 0.0, two_theta / s, torch.full_like(s, 2.0))
    aa = q[:, 1:] * k[:, None]
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)


def _rodrigues(aa):
    ang = torch.norm(aa + 1e-8, p=2, dim=1, keepdim=True)
    h = ang * 0.5
    q = torch.cat([torch.cos(h), torch.sin(h) * (aa / ang)], 1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
                        2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
                        2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z], 1).view(-1, 3, 3)


def _mano_layer(t, pose_aa, betas, p="mano_layer"):
    B, dev = pose_aa.shape[0], pose_aa.device
    full = torch.cat([pose_aa[:, :3], t[p + ".th_hands_mean"] + pose_aa[:, 3:48]], 1)
    R = _rodrigues(full.reshape(-1, 3)).view(B, 16, 3, 3)
    pose_map = (R[:, 1:] - torch.eye(3, device=dev)).reshape(B, 135)
    v_shaped = torch.matmul(t[p + ".th_shapedirs"], betas.t()).permute(2, 0, 1) + t[p + ".th_v_template"]
    Jr = torch.matmul(t[p + ".th_J_regressor"], v_shaped)
    v_posed = v_shaped + torch.matmul(t[p + ".th_posedirs"], pose_map.t()).permute(2, 0, 1)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev).expand(B, 1, 4)
    G = [None] * 16
    for j, par in enumerate(_PARENTS):
        rel = Jr[:, j] if par < 0 else Jr[:, j] - Jr[:, par]
        loc = torch.cat([torch.cat([R[:, j], rel[:, :, None]], 2), bottom], 1)
        G[j] = loc if par < 0 else torch.matmul(G[par], loc)
    G = torch.stack(G, 1)
    corr = torch.matmul(G, torch.cat([Jr, torch.zeros(B, 16, 1, device=dev)], 2)[..., None])
    G2 = G - torch.cat([torch.zeros(B, 16, 4, 3, device=dev), corr], 3)
    T = torch.matmul(G2.permute(0, 2, 3, 1), t[p + ".th_weights"].t())
    rest = torch.cat([v_posed.transpose(2, 1), torch.ones(B, 1, v_posed.shape[1], device=dev)], 1)
    verts = (T * rest[:, None]).sum(2).transpose(2, 1)[:, :, :3]
    jtr = torch.cat([G[:, :, :3, 3], verts[:, list(_TIPS)]], 1)[:, list(_JOINT_ORDER)]
    return verts * 1000, jtr * 1000


def mano_head_train_forward(module, features):
    """model/mano_head.py:208-225 in train mode: the reference's result dict, autograd-connected to the MLP's Parameters."""
    t = _params(module)
    h = features.float()
    i = 0
    while "mano_base_layer.%d.weight" % i in t:
        h = F.leaky_relu(linear_hip(h.contiguous(), t["mano_base_layer.%d.weight" % i], t["mano_base_layer.%d.bias" % i]))
        i += 2
    pose6d = linear_hip(h.contiguous(), t["pose_reg.weight"], t["pose_reg.bias"])
    shape = linear_hip(h.contiguous(), t["shape_reg.weight"], t["shape_reg.bias"])
    R = _rot6d_to_mat(pose6d.reshape(-1, 6))
    aa = _mat_to_aa(R).reshape(-1, 48)
    verts, joints = _mano_layer(t, aa, shape)
    return {"verts3d": verts, "joints3d": joints[:, list(_OBMAN2MANO)], "mano_shape": shape, "mano_pose": R.view(-1, 16, 3, 3), "mano_pose_aa": aa}
