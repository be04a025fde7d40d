"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement (fp32, torch-CPU ATen ops, no autograd, no GPU) of the KeypointFusion forward hot path, written
as pure functions of a state dict that uses the reference's key names.  It exists so that the HIP kernels can be
checked on the GPU box, where /root/reference cannot travel.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; keypointfusion_amd/ never does.

Why torch-CPU and not C/numpy: the path is floating-point convolution/GEMM work and the thing being restated *is*
the reference's PyTorch-CPU forward (BASELINE.json north_star: "match the reference PyTorch-CPU forward"), so ATen's
fp32 CPU kernels are the reference arithmetic itself; the integer/index sub-ops (ball query, top-4) are additionally
restated in plain C in oracle/kpf_index_oracle.c (built by `make -C oracle`, checked in tests/test_index_oracle.py).

Pinning: the reference has no tests or golden vectors for this path, so the oracle is pinned against
*outputs of the reference itself run in the build container*: tests/golden/gen_golden.py imports the reference
(shims in tests/golden/ref_import.py), loads the same synthetic weights, and asserts oracle == reference before it
writes the fixtures under tests/golden/ that `pytest -m "not gpu"` re-checks the oracle against.
One sub-op stays "parity unpinned": pointnet2_ops.ball_query (third-party CUDA extension, pointnet2_ops==3.0.0,
requirements.txt:15, absent from /root/reference and from this image) — restated from its published semantics.

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
import math

import torch
import torch.nn.functional as F

J = 21


# --------------------------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------------------------
def _bn(sd, p, x, eps=1e-5):
    """nn.BatchNorm{1,2}d in eval mode (running statistics)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def layernorm_cf(x, w, b, eps=1e-6):
    """convNeXT/convnext.py:209-214 — LayerNorm over dim 1 of NCHW, biased variance."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def residual(sd, p, x):
    """model/hourglass.py:87-119 — pre-activation bottleneck; conv = model/hourglass.py:64-84 (bias, 'same' pad)."""
    cin = x.shape[1]
    out = F.relu(_bn(sd, p + ".bn1", x))
    out = F.conv2d(out, sd[p + ".conv1.conv.weight"], sd[p + ".conv1.conv.bias"])
    out = F.relu(_bn(sd, p + ".bn2", out))
    out = F.conv2d(out, sd[p + ".conv2.conv.weight"], sd[p + ".conv2.conv.bias"], padding=1)
    out = F.relu(_bn(sd, p + ".bn3", out))
    out = F.conv2d(out, sd[p + ".conv3.conv.weight"], sd[p + ".conv3.conv.bias"])
    cout = out.shape[1]
    if cin != cout:
        x = F.conv2d(x, sd[p + ".skip_layer.conv.weight"], sd[p + ".skip_layer.conv.bias"])
    return out + x


def upsample2x(x):
    """nn.Upsample(scale_factor=2, mode='bilinear') — align_corners False (convNeXT/resnetUnet.py:76)."""
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


# --------------------------------------------------------------------------------------------------------------
# backbones
# --------------------------------------------------------------------------------------------------------------
def convnext_block(sd, p, x):
    """convNeXT/convnext.py:39-52."""
    c = x.shape[1]
    y = F.conv2d(x, sd[p + ".dwconv.weight"], sd[p + ".dwconv.bias"], padding=3, groups=c)
    y = y.permute(0, 2, 3, 1)
    y = F.layer_norm(y, (c,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    y = F.linear(y, sd[p + ".pwconv1.weight"], sd[p + ".pwconv1.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[p + ".pwconv2.weight"], sd[p + ".pwconv2.bias"])
    y = sd[p + ".gamma"] * y
    return x + y.permute(0, 3, 1, 2)


def convnext_features(sd, p, x):
    """convNeXT/convnext.py:111-117 with the 4x4/s4 stem of convNeXT/resnetUnet.py:105-109."""
    feats = []
    i = 0
    while (p + ".downsample_layers.%d.0.weight" % i) in sd:
        q = p + ".downsample_layers.%d" % i
        if i == 0:
            x = F.conv2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"], stride=4)
            x = layernorm_cf(x, sd[q + ".1.weight"], sd[q + ".1.bias"])
        else:
            x = layernorm_cf(x, sd[q + ".0.weight"], sd[q + ".0.bias"])
            x = F.conv2d(x, sd[q + ".1.weight"], sd[q + ".1.bias"], stride=2)
        j = 0
        while (p + ".stages.%d.%d.gamma" % (i, j)) in sd:
            x = convnext_block(sd, p + ".stages.%d.%d" % (i, j), x)
            j += 1
        feats.append(x)
        i += 1
    return feats


def resnet_features(sd, p, x):
    """model/resnet.py:232-244 (stem + maxpool + 4 stages of BasicBlock model/resnet.py:59-76 or Bottleneck :113-135)."""
    x = F.conv2d(x, sd[p + ".conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(sd, p + ".bn1", x))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li in range(1, 5):
        j = 0
        while (p + ".layer%d.%d.conv1.weight" % (li, j)) in sd:
            q = p + ".layer%d.%d" % (li, j)
            stride = 2 if (li > 1 and j == 0) else 1
            idt = x
            if (q + ".conv3.weight") in sd:  # Bottleneck: 1x1 -> 3x3 (carries the stride) -> 1x1 (x4)
                out = F.relu(_bn(sd, q + ".bn1", F.conv2d(x, sd[q + ".conv1.weight"], None)))
                out = F.relu(_bn(sd, q + ".bn2", F.conv2d(out, sd[q + ".conv2.weight"], None, stride=stride, padding=1)))
                out = _bn(sd, q + ".bn3", F.conv2d(out, sd[q + ".conv3.weight"], None))
            else:
                out = F.conv2d(x, sd[q + ".conv1.weight"], None, stride=stride, padding=1)
                out = F.relu(_bn(sd, q + ".bn1", out))
                out = F.conv2d(out, sd[q + ".conv2.weight"], None, padding=1)
                out = _bn(sd, q + ".bn2", out)
            if (q + ".downsample.0.weight") in sd:
                idt = _bn(sd, q + ".downsample.1", F.conv2d(x, sd[q + ".downsample.0.weight"], None, stride=stride))
            x = F.relu(out + idt)
            j += 1
        feats.append(x)
    return feats


def unet(sd, p, img):
    """convNeXT/resnetUnet.py:129-152 / :224-248 and model/resnetUnet.py:309-330 / :393-414.
    Returns (img_result B x 105 x F x F, img_feature B x 128 x F x F)."""
    convnext = (p + ".backbone.downsample_layers.0.0.weight") in sd
    c1, c2, c3, c4 = convnext_features(sd, p + ".backbone", img) if convnext else resnet_features(sd, p + ".backbone", img)
    c4_up = upsample2x(residual(sd, p + ".up4.0", c4))
    c3_skip = residual(sd, p + ".skip_layer4", c3)
    c3_f = residual(sd, p + ".fusion_layer4", torch.cat((c4_up, c3_skip), 1))
    c3_up = upsample2x(residual(sd, p + ".up3.0", c3_f))
    c2_skip = residual(sd, p + ".skip_layer3", c2)
    c2_f = residual(sd, p + ".fusion_layer3", torch.cat((c3_up, c2_skip), 1))
    c2_up = upsample2x(residual(sd, p + ".up2.0", c2_f))
    c1_skip = residual(sd, p + ".skip_layer2", c1)
    feat = residual(sd, p + ".fusion_layer2", torch.cat((c2_up, c1_skip), 1))
    if convnext:  # result_emb is applied once; both returned tensors derive from it (convNeXT/resnetUnet.py:145-146)
        feat = residual(sd, p + ".result_emb", feat)
    res = torch.cat([F.conv2d(feat, sd[p + ".finals.%d.weight" % i], sd[p + ".finals.%d.bias" % i]) for i in range(3)], 1)
    return res, feat


# --------------------------------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------------------------------
def pixel_grid(Fs):
    """u (column) and v (row) pixel-centre coordinates in [-1,1], flattened row-major (model/model.py:477-483)."""
    c = 2.0 * (torch.arange(Fs).float() + 0.5) / Fs - 1.0
    u = c.view(1, Fs).expand(Fs, Fs).reshape(-1)
    v = c.view(Fs, 1).expand(Fs, Fs).reshape(-1)
    return u, v


def offset2joint_weight(offset, depth, kernel):
    """model/model.py:466-500 — masked soft-argmax decode, B x 105 x F x F -> B x 21 x 3 (uvd)."""
    B, _, Fs, _ = offset.shape
    if depth.shape[-1] != Fs:
        depth = F.interpolate(depth, size=[Fs, Fs])  # nearest
    d = depth.reshape(B, 1, Fs * Fs)
    unit = offset[:, :J * 3].reshape(B, J, 3, Fs * Fs)
    heat = offset[:, J * 3:J * 4].reshape(B, J, Fs * Fs)
    w = offset[:, J * 4:].reshape(B, J, Fs * Fs)
    mask = (d < 0.99).float()
    w = w.masked_fill(d > 0.99, -1e8)
    nw = F.softmax(w, dim=-1)
    dist = kernel - (heat * mask) * kernel
    u, v = pixel_grid(Fs)
    coords = torch.stack((u.expand(B, -1), v.expand(B, -1), d[:, 0]), 1).unsqueeze(1)  # B 1 3 P
    val = (unit * mask.unsqueeze(2)) * dist.unsqueeze(2) + coords
    return (val * nw.unsqueeze(2)).sum(-1)


def uvd2xyz(uvd, center, M, cube, cam, img_size=128, flip=1):
    """dataloader/loader.py:775-789 (+ :836-841 get_trans_points, :265-275 pointsImgTo3D).  B x P x 3 -> B x P x 3."""
    B = uvd.shape[0]
    Mi = torch.linalg.inv(M.view(B, 1, 3, 3))
    uv = (uvd[:, :, 0:2] + 1) * (img_size / 2)
    d = uvd[:, :, 2:] * (cube.view(B, 1, 3)[:, :, 2:] / 2.0) + center.view(B, 1, 3)[:, :, 2:]
    hom = torch.cat((uv, torch.ones_like(d)), -1)
    tr = torch.matmul(Mi, hom.unsqueeze(-1)).squeeze(-1)[:, :, 0:2]
    x = (tr[:, :, 0] - cam[:, 2:3]) * d[:, :, 0] / cam[:, 0:1]
    y = flip * (tr[:, :, 1] - cam[:, 3:4]) * d[:, :, 0] / cam[:, 1:2]
    xyz = torch.stack((x, y, d[:, :, 0]), -1)
    return (xyz - center.view(B, 1, 3)) / (cube.view(B, 1, 3) / 2.0)


def img_xyz_grid(img_down, center, M, cube, cam, img_size=128, flip=1):
    """pixel grid + depth -> normalised xyz per feature pixel (dataloader/loader.py:948-953)."""
    B, _, Fs, _ = img_down.shape
    u, v = pixel_grid(Fs)
    uvd = torch.stack((u.expand(B, -1), v.expand(B, -1), img_down.reshape(B, -1)), -1)
    return uvd2xyz(uvd, center, M, cube, cam, img_size, flip)


def img2pcl_index(pcl, img_down, center, M, cube, cam, k=4, img_size=128, flip=1):
    """dataloader/loader.py:936-967 — per point the k nearest feature pixels (ascending squared distance) and
    inverse-distance weights.  Returns (closeness B x N x k fp32, index B x N x k int64)."""
    img_xyz = img_xyz_grid(img_down, center, M, cube, cam, img_size, flip)
    dist = torch.sum(torch.pow(pcl.unsqueeze(2) - img_xyz.unsqueeze(1), 2), dim=-1)
    val, idx = torch.topk(dist, k, largest=False)
    c = 1 / (val + 1e-8)
    return c / (c.sum(-1, keepdim=True) + 1e-8), idx


def img2anchor_dis(joint, img_down, center, M, cube, cam, gamma=10, img_size=128, flip=1):
    """dataloader/loader.py:791-819 — geometry adjacency map 1/(gamma*d^2+1), B x J x F x F.  The joints are pushed
    through the uvd->xyz map again exactly as the reference does (SURVEY a14 note)."""
    B, Jn, _ = joint.shape
    Fs = img_down.shape[-1]
    jx = uvd2xyz(joint, center, M, cube, cam, img_size, flip)
    ix = img_xyz_grid(img_down, center, M, cube, cam, img_size, flip)
    dist = torch.sum(torch.pow(ix.unsqueeze(1) - jx.unsqueeze(2), 2), dim=-1)
    return (1 / (gamma * dist + 1)).view(B, Jn, Fs, Fs)


def joint2heatmap(joint_uv, std, Fs, sigma=1.0):
    """util/generateFeature.py:584-600 — Gaussian on the pixel-centre grid (x = column)."""
    B, Jn, _ = joint_uv.shape
    xs = (torch.arange(Fs).float() + 0.5).view(1, 1, 1, Fs)
    ys = (torch.arange(Fs).float() + 0.5).view(1, 1, Fs, 1)
    jx = ((joint_uv[:, :, 0] + 1) / 2 * Fs).view(B, Jn, 1, 1)
    jy = ((joint_uv[:, :, 1] + 1) / 2 * Fs).view(B, Jn, 1, 1)
    return torch.exp(-(torch.pow((xs - jx) / std, 2) + torch.pow((ys - jy) / std, 2)) / (2 * sigma ** 2))


def pcl_joint2offset(joint, pcl, kernel):
    """model/model.py:503-525 — B x 21 x 3, B x N x 3 -> B x N x 84 (63 masked unit offsets (j,xyz), 21 closeness)."""
    B, Jn, _ = joint.shape
    N = pcl.shape[1]
    off = joint.unsqueeze(2) - pcl.unsqueeze(1)
    dis = torch.sqrt(torch.sum(torch.pow(off, 2), dim=-1))
    unit = (off / (dis.unsqueeze(-1) + 1e-8)).permute(0, 1, 3, 2).reshape(B, Jn * 3, N)
    clos = (kernel - dis) / kernel
    mask = (clos >= 0).float() * (pcl[:, :, 2] < 0.99).float().unsqueeze(1)
    clos = clos * mask
    unit = unit * mask.view(B, Jn, 1, N).expand(B, Jn, 3, N).reshape(B, -1, N)
    return torch.cat((unit, clos), 1).permute(0, 2, 1)


# --------------------------------------------------------------------------------------------------------------
# point-cloud ops (pointnet2_ops restated)
# --------------------------------------------------------------------------------------------------------------
def ball_query(radius, nsample, xyz, new_xyz):
    """pointnet2_ops.ball_query (third party, unpinned): for each query, the first `nsample` point indices in index
    order with d^2 < r^2 (d^2 = dx*dx+dy*dy+dz*dz and r^2 = r*r, all in fp32, products rounded individually); remaining slots repeat the first hit; zeros if none."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    d = new_xyz.unsqueeze(2) - xyz.unsqueeze(1)
    d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
    r32 = torch.tensor(radius, dtype=torch.float32)
    within = d2 < (r32 * r32)  # radius2 = radius * radius in fp32, as the extension's kernel computes it
    # rank of each hit among the hits of its query
    rank = torch.cumsum(within.long(), -1) - 1
    idx = torch.zeros(B, S, nsample, dtype=torch.long)
    first = torch.argmax(within.long(), -1)  # index of first hit (0 if none)
    any_hit = within.any(-1)
    idx[:] = (first * any_hit.long()).unsqueeze(-1)
    sel = within & (rank < nsample)
    b, s, n = torch.nonzero(sel, as_tuple=True)
    idx[b, s, rank[b, s, n]] = n
    return idx


def desa(sd, p, pcl_feat, node_feat, pcl_xyz, node_xyz, ball_override=None):
    """model/model.py:166-204 — multi-radius grouping around the joints.
    ball_override (tests only): list of 3 index tensors to use instead of ball_query (see kpfusion_forward)."""
    B, Jn, C = node_feat.shape
    xyz = torch.cat((pcl_xyz, node_xyz), 1)
    feat = torch.cat((pcl_feat, node_feat), 1)  # B (N+J) C
    outs = []
    idxs = []
    for i, r in enumerate((0.1, 0.2, 0.4)):
        idx = ball_query(r, 64, xyz, node_xyz) if ball_override is None else ball_override[i]  # B J 64
        idxs.append(idx)
        flat = idx.reshape(B, Jn * 64)
        gx = torch.gather(xyz, 1, flat.unsqueeze(-1).expand(-1, -1, 3)).view(B, Jn, 64, 3) - node_xyz.unsqueeze(2)
        gf = torch.gather(feat, 1, flat.unsqueeze(-1).expand(-1, -1, C)).view(B, Jn, 64, C) - node_feat.unsqueeze(2)
        gx = (gx / r).permute(0, 3, 1, 2)  # B 3 J 64
        gf = gf.permute(0, 3, 1, 2)
        loc = _bn(sd, p + ".bn_l0_blocks.%d" % i, F.conv2d(gx, sd[p + ".conv_l0_blocks.%d.weight" % i], sd[p + ".conv_l0_blocks.%d.bias" % i]))
        ft = _bn(sd, p + ".bn_f0_blocks.%d" % i, F.conv2d(gf, sd[p + ".conv_f0_blocks.%d.weight" % i], sd[p + ".conv_f0_blocks.%d.bias" % i]))
        g = F.relu(loc + ft)
        g = F.relu(_bn(sd, p + ".bn_blocks.%d.0" % i, F.conv2d(g, sd[p + ".conv_blocks.%d.0.weight" % i], sd[p + ".conv_blocks.%d.0.bias" % i])))
        outs.append(g.max(-1)[0])  # B C J
    outs.append(node_feat.permute(0, 2, 1))
    cat = torch.cat(outs, 1)
    out = F.relu(_bn(sd, p + ".fusion.1", F.conv1d(cat, sd[p + ".fusion.0.weight"], sd[p + ".fusion.0.bias"])))
    return out.permute(0, 2, 1), idxs


# --------------------------------------------------------------------------------------------------------------
# 21-token transformers
# --------------------------------------------------------------------------------------------------------------
def bert_layer(sd, p, h, heads=4):
    """transformers BertLayer (post-LN, eps 1e-12, GELU-erf), config overrides at model/model.py:224-232."""
    B, T, C = h.shape
    hd = C // heads

    def proj(n):
        return F.linear(h, sd[p + ".attention.self.%s.weight" % n], sd[p + ".attention.self.%s.bias" % n]).view(B, T, heads, hd).transpose(1, 2)

    q, k, v = proj("query"), proj("key"), proj("value")
    a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd), -1)
    ctx = torch.matmul(a, v).transpose(1, 2).reshape(B, T, C)
    o = F.linear(ctx, sd[p + ".attention.output.dense.weight"], sd[p + ".attention.output.dense.bias"])
    h1 = F.layer_norm(o + h, (C,), sd[p + ".attention.output.LayerNorm.weight"], sd[p + ".attention.output.LayerNorm.bias"], 1e-12)
    it = F.gelu(F.linear(h1, sd[p + ".intermediate.dense.weight"], sd[p + ".intermediate.dense.bias"]))
    o2 = F.linear(it, sd[p + ".output.dense.weight"], sd[p + ".output.dense.bias"])
    return F.layer_norm(o2 + h1, (C,), sd[p + ".output.LayerNorm.weight"], sd[p + ".output.LayerNorm.bias"], 1e-12)


def kp_interaction_tr(sd, p, x):
    """model/model.py:106-126 (KP_Interaction_TR) over model/model.py:45-103 (TR_Encoder).  B x 21 x Din -> (B x 21 x 128, B x 21 x 3)."""
    T = x.shape[1]
    h = F.linear(x, sd[p + ".bert.img_embedding.weight"], sd[p + ".bert.img_embedding.bias"]) + sd[p + ".bert.position_embeddings.weight"][:T]
    for l in range(4):
        h = bert_layer(sd, p + ".bert.encoder.layer.%d" % l, h)
    score = F.linear(h, sd[p + ".cls_head.weight"], sd[p + ".cls_head.bias"]) + F.linear(x, sd[p + ".residual.weight"], sd[p + ".residual.bias"])
    return h, score


def decoder_layer(sd, p, query, key, heads=4):
    """model/transfusion_head.py:137-173 with cross_only=True, MHA of :303-556.  query/key B x 21 x 128 -> B x 21 x 128."""
    B, T, C = query.shape
    hd = C // heads
    qe = query + sd[p + ".self_posembed.weight"][:T]
    ke = key + sd[p + ".cross_posembed.weight"][:T]
    W, bqkv = sd[p + ".multihead_attn.in_proj_weight"], sd[p + ".multihead_attn.in_proj_bias"]
    q = F.linear(qe, W[:C], bqkv[:C]) * (float(hd) ** -0.5)
    k = F.linear(ke, W[C:2 * C], bqkv[C:2 * C])
    v = F.linear(ke, W[2 * C:], bqkv[2 * C:])
    q = q.view(B, T, heads, hd).transpose(1, 2)
    k = k.view(B, T, heads, hd).transpose(1, 2)
    v = v.view(B, T, heads, hd).transpose(1, 2)
    a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)), -1)
    ctx = torch.matmul(a, v).transpose(1, 2).reshape(B, T, C)
    o = F.linear(ctx, sd[p + ".multihead_attn.out_proj.weight"], sd[p + ".multihead_attn.out_proj.bias"])
    x = F.layer_norm(query + o, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    f = F.linear(F.relu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])), sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return F.layer_norm(x + f, (C,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5)


# --------------------------------------------------------------------------------------------------------------
# fusion block and full forward
# --------------------------------------------------------------------------------------------------------------
def _emb(sd, p, x):
    """Conv1d(k=1)+BatchNorm1d on B x N x Cin -> B x N x 128 (model/model.py:254-259)."""
    y = F.conv1d(x.permute(0, 2, 1), sd[p + ".0.weight"], sd[p + ".0.bias"])
    return _bn(sd, p + ".1", y).permute(0, 2, 1)


def gather_interp(feat, idx, clos):
    """model/model.py:297-306 — feat B x C x P gathered at idx B x N x K, weighted by clos, summed over K -> B x N x C."""
    B, C, _ = feat.shape
    N, K = idx.shape[1:]
    g = torch.gather(feat, -1, idx.view(B, 1, -1).expand(-1, C, -1)).view(B, C, N, K)
    return torch.sum(g * clos.unsqueeze(1), -1).permute(0, 2, 1)


def block_kpfusion(sd, p, img_feat, img_feat_rgb, pcl, joint_xyz, clos, idx, img_offset, prev_feat, img_down,
                   center, M, cube, cam, img_size=128, flip=1, aux=None, ball_override=None):
    """model/model.py:287-351."""
    B, C, H, W = img_feat.shape
    pcl_off = pcl_joint2offset(joint_xyz, pcl, 0.8)
    pf = gather_interp(img_feat.reshape(B, C, -1), idx, clos)
    pf_rgb = gather_interp(img_feat_rgb.reshape(B, C, -1), idx, clos)
    pw = gather_interp(img_offset[:, J * 4:].reshape(B, J, -1), idx, clos)  # B N 21
    x = _emb(sd, p + ".pcl_feat_emb", pf) + _emb(sd, p + ".pcl_xyz_emb", pcl) + _emb(sd, p + ".pcl_pose_emb", torch.cat((pw, pcl_off), -1))
    x = F.relu(x)
    x = F.relu(x + _emb(sd, p + ".pcl_feat_emb_RGB", pf_rgb))
    att = F.softmax(pw.permute(0, 2, 1), -1)
    jf = torch.matmul(att, x)
    jf = F.relu(_emb(sd, p + ".joint_feat_emb", jf) + _emb(sd, p + ".joint_xyz_emb", joint_xyz))
    jf, ball_idx = desa(sd, p + ".FA", x, jf, pcl, joint_xyz, ball_override)
    h_init, r3d = kp_interaction_tr(sd, p + ".init_TR", jf)
    hm = joint2heatmap(r3d[:, :, :2], 0.8, H, sigma=1)
    gam = img2anchor_dis(r3d, img_down, center, M, cube, cam, 10, img_size, flip)
    sw = torch.sigmoid(F.conv2d(torch.cat([img_feat_rgb, hm], 1), sd[p + ".atten_spatial.weight"], sd[p + ".atten_spatial.bias"]))
    wd = torch.sigmoid(sd[p + ".weight_dis"])
    g = wd * gam + (1 - wd) * sw  # B J H W
    t = F.relu(g.unsqueeze(2) * img_feat_rgb.unsqueeze(1)).view(B, J, C, -1)
    fj = F.linear(t, sd[p + ".fc_spatial2joint_feature.weight"], sd[p + ".fc_spatial2joint_feature.bias"]).view(B, J, C)
    if prev_feat is not None:
        fj = F.relu((fj + prev_feat) / 2)
    dec = decoder_layer(sd, p + ".crossTR.decoder.3", fj, h_init)  # layers 0-2 are unobservable (transfusion_head.py:704-708)
    _, r2d = kp_interaction_tr(sd, p + ".final_TR", torch.cat([r3d, dec], 2))
    if aux is not None:
        aux.update(pcl_off=pcl_off, pf=pf, pf_rgb=pf_rgb, pw=pw, pcl_feat=x, joint_feat_desa=jf, ball_idx=ball_idx,
                   h_init=h_init, hm=hm, gam=gam, gate=g, dec=dec)
    return r3d, r2d, fj, sw


def kpfusion_forward(sd, img_rgb, img, pcl, center, M, cube, cam, kernel=0.8, img_size=128, flip=1, aux=None, overrides=None):
    """model/model.py:395-426.  Returns ([img_offset, img_offset_rgb, r3d1, r2d1, r3d2, r2d2], [sw1, sw2]).

    overrides (tests only): the forward contains two discontinuous integer decisions — the top-4 nearest pixels per point
    and ball-query membership — that flip when two fp32 distances are within rounding of each other.  To compare the
    *rest* of the pipeline across such a flip, a test may inject the index tensors the device chose:
    {"top4": (closeness, index), "ball": {1: [idx_r0, idx_r1, idx_r2], 2: [...]}}.  The decisions themselves are
    checked separately (bit-exact away from near-ties)."""
    with torch.no_grad():
        img_offset, img_feat = unet(sd, "backbone_d", img)
        img_offset_rgb, img_feat_rgb = unet(sd, "backbone_rgb", img_rgb)
        joint_uvd = offset2joint_weight(img_offset, img, kernel)
        H = img_feat.shape[-1]
        img_down = F.interpolate(img, [H, H])
        joint_xyz = uvd2xyz(joint_uvd, center, M, cube, cam, img_size, flip)
        clos, idx = img2pcl_index(pcl, img_down, center, M, cube, cam, 4, img_size, flip)
        if overrides and "top4" in overrides:
            clos, idx = overrides["top4"]
        result = [img_offset, img_offset_rgb]
        sws = []
        prev = None
        if aux is not None:
            aux.update(img_feat=img_feat, img_feat_rgb=img_feat_rgb, joint_uvd=joint_uvd, joint_xyz0=joint_xyz,
                       pcl_closeness=clos, pcl_index=idx, img_down=img_down)
        for i in (1, 2):
            a = {} if aux is not None else None
            r3d, r2d, prev, sw = block_kpfusion(sd, "block%d" % i, img_feat, img_feat_rgb, pcl, joint_xyz, clos, idx,
                                                img_offset, prev, img_down, center, M, cube, cam, img_size, flip, a,
                                                (overrides or {}).get("ball", {}).get(i))
            if aux is not None:
                aux["block%d" % i] = a
                aux["block%d_r2d" % i] = r2d
            result += [r3d, r2d]
            sws.append(sw)
            joint_xyz = r2d
        return result, sws


def backbones_forward(sd, img_rgb, img):
    """The two UNet streams only (BASELINE.json configs[1]; model/model.py:397-398)."""
    with torch.no_grad():
        od, fd = unet(sd, "backbone_d", img)
        orgb, frgb = unet(sd, "backbone_rgb", img_rgb)
    return od, fd, orgb, frgb


def to_torch_sd(np_sd):
    return {k: torch.from_numpy(v) for k, v in np_sd.items()}
