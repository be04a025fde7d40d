"""Parameter/buffer inventory of KPFusion, by state-dict key.

This is the *state-dict contract* of the drop-in boundary: a checkpoint written by the
reference's train.py (`{"model": sd}` with `module.`-prefixed keys, train.py:269-293) must load into the
MI355X model by key intersection exactly as train.py:100-107 does.  The names below therefore follow the
reference's module tree (model/model.py:354-381 KPFusion, :207-273 Block_KPFusion, :129-164 DESA,
:30-43/:106-114 TR_Encoder/KP_Interaction_TR, model/transfusion_head.py:94-110,635-665 decoder,
convNeXT/resnetUnet.py:61-111, convNeXT/convnext.py:16-38,68-103, model/resnetUnet.py:248-289,
model/resnet.py:30-60,137-170, model/hourglass.py:64-104) including every parameter the forward never
touches.  tests/test_host.py checks the result against key/shape lists dumped from the imported reference.

Each entry is (name, shape, dtype, init) where `init` tells keypointfusion_amd.weights how to draw a
synthetic value of sensible scale (there is no checkpoint in this environment).
"""

CONVNEXT = {
    "tiny": ([3, 3, 9, 3], [96, 192, 384, 768]),
    "small": ([3, 3, 27, 3], [96, 192, 384, 768]),
    "base": ([3, 3, 27, 3], [128, 256, 512, 1024]),
    "large": ([3, 3, 27, 3], [192, 384, 768, 1536]),
}
RESNET = {18: ("basic", [2, 2, 2, 2]), 50: ("bottleneck", [3, 4, 6, 3]), 101: ("bottleneck", [3, 4, 23, 3])}  # model/resnetUnet.py:12-16


def parse_net(net):
    """'KPFusion-convnext-tiny' -> ('convnext','tiny'); 'KPFusion-resnet-18' -> ('resnet', 18)."""
    tail = net.split("-")[-1]
    if "convnext" in net:
        if tail not in CONVNEXT:
            raise KeyError(tail)  # the reference raises KeyError for e.g. 'T' (convNeXT/resnetUnet.py:65-66)
        return "convnext", tail
    depth = int(tail)
    if depth not in RESNET:
        raise KeyError(depth)  # the reference indexes its `resnet` dict (model/resnetUnet.py:255); 152 is listed there but never offered
    return "resnet", depth


class _Spec(list):
    def add(self, name, shape, init, dtype="float32"):
        self.append((name, tuple(shape), dtype, init))

    def conv(self, p, cout, cin, kh, kw, bias=True, init="conv"):
        self.add(p + ".weight", (cout, cin, kh, kw), init)
        if bias:
            self.add(p + ".bias", (cout,), "bias")

    def conv1d(self, p, cout, cin, init="conv"):
        self.add(p + ".weight", (cout, cin, 1), init)
        self.add(p + ".bias", (cout,), "bias")

    def linear(self, p, cout, cin, init="linear"):
        self.add(p + ".weight", (cout, cin), init)
        self.add(p + ".bias", (cout,), "bias")

    def norm(self, p, c):  # LayerNorm
        self.add(p + ".weight", (c,), "norm_w")
        self.add(p + ".bias", (c,), "norm_b")

    def bn(self, p, c):
        self.add(p + ".weight", (c,), "norm_w")
        self.add(p + ".bias", (c,), "norm_b")
        self.add(p + ".running_mean", (c,), "bn_mean")
        self.add(p + ".running_var", (c,), "bn_var")
        self.add(p + ".num_batches_tracked", (), "zero_i64", "int64")

    def residual(self, p, cin, cout):
        h = cout // 2
        self.bn(p + ".bn1", cin)
        self.conv(p + ".conv1.conv", h, cin, 1, 1)
        self.bn(p + ".bn2", h)
        self.conv(p + ".conv2.conv", h, h, 3, 3)
        self.bn(p + ".bn3", h)
        self.conv(p + ".conv3.conv", cout, h, 1, 1)
        self.conv(p + ".skip_layer.conv", cout, cin, 1, 1)  # constructed even when unused


def _unet_decoder(s, p, d, deconv_dim, convnext):
    """d = encoder widths at strides 4,8,16,32."""
    s.residual(p + ".skip_layer4", d[2], d[2])
    s.residual(p + ".up4.0", d[3], d[3])
    s.residual(p + ".fusion_layer4", d[2] + d[3], d[2])
    s.residual(p + ".skip_layer3", d[1], d[1])
    s.residual(p + ".up3.0", d[2], d[2])
    s.residual(p + ".fusion_layer3", d[2] + d[1], d[1])
    s.residual(p + ".skip_layer2", d[0], d[0])
    s.residual(p + ".up2.0", d[1], d[1])
    s.residual(p + ".fusion_layer2", d[1] + d[0], deconv_dim)
    if convnext:
        s.residual(p + ".feat_emb", deconv_dim, deconv_dim)
        s.residual(p + ".result_emb", deconv_dim, deconv_dim)
    for i, od in enumerate((63, 21, 21)):
        s.conv(p + ".finals.%d" % i, od, deconv_dim, 1, 1, init="final")


def _convnext_unet(s, p, size, in_ch):
    depths, dims = CONVNEXT[size]
    b = p + ".backbone"
    s.conv(b + ".downsample_layers.0.0", dims[0], in_ch, 4, 4)
    s.norm(b + ".downsample_layers.0.1", dims[0])
    for i in range(1, 4):
        s.norm(b + ".downsample_layers.%d.0" % i, dims[i - 1])
        s.conv(b + ".downsample_layers.%d.1" % i, dims[i], dims[i - 1], 2, 2)
    for i in range(4):
        for j in range(depths[i]):
            q = b + ".stages.%d.%d" % (i, j)
            c = dims[i]
            s.add(q + ".gamma", (c,), "gamma")
            s.conv(q + ".dwconv", c, 1, 7, 7, init="dwconv")
            s.norm(q + ".norm", c)
            s.linear(q + ".pwconv1", 4 * c, c, init="pw")
            s.linear(q + ".pwconv2", c, 4 * c, init="pw")
    s.norm(b + ".norm", dims[3])  # unused by forward_features
    s.linear(b + ".head", 1000, dims[3], init="dead")  # unused
    _unet_decoder(s, p, dims, 128, True)


def _resnet_unet(s, p, depth, in_ch):
    """model/resnetUnet.py:248-289 (OfficialResNetUnet) over model/resnet.py:137-230 (BasicBlock :30-76 / Bottleneck :78-135)."""
    kind, layers = RESNET[depth]
    e = 4 if kind == "bottleneck" else 1
    b = p + ".backbone"
    s.conv(b + ".conv1", 64, in_ch, 7, 7, bias=False)
    s.bn(b + ".bn1", 64)
    inpl = 64
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
        for j in range(n):
            q = b + ".layer%d.%d" % (li + 1, j)
            stride = 2 if (li > 0 and j == 0) else 1
            if kind == "basic":
                s.conv(q + ".conv1", planes, inpl, 3, 3, bias=False)
                s.bn(q + ".bn1", planes)
                s.conv(q + ".conv2", planes, planes, 3, 3, bias=False)
                s.bn(q + ".bn2", planes)
            else:
                s.conv(q + ".conv1", planes, inpl, 1, 1, bias=False)
                s.bn(q + ".bn1", planes)
                s.conv(q + ".conv2", planes, planes, 3, 3, bias=False)
                s.bn(q + ".bn2", planes)
                s.conv(q + ".conv3", planes * 4, planes, 1, 1, bias=False)
                s.bn(q + ".bn3", planes * 4)
            if stride != 1 or inpl != planes * e:
                s.conv(q + ".downsample.0", planes * e, inpl, 1, 1, bias=False)
                s.bn(q + ".downsample.1", planes * e)
            inpl = planes * e
    # decoder widths are fixed; only the encoder-facing inputs grow with the block expansion (model/resnetUnet.py:256-271)
    s.residual(p + ".skip_layer4", 256 * e, 256)
    s.residual(p + ".up4.0", 512 * e, 512)
    s.residual(p + ".fusion_layer4", 512 + 256, 256)
    s.residual(p + ".skip_layer3", 128 * e, 128)
    s.residual(p + ".up3.0", 256, 256)
    s.residual(p + ".fusion_layer3", 256 + 128, 128)
    s.residual(p + ".skip_layer2", 64 * e, 64)
    s.residual(p + ".up2.0", 128, 128)
    s.residual(p + ".fusion_layer2", 128 + 64, 128)
    for i, od in enumerate((63, 21, 21)):
        s.conv(p + ".finals.%d" % i, od, 128, 1, 1, init="final")


def _tr(s, p, din):
    """KP_Interaction_TR (model/model.py:106-114) wrapping TR_Encoder (:30-43)."""
    b = p + ".bert"
    s.add(b + ".embeddings.word_embeddings.weight", (30522, 128), "dead")
    s.add(b + ".embeddings.position_embeddings.weight", (512, 128), "dead")
    s.add(b + ".embeddings.token_type_embeddings.weight", (2, 128), "dead")
    s.norm(b + ".embeddings.LayerNorm", 128)
    for l in range(4):
        q = b + ".encoder.layer.%d" % l
        for n in ("query", "key", "value"):
            s.linear(q + ".attention.self." + n, 128, 128, init="tr")
        s.linear(q + ".attention.output.dense", 128, 128, init="tr")
        s.norm(q + ".attention.output.LayerNorm", 128)
        s.linear(q + ".intermediate.dense", 16, 128, init="tr")
        s.linear(q + ".output.dense", 128, 16, init="tr")
        s.norm(q + ".output.LayerNorm", 128)
    s.linear(b + ".pooler.dense", 128, 128, init="dead")
    s.add(b + ".position_embeddings.weight", (512, 128), "emb")
    s.linear(b + ".img_embedding", 128, din, init="tr")
    s.linear(p + ".cls_head", 3, 128, init="head3")
    s.linear(p + ".residual", 3, din, init="head3")


def _block(s, p, fmap=1024):
    s.add(p + ".weight_dis", (1,), "weight_dis")
    s.linear(p + ".sampling_offsets", 8, 128, init="dead")
    s.linear(p + ".attention_weights", 4, 128, init="dead")
    s.linear(p + ".sampling_feature_embding", 128, 128, init="dead")
    fa = p + ".FA"
    for i in range(3):
        s.conv(fa + ".conv_blocks.%d.0" % i, 128, 128, 1, 1)
    for i in range(3):
        s.bn(fa + ".bn_blocks.%d.0" % i, 128)
    for i in range(3):
        s.conv(fa + ".conv_l0_blocks.%d" % i, 128, 3, 1, 1)
    for i in range(3):
        s.conv(fa + ".conv_f0_blocks.%d" % i, 128, 128, 1, 1)
    for i in range(3):
        s.bn(fa + ".bn_l0_blocks.%d" % i, 128)
    for i in range(3):
        s.bn(fa + ".bn_f0_blocks.%d" % i, 128)
    s.conv1d(fa + ".fusion.0", 128, 512)
    s.bn(fa + ".fusion.1", 128)
    _tr(s, p + ".init_TR", 128)
    _tr(s, p + ".final_TR", 131)
    for l in range(4):
        q = p + ".crossTR.decoder.%d" % l
        s.add(q + ".multihead_attn.in_proj_weight", (384, 128), "tr")
        s.add(q + ".multihead_attn.in_proj_bias", (384,), "bias")
        s.linear(q + ".multihead_attn.out_proj", 128, 128, init="tr")
        s.linear(q + ".linear1", 128, 128, init="tr")
        s.linear(q + ".linear2", 128, 128, init="tr")
        for n in ("norm1", "norm2", "norm3"):
            s.norm(q + "." + n, 128)
        s.add(q + ".self_posembed.weight", (21, 128), "emb")
        s.add(q + ".cross_posembed.weight", (21, 128), "emb")
    for n, cin in (("pcl_feat_emb", 128), ("pcl_xyz_emb", 3), ("pcl_pose_emb", 105),
                   ("joint_feat_emb", 128), ("joint_xyz_emb", 3), ("pcl_feat_emb_RGB", 128)):
        s.conv1d(p + "." + n + ".0", 128, cin)
        s.bn(p + "." + n + ".1", 128)
    s.conv(p + ".atten_spatial", 21, 149, 1, 1)
    s.linear(p + ".fc_spatial2joint_feature", 1, fmap, init="fc_spatial")  # nn.Linear(32 * 32, 1) in the reference (model/model.py:264): F^2 pixels of the feature map
    s.linear(p + ".reduction_joint_feature", 128, 256, init="dead")
    s.conv1d(p + ".reduction_joint_feature_update", 21, 63, init="dead")
    s.linear(p + ".cls_head", 3, 128, init="dead")


def kpfusion_spec(net, crop_size=128):
    """Ordered list of (key, shape, dtype, init) for KPFusion(net, ...).  crop_size = 128 is the reference (its fusion block hard-codes a 32 x 32
    feature map: model/model.py:264); any other multiple of 32 is the labelled WIDE extension of SURVEY section 0 — the same architecture with
    `fc_spatial2joint_feature` sized for (crop_size / 4)^2 pixels, every other key unchanged (not checkpoint-compatible in those two tensors)."""
    if crop_size % 32 or crop_size < 64:
        raise ValueError("crop_size must be a multiple of 32, at least 64 (got %r)" % (crop_size,))
    fam, size = parse_net(net)
    s = _Spec()
    if fam == "convnext":
        _convnext_unet(s, "backbone_rgb", size, 3)
        _convnext_unet(s, "backbone_d", size, 1)
    else:
        _resnet_unet(s, "backbone_rgb", size, 3)
        _resnet_unet(s, "backbone_d", size, 1)
    _block(s, "block1", (crop_size // 4) ** 2)
    _block(s, "block2", (crop_size // 4) ** 2)
    return list(s)


# ----------------------------------------------------------------------------------------------------------------
# Stand-alone heads named by the north star but not wired into KPFusion.forward
# ----------------------------------------------------------------------------------------------------------------
def cbam_spec(gate_channels, reduction_ratio=16, no_spatial=False):
    """model/cbam.py:84-94 CBAM(gate_channels, reduction_ratio, pool_types=['avg','max'], no_spatial)."""
    s = _Spec()
    hid = gate_channels // reduction_ratio
    s.linear("ChannelGate.mlp.1", hid, gate_channels)
    s.linear("ChannelGate.mlp.3", gate_channels, hid)
    if not no_spatial:
        s.conv("SpatialGate.spatial.conv", 1, 2, 7, 7, bias=False)
        s.bn("SpatialGate.spatial.bn", 1)
    return list(s)


def _hourglass(s, p, n, f, increase=0):
    """model/hourglass.py:122-149 (recursive)."""
    nf = f + increase
    s.residual(p + ".up1", f, f)
    s.residual(p + ".low1", f, nf)
    if n > 1:
        _hourglass(s, p + ".low2", n - 1, nf)
    else:
        s.residual(p + ".low2", nf, nf)
    s.residual(p + ".low3", nf, f)


def posenet_spec(nstack, joint_num, inp_dim=256, increase=0):
    """model/hourglass.py:166-209 PoseNet(nstack, joint_num, inp_dim, increase=0) — registration order of __init__."""
    s = _Spec()
    s.conv("pre.0.conv", 64, 1, 7, 7)
    s.bn("pre.0.bn", 64)
    s.residual("pre.1", 64, 128)
    s.residual("pre.3", 128, inp_dim)
    s.residual("pre.4", inp_dim, inp_dim)
    for i in range(nstack):
        _hourglass(s, "hgs.%d" % i, 4, inp_dim, increase)
    for i in range(nstack):
        s.residual("features.%d.0" % i, inp_dim, inp_dim)
        s.conv("features.%d.1.conv" % i, inp_dim, inp_dim, 1, 1)
        s.bn("features.%d.1.bn" % i, inp_dim)
    for name, od in (("outs_1", joint_num * 3), ("outs_2", joint_num), ("outs_3", joint_num)):
        for i in range(nstack):
            s.conv("%s.%d" % (name, i), od, inp_dim, 1, 1, init="final")
    for i in range(nstack):
        s.conv("merge_features.%d.conv.conv" % i, inp_dim, inp_dim, 1, 1)
    for i in range(nstack):
        s.conv("merge_preds.%d.conv.conv" % i, inp_dim, joint_num * 5, 1, 1)
    for i in range(nstack):
        s.conv("merge_all.%d.conv.conv" % i, inp_dim, inp_dim * 2, 1, 1)  # constructed, never called
    return list(s)


MANO_V, MANO_J, MANO_F = 778, 16, 1538


def mano_head_spec(feature_size=1024, mano_neurons=(1024, 512)):
    """model/mano_head.py:177-206 mano_regHead: ManoLayer buffers (util/manopth/manopth/manolayer.py:69-104, use_pca=False,
    flat_hand_mean=True, joint_rot_mode='axisang') followed by the regression MLP."""
    s = _Spec()
    s.add("mano_layer.th_betas", (1, 10), "mano")
    s.add("mano_layer.th_shapedirs", (MANO_V, 3, 10), "mano")
    s.add("mano_layer.th_posedirs", (MANO_V, 3, 135), "mano")
    s.add("mano_layer.th_v_template", (1, MANO_V, 3), "mano")
    s.add("mano_layer.th_J_regressor", (MANO_J, MANO_V), "mano")
    s.add("mano_layer.th_weights", (MANO_V, MANO_J), "mano")
    s.add("mano_layer.th_faces", (MANO_F, 3), "mano", "int64")
    s.add("mano_layer.th_hands_mean", (1, 45), "mano")
    s.add("mano_layer.th_selected_comps", (6, 45), "mano")  # hands_components[:ncomps] with the ctor default ncomps=6
    dims = [feature_size] + list(mano_neurons)
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        s.linear("mano_base_layer.%d" % (2 * i), b, a)
    s.linear("pose_reg", 96, dims[-1])
    s.linear("shape_reg", 10, dims[-1], init="shape_reg")
    return list(s)
