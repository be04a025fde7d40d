"""ctypes binding of libkpf_hip.so (C ABI declared in include/kpf.h).

The library is built in-tree by `make -C keypointfusion_amd/csrc` (or __graft_entry__.build()).  There is no CPU or
PyTorch fallback: if the shared object is missing or a symbol is absent, import of the compute path fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KPF_LIB_PATH") or os.path.join(_HERE, "libkpf_hip.so")  # override: A/B builds while tuning

KPF_ACT_RELU = 1
KPF_ACT_GELU = 2
KPF_RES_ADD = 4
KPF_RES_GAMMA = 8
KPF_RELU_AFTER_RES = 16
KPF_RES_GELU_GRAD = 1024
KPF_ACT_GELU_SAVE = 2048
KPF_PRO_LN = 4096
KPF_DT_F32_MMA_BF16, KPF_DT_F32_MMA_F16 = 8, 9  # weight gradients of fp32 operands on their 16-bit roundings (kpf_conv2d_wgrad_groups / _deferred)
KPF_MMA_BF16, KPF_MMA_F16 = 8192, 16384  # kpf_conv2d_f32: fp32 storage, products on operands rounded to 16 bits in registers (ABI 17)
KPF_OUT_NCHW = 32
KPF_ACT_LEAKY = 64
KPF_IN_SPLIT = 128
KPF_OUT_SPLIT = 256
KPF_W_SPLIT = 512
KPF_DT_F32, KPF_DT_BF16, KPF_DT_F16 = 0, 1, 2
ABI_VERSION = 17  # KPF_ABI_VERSION of include/kpf.h: load() refuses a library built from another revision of the interface


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "B", "IH", "IW", "Cin", "in_ld", "in_coff", "OH", "OW", "N", "KH", "KW", "sh", "sw", "ph", "pw", "Kp",
        "out_ld", "out_coff", "res_ld", "res_coff")] + [("flags", C.c_uint), ("w_unscale", C.c_float), ("tile_cfg", C.c_int), ("groups", C.c_int), ("w_gstride", C.c_long)]


class PackDesc(C.Structure):  # kpf_pack_desc (include/kpf.h)
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p)] + [(n, C.c_int) for n in (
        "N", "Cin", "KH", "KW", "mode", "n_pad", "Kp", "rows", "src_dtype", "dst_dtype", "first_block", "reserved")]


class AdamwDesc(C.Structure):  # kpf_adamw_desc (include/kpf.h)
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_long), ("first_block", C.c_int), ("reserved", C.c_int)]


class WgradGroupDesc(C.Structure):  # kpf_wgrad_group_desc (include/kpf.h)
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("first_block", C.c_int), ("sps", C.c_int), ("ldy", C.c_int)]


class ColsumDesc(C.Structure):  # kpf_colsum_desc (include/kpf.h)
    _fields_ = [("part", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("nblk", C.c_int), ("C", C.c_int), ("first_block", C.c_int), ("reserved", C.c_int)]


class WgradReduceDesc(C.Structure):  # kpf_wgrad_reduce_desc (include/kpf.h)
    _fields_ = [("part", C.c_void_p), ("dbpart", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("g_ws", C.c_long)] + [
        (n, C.c_int) for n in "S N K Cin KHW nkb ndb groups Cin_out N_out kind first_block".split()]


_P = C.c_void_p
_SIGS = {
    "kpf_conv2d_f32": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "kpf_dwconv7_ln_f32": [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P],
    "kpf_layernorm_f32": [_P, _P, _P, _P, C.c_long, C.c_int, C.c_float, _P],
    "kpf_layernorm_split_f32": [_P, _P, _P, _P, C.c_long, C.c_int, C.c_float, _P],
    "kpf_dwconv7_ln_split_f32": [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P],
    "kpf_upsample2x_f32": [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_nchw_to_nhwc_f32": [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_nhwc_to_nchw_f32": [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_maxpool3x3s2_f32": [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_inv3x3_f32": [_P, _P, C.c_int, C.c_int, _P],
    "kpf_offset2joint_f32": [_P] * 8 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _P],
    "kpf_img2pcl_top4_f32": [_P] * 9 + [C.c_int] * 6 + [_P],
    "kpf_point_assemble_f32": [_P] * 9 + [C.c_int, C.c_int, C.c_int, C.c_float, _P],
    "kpf_softmax_pool_f32": [_P, _P, _P, _P, C.c_int, C.c_int, _P],
    "kpf_ball_group_f32": [_P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _P],
    "kpf_group_max_f32": [_P, _P, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_ball_group_stacked_f32": [_P, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _P],
    "kpf_ball_group_bwd_f32": [_P, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P],
    "kpf_slices_sum_relu_forward": [_P, _P, C.c_long, C.c_int, C.c_int, C.c_int, _P],
    "kpf_slices_sum_relu_backward": [_P, _P, _P, _P, C.c_long, C.c_int, C.c_int, C.c_int, _P],
    "kpf_group_max_train_forward": [_P, _P, _P, C.c_long, C.c_int, C.c_int, _P],
    "kpf_group_max_train_backward": [_P, C.c_int, _P, _P, C.c_long, C.c_int, C.c_int, _P],
    "kpf_heat_gam_gate_f32": [_P, _P, _P, C.c_int] + [_P] * 10 + [C.c_int] * 4 + [_P],
    "kpf_gate_reduce_f32": [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P],
    "kpf_tr_encoder_f32": [_P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, _P],
    "kpf_xattn_layer_f32": [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P],
    "kpf_conv2d_h16": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P],
    "kpf_conv2d_h16_uses_8ph": [C.POINTER(ConvDesc), C.c_int],
    "kpf_conv2d_h16_ln_fold_supported": [C.POINTER(ConvDesc)],
    "kpf_dwconv7_ln_h16": [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P],
    "kpf_layernorm_h16": [_P, C.c_int, _P, _P, _P, C.c_int, C.c_long, C.c_int, C.c_float, _P],
    "kpf_upsample2x_h16": [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_cast_h16_f32": [_P, C.c_int, _P, C.c_long, C.c_int, C.c_int, C.c_int, _P],
    "kpf_cast_f32_h16": [_P, _P, C.c_int, C.c_long, _P],
    "kpf_convnext_mlp_f32": [_P] * 8 + [C.c_long, C.c_int, _P],
    "kpf_convnext_mlp_supported": [C.c_int],
    "kpf_convnext_mlp_split_f32": [_P, _P, _P, _P, C.c_float, _P, _P, C.c_float, _P, _P, C.c_long, C.c_int, _P],
    "kpf_convnext_mlp_split_supported": [C.c_int],
    "kpf_convnext_mlp_h16": [_P] * 8 + [C.c_long, C.c_int, C.c_int, _P],
    "kpf_convnext_mlp_h16_supported": [C.c_int],
    "kpf_dwconv7_stats_h16": [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_dwconv7_stats_supported": [C.c_int, C.c_int, C.c_int],
    "kpf_ln_apply_stats_h16": [_P, _P, _P, _P, C.c_long, C.c_int, C.c_float, C.c_int, _P],
    "kpf_ln_stats_merge": [_P, _P, C.c_long, C.c_int, C.c_float, _P],
    "kpf_cbam_channel_gate_f32": [_P] * 7 + [C.c_int] * 4 + [_P],
    "kpf_cbam_spatial_gate_f32": [_P, _P, _P, C.c_float, C.c_float, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_cbam_apply_f32": [_P] * 5 + [C.c_int] * 3 + [_P],
    "kpf_maxpool2x2_f32": [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_upnearest2x_add_f32": [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_mano_forward_f32": [_P, C.c_int, _P, C.c_int] + [_P] * 10 + [C.c_int, _P],
    "kpf_tr_encoder_weight_floats": [C.c_int],
    "kpf_conv_num_tile_cfgs": [],
    "kpf_xattn_weight_floats": [],
    "kpf_conv2d_wgrad_f32": [_P] * 5 + [C.c_long] + [C.c_int] * 15 + [_P],
    "kpf_bn_train_forward": [_P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, _P, C.c_float, C.c_float, C.c_int, _P, C.c_long, C.c_long, C.c_int, _P],
    "kpf_bn_train_backward": [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P, C.c_long, C.c_long, C.c_int, _P],
    "kpf_bn_train_backward_add": [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, C.c_long, C.c_long, C.c_int, _P],
    "kpf_bn_train_forward_f32": [_P] * 8 + [C.c_float, C.c_float, C.c_int, _P, C.c_long, C.c_long, C.c_int, _P],
    "kpf_bn_train_backward_f32": [_P] * 9 + [C.c_int, _P, C.c_long, C.c_long, C.c_int, _P],
    "kpf_dwconv7_f32": [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_dwconv7_add_f32": [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_conv2d_wgrad_h16": [_P, _P, C.c_int] + [_P] * 3 + [C.c_long] + [C.c_int] * 15 + [_P],
    "kpf_conv2d_wgrad_groups": [_P, _P, C.c_int] + [_P] * 3 + [C.c_long] + [C.c_int] * 18 + [_P],
    "kpf_conv2d_wgrad_deferred": [_P, _P, C.c_int] + [_P] * 3 + [C.c_long] + [C.c_int] * 18 + [C.POINTER(WgradReduceDesc), _P],
    "kpf_dwconv7_wgrad_deferred": [_P] * 5 + [C.c_long] + [C.c_int] * 4 + [C.POINTER(WgradReduceDesc), _P],
    "kpf_wgrad_reduce_multi": [C.POINTER(WgradReduceDesc), C.c_int, _P],
    "kpf_row_gather_cols_f32": [_P, C.c_long, C.c_long, C.c_long, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_bn_ssr_forward": [_P] * 7 + [C.c_float, C.c_float, _P, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, _P],
    "kpf_bn_ssr_backward": [_P] * 10 + [C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, _P],
    "kpf_bn_relu_gmax_forward": [_P] * 8 + [C.c_float, C.c_float, _P, C.c_long, C.c_long, C.c_int, C.c_int, _P],
    "kpf_bn_relu_gmax_backward": [_P, C.c_int] + [_P] * 9 + [C.c_long, C.c_long, C.c_int, C.c_int, _P],
    "kpf_bn2_add_relu_forward": [_P] * 12 + [C.c_float, C.c_float, _P, C.c_long, C.c_long, C.c_int, _P],
    "kpf_bn2_add_relu_backward": [_P] * 14 + [C.c_long, C.c_long, C.c_int, _P],
    "kpf_unstack_rows": [_P, C.c_int, _P, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_restack_rows": [C.POINTER(C.c_void_p), _P, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_add_relu_forward": [_P, _P, _P, _P, C.c_long, C.c_float, _P],
    "kpf_add_relu_backward": [_P, _P, _P, C.c_long, C.c_float, _P],
    "kpf_gate_mix_forward": [_P] * 6 + [C.c_int] * 3 + [_P],
    "kpf_gate_mix_backward": [_P] * 11 + [C.c_int] * 3 + [_P],
    "kpf_pad_rows": [_P, C.c_int, _P, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, _P],
    "kpf_pose_tokens_f32": [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P],
    "kpf_ln_train_forward_g": [_P, _P, _P, _P, C.c_int, _P, _P, C.c_long, C.c_int, C.c_int, C.c_float, _P],
    "kpf_ln_train_backward_g": [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, C.c_long, C.c_long, C.c_int, C.c_int, C.POINTER(ColsumDesc), _P],
    "kpf_dwconv7_wgrad_f32": [_P] * 5 + [C.c_long] + [C.c_int] * 4 + [_P],
    "kpf_upsample2x_bwd": [_P, _P] + [C.c_int] * 5 + [_P],
    "kpf_maxpool3x3s2_fwd": [_P, _P, _P] + [C.c_int] * 5 + [_P],
    "kpf_maxpool3x3s2_bwd": [_P, _P, _P] + [C.c_int] * 5 + [_P],
    "kpf_row_gather_fwd_f32": [_P] * 4 + [C.c_int] * 5 + [_P],
    "kpf_pack_conv_weight": [_P, C.c_int, _P, C.c_int] + [C.c_int] * 7 + [_P],
    "kpf_pack_conv_weights_multi": [_P, C.c_int, C.c_int, _P],
    "kpf_joint_heatmap_forward": [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P],
    "kpf_joint_heatmap_backward": [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P],
    "kpf_geom_gate_forward": [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P],
    "kpf_geom_gate_backward": [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P],
    "kpf_geom_gate_uvd_forward": [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P],
    "kpf_geom_gate_uvd_backward": [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P],
    "kpf_linear_wgrad_grouped": [C.POINTER(WgradGroupDesc), C.c_int, _P],
    "kpf_adamw_step_multi": [C.POINTER(AdamwDesc), C.c_int, _P, C.c_float, _P, C.c_double, C.c_double, C.c_float, C.c_float, _P],
    "kpf_adamw_step_multi_scaled": [C.POINTER(AdamwDesc), C.c_int, _P, C.c_float, _P, C.c_double, C.c_double, C.c_float, C.c_float, _P, _P, _P],
    "kpf_grad_finite_check_multi": [C.POINTER(AdamwDesc), C.c_int, _P, _P],
    "kpf_loss_scale_update": [_P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_int, _P],
    "kpf_ln_train_forward": [_P, _P, _P, _P, C.c_int, _P, _P, C.c_long, C.c_int, C.c_float, _P],
    "kpf_ln_train_backward": [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, C.c_long, C.c_long, C.c_int, _P],
    "kpf_gelu_forward": [_P, _P, C.c_int, C.c_long, _P],
    "kpf_dense_loss_forward": [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P],
    "kpf_dense_loss_backward": [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P],
    "kpf_loss_tail_forward": [_P] * 11 + [C.c_int] * 3 + [_P],
    "kpf_loss_tail_backward": [_P] * 11 + [C.c_int] * 3 + [_P],
    "kpf_layer_scale_forward": [_P, _P, C.c_int, _P, _P, C.c_long, C.c_int, _P],
    "kpf_layer_scale_backward": [_P, _P, C.c_int, _P, _P, _P, _P, C.c_long, C.c_long, C.c_int, _P],
    "kpf_drop_add_ln_forward": [_P] * 9 + [C.c_long, C.c_int, C.c_float, C.c_float, _P, C.c_int, _P],
    "kpf_drop_add_ln_backward": [_P] * 11 + [C.c_long, C.c_long, C.c_int, C.c_float, C.c_void_p, _P],
    "kpf_layer_scale_backward_g": [_P, _P, C.c_int, _P, _P, _P, _P, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p, _P],
    "kpf_layer_scale_backward_partial": [_P, _P, C.c_int, _P, _P, _P, _P, C.c_long, C.c_long, C.c_int, C.POINTER(ColsumDesc), _P],
    "kpf_ln_train_backward_partial": [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, C.c_long, C.c_long, C.c_int, C.POINTER(ColsumDesc), _P],
    "kpf_colsum_reduce_grouped": [C.POINTER(ColsumDesc), C.c_int, _P],
    "kpf_tr_stack_set_stamps": [_P],
    "kpf_xattn_train_forward": [_P, _P, _P, _P, C.c_long, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P],
    "kpf_xattn_train_backward": [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_float, C.c_int, C.c_int, _P],
    "kpf_tr_stack_train_forward": [_P, _P, _P, _P, C.c_long, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P],
    "kpf_tr_stack_train_backward": [_P, _P, _P, _P, _P, _P, C.c_int, C.c_float, C.c_int, C.c_int, _P],
    "kpf_bmm_small_k_dx": [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_bmm_small_k_fwd": [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_bmm_small_k_da": [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P],
    "kpf_attn21_forward": [_P] * 6 + [C.c_int] * 5 + [C.c_float, C.c_float, _P, C.c_int, _P],
    "kpf_attn21_backward": [_P] * 9 + [C.c_int] * 5 + [C.c_float, C.c_float, _P],
    "kpf_attn21_forward_ld": [_P] * 6 + [C.c_int] * 6 + [C.c_float, C.c_float, _P, C.c_int, _P],
    "kpf_attn21_backward_ld": [_P] * 9 + [C.c_int] * 6 + [C.c_float, C.c_float, _P],
    "kpf_gelu_backward": [_P, _P, _P, C.c_int, C.c_long, _P],
    "kpf_row_gather_invert": [_P, _P, C.c_long] + [C.c_int] * 4 + [_P],
    "kpf_row_gather_accum_f32": [_P] * 5 + [C.c_int] * 5 + [_P],
    "kpf_row_gather_bwd_f32": [_P] * 5 + [C.c_long] + [C.c_int] * 5 + [_P],
}
_LONG_SIGS = {  # entries returning a long
    "kpf_cbam_workspace_floats": [C.c_int, C.c_int, C.c_int],
    "kpf_conv2d_wgrad_ws_floats": [C.c_long, C.c_int, C.c_int],
    "kpf_dwconv7_wgrad_ws_floats": [C.c_int, C.c_int, C.c_int],
    "kpf_bn_ws_floats": [C.c_long, C.c_int],
    "kpf_row_gather_ws_ints": [C.c_int] * 4,
    "kpf_ln_ws_floats": [C.c_long, C.c_int],
    "kpf_layer_scale_ws_floats": [C.c_long, C.c_int],
    "kpf_dwconv7_stats_floats": [C.c_int] * 4,
    "kpf_pack_desc_blocks": [C.POINTER(PackDesc)],
    "kpf_tr_stack_save_floats": [C.c_int],
    "kpf_bn2_ws_floats": [C.c_long, C.c_int],
    "kpf_bn_ssr_ws_floats": [C.c_long, C.c_int, C.c_int],
    "kpf_bn_relu_gmax_ws_floats": [C.c_long, C.c_int],
    "kpf_xattn_train_save_floats": [C.c_int],
    "kpf_xattn_train_dy_floats": [C.c_int],
    "kpf_xattn_train_offset": [C.c_int, C.c_int],
    "kpf_tr_stack_out_offset": [C.c_int],
    "kpf_tr_stack_dy_floats": [C.c_int],
    "kpf_tr_stack_part_floats": [C.c_int],
    "kpf_tr_stack_offset": [C.c_int, C.c_int, C.c_int],
}
EXPORTS = sorted(list(_SIGS) + list(_LONG_SIGS) + ["kpf_last_error", "kpf_abi_version"])

_lib = None


class KpfError(RuntimeError):
    pass


def load():
    """Load the HIP library once; raise if it is not built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KpfError("libkpf_hip.so is not built: run `make -C keypointfusion_amd/csrc` "
                       "(or __graft_entry__.build()); there is no fallback compute path")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = args
        fn.restype = C.c_int
    lib.kpf_last_error.restype = C.c_char_p
    lib.kpf_abi_version.restype = C.c_int
    if lib.kpf_abi_version() != ABI_VERSION:
        raise KpfError("libkpf_hip.so reports ABI version %d, this binding was written for %d: rebuild it (`make -C keypointfusion_amd/csrc`)"
                       % (lib.kpf_abi_version(), ABI_VERSION))
    for name, args in _LONG_SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_long
    _lib = lib
    return lib


CALLS = [0]  # entry-point calls checked so far (bench.py reports the difference over one step: "library calls per step")


def check(rc, what=""):
    CALLS[0] += 1
    if rc != 0:
        raise KpfError("%s failed (%d): %s" % (what, rc, load().kpf_last_error().decode()))
