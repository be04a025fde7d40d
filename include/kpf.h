/*
 * kpf.h — C ABI of libkpf_hip.so, the MI355X (gfx950) compute library behind keypointfusion_amd.
 *
 * The reference (ru1ven/KeypointFusion) is pure Python on PyTorch: it has no FFI of its own, its "operators" are
 * the nn.Module forwards listed below, and the native code it reaches is cuDNN/cuBLAS/ATen/pointnet2_ops
 * (SURVEY.md §2.1).  Each entry point here replaces the native work behind one such forward, on raw device
 * pointers (fp32, 16-byte aligned) and a hipStream_t passed as void*.  No torch type crosses this boundary; the
 * Python host (keypointfusion_amd/model/model.py) owns memory through torch's caching allocator and passes
 * data_ptr()s.  All activations are NHWC ("pixel-major, channels contiguous"), optionally a channel slice
 * [coff, coff+C) of a wider buffer with pixel stride ld (this is how torch.cat along channels is made free).
 *
 * Every function returns 0 on success or a negative KPF_E* code; kpf_last_error() gives the thread-local message
 * (the Python side raises RuntimeError with it, mirroring the reference's Python-exception error convention).
 * Nothing here allocates, synchronises or touches the default stream: launches are graph-capturable.
 */
#ifndef KPF_H
#define KPF_H

#ifdef __cplusplus
extern "C" {
#endif

#define KPF_OK 0
#define KPF_EINVAL (-1)  /* bad shape / alignment / flag combination */
#define KPF_ELAUNCH (-2) /* HIP launch error */

/* Activation / weight storage types of the reduced-precision path (entry points ending in _h16): fp32 accumulation and fp32
 * elementwise arithmetic everywhere, 16-bit storage in HBM, v_mfma_f32_16x16x32_{bf16,f16} for the GEMMs. */
#define KPF_DT_F32 0
#define KPF_DT_BF16 1
#define KPF_DT_F16 2
/* kpf_conv2d_wgrad_groups / _deferred only (ABI 17): fp32 operands, the products taken on their bf16 / f16 roundings (16-bit MFMA, fp32 accumulation) — the weight
 * gradient of a layer whose forward ran with KPF_MMA_BF16 / _F16; honoured by the 64 x 64 tile form, fp32 products otherwise */
#define KPF_DT_F32_MMA_BF16 8
#define KPF_DT_F32_MMA_F16 9

/* kpf_conv_desc.flags */
#define KPF_ACT_RELU 1u        /* y = relu(acc + bias)                                    */
#define KPF_ACT_GELU 2u        /* y = gelu_erf(acc + bias)        (convNeXT/convnext.py:33) */
#define KPF_RES_ADD 4u         /* y = res + y                                              */
#define KPF_RES_GAMMA 8u       /* y = res + gamma[n] * y          (convNeXT/convnext.py:48-51) */
#define KPF_RELU_AFTER_RES 16u /* y = relu(res + y)               (model/resnet.py:72-73)  */
#define KPF_ACT_LEAKY 64u      /* y = leaky_relu(acc + bias, 0.01) (model/mano_head.py:199) */
#define KPF_IN_SPLIT 128u      /* `in` and `w` are in the split format below; needs a dense 1x1, Cin % 32 == 0, desc.w_unscale   */
#define KPF_W_SPLIT 512u       /* only `w` is split: fp32 activations are split in registers (any conv shape with Cin % 32 == 0)  */
#define KPF_OUT_SPLIT 256u     /* `out` is written in the split format (N, out_ld, out_coff multiples of 32)                     */
#define KPF_OUT_NCHW 32u       /* store out[b][n][oy][ox] (dense), ignoring out_ld/out_coff */
#define KPF_ACT_GELU_SAVE 2048u /* with KPF_ACT_GELU (ABI 13): also store the pre-activation acc + bias to the buffer passed in the `res` slot (res_ld / res_coff describe
                                  it; no residual is read) and use the exact erf GELU — the forward of a training Linear + GELU in one launch */
#define KPF_MMA_BF16 8192u     /* kpf_conv2d_f32 (ABI 17): fp32 storage, but the products MAY take the operands rounded to bf16 in registers (16-bit MFMA, fp32 accumulation) —
                                  what torch.autocast does to a Linear between fp32 tensors; honoured by the plain 1x1 / linear launches of the 256 x 128, 128 x 128 and
                                  128 x 64 tiles (the fusion head's wide Linears in the mixed-precision training step), ignored elsewhere */
#define KPF_MMA_F16 16384u     /* the same with f16 rounding */
#define KPF_PRO_LN 4096u       /* kpf_conv2d_h16 with KPF_ACT_GELU (ABI 15): the LayerNorm in front of the layer (convNeXT/convnext.py:42-44, pwconv1(norm(x))) is folded
                                  into the GEMM: `in` is the un-normalised tensor, `w` holds W diag(ln_w), `bias` W ln_b + b, pro_shift s[n] = sum_k w[n][k] (of the
                                  16-bit weights, in fp32) and pro_scale the (mean, rstd) pair of every pixel (kpf_ln_stats_merge);
                                  out = gelu(rstd * (acc - mean * s[n]) + bias[n]).  Layers gemm16_8ph_kernel covers only (kpf_conv2d_h16_uses_8ph) */
#define KPF_RES_GELU_GRAD 1024u /* with KPF_RES_ADD (ABI 13): y = (acc + bias) * gelu'(res) instead of the sum — the data gradient of Linear(gelu(z)) towards z
                                  in the GEMM's epilogue, res = z (training step: pwconv2 / output.dense backward; dense 1x1 only) */

typedef struct kpf_conv_desc {
  int B, IH, IW, Cin;      /* input: B x IH x IW pixels, Cin channels consumed per pixel            */
  int in_ld, in_coff;      /* floats between consecutive input pixels; first consumed channel       */
  int OH, OW, N;           /* output: B x OH x OW pixels, N channels                                */
  int KH, KW, sh, sw, ph, pw; /* filter taps, strides, zero padding                                 */
  int Kp;                  /* packed weight row length: >= KH*KW*Cin, multiple of 32, zero padded   */
  int out_ld, out_coff;    /* NHWC destination pixel stride / first channel                         */
  int res_ld, res_coff;    /* residual source (NHWC) pixel stride / first channel                   */
  unsigned flags;
  float w_unscale;         /* split operands only: weights were packed as w * 2^s, the accumulator is scaled by 2^-s */
  int tile_cfg;            /* 0: tile shape chosen by the built-in cost model; i+1: use configuration i < kpf_conv_num_tile_cfgs()
                              (the host side times the candidates once per shape and passes the fastest) */
  int groups;              /* 0 / 1: one convolution.  G > 1 (ABI 13): G independent convolutions of this shape in ONE launch (grid.y = group) —
                              group g consumes input channels [in_coff + g*Cin, +Cin) of the same pixels, multiplies by the weights at
                              w + g*w_gstride (bias[g*N + n]) and writes output channels [out_coff + g*N, +N): a grouped convolution over
                              channel-stacked activations.  The training step runs the depth and the RGB backbone (same architecture, two weight
                              sets: model/model.py:287-306) this way; a residual is stacked like the output (res_coff + g*N); no layer scale /
                              prologue / NCHW / split operands in a grouped launch. */
  long w_gstride;          /* elements between the groups' packed weight matrices (multiple of 4; 8 for 16-bit weights) */
} kpf_conv_desc;

/*
 * Implicit-GEMM convolution / linear layer on fp32 MFMA (v_mfma_f32_16x16x4_f32):
 *     out[m][n] = epilogue( sum_k A[m][k] * W[n][k] + bias[n] ),  m = (b,oy,ox),  k = (ky,kx,c) with c fastest,
 *     A[m][k]  = prologue( in[b][oy*sh+ky-ph][ox*sw+kx-pw][c] ), zero outside the image,
 *     prologue(x) = relu(x * pro_scale[c] + pro_shift[c]) when pro_scale != NULL (eval BatchNorm + ReLU on the operand).
 * Replaces: nn.Conv2d / nn.Linear / nn.Conv1d(k=1) forwards reached from model/hourglass.py:64-119 (Conv, Residual),
 * convNeXT/convnext.py:31-34,77,84 (pwconv1/2, stem, downsample), model/resnet.py:52-55,166 (BasicBlock, stem),
 * convNeXT/resnetUnet.py:95-97 (finals) — i.e. cuDNN implicit GEMM + cuBLAS sgemm in the reference.
 * w is [N][Kp] (PyTorch [out][in] order, taps reordered to (ky,kx,c)); bias may be NULL.
 * Requirements: Cin, in_ld, in_coff multiples of 4.  out/res ld and coff multiples of 4 select the float4 epilogue
 * (scalar stores otherwise).
 *
 * Split operands (KPF_IN_SPLIT / KPF_OUT_SPLIT): fp32-accurate GEMM on the f16 matrix cores.  A row of C values (C % 32 == 0)
 * keeps its C*4 bytes but holds, per 32-channel block, [32 x f16 hi | 32 x f16 lo] with x ~= hi + lo (22 significant bits).
 * With KPF_IN_SPLIT both `in` and `w` (rows of Kp) are in that format and each 32-deep K tile is 3 v_mfma_f32_16x16x32_f16
 * (hi*hi + hi*lo + lo*hi) accumulated in fp32: per-product error ~2^-21, below the rounding noise of an fp32 accumulation chain,
 * at 3/16 of the f32-input MFMA's cycles.  Weights are packed as w * 2^s (keeps the lo halves out of the f16 subnormals) and
 * desc.w_unscale = 2^-s.  With KPF_W_SPLIT only the weights are pre-split: activations stay fp32 in memory and are split in registers
 * after the LDS read (and after the operand prologue), at some VALU cost per fragment.  Producers of split activations: this function with KPF_OUT_SPLIT, kpf_layernorm_split_f32,
 * kpf_dwconv7_ln_split_f32.  Values are clamped to +-65504 when split; callers bound their activations (LayerNorm / GELU outputs).
 */
int kpf_conv2d_f32(const kpf_conv_desc* d, const float* in, const float* w, const float* bias,
                   const float* pro_scale, const float* pro_shift, const float* gamma, const float* res,
                   float* out, void* stream);

/*
 * ConvNeXt block front half: depthwise 7x7 (pad 3) + bias, then LayerNorm over channels (eps), fused.
 * x, y: dense NHWC [B][H][W][C]; w_dw: [49][C] (tap-major); replaces convNeXT/convnext.py:41-43
 * (cuDNN depthwise conv + permute + ATen layer_norm).  C % 4 == 0, C <= 2048.
 */
int kpf_dwconv7_ln_f32(const float* x, const float* w_dw, const float* b_dw, const float* ln_w, const float* ln_b,
                       float* y, int B, int H, int W, int C, float eps, void* stream);

/*
 * ConvNeXt block back half, fused: out = x + gamma * (W2 . GELU(W1 . y + b1) + b2)  (convNeXT/convnext.py:44-51: pwconv1, GELU,
 * pwconv2, layer scale, residual).  y, x, out dense [M][C]; w1 [4C][C], w2 [C][4C] in PyTorch layout.  The 4C-wide hidden
 * tensor never leaves registers.  Supported C: see kpf_convnext_mlp_supported(); other widths use two kpf_conv2d_f32 launches.
 * out may alias x.
 */
int kpf_convnext_mlp_f32(const float* y, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                         const float* gamma, float* out, long M, int C, void* stream);
int kpf_convnext_mlp_supported(int C);
/* The same block on the f16 matrix cores with split operands (see kpf_conv2d_f32): y is the split LayerNorm output
 * (kpf_dwconv7_ln_split_f32), w1 [4C][C] and w2 [C][4C] are split-packed and scaled by 1/w*_unscale (powers of two); inside each
 * 32-wide hidden block w2's columns are ordered k = 8g+j -> hidden 16*(j>>2) + 4g + (j&3), the order in which GEMM1's accumulator
 * registers become GEMM2's operand.  x and out are fp32. */
int kpf_convnext_mlp_split_f32(const float* y_split, const float* x, const float* w1_split, const float* b1, float w1_unscale,
                               const float* w2_split_perm, const float* b2, float w2_unscale, const float* gamma, float* out, long M,
                               int C, void* stream);
int kpf_convnext_mlp_split_supported(int C);
/* The same block on 16-bit storage (round 4; replaces the kpf_conv2d_h16 pair pwconv1 + GELU / pwconv2 + layer scale + residual of
 * convNeXT/convnext.py:44-51 where the hidden tensor's HBM round trip dominates — C = 128 / 256, the first two stages of ConvNeXt-B): y, x,
 * out [M][C] and w1 [4C][C] in the storage type `dtype` (KPF_DT_BF16 / KPF_DT_F16), fp32 biases and layer scale, fp32 accumulation and
 * arithmetic, one rounding of the result.  w2_chunks is pwconv2's weight [C][4C] repacked chunk-major, [4C/32][C][32], with the hidden index
 * inside each 32-block in the order k = 8g + j -> hidden 16*(j>>2) + 4g + (j&3) (the order of kpf_convnext_mlp_split_f32).  GELU is evaluated as
 * x * sigmoid(x (c1 + c3 x^2 + c5 x^4)), |error| <= 2.6e-5 absolute: below the rounding of a 16-bit result.  out may alias x. */
int kpf_convnext_mlp_h16(const void* y, const void* x, const void* w1, const float* b1, const void* w2_chunks, const float* b2,
                         const float* gamma, void* out, long M, int C, int dtype, void* stream);
int kpf_convnext_mlp_h16_supported(int C);

/*
 * Depthwise 7x7 + bias as a pure stencil that also emits LayerNorm statistics, and the LayerNorm from those statistics (round 4; together they
 * replace kpf_dwconv7_ln_h16 = convNeXT/convnext.py:41-44 on 16-bit storage for C % 64 == 0 and maps of at least 16 x 16):
 *   kpf_dwconv7_stats_h16   y = dwconv7(x) + b (un-normalised, storage type `dtype`), stats[pixel][C/64][2] = (mean, sum of centred squares) of
 *                           every 64-channel chunk of the fp32 result;  kpf_dwconv7_stats_floats = the floats `stats` needs.
 *   kpf_ln_apply_stats_h16  y <- (y - mean) * rstd * ln_w + ln_b in place, (mean, rstd) merged from the chunk statistics (Chan's update, fixed
 *                           order).  (The statistics are laid out for the following GEMM to consume them instead: out = rstd * (W'y - mean * s) + b'.)
 */
int kpf_dwconv7_stats_h16(const void* x, const float* w_dw, const float* b_dw, void* y, float* stats, int B, int H, int W, int C, int dtype,
                          void* stream);
int kpf_dwconv7_stats_supported(int H, int W, int C);
long kpf_dwconv7_stats_floats(int B, int H, int W, int C);
int kpf_ln_apply_stats_h16(void* y, const float* stats, const float* ln_w, const float* ln_b, long rows, int C, float eps, int dtype, void* stream);
/* mean_rstd[pixel] = (mean, 1 / sqrt(var + eps)) merged from the chunk statistics of kpf_dwconv7_stats_h16 exactly as kpf_ln_apply_stats_h16 merges them:
 * the operand statistics of a kpf_conv2d_h16 launch with KPF_PRO_LN (the nn.LayerNorm of convNeXT/convnext.py:42 folded into pwconv1, ABI 15). */
int kpf_ln_stats_merge(const float* stats, float* mean_rstd, long rows, int C, float eps, void* stream);

/*
 * LayerNorm over the channel dimension of `rows` pixels (biased variance, (x-u)/sqrt(var+eps)*w+b).
 * Replaces convNeXT/convnext.py:205-214 in both data formats (stem/downsample norms).  In-place allowed.
 */
int kpf_layernorm_f32(const float* x, const float* w, const float* b, float* y, long rows, int C, float eps, void* stream);

/* Same two ops writing their output in the split operand format of kpf_conv2d_f32 (C % 32 == 0): the LayerNorm output feeds only
 * pwconv1 / a downsample convolution, so it is split where it is produced. */
int kpf_dwconv7_ln_split_f32(const float* x, const float* w_dw, const float* b_dw, const float* ln_w, const float* ln_b, float* y,
                             int B, int H, int W, int C, float eps, void* stream);
int kpf_layernorm_split_f32(const float* x, const float* w, const float* b, float* y, long rows, int C, float eps, void* stream);

/*
 * Bilinear x2 upsampling, align_corners=False (nn.Upsample(scale_factor=2, mode='bilinear'),
 * convNeXT/resnetUnet.py:76-77): src dense NHWC [B][H][W][C] -> dst channel slice (ld, coff) of [B][2H][2W][*].
 */
int kpf_upsample2x_f32(const float* src, float* dst, int B, int H, int W, int C, int dst_ld, int dst_coff, void* stream);

/* NCHW [B][C][H][W] -> NHWC [B][H][W][Cpad] (channels >= C zero-filled).  Boundary repack of the inputs of
 * KPFusion.forward (model/model.py:395). */
int kpf_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, int Cpad, void* stream);

/* NHWC channel slice -> dense NCHW (boundary repack of returned tensors). */
int kpf_nhwc_to_nchw_f32(const float* src, float* dst, int B, int C, int H, int W, int src_ld, int src_coff, void* stream);

/* 3x3 stride-2 pad-1 max pooling on dense NHWC (model/resnet.py:169,236). */
int kpf_maxpool3x3s2_f32(const float* src, float* dst, int B, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Keypoint-fusion head (Block_KPFusion, model/model.py:287-351) — geometry, point-cloud and 21-token kernels.
 * J = 21 joints, feature maps F x F (P = F*F pixels), N points, C = 128 channels; all buffers dense fp32.
 * ------------------------------------------------------------------------------------------------------------ */

/* Inverse of B 3x3 crop matrices [B][3][3] -> Minv [B][3][3] in the operation order of the reference's torch.linalg.inv on the CPU
 * (dataloader/loader.py:781 -> ATen linalg_solve_ex on A^T -> LAPACK getrf/getrs), bit-identical to it, so that pixel positions and
 * the integer top-4 / ball-query decisions derived from them match the reference bit for bit.  fused = 1: the library's FMA code
 * path (Intel hosts), 0: its separately-rounded path (AMD hosts) — keypointfusion_amd/inv3x3.py::host_mode() tells which one the
 * host's torch.linalg.inv takes.  The geometry entry points below take this Minv (or one computed by torch.linalg.inv itself). */
int kpf_inv3x3_f32(const float* M, float* Minv, int B, int fused, void* stream);

/* Masked soft-argmax decode of the depth stream's offset maps + uvd->xyz (model/model.py:466-500 offset2joint_weight,
 * dataloader/loader.py:775-789 uvd_nl2xyznl_tensor).  offset NCHW [B][105][P]; depth [B][S][S] (nearest-downsampled to
 * F on the fly, model/model.py:471); center [B][3], Minv [B][3][3] (kpf_inv3x3_f32), cube [B][3], cam [B][4].
 * -> joint_uvd, joint_xyz [B][21][3]. */
int kpf_offset2joint_f32(const float* offset, const float* depth, const float* center, const float* Minv, const float* cube,
                         const float* cam, float* joint_uvd, float* joint_xyz, int B, int S, int F, float kernel,
                         int img_size, int flip, void* stream);

/* Per point the 4 nearest feature pixels, ascending squared distance, and inverse-distance weights
 * (dataloader/loader.py:936-967 img2pcl_index; replaces the B x N x P x 3 broadcast + ATen topk).
 * -> closeness [B][N][4] fp32, index [B][N][4] int32 (values in [0,P)), img_xyz [B][P][3] (pixel positions, may be NULL). */
int kpf_img2pcl_top4_f32(const float* pcl, const float* depth, const float* center, const float* Minv, const float* cube,
                         const float* cam, float* closeness, int* index, float* img_xyz, int B, int N, int S, int F,
                         int img_size, int flip, void* stream);

/* Point feature gather + pcl_joint2offset (model/model.py:295-309, 503-525): builds the operand rows of the point
 * embedding GEMMs: A1 [B*N][240] = [pf 128 | xyz 3 | pw 21 | unit offsets 63 | closeness 21 | 0 x4], A2 [B*N][128] = pf_rgb.
 * feat_* NHWC [B][P][128]; offset NCHW [B][105][P]. */
int kpf_point_assemble_f32(const float* feat_d, const float* feat_rgb, const float* offset, const float* pcl,
                           const float* joint_xyz, const float* closeness, const int* index, float* A1, float* A2, int B,
                           int N, int P, float kernel, void* stream);

/* attention = softmax_N(pw), joint_feat = attention @ X (model/model.py:319-320).  X [B*N][128];
 * -> JA [B*21][132] = [joint_feat 128 | joint xyz 3 | 0]. */
int kpf_softmax_pool_f32(const float* A1, const float* X, const float* joint_xyz, float* JA, int B, int N, void* stream);

/* pointnet2_ops ball_query + group_points for the 3 DESA radii (model/model.py:158,171-178): points = cat(pcl, joints),
 * features = cat(X [B*N][128], JF [B*21][jf_ld]).  -> G [3][B*21*64][132] = [feat[idx]-feat_j | (xyz[idx]-xyz_j)/r | 0],
 * idx_out [3][B*21][64] int32 (may be NULL). */
int kpf_ball_group_f32(const float* pcl, const float* joint_xyz, const float* X, const float* JF, int jf_ld, float* G,
                       int* idx_out, int B, int N, float r0, float r1, float r2, void* stream);

/* max over groups of `group` consecutive rows (torch.max(dim=-1), model/model.py:198). */
int kpf_group_max_f32(const float* in, float* out, long rows, int group, int C, int out_ld, int out_coff, void* stream);
/* The same grouping with the three radii CHANNEL-STACKED (ABI 16; the training step's grouped DESA launches): GF [B*21*64][3*128] grouped feature
 * differences, GX [B*21*64][3*4] offsets / radius (3 + a zero channel) — radius i in columns [128 i, 128 i + 128) / [4 i, 4 i + 4). */
int kpf_ball_group_stacked_f32(const float* pcl, const float* joint_xyz, const float* X, const float* JF, int jf_ld, float* GF, float* GX, int* idx_out,
                               int B, int N, float r0, float r1, float r2, void* stream);

/* Heat-map, geometry adjacency map, spatial attention, gate (model/model.py:334-338; util/generateFeature.py:584-600;
 * dataloader/loader.py:791-819).  SF [B*P][sf_ld] = feature part of atten_spatial (a GEMM), Wh [21][21] its heat-map part.
 * -> sw_out NCHW [B][21][P] (returned spatial weight), Gw [B][21][P] = gate * fc_spatial2joint_feature.weight. */
int kpf_heat_gam_gate_f32(const float* r3d, const float* img_xyz, const float* SF, int sf_ld, const float* Wh,
                          const float* bias, const float* weight_dis, const float* wfc, const float* center,
                          const float* Minv, const float* cube, const float* cam, float* sw_out, float* Gw, int B, int F,
                          int img_size, int flip, void* stream);

/* img_feat_j = Gw @ relu(feat) + b (model/model.py:340-344), optional relu((. + prev)/2).  -> out [B][21][128]. */
int kpf_gate_reduce_f32(const float* Gw, const float* feat, const float* bfc, const float* prev, float* out, int B, int P,
                        void* stream);

/* KP_Interaction_TR (model/model.py:106-126): embed + 4 BERT layers + 3-vector heads on 21 tokens, one workgroup per
 * sample.  x [B*21][ldx] (Din used), W = packed weights (keypointfusion_amd/engine.py:pack_tr), -> h [B][21][128],
 * score [B][21][3]; score2 (may be NULL) receives a second copy with row stride score2_ld. */
int kpf_tr_encoder_f32(const float* x, int ldx, int Din, const float* W, float* h, float* score, float* score2,
                       int score2_ld, int B, void* stream);
int kpf_tr_encoder_weight_floats(int Din);

/* The observable layer of updatedDecoder (model/transfusion_head.py:137-173 with cross_only): query/key [B][21][128]
 * -> out rows [B*21] with stride out_ld at column out_coff. */
int kpf_xattn_layer_f32(const float* query, const float* key, const float* W, float* out, int out_ld, int out_coff, int B,
                        void* stream);
int kpf_xattn_weight_floats(void);

/* ---- stand-alone heads: named by the north star, not called by KPFusion.forward ---------------------- */

/* CBAM ChannelGate (model/cbam.py:26-57): scale[b][c] = sigmoid(mlp(avgpool(x)) + mlp(maxpool(x))), mlp = Linear(C,Cr) ReLU
 * Linear(Cr,C) with w1 [Cr][C], w2 [C][Cr] (nn.Linear layout).  x NHWC [B][HW][C]; workspace >= kpf_cbam_workspace_floats(). */
long kpf_cbam_workspace_floats(int B, int HW, int C);
int kpf_cbam_channel_gate_f32(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                              float* workspace, float* scale, int B, int HW, int C, int Cr, void* stream);
/* CBAM SpatialGate value (model/cbam.py:65-81): comp [B][H][W][2] = (max_c, mean_c) of x*scale, then
 * sgate[b][y][x] = sigmoid(bn_scale * conv7x7_pad3(comp; w7 [2][7][7]) + bn_shift) (eval BatchNorm folded by the caller). */
int kpf_cbam_spatial_gate_f32(const float* x, const float* scale, const float* w7, float bn_scale, float bn_shift, float* comp,
                              float* sgate, int B, int H, int W, int C, void* stream);
/* out0 = (x*scale)*sgate, out1 = (x*scale)*(1-sgate) — the tuple SpatialGate.forward returns (model/cbam.py:82);
 * sgate == out1 == NULL: out0 = x*scale (CBAM(no_spatial=True), model/cbam.py:90-94). */
int kpf_cbam_apply_f32(const float* x, const float* scale, const float* sgate, float* out0, float* out1, int B, int HW, int C,
                       void* stream);

/* nn.MaxPool2d(2,2) on NHWC (model/hourglass.py:8,128) and nn.Upsample(scale_factor=2, nearest) fused with the hourglass
 * skip add `up1 + up2` (model/hourglass.py:138-149): out [B][2h][2w][C] = up1 + low[y/2][x/2]. */
int kpf_maxpool2x2_f32(const float* src, float* dst, int B, int H, int W, int C, void* stream);
int kpf_upnearest2x_add_f32(const float* low, const float* up1, float* out, int B, int h, int w, int C, void* stream);

/* mano_regHead tail (model/mano_head.py:212-225) + ManoLayer.forward (util/manopth/manopth/manolayer.py:106-273; use_pca=False,
 * axis-angle joints, right hand, no centring / translation): pose6d rows [B][ld6] (16 x 6D rotations), betas rows [B][ldb] (10)
 * -> verts [B][778][3] mm, joints [B][21][3] mm in OBMAN2MANO order, rotmat [B][16][3][3], pose_aa [B][48].
 * Model arrays: shapedirs_t [10][778*3], posedirs_t [135][778*3] (transposed blend-shape bases), v_template [778*3],
 * j_regressor [16][778], skin_weights [778][16], hands_mean [45]. */
int kpf_mano_forward_f32(const float* pose6d, int ld6, const float* betas, int ldb, const float* shapedirs_t,
                         const float* posedirs_t, const float* v_template, const float* j_regressor, const float* skin_weights,
                         const float* hands_mean, float* verts, float* joints, float* rotmat, float* pose_aa, int B, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Reduced-precision path (BASELINE configs[2] bf16, configs[4] fp16): same operators on 16-bit NHWC activations.  `dtype` is
 * KPF_DT_BF16 or KPF_DT_F16; LayerNorm / bias / layer-scale parameters and the depthwise taps stay fp32.
 * ------------------------------------------------------------------------------------------------------------ */

/* kpf_conv2d_f32 on 16-bit operands: in / w / res / out are `dtype` (w rows [N][Kp], Kp % 64 == 0, Cin % 8 == 0), bias / gamma /
 * prologue scale+shift fp32, fp32 accumulation on v_mfma_f32_16x16x32_{bf16,f16}.  With KPF_OUT_NCHW the output is fp32 NCHW (the
 * heads).  The split-operand flags do not apply. */
int kpf_conv2d_h16(const kpf_conv_desc* desc, const void* in, const void* w, const float* bias, const float* pro_scale,
                   const float* pro_shift, const float* gamma, const void* res, void* out, int dtype, void* stream);
/* 1 when kpf_conv2d_h16 runs `d` on the eight-phase 256 x 256 kernel (gemm16_8ph_kernel: dense 1x1, K % 128 == 0, N % 256 == 0, at least 224 tiles), 0 when
 * on igemm_h16_kernel — for callers that label per-kernel measurements (bench.py's roofline); has_prologue: pro_scale != NULL. */
int kpf_conv2d_h16_uses_8ph(const kpf_conv_desc* d, int has_prologue);
/* 1 when kpf_conv2d_h16 accepts `d` with KPF_PRO_LN (ABI 15): GELU epilogue on a dense 1x1 layer with Cin == Kp, Kp % 128 == 0, N % 256 == 0 — a rule over
 * the layer's shape, never over its batch (the arithmetic of a sample must not depend on the size of its batch). */
int kpf_conv2d_h16_ln_fold_supported(const kpf_conv_desc* d);

/* kpf_dwconv7_ln_f32 with 16-bit activations in and out (fp32 taps, fp32 accumulation and LayerNorm statistics). */
int kpf_dwconv7_ln_h16(const void* x, const float* w_dw, const float* b_dw, const float* ln_w, const float* ln_b, void* y, int B,
                       int H, int W, int C, float eps, int dtype, void* stream);

/* LayerNorm over C: x_dtype -> y_dtype, either fp32 -> 16-bit (behind the fp32 stem convolution) or 16-bit -> the same type. */
int kpf_layernorm_h16(const void* x, int x_dtype, const float* w, const float* b, void* y, int y_dtype, long rows, int C, float eps,
                      void* stream);

/* kpf_upsample2x_f32 on 16-bit activations (dst_ld / dst_coff in elements). */
int kpf_upsample2x_h16(const void* src, void* dst, int B, int H, int W, int C, int dst_ld, int dst_coff, int dtype, void* stream);

/* 16-bit NHWC channel slice -> dense fp32 [rows][C] (feature maps handed to the fp32 fusion head / module boundary). */
int kpf_cast_h16_f32(const void* src, int dtype, float* dst, long rows, int C, int src_ld, int src_coff, void* stream);

/* dense fp32 [n] -> 16-bit (behind the fp32 stem + max-pool of the ResNet backbones); n % 4 == 0. */
int kpf_cast_f32_h16(const float* src, void* dst, int dtype, long n, void* stream);

/*
 * Training: weight and bias gradients of nn.Conv2d / nn.Linear — what autograd's convolution_backward /
 * addmm_backward compute for the layers of kpf_conv2d_f32 when the reference trains (train.py:198-232).
 *     dw[n][c][ky][kx] = sum_{b,oy,ox} dy[b][oy][ox][n] * x[b][oy*sh+ky-ph][ox*sw+kx-pw][c]      (OIHW, PyTorch's layout)
 *     db[n]            = sum_{b,oy,ox} dy[b][oy][ox][n]                                           (db may be NULL)
 * dy: NHWC [B][OH][OW] with pixel stride ldy (N channels used), x: NHWC [B][H][W] with pixel stride ldx (Cin channels used);
 * Cin, N, ldx, ldy multiples of 4, 16-byte aligned pointers.  fp32 on v_mfma_f32_16x16x4_f32, split over the pixel range with a
 * fixed-order (run-to-run deterministic) reduction through `ws` (>= kpf_conv2d_wgrad_ws_floats(B*OH*OW, N, KH*KW*Cin) floats).
 */
long kpf_conv2d_wgrad_ws_floats(long M, int N, int K);
int kpf_conv2d_wgrad_f32(const float* dy, const float* x, float* dw, float* db, float* ws, long ws_floats, int B, int H, int W, int Cin,
                         int ldx, int OH, int OW, int N, int ldy, int KH, int KW, int sh, int sw, int ph, int pw, void* stream);
/* The same with dy and x in 16-bit storage (dtype = KPF_DT_BF16 / KPF_DT_F16; mixed-precision training): values are widened to fp32
 * on the way into LDS, products and sums are the fp32 ones (the master weight's gradient is not rounded).  Cin, N, ldx, ldy % 8 == 0. */
int kpf_conv2d_wgrad_h16(const void* dy, const void* x, int dtype, float* dw, float* db, float* ws, long ws_floats, int B, int H, int W, int Cin,
                         int ldx, int OH, int OW, int N, int ldy, int KH, int KW, int sh, int sw, int ph, int pw, void* stream);

/* G channel-stacked weight gradients in one launch (ABI 13; the data-parallel twin of kpf_conv_desc::groups): group g multiplies
 * dy[.., g*N + n] (pixel stride ldy >= G*N) with x[.., g*Cin + c] (pixel stride ldx >= G*Cin) and writes dw[g][N][Cin][KH][KW], db[g][N]
 * (db may be NULL).  Same kernels as G separate kpf_conv2d_wgrad_f32 / _h16 calls on the channel slices; fp32 operands also keep their split and
 * summation order (bit-identical results), 16-bit operands split the pixel range for 1/G of the chip each (same sums, another order).
 * ws_floats >= G * kpf_conv2d_wgrad_ws_floats(M, N, K).  dtype: KPF_DT_F32 / _BF16 / _F16 (both operands). */
int kpf_conv2d_wgrad_groups(const void* dy, const void* x, int dtype, float* dw, float* db, float* ws, long ws_floats, int groups, int B, int H, int W, int Cin,
                            int ldx, int OH, int OW, int N, int ldy, int KH, int KW, int sh, int sw, int ph, int pw, int cin_valid, int n_valid, void* stream);
/* cin_valid / n_valid (0 = Cin / N): the operands carry zero channels up to whole channel groups (the 3-, 105-, 131-, 149-wide layers of the fusion
 * head, model/model.py:99-104, 254-262) and dw [n_valid][cin_valid][KH][KW], db [n_valid] are written without them. */

/* Row pad / column-slice copy / type change in one launch, and the pose tokens of a fusion block at their padded width (csrc/kpf_train.hip). */
int kpf_pad_rows(const void* src, int src_dtype, void* dst, int dst_dtype, long rows, int C, int src_ld, int Cp, void* stream);
/* Training (ABI 17): channel-stacked maps of G <= 4 paired networks <-> one dense fp32 map per network (train_graph.TrainGraph._forward: the seam between the
 * paired backbones and the fusion head).  src [rows][ld] (KPF_DT_*), group g's C channels at column g * gs; dst [G][rows][C] fp32.  kpf_restack_rows is the
 * gradient: grads[g] fp32 [rows][C] or NULL (zero) -> dst [rows][ld] of dst_dtype, every column (also the pad columns) written once. */
int kpf_unstack_rows(const void* src, int src_dtype, float* dst, long rows, int G, int C, int ld, int gs, int hw, void* stream);
int kpf_restack_rows(const float* const* grads, void* dst, int dst_dtype, long rows, int G, int C, int ld, int gs, int hw, void* stream);
/* hw > 0 (rows % hw == 0): the dense maps are NCHW — dst [G][rows / hw][C][hw], grads[g] [rows / hw][C][hw] — the layout the reference returns its offset maps
 * in (model/model.py:425) and the decode / the loss read; hw == 0: rows of C as above. */
int kpf_pose_tokens_f32(const float* pw, const float* joint, const float* pcl, float* out, int B, int N, int J, int ld, float kernel, void* stream);
/* out = relu(scale * (a + b + c)) (b, c nullable; fp32, n % 4 == 0) and d = out > 0 ? scale * dy : 0 — the gradient of every addend (ABI 13;
 * model/model.py:190, 417-422). */
int kpf_add_relu_forward(const float* a, const float* b, const float* c, float* out, long n, float scale, void* stream);
int kpf_add_relu_backward(const float* dy, const float* out, float* dx, long n, float scale, void* stream);
/* The gate of a fusion block around its two maps (ABI 13; model/model.py:334-341): sw = sigmoid(logits) (returned to the loss), gw = (sigmoid(weight_dis) * gam +
 * (1 - sigmoid(weight_dis)) * sw) * w_fc[p].  logits [B*P][J] (the rows of the atten_spatial GEMM), gam / sw / gw [B][J][P], weight_dis [1], w_fc [P].
 * backward: d_sw nullable; ws >= B*J floats; parameter gradients summed in a fixed order. */
int kpf_gate_mix_forward(const float* logits, const float* gam, const float* weight_dis, const float* w_fc, float* sw, float* gw, int B, int J, int P, void* stream);
int kpf_gate_mix_backward(const float* sw, const float* gam, const float* weight_dis, const float* w_fc, const float* d_sw, const float* d_gw, float* d_gam,
                          float* d_logits, float* d_w_fc, float* d_weight_dis, float* ws, int B, int J, int P, void* stream);
/* Depthwise 7x7 (pad 3) + bias alone, y = dwconv(x): the ConvNeXt block's first op with its output kept for the LayerNorm backward,
 * and its data gradient (dx = the same convolution of dy with w_dw's taps mirrored, zero bias).  w_dw [49][C], x != y. */
int kpf_dwconv7_f32(const float* x, const float* w_dw, const float* b_dw, float* y, int B, int H, int W, int C, void* stream);
/* the same + addend (y's shape) on the output: the data gradient of a ConvNeXt block with the skip path's gradient folded in */
int kpf_dwconv7_add_f32(const float* x, const float* w_dw, const float* b_dw, const float* addend, float* y, int B, int H, int W, int C, void* stream);
/* The same for the ConvNeXt block's depthwise 7x7 (pad 3): dw [C][7][7] (= PyTorch's [C][1][7][7]), db [C] or NULL; dy, x dense NHWC
 * [B][H][W][C], C % 4 == 0; ws >= kpf_dwconv7_wgrad_ws_floats(B, H, C) floats. */
long kpf_dwconv7_wgrad_ws_floats(int B, int H, int C);
int kpf_dwconv7_wgrad_f32(const float* dy, const float* x, float* dw, float* db, float* ws, long ws_floats, int B, int H, int W, int C,
                          void* stream);

/* Split weight gradients with the fixed-order reduce of MANY calls in one launch (ABI 17; `loss.backward()` of train.py:262 issues ~100 weight-gradient
 * GEMMs whose partial sums each needed a 5-us reduce launch).  kpf_conv2d_wgrad_deferred / kpf_dwconv7_wgrad_deferred are kpf_conv2d_wgrad_groups /
 * kpf_dwconv7_wgrad_f32 without the reduce: they launch the split GEMM and describe the pending reduce in *reduce (kind < 0: the call wrote dw itself,
 * nothing is pending).  Until kpf_wgrad_reduce_multi has run on the same stream, dw / db are UNWRITTEN and ws belongs to the pending reduce.
 * kpf_wgrad_reduce_multi sums in the order the per-call reduce does: the gradients have the same bits either way. */
#define KPF_WGRAD_REDUCE_BATCH 24
typedef struct kpf_wgrad_reduce_desc {
  const float* part;   /* [S][N*K] partial tiles ([S][49][C] for the depthwise form) */
  const float* dbpart; /* [S][N] or NULL */
  float* dw;
  float* db;
  long g_ws;           /* floats between the workspaces of consecutive channel groups */
  int S, N, K, Cin, KHW, nkb, ndb, groups, Cin_out, N_out;
  int kind;            /* 0 convolution / Linear, 1 depthwise 7x7 (N = C), < 0 nothing pending */
  int first_block;     /* filled by kpf_wgrad_reduce_multi */
} kpf_wgrad_reduce_desc;
int kpf_conv2d_wgrad_deferred(const void* dy, const void* x, int dtype, float* dw, float* db, float* ws, long ws_floats, int groups, int B, int H, int W, int Cin,
                              int ldx, int OH, int OW, int N, int ldy, int KH, int KW, int sh, int sw, int ph, int pw, int cin_valid, int n_valid,
                              kpf_wgrad_reduce_desc* reduce, void* stream);
int kpf_dwconv7_wgrad_deferred(const float* dy, const float* x, float* dw, float* db, float* ws, long ws_floats, int B, int H, int W, int C,
                               kpf_wgrad_reduce_desc* reduce, void* stream);
int kpf_wgrad_reduce_multi(const kpf_wgrad_reduce_desc* descs, int n, void* stream);

/*
 * BatchNorm with batch statistics (+ optional ReLU) on NHWC rows x [M][C], forward and backward: nn.BatchNorm2d / BatchNorm1d in
 * train mode as the reference's Residual blocks use them (model/hourglass.py:84-119, `BN -> ReLU -> conv`), C % 4 == 0.
 * forward : mean[c], invstd[c] = 1/sqrt(biased var + eps) are written for the backward; running_mean / running_var (may be NULL)
 *           are updated with `momentum` (unbiased variance), y = [relu]((x - mean) * invstd * w + b).
 * backward: dz = dy (masked by y > 0 when relu; y = the forward output), dx = w*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)),
 *           dw = sum dz*xhat, db = sum dz (dw, db may be NULL).
 * ws >= kpf_bn_ws_floats(M, C) floats; sums are added in a fixed order (run-to-run deterministic).
 */
long kpf_bn_ws_floats(long M, int C);
/* Storage-type generic forms (mixed-precision training): x / dx in x_dtype, y / dy in y_dtype, each KPF_DT_F32 or one 16-bit type
 * (the same on both sides when both are 16-bit); statistics, parameters and all arithmetic stay fp32. */
int kpf_bn_train_forward(const void* x, int x_dtype, const float* w, const float* b, void* y, int y_dtype, float* mean, float* invstd,
                         float* running_mean, float* running_var, float momentum, float eps, int relu, float* ws, long ws_floats, long M, int C,
                         void* stream);
int kpf_bn_train_backward(const void* dy, const void* x, const void* y, int x_dtype, int y_dtype, const float* mean, const float* invstd,
                          const float* w, void* dx, float* dw, float* db, int relu, float* ws, long ws_floats, long M, int C, void* stream);
/* the same with dx += addend (x's type and shape; nullable): a second gradient of x — the skip path of a Residual block — folded in */
int kpf_bn_train_backward_add(const void* dy, const void* x, const void* y, int x_dtype, int y_dtype, const float* mean, const float* invstd,
                              const float* w, const void* addend, void* dx, float* dw, float* db, int relu, float* ws, long ws_floats, long M, int C,
                              void* stream);
/* Training (ABI 17): the embedding sums of a fusion block with their BatchNorms (batch statistics) inside: x [rows][n C], n = n1 + n2 <= 4 sibling Linear outputs side by
 * side (C % 4 == 0, C <= 256), out [rows][C] = relu(S1) (n2 == 0) or relu(relu(S1) + S2) with S1 / S2 the sums of the first n1 / next n2 normalised blocks
 * (model/model.py:254-259, 417-422) — one pass over the pre-activations forward, two backward; w, b, dw, db, running statistics: [n C]; stats [2][n C] = mean, invstd. */
long kpf_bn_ssr_ws_floats(long rows, int C, int n);
int kpf_bn_ssr_forward(const float* x, const float* w, const float* b, float* out, float* stats, float* rmean, float* rvar, float momentum, float eps, float* ws,
                       long ws_floats, long rows, int C, int n1, int n2, void* stream);
int kpf_bn_ssr_backward(const float* dout, const float* out, const float* x, const float* stats, const float* w, const float* b, float* dx, float* dw, float* db, float* ws,
                        long ws_floats, long rows, int C, int n1, int n2, void* stream);
/* Training (ABI 17): y[g][c] = max over the `group` consecutive rows of relu(BN(x)) (batch statistics; fp32 rows x [M][C], M % group == 0) — DESA's `bn -> ReLU ->
 * max over a ball's 64 members` (model/model.py:188-192) — in one pass over the pre-activation (the winner's member index in arg [M / group][C]: first maximum);
 * backward from dmax (rows dmax_ld floats apart): dx dense, dw / db; the BatchNorm sums run over the winners only.  stats [2][C] = mean, invstd. */
long kpf_bn_relu_gmax_ws_floats(long M, int C);
int kpf_bn_relu_gmax_forward(const float* x, const float* w, const float* b, float* y, unsigned char* arg, float* stats, float* rmean, float* rvar, float momentum,
                             float eps, float* ws, long ws_floats, long M, int group, int C, void* stream);
int kpf_bn_relu_gmax_backward(const float* dmax, int dmax_ld, const float* y, const unsigned char* arg, const float* x, const float* stats, const float* w, float* dx,
                              float* dw, float* db, float* ws, long ws_floats, long M, int group, int C, void* stream);
/* Training (ABI 17): out = relu(BN_a(xa) + BN_b(xb)) with batch statistics on fp32 rows [M][C] — DESA's local + feature branches (model/model.py:176-190) — in one
 * pass over the two pre-activations (statistics: two partial + finalize pairs, then ONE element-wise launch), and its backward (masked gradient, both branches'
 * sums, both input gradients: four launches).  stats [4][C] receives mean_a, invstd_a, mean_b, invstd_b (kept for the backward); running statistics nullable and
 * updated like kpf_bn_train_forward's; ws >= kpf_bn2_ws_floats(M, C) floats. */
long kpf_bn2_ws_floats(long M, int C);
int kpf_bn2_add_relu_forward(const float* xa, const float* xb, const float* wa, const float* ba, const float* wb, const float* bb, float* out, float* stats,
                             float* rmean_a, float* rvar_a, float* rmean_b, float* rvar_b, float momentum, float eps, float* ws, long ws_floats, long M, int C,
                             void* stream);
int kpf_bn2_add_relu_backward(const float* dy, const float* out, const float* xa, const float* xb, const float* stats, const float* wa, const float* wb, float* dxa,
                              float* dxb, float* dwa, float* dba, float* dwb, float* dbb, float* ws, long ws_floats, long M, int C, void* stream);
int kpf_bn_train_forward_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* invstd, float* running_mean,
                             float* running_var, float momentum, float eps, int relu, float* ws, long ws_floats, long M, int C,
                             void* stream);
int kpf_bn_train_backward_f32(const float* dy, const float* x, const float* y, const float* mean, const float* invstd, const float* w,
                              float* dx, float* dw, float* db, int relu, float* ws, long ws_floats, long M, int C, void* stream);

/*
 * Training: data-movement ops of the train-mode forward with a run-to-run deterministic backward (gather form, fixed
 * summation order, no floating-point atomics) — what torch autograd computes with atomics for F.interpolate(bilinear)
 * (model/resnetUnet.py:259), nn.MaxPool2d(3, 2, 1) (model/resnet.py:168), torch.gather on feature rows (model/model.py:297-306) and
 * pointnet2's group_points (model/model.py:174).  NHWC, C % 4 == 0; `dtype` = KPF_DT_F32 / _BF16 / _F16 storage of the activations.
 *   kpf_upsample2x_bwd    dy [B][2H][2W][C] -> dx [B][H][W][C]: the exact transpose of kpf_upsample2x_f32 / _h16.
 *   kpf_maxpool3x3s2_fwd  y [B][OH][OW][C] and tap [B][OH][OW][C] (uint8: winning tap ky*3+kx, first maximum in scan order).
 *   kpf_maxpool3x3s2_bwd  dx[b][iy][ix][c] = sum of dy over the <= 4 windows whose winning tap is (iy, ix).
 *   kpf_row_gather_fwd_f32  out[b][r][:] = sum_{g<G} w[b][r*G+g] * src[b][idx[b][r*G+g]][:]   (src [B][P][C], idx int32 [B][R*G],
 *                           w [B][R*G] or NULL for unit weights, out [B][R][C]).
 *   kpf_row_gather_bwd_f32  dsrc[b][p][:] = sum over entries e ascending with idx[b][e] == p of w[b][e] * dout[b][e/G][:];
 *                           R*G <= 8192 entries and P <= 2048 source rows per image; ws >= kpf_row_gather_ws_ints(B, P, R, G) ints
 *                           (the per-image inverse lists: a stable counting sort of the entries by source row).
 */
int kpf_upsample2x_bwd(const void* dy, void* dx, int dtype, int B, int H, int W, int C, void* stream);
int kpf_maxpool3x3s2_fwd(const void* x, void* y, unsigned char* tap, int dtype, int B, int H, int W, int C, void* stream);
int kpf_maxpool3x3s2_bwd(const void* dy, const unsigned char* tap, void* dx, int dtype, int B, int H, int W, int C, void* stream);
int kpf_row_gather_fwd_f32(const float* src, const int* idx, const float* w, float* out, int B, int P, int R, int G, int C, void* stream);
/* ABI 17: the forward alone for a slice of any width C and layout — element (b, p, c) of the source at b * sb + p * sp + c * sc floats (out [B][R][C] dense; same sum
 * order over g): the 21 weight-logit channels the pose tokens sample (model/model.py:372-376, detached there), read in place from the NCHW offset map. */
int kpf_row_gather_cols_f32(const float* src, long sb, long sp, long sc, const int* idx, const float* w, float* out, int B, int P, int R, int G, int C, void* stream);
long kpf_row_gather_ws_ints(int B, int P, int R, int G);
int kpf_row_gather_bwd_f32(const float* dout, const int* idx, const float* w, float* dsrc, int* ws, long ws_ints, int B, int P, int R, int G, int C,
                           void* stream);
/* Its two halves as separate calls (ABI 13): invert an index tensor once, accumulate for every gather that used it (start = ws, list = ws + B*(P+1);
 * a sub-range of the images is addressed by offsetting both). */
int kpf_row_gather_invert(const int* idx, int* ws, long ws_ints, int B, int P, int R, int G, void* stream);
int kpf_row_gather_accum_f32(const float* dout, const int* start, const int* list, const float* w, float* dsrc, int B, int P, int R, int G, int C, void* stream);

/* Training: LayerNorm over the last axis (nn.LayerNorm / F.layer_norm of convNeXT/convnext.py:43,199-214 and the fusion head's post-LN
 * layers) and GELU(erf) (convNeXT/convnext.py:33), forward and backward.  x [rows][C] fp32, C % 4 == 0, C <= 1024; y in y_dtype (fp32 or the
 * 16-bit operand type of the following GEMM); mean / rstd [rows] are kept for the backward.  backward: dx fp32, dw / db [C] column sums over
 * all rows added in a fixed order through ws (>= kpf_ln_ws_floats(rows, C) floats).  GELU: element-wise on n % 4 == 0 elements of `dtype`. */
/* Training (ABI 16): backward of kpf_ball_group_stacked_f32 towards the features, all three radii in one launch: d3 [B*Jn*64][ld3] (radius i in columns
 * [128 i, 128 i + 128)), start / list = kpf_row_gather_invert of idx [3][B][Jn*64] as 3 B images of P = N + Jn rows (G = 1); dX [B][N][128] for the point
 * features, dnode [B][Jn][128] for the joint features (their gathered rows minus the group sums of the centre subtraction).  Fixed summation order. */
int kpf_ball_group_bwd_f32(const float* d3, int ld3, const int* start, const int* list, float* dX, float* dnode, int B, int N, int Jn, void* stream);
/* Training (ABI 16): y[r][c] = max over the `group` consecutive rows of x [rows*group][C] (model/model.py:192 `.max(2)` over a ball's 64 members), the
 * first maximum's member index kept in arg [rows][C]; backward: dx[r][m][c] = (m == arg[r][c]) ? dy[r][c] : 0, every element written once. */
/* Training (ABI 16): the embedding sums of a fusion block over one channel-stacked tensor y [rows][(n1 + n2) * C]: out [rows][C] = relu(S1) (n2 == 0) or
 * relu(relu(S1) + S2), S1 / S2 = sums of the first n1 / next n2 column blocks (model/model.py:417-422); backward: dy of the same shape. */
int kpf_slices_sum_relu_forward(const float* y, float* out, long rows, int C, int n1, int n2, void* stream);
int kpf_slices_sum_relu_backward(const float* dout, const float* out, const float* y, float* dy, long rows, int C, int n1, int n2, void* stream);
int kpf_group_max_train_forward(const float* x, float* y, unsigned char* arg, long rows, int group, int C, void* stream);
int kpf_group_max_train_backward(const float* dy, int dy_ld /* floats between rows of dy: it may be a column slice of a wider matrix */, const unsigned char* arg, float* dx,
                                 long rows, int group, int C, void* stream);
long kpf_ln_ws_floats(long rows, int C);
int kpf_ln_train_forward(const float* x, const float* w, const float* b, void* y, int y_dtype, float* mean, float* rstd, long rows, int C, float eps,
                         void* stream);
int kpf_ln_train_backward(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* w, float* dx, float* dw,
                          float* db, float* ws, long ws_floats, long rows, int C, void* stream);
int kpf_gelu_forward(const void* x, void* y, int dtype, long n, void* stream);
int kpf_gelu_backward(const void* dy, const void* x, void* dx, int dtype, long n, void* stream);

/* Training: the attention core of the 21-token stacks (BertSelfAttention of model/model.py:30-126; multi_head_attention_forward,
 * model/transfusion_head.py:527-546): per (sample, head) P = softmax(scale * Q K^T), ctx = dropout(P) V.  q, k, v, ctx, dq, dk, dv are rows
 * [B*T][ld] fp32 with head h in channels [h*hd, (h+1)*hd) (the projections' own layout: no head transposes); T = 21, hd = 32.  P
 * [B][H][T][T] fp32 and the keep mask M (bytes) are kept for the backward.  p_drop > 0: masks from a hash of (rng[0] = seed, rng[1] =
 * counter advanced by the host once per forward, call_id, element); rng is a device pointer to two int64. */
int kpf_attn21_forward(const float* q, const float* k, const float* v, float* ctx, float* P, unsigned char* M, int B, int T, int H, int hd, int ld,
                       float scale, float p_drop, const long* rng, int call_id, void* stream);
int kpf_attn21_backward(const float* dctx, const float* q, const float* k, const float* v, const float* P, const unsigned char* M, float* dq, float* dk,
                        float* dv, int B, int T, int H, int hd, int ld, float scale, float p_drop, void* stream);
/* The same pair with q / k / v (dq / dk / dv) at row stride ld and ctx (dctx) at its own row stride ldc (ABI 13). */
int kpf_attn21_forward_ld(const float* q, const float* k, const float* v, float* ctx, float* P, unsigned char* M, int B, int T, int H, int hd, int ld, int ldc,
                          float scale, float p_drop, const long* rng, int call_id, void* stream);
int kpf_attn21_backward_ld(const float* dctx, const float* q, const float* k, const float* v, const float* P, const unsigned char* M, float* dq, float* dk,
                           float* dv, int B, int T, int H, int hd, int ld, int ldc, float scale, float p_drop, void* stream);

/* Training: dX[b] (P x C) = A[b]^T (P x J) @ dOut[b] (J x C) for small J (<= 64; J = 21 joints): the operand gradient of the per-sample
 * products of model/model.py:318-320 and 336-341 (a K = J batched GEMM the library handles badly).  fp32, C % 4 == 0, J * C * 4 B <= 64 KB. */
int kpf_bmm_small_k_dx(const float* A, const float* dOut, float* dX, int B, int J, int P, int C, void* stream);
/* round 5 (ABI 14): the other two products of the pair — out[b] = A[b] (J x P) @ X[b] (P x C) (J <= 24, C % 4 == 0) and dA[b] = dOut[b] (J x C) @ X[b]^T — what
   torch.bmm computed in `joint_feat = img2joint @ img_feat` (model/model.py:318-320) and the gated reduction (model/model.py:336-341) of the training step and in
   their autograd backward; fixed summation order. */
int kpf_bmm_small_k_fwd(const float* A, const float* X, float* out, int B, int J, int P, int C, void* stream);
int kpf_bmm_small_k_da(const float* dOut, const float* X, float* dA, int B, int J, int P, int C, void* stream);

/* Training: layer scale + residual of the ConvNeXt block, out = x + gamma * y (convNeXT/convnext.py:48-51): x / out / g fp32 [rows][C], y / dy in
 * y_dtype (fp32 or the 16-bit GEMM storage type), gamma / dgamma [C].  backward: dy = g * gamma, dgamma = column sums of g * y added in a
 * fixed order through ws (>= kpf_layer_scale_ws_floats(rows, C) floats); C % 4 == 0, C <= 1024. */
long kpf_layer_scale_ws_floats(long rows, int C);
int kpf_layer_scale_forward(const float* x, const void* y, int y_dtype, const float* gamma, float* out, long rows, int C, void* stream);
int kpf_layer_scale_backward(const float* g, const void* y, int y_dtype, const float* gamma, void* dy, float* dgamma, float* ws, long ws_floats,
                             long rows, int C, void* stream);

/* Training: the dense-stage loss (train.py:211-224 with GFM.joint2offset / offset2joint_weight, util/generateFeature.py:59-84,166-195, and
 * model/loss.py's SmoothL1): pd [B][5J][F][F] fp32 NCHW (3J unit offsets (j, xyz), J heat maps, J weight logits), img [B][1][S][S], uvd_gt
 * [B][J][3].  forward writes part [B][J][2] = per-(sample, joint) sums of the element losses (pixel term over its 4 channels, coordinate
 * term over its 3 coordinates): loss_pixel = sum(part[..., 0]) / (B*4J*F*F), loss_coord = sum(part[..., 1]) / (B*J*3).  backward: grad2 =
 * device pointer to {dL/dloss_pixel, dL/dloss_coord}; every element of dpd [B][5J][F][F] is written.  F*F <= 1024, S % F == 0. */
int kpf_dense_loss_forward(const float* pd, const float* img, const float* uvd_gt, float* part, int B, int J, int F, int S, float kernel_size, void* stream);
int kpf_dense_loss_backward(const float* pd, const float* img, const float* uvd_gt, const float* grad2, float* dpd, int B, int J, int F, int S,
                            float kernel_size, void* stream);

/* Training: the rest of the loss of train.py:225-261 (two launches forward, one backward): SmoothL1 (model/loss.py) between the four
 * fusion-block joint sets joints4[i] [B][J][3] and xyz_gt, the two spatial-weight terms against GFM.joint2heatmap(uvd_gt[..., :2])
 * (util/generateFeature.py:584-600) divided by its global maximum, the epoch gate of train.py:251 read from a device scalar, the weights and
 * the total, together with the dense stages' partial sums dense0 / dense1 (kpf_dense_loss_forward's part; nullable).
 * joints4 / sw2 / djoints4 / dsw2: HOST arrays of device pointers (entries nullable = term absent / gradient not wanted); sw2[t] element
 * (b, j, p = y*F + x) at b*sw_strides6[3t] + j*sw_strides6[3t+1] + p*sw_strides6[3t+2]; epoch: device float (nullable: gates = 1);
 * cfg9 (host) = {std, sigma0, sigma1, coord_weight, deconv_weight, spatial_weight0, spatial_weight1, spatial_epoch0, spatial_epoch1};
 * sp_part: 2*B*J floats of workspace; out16: [0] loss, [1..4] pixel_0 coord_0 pixel_1 coord_1, [5..8] coord_2..5, [9..10] spatial_0..1
 * (weighted, gated), [12..13] gates, [14..15] the heat maps' maxima (read again by the backward).
 * backward: g = device pointer to dL/dloss; writes djoints4[i], dsw2[t] (same strides as sw2[t]) and gdense2 = {g*deconv_weight,
 * g*coord_weight}, the grad2 of kpf_dense_loss_backward. */
int kpf_loss_tail_forward(const float* dense0, const float* dense1, const float* const* joints4, const float* xyz_gt, const float* uvd_gt,
                          const float* const* sw2, const long* sw_strides6, const float* epoch, const float* cfg9, float* sp_part, float* out16,
                          int B, int J, int F, void* stream);
int kpf_loss_tail_backward(const float* const* joints4, const float* xyz_gt, const float* uvd_gt, const float* const* sw2, const long* sw_strides6,
                           const float* cfg9, const float* out16, const float* g, float* const* djoints4, float* const* dsw2, float* gdense2,
                           int B, int J, int F, void* stream);

/* Training: one-launch packing of a reference-layout weight w [N][Cin][KH][KW] (src_dtype: KPF_DT_F32 master, or a 16-bit copy) into an
 * operand of kpf_conv2d_f32 / _h16 (dst_dtype; fp32 -> 16-bit rounds to nearest even), rows zero-padded to Kp:
 *   mode 0  forward rows        dst [n_pad][Kp], k = (ky, kx, c)
 *   mode 1  data-gradient rows  dst [Cin][Kp],   k = (ky, kx, n < n_pad) with mirrored taps (transposed convolution, stride-1 form)
 *   mode 2  patchify data-grad  dst [(ky, kx, c)][Kp], k = n < n_pad        (kernel == stride convolutions: a GEMM + pixel un-shuffle;
 *                               with Cin = 1: the depthwise tap table [KH*KW][C] of kpf_dwconv7_f32)
 *   mode 3  mode 2 with the taps mirrored (the depthwise convolution's data gradient) */
int kpf_pack_conv_weight(const void* w, int src_dtype, void* dst, int dst_dtype, int N, int Cin, int KH, int KW, int mode, int n_pad, int Kp,
                         void* stream);

/* All operands of an iteration in ONE launch: `descs_device` is an array of `ndesc` descriptors in DEVICE memory, sorted by
 * first_block; descriptor i owns workgroups [first_block, first_block + kpf_pack_desc_blocks(i)) of the `total_blocks` launched.
 * Fields as the arguments of kpf_pack_conv_weight (rows = destination rows: n_pad / Cin / KH*KW*Cin for mode 0 / 1 / 2-3). */
typedef struct kpf_pack_desc {
  const void* src;
  void* dst;
  int N, Cin, KH, KW, mode, n_pad, Kp, rows;
  int src_dtype, dst_dtype, first_block, reserved; /* reserved (ABI 13): 0, or the destination's row stride in elements when it exceeds Kp (the operand
                                                      is a column range of a wider stacked matrix); mode 4 (ABI 13): a vector of N elements, rows = 1 */
} kpf_pack_desc;
int kpf_pack_conv_weights_multi(const kpf_pack_desc* descs_device, int ndesc, int total_blocks, void* stream);
/* workgroups descriptor d occupies in that launch (first_block of the next descriptor = first_block + this; ABI 13: the kernel serves 1x1 transposes,
 * k x k rows, tap tables and k x k data-gradient rows with LDS-staged forms of their own geometry) */
long kpf_pack_desc_blocks(const kpf_pack_desc* d);

/* Training: AdamW (train.py:84-91) over many tensors in ceil(n / KPF_ADAMW_BATCH) launches.  descs: HOST array (p, m, v updated in place;
 * g read; all fp32, n elements each; first_block is set by the call); the learning rate is *lr_dev when lr_dev is non-null (device
 * scalar: a captured iteration follows the scheduler) else lr_host; *step_dev is the number of steps taken BEFORE this one (device float;
 * the caller increments it afterwards).  Arithmetic of torch.optim.AdamW(fused=True): decoupled weight decay, bias corrections from
 * step + 1, no amsgrad. */
#define KPF_ADAMW_BATCH 320 /* (round 6: 80 -> 320, a 15-KB kernel-argument segment: the ~1100 tensors of a paired ConvNeXt-T KPFusion take 4 launches instead of 14) */
typedef struct kpf_adamw_desc {
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;
  int first_block, reserved;
} kpf_adamw_desc;
int kpf_adamw_step_multi(const kpf_adamw_desc* descs, int n, const float* lr_dev, float lr_host, const float* step_dev, double beta1, double beta2, float eps,
                         float weight_decay, void* stream);
/* round 5 (ABI 14): loss scaling for fp16 mixed-precision training, on the device (what torch.cuda.amp.GradScaler does around train.py:262-264; a captured
 * iteration keeps following it).  kpf_grad_finite_check_multi: *found_dev |= 1 when any gradient of the descriptors (p / m / v unused) holds an inf or a nan.
 * kpf_adamw_step_multi_scaled: kpf_adamw_step_multi on g * *inv_scale_dev, or a no-op when *skip_dev != 0 (either pointer may be null).
 * kpf_loss_scale_update: found -> scale *= backoff and the growth tracker restarts; otherwise *step_dev += 1 (nullable) and scale *= growth every `interval` clean
 * steps; scale stays in [1, 2^24], *inv_scale_dev = 1 / scale, *found_dev = 0 for the next iteration. */
int kpf_grad_finite_check_multi(const kpf_adamw_desc* descs, int n, int* found_dev, void* stream);
int kpf_adamw_step_multi_scaled(const kpf_adamw_desc* descs, int n, const float* lr_dev, float lr_host, const float* step_dev, double beta1, double beta2,
                                float eps, float weight_decay, const float* inv_scale_dev, const int* skip_dev, void* stream);
int kpf_loss_scale_update(float* scale_dev, float* inv_scale_dev, int* tracker_dev, int* found_dev, float* step_dev, float growth, float backoff, int interval,
                          void* stream);

/* Training: the weight (and bias) gradients of MANY Linear layers over few rows in one launch per KPF_WGRAD_GROUP_BATCH problems:
 * dw[N][K] = dy[M][N]^T x[M][K], db[N] = column sums of dy (db nullable), fp32, N % 4 == K % 4 == 0, rows contiguous.  A workgroup owns a
 * 64 x 64 tile of one problem and walks all M rows (no split, no reduce launch) — meant for M up to ~1000 (the 21-token stacks of the
 * fusion head: M = 21 B), where a launch pair per layer costs more than the arithmetic; the row range is still summed in the splits of
 * kpf_conv2d_wgrad_f32 and the split sums in split order, so both forms return the same bits.  descs: HOST array. */
#define KPF_WGRAD_GROUP_BATCH 64
typedef struct kpf_wgrad_group_desc {
  const float* dy;
  const float* x;
  float* dw;
  float* db;
  int M, N, K, first_block; /* (first_block, sps: set by the call) */
  int sps, ldy;             /* ldy: floats between rows of dy (0 = N; ABI 13: dy may be a column slice of a wider matrix, e.g. one of q | k | v) */
} kpf_wgrad_group_desc;
int kpf_linear_wgrad_grouped(const kpf_wgrad_group_desc* descs, int n, void* stream);

/* Training: the column-sum reduce behind every LayerNorm / layer-scale backward (d gamma, d beta) for many layers in one launch.  The
 * _partial forms of kpf_ln_train_backward / kpf_layer_scale_backward run everything but that reduce, leave the per-workgroup partial sums
 * in `ws` (which must stay alive) and fill `desc`; kpf_colsum_reduce_grouped(descs: HOST array) performs them KPF_COLSUM_BATCH per
 * launch with the arithmetic of the immediate form (same bits). */
#define KPF_COLSUM_BATCH 96
typedef struct kpf_colsum_desc {
  const float* part; /* [nblk][2][C] */
  float* dw;         /* column sums of plane 0 */
  float* db;         /* column sums of plane 1 */
  int nblk, C, first_block;
  int reserved;      /* ABI 17: that many floats behind db[C - 1] are set to zero (0: none) — the sum over the batch of a [B][T*C] gradient of the first T rows of an
                        embedding table [L][C] written as the table's whole gradient: part = the [B][T*C] tensor read as [B][2][T*C/2], dw = the table's gradient,
                        db = dw + T*C/2, reserved = (L - T) * C (model/model.py:84 position embeddings; model/transfusion_head.py:150-152) */
} kpf_colsum_desc;
int kpf_ln_train_backward_partial(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* w, float* dx, float* dw,
                                  float* db, float* ws, long ws_floats, long rows, int C, kpf_colsum_desc* desc, void* stream);

/* LayerNorm with G parameter sets (ABI 13): row r is normalised with set r % G, i.e. w, b, dw, db hold [G][C] — what a LayerNorm over each of
 * the G channel groups of a [pixels][G*C] tensor is when that tensor is read as [pixels*G][C] rows (the paired backbones of the training
 * step).  G in {1, 2, 4}, rows % G == 0; G = 1 is kpf_ln_train_forward / _backward.  ws_floats >= kpf_ln_ws_floats(rows, G*C).
 * desc != NULL: the column-sum reduce is described instead of launched (kpf_ln_train_backward_partial's contract, C -> G*C). */
int kpf_ln_train_forward_g(const float* x, const float* w, const float* b, void* y, int y_dtype, float* mean, float* rstd, long rows, int C, int G,
                           float eps, void* stream);
int kpf_ln_train_backward_g(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* w, float* dx, float* dw,
                            float* db, float* ws, long ws_floats, long rows, int C, int G, kpf_colsum_desc* desc, void* stream);

/* y = LayerNorm(h + dropout(o)) in one launch each way (ABI 13; model/model.py:30-126: dense -> dropout -> residual add -> LayerNorm, both halves of a
 * BERT layer).  fp32 rows of C (C % 4 == 0, C <= 1024); xs = h + dropout(o) [rows][C], the keep mask [rows][C] bytes (NULL when p_drop == 0) and
 * mean / rstd [rows] are kept for the backward; rng: the device-resident (seed, counter) pair of kpf_attn21_forward, call_id a per-site constant.
 * backward: dh = d xs, d_o = d xs * mask / (1 - p), dw / db [C]; ws_floats >= kpf_ln_ws_floats(rows, C); desc as kpf_ln_train_backward_partial. */
int kpf_drop_add_ln_forward(const float* o, const float* h, const float* w, const float* b, float* xs, float* y, unsigned char* mask, float* mean, float* rstd,
                            long rows, int C, float eps, float p_drop, const long* rng, int call_id, void* stream);
int kpf_drop_add_ln_backward(const float* dy, const float* xs, const float* mean, const float* rstd, const float* w, const unsigned char* mask, float* dh,
                             float* d_o, float* dw, float* db, float* ws, long ws_floats, long rows, int C, float p_drop, kpf_colsum_desc* desc, void* stream);
/* The layer-scale backward with G parameter sets (same row-view convention as kpf_ln_train_backward_g: rows counts [rows][C] rows of a
 * [rows / G][G*C] tensor, gamma / dgamma hold [G][C]); ws_floats >= kpf_layer_scale_ws_floats(rows, G*C).  The forward needs no twin
 * (kpf_layer_scale_forward with C := G*C). */
int kpf_layer_scale_backward_g(const float* g, const void* y, int y_dtype, const float* gamma, void* dy, float* dgamma, float* ws, long ws_floats,
                               long rows, int C, int G, kpf_colsum_desc* desc /* nullable: as kpf_layer_scale_backward_partial */, void* stream);
int kpf_layer_scale_backward_partial(const float* g, const void* y, int y_dtype, const float* gamma, void* dy, float* dgamma, float* ws, long ws_floats,
                                     long rows, int C, kpf_colsum_desc* desc, void* stream);
int kpf_colsum_reduce_grouped(const kpf_colsum_desc* descs, int n, void* stream);

/* Training (ABI 16): the four BERT layers of a KP_Interaction_TR stack (21 tokens x 128, 4 heads x 32, intermediate 16, GELU-erf, post-LN eps 1e-12,
 * hidden / attention dropout p_drop) as ONE launch per direction — replaces model/model.py:30-126 (transformers' BertEncoder under .train()) between the
 * embedding Linear and the cls_head / residual heads.  csrc/kpf_trstack.hip: one workgroup per sample, weights streamed by dedicated loader waves from the
 * parameter tensors as they lie in memory ([N][K] fp32, 16-byte aligned).
 *   forward : H[0] = dropout(e + pos), then the four layers; e [B][21][128], pos [21][128]; param_table = DEVICE array of 4 x 16 pointers per layer in the order
 *             Wq bq Wk bk Wv bv Wo bo ln1.w ln1.b Wi bi Wo2 bo2 ln2.w ln2.b; everything the backward needs goes to `save`
 *             (kpf_tr_stack_save_floats(B) floats); the stack's output [B][21][128] is save + kpf_tr_stack_out_offset(B).
 *             rng: the device-resident (seed, counter) pair of kpf_attn21_forward; call0: first of 13 consecutive dropout call ids.
 *   backward: dh [B][21][128] -> dE [B][21][128] (gradient of e, and of pos after a sum over B); every Linear's dY goes to `dys`
 *             (kpf_tr_stack_dy_floats(B)) beside the X kept in `save` (kpf_tr_stack_offset(B, layer, which): which 0-3 = X of q|k|v / attention output /
 *             intermediate / output in `save`, 4-7 = dqkv [M][384] / d(attention output) / d(intermediate) [M][16] / d(output) in `dys`) for
 *             kpf_linear_wgrad_grouped; LayerNorm parameter gradients leave as per-sample partial sums parts[layer][ln][B][2][128]
 *             (kpf_tr_stack_part_floats(B); a kpf_colsum_desc with nblk = B, C = 128 each).  Dropout masks are recomputed from the (seed, counter) the
 *             forward stored in `save`.  Fixed summation order: bit-identical replays. */
/* Training (ABI 16): the decoder layer of the fusion block (updatedDecoder layer 3, model/transfusion_head.py:137-173, 437-554: cross attention of 21 query tokens over 21
 * key tokens, post-LN eps 1e-5, ReLU feed-forward of width 128, dropout) as one launch each way on the engine of kpf_tr_stack_train_*.  query / key [B][21][128];
 * param_table: DEVICE array of 14 pointers — in_proj_weight [384][128], in_proj_bias, out_proj.weight, out_proj.bias, norm2.weight, norm2.bias, linear1.weight, linear1.bias,
 * linear2.weight, linear2.bias, norm3.weight, norm3.bias, self_posembed [21][128], cross_posembed [21][128]; the output is save + kpf_xattn_train_offset(B, 5).
 * backward: dquery (both paths), dqe / dke = gradients of query + qpos / key + kpos (dkey = dke; the position tables' gradients are their sums over B); dys
 * (kpf_xattn_train_dy_floats) receives dqkv [M][384] | d out_proj [M][128] | d linear1 | d linear2 for the weight gradients (X operands: kpf_xattn_train_offset
 * 0 qe, 1 ke, 2 ctx, 3 x, 4 hidden); parts [2][B][2][128] the LayerNorm partial sums (norm2, norm3).  call0: first of 4 consecutive dropout call ids; mma as above. */
long kpf_xattn_train_save_floats(int B);
long kpf_xattn_train_dy_floats(int B);
long kpf_xattn_train_offset(int B, int which);
int kpf_xattn_train_forward(const float* query, const float* key, const void* param_table, float* save, long save_floats, int B, float p_drop, const long* rng,
                            int call0, int mma, void* stream);
int kpf_xattn_train_backward(const float* dout, const void* param_table, const float* save, float* dquery, float* dqe, float* dke, float* dys, float* parts, int B,
                             float p_drop, int call0, int mma, void* stream);
int kpf_tr_stack_set_stamps(void* stamps64 /* tuning aid: 64 device uint64 slots for in-kernel wall-clock stamps of workgroup 0; NULL = off */);
long kpf_tr_stack_save_floats(int B);
long kpf_tr_stack_out_offset(int B);
long kpf_tr_stack_dy_floats(int B);
long kpf_tr_stack_part_floats(int B);
long kpf_tr_stack_offset(int B, int layer, int which);
/* mma: 0 = fp32 products on v_mfma_f32_16x16x4_f32 (the fp32 step), 1 / 2 = operands rounded to bf16 / f16 in registers, fp32 accumulation (the mixed-precision step:
 * what torch.autocast does to these Linears); softmax, LayerNorm, GELU, residual sums and everything stored stay fp32. */
int kpf_tr_stack_train_forward(const float* e, const float* pos, const void* param_table, float* save, long save_floats, int B, float p_drop, const long* rng,
                               int call0, int mma, void* stream);
int kpf_tr_stack_train_backward(const float* dh, const void* param_table, const float* save, float* dE, float* dys, float* parts, int B, float p_drop,
                                int call0, int mma, void* stream);

/* Training: the two analytic maps of a fusion block (model/model.py:300-336) with their gradients towards the joints, one launch each:
 * hm[b][j][y][x] = GFM.joint2heatmap(uvd[..., :2], std, F, sigma) (util/generateFeature.py:584-600), duvd [B][J][3] (z component 0);
 * gam[b][j][p] = 1 / (10 |pix_xyz[b][p] - joint_xyz[b][j]|^2 + 1) (dataloader/loader.py:791-819), djoint [B][J][3].  fp32, contiguous. */
int kpf_joint_heatmap_forward(const float* uvd, float* hm, int B, int J, int F, float std_, float sigma, void* stream);
int kpf_joint_heatmap_backward(const float* uvd, const float* dhm, float* duvd, int B, int J, int F, float std_, float sigma, void* stream);
int kpf_geom_gate_forward(const float* pix_xyz, const float* joint_xyz, float* gam, int B, int J, int P, void* stream);
int kpf_geom_gate_backward(const float* pix_xyz, const float* joint_xyz, const float* dgam, float* djoint, int B, int J, int P, void* stream);

/* The same map with the joints given in crop coordinates uvd in [-1, 1]^3 (ABI 13): the uvd -> camera -> cube-normalised xyz transform of
 * dataloader/loader.py:775-789 (half_size = img_size / 2, flip = +-1) runs inside the kernel and the backward returns d/d(uvd).
 * par16 [B][16] = [M^-1 rows 0 and 1 (6) | fx fy u0 v0 (4) | centre xyz (3) | cube xyz (3)] per sample.  Replaces model/model.py:318-326. */
int kpf_geom_gate_uvd_forward(const float* pix_xyz, const float* joint_uvd, const float* par16, float* gam, int B, int J, int P, float half_size, float flip,
                              void* stream);
int kpf_geom_gate_uvd_backward(const float* pix_xyz, const float* joint_uvd, const float* par16, const float* dgam, float* djoint_uvd, int B, int J, int P,
                               float half_size, float flip, void* stream);

int kpf_conv_num_tile_cfgs(void);

const char* kpf_last_error(void);
/* Library/ABI version, bumped when a signature or the meaning of an argument changes (KPF_ABI_VERSION is what this header
 * describes; the Python binding refuses a library that reports another). */
#define KPF_ABI_VERSION 17
int kpf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
