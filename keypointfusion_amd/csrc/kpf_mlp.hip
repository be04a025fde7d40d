// Fused ConvNeXt MLP (convNeXT/convnext.py:44-51):   out = x + gamma * ( W2 · GELU(W1 · y + b1) + b2 )
// y = LayerNorm output [M][C], x = block input [M][C] (residual), W1 [4C][C], W2 [C][4C] in PyTorch layout, fp32.
//
// Why fuse: at C = 96/192 (the 64x64 and 32x32 stages of a 256x256 crop) the two GEMMs have K = C resp. N = C, only 3-6 K tiles
// per output tile, so tile fill, epilogue and the 4C-wide hidden tensor's HBM round trip (402 MB written + 402 MB read per
// block at B=64) cost as much as the MFMAs (measured 69-88 TF for the separate launches vs 100-119 TF for the large-K stages).
// Fused, the hidden activations never leave registers:
//   * a workgroup (4 waves) owns BM = 64*TM pixel rows; each WAVE owns 16*TM rows for ALL hidden and output channels, so no
//     cross-wave exchange is needed.
//   * the hidden dimension is processed in chunks of HC = 16*HT units.  GEMM1 (transposed issue: A operand = W1 fragment,
//     B operand = y fragment) leaves acc1[ht][tm] with lane (pixel = lane&15, hidden = 16*ht + 4*(lane>>4) + r): exactly the
//     activation-operand layout of a 16-deep k-step of GEMM2, so after bias+GELU the accumulator registers ARE the B operand
//     of GEMM2 (acc2[n][tm] += W2frag[n] x h) — no LDS, no shuffle.
//   * y tile stays in LDS for the whole tile; the W1 / W2 chunk pair is double-buffered and streamed by LDS-DMA
//     (global_load_lds_dwordx4) one chunk ahead; weights are L2-resident (295 KB at C=96).
//   * LDS images are row-major with the 16-byte chunk index XOR-swizzled per row (on the DMA source address) so that the
//     ds_read_b128 fragment reads of 16 consecutive rows hit 16 different bank groups.
#include "kpf_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;

__device__ __attribute__((aligned(16))) float kpf_mlp_zero16[4] = {0.f, 0.f, 0.f, 0.f};

struct MlpArgs {
  const float* y;
  const float* x;
  const float* w1;
  const float* b1;
  const float* w2;
  const float* b2;
  const float* gamma;
  float* out;
  const float* zero;
  int M;
};

__device__ __forceinline__ float gelu_f(float x) {  // same evaluation as kpf_conv.hip (A&S 7.1.26 erfc form, |err| <= 1.5e-7)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float q = 0.5f * x * (p * t) * e;
  return fmaxf(x, 0.f) - fabsf(q);
}

// row swizzle for rows of RC 16-byte chunks: makes 16 consecutive rows at one logical chunk land in 16 distinct 16-byte slots
// of the 256-byte bank row
template <int RC>
__device__ __forceinline__ int row_sw(int r) {
  static_assert(RC % 8 == 0, "row length must be a multiple of 32 floats");
  return (RC % 16 == 0) ? (r & 15) : ((r >> 1) & 7);
}

// DMA a [ROWS][RC chunks] image: LDS linear (row-major, swizzled chunk positions), source rows at src + row*src_ld (floats);
// rows >= valid_rows read the zero page.  All NT threads take part; ROWS*RC must be a multiple of NT.
template <int ROWS, int RC, int NT>
__device__ __forceinline__ void dma_image(float* lds_dst, const float* src, long src_ld, int valid_rows, const float* zero, int tid) {
  static_assert((ROWS * RC) % NT == 0, "image must be a whole number of workgroup-wide passes");
  constexpr int PASSES = ROWS * RC / NT;
  const int wave = tid >> 6;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int pos = p * NT + tid;  // 16-byte slot index in the LDS image
    const int r = pos / RC, cp = pos - r * RC;
    const int c = cp ^ row_sw<RC>(r);
    const float* s = r < valid_rows ? src + (long)r * src_ld + 4 * c : zero;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)s, (lds_void_t*)(lds_dst + (p * NT + wave * 64) * 4), 16, 0, 0);
  }
}

template <int NC, int TM, int HT, int NW>
__global__ __launch_bounds__(64 * NW) void convnext_mlp_kernel(const MlpArgs a) {
  constexpr int NT = 64 * NW;    // NW = 8: two waves per SIMD, so one wave's GELU (VALU) and LDS waits run under its partner's MFMAs
  constexpr int C = 16 * NC;     // channels
  constexpr int H4 = 4 * C;      // hidden width
  constexpr int HC = 16 * HT;    // hidden units per chunk
  constexpr int BM = 16 * TM * NW;  // pixel rows per workgroup
  constexpr int RCY = C / 4;     // 16-byte chunks per y / W1 row
  constexpr int RCW = HC / 4;    // 16-byte chunks per W2-chunk row
  constexpr int NCH = H4 / HC;   // chunks
  constexpr int YF = BM * C, W1F = HC * C, W2F = C * HC;  // floats per image
  static_assert(H4 % HC == 0, "chunking");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ys = lds;                   // [BM][C]
  float* W1s = Ys + YF;              // [2][HC][C]
  float* W2s = W1s + 2 * W1F;        // [2][C][HC]
  float* B1s = W2s + 2 * W2F;        // [4C] pwconv1 bias: read from LDS inside the chunk loop — an ordinary global load there would
                                     // make hipcc wait vmcnt(0), i.e. drain the in-flight weight DMA of the next chunk every chunk

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const long m0 = (long)blockIdx.x * BM;
  const int valid = (int)((a.M - m0) < BM ? (a.M - m0) : BM);

  dma_image<BM, RCY, NT>(Ys, a.y + m0 * C, C, valid, a.zero, tid);
  dma_image<HC, RCY, NT>(W1s, a.w1, C, HC, a.zero, tid);
  dma_image<C, RCW, NT>(W2s, a.w2, H4, C, a.zero, tid);

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc2[NC][TM];
#pragma unroll
  for (int n = 0; n < NC; ++n)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc2[n][j] = zero4;

  float* B2s = B1s + H4;             // [C] pwconv2 bias, [C] layer scale
  float* Gs = B2s + C;
  for (int i = tid; i < H4; i += NT) B1s[i] = a.b1[i];
  for (int i = tid; i < C; i += NT) {
    B2s[i] = a.b2[i];
    Gs[i] = a.gamma[i];
  }
  __syncthreads();

  const int yrow0 = wave * TM * 16 + fr;  // this lane's first y row
  for (int ch = 0; ch < NCH; ++ch) {
    const int cur = ch & 1;
    if (ch + 1 < NCH) {  // next chunk's weights fly into the other buffers under this chunk's MFMAs
      dma_image<HC, RCY, NT>(W1s + (cur ^ 1) * W1F, a.w1 + (long)(ch + 1) * HC * C, C, HC, a.zero, tid);
      dma_image<C, RCW, NT>(W2s + (cur ^ 1) * W2F, a.w2 + (long)(ch + 1) * HC, H4, C, a.zero, tid);
    }
    const float* w1b = W1s + cur * W1F;
    const float* w2b = W2s + cur * W2F;

    // ---- GEMM1: acc1[ht][tm] = W1chunk (HC x C) . y^T ----
    f32x4 acc1[HT][TM];
#pragma unroll
    for (int h = 0; h < HT; ++h)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc1[h][j] = zero4;
#pragma unroll
    for (int s = 0; s < NC; ++s) {  // 16-deep k-steps over the C input channels
      f32x4 yf[TM], wf[HT];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int r = yrow0 + j * 16;
        yf[j] = *reinterpret_cast<const f32x4*>(Ys + r * C + (((4 * s + fg) ^ row_sw<RCY>(r)) << 2));
      }
#pragma unroll
      for (int h = 0; h < HT; ++h) {
        const int r = h * 16 + fr;
        wf[h] = *reinterpret_cast<const f32x4*>(w1b + r * C + (((4 * s + fg) ^ row_sw<RCY>(r)) << 2));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc1[h][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[h][e], yf[j][e], acc1[h][j], 0, 0, 0);
    }

    // ---- bias + GELU in registers; GEMM2: acc2[n][tm] += W2chunk (C x HC) . h ----
#pragma unroll
    for (int h = 0; h < HT; ++h) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(B1s + ch * HC + h * 16 + 4 * fg);
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1[h][j][e] = gelu_f(acc1[h][j][e] + bv[e]);
      f32x4 wf[NC];
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        const int r = n * 16 + fr;
        wf[n] = *reinterpret_cast<const f32x4*>(w2b + r * HC + (((4 * h + fg) ^ row_sw<RCW>(r)) << 2));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc2[n][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][e], acc1[h][j][e], acc2[n][j], 0, 0, 0);
    }
    __syncthreads();  // next chunk's weights landed; everyone is done with the current buffers
  }

  // ---- epilogue: out = x + gamma * (acc2 + b2); lane owns channels n*16 + 4*fg .. +3 of pixel row.  `out` may alias `x`, so
  // the compiler will not move a residual load above an earlier store: issue all of a row's residual loads first, then store.
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const long m = m0 + wave * TM * 16 + j * 16 + fr;
    if (m >= a.M) continue;
    f32x4 xv[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) xv[n] = *reinterpret_cast<const f32x4*>(a.x + m * C + n * 16 + 4 * fg);
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = n * 16 + 4 * fg;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(B2s + c);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(Gs + c);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = xv[n][e] + gv[e] * (acc2[n][j][e] + bv[e]);
      *reinterpret_cast<f32x4*>(a.out + m * C + c) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same fusion on the f16 matrix cores with split operands (include/kpf.h): y arrives split from the depthwise+LayerNorm kernel,
// W1 / W2 are packed split, every 32-deep K step is 3 v_mfma_f32_16x16x32_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulate).
// GEMM1 leaves the lane with hidden units 16*ht + 4*fg + e of pixel fr; after bias + GELU those registers are split into f16 hi/lo
// and two neighbouring hidden tiles form the B operand of one 32-deep K step of GEMM2: k-slot (fg, j) is hidden
// 32q + 16*(j>>2) + 4*fg + (j&3).  W2 is packed with that order inside each 32-block (keypointfusion_amd/engine.py), so the hidden
// tensor still never leaves registers and no lane movement is needed.
// ---------------------------------------------------------------------------------------------------------------
struct MlpSplitArgs {
  const float* y;    // [M][C] split
  const float* x;    // [M][C] fp32 residual
  const float* w1;   // [4C][C] split, scaled by 1/us1
  const float* b1;
  const float* w2;   // [C][4C] split, hidden order permuted per 32-block, scaled by 1/us2
  const float* b2;
  const float* gamma;
  float* out;
  const float* zero;
  float us1, us2;
  int M;
};

// YLDS = false (C >= 192): the y tile is not staged in LDS — every lane reads its y fragments straight from global memory once
// (they live in registers for the whole tile anyway), which leaves the LDS to the W1 / W2 chunk ring.
template <int NC, int HT, int NW, bool YLDS, int TM>
__global__ __launch_bounds__(64 * NW) void convnext_mlp_split_kernel(const MlpSplitArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int C = 16 * NC;      // channels (multiple of 32)
  constexpr int KC = C / 32;      // K steps of GEMM1
  constexpr int H4 = 4 * C;
  constexpr int HC = 16 * HT;     // hidden units per chunk (multiple of 32)
  constexpr int HK = HC / 32;     // K steps of GEMM2 per chunk
  constexpr int BM = 16 * TM * NW;  // TM 16-pixel tiles per wave: every weight fragment read from LDS feeds TM x 3 MFMAs (at TM = 1 the
                                    // kernel needs ~340 B/clk of LDS reads per CU and is LDS-bound)
  constexpr int RCY = C / 4, RCW = HC / 4;
  constexpr int NCH = H4 / HC;
  constexpr int YF = YLDS ? BM * C : 0, W1F = HC * C, W2F = C * HC;
  static_assert(C % 32 == 0 && HC % 32 == 0 && H4 % HC == 0, "split blocks are 32 wide");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ys = lds;
  float* W1s = Ys + YF;
  float* W2s = W1s + 2 * W1F;
  float* B1s = W2s + 2 * W2F;
  float* B2s = B1s + H4;
  float* Gs = B2s + C;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const long m0 = (long)blockIdx.x * BM;
  const int valid = (int)((a.M - m0) < BM ? (a.M - m0) : BM);

  if (YLDS) dma_image<BM, RCY, NT>(Ys, a.y + m0 * C, C, valid, a.zero, tid);
  dma_image<HC, RCY, NT>(W1s, a.w1, C, HC, a.zero, tid);
  dma_image<C, RCW, NT>(W2s, a.w2, H4, C, a.zero, tid);

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc2[NC][TM];
#pragma unroll
  for (int n = 0; n < NC; ++n)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc2[n][j] = zero4;
  for (int i = tid; i < H4; i += NT) B1s[i] = a.b1[i];
  for (int i = tid; i < C; i += NT) {
    B2s[i] = a.b2[i];
    Gs[i] = a.gamma[i];
  }
  __syncthreads();

  // this lane's y fragments (hi, lo per K step and pixel tile) are the same for every chunk: read them once
  f16x8 yh[KC][TM], yl[KC][TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int yr = (wave * TM + j) * 16 + fr;
    if (YLDS) {
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        yh[k][j] = *reinterpret_cast<const f16x8*>(Ys + yr * C + (((8 * k + fg) ^ row_sw<RCY>(yr)) << 2));
        yl[k][j] = *reinterpret_cast<const f16x8*>(Ys + yr * C + (((8 * k + 4 + fg) ^ row_sw<RCY>(yr)) << 2));
      }
    } else {
      const long mr = (m0 + yr < a.M) ? m0 + yr : a.M - 1;  // rows beyond M repeat the last row; their results are never stored
      const float* yrow = a.y + mr * C;
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        yh[k][j] = *reinterpret_cast<const f16x8*>(yrow + ((8 * k + fg) << 2));
        yl[k][j] = *reinterpret_cast<const f16x8*>(yrow + ((8 * k + 4 + fg) << 2));
      }
    }
  }
  const float us1 = a.us1;

  for (int ch = 0; ch < NCH; ++ch) {
    const int cur = ch & 1;
    if (ch + 1 < NCH) {
      dma_image<HC, RCY, NT>(W1s + (cur ^ 1) * W1F, a.w1 + (long)(ch + 1) * HC * C, C, HC, a.zero, tid);
      dma_image<C, RCW, NT>(W2s + (cur ^ 1) * W2F, a.w2 + (long)(ch + 1) * HC, H4, C, a.zero, tid);
    }
    const float* w1b = W1s + cur * W1F;
    const float* w2b = W2s + cur * W2F;

    // ---- GEMM1: acc1[ht][tm] = W1chunk (HC x C) . y^T, 3 MFMAs per (hidden tile, pixel tile, K step) ----
    f32x4 acc1[HT][TM];
#pragma unroll
    for (int h = 0; h < HT; ++h)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc1[h][j] = zero4;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
#pragma unroll
      for (int h = 0; h < HT; ++h) {
        const int r = h * 16 + fr;
        const f16x8 wh = *reinterpret_cast<const f16x8*>(w1b + r * C + (((8 * k + fg) ^ row_sw<RCY>(r)) << 2));
        const f16x8 wl = *reinterpret_cast<const f16x8*>(w1b + r * C + (((8 * k + 4 + fg) ^ row_sw<RCY>(r)) << 2));
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          acc1[h][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, yh[k][j], acc1[h][j], 0, 0, 0);
          acc1[h][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, yl[k][j], acc1[h][j], 0, 0, 0);
          acc1[h][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, yh[k][j], acc1[h][j], 0, 0, 0);
        }
      }
    }

    // ---- bias + GELU, split in registers; GEMM2: acc2[n][tm] += W2chunk (C x HC) . h ----
#pragma unroll
    for (int q = 0; q < HK; ++q) {
      f16x8 hh[TM], hl[TM];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int h = 2 * q + t;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(B1s + ch * HC + h * 16 + 4 * fg);
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = gelu_f(fmaf(acc1[h][j][e], us1, bv[e]));
            const _Float16 hi = (_Float16)g;
            hh[j][4 * t + e] = hi;
            hl[j][4 * t + e] = (_Float16)(g - (float)hi);
          }
      }
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        const int r = n * 16 + fr;
        const f16x8 wh = *reinterpret_cast<const f16x8*>(w2b + r * HC + (((8 * q + fg) ^ row_sw<RCW>(r)) << 2));
        const f16x8 wl = *reinterpret_cast<const f16x8*>(w2b + r * HC + (((8 * q + 4 + fg) ^ row_sw<RCW>(r)) << 2));
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          acc2[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, hh[j], acc2[n][j], 0, 0, 0);
          acc2[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hl[j], acc2[n][j], 0, 0, 0);
          acc2[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hh[j], acc2[n][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  const float us2 = a.us2;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const long m = m0 + (wave * TM + j) * 16 + fr;
    if (m >= a.M) continue;
    f32x4 xv[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) xv[n] = *reinterpret_cast<const f32x4*>(a.x + m * C + n * 16 + 4 * fg);
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = n * 16 + 4 * fg;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(B2s + c);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(Gs + c);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = xv[n][e] + gv[e] * fmaf(acc2[n][j][e], us2, bv[e]);
      *reinterpret_cast<f32x4*>(a.out + m * C + c) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// 16-bit storage (bf16 / f16 activations and weights, fp32 accumulate): kpf_convnext_mlp_h16 (round 4).
//
// The stage-1/2 blocks of ConvNeXt-B at 512^2 (C = 128 / 256, 10^6 / 2.6 10^5 pixels) are HBM-bound as two GEMM launches: the 4C-wide hidden
// tensor is written and read back (2.1 GB per block at C = 128 against 0.8 GB for y, x and out together), and the ablation of the
// eight-phase GEMM (profiles/r04_g8_ablation.txt) shows that a tile's stores cannot be hidden behind its own matrix work.  Fused, the
// hidden tensor never leaves registers — same dataflow as convnext_mlp_split_kernel above, one MFMA per product:
//   * 4 waves per workgroup, two workgroups per CU (<= 256 registers, <= 80 KB of LDS): one workgroup's GELU (vector ALU), prologue and
//     epilogue run under the other's MFMAs; a wave owns PT pixel tiles of 16 for ALL hidden and output channels.
//   * y fragments come straight from global memory once per tile and stay in registers; the LDS holds only the weight ring: chunks of 32
//     hidden units = [32 x C] rows of W1 + [C x 32] of W2 (chunk-major, hidden order permuted per 32-block exactly like the split kernel's:
//     k-slot 8g + j <-> hidden 16 (j >> 2) + 4g + (j & 3)), R stages filled by LDS-DMA with a counted vmcnt (R - 2 chunks stay in flight
//     across the one raw barrier per chunk); every weight fragment read feeds PT MFMAs.
//   * per chunk and wave: GEMM1 2 x PT x C/32 MFMAs -> bias + GELU on 8 PT registers -> packed in place as GEMM2's B operand -> C/16 x PT MFMAs.
//   * epilogue: out = x + gamma * (acc + b2) in fp32 (x read in the accumulator layout), rounded once, staged through the wave's share of
//     the idle ring so that the global stores are whole 2C-byte rows.
// Measured and not adopted (round 4, C = 128, 10^6 pixels, tools/mlp16_bench.py; ablation with KPF_MLP16_DBG: of 484 us the GELU takes 114 — its
// vector-ALU floor —, the chunk-boundary waits / DMA issue 94, the stores 29): (i) a producer / consumer split (8 waves: waves 0-3 GEMM1 + GELU,
// handing the packed operand to waves 4-7 through LDS lane for lane, which run GEMM2 of the previous chunk; one register array reinterpreted per role):
// correct, 578 us — the producer's vector work issues at the single-wave rate (4 cycles per instruction) and the per-step barrier couples the two
// roles; (ii) 8 waves per workgroup (384 pixels, half the weight DMA per pixel, one workgroup per CU): 495 us; (iii) 2 pixel tiles per wave at 3
// workgroups per CU: 490 us, at 2 workgroups: 533 us.  The 4-wave, 3-tile, 2-workgroup form below stays (451-467 us, 590-610 TFLOP/s).
// ---------------------------------------------------------------------------------------------------------------------------------
struct Mlp16Args {
  const void* y;     // [M][C] LayerNorm output, 16-bit
  const void* x;     // [M][C] block input (residual), 16-bit
  const void* w1;    // [4C][C] 16-bit
  const float* b1;
  const void* w2c;   // [4C/32][C][32] 16-bit: chunk-major, permuted hidden order inside a chunk
  const float* b2;
  const float* gamma;
  void* out;         // [M][C] 16-bit (may alias x)
  int M;
  int dbg;           // tuning aid (KPF_MLP16_DBG): 1 = no GELU, 2 = no epilogue stores, 4 = no weight DMA / waits (garbage weights)
};

__device__ __forceinline__ float gelu_h16m(float x) { return kpf_gelu_h16(x); }  // (kpf_common.h: one definition for every 16-bit kernel)

template <int C, int PT, int R, bool BF, int WPS = 2, int NW = 4>
__global__ __launch_bounds__(64 * NW, WPS) void convnext_mlp_h16_kernel(const Mlp16Args a) {
  using TH = typename std::conditional<BF, bf16_t, f16_t>::type;
  constexpr int NT = 64 * NW;
  constexpr int KC = C / 32;        // K steps of GEMM1
  constexpr int NCT = C / 16;       // output-channel tiles of GEMM2
  constexpr int NCH = 4 * C / 32;   // chunks of 32 hidden units
  constexpr int W1B = 32 * C * 2;   // bytes of a chunk's W1 rows (32 rows of 2C bytes)
  constexpr int W2B = C * 64;       // bytes of a chunk's W2 rows (C rows of 64 bytes)
  constexpr int CHB = W1B + W2B;    // one ring stage
  constexpr int G = CHB / (NT * 16);  // DMA instructions per wave per chunk
  constexpr int RC1 = C / 8;        // 16-byte chunks per W1 row
  constexpr int BM = 16 * PT * NW;
  static_assert(C % 128 == 0 && R >= 2 && R <= 4 && W1B % (NT * 16) == 0 && W2B % (NT * 16) == 0, "shape");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const LB = reinterpret_cast<char*>(lds);
  constexpr int RING_B = R * CHB > NW * PT * 16 * 2 * C ? R * CHB : NW * PT * 16 * 2 * C;  // (the epilogue's staging area lives in the ring's bytes)
  float* const B1s = reinterpret_cast<float*>(LB + RING_B);
  float* const B2s = B1s + 4 * C;
  float* const Gs = B2s + C;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const long m0 = (long)blockIdx.x * BM;

  // ---- this lane's y fragments (B operand of GEMM1: k = 32 ks + 8 fg .. + 7 of pixel fr), once per tile, and the small tables ----
  f16x8 yf[KC][PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    long m = m0 + (wave * PT + pt) * 16 + fr;
    m = m < a.M ? m : a.M - 1;  // (rows beyond M repeat the last row; their results are never stored)
    const TH* yrow = reinterpret_cast<const TH*>(a.y) + m * C + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) yf[ks][pt] = *reinterpret_cast<const f16x8*>(yrow + 32 * ks);
  }
  for (int i = tid; i < 4 * C; i += NT) B1s[i] = a.b1[i];
  for (int i = tid; i < C; i += NT) {
    B2s[i] = a.b2[i];
    Gs[i] = a.gamma[i];
  }
  __syncthreads();  // the tables are visible to every wave (a raw s_barrier, as used in the ring below, would NOT wait for these ds_writes: lgkmcnt)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ordinary loads are done before the first LDS-DMA is issued: from here on vmcnt counts DMAs only

  // ---- weight ring ----
  // W1 image: row r (0..31) of 2C bytes, 16-byte chunk c at position c ^ (r & 15) (low four bits); W2 image: row r (0..C-1) of 64 bytes, chunk c at
  // c ^ ((r >> 2) & 3).  The DMA writes LDS linearly, so the permutation is applied to each lane's SOURCE address.
  const TH* const w1g = reinterpret_cast<const TH*>(a.w1);
  const TH* const w2g = reinterpret_cast<const TH*>(a.w2c);
  // element offsets of this lane's sources inside a chunk: pass p moves LDS slots p * 256 + tid.  W1 (rows of RC1 chunks): row r = p * 256 / RC1 +
  // tid / RC1, so with RC1 = 16 the swizzle term (r & 15) does not depend on p, with RC1 = 32 it alternates between two values; W2 (4 chunks per row):
  // row r = 64 p + tid / 4 and ((r >> 2) & 3) does not depend on p.  So one or two offsets per image plus compile-time strides.
  constexpr int P1 = W1B / (NT * 16), P2 = W2B / (NT * 16), RPP1 = NT / RC1;  // passes; W1 rows per pass
  const int r1 = tid / RC1, cp1 = tid % RC1;
  const int s1a = r1 * C + 8 * (cp1 ^ (r1 & 15)), s1b = r1 * C + 8 * (cp1 ^ ((r1 + RPP1) & 15));  // even / odd passes (equal when RPP1 = 16)
  const int s2 = (tid >> 2) * 32 + 8 * ((tid & 3) ^ ((tid >> 4) & 3));
  auto stage = [&](int ch, int slot) {
    char* dst = LB + slot * CHB + wave * 1024;
    const TH* const g1 = w1g + (long)ch * 32 * C;
    const TH* const g2 = w2g + (long)ch * C * 32;
#pragma unroll
    for (int p = 0; p < P1; ++p)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(g1 + ((p & 1) ? s1b : s1a) + p * RPP1 * C), (lds_void_t*)(dst + p * NT * 16), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < P2; ++p)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(g2 + s2 + p * (NT / 4) * 32), (lds_void_t*)(dst + W1B + p * NT * 16), 16, 0, 0);
  };
  if (!(a.dbg & 4))
#pragma unroll
    for (int c = 0; c < R - 1; ++c) stage(c, c);

  f32x4 acc[NCT][PT];
#pragma unroll
  for (int n = 0; n < NCT; ++n)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[n][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment read offsets (bytes inside a stage): W1 row 16 ht + fr, chunk (4 ks + fg) ^ fr; W2 row 16 ct + fr, chunk fg ^ ((fr >> 2) & 3)
  const int w1_rd = fr * (2 * C), w2_rd = W1B + fr * 64 + ((fg ^ ((fr >> 2) & 3)) << 4);

  auto chunk = [&](int ch, auto SLOT) {
    constexpr int slot = decltype(SLOT)::value;
    // chunk ch has landed once at most the younger chunks' DMAs are outstanding: R - 2 in steady state, fewer at the tail
    const int younger = NCH - 1 - ch;
    if (a.dbg & 4) {
    } else if (younger >= R - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * G) : "memory");
    else if (R > 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's part of chunk ch is in LDS; everyone has finished reading chunk ch - 1's stage
    if (ch + R - 1 < NCH && !(a.dbg & 4)) stage(ch + R - 1, (slot + R - 1) % R);
    const char* const sb = LB + slot * CHB;
    f32x4 d1[2][PT];
#pragma unroll
    for (int ht = 0; ht < 2; ++ht)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) d1[ht][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KC; ++ks)
#pragma unroll
      for (int ht = 0; ht < 2; ++ht) {
        const f16x8 w = *reinterpret_cast<const f16x8*>(sb + w1_rd + ht * 16 * (2 * C) + ((((4 * ks + fg) ^ fr) & (RC1 - 1)) << 4));
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          if constexpr (BF) d1[ht][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, yf[ks][pt]), d1[ht][pt], 0, 0, 0);
          else d1[ht][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, yf[ks][pt], d1[ht][pt], 0, 0, 0);
        }
      }
    // bias + GELU; the lane's 8 values of a pixel (hidden 16 ht + 4 fg + e) are one B-operand fragment of GEMM2 in the packed hidden order
    const f32x4 bv0 = *reinterpret_cast<const f32x4*>(B1s + ch * 32 + 4 * fg);
    const f32x4 bv1 = *reinterpret_cast<const f32x4*>(B1s + ch * 32 + 16 + 4 * fg);
    f16x8 hf[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      typename std::conditional<BF, bf16x8, f16x8>::type h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = (TH)((a.dbg & 1) ? d1[0][pt][e] + bv0[e] : gelu_h16m(d1[0][pt][e] + bv0[e]));
        h[4 + e] = (TH)((a.dbg & 1) ? d1[1][pt][e] + bv1[e] : gelu_h16m(d1[1][pt][e] + bv1[e]));
      }
      hf[pt] = __builtin_bit_cast(f16x8, h);
    }
#pragma unroll
    for (int n = 0; n < NCT; ++n) {
      const f16x8 w = *reinterpret_cast<const f16x8*>(sb + w2_rd + n * 16 * 64);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        if constexpr (BF) acc[n][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, hf[pt]), acc[n][pt], 0, 0, 0);
        else acc[n][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, hf[pt], acc[n][pt], 0, 0, 0);
      }
    }
  };
  static_assert(NCH % R == 0, "whole ring revolutions");
  for (int cb = 0; cb < NCH; cb += R) {
    chunk(cb, std::integral_constant<int, 0>{});
    chunk(cb + 1, std::integral_constant<int, 1 % R>{});
    if constexpr (R > 2) chunk(cb + 2, std::integral_constant<int, 2 % R>{});
    if constexpr (R > 3) chunk(cb + 3, std::integral_constant<int, 3 % R>{});
  }
  __builtin_amdgcn_s_barrier();  // every wave is done with the ring (no DMA in flight: the last chunk waited vmcnt(0)): it becomes the staging area

  // ---- epilogue ----
  const TH* const xg = reinterpret_cast<const TH*>(a.x);
  TH* const og = reinterpret_cast<TH*>(a.out);
  constexpr int ROWB = 2 * C;  // staging row (bytes), chunk index XOR-swizzled by (row & 15) in its low four bits
  char* const stg = LB + wave * (PT * 16 * ROWB);
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const long m = m0 + (wave * PT + pt) * 16 + fr;
    const long mr = m < a.M ? m : a.M - 1;
    f32x4 xv[NCT];
#pragma unroll
    for (int n = 0; n < NCT; ++n) xv[n] = kpf_ld4(xg + mr * C + n * 16 + 4 * fg);
#pragma unroll
    for (int n = 0; n < NCT; ++n) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(B2s + n * 16 + 4 * fg);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(Gs + n * 16 + 4 * fg);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = xv[n][e] + gv[e] * (acc[n][pt][e] + bv[e]);
      const int row = pt * 16 + fr, cidx = 2 * n + (fg >> 1);
      kpf_st4(reinterpret_cast<TH*>(stg + row * ROWB + ((cidx ^ fr) << 4) + (fg & 1) * 8), v);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (own region only)
  constexpr int CPR = ROWB / 16, RPI = 64 / CPR;  // chunks per row, rows per store instruction
  const int rl = lane / CPR, p = lane % CPR;
#pragma unroll
  for (int it = 0; it < PT * 16 / RPI; ++it) {
    const int row = it * RPI + rl;
    const int c = p ^ (row & 15);
    const f32x4 q = *reinterpret_cast<const f32x4*>(stg + row * ROWB + (p << 4));
    const long m = m0 + wave * PT * 16 + row;
    if (m < a.M && !(a.dbg & 2)) *reinterpret_cast<f32x4*>(og + m * C + 8 * c) = q;
  }
}

template <int C, int PT, int R, int WPS = 2, int NW = 4>
int launch_mlp_h16(const Mlp16Args& a, int dtype, hipStream_t st) {
  constexpr int CHB = 32 * C * 2 + C * 64, BM = 16 * PT * NW;
  constexpr int RING_B = R * CHB > NW * PT * 16 * 2 * C ? R * CHB : NW * PT * 16 * 2 * C;
  const size_t lds = (size_t)RING_B + 6 * C * sizeof(float);
  void (*kern)(const Mlp16Args) = dtype == KPF_DT_BF16 ? convnext_mlp_h16_kernel<C, PT, R, true, WPS, NW> : convnext_mlp_h16_kernel<C, PT, R, false, WPS, NW>;
  static std::atomic<bool> lds_opt_in[2][KPF_MAX_DEVICES];
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(kern), lds_opt_in[dtype == KPF_DT_BF16 ? 1 : 0])) {
    kpf_set_error("kpf_convnext_mlp_h16: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  const long tiles = ((long)a.M + BM - 1) / BM;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64 * NW), lds, st, a);
  return kpf_check_launch("kpf_convnext_mlp_h16");
}

template <int NC, int HT, int NW, bool YLDS, int TM>
int launch_mlp_split(MlpSplitArgs& a, hipStream_t st) {
  constexpr int C = 16 * NC, HC = 16 * HT, BM = 16 * TM * NW;
  const size_t lds = (size_t)((YLDS ? BM * C : 0) + 2 * HC * C + 2 * C * HC + 6 * C) * sizeof(float);
  auto kern = convnext_mlp_split_kernel<NC, HT, NW, YLDS, TM>;
  static std::atomic<bool> lds_opt_in[KPF_MAX_DEVICES];  // per device: the attribute does not carry over to other GPUs of the process
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(kern), lds_opt_in)) {
    kpf_set_error("kpf_convnext_mlp_split_f32: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  const long tiles = (a.M + BM - 1) / BM;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64 * NW), lds, st, a);
  return kpf_check_launch("kpf_convnext_mlp_split_f32");
}

template <int NC, int TM, int HT, int NW>
int launch_mlp(MlpArgs& a, hipStream_t st) {
  constexpr int C = 16 * NC, HC = 16 * HT, BM = 16 * TM * NW;
  const size_t lds = (size_t)(BM * C + 2 * HC * C + 2 * C * HC + 6 * C) * sizeof(float);
  auto kern = convnext_mlp_kernel<NC, TM, HT, NW>;
  static std::atomic<bool> lds_opt_in[KPF_MAX_DEVICES];  // per device: the attribute does not carry over to other GPUs of the process
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(kern), lds_opt_in)) {
    kpf_set_error("kpf_convnext_mlp_f32: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  const long tiles = (a.M + BM - 1) / BM;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64 * NW), lds, st, a);
  return kpf_check_launch("kpf_convnext_mlp_f32");
}

}  // namespace

// 192 is implemented (4-wave variant: a 128-row y tile does not fit beside the weight ring) but measured slower than the two plain
// GEMM launches at that width (93 vs 102 TF), so it is not advertised; kpf_convnext_mlp_f32 still accepts it.
extern "C" int kpf_convnext_mlp_supported(int C) { return C == 96 || C == 128; }

extern "C" int kpf_convnext_mlp_f32(const float* y, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                    const float* gamma, float* out, long M, int C, void* stream) {
  KPF_REQUIRE(y && x && w1 && b1 && w2 && b2 && gamma && out && M > 0, "kpf_convnext_mlp_f32: null pointer / empty");
  KPF_REQUIRE(C == 96 || C == 128 || C == 192, "kpf_convnext_mlp_f32: C=%d not supported (96, 128, 192)", C);
  KPF_REQUIRE(M < (1l << 31), "kpf_convnext_mlp_f32: too many rows");
  KPF_REQUIRE(kpf_aligned16(y) && kpf_aligned16(x) && kpf_aligned16(w1) && kpf_aligned16(w2) && kpf_aligned16(out) && kpf_aligned16(b1) &&
                  kpf_aligned16(b2) && kpf_aligned16(gamma),
              "kpf_convnext_mlp_f32: pointers must be 16-byte aligned");
  MlpArgs a;
  a.y = y; a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.gamma = gamma; a.out = out; a.M = (int)M;
  {
    static const float* zero_of_dev[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!zero_of_dev[dev]) {
      void* p = nullptr;
      if (hipGetSymbolAddress(&p, HIP_SYMBOL(kpf_mlp_zero16)) != hipSuccess || !p) {
        kpf_set_error("kpf_convnext_mlp_f32: cannot resolve the zero page");
        return KPF_ELAUNCH;
      }
      zero_of_dev[dev] = static_cast<const float*>(p);
    }
    a.zero = zero_of_dev[dev];
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (C) {
    case 96: return launch_mlp<6, 1, 4, 8>(a, st);    // 8 waves, BM 128, HC 64: 48 + 96 KB LDS
    case 128: return launch_mlp<8, 1, 2, 8>(a, st);   // 8 waves, BM 128, HC 32: 64 + 64 KB
    default: return launch_mlp<12, 1, 2, 4>(a, st);   // C = 192: 4 waves, BM 64, HC 32: 48 + 96 KB (a 128-row y tile would not fit)
  }
}

static const float* mlp_zero_page() {
  static const float* zero_of_dev[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!zero_of_dev[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(kpf_mlp_zero16)) != hipSuccess || !p) return nullptr;
    zero_of_dev[dev] = static_cast<const float*>(p);
  }
  return zero_of_dev[dev];
}

extern "C" int kpf_convnext_mlp_h16_supported(int C) { return C == 128 || C == 256; }

extern "C" int kpf_convnext_mlp_h16(const void* y, const void* x, const void* w1, const float* b1, const void* w2_chunks, const float* b2,
                                    const float* gamma, void* out, long M, int C, int dtype, void* stream) {
  KPF_REQUIRE(y && x && w1 && b1 && w2_chunks && b2 && gamma && out && M > 0, "kpf_convnext_mlp_h16: null pointer / empty");
  KPF_REQUIRE(kpf_convnext_mlp_h16_supported(C), "kpf_convnext_mlp_h16: C=%d not supported (128, 256)", C);
  KPF_REQUIRE(dtype == KPF_DT_BF16 || dtype == KPF_DT_F16, "kpf_convnext_mlp_h16: dtype must be KPF_DT_BF16 or KPF_DT_F16");
  KPF_REQUIRE(M < (1l << 31), "kpf_convnext_mlp_h16: too many rows");
  KPF_REQUIRE(kpf_aligned16(y) && kpf_aligned16(x) && kpf_aligned16(w1) && kpf_aligned16(w2_chunks) && kpf_aligned16(out) && kpf_aligned16(b1) &&
                  kpf_aligned16(b2) && kpf_aligned16(gamma),
              "kpf_convnext_mlp_h16: pointers must be 16-byte aligned");
  Mlp16Args a;
  a.y = y; a.x = x; a.w1 = w1; a.b1 = b1; a.w2c = w2_chunks; a.b2 = b2; a.gamma = gamma; a.out = out; a.M = (int)M;
  static const int dbg = []() { const char* e = getenv("KPF_MLP16_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static const int cfg = []() { const char* e = getenv("KPF_MLP16_CFG"); return e ? atoi(e) : 0; }();  // tuning aid
  if (C == 128 && cfg == 5) return launch_mlp_h16<128, 3, 4, 2, 8>(a, dtype, st);  // 8 waves, 384 pixels per workgroup, one workgroup per CU: slower (495 vs 451 us)
  if (C == 128) return launch_mlp_h16<128, 3, 4>(a, dtype, st);  // 48 pixels per wave, 4-stage ring of 16-KB chunks (67 KB of LDS)
  return launch_mlp_h16<256, 2, 2>(a, dtype, st);                 // 32 pixels per wave, 2 stages of 32 KB (70 KB)
}

extern "C" int kpf_convnext_mlp_split_supported(int C) { return C == 96 || C == 128 || C == 192 || C == 256; }

extern "C" int kpf_convnext_mlp_split_f32(const float* y_split, const float* x, const float* w1_split, const float* b1, float w1_unscale,
                                          const float* w2_split_perm, const float* b2, float w2_unscale, const float* gamma, float* out, long M,
                                          int C, void* stream) {
  KPF_REQUIRE(y_split && x && w1_split && b1 && w2_split_perm && b2 && gamma && out && M > 0, "kpf_convnext_mlp_split_f32: null pointer / empty");
  KPF_REQUIRE(kpf_convnext_mlp_split_supported(C), "kpf_convnext_mlp_split_f32: C=%d not supported (96, 128, 192, 256)", C);
  KPF_REQUIRE(M < (1l << 31) && w1_unscale > 0.f && w2_unscale > 0.f, "kpf_convnext_mlp_split_f32: bad M / scales");
  KPF_REQUIRE(kpf_aligned16(y_split) && kpf_aligned16(x) && kpf_aligned16(w1_split) && kpf_aligned16(w2_split_perm) && kpf_aligned16(out) &&
                  kpf_aligned16(b1) && kpf_aligned16(b2) && kpf_aligned16(gamma),
              "kpf_convnext_mlp_split_f32: pointers must be 16-byte aligned");
  MlpSplitArgs a;
  a.y = y_split; a.x = x; a.w1 = w1_split; a.b1 = b1; a.w2 = w2_split_perm; a.b2 = b2; a.gamma = gamma; a.out = out; a.M = (int)M;
  a.us1 = w1_unscale; a.us2 = w2_unscale;
  a.zero = mlp_zero_page();
  KPF_REQUIRE(a.zero, "kpf_convnext_mlp_split_f32: cannot resolve the zero page");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static const int tm_env = []() { const char* e = getenv("KPF_MLP_TM"); return e ? atoi(e) : 2; }();  // tuning aid
  if (tm_env == 1) {
    if (C == 96) return launch_mlp_split<6, 4, 8, true, 1>(a, st);    // BM 128, HC 64: 48 + 96 KB LDS
    if (C == 128) return launch_mlp_split<8, 2, 8, true, 1>(a, st);   // BM 128, HC 32: 64 + 64 KB
    if (C == 192) return launch_mlp_split<12, 2, 8, false, 1>(a, st);  // y fragments from global; HC 32: 96 KB of weight chunks
    return launch_mlp_split<16, 2, 8, false, 1>(a, st);                // C = 256: 128 KB of weight chunks
  }
  // two pixel tiles per wave (BM 256): y fragments from global memory, the LDS holds only the weight ring
  if (C == 96) return launch_mlp_split<6, 4, 8, false, 2>(a, st);
  if (C == 128) return launch_mlp_split<8, 2, 8, false, 2>(a, st);
  if (C == 192) return launch_mlp_split<12, 2, 8, false, 2>(a, st);
  return launch_mlp_split<16, 2, 8, false, 1>(a, st);  // C = 256: registers only allow one tile per wave
}
