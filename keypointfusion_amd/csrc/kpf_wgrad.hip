// Weight gradients of the training step on gfx950.
//
// (1) wgrad_f32_kernel — dense convolution / Linear:
//         dW[n][(ky,kx,c)] = sum over pixels m = (b,oy,ox) of dY[m][n] * X[b][oy*sh+ky-ph][ox*sw+kx-pw][c]
//     = a GEMM whose REDUCTION dimension is the pixel index: both operands are stored with the reduction index as the slow (row)
//     dimension (dY rows are n-contiguous, X pixels c-contiguous), which is exactly the fragment order of v_mfma_f32_16x16x4_f32
//     when LDS holds the tiles as they are in memory: lane l reads 16 bytes at row (l/16 + 4*kstep), column 4*(l%16) and gets the
//     operands of four MFMA tiles at once (MFMA row r of tile i <-> n = 4r+i: a permutation of the OUTPUT only).  No transposes
//     anywhere; the tiles are staged by LDS-DMA (global_load_lds_dwordx4) straight from dY and from the im2col addresses of X
//     (out-of-image / out-of-range chunks read a zero page).  The pixel range is split over gridDim.y workgroups per output tile
//     (split-K); partial tiles go to a workspace and wgrad_reduce_kernel sums them in a fixed order (deterministic, no atomics)
//     while permuting (ky,kx,c) -> the reference's OIHW weight layout.  db = sum_m dY[m][n] falls out of the dY fragments.
//     With 16-bit dY / X (mixed-precision step) the same kernel stages through registers instead (16-byte loads of 8 values in
//     flight under the previous tile's MFMAs, widened to fp32 on the way into LDS): kpf_conv2d_wgrad_h16.
// (2) dwconv7_wgrad_kernel — depthwise 7x7: 49 x C outputs, HBM/L2-bound; thread = (4 channels, tap row ky), sliding 14-pixel
//     window in registers, partial sums per (image, row band) chunk, same fixed-order reduce.
// (3) bn_*_kernel — train-mode BatchNorm (+ ReLU) forward / backward on NHWC rows, fp32 or 16-bit storage on either side
//     (kpf_bn_train_forward / kpf_bn_train_backward), at the end of this file.
#include "kpf_common.h"

#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;

__device__ __attribute__((aligned(16))) float wg_zero16[4] = {0.f, 0.f, 0.f, 0.f};

struct WgradArgs {
  const void* dy;    // [M][ldy]                 fp32, or the 16-bit storage type of the mixed-precision step (TIN)
  const void* x;     // NHWC [B][H][W][ldx]
  float* part;       // [S][N][K]
  float* dbpart;     // [S][N] or nullptr
  const float* zero;
  int H, W, Cin, ldx, OH, OW, N, ldy, KH, KW, sh, sw, ph, pw;
  int M, K, tilesK, stages_per_split;
  int is1x1;
  int xcd;       // wgrad_h16s_kernel: XCD-aware tile order (xcd_slab_remap)
  int emul_sps;  // grouped form only: stages per split of the split + reduce form whose summation order it reproduces
  // channel-grouped launch (kpf_conv2d_wgrad_groups, grid.z = group): group g reads dy + g*g_dy / x + g*g_x (elements) and writes part + g*g_part,
  // dbpart + g*g_db (floats).  All zero for an ordinary launch (blockIdx.z is 0 there).
  int groups, g_dy, g_x;
  long g_part, g_db;
};

template <typename TIN>
__device__ __forceinline__ WgradArgs wg_group_args(const WgradArgs& a0) {
  WgradArgs a = a0;
  const int g = blockIdx.z;
  a.dy = static_cast<const TIN*>(a0.dy) + g * a0.g_dy;
  a.x = static_cast<const TIN*>(a0.x) + g * a0.g_x;
  a.part += g * a0.g_part;
  if (a.dbpart) a.dbpart += g * a0.g_db;
  return a;
}

constexpr int RB = 32;  // pixels (reduction rows) per LDS stage

// 8 consecutive 16-bit values (one 16-byte load) -> 8 floats
__device__ __forceinline__ void unpack8(const uint4 u, bf16_t, f32x4& lo, f32x4& hi) {
  lo = f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
  hi = f32x4{__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
}
__device__ __forceinline__ void unpack8(const uint4 u, f16_t, f32x4& lo, f32x4& hi) {
  const f16x8 h = __builtin_bit_cast(f16x8, u);
  lo = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  hi = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
}

__device__ __forceinline__ void unpack8(const uint4, float, f32x4&, f32x4&) {}  // (fp32 operands take the DMA path: never called)

// TIN = float: operands staged by LDS-DMA.  TIN = bf16_t / f16_t (mixed-precision training: dY and the saved activations are 16-bit):
// 16-byte global loads of 8 values into registers while the previous tile is multiplied, widened to fp32 on the way into LDS — the
// MFMAs, accumulation and reduction are the fp32 ones (the master weight's gradient is not rounded), and the separate
// 16-bit -> fp32 passes over dY and X disappear.
// R16 (round 6; fp32 operands only): 0 = fp32 products; 1 / 2 = the operands rounded to bf16 / f16 in registers, four reduction rows per v_mfma_f32_16x16x16 (the weight
// gradient of a Linear whose forward multiplied rounded operands: DESA's wide Linears in the mixed-precision step); accumulation, split and reduce unchanged.
template <int VN, int VK, typename TIN, bool GROUPED = false, int R16 = 0>
__device__ __forceinline__ void wgrad_f32_body(const WgradArgs& a, const int bx, const int split) {
  constexpr bool DMA = std::is_same<TIN, float>::value;
  constexpr int BN = 32 * VN, BK = 32 * VK;     // output tile; 2 x 2 waves, wave tile (16 VN) x (16 VK)
  constexpr int GA = BN / 4, GB = BK / 4;       // 16-byte granules per staged row
  constexpr int RA = 64 / GA, RBW = 64 / GB;    // rows one wave-DMA (64 lanes x 16 B) covers
  constexpr int TILE = RB * (BN + BK);
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][TILE]: A = dY tile [RB][BN], B = X tile [RB][BK]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wk = wave >> 1;
  const int nt = bx / a.tilesK, kt = bx % a.tilesK;
  const int n0 = nt * BN, k0 = kt * BK;
  const int total_stages = (a.M + RB - 1) / RB;
  const int s_begin = split * a.stages_per_split;
  const int ns = min(a.stages_per_split, total_stages - s_begin);
  const int m_begin = s_begin * RB;

  // --- per-thread staging constants ---------------------------------------------------------------------------------------
  const int ga = lane % GA, ra = lane / GA;     // A: granule column / row inside one wave-DMA
  const int gb = lane % GB, rb = lane / GB;
  const bool a_ok = n0 + 4 * ga < a.N;
  const float* a_src = static_cast<const float*>(a.dy) + n0 + 4 * ga;
  const float* xf = static_cast<const float*>(a.x);
  const int kcol = k0 + 4 * gb;
  const bool b_ok = kcol < a.K;
  const int tap = kcol / a.Cin, cch = kcol - tap * a.Cin;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
  const int ohw = a.OH * a.OW;

  auto stage = [&](int s, float* buf) {
    const int mb = m_begin + s * RB;
#pragma unroll
    for (int i = 0; i < VN; ++i) {  // A: VN wave-DMAs per wave
      const int d = i * 4 + wave;
      const int m = mb + d * RA + ra;
      const float* src = (a_ok && m < a.M) ? a_src + (size_t)m * a.ldy : a.zero;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(buf + d * 256), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < VK; ++i) {
      const int d = i * 4 + wave;
      const int m = mb + d * RBW + rb;
      const float* src = a.zero;
      if (b_ok && m < a.M) {
        if (a.is1x1) {
          src = xf + (size_t)m * a.ldx + cch;
        } else {
          const int b = m / ohw, r = m - b * ohw;
          const int oy = r / a.OW, ox = r - oy * a.OW;
          const int iy = oy * a.sh + ky - a.ph, ix = ox * a.sw + kx - a.pw;
          if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) src = xf + ((size_t)(b * a.H + iy) * a.W + ix) * a.ldx + cch;
        }
      }
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(buf + RB * BN + d * 256), 16, 0, 0);
    }
  };

  typedef float fvn __attribute__((ext_vector_type(VN)));
  typedef float fvk __attribute__((ext_vector_type(VK)));
  f32x4 acc[VN][VK];
#pragma unroll
  for (int i = 0; i < VN; ++i)
#pragma unroll
    for (int j = 0; j < VK; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  fvn dbs = 0.f;
  const bool want_db = a.dbpart != nullptr && kt == 0 && wk == 0;
  // GROUPED (one workgroup walks every stage): the stages are still summed split by split — a fresh accumulator per emul_sps stages, the
  // split sums added in split order — which is exactly what the split kernels + wgrad_reduce_kernel compute: same bits either way
  f32x4 tot[VN][VK];
  float dbt[VN];
#pragma unroll
  for (int i = 0; i < VN; ++i) {
    dbt[i] = 0.f;
#pragma unroll
    for (int j = 0; j < VK; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto fold = [&]() {
#pragma unroll
    for (int i = 0; i < VN; ++i) {
#pragma unroll
      for (int j = 0; j < VK; ++j) {
        tot[i][j] += acc[i][j];
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      float v = dbs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      dbt[i] += v;
      dbs[i] = 0.f;
    }
  };

  const int fr = lane >> 4, fc = lane & 15;  // fragment row inside a 4-pixel k-step / 16-lane column index
  const int a_off = fr * BN + wn * (16 * VN) + VN * fc;
  const int b_off = RB * BN + fr * BK + wk * (16 * VK) + VK * fc;

  // ---- 16-bit operands: register staging (8-element granules) ----
  constexpr int GA8 = BN / 8, GB8 = BK / 8;          // granules per staged row
  constexpr int NA = RB * GA8 / 256 > 0 ? RB * GA8 / 256 : 1, NB = RB * GB8 / 256 > 0 ? RB * GB8 / 256 : 1;
  const TIN* dyh = static_cast<const TIN*>(a.dy);
  const TIN* xh = static_cast<const TIN*>(a.x);
  const int gcolb = tid % GB8;                        // (256 % GB8 == 0: a thread's X granule column is the same for every pass)
  const int kcol8 = k0 + 8 * gcolb;
  const bool b8_ok = kcol8 < a.K;
  const int tap8 = kcol8 / a.Cin, cch8 = kcol8 - tap8 * a.Cin;
  const int ky8 = tap8 / a.KW, kx8 = tap8 - ky8 * a.KW;
  uint4 rga[NA], rgb[NB];
  auto load_regs = [&](int s) {
    const int mb = m_begin + s * RB;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int g = i * 256 + tid, row = g / GA8, gc = g % GA8;
      const int m = mb + row, n = n0 + 8 * gc;
      uint4 v = {0u, 0u, 0u, 0u};
      if (row < RB && m < a.M && n < a.N) v = *reinterpret_cast<const uint4*>(dyh + (size_t)m * a.ldy + n);
      rga[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int g = i * 256 + tid, row = g / GB8;
      const int m = mb + row;
      uint4 v = {0u, 0u, 0u, 0u};
      if (row < RB && b8_ok && m < a.M) {
        if (a.is1x1) {
          v = *reinterpret_cast<const uint4*>(xh + (size_t)m * a.ldx + cch8);
        } else {
          const int b = m / ohw, r = m - b * ohw;
          const int oy = r / a.OW, ox = r - oy * a.OW;
          const int iy = oy * a.sh + ky8 - a.ph, ix = ox * a.sw + kx8 - a.pw;
          if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
            v = *reinterpret_cast<const uint4*>(xh + ((size_t)(b * a.H + iy) * a.W + ix) * a.ldx + cch8);
        }
      }
      rgb[i] = v;
    }
  };
  auto write_lds = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int g = i * 256 + tid, row = g / GA8, gc = g % GA8;
      if (row < RB) {
        f32x4 lo, hi;
        unpack8(rga[i], TIN(), lo, hi);
        *reinterpret_cast<f32x4*>(buf + row * BN + 8 * gc) = lo;
        *reinterpret_cast<f32x4*>(buf + row * BN + 8 * gc + 4) = hi;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int g = i * 256 + tid, row = g / GB8;
      if (row < RB) {
        f32x4 lo, hi;
        unpack8(rgb[i], TIN(), lo, hi);
        *reinterpret_cast<f32x4*>(buf + RB * BN + row * BK + 8 * gcolb) = lo;
        *reinterpret_cast<f32x4*>(buf + RB * BN + row * BK + 8 * gcolb + 4) = hi;
      }
    }
  };

  if (ns > 0) {
    if constexpr (DMA) {
      stage(0, lds);
    } else {
      load_regs(0);
      write_lds(lds);
    }
  }
  for (int s = 0; s < ns; ++s) {
    if (GROUPED && s > 0 && s % a.emul_sps == 0) fold();
    __syncthreads();  // tile s is in LDS (DMA: the barrier's fence drains it); every wave is done with tile s-1
    float* cur = lds + (s & 1) * TILE;
    if (s + 1 < ns) {
      if constexpr (DMA) stage(s + 1, lds + ((s + 1) & 1) * TILE);
      else load_regs(s + 1);  // in flight under this tile's MFMAs, written to the other buffer below
    }
    if constexpr (R16 != 0) {
      static_assert(RB % 16 == 0, "four MFMA k-groups of four rows");
      typedef short r16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int q = 0; q < RB / 16; ++q) {  // rows 16 q + 4 e + (lane / 16), e = 0..3: this lane's four k values of one 16-deep MFMA (both operands alike)
        fvn af[4];
        fvk bf[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          af[e] = *reinterpret_cast<const fvn*>(cur + a_off + 4 * (4 * q + e) * BN);
          bf[e] = *reinterpret_cast<const fvk*>(cur + b_off + 4 * (4 * q + e) * BK);
          if (want_db) dbs += af[e];
        }
        r16x4 ah[VN], bh[VK];
#pragma unroll
        for (int i = 0; i < VN; ++i) {
          if constexpr (R16 == 1) ah[i] = __builtin_bit_cast(r16x4, bf16x4{(bf16_t)af[0][i], (bf16_t)af[1][i], (bf16_t)af[2][i], (bf16_t)af[3][i]});
          else ah[i] = __builtin_bit_cast(r16x4, f16x4{(f16_t)af[0][i], (f16_t)af[1][i], (f16_t)af[2][i], (f16_t)af[3][i]});
        }
#pragma unroll
        for (int j = 0; j < VK; ++j) {
          if constexpr (R16 == 1) bh[j] = __builtin_bit_cast(r16x4, bf16x4{(bf16_t)bf[0][j], (bf16_t)bf[1][j], (bf16_t)bf[2][j], (bf16_t)bf[3][j]});
          else bh[j] = __builtin_bit_cast(r16x4, f16x4{(f16_t)bf[0][j], (f16_t)bf[1][j], (f16_t)bf[2][j], (f16_t)bf[3][j]});
        }
#pragma unroll
        for (int i = 0; i < VN; ++i)
#pragma unroll
          for (int j = 0; j < VK; ++j) {
            if constexpr (R16 == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[i], bh[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, ah[i]), __builtin_bit_cast(f16x4, bh[j]), acc[i][j], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int q = 0; q < RB / 4; ++q) {
        const fvn af = *reinterpret_cast<const fvn*>(cur + a_off + 4 * q * BN);
        const fvk bf = *reinterpret_cast<const fvk*>(cur + b_off + 4 * q * BK);
        if (want_db) dbs += af;
#pragma unroll
        for (int i = 0; i < VN; ++i)
#pragma unroll
          for (int j = 0; j < VK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (!DMA) {
      if (s + 1 < ns) write_lds(lds + ((s + 1) & 1) * TILE);
    }
  }

  if (GROUPED) fold();
  // --- partial tile -> workspace: lane holds, for tile (i, j) and r = 0..3: n = 4*(4*fr + r) + i (VN-interleaved), k = VK*fc + j ---
  float* part = a.part + (size_t)split * a.N * a.K;
  const int kk = k0 + wk * (16 * VK) + VK * fc;
#pragma unroll
  for (int i = 0; i < VN; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * (16 * VN) + VN * (4 * fr + r) + i;
      if (n < a.N && kk < a.K) {
        fvk v;
#pragma unroll
        for (int j = 0; j < VK; ++j) v[j] = GROUPED ? tot[i][j][r] : acc[i][j][r];
        *reinterpret_cast<fvk*>(part + (size_t)n * a.K + kk) = v;
      }
    }
  if (want_db) {
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      float v = dbs[i];
      if (GROUPED) {
        v = dbt[i];
      } else {
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
      }
      const int n = n0 + wn * (16 * VN) + VN * fc + i;
      if (fr == 0 && n < a.N) a.dbpart[(size_t)split * a.N + n] = v;
    }
  }
}

template <int VN, int VK, typename TIN>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgradArgs a) {
  wgrad_f32_body<VN, VK, TIN>(wg_group_args<TIN>(a), blockIdx.x, blockIdx.y);
}
template <int R16>
__global__ __launch_bounds__(256) void wgrad_r16_kernel(const WgradArgs a) {  // 64 x 64 tiles, fp32 operands, 16-bit products
  wgrad_f32_body<2, 2, float, false, R16>(wg_group_args<float>(a), blockIdx.x, blockIdx.y);
}

// Grouped form for Linear layers over few rows (the 21-token stacks of the fusion head: M = 21 B): ~80 such weight gradients per
// iteration cost 8 us + a 6-us reduce launch EACH as separate launches, for 0.1 GFLOP in all.  Here one launch carries up to
// KPF_WGRAD_GROUP_BATCH problems (descriptors by value, read from the kernel-argument segment); a workgroup owns one 64 x 64 tile of
// one problem, walks all its pixel stages (no split, no reduce: for a 1x1 the tile IS a piece of dW) and every problem's tiles run side
// by side.  Issued once after backward by training.GroupedLinearWgrad with the (dY, X) pairs it kept alive.
struct WgradGroupBatch {
  kpf_wgrad_group_desc d[KPF_WGRAD_GROUP_BATCH];
  const float* zero;
  int nd;
};
typedef const __attribute__((address_space(4))) WgradGroupBatch* wgrad_group_kernarg_t;

__global__ __launch_bounds__(256) void wgrad_grouped_kernel(const WgradGroupBatch) {
  wgrad_group_kernarg_t bp = (wgrad_group_kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int nd = bp->nd;
  int lo = 0, hi = nd - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= bp->d[mid].first_block) lo = mid;
    else hi = mid - 1;
  }
  WgradArgs a;
  a.dy = bp->d[lo].dy, a.x = bp->d[lo].x, a.part = bp->d[lo].dw, a.dbpart = bp->d[lo].db, a.zero = bp->zero;
  a.M = bp->d[lo].M, a.N = bp->d[lo].N, a.K = bp->d[lo].K;
  a.H = 1, a.W = a.M, a.Cin = a.K, a.ldx = a.K, a.OH = 1, a.OW = a.M, a.ldy = bp->d[lo].ldy ? bp->d[lo].ldy : a.N, a.KH = 1, a.KW = 1, a.sh = 1, a.sw = 1, a.ph = 0, a.pw = 0;
  a.groups = 1, a.g_dy = a.g_x = 0, a.g_part = a.g_db = 0;
  a.tilesK = (a.K + 63) / 64;
  a.stages_per_split = (a.M + RB - 1) / RB;
  a.is1x1 = 1;
  a.emul_sps = bp->d[lo].sps;
  wgrad_f32_body<2, 2, float, true>(a, (int)blockIdx.x - bp->d[lo].first_block, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// (1b) wgrad_h16_kernel — the same GEMM for 16-bit dY / X (mixed-precision step) on v_mfma_f32_16x16x32_{bf16,f16}.  The MFMA wants 8
// consecutive REDUCTION indices (pixels) per lane, the operands are stored pixel-major — a transpose.  LDS holds the tiles as they lie in
// memory ([32 pixels][128 channels] = 256-byte rows, filled by LDS-DMA, 16-byte chunks XOR-swizzled by the row so that both the DMA
// writes and the reads are conflict-free) and ds_read_b64_tr_b16 delivers them column-major: two reads give a lane its 8 pixels of one
// channel.  Workgroup tile 128 (n) x 128 (k), wave tile 64 x 64 (16 MFMAs + 16 transposed reads per 32-pixel stage), 3-stage DMA ring
// with a counted vmcnt (two stages in flight).  Products of 16-bit values are exact in fp32 and the accumulation is fp32, as in the
// widening form this replaces (which ran the fp32 MFMA at 16x fewer FLOP per instruction); partial tiles and wgrad_reduce_kernel as above.
// ---------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
typedef __attribute__((address_space(3))) char lds_char_t;

// The transposed read as inline asm, with the wait placed by hand (tr_wait8): through the compiler builtin the waitcnt pass cannot tell
// the read from the LDS-DMAs still in flight into the OTHER ring buffers and puts `s_waitcnt vmcnt(0)` in front of the first read of
// every stage — the stage just issued is then waited for as well and the ring never overlaps anything (seen in the .s of both kernels;
// cdna_hip_programming.md "three .s-level traps").  The asm reads are invisible to that pass; the counted vmcnt in the loop is the
// only wait on the DMA queue.
__device__ __forceinline__ s16x4 tr_read(const char* p) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)(lds_char_t*)p));
  return v;
}
// lgkmcnt(0) with the eight 64-bit results tied to it, so that nothing that consumes them can be scheduled above the wait
__device__ __forceinline__ void tr_wait8(s16x4& a, s16x4& b, s16x4& c, s16x4& d, s16x4& e, s16x4& f, s16x4& g, s16x4& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}

constexpr int HB = 128;   // tile edge: one 256-byte LDS row of 16-bit channels

__device__ __forceinline__ int h16_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ f32x4 mfma_h16(const s16x8 x, const s16x8 y, const f32x4 c, bf16_t) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_h16(const s16x8 x, const s16x8 y, const f32x4 c, f16_t) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
}
__device__ __forceinline__ float sum8(const s16x8 v, bf16_t) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += __uint_as_float(((unsigned)(unsigned short)v[e]) << 16);
  return s;
}
__device__ __forceinline__ float sum8(const s16x8 v, f16_t) {
  const f16x8 h = __builtin_bit_cast(f16x8, v);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += (float)h[e];
  return s;
}

template <typename TIN, int HNS>  // HNS: LDS ring stages (RB pixels x (A + B) rows of 256 B = 16 KB each), HNS - 1 of them in flight
__global__ __launch_bounds__(256) void wgrad_h16_kernel(const WgradArgs a0) {
  const WgradArgs a = wg_group_args<TIN>(a0);
  extern __shared__ __attribute__((aligned(16))) char hl[];  // [HNS][A 8 KB | B 8 KB]
  constexpr int STAGE = 2 * RB * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wk = wave >> 1;
  const int nt = blockIdx.x / a.tilesK, kt = blockIdx.x % a.tilesK;
  const int n0 = nt * HB, k0 = kt * HB;
  const int split = blockIdx.y;
  const int total_stages = (a.M + RB - 1) / RB;
  const int s_begin = split * a.stages_per_split;
  const int ns = min(a.stages_per_split, total_stages - s_begin);
  const int m_begin = s_begin * RB;
  const TIN* dyh = static_cast<const TIN*>(a.dy);
  const TIN* xh = static_cast<const TIN*>(a.x);
  const int ohw = a.OH * a.OW;

  // --- staging: DMA instruction d (0..7 per operand) fills LDS rows 4d .. 4d+3 (1 KB, lane-linear); lane -> row 4d + lane/16, and the
  // 16-byte slot lane%16 of that row receives global chunk (lane%16) ^ swz(row).  A wave issues d = 2*wave + i, i = 0, 1, per operand.
  const int jr = lane >> 4, jc = lane & 15;
  const TIN* a_src[2];
  bool a_ok[2], b_ok[2];
  int b_cch[2], b_ky[2], b_kx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * wave + 4 * i + jr;
    const int gc = jc ^ h16_swz(row);
    const int n = n0 + 8 * gc;
    a_ok[i] = n < a.N;
    a_src[i] = dyh + n;
    const int kcol = k0 + 8 * gc;
    b_ok[i] = kcol < a.K;
    const int tap = kcol / a.Cin;
    b_cch[i] = kcol - tap * a.Cin;
    b_ky[i] = tap / a.KW;
    b_kx[i] = tap - b_ky[i] * a.KW;
  }
  auto stage = [&](int s) {
    char* buf = hl + (s % HNS) * STAGE;
    const int mb = m_begin + s * RB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mb + 8 * wave + 4 * i + jr;
      const void* src = (a_ok[i] && m < a.M) ? (const void*)(a_src[i] + (size_t)m * a.ldy) : (const void*)a.zero;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(buf + (2 * wave + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mb + 8 * wave + 4 * i + jr;
      const void* src = (const void*)a.zero;
      if (b_ok[i] && m < a.M) {
        if (a.is1x1) {
          src = xh + (size_t)m * a.ldx + b_cch[i];
        } else {
          const int b = m / ohw, r = m - b * ohw;
          const int oy = r / a.OW, ox = r - oy * a.OW;
          const int iy = oy * a.sh + b_ky[i] - a.ph, ix = ox * a.sw + b_kx[i] - a.pw;
          if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) src = xh + ((size_t)(b * a.H + iy) * a.W + ix) * a.ldx + b_cch[i];
        }
      }
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(buf + RB * 256 + (2 * wave + i) * 1024), 16, 0, 0);
    }
  };

  // --- fragment addresses: k-group g = lane/16 owns pixels 8g .. 8g+7; lane 4q+p of the group addresses row 8g + 4h + q, columns 4p .. 4p+3
  // of the 16-channel block (two 16-byte chunks c0, c0+1) — ds_read_b64_tr_b16 hands lane i of the group channel i of those 4 rows
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  int addrA[4][2], addrB[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * g + 4 * h + q, sw = h16_swz(row);
      addrA[i][h] = row * 256 + 16 * ((wn * 8 + 2 * i + (pp >> 1)) ^ sw) + 8 * (pp & 1);
      addrB[i][h] = RB * 256 + row * 256 + 16 * ((wk * 8 + 2 * i + (pp >> 1)) ^ sw) + 8 * (pp & 1);
    }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_db = a.dbpart != nullptr && kt == 0 && wk == 0;

#pragma unroll
  for (int i = 0; i < HNS - 1; ++i)
    if (ns > i) stage(i);
  for (int s = 0; s < ns; ++s) {
    // stage s has landed; the younger stages (4 DMAs per wave each) may still fly
    if (HNS > 3 && s + 2 < ns) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (s + 1 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's part of stage s is in LDS, and every wave is done reading stage s-1
    asm volatile("" ::: "memory");
    if (s + HNS - 1 < ns) stage(s + HNS - 1);  // into the buffer of stage s-1
    const char* cur = hl + (s % HNS) * STAGE;
    s16x4 ra[4][2], rb[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i][0] = tr_read(cur + addrA[i][0]);
      ra[i][1] = tr_read(cur + addrA[i][1]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rb[j][0] = tr_read(cur + addrB[j][0]);
      rb[j][1] = tr_read(cur + addrB[j][1]);
    }
    tr_wait8(ra[0][0], ra[0][1], ra[1][0], ra[1][1], ra[2][0], ra[2][1], ra[3][0], ra[3][1]);
    tr_wait8(rb[0][0], rb[0][1], rb[1][0], rb[1][1], rb[2][0], rb[2][1], rb[3][0], rb[3][1]);
    s16x8 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i] = __builtin_shufflevector(ra[i][0], ra[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
      bf[i] = __builtin_shufflevector(rb[i][0], rb[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
    }
    if (want_db) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dbs[i] += sum8(af[i], TIN());
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma_h16(af[i], bf[j], acc[i][j], TIN());
  }

  // --- partial tile -> workspace: lane holds for tile (i, j), r = 0..3: n = 16 i + 4 g + r, k = 16 j + lane%16 ---
  float* part = a.part + (size_t)split * a.N * a.K;
  const int idx = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * 64 + 16 * i + 4 * g + r;
      if (n < a.N) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + wk * 64 + 16 * j + idx;
          if (k < a.K) part[(size_t)n * a.K + k] = acc[i][j][r];
        }
      }
    }
  if (want_db) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = dbs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n = n0 + wn * 64 + 16 * i + idx;
      if (g == 0 && n < a.N) a.dbpart[(size_t)split * a.N + n] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// (1c) wgrad_h16s_kernel — the same product with a 64 x 64 output tile whose PIXEL range is split over the four waves.  The 128-tile
// form above fills the chip with ~512 workgroups of 64 KB partial tile each: 33 MB of partial sums written and read back per layer,
// whatever its shape (~9 GB per training iteration) — more than the operands.  Here a stage is 128 pixels, wave w stages and consumes
// rows 32 w .. 32 w + 31 of it (its own DMA, its own 8-KB slice of every LDS buffer: NO barrier in the main loop, only counted
// vmcnt), every wave accumulates the whole 64 x 64 tile for its pixels, and the four wave tiles are added through LDS once at the end:
// 16 KB of partial sums per workgroup (4 x less traffic at equal parallelism), and output tiles four times as many, so large
// layers need no split at all (1x1: the tile goes straight into dW, no reduce launch).  LDS rows are 128 B (64 channels); 16-byte
// chunk c of row r lives in slot c ^ f(r), f(r) = 2 ((r >> 1 & 1) | (r >> 3 & 1) << 1): a half-wave's transposed read covers 8 rows in 8
// distinct 32-byte bank groups.
// ---------------------------------------------------------------------------------------------------------------
constexpr int HS_TB = 64;    // tile edge
constexpr int HS_SP = 128;   // pixels per stage (32 per wave)

__device__ __forceinline__ int hs_swz(int row) { return 2 * (((row >> 1) & 1) | (((row >> 3) & 1) << 1)); }

// XCD-aware order of the tiles of one (split, group) slab of the grid: workgroups are handed to the 8 XCDs (private 4-MB L2s) round-robin in dispatch
// order (x fastest, then y, z); the slab's workgroups that land on one XCD get a CONTIGUOUS range of (n tile, k tile) pairs, k tiles fastest — so an XCD
// reads 1/8 of dY's columns (and all of X) instead of every XCD reading everything.  Bijective on [0, gx) for any gx and slab offset.
__device__ __forceinline__ int xcd_slab_remap(int x, int gx, int slab) {
  const int o = (int)(((long)gx * slab) & 7);
  const int c = (x + o) & 7;
  int before = 0;
  for (int cc = 0; cc < c; ++cc) before += (o + gx - cc + 7) / 8 - (o - cc + 7) / 8;
  return before + (x - ((c - o) & 7)) / 8;
}

template <typename TIN, int HS_NS>  // HS_NS: LDS ring depth; per wave and buffer 32 rows x 128 B x (A + B) = 8 KB (3: 96 KB per workgroup, 2: 64 KB)
__global__ __launch_bounds__(256) void wgrad_h16s_kernel(const WgradArgs a0) {
  const WgradArgs a = wg_group_args<TIN>(a0);
  extern __shared__ __attribute__((aligned(16))) char hl[];  // [4 waves][HS_NS][A 4 KB | B 4 KB] + 1 KB; reused for the final cross-wave sum
  constexpr int WBUF = 2 * 32 * 128;           // one wave's A + B rows of one stage
  constexpr int WREG = HS_NS * WBUF;           // one wave's region (24 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = a.xcd ? xcd_slab_remap((int)blockIdx.x, (int)gridDim.x, (int)(blockIdx.y + gridDim.y * blockIdx.z)) : (int)blockIdx.x;
  const int nt = bid / a.tilesK, kt = bid % a.tilesK;
  const int n0 = nt * HS_TB, k0 = kt * HS_TB;
  const int split = blockIdx.y;
  const int total_stages = (a.M + HS_SP - 1) / HS_SP;
  const int s_begin = split * a.stages_per_split;
  const int ns = min(a.stages_per_split, total_stages - s_begin);
  const int m_begin = s_begin * HS_SP + 32 * wave;  // this wave's first pixel
  const TIN* dyh = static_cast<const TIN*>(a.dy);
  const TIN* xh = static_cast<const TIN*>(a.x);
  const int ohw = a.OH * a.OW;
  char* mine = hl + wave * WREG;

  // --- staging: wave-DMA i (0..3 per operand) fills rows 8 i .. 8 i + 7 of the wave's 32 (1 KB, lane-linear): lane -> row 8 i + lane / 8,
  // slot lane % 8 of that row receives global chunk (lane % 8) ^ f(row)
  const int jr = lane >> 3, jc = lane & 7;
  const TIN* a_src[4];
  bool a_ok[4], b_ok[4];
  int b_cch[4], b_ky[4], b_kx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + jr;
    const int gc = jc ^ hs_swz(row);
    const int n = n0 + 8 * gc;
    a_ok[i] = n < a.N;
    a_src[i] = dyh + n;
    const int kcol = k0 + 8 * gc;
    b_ok[i] = kcol < a.K;
    const int tap = kcol / a.Cin;
    b_cch[i] = kcol - tap * a.Cin;
    b_ky[i] = tap / a.KW;
    b_kx[i] = tap - b_ky[i] * a.KW;
  }
  auto stage = [&](int s) {
    char* buf = mine + (s % HS_NS) * WBUF;
    const int mb = m_begin + s * HS_SP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + 8 * i + jr;
      const void* src = (a_ok[i] && m < a.M) ? (const void*)(a_src[i] + (size_t)m * a.ldy) : (const void*)a.zero;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(buf + i * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + 8 * i + jr;
      const void* src = (const void*)a.zero;
      if (b_ok[i] && m < a.M) {
        if (a.is1x1) {
          src = xh + (size_t)m * a.ldx + b_cch[i];
        } else {
          const int b = m / ohw, r = m - b * ohw;
          const int oy = r / a.OW, ox = r - oy * a.OW;
          const int iy = oy * a.sh + b_ky[i] - a.ph, ix = ox * a.sw + b_kx[i] - a.pw;
          if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) src = xh + ((size_t)(b * a.H + iy) * a.W + ix) * a.ldx + b_cch[i];
        }
      }
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(buf + 4096 + i * 1024), 16, 0, 0);
    }
  };

  // --- fragment addresses inside a wave buffer: k-group g = lane / 16 owns pixels 8 g .. 8 g + 7 of the wave's 32
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  int addrA[4][2], addrB[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * g + 4 * h + q;
      const int off = row * 128 + 16 * ((2 * i + (pp >> 1)) ^ hs_swz(row)) + 8 * (pp & 1);
      addrA[i][h] = off;
      addrB[i][h] = 4096 + off;
    }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_db = a.dbpart != nullptr && kt == 0;

#pragma unroll
  for (int i = 0; i < HS_NS - 1; ++i)
    if (ns > i) stage(i);
  for (int s = 0; s < ns; ++s) {
    // this wave's rows of stage s have landed (8 DMAs per wave and stage; with three buffers one younger stage may still fly)
    if (HS_NS > 2 && s + 1 < ns) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* cur = mine + (s % HS_NS) * WBUF;
    s16x4 ra[4][2], rb[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i][0] = tr_read(cur + addrA[i][0]);
      ra[i][1] = tr_read(cur + addrA[i][1]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rb[j][0] = tr_read(cur + addrB[j][0]);
      rb[j][1] = tr_read(cur + addrB[j][1]);
    }
    tr_wait8(ra[0][0], ra[0][1], ra[1][0], ra[1][1], ra[2][0], ra[2][1], ra[3][0], ra[3][1]);
    tr_wait8(rb[0][0], rb[0][1], rb[1][0], rb[1][1], rb[2][0], rb[2][1], rb[3][0], rb[3][1]);
    s16x8 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i] = __builtin_shufflevector(ra[i][0], ra[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
      bf[i] = __builtin_shufflevector(rb[i][0], rb[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
    }
    // the buffer of stage s-1 is free (its reads fed last iteration's MFMAs): refill it while this stage is multiplied
    if (s + HS_NS - 1 < ns) stage(s + HS_NS - 1);
    if (want_db) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dbs[i] += sum8(af[i], TIN());
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma_h16(af[i], bf[j], acc[i][j], TIN());
  }

  // --- the four wave tiles -> LDS ([wave][64 n][64 k] floats in the wave's own region), added in wave order, -> workspace / dW ---
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float* wt = reinterpret_cast<float*>(mine);
  const int idx = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) wt[(16 * i + 4 * g + r) * 64 + 16 * j + idx] = acc[i][j][r];
  float* dbl = reinterpret_cast<float*>(hl + 4 * WREG) + 64 * wave;  // [64] column sums of this wave's pixels (behind the four regions)
  if (want_db) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = dbs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) dbl[16 * i + idx] = v;
    }
  }
  __syncthreads();
  const float* w0 = reinterpret_cast<const float*>(hl);
  constexpr int WF = WREG / 4;  // floats between two waves' regions
  float* part = a.part + (size_t)split * a.N * a.K;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int o = tid + 256 * e;
    const int n = n0 + (o >> 6), k = k0 + (o & 63);
    if (n < a.N && k < a.K) part[(size_t)n * a.K + k] = (w0[o] + w0[WF + o]) + (w0[2 * WF + o] + w0[3 * WF + o]);
  }
  if (want_db && tid < 64) {
    const int n = n0 + tid;
    const float* d0 = reinterpret_cast<const float*>(hl + 4 * WREG);
    if (n < a.N) a.dbpart[(size_t)split * a.N + n] = (d0[tid] + d0[64 + tid]) + (d0[128 + tid] + d0[192 + tid]);
  }
}

// Fixed-order sum of S partial arrays of n floats (n % 4 == 0, 16-byte aligned), four consecutive outputs per thread.  S > SEQ_SPLITS: a workgroup
// owns 256 consecutive outputs, its four waves sum the partials p = w, w+4, w+8, ... (four independent chains each) and the four wave sums are
// combined through LDS — the same order every run.  Up to SEQ_SPLITS partials are simply added in split order, every wave on its own 256 outputs
// (1024 per workgroup): that is the order wgrad_grouped_kernel reproduces inside one workgroup (its problems have at most 8 splits), so a small
// layer's gradient has the same bits whichever form computed it.  (Round 6: float4 per thread instead of one float — a quarter of the workgroups
// and load instructions for the same additions in the same order; the ~100 reduces of a training iteration took 0.63 ms before.)
constexpr int SEQ_SPLITS = 8;
inline int reduce_blocks(long n, int S) { return (int)(S <= SEQ_SPLITS ? (n + 1023) / 1024 : (n + 255) / 256); }
// -> true when this thread holds the sums v of outputs [i, i + 4)
__device__ __forceinline__ bool sum_partials4(const float* __restrict__ part, long n, int S, int bx, long& i, f32x4& v, f32x4 (*red)[64]) {
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (S <= SEQ_SPLITS) {
    i = ((long)bx * 256 + threadIdx.x) * 4;
    v = z4;
    if (i < n)
      for (int p = 0; p < S; ++p) v += kpf_ld4(part + (size_t)p * n + i);
    return i < n;
  }
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  i = ((long)bx * 64 + o) * 4;
  f32x4 s0 = z4, s1 = z4, s2 = z4, s3 = z4;
  if (i < n) {
    int p = g;
#pragma unroll 2
    for (; p + 12 < S; p += 16) {
      s0 += kpf_ld4(part + (size_t)p * n + i);
      s1 += kpf_ld4(part + (size_t)(p + 4) * n + i);
      s2 += kpf_ld4(part + (size_t)(p + 8) * n + i);
      s3 += kpf_ld4(part + (size_t)(p + 12) * n + i);
    }
    for (; p < S; p += 4) s0 += kpf_ld4(part + (size_t)p * n + i);
  }
  red[g][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  v = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
  return g == 0 && i < n;
}

// dw[n][c][ky][kx] = sum_s part[s][n][(ky,kx,c)];  db[n] = sum_s dbpart[s][n]   (blocks [0, nkb) reduce dw, the rest db)
// (grp = channel group of a grouped launch: its partial sums start g_ws floats further, its dw / db N_out*K_out / N_out floats further.
//  Cin_out <= Cin, N_out <= N: the operands carried zero channels up to whole channel groups; dw [N_out][Cin_out][KH][KW] drops them)
__device__ __forceinline__ void reduce_conv_body(const float* __restrict__ part, const float* __restrict__ dbpart, float* __restrict__ dw, float* __restrict__ db,
                                                 int S, int N, int K, int Cin, int KHW, int nkb, long g_ws, int Cin_out, int N_out, int bx, int grp,
                                                 f32x4 (*red)[64]) {
  part += grp * g_ws, dw += (long)grp * N_out * Cin_out * KHW;
  if (dbpart) dbpart += grp * g_ws, db += grp * N_out;
  long i;
  f32x4 v;
  if (bx < nkb) {
    if (sum_partials4(part, (long)N * K, S, bx, i, v, red)) {  // (K % 4 == 0: the four outputs share n and the tap, c .. c + 3)
      const int n = (int)(i / K), k = (int)(i - (long)n * K);
      const int tap = k / Cin, c = k - tap * Cin;
      if (n < N_out) {
        if (KHW == 1 && Cin_out == Cin) {
          kpf_st4(dw + (size_t)n * Cin + c, v);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c + j < Cin_out) dw[((size_t)n * Cin_out + c + j) * KHW + tap] = v[j];
        }
      }
    }
  } else {
    if (sum_partials4(dbpart, N, S, bx - nkb, i, v, red)) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j < N_out) db[i + j] = v[j];
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ dbpart,
                                                           float* __restrict__ dw, float* __restrict__ db, int S, int N, int K, int Cin,
                                                           int KHW, int nkb, long g_ws, int Cin_out, int N_out) {
  __shared__ f32x4 red[4][64];
  reduce_conv_body(part, dbpart, dw, db, S, N, K, Cin, KHW, nkb, g_ws, Cin_out, N_out, (int)blockIdx.x, (int)blockIdx.y, red);
}

// depthwise 7x7: dw[c][tap] = sum_s part[s][tap][c];  db[c] = sum_s dbpart[s][c]
__device__ __forceinline__ void reduce_dw7_body(const float* __restrict__ part, const float* __restrict__ dbpart, float* __restrict__ dw, float* __restrict__ db,
                                                int S, int C, int nkb, int bx, f32x4 (*red)[64]) {
  long i;
  f32x4 v;
  if (bx < nkb) {
    if (sum_partials4(part, 49L * C, S, bx, i, v, red)) {
      const int tap = (int)(i / C), c = (int)(i - (long)tap * C);
#pragma unroll
      for (int j = 0; j < 4; ++j) dw[(c + j) * 49 + tap] = v[j];
    }
  } else {
    if (sum_partials4(dbpart, C, S, bx - nkb, i, v, red)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) db[i + j] = v[j];
    }
  }
}

// The reduces of up to KPF_WGRAD_REDUCE_BATCH weight-gradient calls in ONE launch (kpf_wgrad_reduce_multi): a workgroup finds the call it
// belongs to from the calls' first blocks and runs that call's own reduce body — same order of additions, same bits as the per-call launches.
struct ReduceBatch {
  kpf_wgrad_reduce_desc d[KPF_WGRAD_REDUCE_BATCH];
  int nd;
};
typedef const __attribute__((address_space(4))) ReduceBatch* reduce_kernarg_t;

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const ReduceBatch) {
  reduce_kernarg_t bp = (reduce_kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ f32x4 red[4][64];
  const int b = (int)blockIdx.x;
  int k = 0;
  const int nd = bp->nd;
  while (k + 1 < nd && b >= bp->d[k + 1].first_block) ++k;
  const int local = b - bp->d[k].first_block;
  const int per = bp->d[k].nkb + bp->d[k].ndb;
  if (bp->d[k].kind == 1) {
    reduce_dw7_body(bp->d[k].part, bp->d[k].dbpart, bp->d[k].dw, bp->d[k].db, bp->d[k].S, bp->d[k].N, bp->d[k].nkb, local, red);
  } else {
    const int grp = local / per;
    reduce_conv_body(bp->d[k].part, bp->d[k].dbpart, bp->d[k].dw, bp->d[k].db, bp->d[k].S, bp->d[k].N, bp->d[k].K, bp->d[k].Cin, bp->d[k].KHW, bp->d[k].nkb,
                     bp->d[k].g_ws, bp->d[k].Cin_out, bp->d[k].N_out, local - grp * per, grp, red);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// depthwise 7x7 (pad 3, stride 1): dW[c][ky][kx] = sum_{b,y,x} dY[b][y][x][c] * X[b][y+ky-3][x+kx-3][c],  db[c] = sum dY
// thread = (channel quad q, tap row ky); chunk (blockIdx.y) = a range of (b, y) output rows.
// part: [S][7][7][C] (+ [S][C] for db, accumulated by the ky == 3 threads)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ part, float* __restrict__ dbpart, int B, int H, int W,
                                                            int C, int rows_per_chunk) {
  const int Q = C >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 7 * Q) return;
  const int ky = t / Q, q = t - ky * Q;
  const int c = 4 * q;
  const long rows = (long)B * H;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  const long r1 = min(rows, r0 + rows_per_chunk);
  f32x4 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dbs = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long r = r0; r < r1; ++r) {
    const int b = (int)(r / H), y = (int)(r - (long)b * H);
    const int iy = y + ky - 3;
    const float* dyr = dy + ((size_t)r * W) * C + c;
    if (ky == 3) {
      for (int xx = 0; xx < W; ++xx) dbs += kpf_ld4(dyr + (size_t)xx * C);
    }
    if ((unsigned)iy >= (unsigned)H) continue;
    const float* xr = x + (((size_t)b * H + iy) * W) * C + c;
    for (int x0 = 0; x0 < W; x0 += 8) {
      f32x4 win[14], g[8];
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const int ix = x0 + j - 3;
        win[j] = (unsigned)ix < (unsigned)W ? kpf_ld4(xr + (size_t)ix * C) : z4;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = x0 + j < W ? kpf_ld4(dyr + (size_t)(x0 + j) * C) : z4;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) acc[kx] += g[j] * win[j + kx];
    }
  }
  float* p = part + ((size_t)blockIdx.y * 49 + ky * 7) * C + c;
#pragma unroll
  for (int kx = 0; kx < 7; ++kx) kpf_st4(p + (size_t)kx * C, acc[kx]);
  if (ky == 3) kpf_st4(dbpart + (size_t)blockIdx.y * C + c, dbs);
}

// (2b) dwconv7_wgrad_lds_kernel (round 6) — the same sums with the operands staged through LDS.  The kernel above has every (channel quad, tap row) thread
// fetch its own x and dY rows: each byte of both tensors is requested seven times, and with whole-width channel blocks per workgroup the seven-row window (170 KB
// at 32 x 192 channels) misses the L1 — the stage-1 layers of the training step ran at 64 us for 50 MB of operands (L2-bound at ~10 x the unique bytes).  Here a
// workgroup owns CQ channel quads (128-512 B of every pixel) and a chunk of output rows; an 8-slot LDS ring holds the x rows r-3 .. r+4 of that channel block
// (slot = global row & 7; three zero halo columns either side), two slots hold dY rows r, r+1; per output row ONE new x row and ONE dY row are fetched (registers
// -> LDS, issued before the row's arithmetic, stored after it), and thread (quad, tap row ky, 8-column segment) reads its 14-pixel window and 8 gradients from LDS.
// The segments' sums are added in segment order through LDS at the end: partial layout, summation order across chunks and the reduce kernel are unchanged.
constexpr int DWL_RING = 8;
struct DwlGeom { int nseg, CQ, Wp, xrow, lds_f4; };
__host__ __device__ inline DwlGeom dwl_geom(int W, int C) {
  DwlGeom g;
  g.nseg = (W + 7) / 8;
  int cq = 256 / (7 * g.nseg);
  int p2 = 1;
  while (p2 * 2 <= cq) p2 *= 2;
  const int Q = C >> 2;
  while (p2 > 1 && p2 / 2 >= Q) p2 /= 2;  // (narrow layers: no more quads per workgroup than the layer has, rounded up to a power of two)
  g.CQ = cq < 1 ? 0 : p2;
  g.Wp = g.nseg * 8;
  g.xrow = (g.Wp + 6) * g.CQ;
  g.xrow += (8 - g.xrow % 16 + 16) % 16;  // row stride = 128 B mod 256 B: the tap rows of a 16-lane LDS pass alternate between the two halves of the banks
  g.lds_f4 = DWL_RING * g.xrow + 2 * g.Wp * g.CQ;
  return g;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void dwconv7_wgrad_lds_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                                float* __restrict__ dbpart, int B, int H, int W, int C, int rows_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) f32x4 dwl_sm[];
  const DwlGeom gm = dwl_geom(W, C);
  const int CQ = gm.CQ, nseg = gm.nseg, Wp = gm.Wp, xrow = gm.xrow;
  f32x4* xs = dwl_sm;                    // [DWL_RING][xrow]: column c of a row at (c + 3) * CQ
  f32x4* ds = dwl_sm + DWL_RING * xrow;  // [2][Wp * CQ]
  const int tid = threadIdx.x;
  const int Q = C >> 2;
  const int q0 = blockIdx.x * CQ;
  const long rows = (long)B * H;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  const long r1 = min(rows, r0 + rows_per_chunk);
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < gm.lds_f4; i += 256) dwl_sm[i] = z4;  // halo columns, pad columns of dY, stride padding: zero for the whole kernel
  __syncthreads();
  const int nx = W * CQ;  // float4 elements of one staged row of this channel block (x and dY alike)
  // element e of a row: column e / CQ, quad e % CQ; at most two per thread (W * CQ <= 8 nseg * 256 / (7 nseg) < 512)
  int e_col[2], e_cq[2];
  bool e_ok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = tid + 256 * k;
    e_col[k] = e / CQ, e_cq[k] = e - e_col[k] * CQ;
    e_ok[k] = e < nx && q0 + e_cq[k] < Q;
  }
  auto fetch = [&](const float* __restrict__ src, long rg, f32x4* v) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      v[k] = (e_ok[k] && rg >= 0 && rg < rows) ? kpf_ld4(src + ((size_t)rg * W + e_col[k]) * C + 4 * (q0 + e_cq[k])) : z4;
  };
  auto put_x = [&](long rg, const f32x4* v) {
    f32x4* row = xs + (int)(rg & (DWL_RING - 1)) * xrow;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 256 * k < nx) row[(e_col[k] + 3) * CQ + e_cq[k]] = v[k];
  };
  auto put_d = [&](long rg, const f32x4* v) {
    f32x4* row = ds + (int)(rg & 1) * Wp * CQ;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 256 * k < nx) row[e_col[k] * CQ + e_cq[k]] = v[k];
  };
  f32x4 pxv[2], pdv[2];  // the rows in flight: x row r + 4 and dY row r + 1 while row r is being processed (requested one iteration earlier)
  {  // prologue: x rows r0 - 3 .. r0 + 3 and dY row r0 — all requested before the first is stored — and the first pair in flight
    f32x4 v[8][2];
#pragma unroll
    for (int i = 0; i < 7; ++i) fetch(x, r0 - 3 + i, v[i]);
    fetch(dy, r0, v[7]);
    fetch(x, r0 + 4, pxv);
    fetch(dy, r0 + 1 < r1 ? r0 + 1 : -1, pdv);
#pragma unroll
    for (int i = 0; i < 7; ++i) put_x(r0 - 3 + i, v[i]);
    put_d(r0, v[7]);
  }
  const int cq = tid % CQ, rest = tid / CQ;
  const int ky = rest % 7, seg = rest / 7;
  const bool active = seg < nseg;
  f32x4 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) acc[j] = z4;
  f32x4 dbs = z4;
  for (long r = r0; r < r1; ++r) {
    __syncthreads();  // the rows stored at the end of the last iteration (or by the prologue) are visible; everyone is done with the slots about to be refilled
    f32x4 nxv[2], ndv[2];  // requested now, stored at the end of the NEXT iteration: two iterations for the round trip
    fetch(x, r + 2 < r1 ? r + 5 : -1, nxv);
    fetch(dy, r + 2 < r1 ? r + 2 : -1, ndv);
    if (active) {
      const int b = (int)(r / H), y = (int)(r - (long)b * H);
      const int iy = y + ky - 3;
      const f32x4* dr = ds + (int)(r & 1) * Wp * CQ + (seg * 8) * CQ + cq;
      f32x4 g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = dr[j * CQ];
      if (ky == 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dbs += g[j];
      }
      if ((unsigned)iy < (unsigned)H) {
        const f32x4* xr = xs + (int)((r + ky - 3) & (DWL_RING - 1)) * xrow + (seg * 8) * CQ + cq;
        f32x4 win[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) win[j] = xr[j * CQ];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) acc[kx] += g[j] * win[j + kx];
      }
    }
    if (r + 1 < r1) {
      put_x(r + 4, pxv);
      put_d(r + 1, pdv);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) pxv[k] = nxv[k], pdv[k] = ndv[k];
  }
  // --- the segments' sums, added in segment order: red[seg][ky][kx][cq] over the ring, the bias sums behind them ---
  __syncthreads();
  f32x4* red = dwl_sm;
  f32x4* dred = dwl_sm + nseg * 49 * CQ;
  if (active) {
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) red[((seg * 7 + ky) * 7 + kx) * CQ + cq] = acc[kx];
    if (ky == 3) dred[seg * CQ + cq] = dbs;
  }
  __syncthreads();
  for (int o = tid; o < 49 * CQ; o += 256) {
    const int c = o % CQ, tap = o / CQ;
    if (q0 + c >= Q) continue;
    f32x4 t = red[tap * CQ + c];
    for (int sg = 1; sg < nseg; ++sg) t += red[(sg * 49 + tap) * CQ + c];
    kpf_st4(part + ((size_t)blockIdx.y * 49 + tap) * C + 4 * (q0 + c), t);
  }
  if (tid < CQ && q0 + tid < Q) {
    f32x4 t = dred[tid];
    for (int sg = 1; sg < nseg; ++sg) t += dred[sg * CQ + tid];
    kpf_st4(dbpart + (size_t)blockIdx.y * C + 4 * (q0 + tid), t);
  }
}

__global__ __launch_bounds__(256) void dwconv7_wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ dbpart,
                                                                   float* __restrict__ dw, float* __restrict__ db, int S, int C, int nkb) {
  __shared__ f32x4 red[4][64];
  reduce_dw7_body(part, dbpart, dw, db, S, C, nkb, (int)blockIdx.x, red);
}

struct Plan { int vn, vk, tilesN, tilesK, S, sps; };

// Tile shape and split count from a small cost model: MFMA time of the padded tiles, L2 -> LDS bytes (every dY row is staged once
// per K tile column, every X row once per N tile row), and the workspace round trip of S partial outputs; S fills the chip
// (~3 workgroups per CU) with at least four stages per workgroup.
// direct (1x1 convolution / Linear over at most 128 rows): one workgroup per 64 x 64 tile walks the (<= 4) stages and writes dW itself
// — the partial layout [N][K] IS dW's for a 1x1 — without a reduce launch.  (Tried up to 24 stages for the 21-token stacks of the fusion
// head, M = 21 B = 672: the serial walk costs 19 us against 8 + 6 for split + reduce.)
constexpr int DIRECT_MAX_STAGES = 4;
Plan plan_wgrad(long M, int N, int K, bool direct_ok = false) {
  const int total = (int)((M + RB - 1) / RB);
  if (direct_ok && total <= DIRECT_MAX_STAGES) {
    Plan p{};
    p.vn = p.vk = 2;
    p.tilesN = (N + 63) / 64, p.tilesK = (K + 63) / 64;
    p.S = 1, p.sps = total;
    return p;
  }
  Plan best{};
  double best_t = 1e30;
  for (int vn = 2; vn <= 4; vn += 2)
    for (int vk = 2; vk <= 4; vk += 2) {
      Plan p;
      p.vn = vn, p.vk = vk;
      p.tilesN = (N + 32 * vn - 1) / (32 * vn);
      p.tilesK = (K + 32 * vk - 1) / (32 * vk);
      const int tiles = p.tilesN * p.tilesK;
      int S = (768 + tiles - 1) / tiles;
      const int smax = (total + 3) / 4;
      if (S > smax) S = smax < 1 ? 1 : smax;
      p.sps = (total + S - 1) / S;
      p.S = (total + p.sps - 1) / p.sps;
      const double np = 32.0 * vn * p.tilesN, kp = 32.0 * vk * p.tilesK;
      const double t_mfma = 2.0 * M * np * kp / 110e12;
      const double t_l2 = 4.0 * M * (np * p.tilesK + kp * p.tilesN) / 5e12;
      const double t_ws = 8.0 * p.S * (double)N * K / 3e12;
      const double occ = (double)tiles * p.S / 512.0;  // fewer than two workgroups per CU: the chip is not full
      const double t = (t_mfma > t_l2 ? t_mfma : t_l2) / (occ < 1.0 ? occ : 1.0) + t_ws;
      if (t < best_t) best_t = t, best = p;
    }
  return best;
}

// 128 x 128 tiles of wgrad_h16_kernel: S fills the chip (two 48-KB workgroups per CU) with at least four stages per workgroup
Plan plan_wgrad_h16(long M, int N, int K) {
  Plan p{};
  p.vn = p.vk = 4;
  p.tilesN = (N + HB - 1) / HB;
  p.tilesK = (K + HB - 1) / HB;
  const int total = (int)((M + RB - 1) / RB);
  const int tiles = p.tilesN * p.tilesK;
  static const int target = []() { const char* e = getenv("KPF_WG16_TARGET"); return e ? atoi(e) : 640; }();   // tuning aids
  static const int minst = []() { const char* e = getenv("KPF_WG16_MINSTAGES"); return e ? atoi(e) : 4; }();
  int S = (target + tiles - 1) / tiles;
  const int smax = (total + minst - 1) / minst;
  if (S > smax) S = smax < 1 ? 1 : smax;
  p.sps = (total + S - 1) / S;
  p.S = (total + p.sps - 1) / p.sps;
  return p;
}

// 64 x 64 tiles of wgrad_h16s_kernel (two 64-KB workgroups per CU): S brings the grid to ~1.5 workgroups per CU, at least two stages each
Plan plan_wgrad_h16s(long M, int N, int K, int groups = 1) {
  Plan p{};
  p.vn = p.vk = 2;
  p.tilesN = (N + HS_TB - 1) / HS_TB;
  p.tilesK = (K + HS_TB - 1) / HS_TB;
  const int total = (int)((M + HS_SP - 1) / HS_SP);
  const int tiles = p.tilesN * p.tilesK;
  static const int target = []() { const char* e = getenv("KPF_WG16S_TARGET"); return e ? atoi(e) : 384; }();  // tuning aid (256 / 384 / 512 / 768: 3.13 / 3.01 / 3.19 / 3.40 ms of GEMM + reduce per iteration)
  // (a grouped launch runs `groups` problems side by side: each gets its share of the chip — train128_bf16 with the paired backbones: 18.65 ms per
  //  iteration with every problem split for the whole chip, 18.36 with the share)
  const int tgt = target / (groups > 1 ? groups : 1);
  int S = (tgt + tiles - 1) / tiles;
  if (tiles >= tgt / 2) S = 1;  // enough tiles: no split, no reduce launch for a 1x1
  const int smax = (total + 1) / 2;
  if (S > smax) S = smax < 1 ? 1 : smax;
  p.sps = (total + S - 1) / S;
  p.S = (total + p.sps - 1) / p.sps;
  return p;
}

const float* zero_page() {
  static std::atomic<const float*> cache[KPF_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= KPF_MAX_DEVICES) return nullptr;
  const float* p = cache[dev].load(std::memory_order_acquire);
  if (p) return p;
  void* q = nullptr;
  if (hipGetSymbolAddress(&q, HIP_SYMBOL(wg_zero16)) != hipSuccess || !q) return nullptr;
  cache[dev].store((const float*)q, std::memory_order_release);
  return (const float*)q;
}

template <int VN, int VK, typename TIN>
int launch_wgrad(const WgradArgs& a, const Plan& p, hipStream_t st) {
  constexpr int LDS = 2 * RB * 32 * (VN + VK) * 4;
  hipLaunchKernelGGL((wgrad_f32_kernel<VN, VK, TIN>), dim3(p.tilesN * p.tilesK, p.S, a.groups), dim3(256), LDS, st, a);
  return kpf_check_launch("kpf_conv2d_wgrad");
}

template <typename TIN>
int launch_wgrad_any(const WgradArgs& a, const Plan& p, hipStream_t st, int r16 = 0) {
  if constexpr (std::is_same<TIN, float>::value) {
    if (r16 && p.vn == 2 && p.vk == 2) {  // (the 64 x 64 form only: what the plan gives the layers that ask; other tile shapes keep fp32 products)
      constexpr int LDS = 2 * RB * 32 * (2 + 2) * 4;
      if (r16 == 1) hipLaunchKernelGGL(wgrad_r16_kernel<1>, dim3(p.tilesN * p.tilesK, p.S, a.groups), dim3(256), LDS, st, a);
      else hipLaunchKernelGGL(wgrad_r16_kernel<2>, dim3(p.tilesN * p.tilesK, p.S, a.groups), dim3(256), LDS, st, a);
      return kpf_check_launch("kpf_conv2d_wgrad");
    }
  }
  if (p.vn == 4 && p.vk == 4) return launch_wgrad<4, 4, TIN>(a, p, st);
  if (p.vn == 4) return launch_wgrad<4, 2, TIN>(a, p, st);
  if (p.vk == 4) return launch_wgrad<2, 4, TIN>(a, p, st);
  return launch_wgrad<2, 2, TIN>(a, p, st);
}

// chunks of the LDS form from (B, H, C) alone (the workspace query has no W: a square map is assumed for the estimate — any W gives correct results, the
// estimate only steers the grid towards ~3 workgroups per CU), at least four output rows per chunk (each chunk fetches six halo rows on top of its own)
int dwl_chunk_rows(int B, int H, int C) {
  const DwlGeom g = dwl_geom(H, C);
  const long rows = (long)B * H;
  const int cb = g.CQ > 0 ? ((C >> 2) + g.CQ - 1) / g.CQ : 1;
  static const int target = []() { const char* e = getenv("KPF_DW7_TARGET"); return e ? atoi(e) : 768; }();  // tuning aid: workgroups aimed at
  static const int minrows = []() { const char* e = getenv("KPF_DW7_MINROWS"); return e ? atoi(e) : 4; }();
  long S = (target + cb - 1) / cb;
  const long smax = (rows + minrows - 1) / minrows;
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  return (int)((rows + S - 1) / S);
}

int dw_chunk_rows(int B, int H, int C) {
  const long rows = (long)B * H;
  const int bx = (7 * (C / 4) + 255) / 256;
  long S = 768 / bx;  // ~3 workgroups per CU
  if (S < 1) S = 1;
  if (S > rows) S = rows;
  return (int)((rows + S - 1) / S);
}

}  // namespace

extern "C" {

long kpf_conv2d_wgrad_ws_floats(long M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const Plan p = plan_wgrad(M, N, K), h = plan_wgrad_h16(M, N, K), hs = plan_wgrad_h16s(M, N, K);  // (any of the kernels may take the call)
  const int S = p.S > h.S ? (p.S > hs.S ? p.S : hs.S) : (h.S > hs.S ? h.S : hs.S);
  return (long)S * N * K + (long)S * N;
}

static int conv2d_wgrad_impl(const void* dy, const void* x, int dtype, float* dw, float* db, float* ws, long ws_floats, int B, int H, int W,
                             int Cin, int ldx, int OH, int OW, int N, int ldy, int KH, int KW, int sh, int sw, int ph, int pw,
                             void* stream, int groups = 1, int cin_valid = 0, int n_valid = 0, kpf_wgrad_reduce_desc* defer = nullptr) {
  if (defer) defer->kind = -1;  // (nothing to reduce unless the split path below says so)
  if (cin_valid <= 0) cin_valid = Cin;
  if (n_valid <= 0) n_valid = N;
  KPF_REQUIRE(cin_valid <= Cin && n_valid <= N, "kpf_conv2d_wgrad: cin_valid / n_valid exceed Cin / N");
  const bool trimmed = cin_valid != Cin || n_valid != N;
  KPF_REQUIRE(groups >= 1 && (long)groups * Cin <= ldx && (long)groups * N <= ldy, "kpf_conv2d_wgrad: %d groups of %d / %d channels exceed the pixel strides %d / %d", groups, Cin, N, ldx, ldy);
  KPF_REQUIRE(dy && x && dw && ws, "kpf_conv2d_wgrad_f32: null pointer");
  int r16 = 0;  // KPF_DT_F32_MMA_BF16 / _F16: fp32 operands, products on their 16-bit roundings
  if (dtype == KPF_DT_F32_MMA_BF16 || dtype == KPF_DT_F32_MMA_F16) r16 = dtype == KPF_DT_F32_MMA_BF16 ? 1 : 2, dtype = KPF_DT_F32;
  KPF_REQUIRE(dtype == KPF_DT_F32 || dtype == KPF_DT_BF16 || dtype == KPF_DT_F16, "kpf_conv2d_wgrad: unknown dtype %d", dtype);
  if (dtype != KPF_DT_F32)
    KPF_REQUIRE(Cin % 8 == 0 && ldx % 8 == 0 && N % 8 == 0 && ldy % 8 == 0, "kpf_conv2d_wgrad_h16: Cin, N, ldx, ldy must be multiples of 8");
  KPF_REQUIRE(B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && KH > 0 && KW > 0 && sh > 0 && sw > 0, "kpf_conv2d_wgrad_f32: bad shape");
  KPF_REQUIRE(Cin > 0 && Cin % 4 == 0 && ldx % 4 == 0 && ldx >= Cin, "kpf_conv2d_wgrad_f32: Cin and ldx must be multiples of 4 (got %d, %d)", Cin, ldx);
  KPF_REQUIRE(N > 0 && N % 4 == 0 && ldy % 4 == 0 && ldy >= N, "kpf_conv2d_wgrad_f32: N and ldy must be multiples of 4 (got %d, %d)", N, ldy);
  KPF_REQUIRE(kpf_aligned16(dy) && kpf_aligned16(x) && kpf_aligned16(ws), "kpf_conv2d_wgrad_f32: dy, x, ws must be 16-byte aligned");
  KPF_REQUIRE(OH == (H + 2 * ph - KH) / sh + 1 && OW == (W + 2 * pw - KW) / sw + 1, "kpf_conv2d_wgrad_f32: output size %dx%d does not match the convolution", OH, OW);
  const long M = (long)B * OH * OW;
  const long K = (long)KH * KW * Cin;
  KPF_REQUIRE(M < (1L << 31) && K < (1L << 24) && (long)B * H * W < (1L << 31), "kpf_conv2d_wgrad_f32: problem too large");
  static const int h16_widen = []() { const char* e = getenv("KPF_WGRAD_H16_WIDEN"); return e ? atoi(e) : 0; }();  // tuning aid: the old widening kernel
  // 16-bit operands: the 64-tile form (wgrad_h16s_kernel, two buffers = two workgroups per CU) — GEMM + reduce of a ConvNeXt-T iteration
  // 3.01 ms against 3.51 for the 128-tile form (a quarter of the partial-sum traffic, no split at all for the layers whose tiles fill the
  // chip); with three buffers (one workgroup per CU) it loses (3.61).  KPF_WG16_FORM=128 selects the 128-tile form (tuning aid).
  static const int h16_form = []() { const char* e = getenv("KPF_WG16_FORM"); return e ? atoi(e) : 64; }();
  const bool h16 = dtype != KPF_DT_F32 && !h16_widen;
  const bool one = KH == 1 && KW == 1;
  // few pixels, many output tiles (the 4 x 4 maps of the last ConvNeXt stage: M = 512, 3072 x 768 outputs per backbone): the 128-tile kernel without a split — a
  // quarter of the workgroups of the 64-tile form, each walking all pixels, no cross-wave sum — when its tiles alone fill the chip (KPF_WG16_BIG_M: pixel limit)
  static const int big_m = []() { const char* e = getenv("KPF_WG16_BIG_M"); return e ? atoi(e) : 512; }();
  const bool big = h16 && one && M <= big_m && (long)((N + HB - 1) / HB) * ((K + HB - 1) / HB) * groups >= 256;
  const bool h16s = h16 && h16_form != 128 && !big;
  Plan p = h16s ? plan_wgrad_h16s(M, N, (int)K, groups) : (h16 ? plan_wgrad_h16(M, N, (int)K) : plan_wgrad(M, N, (int)K, one));
  if (big) p.S = 1, p.sps = (int)((M + RB - 1) / RB);
  const bool direct = one && p.S == 1 && !trimmed;  // the single partial array is dW
  const long wsg = (long)p.S * N * K + (long)p.S * N;  // one group's workspace
  KPF_REQUIRE(ws_floats >= groups * wsg, "kpf_conv2d_wgrad_f32: workspace too small (%ld floats, need %ld)", ws_floats, groups * wsg);
  WgradArgs a;
  a.dy = dy, a.x = x, a.part = direct ? dw : ws, a.dbpart = db ? (direct ? db : ws + (size_t)p.S * N * K) : nullptr;
  a.groups = groups, a.g_dy = groups > 1 ? N : 0, a.g_x = groups > 1 ? Cin : 0;
  a.g_part = groups > 1 ? (direct ? (long)N * K : wsg) : 0, a.g_db = groups > 1 ? (direct ? (long)N : wsg) : 0;
  a.zero = zero_page();
  KPF_REQUIRE(a.zero, "kpf_conv2d_wgrad_f32: cannot resolve the zero page");
  a.H = H, a.W = W, a.Cin = Cin, a.ldx = ldx, a.OH = OH, a.OW = OW, a.N = N, a.ldy = ldy, a.KH = KH, a.KW = KW;
  a.sh = sh, a.sw = sw, a.ph = ph, a.pw = pw, a.M = (int)M, a.K = (int)K, a.tilesK = p.tilesK, a.stages_per_split = p.sps;
  a.is1x1 = KH == 1 && KW == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0 && OH == H && OW == W;
  a.emul_sps = 0;
  static const int xcd_order = []() { const char* e = getenv("KPF_WG16S_XCD"); return e ? atoi(e) : 1; }();  // tuning aid: 0 = tiles in dispatch order
  a.xcd = xcd_order;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (h16s) {
    static std::atomic<bool> lds_ok[2][KPF_MAX_DEVICES];
    static const int sring = []() { const char* e = getenv("KPF_WG16S_RING"); return e ? atoi(e) : 2; }();  // tuning aid: 2 (default) or 3 buffers
    const dim3 grid(p.tilesN * p.tilesK, p.S, groups);
    constexpr int WB = 4 * 2 * 32 * 128;  // one buffer of the four waves
    if (sring == 2) {
      if (dtype == KPF_DT_BF16) hipLaunchKernelGGL((wgrad_h16s_kernel<bf16_t, 2>), grid, dim3(256), 2 * WB + 1024, st, a);
      else hipLaunchKernelGGL((wgrad_h16s_kernel<f16_t, 2>), grid, dim3(256), 2 * WB + 1024, st, a);
    } else if (dtype == KPF_DT_BF16) {
      KPF_REQUIRE(kpf_raise_lds_limit(reinterpret_cast<const void*>(&wgrad_h16s_kernel<bf16_t, 3>), lds_ok[0]), "kpf_conv2d_wgrad_h16: cannot raise the LDS limit");
      hipLaunchKernelGGL((wgrad_h16s_kernel<bf16_t, 3>), grid, dim3(256), 3 * WB + 1024, st, a);
    } else {
      KPF_REQUIRE(kpf_raise_lds_limit(reinterpret_cast<const void*>(&wgrad_h16s_kernel<f16_t, 3>), lds_ok[1]), "kpf_conv2d_wgrad_h16: cannot raise the LDS limit");
      hipLaunchKernelGGL((wgrad_h16s_kernel<f16_t, 3>), grid, dim3(256), 3 * WB + 1024, st, a);
    }
    rc = kpf_check_launch("kpf_conv2d_wgrad_h16");
  } else if (h16) {
    static const int ring = []() { const char* e = getenv("KPF_WG16_RING"); return e ? atoi(e) : 3; }();  // tuning aid: 3 or 4 stages
    const dim3 grid(p.tilesN * p.tilesK, p.S, groups);
    if (ring == 4) {
      if (dtype == KPF_DT_BF16) hipLaunchKernelGGL((wgrad_h16_kernel<bf16_t, 4>), grid, dim3(256), 4 * 2 * RB * 256, st, a);
      else hipLaunchKernelGGL((wgrad_h16_kernel<f16_t, 4>), grid, dim3(256), 4 * 2 * RB * 256, st, a);
    } else {
      if (dtype == KPF_DT_BF16) hipLaunchKernelGGL((wgrad_h16_kernel<bf16_t, 3>), grid, dim3(256), 3 * 2 * RB * 256, st, a);
      else hipLaunchKernelGGL((wgrad_h16_kernel<f16_t, 3>), grid, dim3(256), 3 * 2 * RB * 256, st, a);
    }
    rc = kpf_check_launch("kpf_conv2d_wgrad_h16");
  } else {
    rc = dtype == KPF_DT_F32 ? launch_wgrad_any<float>(a, p, st, r16)
                             : (dtype == KPF_DT_BF16 ? launch_wgrad_any<bf16_t>(a, p, st) : launch_wgrad_any<f16_t>(a, p, st));
  }
  if (rc != KPF_OK || direct) return rc;
  const long NK = (long)N * K;
  const int nkb = reduce_blocks(NK, p.S), ndb = db ? reduce_blocks(N, p.S) : 0;
  if (defer) {  // the caller batches this reduce with others (kpf_wgrad_reduce_multi): ws stays the caller's until then
    defer->part = ws, defer->dbpart = a.dbpart, defer->dw = dw, defer->db = db, defer->g_ws = wsg;
    defer->S = p.S, defer->N = N, defer->K = (int)K, defer->Cin = Cin, defer->KHW = KH * KW, defer->nkb = nkb, defer->ndb = ndb;
    defer->groups = groups, defer->Cin_out = cin_valid, defer->N_out = n_valid, defer->kind = 0, defer->first_block = 0;
    return KPF_OK;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nkb + ndb, groups), dim3(256), 0, st, ws, a.dbpart, dw, db, p.S, N, (int)K, Cin,
                     KH * KW, nkb, wsg, cin_valid, n_valid);
  return kpf_check_launch("kpf_conv2d_wgrad_f32 (reduce)");
}

int kpf_conv2d_wgrad_f32(const float* dy, const float* x, float* dw, float* db, float* ws, long ws_floats, int B, int H, int W, int Cin,
                         int ldx, int OH, int OW, int N, int ldy, int KH, int KW, int sh, int sw, int ph, int pw, void* stream) {
  return conv2d_wgrad_impl(dy, x, KPF_DT_F32, dw, db, ws, ws_floats, B, H, W, Cin, ldx, OH, OW, N, ldy, KH, KW, sh, sw, ph, pw, stream);
}

int kpf_conv2d_wgrad_h16(const void* dy, const void* x, int dtype, float* dw, float* db, float* ws, long ws_floats, int B, int H, int W, int Cin,
                         int ldx, int OH, int OW, int N, int ldy, int KH, int KW, int sh, int sw, int ph, int pw, void* stream) {
  KPF_REQUIRE(dtype == KPF_DT_BF16 || dtype == KPF_DT_F16, "kpf_conv2d_wgrad_h16: dtype must be KPF_DT_BF16 or KPF_DT_F16");
  return conv2d_wgrad_impl(dy, x, dtype, dw, db, ws, ws_floats, B, H, W, Cin, ldx, OH, OW, N, ldy, KH, KW, sh, sw, ph, pw, stream);
}

int kpf_conv2d_wgrad_groups(const void* dy, const void* x, int dtype, float* dw, float* db, float* ws, long ws_floats, int groups, int B, int H, int W, int Cin,
                            int ldx, int OH, int OW, int N, int ldy, int KH, int KW, int sh, int sw, int ph, int pw, int cin_valid, int n_valid, void* stream) {
  KPF_REQUIRE(groups >= 1 && groups <= 64, "kpf_conv2d_wgrad_groups: 1..64 groups");
  return conv2d_wgrad_impl(dy, x, dtype, dw, db, ws, ws_floats, B, H, W, Cin, ldx, OH, OW, N, ldy, KH, KW, sh, sw, ph, pw, stream, groups, cin_valid, n_valid);
}

int kpf_conv2d_wgrad_deferred(const void* dy, const void* x, int dtype, float* dw, float* db, float* ws, long ws_floats, int groups, int B, int H, int W, int Cin,
                              int ldx, int OH, int OW, int N, int ldy, int KH, int KW, int sh, int sw, int ph, int pw, int cin_valid, int n_valid,
                              kpf_wgrad_reduce_desc* reduce, void* stream) {
  KPF_REQUIRE(groups >= 1 && groups <= 64, "kpf_conv2d_wgrad_deferred: 1..64 groups");
  KPF_REQUIRE(reduce, "kpf_conv2d_wgrad_deferred: null descriptor");
  return conv2d_wgrad_impl(dy, x, dtype, dw, db, ws, ws_floats, B, H, W, Cin, ldx, OH, OW, N, ldy, KH, KW, sh, sw, ph, pw, stream, groups, cin_valid, n_valid, reduce);
}

int kpf_wgrad_reduce_multi(const kpf_wgrad_reduce_desc* descs, int n, void* stream) {
  KPF_REQUIRE(n >= 0 && (descs || n == 0), "kpf_wgrad_reduce_multi: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < n;) {
    ReduceBatch b;
    b.nd = 0;
    long blocks = 0;
    for (; base < n && b.nd < KPF_WGRAD_REDUCE_BATCH; ++base) {
      const kpf_wgrad_reduce_desc& d = descs[base];
      if (d.kind < 0) continue;  // (a call that wrote its gradient directly)
      KPF_REQUIRE((d.kind == 0 || d.kind == 1) && d.part && d.dw && d.S > 0 && d.N > 0 && d.nkb > 0 && d.ndb >= 0 && (d.ndb == 0 || (d.db && d.dbpart)),
                  "kpf_wgrad_reduce_multi: bad descriptor %d", base);
      KPF_REQUIRE(d.kind == 1 || (d.K > 0 && d.Cin > 0 && d.KHW > 0 && d.groups >= 1 && d.Cin_out > 0 && d.N_out > 0), "kpf_wgrad_reduce_multi: bad descriptor %d", base);
      b.d[b.nd] = d;
      b.d[b.nd].first_block = (int)blocks;
      blocks += (long)(d.nkb + d.ndb) * (d.kind == 1 ? 1 : d.groups);
      KPF_REQUIRE(blocks < (1L << 31), "kpf_wgrad_reduce_multi: too many blocks");
      ++b.nd;
    }
    if (b.nd == 0) break;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
    const int rc = kpf_check_launch("kpf_wgrad_reduce_multi");
    if (rc != KPF_OK) return rc;
  }
  return KPF_OK;
}

int kpf_linear_wgrad_grouped(const kpf_wgrad_group_desc* descs, int n, void* stream) {
  KPF_REQUIRE(n >= 0 && (descs || n == 0), "kpf_linear_wgrad_grouped: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < n; base += KPF_WGRAD_GROUP_BATCH) {
    WgradGroupBatch b;
    b.nd = n - base < KPF_WGRAD_GROUP_BATCH ? n - base : KPF_WGRAD_GROUP_BATCH;
    long blocks = 0;
    for (int k = 0; k < b.nd; ++k) {
      b.d[k] = descs[base + k];
      const kpf_wgrad_group_desc& d = b.d[k];
      KPF_REQUIRE(d.dy && d.x && d.dw && d.M > 0 && d.N > 0 && d.K > 0 && d.N % 4 == 0 && d.K % 4 == 0, "kpf_linear_wgrad_grouped: bad descriptor %d (N, K multiples of 4)",
                  base + k);
      KPF_REQUIRE(kpf_aligned16(d.dy) && kpf_aligned16(d.x) && kpf_aligned16(d.dw), "kpf_linear_wgrad_grouped: dy, x, dw must be 16-byte aligned (descriptor %d)", base + k);
      KPF_REQUIRE(d.ldy == 0 || (d.ldy >= d.N && d.ldy % 4 == 0), "kpf_linear_wgrad_grouped: ldy must be 0 or a multiple of 4 >= N (descriptor %d)", base + k);
      b.d[k].first_block = (int)blocks;
      b.d[k].sps = plan_wgrad(d.M, d.N, d.K, true).sps;  // (the split the per-layer form would use: its summation order is reproduced)
      blocks += (long)((d.N + 63) / 64) * ((d.K + 63) / 64);
    }
    b.zero = zero_page();
    KPF_REQUIRE(b.zero, "kpf_linear_wgrad_grouped: cannot resolve the zero page");
    hipLaunchKernelGGL(wgrad_grouped_kernel, dim3((unsigned)blocks), dim3(256), 2 * RB * 32 * (2 + 2) * 4, st, b);
    const int rc = kpf_check_launch("kpf_linear_wgrad_grouped");
    if (rc != KPF_OK) return rc;
  }
  return KPF_OK;
}

long kpf_dwconv7_wgrad_ws_floats(int B, int H, int C) {
  if (B <= 0 || H <= 0 || C <= 0) return 0;
  const int rpc = dw_chunk_rows(B, H, C), rpl = dwl_chunk_rows(B, H, C);
  const long S = ((long)B * H + rpc - 1) / rpc, Sl = ((long)B * H + rpl - 1) / rpl;  // (either kernel may take the call)
  return (S > Sl ? S : Sl) * 50 * C;
}

static int dwconv7_wgrad_impl(const float* dy, const float* x, float* dw, float* db, float* ws, long ws_floats, int B, int H, int W, int C,
                              kpf_wgrad_reduce_desc* defer, void* stream) {
  if (defer) defer->kind = -1;
  KPF_REQUIRE(dy && x && dw && ws, "kpf_dwconv7_wgrad_f32: null pointer");
  KPF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "kpf_dwconv7_wgrad_f32: bad shape (C %% 4 == 0)");
  KPF_REQUIRE(kpf_aligned16(dy) && kpf_aligned16(x) && kpf_aligned16(ws), "kpf_dwconv7_wgrad_f32: dy, x, ws must be 16-byte aligned");
  static const int lds_form = []() { const char* e = getenv("KPF_DW7_WGRAD_LDS"); return e ? atoi(e) : 1; }();  // tuning aid: 0 = the register-window kernel
  const DwlGeom gm = dwl_geom(W, C);
  bool use_lds = lds_form && gm.CQ >= 1 && (size_t)gm.lds_f4 * 16 <= 80 * 1024 && W * gm.CQ <= 512;  // (80 KB: two workgroups per CU)
  if (use_lds && (size_t)gm.lds_f4 * 16 > 64 * 1024) {  // the 32-quad channel blocks of the 8 x 8 and 4 x 4 maps: 66.5 KB
    static std::atomic<bool> lds_ok[KPF_MAX_DEVICES];
    use_lds = kpf_raise_lds_limit(reinterpret_cast<const void*>(&dwconv7_wgrad_lds_kernel), lds_ok);
  }
  const int rpc = use_lds ? dwl_chunk_rows(B, H, C) : dw_chunk_rows(B, H, C);
  const int S = (int)(((long)B * H + rpc - 1) / rpc);
  KPF_REQUIRE(ws_floats >= (long)S * 50 * C, "kpf_dwconv7_wgrad_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* dbpart = ws + (size_t)S * 49 * C;
  if (use_lds)
    hipLaunchKernelGGL(dwconv7_wgrad_lds_kernel, dim3(((C >> 2) + gm.CQ - 1) / gm.CQ, S), dim3(256), (size_t)gm.lds_f4 * 16, st, dy, x, ws, dbpart, B, H, W, C, rpc);
  else
    hipLaunchKernelGGL(dwconv7_wgrad_kernel, dim3((7 * (C / 4) + 255) / 256, S), dim3(256), 0, st, dy, x, ws, dbpart, B, H, W, C, rpc);
  int rc = kpf_check_launch("kpf_dwconv7_wgrad_f32");
  if (rc != KPF_OK) return rc;
  const int nkb = reduce_blocks(49L * C, S), ndb = db ? reduce_blocks(C, S) : 0;
  if (defer) {
    *defer = kpf_wgrad_reduce_desc{};
    defer->part = ws, defer->dbpart = dbpart, defer->dw = dw, defer->db = db;
    defer->S = S, defer->N = C, defer->nkb = nkb, defer->ndb = ndb, defer->groups = 1, defer->kind = 1;
    return KPF_OK;
  }
  hipLaunchKernelGGL(dwconv7_wgrad_reduce_kernel, dim3(nkb + ndb), dim3(256), 0, st, ws, dbpart, dw, db, S, C, nkb);
  return kpf_check_launch("kpf_dwconv7_wgrad_f32 (reduce)");
}

int kpf_dwconv7_wgrad_f32(const float* dy, const float* x, float* dw, float* db, float* ws, long ws_floats, int B, int H, int W, int C,
                          void* stream) {
  return dwconv7_wgrad_impl(dy, x, dw, db, ws, ws_floats, B, H, W, C, nullptr, stream);
}

int kpf_dwconv7_wgrad_deferred(const float* dy, const float* x, float* dw, float* db, float* ws, long ws_floats, int B, int H, int W, int C,
                               kpf_wgrad_reduce_desc* reduce, void* stream) {
  KPF_REQUIRE(reduce, "kpf_dwconv7_wgrad_deferred: null descriptor");
  return dwconv7_wgrad_impl(dy, x, dw, db, ws, ws_floats, B, H, W, C, reduce, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm (batch statistics) + optional ReLU on NHWC rows [M][C], forward and backward — the `BN -> ReLU` pairs of the
// pre-activation Residual blocks (model/hourglass.py:84-119) when the reference trains.  Three launches each way: per-chunk partial
// sums (thread = channel quad x row lane, coalesced float4 rows), a per-channel finalize that adds the partials in a fixed order,
// and the elementwise pass.  Sums are taken about the first row's value (shifted data: no cancellation for |mean| >> std).
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct BnGeom { int Q, QL, RL; };
__host__ __device__ inline BnGeom bn_geom(int C) {
  BnGeom g;
  g.Q = C >> 2;
  g.QL = g.Q < 64 ? g.Q : 64;
  g.RL = 256 / g.QL;
  return g;
}

// MODE 0: forward statistics  p0 = sum(x - K), p1 = sum((x - K)^2)          (K = x[0][c])
// MODE 1: backward sums       p0 = sum(dz),    p1 = sum(dz * (x - mean))     (dz = dy masked by y > 0 when RELU)
// TX: storage type of the layer's input side (x, dx); TY: of its output side (y, dy) — fp32 or the 16-bit type of the
// mixed-precision step; the arithmetic is fp32 either way.
template <int MODE, bool RELU, typename TX, typename TY>
__global__ __launch_bounds__(256) void bn_partial_kernel(const TX* __restrict__ x, const TY* __restrict__ dy, const TY* __restrict__ y,
                                                         const float* __restrict__ mean, float* __restrict__ ws, long M, int C,
                                                         int rows_per_chunk) {
  __shared__ f32x4 red[2][256];
  const BnGeom g = bn_geom(C);
  const int ql = threadIdx.x % g.QL, rl = threadIdx.x / g.QL;
  const int q = blockIdx.y * 64 + ql;
  const bool active = rl < g.RL && q < g.Q;
  f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const long r0 = (long)blockIdx.x * rows_per_chunk;
    const long r1 = min(M, r0 + rows_per_chunk);
    const f32x4 k = MODE == 0 ? kpf_ld4(x + 4 * q) : kpf_ld4(mean + 4 * q);
    for (long r = r0 + rl; r < r1; r += g.RL) {
      const f32x4 xv = kpf_ld4(x + r * C + 4 * q) - k;
      if (MODE == 0) {
        p0 += xv;
        p1 += xv * xv;
      } else {
        f32x4 d = kpf_ld4(dy + r * C + 4 * q);
        if (RELU) {
          const f32x4 yv = kpf_ld4(y + r * C + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = yv[e] > 0.f ? d[e] : 0.f;
        }
        p0 += d;
        p1 += d * xv;
      }
    }
  }
  red[0][threadIdx.x] = p0;
  red[1][threadIdx.x] = p1;
  __syncthreads();
  if (rl == 0 && q < g.Q) {
    f32x4 s0 = red[0][ql], s1 = red[1][ql];
    for (int i = 1; i < g.RL; ++i) {
      s0 += red[0][i * g.QL + ql];
      s1 += red[1][i * g.QL + ql];
    }
    float* o = ws + (size_t)blockIdx.x * 2 * C;
    kpf_st4(o + 4 * q, s0);
    kpf_st4(o + C + 4 * q, s1);
  }
}

// sums of the S per-chunk partials ws[p][{0,1}][C] for channel c, in a fixed order: wave g of the 16 adds p = g, g+16, ... in four
// chains (S/64 dependent loads per thread: the finalize kernels are pure latency), then the 16 wave sums are combined through LDS
constexpr int BN_FW = 16;  // waves per finalize workgroup
__device__ __forceinline__ void bn_sum_partials(const float* __restrict__ ws, int S, int C, int c, float (*red)[2][64], float& r0, float& r1) {
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
  if (c < C) {
    int p = g;
    for (; p + 3 * BN_FW < S; p += 4 * BN_FW) {
      const float* w0 = ws + (size_t)p * 2 * C + c;
      a0 += w0[0], b0 += w0[C];
      a1 += w0[(size_t)2 * BN_FW * C], b1 += w0[(size_t)(2 * BN_FW + 1) * C];
      a2 += w0[(size_t)4 * BN_FW * C], b2 += w0[(size_t)(4 * BN_FW + 1) * C];
      a3 += w0[(size_t)6 * BN_FW * C], b3 += w0[(size_t)(6 * BN_FW + 1) * C];
    }
    for (; p < S; p += BN_FW) a0 += ws[(size_t)p * 2 * C + c], b0 += ws[(size_t)p * 2 * C + C + c];
  }
  red[g][0][o] = (a0 + a1) + (a2 + a3);
  red[g][1][o] = (b0 + b1) + (b2 + b3);
  __syncthreads();
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int i = 0; i < BN_FW; ++i) {
    s0 += red[i][0][o];
    s1 += red[i][1][o];
  }
  r0 = s0;
  r1 = s1;
}

// forward finalize: batch mean / 1/sqrt(biased var + eps), running statistics (unbiased variance); 64 channels per workgroup
template <typename TX>
__global__ __launch_bounds__(64 * BN_FW) void bn_stats_finalize_kernel(const TX* __restrict__ x, const float* __restrict__ ws, int S, long M, int C,
                                                                float* __restrict__ mean, float* __restrict__ invstd,
                                                                float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps) {
  __shared__ float red[BN_FW][2][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  float s, ss;
  bn_sum_partials(ws, S, C, c, red, s, ss);
  if (threadIdx.x >= 64 || c >= C) return;
  const float n = (float)M;
  const float ms = s / n;
  const float var = fmaxf(ss / n - ms * ms, 0.f);
  const float m = (float)x[c] + ms;
  mean[c] = m;
  invstd[c] = 1.0f / sqrtf(var + eps);
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (M > 1 ? var * n / (n - 1.f) : var);
}

// backward finalize: db = sum dz, dw = invstd * sum dz (x - mean); coef[c] = (sum dz / M, invstd^2 * sum dz (x-mean) / M)
__global__ __launch_bounds__(64 * BN_FW) void bn_bwd_finalize_kernel(const float* __restrict__ ws, int S, long M, int C, const float* __restrict__ invstd,
                                                              float* __restrict__ dw, float* __restrict__ db, float* __restrict__ coef) {
  __shared__ float red[BN_FW][2][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  float s1, s2;
  bn_sum_partials(ws, S, C, c, red, s1, s2);
  if (threadIdx.x >= 64 || c >= C) return;
  const float is = invstd[c];
  if (db) db[c] = s1;
  if (dw) dw[c] = s2 * is;
  coef[c] = s1 / (float)M;
  coef[C + c] = s2 * is * is / (float)M;
}

// MODE 0: y = [relu]((x - mean) * invstd * w + b)
// MODE 1: dx = w * invstd * (dz - coef0 - (x - mean) * coef1)
// (MODE 0 writes y: TY; MODE 1 writes dx: TX)
template <int MODE, bool RELU, typename TX, typename TY>
__global__ __launch_bounds__(256) void bn_elem_kernel(const TX* __restrict__ x, const TY* __restrict__ dy, const TY* __restrict__ y,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ coef,
                                                      typename std::conditional<MODE == 0, TY, TX>::type* __restrict__ out, long M, int C,
                                                      int rows_per_block, const TX* __restrict__ addend = nullptr) {
  // addend (MODE 1, nullable): a second gradient of x — the skip path of a Residual block — added here instead of by a separate launch
  const BnGeom g = bn_geom(C);
  const int ql = threadIdx.x % g.QL, rl = threadIdx.x / g.QL;
  const int q = blockIdx.y * 64 + ql;
  if (rl >= g.RL || q >= g.Q) return;
  const f32x4 mu = kpf_ld4(mean + 4 * q), is = kpf_ld4(invstd + 4 * q), wv = kpf_ld4(w + 4 * q);
  const f32x4 a = is * wv;
  f32x4 t0, t1;
  if (MODE == 0) t0 = kpf_ld4(b + 4 * q);
  else t0 = kpf_ld4(coef + 4 * q), t1 = kpf_ld4(coef + C + 4 * q);
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(M, r0 + rows_per_block);
  for (long r = r0 + rl; r < r1; r += g.RL) {
    const f32x4 xv = kpf_ld4(x + r * C + 4 * q) - mu;
    f32x4 o;
    if (MODE == 0) {
      o = xv * a + t0;
      if (RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
      }
    } else {
      f32x4 d = kpf_ld4(dy + r * C + 4 * q);
      if (RELU) {
        const f32x4 yv = kpf_ld4(y + r * C + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = yv[e] > 0.f ? d[e] : 0.f;
      }
      o = a * (d - t0 - xv * t1);
      if (addend) o += kpf_ld4(addend + r * C + 4 * q);
    }
    kpf_st4(out + r * C + 4 * q, o);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: out = relu(BN_a(xa) + BN_b(xb)) — DESA's `relu(bn_l0(conv_l0(offsets)) + bn_f0(conv_f0(features)))` (model/model.py:176-190) on fp32 rows [M][C] — with the
// two normalisations, the sum and the ReLU in ONE pass over the two pre-activations (the three-kernel form wrote both normalised tensors and read them back:
// 264 MB of 43008 x 384 fp32 rows per fusion block), and the backward's masked gradient, both branches' partial sums and both input gradients in two passes
// instead of seven.  Geometry, chunking and the fixed-order finalize are the BatchNorm kernels' above (their workspaces: [S][2][C] per branch).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn2_add_relu_fwd_kernel(const float* __restrict__ xa, const float* __restrict__ xb, const float* __restrict__ stats /* [4][C]: mean_a, invstd_a, mean_b, invstd_b */,
                                                               const float* __restrict__ wa, const float* __restrict__ ba, const float* __restrict__ wb, const float* __restrict__ bb,
                                                               float* __restrict__ out, long M, int C, int rows_per_block) {
  const BnGeom g = bn_geom(C);
  const int ql = threadIdx.x % g.QL, rl = threadIdx.x / g.QL;
  const int q = blockIdx.y * 64 + ql;
  if (rl >= g.RL || q >= g.Q) return;
  const f32x4 ma = kpf_ld4(stats + 4 * q), sa = kpf_ld4(stats + C + 4 * q) * kpf_ld4(wa + 4 * q);
  const f32x4 mb = kpf_ld4(stats + 2 * C + 4 * q), sb = kpf_ld4(stats + 3 * C + 4 * q) * kpf_ld4(wb + 4 * q);
  const f32x4 t0 = kpf_ld4(ba + 4 * q) + kpf_ld4(bb + 4 * q);
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (long r = r0 + rl; r < r1; r += g.RL) {
    const f32x4 va = kpf_ld4(xa + r * C + 4 * q) - ma, vb = kpf_ld4(xb + r * C + 4 * q) - mb;
    f32x4 o = va * sa + (vb * sb + t0);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    kpf_st4(out + r * C + 4 * q, o);
  }
}

// partial sums of the backward for BOTH branches: dz = out > 0 ? dy : 0;  wsa[chunk] = {sum dz, sum dz (xa - mean_a)}, wsb[chunk] = {sum dz, sum dz (xb - mean_b)}
__global__ __launch_bounds__(256) void bn2_partial_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out, const float* __restrict__ xa, const float* __restrict__ xb,
                                                              const float* __restrict__ stats, float* __restrict__ wsa, float* __restrict__ wsb, long M, int C, int rows_per_chunk) {
  __shared__ f32x4 red[3][256];
  const BnGeom g = bn_geom(C);
  const int ql = threadIdx.x % g.QL, rl = threadIdx.x / g.QL;
  const int q = blockIdx.y * 64 + ql;
  const bool active = rl < g.RL && q < g.Q;
  f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const f32x4 ma = kpf_ld4(stats + 4 * q), mb = kpf_ld4(stats + 2 * C + 4 * q);
    for (long r = r0 + rl; r < r1; r += g.RL) {
      f32x4 d = kpf_ld4(dy + r * C + 4 * q);
      const f32x4 o = kpf_ld4(out + r * C + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
      p0 += d;
      pa += d * (kpf_ld4(xa + r * C + 4 * q) - ma);
      pb += d * (kpf_ld4(xb + r * C + 4 * q) - mb);
    }
  }
  red[0][threadIdx.x] = p0;
  red[1][threadIdx.x] = pa;
  red[2][threadIdx.x] = pb;
  __syncthreads();
  if (rl == 0 && q < g.Q) {
    f32x4 s0 = red[0][ql], s1 = red[1][ql], s2 = red[2][ql];
    for (int i = 1; i < g.RL; ++i) {
      s0 += red[0][i * g.QL + ql];
      s1 += red[1][i * g.QL + ql];
      s2 += red[2][i * g.QL + ql];
    }
    float* oa = wsa + (size_t)blockIdx.x * 2 * C;
    float* ob = wsb + (size_t)blockIdx.x * 2 * C;
    kpf_st4(oa + 4 * q, s0);
    kpf_st4(oa + C + 4 * q, s1);
    kpf_st4(ob + 4 * q, s0);
    kpf_st4(ob + C + 4 * q, s2);
  }
}

// dxa = wa invstd_a (dz - c0a - (xa - mean_a) c1a), dxb likewise (coef_* from bn_bwd_finalize_kernel)
__global__ __launch_bounds__(256) void bn2_elem_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out, const float* __restrict__ xa, const float* __restrict__ xb,
                                                           const float* __restrict__ stats, const float* __restrict__ wa, const float* __restrict__ wb,
                                                           const float* __restrict__ coefa, const float* __restrict__ coefb, float* __restrict__ dxa, float* __restrict__ dxb,
                                                           long M, int C, int rows_per_block) {
  const BnGeom g = bn_geom(C);
  const int ql = threadIdx.x % g.QL, rl = threadIdx.x / g.QL;
  const int q = blockIdx.y * 64 + ql;
  if (rl >= g.RL || q >= g.Q) return;
  const f32x4 ma = kpf_ld4(stats + 4 * q), sa = kpf_ld4(stats + C + 4 * q) * kpf_ld4(wa + 4 * q);
  const f32x4 mb = kpf_ld4(stats + 2 * C + 4 * q), sb = kpf_ld4(stats + 3 * C + 4 * q) * kpf_ld4(wb + 4 * q);
  const f32x4 a0 = kpf_ld4(coefa + 4 * q), a1 = kpf_ld4(coefa + C + 4 * q), b0 = kpf_ld4(coefb + 4 * q), b1 = kpf_ld4(coefb + C + 4 * q);
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (long r = r0 + rl; r < r1; r += g.RL) {
    f32x4 d = kpf_ld4(dy + r * C + 4 * q);
    const f32x4 o = kpf_ld4(out + r * C + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
    const f32x4 va = kpf_ld4(xa + r * C + 4 * q) - ma, vb = kpf_ld4(xb + r * C + 4 * q) - mb;
    kpf_st4(dxa + r * C + 4 * q, sa * (d - a0 - va * a1));
    kpf_st4(dxb + r * C + 4 * q, sb * (d - b0 - vb * b1));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: y[g][c] = max over the `group` consecutive rows of relu(BN(x)) — DESA's `bn_blocks -> ReLU -> max over a ball's 64 members` (model/model.py:188-192) — with
// the normalisation, the ReLU and the maximum in ONE pass over the pre-activation (the normalised 43008 x 384 tensor is never written), the winner kept like
// group_max_train_fwd_kernel (first maximum; all members <= 0: member 0 with value 0).  Backward: only the winners carry a gradient, so the two BatchNorm sums
// are sums over G = rows / group entries per channel (x gathered at the winner's row) instead of passes over the tensor, and dx = w invstd (dz - c0 - xhat c1) is
// one dense pass that rebuilds dz from (arg, y > 0).  Traffic per fusion block: 0.86 GB -> 0.26 GB.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_relu_gmax_fwd_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ w, const float* __restrict__ b,
                                                               float* __restrict__ y, unsigned char* __restrict__ arg, long n4, int group, int C4) {
  const int C = 4 * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const int q = (int)(i - r * C4);
    const f32x4 mu = kpf_ld4(stats + 4 * q), a = kpf_ld4(stats + C + 4 * q) * kpf_ld4(w + 4 * q), be = kpf_ld4(b + 4 * q);
    const float* p = x + (r * group * C4 + q) * 4;
    f32x4 best = {0.f, 0.f, 0.f, 0.f};
    int bi[4] = {0, 0, 0, 0};
#pragma unroll 8
    for (int m = 0; m < group; ++m) {
      const f32x4 v = (kpf_ld4(p + (long)m * C4 * 4) - mu) * a + be;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float z = fmaxf(v[t], 0.f);
        if (m == 0 || z > best[t]) {  // strictly greater: the first maximum wins
          best[t] = z;
          bi[t] = m;
        }
      }
    }
    kpf_st4(y + i * 4, best);
    *reinterpret_cast<uchar4*>(arg + i * 4) = uchar4{(unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]};
  }
}

// partial sums over chunks of groups: ws[chunk] = {sum dz, sum dz (x_winner - mean)} per channel, dz = y > 0 ? dmax : 0;  grid (ceil(C / 64), S), thread = (channel, 4 lanes)
__global__ __launch_bounds__(256) void bn_gmax_partial_bwd_kernel(const float* __restrict__ dmx, int dmx_ld, const float* __restrict__ y, const unsigned char* __restrict__ arg,
                                                                  const float* __restrict__ x, const float* __restrict__ stats, float* __restrict__ ws, long G, int group, int C,
                                                                  int groups_per_chunk) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long g0 = (long)blockIdx.y * groups_per_chunk, g1 = min(G, g0 + groups_per_chunk);
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float mu = stats[c];
    for (long g = g0 + gl; g < g1; g += 4) {
      const float dz = y[g * C + c] > 0.f ? dmx[g * dmx_ld + c] : 0.f;
      const float xv = x[(g * group + arg[g * C + c]) * C + c] - mu;
      s0 += dz;
      s1 = fmaf(dz, xv, s1);
    }
  }
  red[0][gl][cl] = s0;
  red[1][gl][cl] = s1;
  __syncthreads();
  if (gl == 0 && c < C) {
    float* o = ws + (size_t)blockIdx.y * 2 * C;
    o[c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    o[C + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  }
}

// dx[r][c] = w invstd (dz - c0 - (x - mean) c1), dz = (member of r == winner and y > 0) ? dmax : 0
__global__ __launch_bounds__(256) void bn_gmax_elem_bwd_kernel(const float* __restrict__ dmx, int dmx_ld, const float* __restrict__ y, const unsigned char* __restrict__ arg,
                                                               const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ w, const float* __restrict__ coef,
                                                               float* __restrict__ dx, long n4, int group, int C4) {
  const int C = 4 * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {  // i over [G][group][C4]
    const int q = (int)(i % C4);
    const long rm = i / C4;
    const int m = (int)(rm % group);
    const long g = rm / group;
    const f32x4 dm = kpf_ld4(dmx + g * dmx_ld + 4 * q), yv = kpf_ld4(y + (g * C4 + q) * 4);
    const uchar4 a = *reinterpret_cast<const uchar4*>(arg + (g * C4 + q) * 4);
    const f32x4 dz = {(a.x == m && yv[0] > 0.f) ? dm[0] : 0.f, (a.y == m && yv[1] > 0.f) ? dm[1] : 0.f, (a.z == m && yv[2] > 0.f) ? dm[2] : 0.f,
                      (a.w == m && yv[3] > 0.f) ? dm[3] : 0.f};
    const f32x4 xv = kpf_ld4(x + i * 4) - kpf_ld4(stats + 4 * q);
    const f32x4 sc = kpf_ld4(stats + C + 4 * q) * kpf_ld4(w + 4 * q);
    kpf_st4(dx + i * 4, sc * (dz - kpf_ld4(coef + 4 * q) - xv * kpf_ld4(coef + C + 4 * q)));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: the embedding sums of a fusion block with their BatchNorms inside: x [rows][n C] holds n = n1 + n2 <= 4 sibling Linear outputs side by side,
//   out [rows][C] = relu(S1)  (n2 == 0)   or   relu(relu(S1) + S2),   S1 = sum of BN_k(x_k) for k < n1,  S2 = sum for k >= n1      (model/model.py:254-259, 417-422)
// in ONE pass over the pre-activations (BatchNorm pass + SlicesSumRelu wrote and re-read the normalised [rows][n C] tensor), and the backward — the masks, every
// branch's two BatchNorm sums, every branch's input gradient — in two passes.  thread = (row lane, channel quad of C), all n branches; C4 <= 64.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SSR_MAXN = 4;
__device__ __forceinline__ void ssr_norm(const float* __restrict__ xr, const float* __restrict__ stats, const float* __restrict__ w, const float* __restrict__ b, int nC, int C, int q,
                                         int n, f32x4* xc, f32x4* yv) {  // xc = x - mean, yv = BN(x) for the n branches of (row, quad)
#pragma unroll
  for (int k = 0; k < SSR_MAXN; ++k)
    if (k < n) {
      const int col = k * C + 4 * q;
      xc[k] = kpf_ld4(xr + col) - kpf_ld4(stats + col);
      yv[k] = xc[k] * (kpf_ld4(stats + nC + col) * kpf_ld4(w + col)) + kpf_ld4(b + col);
    }
}
__global__ __launch_bounds__(256) void bn_ssr_fwd_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ w, const float* __restrict__ b,
                                                         float* __restrict__ out, long n4, int C4, int n1, int n2) {
  const int C = 4 * C4, n = n1 + n2, nC = n * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const int q = (int)(i - r * C4);
    f32x4 xc[SSR_MAXN], yv[SSR_MAXN];
    ssr_norm(x + r * nC, stats, w, b, nC, C, q, n, xc, yv);
    f32x4 s1 = yv[0];
#pragma unroll
    for (int k = 1; k < SSR_MAXN; ++k)
      if (k < n1) s1 += yv[k];
    f32x4 o = {fmaxf(s1[0], 0.f), fmaxf(s1[1], 0.f), fmaxf(s1[2], 0.f), fmaxf(s1[3], 0.f)};
    if (n2 > 0) {
#pragma unroll
      for (int k = 1; k < SSR_MAXN; ++k)
        if (k >= n1 && k < n) o += yv[k];
      o = f32x4{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
    }
    kpf_st4(out + i * 4, o);
  }
}
// the gradient of every branch's BN output at (row, quad): d = dout (out > 0); branches < n1 additionally masked by S1 > 0 when n2 > 0
__device__ __forceinline__ void ssr_masks(const f32x4 g, const f32x4 o, const f32x4* yv, int n1, int n2, f32x4& d, f32x4& d1) {
  d = f32x4{o[0] > 0.f ? g[0] : 0.f, o[1] > 0.f ? g[1] : 0.f, o[2] > 0.f ? g[2] : 0.f, o[3] > 0.f ? g[3] : 0.f};
  d1 = d;
  if (n2 > 0) {
    f32x4 s1 = yv[0];
#pragma unroll
    for (int k = 1; k < SSR_MAXN; ++k)
      if (k < n1) s1 += yv[k];
    d1 = f32x4{s1[0] > 0.f ? d[0] : 0.f, s1[1] > 0.f ? d[1] : 0.f, s1[2] > 0.f ? d[2] : 0.f, s1[3] > 0.f ? d[3] : 0.f};
  }
}
// grid (S): 256 threads = C4 quads x RL row lanes (RL = 256 / C4 rounded down); ws[chunk] = [2][n C]
__global__ __launch_bounds__(256) void bn_ssr_partial_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ x,
                                                                 const float* __restrict__ stats, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ ws,
                                                                 long rows, int C4, int n1, int n2, int rows_per_chunk) {
  extern __shared__ f32x4 ssr_red[];  // [2 n][256]
  const int C = 4 * C4, n = n1 + n2, nC = n * C;
  const int RL = 256 / C4;
  const int q = threadIdx.x % C4, rl = threadIdx.x / C4;
  const bool active = rl < RL;
  f32x4 p0[SSR_MAXN], p1[SSR_MAXN];
#pragma unroll
  for (int k = 0; k < SSR_MAXN; ++k) p0[k] = p1[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (active) {
    const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    for (long r = r0 + rl; r < r1; r += RL) {
      f32x4 xc[SSR_MAXN], yv[SSR_MAXN], d, d1;
      ssr_norm(x + r * nC, stats, w, b, nC, C, q, n, xc, yv);
      ssr_masks(kpf_ld4(dout + (r * C4 + q) * 4), kpf_ld4(out + (r * C4 + q) * 4), yv, n1, n2, d, d1);
#pragma unroll
      for (int k = 0; k < SSR_MAXN; ++k)
        if (k < n) {
          const f32x4 dk = k < n1 ? d1 : d;
          p0[k] += dk;
          p1[k] += dk * xc[k];
        }
    }
  }
#pragma unroll
  for (int k = 0; k < SSR_MAXN; ++k)
    if (k < n) {
      ssr_red[(2 * k) * 256 + threadIdx.x] = p0[k];
      ssr_red[(2 * k + 1) * 256 + threadIdx.x] = p1[k];
    }
  __syncthreads();
  if (rl == 0) {
    float* o = ws + (size_t)blockIdx.x * 2 * nC;
    for (int k = 0; k < n; ++k) {
      f32x4 s0 = ssr_red[(2 * k) * 256 + q], s1 = ssr_red[(2 * k + 1) * 256 + q];
      for (int i = 1; i < RL; ++i) {
        s0 += ssr_red[(2 * k) * 256 + i * C4 + q];
        s1 += ssr_red[(2 * k + 1) * 256 + i * C4 + q];
      }
      kpf_st4(o + k * C + 4 * q, s0);
      kpf_st4(o + nC + k * C + 4 * q, s1);
    }
  }
}
__global__ __launch_bounds__(256) void bn_ssr_elem_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ x,
                                                              const float* __restrict__ stats, const float* __restrict__ w, const float* __restrict__ b,
                                                              const float* __restrict__ coef, float* __restrict__ dx, long n4, int C4, int n1, int n2) {
  const int C = 4 * C4, n = n1 + n2, nC = n * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const int q = (int)(i - r * C4);
    f32x4 xc[SSR_MAXN], yv[SSR_MAXN], d, d1;
    ssr_norm(x + r * nC, stats, w, b, nC, C, q, n, xc, yv);
    ssr_masks(kpf_ld4(dout + i * 4), kpf_ld4(out + i * 4), yv, n1, n2, d, d1);
#pragma unroll
    for (int k = 0; k < SSR_MAXN; ++k)
      if (k < n) {
        const int col = k * C + 4 * q;
        const f32x4 dk = k < n1 ? d1 : d;
        const f32x4 sc = kpf_ld4(stats + nC + col) * kpf_ld4(w + col);
        kpf_st4(dx + r * nC + col, sc * (dk - kpf_ld4(coef + col) - xc[k] * kpf_ld4(coef + nC + col)));
      }
  }
}

int bn_chunks(long M, int C, int* rows_per_chunk) {
  const BnGeom g = bn_geom(C);
  const int cg = (g.Q + 63) / 64;
  static const int target = []() { const char* e = getenv("KPF_BN_TARGET"); return e ? atoi(e) : 1024; }();  // tuning aid: workgroups aimed at (round 6: 512 -> 1024, - 0.1 ms per training iteration)
  long S = target / cg;  // ~4 workgroups per CU
  const long smax = (M + 4L * g.RL - 1) / (4L * g.RL);  // at least four rows per thread
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  *rows_per_chunk = (int)((M + S - 1) / S);
  return (int)((M + *rows_per_chunk - 1) / *rows_per_chunk);
}

}  // namespace

extern "C" {

long kpf_bn_ws_floats(long M, int C) {
  if (M <= 0 || C <= 0) return 0;
  int rpc;
  const int S = bn_chunks(M, C, &rpc);
  return (long)S * 2 * C + 2 * C;
}


/* the embedding sums with their BatchNorms inside (see bn_ssr_fwd_kernel): workspace floats for either direction */
static int ssr_chunks(long rows, int C4, int* rpc) {
  const int RL = 256 / C4;
  long S = 1024;
  const long smax = (rows + 2L * RL - 1) / (2L * RL);  // at least two rows per thread
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  *rpc = (int)((rows + S - 1) / S);
  return (int)((rows + *rpc - 1) / *rpc);
}
long kpf_bn_ssr_ws_floats(long rows, int C, int n) {
  if (rows <= 0 || C <= 0 || n <= 0 || C % 4) return 0;
  int rpc, rp2;
  const long S1 = bn_chunks(rows, n * C, &rpc), S2 = ssr_chunks(rows, C / 4, &rp2);
  return (S1 > S2 ? S1 : S2) * 2 * n * C + 2L * n * C;
}

int kpf_bn_ssr_forward(const float* x, const float* w, const float* b, float* out, float* stats, float* rmean, float* rvar, float momentum, float eps, float* ws,
                       long ws_floats, long rows, int C, int n1, int n2, void* stream) {
  const int n = n1 + n2;
  KPF_REQUIRE(x && w && b && out && stats && ws && rows > 0 && C > 0 && C % 4 == 0 && C / 4 <= 64 && n1 >= 1 && n2 >= 0 && n <= SSR_MAXN,
              "kpf_bn_ssr_forward: bad arguments (C %% 4 == 0, C <= 256, 1 <= n1, n1 + n2 <= 4)");
  KPF_REQUIRE(kpf_aligned16(x) && kpf_aligned16(out) && kpf_aligned16(stats) && kpf_aligned16(ws) && kpf_aligned16(w) && kpf_aligned16(b), "kpf_bn_ssr_forward: pointers must be 16-byte aligned");
  const int nC = n * C;
  int rpc;
  const int S = bn_chunks(rows, nC, &rpc);
  KPF_REQUIRE(ws_floats >= (long)S * 2 * nC, "kpf_bn_ssr_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int cg = (nC / 4 + 63) / 64;
  hipLaunchKernelGGL((bn_partial_kernel<0, false, float, float>), dim3(S, cg), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, ws, rows, nC,
                     rpc);
  hipLaunchKernelGGL(bn_stats_finalize_kernel<float>, dim3((nC + 63) / 64), dim3(64 * BN_FW), 0, st, x, ws, S, rows, nC, stats, stats + nC, rmean, rvar, momentum, eps);
  const long n4 = rows * (C / 4);
  long nb = (n4 + 255) / 256;
  hipLaunchKernelGGL(bn_ssr_fwd_kernel, dim3((unsigned)(nb > 65535 ? 65535 : nb)), dim3(256), 0, st, x, stats, w, b, out, n4, C / 4, n1, n2);
  return kpf_check_launch("kpf_bn_ssr_forward");
}

int kpf_bn_ssr_backward(const float* dout, const float* out, const float* x, const float* stats, const float* w, const float* b, float* dx, float* dw, float* db, float* ws,
                        long ws_floats, long rows, int C, int n1, int n2, void* stream) {
  const int n = n1 + n2;
  KPF_REQUIRE(dout && out && x && stats && w && b && dx && dw && db && ws && rows > 0 && C > 0 && C % 4 == 0 && C / 4 <= 64 && n1 >= 1 && n2 >= 0 && n <= SSR_MAXN,
              "kpf_bn_ssr_backward: bad arguments");
  KPF_REQUIRE(kpf_aligned16(dout) && kpf_aligned16(out) && kpf_aligned16(x) && kpf_aligned16(dx) && kpf_aligned16(stats) && kpf_aligned16(ws) && kpf_aligned16(w) && kpf_aligned16(b),
              "kpf_bn_ssr_backward: pointers must be 16-byte aligned");
  const int nC = n * C;
  int rpc;
  const int S = ssr_chunks(rows, C / 4, &rpc);
  KPF_REQUIRE(ws_floats >= (long)S * 2 * nC + 2 * nC, "kpf_bn_ssr_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* coef = ws + (size_t)S * 2 * nC;
  hipLaunchKernelGGL(bn_ssr_partial_bwd_kernel, dim3(S), dim3(256), (size_t)2 * n * 256 * sizeof(f32x4), st, dout, out, x, stats, w, b, ws, rows, C / 4, n1, n2, rpc);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((nC + 63) / 64), dim3(64 * BN_FW), 0, st, ws, S, rows, nC, stats + nC, dw, db, coef);
  const long n4 = rows * (C / 4);
  long nb = (n4 + 255) / 256;
  hipLaunchKernelGGL(bn_ssr_elem_bwd_kernel, dim3((unsigned)(nb > 65535 ? 65535 : nb)), dim3(256), 0, st, dout, out, x, stats, w, b, coef, dx, n4, C / 4, n1, n2);
  return kpf_check_launch("kpf_bn_ssr_backward");
}

/* BatchNorm + ReLU + maximum over `group` consecutive rows (see bn_relu_gmax_fwd_kernel): workspace floats for either direction */
long kpf_bn_relu_gmax_ws_floats(long M, int C) {
  if (M <= 0 || C <= 0) return 0;
  int rpc;
  const int S = bn_chunks(M, C, &rpc);
  const long f = (long)S * 2 * C + 2 * C, bwd = 64L * 2 * C + 2 * C;
  return f > bwd ? f : bwd;
}

int kpf_bn_relu_gmax_forward(const float* x, const float* w, const float* b, float* y, unsigned char* arg, float* stats, float* rmean, float* rvar, float momentum,
                             float eps, float* ws, long ws_floats, long M, int group, int C, void* stream) {
  KPF_REQUIRE(x && w && b && y && arg && stats && ws && M > 0 && group > 0 && group <= 256 && M % group == 0 && C > 0 && C % 4 == 0,
              "kpf_bn_relu_gmax_forward: bad arguments (C %% 4 == 0, M %% group == 0, group <= 256)");
  KPF_REQUIRE(kpf_aligned16(x) && kpf_aligned16(y) && kpf_aligned16(stats) && kpf_aligned16(ws), "kpf_bn_relu_gmax_forward: pointers must be 16-byte aligned");
  int rpc;
  const int S = bn_chunks(M, C, &rpc);
  KPF_REQUIRE(ws_floats >= (long)S * 2 * C, "kpf_bn_relu_gmax_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int cg = (C / 4 + 63) / 64;
  hipLaunchKernelGGL((bn_partial_kernel<0, false, float, float>), dim3(S, cg), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, ws, M, C, rpc);
  hipLaunchKernelGGL(bn_stats_finalize_kernel<float>, dim3((C + 63) / 64), dim3(64 * BN_FW), 0, st, x, ws, S, M, C, stats, stats + C, rmean, rvar, momentum, eps);
  const long n4 = (M / group) * (C / 4);
  long nb = (n4 + 255) / 256;
  hipLaunchKernelGGL(bn_relu_gmax_fwd_kernel, dim3((unsigned)(nb > 65535 ? 65535 : nb)), dim3(256), 0, st, x, stats, w, b, y, arg, n4, group, C / 4);
  return kpf_check_launch("kpf_bn_relu_gmax_forward");
}

int kpf_bn_relu_gmax_backward(const float* dmax, int dmax_ld, const float* y, const unsigned char* arg, const float* x, const float* stats, const float* w, float* dx,
                              float* dw, float* db, float* ws, long ws_floats, long M, int group, int C, void* stream) {
  KPF_REQUIRE(dmax && y && arg && x && stats && w && dx && dw && db && ws && M > 0 && group > 0 && group <= 256 && M % group == 0 && C > 0 && C % 4 == 0 && dmax_ld >= C &&
                  dmax_ld % 4 == 0,
              "kpf_bn_relu_gmax_backward: bad arguments");
  KPF_REQUIRE(kpf_aligned16(dmax) && kpf_aligned16(y) && kpf_aligned16(x) && kpf_aligned16(dx) && kpf_aligned16(stats) && kpf_aligned16(ws),
              "kpf_bn_relu_gmax_backward: pointers must be 16-byte aligned");
  const long G = M / group;
  long S = G < 64 ? G : 64;  // chunks of groups: their partial sums are added in chunk order by bn_bwd_finalize_kernel
  const int gpc = (int)((G + S - 1) / S);
  S = (G + gpc - 1) / gpc;
  KPF_REQUIRE(ws_floats >= S * 2 * C + 2 * C, "kpf_bn_relu_gmax_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* coef = ws + (size_t)S * 2 * C;
  hipLaunchKernelGGL(bn_gmax_partial_bwd_kernel, dim3((C + 63) / 64, (unsigned)S), dim3(256), 0, st, dmax, dmax_ld, y, arg, x, stats, ws, G, group, C, gpc);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64 * BN_FW), 0, st, ws, (int)S, M, C, stats + C, dw, db, coef);
  const long n4 = M * (C / 4);
  long nb = (n4 + 255) / 256;
  hipLaunchKernelGGL(bn_gmax_elem_bwd_kernel, dim3((unsigned)(nb > 262144 ? 262144 : nb)), dim3(256), 0, st, dmax, dmax_ld, y, arg, x, stats, w, coef, dx, n4, group, C / 4);
  return kpf_check_launch("kpf_bn_relu_gmax_backward");
}

/* the fused two-branch form (see bn2_add_relu_fwd_kernel): workspace floats */
long kpf_bn2_ws_floats(long M, int C) {
  if (M <= 0 || C <= 0) return 0;
  int rpc;
  const int S = bn_chunks(M, C, &rpc);
  return 2 * ((long)S * 2 * C + 2 * C);
}

int kpf_bn2_add_relu_forward(const float* xa, const float* xb, const float* wa, const float* ba, const float* wb, const float* bb, float* out, float* stats,
                             float* rmean_a, float* rvar_a, float* rmean_b, float* rvar_b, float momentum, float eps, float* ws, long ws_floats, long M, int C,
                             void* stream) {
  KPF_REQUIRE(xa && xb && wa && ba && wb && bb && out && stats && ws && M > 0 && C > 0 && C % 4 == 0, "kpf_bn2_add_relu_forward: bad arguments (C %% 4 == 0)");
  KPF_REQUIRE(kpf_aligned16(xa) && kpf_aligned16(xb) && kpf_aligned16(out) && kpf_aligned16(stats) && kpf_aligned16(ws), "kpf_bn2_add_relu_forward: pointers must be 16-byte aligned");
  int rpc;
  const int S = bn_chunks(M, C, &rpc);
  KPF_REQUIRE(ws_floats >= (long)S * 2 * C, "kpf_bn2_add_relu_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int cg = (C / 4 + 63) / 64;
  const float* xs[2] = {xa, xb};
  float* rm[2] = {rmean_a, rmean_b};
  float* rv[2] = {rvar_a, rvar_b};
  for (int k = 0; k < 2; ++k) {  // (the two statistics passes reuse one workspace: same stream)
    hipLaunchKernelGGL((bn_partial_kernel<0, false, float, float>), dim3(S, cg), dim3(256), 0, st, xs[k], (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, ws, M,
                       C, rpc);
    hipLaunchKernelGGL(bn_stats_finalize_kernel<float>, dim3((C + 63) / 64), dim3(64 * BN_FW), 0, st, xs[k], ws, S, M, C, stats + 2 * k * C, stats + (2 * k + 1) * C, rm[k],
                       rv[k], momentum, eps);
  }
  hipLaunchKernelGGL(bn2_add_relu_fwd_kernel, dim3(S, cg), dim3(256), 0, st, xa, xb, stats, wa, ba, wb, bb, out, M, C, rpc);
  return kpf_check_launch("kpf_bn2_add_relu_forward");
}

int kpf_bn2_add_relu_backward(const float* dy, const float* out, const float* xa, const float* xb, const float* stats, const float* wa, const float* wb, float* dxa,
                              float* dxb, float* dwa, float* dba, float* dwb, float* dbb, float* ws, long ws_floats, long M, int C, void* stream) {
  KPF_REQUIRE(dy && out && xa && xb && stats && wa && wb && dxa && dxb && dwa && dba && dwb && dbb && ws && M > 0 && C > 0 && C % 4 == 0,
              "kpf_bn2_add_relu_backward: bad arguments (C %% 4 == 0)");
  KPF_REQUIRE(kpf_aligned16(dy) && kpf_aligned16(out) && kpf_aligned16(xa) && kpf_aligned16(xb) && kpf_aligned16(dxa) && kpf_aligned16(dxb) && kpf_aligned16(ws),
              "kpf_bn2_add_relu_backward: pointers must be 16-byte aligned");
  int rpc;
  const int S = bn_chunks(M, C, &rpc);
  const long per = (long)S * 2 * C + 2 * C;
  KPF_REQUIRE(ws_floats >= 2 * per, "kpf_bn2_add_relu_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int cg = (C / 4 + 63) / 64;
  float* wsa = ws;
  float* wsb = ws + per;
  float* coefa = wsa + (size_t)S * 2 * C;
  float* coefb = wsb + (size_t)S * 2 * C;
  hipLaunchKernelGGL(bn2_partial_bwd_kernel, dim3(S, cg), dim3(256), 0, st, dy, out, xa, xb, stats, wsa, wsb, M, C, rpc);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64 * BN_FW), 0, st, wsa, S, M, C, stats + C, dwa, dba, coefa);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64 * BN_FW), 0, st, wsb, S, M, C, stats + 3 * C, dwb, dbb, coefb);
  hipLaunchKernelGGL(bn2_elem_bwd_kernel, dim3(S, cg), dim3(256), 0, st, dy, out, xa, xb, stats, wa, wb, coefa, coefb, dxa, dxb, M, C, rpc);
  return kpf_check_launch("kpf_bn2_add_relu_backward");
}

}  // extern "C"

template <typename TX, typename TY>
static int bn_forward_impl(const void* xv, const float* w, const float* b, void* yv, float* mean, float* invstd, float* running_mean,
                           float* running_var, float momentum, float eps, int relu, float* ws, long ws_floats, long M, int C, void* stream) {
  const TX* x = static_cast<const TX*>(xv);
  TY* y = static_cast<TY*>(yv);
  int rpc;
  const int S = bn_chunks(M, C, &rpc);
  KPF_REQUIRE(ws_floats >= (long)S * 2 * C + 2 * C, "kpf_bn_train_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int cg = (C / 4 + 63) / 64;
  hipLaunchKernelGGL((bn_partial_kernel<0, false, TX, TY>), dim3(S, cg), dim3(256), 0, st, x, (const TY*)nullptr, (const TY*)nullptr,
                     (const float*)nullptr, ws, M, C, rpc);
  hipLaunchKernelGGL(bn_stats_finalize_kernel<TX>, dim3((C + 63) / 64), dim3(64 * BN_FW), 0, st, x, ws, S, M, C, mean, invstd, running_mean, running_var,
                     momentum, eps);
  if (relu)
    hipLaunchKernelGGL((bn_elem_kernel<0, true, TX, TY>), dim3(S, cg), dim3(256), 0, st, x, (const TY*)nullptr, (const TY*)nullptr, mean, invstd, w, b,
                       (const float*)nullptr, y, M, C, rpc);
  else
    hipLaunchKernelGGL((bn_elem_kernel<0, false, TX, TY>), dim3(S, cg), dim3(256), 0, st, x, (const TY*)nullptr, (const TY*)nullptr, mean, invstd, w, b,
                       (const float*)nullptr, y, M, C, rpc);
  return kpf_check_launch("kpf_bn_train_forward");
}

template <typename TX, typename TY>
static int bn_backward_impl(const void* dyv, const void* xv, const void* yv, const float* mean, const float* invstd, const float* w, void* dxv,
                            float* dw, float* db, int relu, float* ws, long ws_floats, long M, int C, void* stream, const void* addv) {
  const TX* addend = static_cast<const TX*>(addv);
  const TY* dy = static_cast<const TY*>(dyv);
  const TY* y = static_cast<const TY*>(yv);
  const TX* x = static_cast<const TX*>(xv);
  TX* dx = static_cast<TX*>(dxv);
  int rpc;
  const int S = bn_chunks(M, C, &rpc);
  KPF_REQUIRE(ws_floats >= (long)S * 2 * C + 2 * C, "kpf_bn_train_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int cg = (C / 4 + 63) / 64;
  float* coef = ws + (size_t)S * 2 * C;
  if (relu) hipLaunchKernelGGL((bn_partial_kernel<1, true, TX, TY>), dim3(S, cg), dim3(256), 0, st, x, dy, y, mean, ws, M, C, rpc);
  else hipLaunchKernelGGL((bn_partial_kernel<1, false, TX, TY>), dim3(S, cg), dim3(256), 0, st, x, dy, (const TY*)nullptr, mean, ws, M, C, rpc);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64 * BN_FW), 0, st, ws, S, M, C, invstd, dw, db, coef);
  if (relu)
    hipLaunchKernelGGL((bn_elem_kernel<1, true, TX, TY>), dim3(S, cg), dim3(256), 0, st, x, dy, y, mean, invstd, w, (const float*)nullptr, coef, dx, M,
                       C, rpc, addend);
  else
    hipLaunchKernelGGL((bn_elem_kernel<1, false, TX, TY>), dim3(S, cg), dim3(256), 0, st, x, dy, (const TY*)nullptr, mean, invstd, w,
                       (const float*)nullptr, coef, dx, M, C, rpc, addend);
  return kpf_check_launch("kpf_bn_train_backward");
}

// dispatch on (input-side dtype, output-side dtype): fp32 or ONE 16-bit type on either side
#define KPF_BN_DISPATCH(FN, xdt, ydt, ...)                                                                             \
  do {                                                                                                                 \
    const int h = (xdt) != KPF_DT_F32 ? (xdt) : (ydt);                                                                 \
    KPF_REQUIRE(((xdt) == KPF_DT_F32 || (xdt) == h) && ((ydt) == KPF_DT_F32 || (ydt) == h) && h >= 0 && h <= KPF_DT_F16, \
                "kpf_bn_train: unsupported dtype pair (%d, %d)", (int)(xdt), (int)(ydt));                               \
    if ((xdt) == KPF_DT_F32 && (ydt) == KPF_DT_F32) return FN<float, float>(__VA_ARGS__);                             \
    if (h == KPF_DT_BF16) {                                                                                            \
      if ((xdt) == KPF_DT_F32) return FN<float, bf16_t>(__VA_ARGS__);                                                 \
      if ((ydt) == KPF_DT_F32) return FN<bf16_t, float>(__VA_ARGS__);                                                 \
      return FN<bf16_t, bf16_t>(__VA_ARGS__);                                                                         \
    }                                                                                                                  \
    if ((xdt) == KPF_DT_F32) return FN<float, f16_t>(__VA_ARGS__);                                                    \
    if ((ydt) == KPF_DT_F32) return FN<f16_t, float>(__VA_ARGS__);                                                    \
    return FN<f16_t, f16_t>(__VA_ARGS__);                                                                             \
  } while (0)

extern "C" {

int kpf_bn_train_forward(const void* x, int x_dtype, const float* w, const float* b, void* y, int y_dtype, float* mean, float* invstd,
                         float* running_mean, float* running_var, float momentum, float eps, int relu, float* ws, long ws_floats, long M, int C,
                         void* stream) {
  KPF_REQUIRE(x && w && b && y && mean && invstd && ws, "kpf_bn_train_forward: null pointer");
  KPF_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "kpf_bn_train_forward: bad shape (C %% 4 == 0)");
  KPF_REQUIRE(kpf_aligned16(x) && kpf_aligned16(y) && kpf_aligned16(w) && kpf_aligned16(b) && kpf_aligned16(mean) && kpf_aligned16(invstd) &&
                  kpf_aligned16(ws), "kpf_bn_train_forward: pointers must be 16-byte aligned");
  KPF_BN_DISPATCH(bn_forward_impl, x_dtype, y_dtype, x, w, b, y, mean, invstd, running_mean, running_var, momentum, eps, relu, ws, ws_floats, M, C,
                  stream);
}

static int bn_train_backward_any(const void* dy, const void* x, const void* y, int x_dtype, int y_dtype, const float* mean, const float* invstd,
                                 const float* w, void* dx, float* dw, float* db, int relu, float* ws, long ws_floats, long M, int C, void* stream,
                                 const void* addend) {
  KPF_REQUIRE(dy && x && mean && invstd && w && dx && ws && (!relu || y), "kpf_bn_train_backward: null pointer");
  KPF_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "kpf_bn_train_backward: bad shape (C %% 4 == 0)");
  KPF_REQUIRE(kpf_aligned16(dy) && kpf_aligned16(x) && kpf_aligned16(dx) && kpf_aligned16(ws) && (!relu || kpf_aligned16(y)) && kpf_aligned16(addend),
              "kpf_bn_train_backward: pointers must be 16-byte aligned");
  KPF_BN_DISPATCH(bn_backward_impl, x_dtype, y_dtype, dy, x, y, mean, invstd, w, dx, dw, db, relu, ws, ws_floats, M, C, stream, addend);
}

int kpf_bn_train_backward(const void* dy, const void* x, const void* y, int x_dtype, int y_dtype, const float* mean, const float* invstd,
                          const float* w, void* dx, float* dw, float* db, int relu, float* ws, long ws_floats, long M, int C, void* stream) {
  return bn_train_backward_any(dy, x, y, x_dtype, y_dtype, mean, invstd, w, dx, dw, db, relu, ws, ws_floats, M, C, stream, nullptr);
}

int kpf_bn_train_backward_add(const void* dy, const void* x, const void* y, int x_dtype, int y_dtype, const float* mean, const float* invstd,
                              const float* w, const void* addend, void* dx, float* dw, float* db, int relu, float* ws, long ws_floats, long M, int C,
                              void* stream) {
  return bn_train_backward_any(dy, x, y, x_dtype, y_dtype, mean, invstd, w, dx, dw, db, relu, ws, ws_floats, M, C, stream, addend);
}

int kpf_bn_train_forward_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* invstd, float* running_mean,
                             float* running_var, float momentum, float eps, int relu, float* ws, long ws_floats, long M, int C,
                             void* stream) {
  return kpf_bn_train_forward(x, KPF_DT_F32, w, b, y, KPF_DT_F32, mean, invstd, running_mean, running_var, momentum, eps, relu, ws, ws_floats, M, C,
                              stream);
}

int kpf_bn_train_backward_f32(const float* dy, const float* x, const float* y, const float* mean, const float* invstd, const float* w,
                              float* dx, float* dw, float* db, int relu, float* ws, long ws_floats, long M, int C, void* stream) {
  return kpf_bn_train_backward(dy, x, y, KPF_DT_F32, KPF_DT_F32, mean, invstd, w, dx, dw, db, relu, ws, ws_floats, M, C, stream);
}

}  // extern "C"
