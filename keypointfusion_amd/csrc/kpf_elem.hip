// HBM-bound kernels of the backbone path: depthwise 7x7 + LayerNorm (fused), LayerNorm, bilinear x2 upsample,
// layout repacks, max-pool.  All fp32, NHWC, 16-byte vector accesses (float4 over channels), wave64 reductions.
#include "kpf_common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------------------------------
// depthwise 7x7 + bias + LayerNorm(C)
//   grid.x = B*H*ceil(W/(8*S)) ; block = S * (C/4) threads ; thread (strip s, channel quad q) produces 8 consecutive
//   output pixels of one image row for 4 channels with a register sliding window (7 rows x 14 input float4), parks the
//   un-normalised result in an LDS tile [8*S pixels][C], then each wave normalises whole pixels from LDS (shuffle
//   reductions) and writes them out fully coalesced.  Algorithmic traffic: read x once, write y once.
// ---------------------------------------------------------------------------------------------------------------
constexpr int DW_T = 8;

template <typename TA>
__global__ __launch_bounds__(512) void dwconv7_ln_kernel(const TA* __restrict__ x, const float* __restrict__ wdw,
                                                         const float* __restrict__ bdw, const float* __restrict__ lw,
                                                         const float* __restrict__ lb, TA* __restrict__ y, int H, int W,
                                                         int C, int S, int xblocks, float eps) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [8*S][C]
  const int C4 = C >> 2;
  const int tid = threadIdx.x;
  const int s = tid / C4;
  const int q = tid - s * C4;
  int bid = blockIdx.x;
  const int xb = bid % xblocks;
  bid /= xblocks;
  const int oy = bid % H;
  const int b = bid / H;
  const int x0 = (xb * S + s) * DW_T;

  if (s < S && x0 < W) {
    f32x4 acc[DW_T];
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bdw + 4 * q);
#pragma unroll
    for (int t = 0; t < DW_T; ++t) acc[t] = bias;
    const TA* xb_ptr = x + (long)b * H * W * C + 4 * q;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
      const int iy = oy + ky - 3;
      if ((unsigned)iy >= (unsigned)H) continue;
      f32x4 in[DW_T + 6];
      const TA* row = xb_ptr + (long)iy * W * C;
#pragma unroll
      for (int i = 0; i < DW_T + 6; ++i) {
        const int ix = x0 + i - 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ix < (unsigned)W) v = kpf_ld4(row + (long)ix * C);
        in[i] = v;
      }
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wdw + (ky * 7 + kx) * C + 4 * q);
#pragma unroll
        for (int t = 0; t < DW_T; ++t) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][e] = fmaf(in[t + kx][e], wv[e], acc[t][e]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < DW_T; ++t) *reinterpret_cast<f32x4*>(tile + (s * DW_T + t) * C + 4 * q) = acc[t];
  }
  __syncthreads();

  // LayerNorm: one wave per pixel, round robin
  const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int npix = S * DW_T;
  const float invC = 1.0f / (float)C;
  for (int p = wave; p < npix; p += nwaves) {
    const int px = xb * S * DW_T + p;
    if (px >= W) break;
    const float* src = tile + p * C;
    float sum = 0.f;
    for (int i = lane; i < C4; i += 64) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
      sum += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = wave_sum(sum) * invC;
    float sq = 0.f;
    for (int i = lane; i < C4; i += 64) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[e] - mean;
        sq = fmaf(d, d, sq);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * invC + eps);
    TA* dst = y + (((long)b * H + oy) * W + px) * C;
    for (int i = lane; i < C4; i += 64) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
      const f32x4 g = *reinterpret_cast<const f32x4*>(lw + 4 * i);
      const f32x4 be = *reinterpret_cast<const f32x4*>(lb + 4 * i);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * g[e] + be[e];
      kpf_st4(dst + 4 * i, o);
    }
  }
}

// Workgroup ids b, b+8, b+16, .. run on one XCD (private 4 MB L2): give each XCD a contiguous range of logical ids, so that the
// workgroups sharing halo rows share an L2 (un-remapped, the 7x7 halo re-reads went to HBM: 3.9x the algorithmic bytes measured).
__device__ __forceinline__ unsigned xcd_contiguous_block_id() {
  const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---------------------------------------------------------------------------------------------------------------
// depthwise 7x7 + bias (no norm): thread = 4 channels x (2 output rows x 8 px).  Each of the 8 input rows it touches is loaded
// once (14 float4) and feeds both output rows, the 49-tap weights of its channel quad come from LDS (when 49*C*4 B fits) so
// the CU's address unit only carries activations: 7 activation loads per output float4 instead of 12.25 + 6.1 weight loads.
// No LDS output tile -> occupancy is set by registers only.
// ---------------------------------------------------------------------------------------------------------------
// R2 = false (round 6): ONE output row per thread (`ypairs` then counts rows) — twice the threads for the small maps of the training step, whose launches
// were 96 - 384 workgroups of long serial chains on a 256-CU chip.
template <typename TA, bool WLDS, int PX, bool R2 = true>
__global__ __launch_bounds__(256) void dwconv7_kernel(const TA* __restrict__ x, const float* __restrict__ wdw,
                                                      const float* __restrict__ bdw, TA* __restrict__ y, int B, int H, int W, int C,
                                                      int xstrips, int ypairs, const TA* __restrict__ addend = nullptr) {
  // addend (nullable, y's shape): added to the output — the data gradient of a ConvNeXt block picks up the skip path's gradient here
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [49][C] when WLDS
  const int C4 = C >> 2;
  if (WLDS) {
    for (int i = threadIdx.x; i < 49 * C4; i += 256)
      *reinterpret_cast<f32x4*>(wl + 4 * i) = *reinterpret_cast<const f32x4*>(wdw + 4 * i);
    __syncthreads();
  }
  const long total = (long)B * ypairs * xstrips * C4;
  const long gid = (long)xcd_contiguous_block_id() * 256 + threadIdx.x;
  if (gid >= total) return;
  const int q = (int)(gid % C4);
  long r = gid / C4;
  const int xs = (int)(r % xstrips);
  r /= xstrips;
  const int yp = (int)(r % ypairs);
  const int b = (int)(r / ypairs);
  const int x0 = xs * PX, y0 = yp * (R2 ? 2 : 1);
  const float* wsrc = WLDS ? wl : wdw;
  f32x4 acc0[PX], acc1[PX];
  const f32x4 bias = *reinterpret_cast<const f32x4*>(bdw + 4 * q);
#pragma unroll
  for (int t = 0; t < PX; ++t) {
    acc0[t] = bias;
    acc1[t] = bias;
  }
  const TA* xb = x + (long)b * H * W * C + 4 * q;
#pragma unroll 1
  for (int ir = 0; ir < (R2 ? 8 : 7); ++ir) {  // input rows y0-3 .. y0+4 (y0+3 for a single output row)
    const int iy = y0 + ir - 3;
    if ((unsigned)iy >= (unsigned)H) continue;
    f32x4 in[PX + 6];
    const TA* row = xb + (long)iy * W * C;
#pragma unroll
    for (int i = 0; i < PX + 6; ++i) {
      const int ix = x0 + i - 3;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)ix < (unsigned)W) v = kpf_ld4(row + (long)ix * C);
      in[i] = v;
    }
    if (ir < 7) {  // output row y0: tap row ky = ir
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wsrc + (ir * 7 + kx) * C + 4 * q);
#pragma unroll
        for (int t = 0; t < PX; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc0[t][e] = fmaf(in[t + kx][e], wv[e], acc0[t][e]);
      }
    }
    if (R2 && ir >= 1) {  // output row y0+1: tap row ky = ir-1
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wsrc + ((ir - 1) * 7 + kx) * C + 4 * q);
#pragma unroll
        for (int t = 0; t < PX; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc1[t][e] = fmaf(in[t + kx][e], wv[e], acc1[t][e]);
      }
    }
  }
  TA* yb = y + (long)b * H * W * C + 4 * q;
  const TA* ab = addend ? addend + (long)b * H * W * C + 4 * q : nullptr;
#pragma unroll
  for (int t = 0; t < PX; ++t) {
    if (x0 + t < W) {
      const long o0 = ((long)y0 * W + x0 + t) * C, o1 = ((long)(y0 + 1) * W + x0 + t) * C;
      kpf_st4(yb + o0, ab ? acc0[t] + kpf_ld4(ab + o0) : acc0[t]);
      if (R2 && y0 + 1 < H) kpf_st4(yb + o1, ab ? acc1[t] + kpf_ld4(ab + o1) : acc1[t]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// depthwise 7x7 + bias + LayerNorm fused, for 256 < C <= 1024 (ConvNeXt stages 3-4): one workgroup of NT = C/4 rounded up to whole
// waves owns a 2-row x 8-pixel strip, thread q computes channel quad q exactly like dwconv7_kernel (taps from global: 49*C*4 B does
// not fit beside a useful occupancy; they stay L2 / L1 resident), and the per-pixel LayerNorm statistics cross the 2-4 waves
// through LDS: every thread writes its 16 per-pixel partial sums, 8 threads per pixel add NT/8 of them each and finish with three
// xor-shuffles (mean first, then the centred squares: two-pass variance like ATen's).  One pass over the activation instead of
// the conv + LayerNorm pair's two.
// ---------------------------------------------------------------------------------------------------------------
// ROWS = 2 (default): a 2-row x 8-pixel strip per workgroup.  ROWS = 1 (round 5): one row — for launches that leave the chip mostly idle (B x 8 x 8 x 384 at
// B = 32: 128 workgroups of two waves, each thread a chain of eight input-row round trips and 3136 multiply-adds): twice the workgroups, seven rows and half the
// arithmetic per thread.  Same order of operations per output and per LayerNorm sum: same bits.
template <typename TA, int NT, bool SPLIT, int ROWS = 2>
__global__ __launch_bounds__(NT) void dwconv7_ln_wide_kernel(const TA* __restrict__ x, const float* __restrict__ wdw,
                                                             const float* __restrict__ bdw, const float* __restrict__ lw,
                                                             const float* __restrict__ lb, TA* __restrict__ y, int B, int H, int W,
                                                             int C, int xstrips, int ypairs, float eps) {
  constexpr int NP = 8 * ROWS;  // pixels of the strip
  __shared__ float red[NP * NT];
  __shared__ float stat[NP];
  const int C4 = C >> 2;
  const int q = threadIdx.x;
  const bool live = q < C4;
  const int qc = live ? q : 0;
  long r = xcd_contiguous_block_id();
  const int xs = (int)(r % xstrips);
  r /= xstrips;
  const int yp = (int)(r % ypairs);
  const int b = (int)(r / ypairs);
  const int x0 = xs * 8, y0 = yp * ROWS;
  f32x4 acc0[8], acc1[8];
  const f32x4 bias = *reinterpret_cast<const f32x4*>(bdw + 4 * qc);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    acc0[t] = bias;
    acc1[t] = bias;
  }
  const TA* xb = x + (long)b * H * W * C + 4 * qc;
  if (live) {
#pragma unroll 1
    for (int ir = 0; ir < 6 + ROWS; ++ir) {  // input rows y0-3 .. y0+2+ROWS
      const int iy = y0 + ir - 3;
      if ((unsigned)iy >= (unsigned)H) continue;
      f32x4 in[14];
      const TA* row = xb + (long)iy * W * C;
#pragma unroll
      for (int i = 0; i < 14; ++i) {
        const int ix = x0 + i - 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ix < (unsigned)W) v = kpf_ld4(row + (long)ix * C);
        in[i] = v;
      }
      if (ir < 7) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wdw + (ir * 7 + kx) * C + 4 * qc);
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc0[t][e] = fmaf(in[t + kx][e], wv[e], acc0[t][e]);
        }
      }
      if (ROWS == 2 && ir >= 1) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wdw + ((ir - 1) * 7 + kx) * C + 4 * qc);
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc1[t][e] = fmaf(in[t + kx][e], wv[e], acc1[t][e]);
        }
      }
    }
  }
  // ---- LayerNorm statistics of the strip's 16 pixels across the workgroup ----
  const float invC = 1.0f / (float)C;
  const int rp = threadIdx.x >> 3, rg = threadIdx.x & 7;  // reducer role (threads 0 .. 8 NP - 1): pixel rp, one of its 8 adders
  float mean[NP];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const f32x4 v = p < 8 ? acc0[p] : acc1[p & 7];
      float s = 0.f;
      if (live) {
        if (pass == 0) {
          s = (v[0] + v[1]) + (v[2] + v[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = v[e] - mean[p];
            s = fmaf(d, d, s);
          }
        }
      }
      red[p * NT + threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x < 8 * NP) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NT / 8; ++j) s += red[rp * NT + rg + 8 * j];
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 1, 64);
      if (rg == 0) stat[rp] = pass == 0 ? s * invC : 1.0f / sqrtf(s * invC + eps);
    }
    __syncthreads();
    if (pass == 0) {
#pragma unroll
      for (int p = 0; p < NP; ++p) mean[p] = stat[p];
      __syncthreads();  // stat is rewritten by the second pass
    }
  }
  if (!live) return;
  const f32x4 g = *reinterpret_cast<const f32x4*>(lw + 4 * q);
  const f32x4 be = *reinterpret_cast<const f32x4*>(lb + 4 * q);
  TA* yb = y + (long)b * H * W * C;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int t = p & 7, rr = p >> 3;
    if (x0 + t < W && y0 + rr < H) {
      const f32x4 v = rr ? acc1[t] : acc0[t];
      const float rstd = stat[p];
      f32x4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = (v[e] - mean[p]) * rstd * g[e] + be[e];
      TA* dst = yb + ((long)(y0 + rr) * W + x0 + t) * C;
      if constexpr (SPLIT)
        kpf_store_split4(dst, 4 * q, o4);
      else
        kpf_st4(dst + 4 * q, o4);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// depthwise 7x7 + bias + LayerNorm fused, for C <= 256: a pixel's C/4 channel quads sit on the lanes of one LG-lane group
// (LG = 32 or 64), every lane computes 4 channels x (2 rows x 8 px) exactly like dwconv7_kernel, and the per-pixel mean /
// variance are xor-shuffle reductions inside the group — no LDS tile (which capped the occupancy of the first fused kernel),
// and the activation is read once and written once (the unfused pair reads and writes it twice).
// ---------------------------------------------------------------------------------------------------------------
template <typename TA, int LG, bool SPLIT>
__global__ __launch_bounds__(256) void dwconv7_ln_wave_kernel(const TA* __restrict__ x, const float* __restrict__ wdw,
                                                              const float* __restrict__ bdw, const float* __restrict__ lw,
                                                              const float* __restrict__ lb, TA* __restrict__ y, int B, int H, int W,
                                                              int C, int xstrips, int ypairs, float eps) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [49][C]
  const int C4 = C >> 2;
  for (int i = threadIdx.x; i < 49 * C4; i += 256)
    *reinterpret_cast<f32x4*>(wl + 4 * i) = *reinterpret_cast<const f32x4*>(wdw + 4 * i);
  __syncthreads();
  const long groups = (long)B * ypairs * xstrips;
  const long gid = ((long)xcd_contiguous_block_id() * 256 + threadIdx.x) / LG;
  const int q = threadIdx.x % LG;
  const bool live = q < C4 && gid < groups;  // idle lanes still take part in the shuffles (with zeros)
  const long gc = gid < groups ? gid : groups - 1;
  const int xs = (int)(gc % xstrips);
  long r = gc / xstrips;
  const int yp = (int)(r % ypairs);
  const int b = (int)(r / ypairs);
  const int x0 = xs * 8, y0 = yp * 2;
  const int qc = q < C4 ? q : 0;
  f32x4 acc0[8], acc1[8];
  const f32x4 bias = *reinterpret_cast<const f32x4*>(bdw + 4 * qc);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    acc0[t] = bias;
    acc1[t] = bias;
  }
  const TA* xb = x + (long)b * H * W * C + 4 * qc;
  if (live) {
#pragma unroll 1
    for (int ir = 0; ir < 8; ++ir) {  // input rows y0-3 .. y0+4
      const int iy = y0 + ir - 3;
      if ((unsigned)iy >= (unsigned)H) continue;
      f32x4 in[14];
      const TA* row = xb + (long)iy * W * C;
#pragma unroll
      for (int i = 0; i < 14; ++i) {
        const int ix = x0 + i - 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ix < (unsigned)W) v = kpf_ld4(row + (long)ix * C);
        in[i] = v;
      }
      if (ir < 7) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + (ir * 7 + kx) * C + 4 * qc);
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc0[t][e] = fmaf(in[t + kx][e], wv[e], acc0[t][e]);
        }
      }
      if (ir >= 1) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + ((ir - 1) * 7 + kx) * C + 4 * qc);
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc1[t][e] = fmaf(in[t + kx][e], wv[e], acc1[t][e]);
        }
      }
    }
  }
  // LayerNorm of the 16 pixels this lane group holds (two-pass variance like ATen's)
  const float invC = 1.0f / (float)C;
  const f32x4 g = *reinterpret_cast<const f32x4*>(lw + 4 * qc);
  const f32x4 be = *reinterpret_cast<const f32x4*>(lb + 4 * qc);
  TA* yb = y + (long)b * H * W * C + 4 * qc;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x4 v = rr ? acc1[t] : acc0[t];
      float s = live ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
#pragma unroll
      for (int o = LG / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mean = s * invC;
      float sq = 0.f;
      if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[e] - mean;
          sq = fmaf(d, d, sq);
        }
      }
#pragma unroll
      for (int o = LG / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
      const float rstd = 1.0f / sqrtf(sq * invC + eps);
      if (live && x0 + t < W && y0 + rr < H) {
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (v[e] - mean) * rstd * g[e] + be[e];
        if constexpr (SPLIT)
          kpf_store_split4(yb - 4 * qc + ((long)(y0 + rr) * W + x0 + t) * C, 4 * qc, o4);
        else
          kpf_st4(yb + ((long)(y0 + rr) * W + x0 + t) * C, o4);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// depthwise 7x7 + bias + LayerNorm, LDS-tiled (C = 96 / 128, large maps).  The register-tiled kernels above fetch every input pixel
// ~7 times through L2 -> L1; here a workgroup stages the (8+6) x (32+6) pixel halo of its 8 x 32 output tile once per 32-channel
// chunk in LDS and all 49 taps read LDS.  Thread = (channel quad of the chunk, 1 x 8 pixel strip); the outputs of all NCHK chunks
// stay in registers, so the LayerNorm over the pixel's C channels is 8-lane shuffles at the end: one pass over the activation.
// LDS rows are padded by one pixel (39 x 128 B): the two strips of a 16-lane group are vertical neighbours, a row apart = 128 B
// apart modulo the 256-byte bank row, so the ds_read_b128 of 8 quads x 2 strips is conflict-free.
// ---------------------------------------------------------------------------------------------------------------
constexpr int DT_TY = 8, DT_HR = DT_TY + 6;  // tile rows, halo rows

// PXS = strip width (8: C <= 128, 4: C = 192 so that NCHK x PXS accumulators fit); the tile is 8 rows x 4 strips.
template <typename TA, int NCHK, int PXS, bool SPLIT>
__global__ __launch_bounds__(256) void dwconv7_ln_tile_kernel(const TA* __restrict__ x, const float* __restrict__ wdw,
                                                              const float* __restrict__ bdw, const float* __restrict__ lw,
                                                              const float* __restrict__ lb, TA* __restrict__ y, int B, int H, int W,
                                                              int tiles_x, int tiles_y, float eps) {
  constexpr int C = 32 * NCHK;
  constexpr int DT_TX = 4 * PXS, DT_HP = DT_TX + 6, DT_RS = (DT_HP + 1) * 32;  // halo pixels per row; row stride in floats (1 pad pixel)
  static_assert(DT_HP % 2 == 0, "the padded row must be an odd number of 128-byte pixels");
  __shared__ __attribute__((aligned(16))) float tile[DT_HR * DT_RS];  // [14][39][32]
  __shared__ __attribute__((aligned(16))) float wl[49 * 32];          // this chunk's taps [49][32]
  const int tid = threadIdx.x;
  const int q8 = tid & 7, strip = tid >> 3;
  const int srow = strip & 7, sx = (strip >> 3) * PXS;  // consecutive strips are vertical neighbours (bank layout, see above)
  unsigned bid = xcd_contiguous_block_id();
  const int tx = (int)(bid % tiles_x);
  bid /= tiles_x;
  const int ty = (int)(bid % tiles_y);
  const int b = (int)(bid / tiles_y);
  const int y0 = ty * DT_TY, x0 = tx * DT_TX;
  const TA* xb = x + (long)b * H * W * C;

  f32x4 acc[NCHK][PXS];
#pragma unroll  // (acc is indexed by ch: it must be a compile-time index to stay in registers)
  for (int ch = 0; ch < NCHK; ++ch) {
    // stage the halo tile of this chunk: 14 x 38 pixels x 8 quads, zero outside the image.  All of a thread's loads are issued
    // before the first LDS write (unconditional loads from a clamped address, zeroed by a select: a branch around a load would
    // make the compiler wait for each one), so a chunk exposes one memory round trip, not seventeen.
    constexpr int NIT = (DT_HR * DT_HP * 8 + 255) / 256;
    f32x4 sv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int q = i & 7, p = i >> 3;
      const int pr = p / DT_HP, pc = p - pr * DT_HP;
      const int iy = y0 + pr - 3, ix = x0 + pc - 3;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && i < DT_HR * DT_HP * 8;
      const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
      const f32x4 v = kpf_ld4(xb + ((long)cy * W + cx) * C + ch * 32 + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) sv[it][e] = ok ? v[e] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int q = i & 7, p = i >> 3;
      const int pr = p / DT_HP, pc = p - pr * DT_HP;
      if (i < DT_HR * DT_HP * 8) *reinterpret_cast<f32x4*>(tile + pr * DT_RS + pc * 32 + q * 4) = sv[it];
    }
    for (int i = tid; i < 49 * 8; i += 256) {
      const int q = i & 7, t = i >> 3;
      *reinterpret_cast<f32x4*>(wl + t * 32 + q * 4) = *reinterpret_cast<const f32x4*>(wdw + (long)t * C + ch * 32 + q * 4);
    }
    __syncthreads();
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bdw + ch * 32 + q8 * 4);
    f32x4 a8[PXS];
#pragma unroll
    for (int t = 0; t < PXS; ++t) a8[t] = bias;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
      const float* row = tile + (srow + ky) * DT_RS + sx * 32 + q8 * 4;
      f32x4 in[PXS + 6];
#pragma unroll
      for (int i = 0; i < PXS + 6; ++i) in[i] = *reinterpret_cast<const f32x4*>(row + i * 32);
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + (ky * 7 + kx) * 32 + q8 * 4);
#pragma unroll
        for (int t = 0; t < PXS; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) a8[t][e] = fmaf(in[t + kx][e], wv[e], a8[t][e]);
      }
    }
#pragma unroll
    for (int t = 0; t < PXS; ++t) acc[ch][t] = a8[t];
    __syncthreads();  // everyone is done with the tile before the next chunk overwrites it
  }

  // LayerNorm of this strip's pixels: the pixel's C channels are the NCHK x 4 values of 8 neighbouring lanes
  const float invC = 1.0f / (float)C;
  const int oy = y0 + srow;
#pragma unroll
  for (int t = 0; t < PXS; ++t) {
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCHK; ++ch) s += (acc[ch][t][0] + acc[ch][t][1]) + (acc[ch][t][2] + acc[ch][t][3]);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    const float mean = s * invC;
    float sq = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCHK; ++ch)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = acc[ch][t][e] - mean;
        sq = fmaf(d, d, sq);
      }
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    sq += __shfl_xor(sq, 4, 64);
    const float rstd = 1.0f / sqrtf(sq * invC + eps);
    const int ox = x0 + sx + t;
    if (oy < H && ox < W) {
      TA* dst = y + (((long)b * H + oy) * W + ox) * C;
#pragma unroll
      for (int ch = 0; ch < NCHK; ++ch) {
        const int c = ch * 32 + q8 * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(lw + c);
        const f32x4 be = *reinterpret_cast<const f32x4*>(lb + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (acc[ch][t][e] - mean) * rstd * g[e] + be[e];
        if constexpr (SPLIT)
          kpf_store_split4(dst, c, o);
        else
          kpf_st4(dst + c, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over C for `rows` pixels.  A row is handled by LPR = 16/32/64 lanes (the power of two covering C/4 float4s, so a
// 96-channel row uses a 32-lane half wave instead of idling 40 of 64 lanes), rows stay in registers (two-pass variance),
// reductions are xor-shuffles inside the LPR-lane group.  C <= 64*4*LN_MAXV.
// ---------------------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 8;  // float4 per lane at LPR = 64 -> C <= 2048

template <typename TI, typename TO, int LPR, int NV, bool SPLIT>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, TO* __restrict__ y, long rows, int C,
                                                        float eps) {
  constexpr int RPB = 256 / LPR;  // rows per block
  const int sub = threadIdx.x % LPR;
  const long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
  const bool live = row < rows;
  const int C4 = C >> 2;
  const TI* src = x + (live ? row : 0) * C;
  f32x4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = sub + LPR * i;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < C4) {
      v[i] = kpf_ld4(src + 4 * c);
      sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float invC = 1.0f / (float)C;
  const float mean = sum * invC;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (sub + LPR * i < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        sq = fmaf(d, d, sq);
      }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / sqrtf(sq * invC + eps);
  if (!live) return;
  TO* dst = y + row * C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = sub + LPR * i;
    if (c < C4) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(w + 4 * c);
      const f32x4 be = *reinterpret_cast<const f32x4*>(b + 4 * c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + be[e];
      if constexpr (SPLIT)  // in place is safe: a row's lanes share a wave and every load precedes the reduction shuffles
        kpf_store_split4(dst, 4 * c, o);
      else
        kpf_st4(dst + 4 * c, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bilinear x2, align_corners = False:  src = (dst + 0.5) / 2 - 0.5 clamped at 0 ; i1 = min(i0 + 1, n - 1)
// ---------------------------------------------------------------------------------------------------------------
// NQ channel quads per thread: 1 (16 bytes of fp32 / 8 bytes of 16-bit storage) or 2 for 16-bit storage (16-byte accesses: the 8-byte form moved the
// B x 128 x 128 x 128 f16 map of ConvNeXt-B's decoder at 1.8 TB/s)
template <typename TA, int NQ = 1>
__global__ __launch_bounds__(256) void upsample2x_kernel(const TA* __restrict__ src, TA* __restrict__ dst, int B, int H,
                                                         int W, int C4, int dst_ld, int dst_coff) {
  const int CQ = C4 / NQ;
  const long total = (long)B * 2 * H * 2 * W * CQ;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % CQ) * NQ;
    long p = i / CQ;
    const int ox = (int)(p % (2 * W));
    p /= (2 * W);
    const int oy = (int)(p % (2 * H));
    const int b = (int)(p / (2 * H));
    float fy = (oy + 0.5f) * 0.5f - 0.5f;
    float fx = (ox + 0.5f) * 0.5f - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const TA* sb = src + (long)b * H * W * C4 * 4 + 4 * q;
    const TA* p00 = sb + ((long)y0 * W + x0) * C4 * 4;
    const TA* p01 = sb + ((long)y0 * W + x1) * C4 * 4;
    const TA* p10 = sb + ((long)y1 * W + x0) * C4 * 4;
    const TA* p11 = sb + ((long)y1 * W + x1) * C4 * 4;
    TA* po = dst + (((long)b * 2 * H + oy) * 2 * W + ox) * dst_ld + dst_coff + 4 * q;
    f32x4 v00[NQ], v01[NQ], v10[NQ], v11[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      v00[u] = kpf_ld4(p00 + 4 * u);
      v01[u] = kpf_ld4(p01 + 4 * u);
      v10[u] = kpf_ld4(p10 + 4 * u);
      v11[u] = kpf_ld4(p11 + 4 * u);
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = hy * (hx * v00[u][e] + lx * v01[u][e]) + ly * (hx * v10[u][e] + lx * v11[u][e]);
      kpf_st4(po + 4 * u, o);
    }
  }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C,
                                                           int HW, int Cpad) {
  const long total = (long)B * HW * Cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const long p = i / Cpad;
    const int pix = (int)(p % HW);
    const int b = (int)(p / HW);
    dst[i] = c < C ? src[((long)b * C + c) * HW + pix] : 0.f;
  }
}

// tile transpose through LDS: NHWC slice -> NCHW
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW,
                                                           int src_ld, int src_coff) {
  __shared__ float t[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    t[r][tx] = (p < HW && c < C) ? src[((long)b * HW + p) * src_ld + src_coff + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    if (c < C && p < HW) dst[((long)b * C + c) * HW + p] = t[tx][r];
  }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H,
                                                           int W, int OH, int OW, int C4) {
  const long total = (long)B * OH * OW * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % C4);
    long p = i / C4;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 + ky - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 + kx - 1;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (((long)b * H + iy) * W + ix) * C4 * 4 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    *reinterpret_cast<f32x4*>(dst + i * 4) = m;
  }
}

template <typename TA>
__global__ __launch_bounds__(256) void cast_h16_f32_kernel(const TA* __restrict__ src, float* __restrict__ dst, long total, int C4, int src_ld,
                                                           int src_coff) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C4;
    const int q = (int)(i - row * C4);
    kpf_st4(dst + i * 4, kpf_ld4(src + row * src_ld + src_coff + 4 * q));
  }
}

template <typename TA>
__global__ __launch_bounds__(256) void cast_f32_h16_kernel(const float* __restrict__ src, TA* __restrict__ dst, long total4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) kpf_st4(dst + i * 4, kpf_ld4(src + i * 4));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4: depthwise 7x7 as a pure LDS-tiled stencil that also emits LayerNorm statistics (16-bit storage), and the LayerNorm as a light second
// pass over them (kpf_dwconv7_stats_h16 / kpf_ln_apply_stats_h16; the statistics can alternatively be consumed by the following GEMM).
//
// Why: the one-pass kernels above hold a pixel's whole channel vector (x 16 pixels) in one workgroup's registers — 182 VGPRs, two waves per SIMD,
// every input row fetched through the texture path by 14 dependent 8-byte loads — and run 5.5 x over the vector-ALU floor of the stencil
// (49 FMA per output element: 25 us for 64 x 32 x 32 x 512 at the chip's full fp32 rate; measured 138 us in f16).  Here:
//  * a workgroup owns a 16 x 16-pixel tile of ONE 64-channel chunk: the (16+6) x (16+6) halo tile (62 KB) is filled once by LDS-DMA (zero page outside
//    the image), so every input element crosses L2 -> LDS 1.9 x and is then read from LDS at 256 B/clk (ds_read_b64 of a channel quad);
//  * a thread owns one channel quad of one 8-pixel strip of one row: 32 accumulators, 14 packed input quads and 7 weight quads per filter row —
//    under 100 registers, so four to five waves per SIMD cover each other's LDS latencies; f16 inputs feed v_fma_mix_f32 straight from the
//    packed registers (no conversion instructions, fp32 accumulation);
//  * the 16 lanes that hold a pixel's 64 channels of the chunk reduce (mean, centred sum of squares) by DPP shuffles; chunk statistics go to a
//    side buffer [pixels][C/64][2] (6 % of the tensor) and are merged with Chan's update where they are consumed — no cancellation for |mean| >> sigma.
//  LDS rows are padded to 23 pixels so that the two strips a 32-lane half reads (same columns, neighbouring rows) sit in opposite halves of the
//  256-byte bank row.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int DS_T = 16, DS_HW = DS_T + 6, DS_ROW = DS_HW + 1, DS_CH = 64;  // tile edge, halo edge, padded LDS row (pixels), channels per chunk
typedef __attribute__((address_space(3))) void ds_lds_void_t;
typedef __attribute__((address_space(1))) void ds_gbl_void_t;
__device__ __attribute__((aligned(16))) float kpf_elem_zero16[4] = {0.f, 0.f, 0.f, 0.f};

template <typename TA>
__global__ __launch_bounds__(512, 2) void dwconv7_stats_kernel(const TA* __restrict__ x, const float* __restrict__ wdw, const float* __restrict__ bdw,
                                                               TA* __restrict__ y, float* __restrict__ stats, int H, int W, int C, int tiles_x,
                                                               int tiles_y, const float* __restrict__ zero, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float dsl[];
  char* const LB = reinterpret_cast<char*>(dsl);
  constexpr int SLOTS = DS_HW * DS_ROW * 8;                // 16-byte pieces of the halo tile (128 bytes per pixel)
  constexpr int HALO_B = (SLOTS + 63) / 64 * 1024;         // rounded up to whole 1-KiB DMA instructions: the last one writes past the tile
  float* const Ws = reinterpret_cast<float*>(LB + HALO_B);  // [49][64] taps of the chunk, then [64] bias
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunk = C / DS_CH;
  long r = xcd_contiguous_block_id();
  const int chunk = (int)(r % nchunk);  // chunks of one tile run back to back on one XCD: the stats rows of a pixel are written together
  r /= nchunk;
  const int tx = (int)(r % tiles_x);
  r /= tiles_x;
  const int ty = (int)(r % tiles_y);
  const int b = (int)(r / tiles_y);
  const int x0 = tx * DS_T, y0 = ty * DS_T, c0 = chunk * DS_CH;

  // ---- halo tile by LDS-DMA: slot = (pixel of the padded tile, 16-byte piece); LDS is linear in slots, the source is per lane ----
  const TA* const xb = x + (long)b * H * W * C + c0;
#pragma unroll 1
  for (int s0 = (dbg & 2) ? SLOTS : 0; s0 < SLOTS; s0 += 512) {
    const int slot = s0 + tid;
    const int px = slot >> 3, pc = slot & 7;
    const int hy = px / DS_ROW, hx = px - hy * DS_ROW;
    const int iy = y0 + hy - 3, ix = x0 + hx - 3;
    const bool in = slot < SLOTS && hx < DS_HW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const void* src = in ? static_cast<const void*>(xb + ((long)iy * W + ix) * C + pc * 8) : static_cast<const void*>(zero);
    if (s0 + wave * 64 < SLOTS)  // (wave-uniform; the last pass is partial)
      __builtin_amdgcn_global_load_lds((ds_gbl_void_t*)src, (ds_lds_void_t*)(LB + (s0 + wave * 64) * 16), 16, 0, 0);
  }
  for (int i = tid; i < 49 * 16; i += 512) {
    const int tap = i >> 4, q4 = i & 15;
    *reinterpret_cast<f32x4*>(Ws + tap * 64 + 4 * q4) = *reinterpret_cast<const f32x4*>(wdw + (long)tap * C + c0 + 4 * q4);
  }
  if (tid < 16) *reinterpret_cast<f32x4*>(Ws + 49 * 64 + 4 * tid) = *reinterpret_cast<const f32x4*>(bdw + c0 + 4 * tid);
  __syncthreads();  // (its fence drains the DMA: vmcnt(0))

  // ---- thread = channel quad q of strip (sx, row): lanes 0-15 / 16-31 of a half-wave are rows 2k / 2k+1 of the same strip column ----
  const int q = tid & 15, sid = tid >> 4;
  const int row = ((sid >> 2) << 1) | (sid & 1), sx = (sid >> 1) & 1;
  f32x4 acc[8];
  {
    const f32x4 bias = *reinterpret_cast<const f32x4*>(Ws + 49 * 64 + 4 * q);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = bias;
  }
  const char* const ib = LB + ((row * DS_ROW + sx * 8) * DS_CH + 4 * q) * 2;
  if (!(dbg & 1))
#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
    uint2 in[14];  // four packed 16-bit channels of 14 input pixels
#pragma unroll
    for (int i = 0; i < 14; ++i) in[i] = *reinterpret_cast<const uint2*>(ib + (ky * DS_ROW + i) * (DS_CH * 2));
    // widen the 14 x 4 inputs ONCE per filter row (56 conversions feed 224 multiply-adds), then plain v_fma_f32.  Round 4 used v_fma_mix_f32 on the packed
    // halves to save the conversions; tools/valu_rate.hip (round 5) measures that form at 2.4 ns per wave-instruction and SIMD against 1.4 for v_fma_f32 and
    // 1.5 for a 4 : 16 mix of v_cvt_f32_f16 and v_fma_f32 — the kernel sat at 85 % of the v_fma_mix roof (61 us for 1.64 G multiply-adds).  Same values
    // (the conversion is exact either way): same bits.  bf16: shift / mask, as before.
    f32x4 inf[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      if constexpr (std::is_same<TA, f16_t>::value) {
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        const h2_t lo = __builtin_bit_cast(h2_t, in[i].x), hi = __builtin_bit_cast(h2_t, in[i].y);
        inf[i] = f32x4{(float)lo[0], (float)lo[1], (float)hi[0], (float)hi[1]};
      } else {
        inf[i] = f32x4{__uint_as_float(in[i].x << 16), __uint_as_float(in[i].x & 0xffff0000u), __uint_as_float(in[i].y << 16), __uint_as_float(in[i].y & 0xffff0000u)};
      }
    }
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(Ws + (ky * 7 + kx) * 64 + 4 * q);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        // (written as instructions: left to itself hipcc SLP-packs the products into v_pk_fma_f32 with register shuffles around them)
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[t][0]) : "v"(inf[t + kx][0]), "v"(wv[0]));
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[t][1]) : "v"(inf[t + kx][1]), "v"(wv[1]));
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[t][2]) : "v"(inf[t + kx][2]), "v"(wv[2]));
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[t][3]) : "v"(inf[t + kx][3]), "v"(wv[3]));
      }
    }
  }

  // ---- outputs + chunk statistics ----
  // The 16 lanes q = 0..15 of a strip hold the 64 channels of each of its 8 pixels.  Sum and sum of squares are reduced by a TRANSPOSED butterfly:
  // at distance 8 a lane hands over four pixels' partials and keeps four, at distance 4 two, at distance 2 one, and a last exchange at distance 1
  // completes them — 2 x (4 + 2 + 1 + 1) = 16 cross-lane operations per thread instead of 64 (a 16-us item of a 75-us kernel when every pixel was
  // reduced on its own).  Pixel t ends up complete in lanes q with (q >> 1) == bit-reversed-ish index below; lane pairs hold the same total.
  // M2 = S2 - S1^2 / 64 in fp32: the cancellation error is 2^-24 (mean / sigma)^2 of M2, harmless up to |mean| ~ 100 sigma inside a 64-channel chunk.
  const int oy = y0 + row;
  TA* const yb = y + ((long)b * H + oy) * W * C + c0 + 4 * q;
  float s1[8], s2[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    s1[t] = (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    s2[t] = fmaf(acc[t][0], acc[t][0], acc[t][1] * acc[t][1]) + fmaf(acc[t][2], acc[t][2], acc[t][3] * acc[t][3]);
    if (oy < H && x0 + sx * 8 + t < W && !(dbg & 8)) kpf_st4(yb + (long)(x0 + sx * 8 + t) * C, acc[t]);
  }
  if (!(dbg & 4)) {
    // distance 8: lanes with (q & 8) == 0 keep pixels 0-3, the others pixels 4-7
    float a1[4], a2[4];
    {
      const bool up = q & 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float give1 = up ? s1[i] : s1[4 + i], give2 = up ? s2[i] : s2[4 + i];
        const float keep1 = up ? s1[4 + i] : s1[i], keep2 = up ? s2[4 + i] : s2[i];
        a1[i] = keep1 + __shfl_xor(give1, 8, 64);
        a2[i] = keep2 + __shfl_xor(give2, 8, 64);
      }
    }
    float b1[2], b2[2];
    {
      const bool up = q & 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float give1 = up ? a1[i] : a1[2 + i], give2 = up ? a2[i] : a2[2 + i];
        const float keep1 = up ? a1[2 + i] : a1[i], keep2 = up ? a2[2 + i] : a2[i];
        b1[i] = keep1 + __shfl_xor(give1, 4, 64);
        b2[i] = keep2 + __shfl_xor(give2, 4, 64);
      }
    }
    float c1, c2;
    {
      const bool up = q & 2;
      const float give1 = up ? b1[0] : b1[1], give2 = up ? b2[0] : b2[1];
      c1 = (up ? b1[1] : b1[0]) + __shfl_xor(give1, 2, 64);
      c2 = (up ? b2[1] : b2[0]) + __shfl_xor(give2, 2, 64);
    }
    c1 += __shfl_xor(c1, 1, 64);
    c2 += __shfl_xor(c2, 1, 64);
    // this lane pair owns pixel t = 4 * bit3(q) + 2 * bit2(q) + bit1(q)
    const int t = ((q >> 3) & 1) * 4 + ((q >> 2) & 1) * 2 + ((q >> 1) & 1);
    const int ox = x0 + sx * 8 + t;
    if (!(q & 1) && oy < H && ox < W && !(dbg & 16)) {
      const float mean = c1 * (1.0f / DS_CH);
      const float m2 = fmaxf(c2 - c1 * mean, 0.f);
      *reinterpret_cast<float2*>(stats + (((long)b * H + oy) * W + ox) * (2 * nchunk) + 2 * chunk) = make_float2(mean, m2);
    }
  }
}

// LayerNorm from the chunk statistics, in place on the stencil's output.  A workgroup takes 64 pixels: 64 threads merge the C/64 (mean, M2) pairs of
// one pixel each in a fixed order (equal counts: mean = average of the chunk means, M2 = sum M2_c + 64 sum (mean_c - mean)^2) into LDS, then all
// threads stream the 64 x C block once (8 channels = 16 bytes per lane).
template <typename TA>
__global__ __launch_bounds__(256) void ln_apply_stats_kernel(TA* __restrict__ y, const float* __restrict__ stats, const float* __restrict__ lw,
                                                             const float* __restrict__ lb, long rows, int C, float eps) {
  __shared__ float2 mr[64];
  const int nchunk = C / DS_CH, C8 = C >> 3;
  const long p0 = (long)blockIdx.x * 64;
  if (threadIdx.x < 64 && p0 + threadIdx.x < rows) {
    const float* sp = stats + (p0 + threadIdx.x) * (2 * nchunk);
    float mean = 0.f;
    for (int c = 0; c < nchunk; ++c) mean += sp[2 * c];
    mean *= 1.0f / (float)nchunk;
    float m2 = 0.f;
    for (int c = 0; c < nchunk; ++c) {
      const float d = sp[2 * c] - mean;
      m2 += sp[2 * c + 1] + (float)DS_CH * d * d;
    }
    mr[threadIdx.x] = make_float2(mean, 1.0f / sqrtf(m2 / (float)C + eps));
  }
  __syncthreads();
  const int n = (int)((rows - p0) < 64 ? (rows - p0) : 64) * C8;
  TA* const yb = y + p0 * C;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int px = i / C8, q = i - px * C8;
    const float2 s = mr[px];
    const f32x4 v0 = kpf_ld4(yb + (long)i * 8), v1 = kpf_ld4(yb + (long)i * 8 + 4);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(lw + 8 * q), g1 = *reinterpret_cast<const f32x4*>(lw + 8 * q + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(lb + 8 * q), b1 = *reinterpret_cast<const f32x4*>(lb + 8 * q + 4);
    f32x4 o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o0[e] = (v0[e] - s.x) * s.y * g0[e] + b0[e];
      o1[e] = (v1[e] - s.x) * s.y * g1[e] + b1[e];
    }
    kpf_st4(yb + (long)i * 8, o0);
    kpf_st4(yb + (long)i * 8 + 4, o1);
  }
}

// (mean, rstd) of every pixel from its chunk statistics, in the order ln_apply_stats_kernel merges them (same bits): the form the GEMM behind a folded
// LayerNorm reads (gemm16_8ph_kernel LNF: eight pairs per lane and tile instead of eight times C/64).  One thread per pixel.
__global__ __launch_bounds__(256) void ln_stats_merge_kernel(const float* __restrict__ stats, float2* __restrict__ mr, long rows, int C, float eps) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= rows) return;
  const int nchunk = C / DS_CH;
  const float* sp = stats + p * (2 * nchunk);
  float mean = 0.f;
  for (int c = 0; c < nchunk; ++c) mean += sp[2 * c];
  mean *= 1.0f / (float)nchunk;
  float m2 = 0.f;
  for (int c = 0; c < nchunk; ++c) {
    const float d = sp[2 * c] - mean;
    m2 += sp[2 * c + 1] + (float)DS_CH * d * d;
  }
  mr[p] = make_float2(mean, 1.0f / sqrtf(m2 / (float)C + eps));
}

inline int grid_for(long total, int block = 256, int cap = 256 * 16) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

template <typename TI, typename TO>
static int layernorm_impl(const TI* x, const float* w, const float* b, TO* y, long rows, int C, float eps, void* stream, bool split);

template <typename TA>
static int dwconv7_ln_impl(const TA* x, const float* w_dw, const float* b_dw, const float* ln_w, const float* ln_b, TA* y, int B, int H,
                           int W, int C, float eps, void* stream, bool split) {
  constexpr bool F32 = sizeof(TA) == 4;  // (the split operand format exists for fp32 storage only)
  KPF_REQUIRE(x && w_dw && b_dw && ln_w && ln_b && y, "kpf_dwconv7_ln: null pointer");
  KPF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C <= 2048, "kpf_dwconv7_ln: bad shape B=%d H=%d W=%d C=%d", B, H, W, C);
  KPF_REQUIRE(kpf_aligned16(x) && kpf_aligned16(y) && kpf_aligned16(w_dw), "kpf_dwconv7_ln: unaligned pointer");
  KPF_REQUIRE(!split || (F32 && C % 32 == 0), "kpf_dwconv7_ln_split_f32: C=%d must be a multiple of 32 (fp32 storage)", C);
  const int C4 = C / 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static const int tile_env = []() { const char* e = getenv("KPF_DW_TILE"); return e ? atoi(e) : 1; }();  // tuning aid
  if (tile_env && (C == 96 || C == 128 || C == 192) && H >= 16 && W >= 16) {
    // LDS-tiled, fused: every input pixel crosses L2 -> L1 about twice instead of seven times
    const int TX = C == 192 ? 16 : 32;
    const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + DT_TY - 1) / DT_TY;
    const dim3 grid((unsigned)((long)B * tiles_y * tiles_x));
#define KPF_DWT(NCHK, PXS)                                                                                                                     \
  do {                                                                                                                                         \
    if constexpr (F32) {                                                                                                                       \
      if (split) {                                                                                                                             \
        hipLaunchKernelGGL((dwconv7_ln_tile_kernel<TA, NCHK, PXS, true>), grid, dim3(256), 0, st, x, w_dw, b_dw, ln_w, ln_b, y, B, H, W,       \
                           tiles_x, tiles_y, eps);                                                                                             \
        break;                                                                                                                                 \
      }                                                                                                                                        \
    }                                                                                                                                          \
    hipLaunchKernelGGL((dwconv7_ln_tile_kernel<TA, NCHK, PXS, false>), grid, dim3(256), 0, st, x, w_dw, b_dw, ln_w, ln_b, y, B, H, W, tiles_x, \
                       tiles_y, eps);                                                                                                          \
  } while (0)
    if (C == 96) KPF_DWT(3, 8);
    else if (C == 128) KPF_DWT(4, 8);
    else KPF_DWT(6, 4);
#undef KPF_DWT
    return kpf_check_launch("kpf_dwconv7_ln");
  }
  if (H * W >= 64 && C4 <= 64 && (size_t)49 * C * sizeof(float) <= 64 * 1024 && !getenv("KPF_DW_UNFUSED")) {
    // fused, one pass over the activation: channel quads of a pixel on one 32- or 64-lane group, LayerNorm by shuffles
    const int xstrips = (W + 7) / 8, ypairs = (H + 1) / 2;
    const long groups = (long)B * ypairs * xstrips;
    const size_t wbytes = (size_t)49 * C * sizeof(float);
#define KPF_DWW(LG)                                                                                                                            \
  do {                                                                                                                                         \
    const long threads = groups * LG;                                                                                                          \
    const dim3 grid((unsigned)((threads + 255) / 256));                                                                                        \
    if constexpr (F32) {                                                                                                                       \
      if (split) {                                                                                                                             \
        hipLaunchKernelGGL((dwconv7_ln_wave_kernel<TA, LG, true>), grid, dim3(256), wbytes, st, x, w_dw, b_dw, ln_w, ln_b, y, B, H, W, C,      \
                           xstrips, ypairs, eps);                                                                                              \
        break;                                                                                                                                 \
      }                                                                                                                                        \
    }                                                                                                                                          \
    hipLaunchKernelGGL((dwconv7_ln_wave_kernel<TA, LG, false>), grid, dim3(256), wbytes, st, x, w_dw, b_dw, ln_w, ln_b, y, B, H, W, C,         \
                       xstrips, ypairs, eps);                                                                                                  \
  } while (0)
    if (C4 <= 32) KPF_DWW(32);
    else KPF_DWW(64);
#undef KPF_DWW
    return kpf_check_launch("kpf_dwconv7_ln");
  }
  static const int wide_env = []() { const char* e = getenv("KPF_DW_WIDE"); return e ? atoi(e) : 1; }();  // tuning aid
  if (wide_env && C4 > 64 && C4 <= 256 && H * W >= 16) {
    const int xstrips = (W + 7) / 8;
    // one-row strips while two-row strips would leave half the CUs without a workgroup (KPF_DW_WIDE_ROWS=1|2 forces a form: tuning aid)
    static const int rows_env = []() { const char* e = getenv("KPF_DW_WIDE_ROWS"); return e ? atoi(e) : 0; }();
    const bool one_row = rows_env ? rows_env == 1 : (long)B * ((H + 1) / 2) * xstrips < 256;
    const int ypairs = one_row ? H : (H + 1) / 2;  // row groups
    const dim3 grid((unsigned)((long)B * ypairs * xstrips));
#define KPF_DWX2(NT, ROWS)                                                                                                                     \
  do {                                                                                                                                         \
    if constexpr (F32) {                                                                                                                       \
      if (split) {                                                                                                                             \
        hipLaunchKernelGGL((dwconv7_ln_wide_kernel<TA, NT, true, ROWS>), grid, dim3(NT), 0, st, x, w_dw, b_dw, ln_w, ln_b, y, B, H, W, C,      \
                           xstrips, ypairs, eps);                                                                                              \
        break;                                                                                                                                 \
      }                                                                                                                                        \
    }                                                                                                                                          \
    hipLaunchKernelGGL((dwconv7_ln_wide_kernel<TA, NT, false, ROWS>), grid, dim3(NT), 0, st, x, w_dw, b_dw, ln_w, ln_b, y, B, H, W, C,         \
                       xstrips, ypairs, eps);                                                                                                  \
  } while (0)
#define KPF_DWX(NT)                                                                                                                            \
  do {                                                                                                                                         \
    if (one_row) KPF_DWX2(NT, 1);                                                                                                              \
    else KPF_DWX2(NT, 2);                                                                                                                      \
  } while (0)
    if (C4 <= 128) KPF_DWX(128);
    else if (C4 <= 192) KPF_DWX(192);
    else KPF_DWX(256);
#undef KPF_DWX2
#undef KPF_DWX
    return kpf_check_launch("kpf_dwconv7_ln");
  }
  if (H * W >= 64 || split) {
    // two launches: register-tiled depthwise conv, then the row LayerNorm in place (each streams the tensor once; measured faster
    // than the single fused kernel, whose LDS output tile caps occupancy).  Tiny maps keep the fused kernel (one launch).
    static const int px_env = []() { const char* e = getenv("KPF_DW_PX"); return e ? atoi(e) : 8; }();  // tuning aid
    const int PXr = px_env == 4 ? 4 : 8;
    const int xstrips = (W + PXr - 1) / PXr, ypairs = (H + 1) / 2;
    const long total = (long)B * ypairs * xstrips * C4;
    const size_t wbytes = (size_t)49 * C * sizeof(float);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (wbytes <= 48 * 1024) {
      if (PXr == 4) hipLaunchKernelGGL((dwconv7_kernel<TA, true, 4>), grid, dim3(256), wbytes, st, x, w_dw, b_dw, y, B, H, W, C, xstrips, ypairs);
      else hipLaunchKernelGGL((dwconv7_kernel<TA, true, 8>), grid, dim3(256), wbytes, st, x, w_dw, b_dw, y, B, H, W, C, xstrips, ypairs);
    } else {
      if (PXr == 4) hipLaunchKernelGGL((dwconv7_kernel<TA, false, 4>), grid, dim3(256), 0, st, x, w_dw, b_dw, y, B, H, W, C, xstrips, ypairs);
      else hipLaunchKernelGGL((dwconv7_kernel<TA, false, 8>), grid, dim3(256), 0, st, x, w_dw, b_dw, y, B, H, W, C, xstrips, ypairs);
    }
    int rc = kpf_check_launch("kpf_dwconv7_ln");
    if (rc) return rc;
    return layernorm_impl<TA, TA>(y, ln_w, ln_b, y, (long)B * H * W, C, eps, stream, split);
  }
  int S = 1;
  while (S * 2 * C4 <= 256 && S * DW_T < W) S *= 2;  // strips per block: <= 256 threads, no wider than the row
  const int threads = ((S * C4 + 63) / 64) * 64;
  KPF_REQUIRE(threads <= 512, "kpf_dwconv7_ln: C too large");
  const int xblocks = (W + S * DW_T - 1) / (S * DW_T);
  const size_t lds = (size_t)S * DW_T * C * sizeof(float);
  hipLaunchKernelGGL(dwconv7_ln_kernel<TA>, dim3((unsigned)((long)B * H * xblocks)), dim3(threads), lds, st, x, w_dw, b_dw, ln_w, ln_b, y, H, W, C,
                     S, xblocks, eps);
  return kpf_check_launch("kpf_dwconv7_ln");
}

// depthwise 7x7 + bias alone (training: forward of the ConvNeXt block's dwconv with the pre-norm activation kept for backward, and
// its data gradient = the same convolution of dY with the taps mirrored)
static int dwconv7_impl(const float* x, const float* w_dw, const float* b_dw, const float* addend, float* y, int B, int H, int W, int C, void* stream) {
  KPF_REQUIRE(x && w_dw && b_dw && y && x != y, "kpf_dwconv7_f32: null pointer or in-place call");
  KPF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "kpf_dwconv7_f32: bad shape (C %% 4 == 0)");
  KPF_REQUIRE(kpf_aligned16(x) && kpf_aligned16(y) && kpf_aligned16(w_dw) && kpf_aligned16(b_dw) && kpf_aligned16(addend),
              "kpf_dwconv7_f32: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  // work per thread: 2 rows x 8 pixels while that fills the chip (>= KPF_DW7_MIN_BLOCKS workgroups), else 4 pixels, else 1 row x 4 pixels — the 16 x 16 ... 4 x 4
  // maps of a training iteration are 96 - 192 workgroups in the widest form (tuning aids: KPF_DW7_PX = 8 | 4, KPF_DW7_ROWS = 2 | 1 force a form)
  static const int px_env = []() { const char* e = getenv("KPF_DW7_PX"); return e ? atoi(e) : 0; }();
  static const int rows_env = []() { const char* e = getenv("KPF_DW7_ROWS"); return e ? atoi(e) : 0; }();
  static const int min_blocks = []() { const char* e = getenv("KPF_DW7_MIN_BLOCKS"); return e ? atoi(e) : 768; }();
  auto blocks = [&](int px, int rows) { return ((long)B * ((H + rows - 1) / rows) * ((W + px - 1) / px) * (C / 4) + 255) / 256; };
  int px = 8, rows = 2;
  if (blocks(8, 2) < min_blocks || W <= 4) px = 4;
  if (blocks(px, 2) < min_blocks) rows = 1;
  if (px_env == 4 || px_env == 8) px = px_env;
  if (rows_env == 1 || rows_env == 2) rows = rows_env;
  const int xstrips = (W + px - 1) / px, ypairs = (H + rows - 1) / rows;
  const long total = (long)B * ypairs * xstrips * (C / 4);
  const size_t wbytes = (size_t)49 * C * sizeof(float);
  const dim3 grid((unsigned)((total + 255) / 256));
  const bool wl = wbytes <= 48 * 1024;
#define KPF_DW7(WL, PXV, R2V) hipLaunchKernelGGL((dwconv7_kernel<float, WL, PXV, R2V>), grid, dim3(256), WL ? wbytes : 0, st, x, w_dw, b_dw, y, B, H, W, C, xstrips, ypairs, addend)
  if (wl) {
    if (px == 8) { if (rows == 2) KPF_DW7(true, 8, true); else KPF_DW7(true, 8, false); }
    else { if (rows == 2) KPF_DW7(true, 4, true); else KPF_DW7(true, 4, false); }
  } else {
    if (px == 8) { if (rows == 2) KPF_DW7(false, 8, true); else KPF_DW7(false, 8, false); }
    else { if (rows == 2) KPF_DW7(false, 4, true); else KPF_DW7(false, 4, false); }
  }
#undef KPF_DW7
  return kpf_check_launch("kpf_dwconv7_f32");
}

extern "C" int kpf_dwconv7_f32(const float* x, const float* w_dw, const float* b_dw, float* y, int B, int H, int W, int C, void* stream) {
  return dwconv7_impl(x, w_dw, b_dw, nullptr, y, B, H, W, C, stream);
}

extern "C" int kpf_dwconv7_add_f32(const float* x, const float* w_dw, const float* b_dw, const float* addend, float* y, int B, int H, int W, int C, void* stream) {
  KPF_REQUIRE(addend && addend != y, "kpf_dwconv7_add_f32: addend missing or aliasing the output");
  return dwconv7_impl(x, w_dw, b_dw, addend, y, B, H, W, C, stream);
}

extern "C" int kpf_dwconv7_ln_f32(const float* x, const float* w_dw, const float* b_dw, const float* ln_w, const float* ln_b,
                                  float* y, int B, int H, int W, int C, float eps, void* stream) {
  return dwconv7_ln_impl<float>(x, w_dw, b_dw, ln_w, ln_b, y, B, H, W, C, eps, stream, false);
}
extern "C" int kpf_dwconv7_ln_split_f32(const float* x, const float* w_dw, const float* b_dw, const float* ln_w, const float* ln_b,
                                        float* y, int B, int H, int W, int C, float eps, void* stream) {
  return dwconv7_ln_impl<float>(x, w_dw, b_dw, ln_w, ln_b, y, B, H, W, C, eps, stream, true);
}
extern "C" int kpf_dwconv7_ln_h16(const void* x, const float* w_dw, const float* b_dw, const float* ln_w, const float* ln_b, void* y, int B,
                                  int H, int W, int C, float eps, int dtype, void* stream) {
  if (dtype == KPF_DT_BF16)
    return dwconv7_ln_impl<bf16_t>(static_cast<const bf16_t*>(x), w_dw, b_dw, ln_w, ln_b, static_cast<bf16_t*>(y), B, H, W, C, eps, stream, false);
  if (dtype == KPF_DT_F16)
    return dwconv7_ln_impl<f16_t>(static_cast<const f16_t*>(x), w_dw, b_dw, ln_w, ln_b, static_cast<f16_t*>(y), B, H, W, C, eps, stream, false);
  kpf_set_error("kpf_dwconv7_ln_h16: dtype must be KPF_DT_BF16 or KPF_DT_F16");
  return KPF_EINVAL;
}

template <typename TI, typename TO>
static int layernorm_impl(const TI* x, const float* w, const float* b, TO* y, long rows, int C, float eps, void* stream, bool split) {
  constexpr bool F32 = sizeof(TO) == 4;
  KPF_REQUIRE(x && w && b && y && rows > 0, "kpf_layernorm: null pointer / empty");
  KPF_REQUIRE(C % 4 == 0 && C > 0 && C <= 64 * 4 * LN_MAXV, "kpf_layernorm: C=%d unsupported", C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int C4 = C / 4;
  KPF_REQUIRE(!split || (F32 && C % 32 == 0), "kpf_layernorm_split_f32: C=%d must be a multiple of 32 (fp32 storage)", C);
#define KPF_LN(LPR, NV)                                                                                                                          \
  do {                                                                                                                                           \
    const dim3 g((unsigned)((rows + 256 / LPR - 1) / (256 / LPR)));                                                                              \
    if constexpr (F32) {                                                                                                                         \
      if (split) {                                                                                                                               \
        hipLaunchKernelGGL((layernorm_kernel<TI, TO, LPR, NV, true>), g, dim3(256), 0, st, x, w, b, y, rows, C, eps);                            \
        break;                                                                                                                                   \
      }                                                                                                                                          \
    }                                                                                                                                            \
    hipLaunchKernelGGL((layernorm_kernel<TI, TO, LPR, NV, false>), g, dim3(256), 0, st, x, w, b, y, rows, C, eps);                               \
  } while (0)
  if (C4 <= 16) KPF_LN(16, 1);
  else if (C4 <= 32) KPF_LN(32, 1);
  else if (C4 <= 64) KPF_LN(64, 1);
  else if (C4 <= 128) KPF_LN(64, 2);
  else if (C4 <= 256) KPF_LN(64, 4);
  else KPF_LN(64, LN_MAXV);
#undef KPF_LN
  return kpf_check_launch("kpf_layernorm");
}

extern "C" int kpf_layernorm_f32(const float* x, const float* w, const float* b, float* y, long rows, int C, float eps, void* stream) {
  return layernorm_impl<float, float>(x, w, b, y, rows, C, eps, stream, false);
}
extern "C" int kpf_layernorm_split_f32(const float* x, const float* w, const float* b, float* y, long rows, int C, float eps, void* stream) {
  return layernorm_impl<float, float>(x, w, b, y, rows, C, eps, stream, true);
}
extern "C" int kpf_layernorm_h16(const void* x, int x_dtype, const float* w, const float* b, void* y, int y_dtype, long rows, int C, float eps,
                                 void* stream) {
  // fp32 -> 16-bit (the stem's LayerNorm: its convolution runs in fp32 on the fp32 image) or 16-bit -> the same 16-bit type
  if (y_dtype == KPF_DT_BF16 && x_dtype == KPF_DT_F32)
    return layernorm_impl<float, bf16_t>(static_cast<const float*>(x), w, b, static_cast<bf16_t*>(y), rows, C, eps, stream, false);
  if (y_dtype == KPF_DT_F16 && x_dtype == KPF_DT_F32)
    return layernorm_impl<float, f16_t>(static_cast<const float*>(x), w, b, static_cast<f16_t*>(y), rows, C, eps, stream, false);
  if (y_dtype == KPF_DT_BF16 && x_dtype == KPF_DT_BF16)
    return layernorm_impl<bf16_t, bf16_t>(static_cast<const bf16_t*>(x), w, b, static_cast<bf16_t*>(y), rows, C, eps, stream, false);
  if (y_dtype == KPF_DT_F16 && x_dtype == KPF_DT_F16)
    return layernorm_impl<f16_t, f16_t>(static_cast<const f16_t*>(x), w, b, static_cast<f16_t*>(y), rows, C, eps, stream, false);
  kpf_set_error("kpf_layernorm_h16: unsupported dtype pair (%d -> %d)", x_dtype, y_dtype);
  return KPF_EINVAL;
}

template <typename TA>
static int upsample2x_impl(const TA* src, TA* dst, int B, int H, int W, int C, int dst_ld, int dst_coff, void* stream) {
  KPF_REQUIRE(src && dst && B > 0 && H > 0 && W > 0, "kpf_upsample2x: null pointer / empty");
  KPF_REQUIRE(C % 4 == 0 && dst_ld % 4 == 0 && dst_coff % 4 == 0 && dst_coff + C <= dst_ld, "kpf_upsample2x: bad channel slice");
  const long total = (long)B * 4 * H * W * (C / 4);
  if (sizeof(TA) == 2 && C % 8 == 0 && dst_ld % 8 == 0 && dst_coff % 8 == 0)  // 16-byte accesses
    hipLaunchKernelGGL((upsample2x_kernel<TA, 2>), dim3(grid_for(total / 2, 256, 256 * 32)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, B, H, W,
                       C / 4, dst_ld, dst_coff);
  else
    hipLaunchKernelGGL((upsample2x_kernel<TA, 1>), dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, B, H, W, C / 4,
                       dst_ld, dst_coff);
  return kpf_check_launch("kpf_upsample2x");
}
extern "C" int kpf_upsample2x_f32(const float* src, float* dst, int B, int H, int W, int C, int dst_ld, int dst_coff,
                                  void* stream) {
  return upsample2x_impl<float>(src, dst, B, H, W, C, dst_ld, dst_coff, stream);
}
extern "C" int kpf_upsample2x_h16(const void* src, void* dst, int B, int H, int W, int C, int dst_ld, int dst_coff, int dtype, void* stream) {
  if (dtype == KPF_DT_BF16) return upsample2x_impl<bf16_t>(static_cast<const bf16_t*>(src), static_cast<bf16_t*>(dst), B, H, W, C, dst_ld, dst_coff, stream);
  if (dtype == KPF_DT_F16) return upsample2x_impl<f16_t>(static_cast<const f16_t*>(src), static_cast<f16_t*>(dst), B, H, W, C, dst_ld, dst_coff, stream);
  kpf_set_error("kpf_upsample2x_h16: dtype must be KPF_DT_BF16 or KPF_DT_F16");
  return KPF_EINVAL;
}

// 16-bit NHWC slice -> dense fp32 (the 128-channel feature maps handed to the fp32 fusion head / returned at the module boundary)
extern "C" int kpf_cast_h16_f32(const void* src, int dtype, float* dst, long rows, int C, int src_ld, int src_coff, void* stream) {
  KPF_REQUIRE(src && dst && rows > 0 && C > 0 && C % 4 == 0 && src_ld % 4 == 0 && src_coff % 4 == 0 && src_coff + C <= src_ld, "kpf_cast_h16_f32: bad arguments");
  const long total = rows * (C / 4);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == KPF_DT_BF16)
    hipLaunchKernelGGL(cast_h16_f32_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, static_cast<const bf16_t*>(src), dst, total, C / 4, src_ld, src_coff);
  else if (dtype == KPF_DT_F16)
    hipLaunchKernelGGL(cast_h16_f32_kernel<f16_t>, dim3(grid_for(total)), dim3(256), 0, st, static_cast<const f16_t*>(src), dst, total, C / 4, src_ld, src_coff);
  else {
    kpf_set_error("kpf_cast_h16_f32: dtype must be KPF_DT_BF16 or KPF_DT_F16");
    return KPF_EINVAL;
  }
  return kpf_check_launch("kpf_cast_h16_f32");
}

// dense fp32 -> 16-bit (behind the fp32 stem + max-pool of the ResNet backbones)
extern "C" int kpf_cast_f32_h16(const float* src, void* dst, int dtype, long n, void* stream) {
  KPF_REQUIRE(src && dst && n > 0 && n % 4 == 0, "kpf_cast_f32_h16: bad arguments (n %% 4 == 0)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == KPF_DT_BF16)
    hipLaunchKernelGGL(cast_f32_h16_kernel<bf16_t>, dim3(grid_for(n / 4)), dim3(256), 0, st, src, static_cast<bf16_t*>(dst), n / 4);
  else if (dtype == KPF_DT_F16)
    hipLaunchKernelGGL(cast_f32_h16_kernel<f16_t>, dim3(grid_for(n / 4)), dim3(256), 0, st, src, static_cast<f16_t*>(dst), n / 4);
  else {
    kpf_set_error("kpf_cast_f32_h16: dtype must be KPF_DT_BF16 or KPF_DT_F16");
    return KPF_EINVAL;
  }
  return kpf_check_launch("kpf_cast_f32_h16");
}

extern "C" int kpf_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, int Cpad, void* stream) {
  KPF_REQUIRE(src && dst && B > 0 && C > 0 && Cpad >= C, "kpf_nchw_to_nhwc_f32: bad arguments");
  const long total = (long)B * H * W * Cpad;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, B, C,
                     H * W, Cpad);
  return kpf_check_launch("kpf_nchw_to_nhwc_f32");
}

extern "C" int kpf_nhwc_to_nchw_f32(const float* src, float* dst, int B, int C, int H, int W, int src_ld, int src_coff,
                                    void* stream) {
  KPF_REQUIRE(src && dst && B > 0 && C > 0 && src_coff + C <= src_ld, "kpf_nhwc_to_nchw_f32: bad arguments");
  const int HW = H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, dst, C, HW, src_ld, src_coff);
  return kpf_check_launch("kpf_nhwc_to_nchw_f32");
}

extern "C" int kpf_maxpool3x3s2_f32(const float* src, float* dst, int B, int H, int W, int C, void* stream) {
  KPF_REQUIRE(src && dst && B > 0 && C % 4 == 0, "kpf_maxpool3x3s2_f32: bad arguments");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long total = (long)B * OH * OW * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, B, H,
                     W, OH, OW, C / 4);
  return kpf_check_launch("kpf_maxpool3x3s2_f32");
}

// ---- round 4: stencil + statistics, LayerNorm from statistics (16-bit storage) ----
extern "C" int kpf_dwconv7_stats_supported(int H, int W, int C) { return C % 64 == 0 && C <= 2048 && H >= 16 && W >= 16; }
extern "C" long kpf_dwconv7_stats_floats(int B, int H, int W, int C) { return (long)B * H * W * (C / 64) * 2; }

template <typename TA>
static int dwconv7_stats_impl(const TA* x, const float* w_dw, const float* b_dw, TA* y, float* stats, int B, int H, int W, int C, void* stream) {
  const int tiles_x = (W + DS_T - 1) / DS_T, tiles_y = (H + DS_T - 1) / DS_T;
  const long blocks = (long)B * tiles_y * tiles_x * (C / DS_CH);
  KPF_REQUIRE(blocks < (1l << 31), "kpf_dwconv7_stats_h16: grid too large");
  static const float* zero_of_dev[KPF_MAX_DEVICES] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= KPF_MAX_DEVICES) dev = 0;
  if (!zero_of_dev[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(kpf_elem_zero16)) != hipSuccess || !p) {
      kpf_set_error("kpf_dwconv7_stats_h16: cannot resolve the zero page");
      return KPF_ELAUNCH;
    }
    zero_of_dev[dev] = static_cast<const float*>(p);
  }
  const size_t lds = (size_t)(DS_HW * DS_ROW * 8 + 63) / 64 * 1024 + (49 * 64 + 64) * sizeof(float);
  auto kern = dwconv7_stats_kernel<TA>;
  static std::atomic<bool> lds_opt_in[KPF_MAX_DEVICES];
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(kern), lds_opt_in)) {
    kpf_set_error("kpf_dwconv7_stats_h16: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  static const int dbg = []() { const char* e = getenv("KPF_DWS_DBG"); return e ? atoi(e) : 0; }();  // tuning aid (ablation bits: see the kernel)
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), x, w_dw, b_dw, y, stats, H, W, C, tiles_x, tiles_y,
                     zero_of_dev[dev], dbg);
  return kpf_check_launch("kpf_dwconv7_stats_h16");
}

extern "C" int kpf_dwconv7_stats_h16(const void* x, const float* w_dw, const float* b_dw, void* y, float* stats, int B, int H, int W, int C, int dtype,
                                     void* stream) {
  KPF_REQUIRE(x && w_dw && b_dw && y && stats && x != y, "kpf_dwconv7_stats_h16: null pointer or in-place call");
  KPF_REQUIRE(B > 0 && kpf_dwconv7_stats_supported(H, W, C), "kpf_dwconv7_stats_h16: shape B=%d H=%d W=%d C=%d not supported (C %% 64 == 0, H, W >= 16)", B, H, W, C);
  KPF_REQUIRE(kpf_aligned16(x) && kpf_aligned16(y) && kpf_aligned16(w_dw) && kpf_aligned16(b_dw) && kpf_aligned16(stats), "kpf_dwconv7_stats_h16: unaligned pointer");
  if (dtype == KPF_DT_BF16) return dwconv7_stats_impl<bf16_t>(static_cast<const bf16_t*>(x), w_dw, b_dw, static_cast<bf16_t*>(y), stats, B, H, W, C, stream);
  if (dtype == KPF_DT_F16) return dwconv7_stats_impl<f16_t>(static_cast<const f16_t*>(x), w_dw, b_dw, static_cast<f16_t*>(y), stats, B, H, W, C, stream);
  kpf_set_error("kpf_dwconv7_stats_h16: dtype must be KPF_DT_BF16 or KPF_DT_F16");
  return KPF_EINVAL;
}

extern "C" int kpf_ln_stats_merge(const float* stats, float* mean_rstd, long rows, int C, float eps, void* stream) {
  KPF_REQUIRE(stats && mean_rstd && rows > 0 && C > 0 && C % 64 == 0, "kpf_ln_stats_merge: bad arguments (C %% 64 == 0)");
  KPF_REQUIRE(reinterpret_cast<uintptr_t>(stats) % 8 == 0 && reinterpret_cast<uintptr_t>(mean_rstd) % 8 == 0, "kpf_ln_stats_merge: unaligned pointer");
  KPF_REQUIRE((rows + 255) / 256 < (1l << 31), "kpf_ln_stats_merge: too many rows");
  hipLaunchKernelGGL(ln_stats_merge_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), stats,
                     reinterpret_cast<float2*>(mean_rstd), rows, C, eps);
  return kpf_check_launch("kpf_ln_stats_merge");
}

extern "C" int kpf_ln_apply_stats_h16(void* y, const float* stats, const float* ln_w, const float* ln_b, long rows, int C, float eps, int dtype, void* stream) {
  KPF_REQUIRE(y && stats && ln_w && ln_b && rows > 0 && C > 0 && C % 64 == 0, "kpf_ln_apply_stats_h16: bad arguments (C %% 64 == 0)");
  KPF_REQUIRE(kpf_aligned16(y) && kpf_aligned16(ln_w) && kpf_aligned16(ln_b), "kpf_ln_apply_stats_h16: unaligned pointer");
  KPF_REQUIRE((rows + 63) / 64 < (1l << 31), "kpf_ln_apply_stats_h16: too many rows");
  const dim3 grid((unsigned)((rows + 63) / 64));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == KPF_DT_BF16)
    hipLaunchKernelGGL(ln_apply_stats_kernel<bf16_t>, grid, dim3(256), 0, st, static_cast<bf16_t*>(y), stats, ln_w, ln_b, rows, C, eps);
  else if (dtype == KPF_DT_F16)
    hipLaunchKernelGGL(ln_apply_stats_kernel<f16_t>, grid, dim3(256), 0, st, static_cast<f16_t*>(y), stats, ln_w, ln_b, rows, C, eps);
  else {
    kpf_set_error("kpf_ln_apply_stats_h16: dtype must be KPF_DT_BF16 or KPF_DT_F16");
    return KPF_EINVAL;
  }
  return kpf_check_launch("kpf_ln_apply_stats_h16");
}
