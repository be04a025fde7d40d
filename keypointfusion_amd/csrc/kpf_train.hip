// Deterministic backward (and the matching forward) kernels of the training step's data-movement ops:
//
//   kpf_upsample2x_bwd      gradient of the bilinear x2 of the UNet decoder (model/resnetUnet.py:259, align_corners False)
//   kpf_maxpool3x3s2_fwd/bwd ResNet's 3x3 / s2 / pad 1 max-pool (model/resnet.py:168) with the argmax tap kept for the backward
//   kpf_row_gather_fwd/bwd  weighted row gathers: the 4-nearest-pixel feature sampling of the points (model/model.py:297-306) and the
//                           ball-query grouping of DESA (model/model.py:174 -> pointnet2 group_points)
//
// The library kernels behind torch's autograd for these ops accumulate with float atomics, so two runs of the same iteration
// differ in the last bits and, through the integer decisions of the fusion head, occasionally in whole joints.  Everything here is
// written in GATHER form: one thread (or wave) owns an output element and adds its contributions in a fixed order — no atomics on
// floating-point data, run-to-run bit-identical.  All of it is HBM-bound data movement (read the gradient once, write once).
#include "kpf_common.h"

namespace {

inline int grid_for(long total, int block = 256, int cap = 256 * 16) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------------------------------------------------------
// bilinear x2 backward.  Forward (kpf_elem.hip, upsample2x_kernel): f = (o + 0.5) / 2 - 0.5 clamped at 0, i0 = floor(f),
// i1 = min(i0 + 1, n - 1), weights (1 - l, l).  A source row y receives from destination rows 2y-1 .. 2y+2; the weight is recomputed
// with the forward's own expressions so that forward and backward are exact transposes of each other.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float up2_weight(int o, int i, int n) {
  float f = (o + 0.5f) * 0.5f - 0.5f;
  f = f < 0.f ? 0.f : f;
  const int i0 = (int)f;
  const int i1 = i0 + (i0 < n - 1 ? 1 : 0);
  const float l = f - (float)i0;
  return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}

template <typename TA>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const TA* __restrict__ dy, TA* __restrict__ dx, int B, int H, int W, int C4) {
  const long total = (long)B * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % C4);
    long p = i / C4;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dyo = -1; dyo <= 2; ++dyo) {
      const int oy = 2 * y + dyo;
      if ((unsigned)oy >= (unsigned)(2 * H)) continue;
      const float wy = up2_weight(oy, y, H);
      if (wy == 0.f) continue;
#pragma unroll
      for (int dxo = -1; dxo <= 2; ++dxo) {
        const int ox = 2 * x + dxo;
        if ((unsigned)ox >= (unsigned)(2 * W)) continue;
        const float wx = up2_weight(ox, x, W);
        if (wx == 0.f) continue;
        const f32x4 g = kpf_ld4(dy + ((((long)b * 2 * H + oy) * 2 * W + ox) * C4 + q) * 4);
        const float w = wy * wx;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += w * g[e];
      }
    }
    kpf_st4(dx + i * 4, acc);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// max-pool 3x3 / s2 / pad 1 with the winning tap (0..8, scan order ky, kx; the FIRST maximum wins, NaN propagates like ATen's
// `val > max || isnan(val)`) stored per output element; the backward lets every input element look at the <= 4 windows that cover it.
// ---------------------------------------------------------------------------------------------------------------
template <typename TA>
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_kernel(const TA* __restrict__ src, TA* __restrict__ dst, unsigned char* __restrict__ tap,
                                                               int B, int H, int W, int OH, int OW, int C4) {
  const long total = (long)B * OH * OW * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % C4);
    long p = i / C4;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int t[4] = {-1, -1, -1, -1};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 + ky - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 + kx - 1;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = kpf_ld4(src + ((((long)b * H + iy) * W + ix) * C4 + q) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t[e] < 0 || v[e] > m[e] || v[e] != v[e]) {
            m[e] = v[e];
            t[e] = ky * 3 + kx;
          }
      }
    }
    kpf_st4(dst + i * 4, m);
    *reinterpret_cast<uchar4*>(tap + i * 4) = make_uchar4((unsigned char)t[0], (unsigned char)t[1], (unsigned char)t[2], (unsigned char)t[3]);
  }
}

template <typename TA>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const TA* __restrict__ dy, const unsigned char* __restrict__ tap, TA* __restrict__ dx,
                                                               int B, int H, int W, int OH, int OW, int C4) {
  const long total = (long)B * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % C4);
    long p = i / C4;
    const int ix = (int)(p % W);
    p /= W;
    const int iy = (int)(p % H);
    const int b = (int)(p / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // windows oy with 2*oy - 1 <= iy <= 2*oy + 1
    const int oy_lo = iy >> 1, oy_hi = (iy + 1) >> 1;
    const int ox_lo = ix >> 1, ox_hi = (ix + 1) >> 1;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      if (oy >= OH) continue;
      const int ky = iy - 2 * oy + 1;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        if (ox >= OW) continue;
        const int kx = ix - 2 * ox + 1;
        const long o = (((long)b * OH + oy) * OW + ox) * C4 + q;
        const uchar4 t = *reinterpret_cast<const uchar4*>(tap + o * 4);
        const f32x4 g = kpf_ld4(dy + o * 4);
        const int want = ky * 3 + kx;
        if (t.x == want) acc[0] += g[0];
        if (t.y == want) acc[1] += g[1];
        if (t.z == want) acc[2] += g[2];
        if (t.w == want) acc[3] += g[3];
      }
    }
    kpf_st4(dx + i * 4, acc);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weighted row gather:  out[b][r][:] = sum_{g < G} w[b][r*G + g] * src[b][idx[b][r*G + g]][:]      (w == nullptr: weights 1)
// One wave per output row; lanes over channel quads (C % 4 == 0), rows of C fp32.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_gather_fwd_kernel(const float* __restrict__ src, const int* __restrict__ idx, const float* __restrict__ w,
                                                             float* __restrict__ out, int B, int P, int R, int G, int C4) {
  const int lane = threadIdx.x & 63;
  const long rows = (long)B * R;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
    const int b = (int)(r / R);
    const int* ip = idx + r * G;
    const float* wp = w ? w + r * G : nullptr;
    for (int q = lane; q < C4; q += 64) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int g = 0; g < G; ++g) {
        const f32x4 v = kpf_ld4(src + ((long)b * P + ip[g]) * C4 * 4 + 4 * q);
        const float ww = wp ? wp[g] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += ww * v[e];
      }
      kpf_st4(out + r * C4 * 4 + 4 * q, acc);
    }
  }
}

// backward:  dsrc[b][p][:] = sum over entries e (ascending) with idx[b][e] == p of w[b][e] * dout[b][e / G][:]
// Two kernels.  (1) row_gather_invert_kernel, one 16-wave workgroup per image: a STABLE counting sort of the entry numbers by source row.
// Each wave owns a contiguous range of entries: per-wave counts by integer LDS atomics, a workgroup scan that gives every (wave, row) its
// first slot (lower waves first), then each wave ranks its range 64 entries at a time in entry order — the loop visits the distinct rows
// among its 64 lanes (shuffle + ballot), a lane's slot is its row's next free slot + the number of lower lanes with the same row.
// The lists therefore hold ascending entry numbers whatever the timing: the summation order below is fixed.
// (2) row_gather_accum_kernel, one wave per source row: lanes over channel quads; with C <= 128 the two half-waves take alternate list
// positions (two accumulators, added lower + upper at the end: still one fixed order).  E <= 8192 entries, P <= 4096 rows per image (round 5: the 64 x 64
// feature map of the wide model; 16 waves per image up to 2048 rows, 8 beyond — the per-wave count table is INV_W * P ints of LDS).
constexpr int GATHER_MAX_E = 8192, GATHER_MAX_P = 4096;

template <int INV_W>  // waves per image in the inversion
__global__ __launch_bounds__(64 * INV_W) void row_gather_invert_kernel(const int* __restrict__ idx, int* __restrict__ start, int* __restrict__ list, int P,
                                                                      int E) {
  extern __shared__ int inv_lds[];  // cnt[INV_W][P] (per-wave counts, then per-wave next-free slots) | tot[P] | wsum[INV_W]
  int* cnt = inv_lds;
  int* tot = inv_lds + INV_W * P;
  int* wsum = tot + P;  // (no static LDS next to the dynamic block: the opt-in above 64 KB covers the dynamic size only)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int* ib = idx + (long)b * E;
  int* sb = start + (long)b * (P + 1);
  int* lb = list + (long)b * E;
  const int seg = ((E + INV_W - 1) / INV_W + 63) / 64 * 64;  // entries per wave: a contiguous range, whole 64-chunks
  const int e_lo = wave * seg, e_hi = min(E, e_lo + seg);
  for (int i = tid; i < INV_W * P; i += 64 * INV_W) cnt[i] = 0;
  __syncthreads();
  int* mine = cnt + wave * P;
  for (int e = e_lo + lane; e < e_hi; e += 64) atomicAdd(&mine[ib[e]], 1);  // integer counts: order-independent
  __syncthreads();
  for (int p = tid; p < P; p += 64 * INV_W) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < INV_W; ++w) t += cnt[w * P + p];
    tot[p] = t;
  }
  __syncthreads();
  // exclusive scan of tot[0..P) by the whole workgroup: thread t owns the contiguous range [t*per, (t+1)*per)
  const int per = (P + 64 * INV_W - 1) / (64 * INV_W);
  int local = 0;
  for (int j = 0; j < per; ++j) {
    const int p = tid * per + j;
    if (p < P) local += tot[p];
  }
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - local;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  for (int j = 0; j < per; ++j) {
    const int p = tid * per + j;
    if (p < P) {
      sb[p] = base;
      int run = base;
#pragma unroll
      for (int w = 0; w < INV_W; ++w) {  // wave w's entries of row p go behind those of the waves before it (they hold lower entry numbers)
        const int c = cnt[w * P + p];
        cnt[w * P + p] = run;
        run += c;
      }
      base = run;
    }
  }
  if (tid == 0) sb[P] = E;
  __syncthreads();
  // every wave ranks its own range, 64 entries at a time in entry order: the loop visits the distinct rows among the 64 lanes; a lane's
  // slot is its row's next free slot + the number of lower lanes with the same row (only this wave touches cnt[wave][*] from here on)
  for (int e0 = e_lo; e0 < e_hi; e0 += 64) {
    const int e = e0 + lane;
    const bool valid = e < e_hi;
    const int key = valid ? ib[e] : -1;
    unsigned long long todo = __ballot(valid);
    int slot = 0;
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const int k = __shfl(key, lead, 64);
      const unsigned long long same = __ballot(valid && key == k);
      const int first = mine[k];  // (every lane reads the same word: broadcast)
      if (valid && key == k) slot = first + __popcll(same & ((1ull << lane) - 1ull));
      __builtin_amdgcn_wave_barrier();
      if (lane == lead) mine[k] = first + __popcll(same);
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the LDS write has landed before the next iteration's read
      __builtin_amdgcn_wave_barrier();
      todo &= ~same;
    }
    if (valid) lb[slot] = e;
  }
}

__global__ __launch_bounds__(256) void row_gather_accum_kernel(const float* __restrict__ dout, const int* __restrict__ start, const int* __restrict__ list,
                                                               const float* __restrict__ w, float* __restrict__ dsrc, int B, int P, int E, int G, int C4) {
  const int lane = threadIdx.x & 63;
  const long rows = (long)B * P;
  const bool halves = C4 <= 32;  // two half-waves on alternate list positions
  const int q0 = halves ? (lane & 31) : lane;
  const int phase = halves ? (lane >> 5) : 0, nph = halves ? 2 : 1;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
    const int b = (int)(r / P), p = (int)(r - (long)b * P);
    const int s0 = start[(long)b * (P + 1) + p], s1 = start[(long)b * (P + 1) + p + 1];
    const int* lb = list + (long)b * E;
    const float* wb = w ? w + (long)b * E : nullptr;
    for (int q = q0; q < (halves ? 32 : C4 + 63 - (C4 + 63) % 64); q += (halves ? 32 : 64)) {
      const bool live = q < C4;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // The list is walked 64 positions at a time: every lane fetches ONE index (and weight) of the chunk, the wave then broadcasts them position by
      // position (no dependent index load in front of every row load), four row loads in flight, added in position order (same sums as a plain walk).
      for (int c0 = s0; c0 < s1; c0 += 64) {
        const int m = min(64, s1 - c0);
        const int ev = lane < m ? lb[c0 + lane] : 0;
        const float wv = (wb && lane < m) ? wb[ev] : 1.f;
        int k = phase;
        for (; (k - phase) + 4 * nph - 1 < m; k += 4 * nph) {  // (the same trip count for both half-waves: every shuffle runs with the whole wave active)
          int e[4];
          float ww[4];
          f32x4 g[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            e[u] = __shfl(ev, k + u * nph, 64);
            ww[u] = __shfl(wv, k + u * nph, 64);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) g[u] = live ? kpf_ld4(dout + (((long)b * E + e[u]) / G) * C4 * 4 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += ww[u] * g[u][t];
        }
        for (; k < m + phase; k += nph) {  // (uniform trip count for both half-waves: the shuffles are executed by all lanes; positions >= m add nothing)
          const int kk = min(k, m - 1);
          const int e1 = __shfl(ev, kk, 64);
          const float w1 = __shfl(wv, kk, 64);
          if (live && k < m) {
            const f32x4 g1 = kpf_ld4(dout + (((long)b * E + e1) / G) * C4 * 4 + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += w1 * g1[t];
          }
        }
      }
      if (halves) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += __shfl_down(acc[k], 32, 64);  // even positions + odd positions, always in this order
        if (phase == 0 && live) kpf_st4(dsrc + r * C4 * 4 + 4 * q, acc);
      } else if (live) {
        kpf_st4(dsrc + r * C4 * 4 + 4 * q, acc);
      }
    }
  }
}

template <typename TA>
int up_bwd(const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(upsample2x_bwd_kernel<TA>, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), static_cast<const TA*>(dy),
                     static_cast<TA*>(dx), B, H, W, C / 4);
  return kpf_check_launch("kpf_upsample2x_bwd");
}
template <typename TA>
int mp_fwd(const void* x, void* y, unsigned char* tap, int B, int H, int W, int C, void* stream) {
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long total = (long)B * OH * OW * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel<TA>, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), static_cast<const TA*>(x),
                     static_cast<TA*>(y), tap, B, H, W, OH, OW, C / 4);
  return kpf_check_launch("kpf_maxpool3x3s2_fwd");
}
template <typename TA>
int mp_bwd(const void* dy, const unsigned char* tap, void* dx, int B, int H, int W, int C, void* stream) {
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel<TA>, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), static_cast<const TA*>(dy), tap,
                     static_cast<TA*>(dx), B, H, W, OH, OW, C / 4);
  return kpf_check_launch("kpf_maxpool3x3s2_bwd");
}

}  // namespace

#define KPF_DISPATCH_DT(dtype, what, CALL)                                 \
  do {                                                                     \
    if ((dtype) == KPF_DT_F32) return CALL(float);                         \
    if ((dtype) == KPF_DT_BF16) return CALL(bf16_t);                       \
    if ((dtype) == KPF_DT_F16) return CALL(f16_t);                         \
    kpf_set_error(what ": dtype must be KPF_DT_F32 / _BF16 / _F16");        \
    return KPF_EINVAL;                                                     \
  } while (0)

extern "C" int kpf_upsample2x_bwd(const void* dy, void* dx, int dtype, int B, int H, int W, int C, void* stream) {
  KPF_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "kpf_upsample2x_bwd: bad arguments");
#define CALL(T) up_bwd<T>(dy, dx, B, H, W, C, stream)
  KPF_DISPATCH_DT(dtype, "kpf_upsample2x_bwd", CALL);
#undef CALL
}

extern "C" int kpf_maxpool3x3s2_fwd(const void* x, void* y, unsigned char* tap, int dtype, int B, int H, int W, int C, void* stream) {
  KPF_REQUIRE(x && y && tap && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "kpf_maxpool3x3s2_fwd: bad arguments");
#define CALL(T) mp_fwd<T>(x, y, tap, B, H, W, C, stream)
  KPF_DISPATCH_DT(dtype, "kpf_maxpool3x3s2_fwd", CALL);
#undef CALL
}

extern "C" int kpf_maxpool3x3s2_bwd(const void* dy, const unsigned char* tap, void* dx, int dtype, int B, int H, int W, int C, void* stream) {
  KPF_REQUIRE(dy && dx && tap && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "kpf_maxpool3x3s2_bwd: bad arguments");
#define CALL(T) mp_bwd<T>(dy, tap, dx, B, H, W, C, stream)
  KPF_DISPATCH_DT(dtype, "kpf_maxpool3x3s2_bwd", CALL);
#undef CALL
}

// the same gather for a slice of any width and layout (no channel quads; element (b, p, c) at b * sb + p * sp + c * sc: a column slice of NHWC rows, channel
// planes of an NCHW map): the 21 weight-logit channels of the offset map that the pose tokens
// sample (model/model.py:372-376; no gradient: the reference detaches them) — one thread per output element; the library path was cast + gather + mul + sum
namespace {
__global__ __launch_bounds__(256) void row_gather_cols_kernel(const float* __restrict__ src, const int* __restrict__ idx, const float* __restrict__ w,
                                                              float* __restrict__ out, long total, int P, int R, int G, int C, long sb, long sp, long sc) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const int b = (int)(r / R);
    const int* ip = idx + r * G;
    float acc = 0.f;
    for (int g = 0; g < G; ++g) acc += (w ? w[r * G + g] : 1.f) * src[b * sb + ip[g] * sp + c * sc];
    out[i] = acc;
  }
}
}  // namespace

extern "C" int kpf_row_gather_cols_f32(const float* src, long sb, long sp, long sc, const int* idx, const float* w, float* out, int B, int P, int R, int G, int C,
                                       void* stream) {
  KPF_REQUIRE(src && idx && out && B > 0 && P > 0 && R > 0 && G > 0 && C > 0 && sb > 0 && sp > 0 && sc > 0, "kpf_row_gather_cols_f32: bad arguments");
  const long total = (long)B * R * C;
  hipLaunchKernelGGL(row_gather_cols_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, idx, w, out, total, P, R, G, C, sb, sp, sc);
  return kpf_check_launch("kpf_row_gather_cols_f32");
}

extern "C" int kpf_row_gather_fwd_f32(const float* src, const int* idx, const float* w, float* out, int B, int P, int R, int G, int C, void* stream) {
  KPF_REQUIRE(src && idx && out && B > 0 && P > 0 && R > 0 && G > 0 && C > 0 && C % 4 == 0, "kpf_row_gather_fwd_f32: bad arguments");
  const long rows = (long)B * R;
  hipLaunchKernelGGL(row_gather_fwd_kernel, dim3(grid_for(rows, 4, 256 * 32)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, idx, w, out, B, P, R,
                     G, C / 4);
  return kpf_check_launch("kpf_row_gather_fwd_f32");
}


// the inversion's launch: 16 waves per image while the count table fits (P <= 2048), 8 beyond
static int launch_row_gather_invert(const int* idx, int* start, int* list, int B, int P, int E, hipStream_t st, const char* who) {
  static std::atomic<bool> lds_opt_in[2][KPF_MAX_DEVICES];
  if (P <= 2048) {
    const size_t inv_lds = (size_t)(16 * P + P + 16) * sizeof(int);
    if (inv_lds > 64 * 1024 && !kpf_raise_lds_limit(reinterpret_cast<const void*>(&row_gather_invert_kernel<16>), lds_opt_in[0])) {
      kpf_set_error("%s: cannot raise the dynamic LDS limit", who);
      return KPF_ELAUNCH;
    }
    hipLaunchKernelGGL(row_gather_invert_kernel<16>, dim3(B), dim3(64 * 16), inv_lds, st, idx, start, list, P, E);
  } else {
    const size_t inv_lds = (size_t)(8 * P + P + 8) * sizeof(int);
    if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(&row_gather_invert_kernel<8>), lds_opt_in[1])) {
      kpf_set_error("%s: cannot raise the dynamic LDS limit", who);
      return KPF_ELAUNCH;
    }
    hipLaunchKernelGGL(row_gather_invert_kernel<8>, dim3(B), dim3(64 * 8), inv_lds, st, idx, start, list, P, E);
  }
  return kpf_check_launch(who);
}

extern "C" long kpf_row_gather_ws_ints(int B, int P, int R, int G) { return (long)B * (P + 1) + (long)B * R * G; }

extern "C" int kpf_row_gather_bwd_f32(const float* dout, const int* idx, const float* w, float* dsrc, int* ws, long ws_ints, int B, int P, int R, int G, int C,
                                      void* stream) {
  KPF_REQUIRE(dout && idx && dsrc && ws && B > 0 && P > 0 && R > 0 && G > 0 && C > 0 && C % 4 == 0, "kpf_row_gather_bwd_f32: bad arguments");
  const long E = (long)R * G;
  KPF_REQUIRE(E <= GATHER_MAX_E && P <= GATHER_MAX_P, "kpf_row_gather_bwd_f32: at most %d gathered entries and %d source rows per image", GATHER_MAX_E,
              GATHER_MAX_P);
  KPF_REQUIRE(ws_ints >= kpf_row_gather_ws_ints(B, P, R, G), "kpf_row_gather_bwd_f32: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int* start = ws;
  int* list = ws + (long)B * (P + 1);
  int rc = launch_row_gather_invert(idx, start, list, B, P, (int)E, st, "kpf_row_gather_bwd_f32 (invert)");
  if (rc != KPF_OK) return rc;
  hipLaunchKernelGGL(row_gather_accum_kernel, dim3(grid_for((long)B * P, 4, 256 * 32)), dim3(256), 0, st, dout, start, list, w, dsrc, B, P, (int)E, G, C / 4);
  return kpf_check_launch("kpf_row_gather_bwd_f32");
}

/* The two halves of kpf_row_gather_bwd_f32 on their own (ABI 13): several gathers that share one index tensor (the point sampling of both feature maps in both
 * fusion blocks; the three radii of DESA as 3B "images") invert it ONCE.  ws: [B*(P+1)] row starts, then [B*R*G] entry lists (kpf_row_gather_ws_ints). */
extern "C" int kpf_row_gather_invert(const int* idx, int* ws, long ws_ints, int B, int P, int R, int G, void* stream) {
  KPF_REQUIRE(idx && ws && B > 0 && P > 0 && R > 0 && G > 0, "kpf_row_gather_invert: bad arguments");
  const long E = (long)R * G;
  KPF_REQUIRE(E <= GATHER_MAX_E && P <= GATHER_MAX_P, "kpf_row_gather_invert: at most %d gathered entries and %d source rows per image", GATHER_MAX_E, GATHER_MAX_P);
  KPF_REQUIRE(ws_ints >= kpf_row_gather_ws_ints(B, P, R, G), "kpf_row_gather_invert: workspace too small");
  return launch_row_gather_invert(idx, ws, ws + (long)B * (P + 1), B, P, (int)E, reinterpret_cast<hipStream_t>(stream), "kpf_row_gather_invert");
}
extern "C" int kpf_row_gather_accum_f32(const float* dout, const int* start, const int* list, const float* w, float* dsrc, int B, int P, int R, int G, int C,
                                        void* stream) {
  KPF_REQUIRE(dout && start && list && dsrc && B > 0 && P > 0 && R > 0 && G > 0 && C > 0 && C % 4 == 0, "kpf_row_gather_accum_f32: bad arguments");
  hipLaunchKernelGGL(row_gather_accum_kernel, dim3(grid_for((long)B * P, 4, 256 * 32)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dout, start, list, w, dsrc,
                     B, P, R * G, G, C / 4);
  return kpf_check_launch("kpf_row_gather_accum_f32");
}

// ---------------------------------------------------------------------------------------------------------------
// Weight packing for the training step: the reference-layout OIHW master weight -> the kernel operand of the implicit GEMM, ONE launch
// per operand (the torch expressions this replaces were flip + permute-clone + pad [+ cast]: 3-4 small launches per convolution per
// step, ~500 per iteration).  mode 0: forward rows [n_pad][Kp], k = (ky, kx, c);  mode 1: data-gradient rows [Cin][Kp],
// k = (ky, kx, n) with the taps mirrored (the transposed convolution);  mode 2: patchify data-gradient rows [(ky, kx, c)][Kp], k = n
// (with Cin = 1 this is the depthwise convolution's tap table [KH*KW][C]);  mode 3: mode 2 with mirrored taps.
// Rows are zero-padded to Kp, output channels beyond N are zero.  TS / TD: source / destination element types (fp32 master -> fp32
// or 16-bit operand; a 16-bit source is copied as it is).
// ---------------------------------------------------------------------------------------------------------------
namespace {
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void pack_weight_kernel(const TS* __restrict__ w, TD* __restrict__ dst, int N, int Cin, int KH, int KW, int mode, int n_pad,
                                                          int Kp, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp);
    const int row = (int)(i / Kp);
    float v = 0.f;
    if (mode == 0) {
      const int c = k % Cin, t = k / Cin;  // t = ky*KW + kx
      if (t < KH * KW && row < N) v = (float)w[((long)row * Cin + c) * KH * KW + t];
    } else if (mode == 1) {
      const int n = k % n_pad, t = k / n_pad;
      if (t < KH * KW && n < N) v = (float)w[((long)n * Cin + row) * KH * KW + (KH * KW - 1 - t)];
    } else {
      const int c = row % Cin, t = row / Cin;
      if (k < N) v = (float)w[((long)k * Cin + c) * KH * KW + (mode == 3 ? KH * KW - 1 - t : t)];
    }
    dst[i] = (TD)v;
  }
}
}  // namespace

extern "C" int kpf_pack_conv_weight(const void* w, int src_dtype, void* dst, int dst_dtype, int N, int Cin, int KH, int KW, int mode, int n_pad, int Kp,
                                    void* stream) {
  KPF_REQUIRE(w && dst && N > 0 && Cin > 0 && KH > 0 && KW > 0 && mode >= 0 && mode <= 3 && n_pad >= N, "kpf_pack_conv_weight: bad arguments");
  const long rows = mode == 0 ? n_pad : (mode == 1 ? Cin : (long)KH * KW * Cin);
  const long kmin = mode == 0 ? (long)KH * KW * Cin : (mode == 1 ? (long)KH * KW * n_pad : n_pad);
  KPF_REQUIRE(Kp >= kmin, "kpf_pack_conv_weight: Kp = %d is shorter than the row (%ld)", Kp, kmin);
  const long total = rows * Kp;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(total)), block(256);
#define KPF_PACK(TS, TD)                                                                                                                              \
  hipLaunchKernelGGL((pack_weight_kernel<TS, TD>), grid, block, 0, st, static_cast<const TS*>(w), static_cast<TD*>(dst), N, Cin, KH, KW, mode, n_pad, Kp, \
                     total)
  if (src_dtype == KPF_DT_F32 && dst_dtype == KPF_DT_F32) KPF_PACK(float, float);
  else if (src_dtype == KPF_DT_F32 && dst_dtype == KPF_DT_BF16) KPF_PACK(float, bf16_t);
  else if (src_dtype == KPF_DT_F32 && dst_dtype == KPF_DT_F16) KPF_PACK(float, f16_t);
  else if (src_dtype == KPF_DT_BF16 && dst_dtype == KPF_DT_BF16) KPF_PACK(bf16_t, bf16_t);
  else if (src_dtype == KPF_DT_F16 && dst_dtype == KPF_DT_F16) KPF_PACK(f16_t, f16_t);
  else {
    kpf_set_error("kpf_pack_conv_weight: unsupported dtype pair %d -> %d", src_dtype, dst_dtype);
    return KPF_EINVAL;
  }
#undef KPF_PACK
  return kpf_check_launch("kpf_pack_conv_weight");
}

// ---------------------------------------------------------------------------------------------------------------
// The same packing for EVERY operand of the training step in one launch: a table of descriptors in device memory (built once by the
// host when the set of layers is known; sources are the parameters' own storage, destinations persistent operand buffers), refreshed
// at the start of each iteration — "weights kept in kernel layout across steps" without giving up the reference's state-dict layout.
// Workgroup b serves the descriptor whose [first_block, first_block + nblocks) range holds b (binary search, uniform per workgroup).
// ---------------------------------------------------------------------------------------------------------------
namespace {
template <typename TS>
__device__ __forceinline__ float pack_value(const TS* __restrict__ w, const kpf_pack_desc& d, int row, int k) {
  const int T = d.KH * d.KW;
  if (d.mode == 0) {
    const int c = k % d.Cin, t = k / d.Cin;
    return (t < T && row < d.N) ? (float)w[((long)row * d.Cin + c) * T + t] : 0.f;
  }
  if (d.mode == 1) {
    const int n = k % d.n_pad, t = k / d.n_pad;
    return (t < T && n < d.N) ? (float)w[((long)n * d.Cin + row) * T + (T - 1 - t)] : 0.f;
  }
  if (d.mode == 4) return k < d.N ? (float)w[k] : 0.f;  // a vector (bias) copied into a slot of a stacked operand: rows = 1, Kp = slot length
  const int c = row % d.Cin, t = row / d.Cin;
  return k < d.N ? (float)w[((long)k * d.Cin + c) * T + (d.mode == 3 ? T - 1 - t : t)] : 0.f;
}

// Which form of the refresh kernel serves a descriptor (ONE rule for the kernel and for the host's block count: kpf_pack_desc_blocks)
enum { PACK_GENERIC = 0, PACK_T_1X1 = 1, PACK_DGRAD_KXK = 4 };
__host__ __device__ inline int pack_form(const kpf_pack_desc& d) {
  const int T = d.KH * d.KW;
  if (d.src_dtype != KPF_DT_F32) return PACK_GENERIC;
  if (d.mode == 1 && T == 1 && (d.Cin & 3) == 0 && (d.Kp & 3) == 0) return PACK_T_1X1;
  // (LDS-staged forms of the k x k forward rows and of the depthwise tap tables were measured too: 26 -> 57 us and 31 -> 31 us — their element-per-thread
  //  reads already hit the same cache lines from neighbouring threads — and removed)
  if (d.mode == 1 && T > 1 && T <= 16 && d.reserved == 0) return PACK_DGRAD_KXK;
  return PACK_GENERIC;
}
__host__ __device__ inline long pack_blocks(const kpf_pack_desc& d) {
  switch (pack_form(d)) {
    case PACK_T_1X1: return (long)((d.rows + 63) / 64) * ((d.Kp + 63) / 64);
    case PACK_DGRAD_KXK: return (long)((d.Cin + 7) / 8) * ((d.n_pad + 31) / 32);
    default: return ((long)d.rows * d.Kp + 1023) / 1024;
  }
}
__device__ __forceinline__ void pack_store(const kpf_pack_desc& d, long i, float v) {
  if (d.dst_dtype == KPF_DT_F32) static_cast<float*>(d.dst)[i] = v;
  else if (d.dst_dtype == KPF_DT_BF16) static_cast<bf16_t*>(d.dst)[i] = (bf16_t)v;
  else static_cast<f16_t*>(d.dst)[i] = (f16_t)v;
}

__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const kpf_pack_desc* __restrict__ descs, int ndesc) {
  int lo = 0, hi = ndesc - 1;
  const int b = blockIdx.x;
  while (lo < hi) {  // last descriptor with first_block <= b
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= b) lo = mid;
    else hi = mid - 1;
  }
  __shared__ float pack_lds[64 * 65];  // ONE staging buffer for the two LDS-staged forms (16.6 KB: separate arrays per form cost every form its occupancy)
  const kpf_pack_desc d = descs[lo];
  // d.reserved != 0: the destination rows are `reserved` elements apart (> Kp) — this operand fills a column range [dst, dst + Kp) of a wider,
  // stacked matrix (the q | k | v data-gradient operand of training.SelfAttention21); only those Kp columns are written.
  const long dld = d.reserved ? d.reserved : d.Kp;
  if (pack_form(d) == PACK_T_1X1) {
    // 1x1 data-gradient operand = the transpose of the weight: 64 x 64 tiles through LDS, 16-byte reads of the weight's rows, 4-element (8- / 16-byte)
    // writes of the operand's rows (blocks of such a descriptor: ceil(rows / 64) * ceil(Kp / 64), see training.PackCache; round 4: 32 x 32 tiles
    // with scalar accesses before — 128 us of the refresh launch for the 260 operands of a ConvNeXt-T iteration)
    float (*tl)[65] = reinterpret_cast<float (*)[65]>(pack_lds);
    const int tiles_k = (d.Kp + 63) / 64;
    const int t = b - d.first_block, tr = t / tiles_k, tc = t - tr * tiles_k;
    const int q = threadIdx.x & 15, r16 = threadIdx.x >> 4;
    const float* w = static_cast<const float*>(d.src);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int n = tc * 64 + r16 + 16 * p, c = tr * 64 + 4 * q;  // read w[n][c .. c+3]
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < d.N && c < d.Cin) v = *reinterpret_cast<const f32x4*>(w + (long)n * d.Cin + c);
      tl[r16 + 16 * p][4 * q + 0] = v[0];
      tl[r16 + 16 * p][4 * q + 1] = v[1];
      tl[r16 + 16 * p][4 * q + 2] = v[2];
      tl[r16 + 16 * p][4 * q + 3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int c = tr * 64 + r16 + 16 * p, k = tc * 64 + 4 * q;  // write dst[c][k .. k+3], k = n
      if (c < d.rows && k < d.Kp) {
        const f32x4 v = {tl[4 * q + 0][r16 + 16 * p], tl[4 * q + 1][r16 + 16 * p], tl[4 * q + 2][r16 + 16 * p], tl[4 * q + 3][r16 + 16 * p]};
        const long i = (long)c * dld + k;
        if (d.dst_dtype == KPF_DT_F32) *reinterpret_cast<f32x4*>(static_cast<float*>(d.dst) + i) = v;
        else if (d.dst_dtype == KPF_DT_BF16) kpf_st4(static_cast<bf16_t*>(d.dst) + i, v);
        else kpf_st4(static_cast<f16_t*>(d.dst) + i, v);
      }
    }
    return;
  }
  if (d.mode == 0 && d.KH * d.KW == 1 && d.src_dtype == KPF_DT_F32 && (d.Cin & 3) == 0 && (d.Kp & 3) == 0 && d.reserved == 0) {
    // forward rows of a 1x1 layer = the weight's rows, zero-extended to Kp and rounded to the operand type: a quad of elements per thread, 16-byte loads,
    // 32-bit index arithmetic (272 operands / 58 M elements of a ConvNeXt-T iteration: 166 -> ~70 us of the refresh launch)
    const unsigned total = (unsigned)d.rows * (unsigned)d.Kp;
    const unsigned i = (unsigned)(b - d.first_block) * 1024u + 4u * threadIdx.x;
    if (i < total) {
      const unsigned row = i / (unsigned)d.Kp, k = i - row * (unsigned)d.Kp;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < (unsigned)d.N && k < (unsigned)d.Cin) v = *reinterpret_cast<const f32x4*>(static_cast<const float*>(d.src) + (long)row * d.Cin + k);
      if (d.dst_dtype == KPF_DT_F32) *reinterpret_cast<f32x4*>(static_cast<float*>(d.dst) + i) = v;
      else if (d.dst_dtype == KPF_DT_BF16) kpf_st4(static_cast<bf16_t*>(d.dst) + i, v);
      else kpf_st4(static_cast<f16_t*>(d.dst) + i, v);
    }
    return;
  }
  if (pack_form(d) == PACK_DGRAD_KXK) {
    const int T = d.KH * d.KW;
    // data-gradient rows of a k x k layer: dst[c][t'*n_pad + n] = w[n][c][T-1-t'].  A block = (8 input channels, 32 output channels): per output channel
    // 8*T contiguous source floats through LDS, then 8*T runs of 32 consecutive destination elements
    float (*sm)[8 * 16 + 1] = reinterpret_cast<float (*)[8 * 16 + 1]>(pack_lds);
    const int nchunks = (d.n_pad + 31) / 32;
    const int t = b - d.first_block, cc = t / nchunks, c0 = cc * 8, n0 = (t - cc * nchunks) * 32;
    const int cw = min(8, d.Cin - c0);
    const float* w = static_cast<const float*>(d.src);
    for (int i = threadIdx.x; i < 32 * cw * T; i += 256) {
      const int nl = i / (cw * T), r = i - nl * (cw * T);
      sm[nl][r] = (n0 + nl) < d.N ? w[((long)(n0 + nl) * d.Cin + c0) * T + r] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cw * T * 32; i += 256) {
      const int nl = i & 31, ct = i >> 5;  // ct = cl * T + t'
      const int cl = ct / T, tp = ct - cl * T;
      if (n0 + nl < d.n_pad) pack_store(d, (long)(c0 + cl) * dld + (long)tp * d.n_pad + n0 + nl, sm[nl][cl * T + (T - 1 - tp)]);
    }
    if (n0 == 0)
      for (int i = threadIdx.x; i < cw * (d.Kp - T * d.n_pad); i += 256) {
        const int cl = i / (d.Kp - T * d.n_pad), k = T * d.n_pad + (i - cl * (d.Kp - T * d.n_pad));
        pack_store(d, (long)(c0 + cl) * dld + k, 0.f);
      }
    return;
  }
  const long total = (long)d.rows * d.Kp;
  for (long i = (long)(b - d.first_block) * 1024 + threadIdx.x; i < total && i < (long)(b - d.first_block + 1) * 1024; i += 256) {
    const int k = (int)(i % d.Kp), row = (int)(i / d.Kp);
    float v;
    if (d.src_dtype == KPF_DT_F32) v = pack_value(static_cast<const float*>(d.src), d, row, k);
    else if (d.src_dtype == KPF_DT_BF16) v = pack_value(static_cast<const bf16_t*>(d.src), d, row, k);
    else v = pack_value(static_cast<const f16_t*>(d.src), d, row, k);
    const long o = (long)row * dld + k;
    if (d.dst_dtype == KPF_DT_F32) static_cast<float*>(d.dst)[o] = v;
    else if (d.dst_dtype == KPF_DT_BF16) static_cast<bf16_t*>(d.dst)[o] = (bf16_t)v;
    else static_cast<f16_t*>(d.dst)[o] = (f16_t)v;
  }
}
}  // namespace

/* workgroups a descriptor occupies in kpf_pack_conv_weights_multi (its first_block spacing): the host builds the table with this */
extern "C" long kpf_pack_desc_blocks(const kpf_pack_desc* d) { return d ? pack_blocks(*d) : 0; }

extern "C" int kpf_pack_conv_weights_multi(const kpf_pack_desc* descs_device, int ndesc, int total_blocks, void* stream) {
  KPF_REQUIRE(descs_device && ndesc > 0 && total_blocks > 0, "kpf_pack_conv_weights_multi: bad arguments");
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), descs_device, ndesc);
  return kpf_check_launch("kpf_pack_conv_weights_multi");
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over the last axis and GELU(erf), forward and backward, for the training step (convNeXT/convnext.py:43-46, 199-214; the
// post-LN BERT / decoder layers of the fusion head).  LayerNorm: one wave per row, a lane holds up to 4 channel quads (C <= 1024), two-pass
// statistics in registers (mean, then centred squares) like ATen; the backward's input gradient needs two more wave reductions per row,
// the parameter gradients are column sums over all rows: every workgroup adds its rows in registers (fixed row order), its four waves
// combine through LDS in wave order, and a second kernel adds the per-workgroup partials in workgroup order — no atomics, two launches
// (ATen: three).  x / dy / dx fp32; y in fp32 or the 16-bit storage type of the following GEMM (no separate cast pass).
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr int LN_MAXQ = 4;  // channel quads per lane: C <= 64 * 4 * LN_MAXQ = 1024

template <typename TY>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, TY* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long rows, int C4, float eps, int G) {
  // G > 1 (kpf_ln_train_forward_g): row r is normalised with parameter set r % G (w, b hold G sets of C) — channel-stacked groups seen as rows
  const int lane = threadIdx.x & 63;
  const float invC = 1.0f / (float)(4 * C4);
  const int po = (int)(((long)blockIdx.x * 4 + (threadIdx.x >> 6)) % G) * 4 * C4;  // (the row stride is a multiple of 4: one parameter set per wave)
  w += po, b += po;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
    f32x4 v[LN_MAXQ];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (q < C4) v[i] = kpf_ld4(x + r * C4 * 4 + 4 * q);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float m = wave_sum(s) * invC;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i)
      if (lane + 64 * i < C4)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[i][e] - m;
          sq = fmaf(d, d, sq);
        }
    const float rs = 1.0f / sqrtf(wave_sum(sq) * invC + eps);
    if (lane == 0) {
      mean[r] = m;
      rstd[r] = rs;
    }
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      if (q < C4) {
        const f32x4 g = kpf_ld4(w + 4 * q), be = kpf_ld4(b + 4 * q);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - m) * rs * g[e] + be[e];
        kpf_st4(y + r * C4 * 4 + 4 * q, o);
      }
    }
  }
}

// grid.x workgroups, each owning the rows r = blockIdx.x*4 + wave, + gridDim.x*4, ...  part: [gridDim.x][2][C] (dw | db)
template <typename TD>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TD* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ w, float* __restrict__ dx, float* __restrict__ part,
                                                     long rows, int C4, int G) {
  extern __shared__ float ln_lds[];  // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = 4 * C4;
  w += (wave % G) * C;  // G > 1: parameter set r % G per row; a wave's rows are 4 * gridDim.x apart, i.e. all of one set
  const float invC = 1.0f / (float)C;
  f32x4 aw[LN_MAXQ], ab[LN_MAXQ], g[LN_MAXQ];
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    aw[i] = ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    g[i] = lane + 64 * i < C4 ? kpf_ld4(w + 4 * (lane + 64 * i)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const float m = mean[r], rs = rstd[r];
    f32x4 xh[LN_MAXQ], d[LN_MAXQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      xh[i] = d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (q < C4) {
        const f32x4 xv = kpf_ld4(x + r * C + 4 * q);
        d[i] = kpf_ld4(dy + r * C + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xv[e] - m) * rs;
          const float gd = d[i][e] * g[i][e];
          s1 += gd;
          s2 = fmaf(gd, xh[i][e], s2);
          aw[i][e] = fmaf(d[i][e], xh[i][e], aw[i][e]);
          ab[i][e] += d[i][e];
        }
      }
    }
    const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      if (q < C4) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (d[i][e] * g[i][e] - m1 - xh[i][e] * m2);
        kpf_st4(dx + r * C + 4 * q, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    if (q < C4) {
      kpf_st4(ln_lds + (wave * 2 + 0) * C + 4 * q, aw[i]);
      kpf_st4(ln_lds + (wave * 2 + 1) * C + 4 * q, ab[i]);
    }
  }
  __syncthreads();
  const int GC = G * C;  // part: [gridDim.x][2][G][C]
  for (int i = threadIdx.x; i < 2 * GC; i += 256) {
    const int which = i / GC, cc = i - which * GC;
    const int grp = cc / C, c = cc - grp * C;
    float s = 0.f;
    for (int wv = grp; wv < 4; wv += G) s += ln_lds[(wv * 2 + which) * C + c];  // (G = 1: waves 0..3 in order)
    part[((long)blockIdx.x * 2 + which) * GC + cc] = s;
  }
}

// 64 columns x 8 segments per workgroup: segment g adds the partials of workgroups g*per .. (g+1)*per - 1 in order, the eight segment
// sums are added in segment order through LDS (fixed order, 8 x fewer dependent loads per thread than one thread per column)
// (tail: that many floats behind db's C are set to zero — the gradient of an embedding table of which only a prefix was used, kpf_colsum_desc::reserved)
__device__ __forceinline__ void ln_bwd_reduce_body(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int nblk, int C, int bx, int tail = 0) {
  __shared__ float seg[8][64];
  const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = bx * 64 + col;
  const int per = (nblk + 7) / 8;
  float s = 0.f;
  if (i < 2 * C) {
    const int which = i / C, c = i - which * C;
    const int b1 = min(nblk, (g + 1) * per);
    int bk = g * per;
    for (; bk + 4 <= b1; bk += 4) {  // four loads in flight, added in index order
      const float p0 = part[((long)bk * 2 + which) * C + c], p1 = part[((long)(bk + 1) * 2 + which) * C + c];
      const float p2 = part[((long)(bk + 2) * 2 + which) * C + c], p3 = part[((long)(bk + 3) * 2 + which) * C + c];
      s = (((s + p0) + p1) + p2) + p3;
    }
    for (; bk < b1; ++bk) s += part[((long)bk * 2 + which) * C + c];
  }
  seg[g][col] = s;
  __syncthreads();
  if (g == 0 && i < 2 * C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += seg[k][col];
    const int which = i / C, c = i - which * C;
    (which ? db : dw)[c] = t;
  } else if (g == 0 && i < 2 * C + tail) {
    db[i - C] = 0.f;
  }
}

__global__ __launch_bounds__(512) void ln_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int nblk, int C) {
  ln_bwd_reduce_body(part, dw, db, nblk, C, blockIdx.x);
}

// The same reduce for MANY layers in one launch (descriptors by value, read from the kernel-argument segment): a training iteration runs
// ~115 of these 5-us launches, one behind every LayerNorm / layer-scale backward; the _partial entry points skip it and describe it, and
// training.DeferredParamGrads issues them together after backward.  Same arithmetic per column: same bits.
struct ColsumBatch {
  kpf_colsum_desc d[KPF_COLSUM_BATCH];
  int nd;
};
typedef const __attribute__((address_space(4))) ColsumBatch* colsum_kernarg_t;

__global__ __launch_bounds__(512) void colsum_reduce_grouped_kernel(const ColsumBatch) {
  colsum_kernarg_t bp = (colsum_kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int nd = bp->nd;
  int lo = 0, hi = nd - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= bp->d[mid].first_block) lo = mid;
    else hi = mid - 1;
  }
  ln_bwd_reduce_body(bp->d[lo].part, bp->d[lo].dw, bp->d[lo].db, bp->d[lo].nblk, bp->d[lo].C, (int)blockIdx.x - bp->d[lo].first_block, bp->d[lo].reserved);
}

__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float v) {
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
  return cdf + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}

template <typename TA>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const TA* __restrict__ x, TA* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = kpf_ld4(x + 4 * i);
    kpf_st4(y + 4 * i, f32x4{gelu_exact(v[0]), gelu_exact(v[1]), gelu_exact(v[2]), gelu_exact(v[3])});
  }
}
template <typename TA>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const TA* __restrict__ dy, const TA* __restrict__ x, TA* __restrict__ dx, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = kpf_ld4(x + 4 * i), g = kpf_ld4(dy + 4 * i);
    kpf_st4(dx + 4 * i, f32x4{g[0] * gelu_grad(v[0]), g[1] * gelu_grad(v[1]), g[2] * gelu_grad(v[2]), g[3] * gelu_grad(v[3])});
  }
}

inline int ln_blocks(long rows) {  // one row per wave for the small (B x 21)-row tensors of the fusion head (a row is a dependent chain of two wave
  static const int rpw = []() { const char* e = getenv("KPF_LN_ROWS_PER_WAVE"); return e ? atoi(e) : 1; }();  // tuning aid (rows >= 2048 only)
  long g = (rows + 3) / 4;          // reductions: round 4 went from four rows per wave to one), at most LN_MAX_BLOCKS workgroups' partials to add
  if (rows >= 2048 && rpw > 1) g = (rows + 4 * rpw - 1) / (4 * rpw);
  // (round 5: 256 -> 1024.  With one workgroup per CU a wave walks rows / 1024 rows one dependent memory round trip at a time: the backward passes of the
  //  32768 x 192 / 8192 x 384 LayerNorms ran at a quarter of what their traffic takes; four workgroups per CU put four times as many rows in flight.)
  static const int cap = []() { const char* e = getenv("KPF_LN_MAX_BLOCKS"); return e ? atoi(e) : 1024; }();
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
}  // namespace

extern "C" long kpf_ln_ws_floats(long rows, int C) { return (long)ln_blocks(rows) * 2 * C; }

extern "C" int kpf_ln_train_forward_g(const float* x, const float* w, const float* b, void* y, int y_dtype, float* mean, float* rstd, long rows, int C, int G,
                                      float eps, void* stream) {
  KPF_REQUIRE(x && w && b && y && mean && rstd && rows > 0 && C > 0 && C % 4 == 0 && C <= 256 * LN_MAXQ, "kpf_ln_train_forward: bad arguments (C %% 4 == 0, C <= 1024)");
  KPF_REQUIRE((G == 1 || G == 2 || G == 4) && rows % G == 0, "kpf_ln_train_forward_g: G must be 1, 2 or 4 and divide the row count");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(rows, 4, 256 * 16));
  if (y_dtype == KPF_DT_F32) hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, dim3(256), 0, st, x, w, b, static_cast<float*>(y), mean, rstd, rows, C / 4, eps, G);
  else if (y_dtype == KPF_DT_BF16) hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, x, w, b, static_cast<bf16_t*>(y), mean, rstd, rows, C / 4, eps, G);
  else if (y_dtype == KPF_DT_F16) hipLaunchKernelGGL(ln_fwd_kernel<f16_t>, grid, dim3(256), 0, st, x, w, b, static_cast<f16_t*>(y), mean, rstd, rows, C / 4, eps, G);
  else {
    kpf_set_error("kpf_ln_train_forward: unknown y_dtype %d", y_dtype);
    return KPF_EINVAL;
  }
  return kpf_check_launch("kpf_ln_train_forward");
}

extern "C" int kpf_ln_train_forward(const float* x, const float* w, const float* b, void* y, int y_dtype, float* mean, float* rstd, long rows, int C, float eps,
                                    void* stream) {
  return kpf_ln_train_forward_g(x, w, b, y, y_dtype, mean, rstd, rows, C, 1, eps, stream);
}

static int ln_train_backward_impl(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* w, float* dx, float* dw,
                                  float* db, float* ws, long ws_floats, long rows, int C, void* stream, kpf_colsum_desc* defer, int G = 1) {
  // G > 1: w, dw, db hold G parameter sets of C; row r belongs to set r % G
  KPF_REQUIRE(dy && x && mean && rstd && w && dx && dw && db && ws && rows > 0 && C > 0 && C % 4 == 0 && C <= 256 * LN_MAXQ, "kpf_ln_train_backward: bad arguments");
  KPF_REQUIRE((G == 1 || G == 2 || G == 4) && rows % G == 0, "kpf_ln_train_backward_g: G must be 1, 2 or 4 and divide the row count");
  const int nblk = ln_blocks(rows);
  KPF_REQUIRE(ws_floats >= (long)nblk * 2 * C * G, "kpf_ln_train_backward: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = (size_t)8 * C * sizeof(float);
  if (dy_dtype == KPF_DT_F32) hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(nblk), dim3(256), lds, st, static_cast<const float*>(dy), x, mean, rstd, w, dx, ws, rows, C / 4, G);
  else if (dy_dtype == KPF_DT_BF16) hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), lds, st, static_cast<const bf16_t*>(dy), x, mean, rstd, w, dx, ws, rows, C / 4, G);
  else if (dy_dtype == KPF_DT_F16) hipLaunchKernelGGL(ln_bwd_kernel<f16_t>, dim3(nblk), dim3(256), lds, st, static_cast<const f16_t*>(dy), x, mean, rstd, w, dx, ws, rows, C / 4, G);
  else {
    kpf_set_error("kpf_ln_train_backward: unknown dy_dtype %d", dy_dtype);
    return KPF_EINVAL;
  }
  int rc = kpf_check_launch("kpf_ln_train_backward");
  if (rc != KPF_OK) return rc;
  if (defer) {
    defer->part = ws, defer->dw = dw, defer->db = db, defer->nblk = nblk, defer->C = G * C, defer->first_block = 0, defer->reserved = 0;
    return KPF_OK;
  }
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * G * C + 63) / 64), dim3(512), 0, st, ws, dw, db, nblk, G * C);
  return kpf_check_launch("kpf_ln_train_backward (reduce)");
}

extern "C" int kpf_ln_train_backward_g(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* w, float* dx, float* dw,
                                       float* db, float* ws, long ws_floats, long rows, int C, int G, kpf_colsum_desc* desc, void* stream) {
  return ln_train_backward_impl(dy, dy_dtype, x, mean, rstd, w, dx, dw, db, ws, ws_floats, rows, C, stream, desc, G);
}

extern "C" int kpf_ln_train_backward(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* w, float* dx, float* dw,
                                     float* db, float* ws, long ws_floats, long rows, int C, void* stream) {
  return ln_train_backward_impl(dy, dy_dtype, x, mean, rstd, w, dx, dw, db, ws, ws_floats, rows, C, stream, nullptr);
}

extern "C" int kpf_ln_train_backward_partial(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* w, float* dx, float* dw,
                                             float* db, float* ws, long ws_floats, long rows, int C, kpf_colsum_desc* desc, void* stream) {
  KPF_REQUIRE(desc, "kpf_ln_train_backward_partial: desc missing");
  return ln_train_backward_impl(dy, dy_dtype, x, mean, rstd, w, dx, dw, db, ws, ws_floats, rows, C, stream, desc);
}

extern "C" int kpf_colsum_reduce_grouped(const kpf_colsum_desc* descs, int n, void* stream) {
  KPF_REQUIRE(n >= 0 && (descs || n == 0), "kpf_colsum_reduce_grouped: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int base = 0; base < n; base += KPF_COLSUM_BATCH) {
    ColsumBatch b;
    b.nd = n - base < KPF_COLSUM_BATCH ? n - base : KPF_COLSUM_BATCH;
    long blocks = 0;
    for (int k = 0; k < b.nd; ++k) {
      b.d[k] = descs[base + k];
      KPF_REQUIRE(b.d[k].part && b.d[k].dw && b.d[k].db && b.d[k].nblk > 0 && b.d[k].C > 0 && b.d[k].reserved >= 0, "kpf_colsum_reduce_grouped: bad descriptor %d", base + k);
      b.d[k].first_block = (int)blocks;
      blocks += (2 * b.d[k].C + b.d[k].reserved + 63) / 64;
    }
    hipLaunchKernelGGL(colsum_reduce_grouped_kernel, dim3((unsigned)blocks), dim3(512), 0, st, b);
    const int rc = kpf_check_launch("kpf_colsum_reduce_grouped");
    if (rc != KPF_OK) return rc;
  }
  return KPF_OK;
}

namespace {
template <typename T>
int gelu_fwd_launch(const void* x, void* y, long n, void* stream) {
  hipLaunchKernelGGL(gelu_fwd_kernel<T>, dim3(grid_for(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), static_cast<const T*>(x), static_cast<T*>(y),
                     n / 4);
  return kpf_check_launch("kpf_gelu_forward");
}
template <typename T>
int gelu_bwd_launch(const void* dy, const void* x, void* dx, long n, void* stream) {
  hipLaunchKernelGGL(gelu_bwd_kernel<T>, dim3(grid_for(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), static_cast<const T*>(dy),
                     static_cast<const T*>(x), static_cast<T*>(dx), n / 4);
  return kpf_check_launch("kpf_gelu_backward");
}
}  // namespace

extern "C" int kpf_gelu_forward(const void* x, void* y, int dtype, long n, void* stream) {
  KPF_REQUIRE(x && y && n > 0 && n % 4 == 0, "kpf_gelu_forward: bad arguments (n %% 4 == 0)");
#define CALL(T) gelu_fwd_launch<T>(x, y, n, stream)
  KPF_DISPATCH_DT(dtype, "kpf_gelu_forward", CALL);
#undef CALL
}

extern "C" int kpf_gelu_backward(const void* dy, const void* x, void* dx, int dtype, long n, void* stream) {
  KPF_REQUIRE(dy && x && dx && n > 0 && n % 4 == 0, "kpf_gelu_backward: bad arguments (n %% 4 == 0)");
#define CALL(T) gelu_bwd_launch<T>(dy, x, dx, n, stream)
  KPF_DISPATCH_DT(dtype, "kpf_gelu_backward", CALL);
#undef CALL
}

// ---------------------------------------------------------------------------------------------------------------
// The attention core of the 21-token stacks in train mode (BertSelfAttention, model/model.py:30-126 via transformers; the decoder
// layer's multi_head_attention_forward, model/transfusion_head.py:527-546): per (sample, head) S = scale * Q K^T, P = softmax(S),
// P' = dropout(P), ctx = P' V, and its backward — one wave per (sample, head), everything in LDS / registers.  Q, K, V are read where
// the projections left them ([B*T][C] rows, head h = channels h*hd .. h*hd+hd-1), ctx is written in the layout the output projection
// reads: no head transposes, no batched-GEMM launches for 21 x 21 x 32 problems (library path: ~15 launches per layer).
// Dropout: keep = hash(seed, counter, call, element) >= p * 2^32 with a counter the host advances once per forward (device-resident, so
// a captured iteration draws new masks at every replay); the mask is kept for the backward (one byte per probability).
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr int AT_T = 21, AT_HD = 32;
constexpr int AT_NT = 256;  // threads per (sample, head): every phase is a loop over 441 or 672 independent outputs, each computed by ONE thread in the same
                            // sequential order whatever the thread count (round 4: 64 -> 256 threads; the kernels are latency chains of five phases)

__device__ __forceinline__ unsigned hash32(unsigned x) {  // "lowbias32" integer hash: full avalanche in three multiplies
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(AT_NT) void attn21_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ ctx,
                                                        float* __restrict__ P, unsigned char* __restrict__ M, int H, int ld, float scale, float p_drop,
                                                        const long* __restrict__ rng, int call_id, int ldc) {
  __shared__ float sq[AT_T][AT_HD + 1], sk[AT_T][AT_HD + 1], sv[AT_T][AT_HD + 1], sp[AT_T][AT_T + 1];
  const int b = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
  const long row0 = (long)b * AT_T;
  for (int i = lane; i < AT_T * AT_HD; i += AT_NT) {
    const int t = i / AT_HD, d = i % AT_HD;
    const long off = (row0 + t) * ld + h * AT_HD + d;
    sq[t][d] = q[off];
    sk[t][d] = k[off];
    sv[t][d] = v[off];
  }
  __syncthreads();
  for (int e = lane; e < AT_T * AT_T; e += AT_NT) {
    const int i = e / AT_T, j = e % AT_T;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < AT_HD; ++d) s = fmaf(sq[i][d], sk[j][d], s);
    sp[i][j] = s * scale;
  }
  __syncthreads();
  if (lane < AT_T) {  // row softmax
    float mx = -INFINITY;
    for (int j = 0; j < AT_T; ++j) mx = fmaxf(mx, sp[lane][j]);
    float sum = 0.f;
    for (int j = 0; j < AT_T; ++j) {
      const float e = __expf(sp[lane][j] - mx);
      sp[lane][j] = e;
      sum += e;
    }
    const float inv = 1.0f / sum;
    for (int j = 0; j < AT_T; ++j) sp[lane][j] *= inv;
  }
  __syncthreads();
  const long pbase = (long)blockIdx.x * AT_T * AT_T;
  const float keep_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  const unsigned thr = p_drop > 0.f ? (unsigned)fminf(p_drop * 4294967296.0f, 4294967295.0f) : 0u;
  const unsigned seed = rng ? (unsigned)rng[0] : 0u, ctr = rng ? (unsigned)rng[1] : 0u;
  for (int e = lane; e < AT_T * AT_T; e += AT_NT) {
    const int i = e / AT_T, j = e % AT_T;
    const float pv = sp[i][j];
    P[pbase + e] = pv;
    unsigned char keep = 1;
    if (p_drop > 0.f) keep = hash32(hash32(seed ^ (ctr * 0x9e3779b9U)) ^ hash32((unsigned)call_id * 0x85ebca6bU + (unsigned)(pbase + e))) >= thr;
    M[pbase + e] = keep;
    sp[i][j] = keep ? pv * keep_scale : 0.f;
  }
  __syncthreads();
  for (int e = lane; e < AT_T * AT_HD; e += AT_NT) {
    const int i = e / AT_HD, d = e % AT_HD;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < AT_T; ++j) s = fmaf(sp[i][j], sv[j][d], s);
    ctx[(row0 + i) * ldc + h * AT_HD + d] = s;  // (ldc: the context's own row stride — q / k / v may be column slices of one [rows][3C] projection)
  }
}

__global__ __launch_bounds__(AT_NT) void attn21_bwd_kernel(const float* __restrict__ dctx, const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ P, const unsigned char* __restrict__ M,
                                                        float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv, int H, int ld, float scale,
                                                        float p_drop, int ldc) {
  __shared__ float sq[AT_T][AT_HD + 1], sk[AT_T][AT_HD + 1], sv[AT_T][AT_HD + 1], sg[AT_T][AT_HD + 1], sp[AT_T][AT_T + 1], sd[AT_T][AT_T + 1];
  const int b = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
  const long row0 = (long)b * AT_T;
  for (int i = lane; i < AT_T * AT_HD; i += AT_NT) {
    const int t = i / AT_HD, d = i % AT_HD;
    const long off = (row0 + t) * ld + h * AT_HD + d;
    sq[t][d] = q[off];
    sk[t][d] = k[off];
    sv[t][d] = v[off];
    sg[t][d] = dctx[(row0 + t) * ldc + h * AT_HD + d];
  }
  const long pbase = (long)blockIdx.x * AT_T * AT_T;
  const float keep_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  __syncthreads();
  // dP' = dctx V^T ; dP = dP' * mask ; sd <- dP, sp <- P' (dropped probabilities, for dV)
  for (int e = lane; e < AT_T * AT_T; e += AT_NT) {
    const int i = e / AT_T, j = e % AT_T;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < AT_HD; ++d) s = fmaf(sg[i][d], sv[j][d], s);
    const float m = M[pbase + e] ? keep_scale : 0.f;
    sd[i][j] = s * m;
    sp[i][j] = P[pbase + e];
  }
  __syncthreads();
  // dV[j][d] = sum_i P'[i][j] dctx[i][d]
  for (int e = lane; e < AT_T * AT_HD; e += AT_NT) {
    const int j = e / AT_HD, d = e % AT_HD;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < AT_T; ++i) s = fmaf(sp[i][j] * (M[pbase + i * AT_T + j] ? keep_scale : 0.f), sg[i][d], s);
    dv[(row0 + j) * ld + h * AT_HD + d] = s;
  }
  __syncthreads();
  if (lane < AT_T) {  // dS = P * (dP - sum_j dP * P), scaled
    float dot = 0.f;
    for (int j = 0; j < AT_T; ++j) dot = fmaf(sd[lane][j], sp[lane][j], dot);
    for (int j = 0; j < AT_T; ++j) sd[lane][j] = sp[lane][j] * (sd[lane][j] - dot) * scale;
  }
  __syncthreads();
  for (int e = lane; e < AT_T * AT_HD; e += AT_NT) {
    const int i = e / AT_HD, d = e % AT_HD;
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int j = 0; j < AT_T; ++j) {
      a = fmaf(sd[i][j], sk[j][d], a);  // dQ[i] = sum_j dS[i][j] K[j]
      c = fmaf(sd[j][i], sq[j][d], c);  // dK[i] = sum_j dS[j][i] Q[j]
    }
    dq[(row0 + i) * ld + h * AT_HD + d] = a;
    dk[(row0 + i) * ld + h * AT_HD + d] = c;
  }
}
}  // namespace

extern "C" int kpf_attn21_forward(const float* q, const float* k, const float* v, float* ctx, float* P, unsigned char* M, int B, int T, int H, int hd, int ld,
                                  float scale, float p_drop, const long* rng, int call_id, void* stream) {
  KPF_REQUIRE(q && k && v && ctx && P && M && B > 0 && T == AT_T && hd == AT_HD && H > 0 && ld >= H * hd, "kpf_attn21_forward: needs 21 tokens and 32-wide heads");
  KPF_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng), "kpf_attn21_forward: dropout needs 0 <= p < 1 and the rng state");
  hipLaunchKernelGGL(attn21_fwd_kernel, dim3(B * H), dim3(AT_NT), 0, reinterpret_cast<hipStream_t>(stream), q, k, v, ctx, P, M, H, ld, scale, p_drop, rng, call_id, ld);
  return kpf_check_launch("kpf_attn21_forward");
}

/* q / k / v (and dq / dk / dv) with row stride ld, ctx (and dctx) with its own row stride ldc: the three projections as column slices of ONE
 * [rows][3C] GEMM output (training.SelfAttention21: one projection launch, one data-gradient launch per layer instead of three each). */
extern "C" int kpf_attn21_forward_ld(const float* q, const float* k, const float* v, float* ctx, float* P, unsigned char* M, int B, int T, int H, int hd, int ld,
                                     int ldc, float scale, float p_drop, const long* rng, int call_id, void* stream) {
  KPF_REQUIRE(q && k && v && ctx && P && M && B > 0 && T == AT_T && hd == AT_HD && H > 0 && ld >= H * hd && ldc >= H * hd, "kpf_attn21_forward_ld: needs 21 tokens and 32-wide heads");
  KPF_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng), "kpf_attn21_forward_ld: dropout needs 0 <= p < 1 and the rng state");
  hipLaunchKernelGGL(attn21_fwd_kernel, dim3(B * H), dim3(AT_NT), 0, reinterpret_cast<hipStream_t>(stream), q, k, v, ctx, P, M, H, ld, scale, p_drop, rng, call_id, ldc);
  return kpf_check_launch("kpf_attn21_forward_ld");
}
extern "C" int kpf_attn21_backward_ld(const float* dctx, const float* q, const float* k, const float* v, const float* P, const unsigned char* M, float* dq, float* dk,
                                      float* dv, int B, int T, int H, int hd, int ld, int ldc, float scale, float p_drop, void* stream) {
  KPF_REQUIRE(dctx && q && k && v && P && M && dq && dk && dv && B > 0 && T == AT_T && hd == AT_HD && H > 0 && ld >= H * hd && ldc >= H * hd, "kpf_attn21_backward_ld: bad arguments");
  hipLaunchKernelGGL(attn21_bwd_kernel, dim3(B * H), dim3(AT_NT), 0, reinterpret_cast<hipStream_t>(stream), dctx, q, k, v, P, M, dq, dk, dv, H, ld, scale, p_drop, ldc);
  return kpf_check_launch("kpf_attn21_backward_ld");
}

extern "C" int kpf_attn21_backward(const float* dctx, const float* q, const float* k, const float* v, const float* P, const unsigned char* M, float* dq, float* dk,
                                   float* dv, int B, int T, int H, int hd, int ld, float scale, float p_drop, void* stream) {
  KPF_REQUIRE(dctx && q && k && v && P && M && dq && dk && dv && B > 0 && T == AT_T && hd == AT_HD && H > 0 && ld >= H * hd, "kpf_attn21_backward: bad arguments");
  hipLaunchKernelGGL(attn21_bwd_kernel, dim3(B * H), dim3(AT_NT), 0, reinterpret_cast<hipStream_t>(stream), dctx, q, k, v, P, M, dq, dk, dv, H, ld, scale, p_drop, ld);
  return kpf_check_launch("kpf_attn21_backward");
}

// ---------------------------------------------------------------------------------------------------------------
// dX[b] (P x C) = A[b]^T (P x 21) @ dOut[b] (21 x C): the gradient of the per-sample products with 21 rows (softmax pooling of the point
// features, model/model.py:318-320; the gated F^2 -> 1 reduction in GEMM form, model/model.py:336-341) with respect to their P x C
// operand.  A K = 21 batched GEMM for which the library picks a 256 x 16 macro tile (443 us per call at B = 32, P = 1024, C = 128);
// here it is 21 multiply-adds per output with dOut[b] in LDS (the other two products of the pair stay on the library: 10-12 us).
// ---------------------------------------------------------------------------------------------------------------
namespace {
// (round 6) workgroup = 32 rows p x all channels; A[b][:, p0 .. p0+31] and dOut[b] in LDS; thread = (4 rows, channel quad): per joint ONE 16-byte LDS read of
// four A values (a broadcast across the 32 quad lanes) and one of dOut feed 16 multiply-adds.  The first form read A from global memory once per output and
// joint (21 load instructions per output quad): 30 us per call at B = 32, P = 1024, C = 128 for 17 MB of results; this one is bound by its stores.
__global__ __launch_bounds__(256) void bmm21_dx_kernel(const float* __restrict__ A, const float* __restrict__ dOut, float* __restrict__ dX, int J, int P, int C) {
  extern __shared__ __attribute__((aligned(16))) float sd[];  // dOut[b]: [J][C], then A tile [J][32]
  float* as = sd + J * C;
  const int b = blockIdx.y, p0 = blockIdx.x * 32;
  for (int i = threadIdx.x; i < J * C; i += 256) sd[i] = dOut[(long)b * J * C + i];
  const float* Ab = A + (long)b * J * P;
  for (int i = threadIdx.x; i < J * 32; i += 256) {
    const int j = i >> 5, pp = i & 31;
    as[i] = p0 + pp < P ? Ab[(long)j * P + p0 + pp] : 0.f;
  }
  __syncthreads();
  const int C4 = C >> 2;
  const int pg = threadIdx.x >> 5, ql = threadIdx.x & 31;
  for (int q = ql; q < C4; q += 32) {
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < J; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(as + j * 32 + 4 * pg);
      const f32x4 d = *reinterpret_cast<const f32x4*>(sd + j * C + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r][e] = fmaf(a[r], d[e], acc[r][e]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pr = p0 + 4 * pg + r;
      if (pr < P) *reinterpret_cast<f32x4*>(dX + ((long)b * P + pr) * C + 4 * q) = acc[r];
    }
  }
}
}  // namespace

extern "C" int kpf_bmm_small_k_dx(const float* A, const float* dOut, float* dX, int B, int J, int P, int C, void* stream) {
  KPF_REQUIRE(A && dOut && dX && B > 0 && J > 0 && J <= 64 && P > 0 && C > 0 && C % 4 == 0 && (long)J * (C + 32) * 4 <= 64 * 1024 && kpf_aligned16(dOut) && kpf_aligned16(dX),
              "kpf_bmm_small_k_dx: bad arguments");
  hipLaunchKernelGGL(bmm21_dx_kernel, dim3((P + 31) / 32, B), dim3(256), (size_t)J * (C + 32) * sizeof(float), reinterpret_cast<hipStream_t>(stream), A, dOut, dX, J, P, C);
  return kpf_check_launch("kpf_bmm_small_k_dx");
}

// ---------------------------------------------------------------------------------------------------------------
// The other two products of the same pair (round 5: they sat on the library's batched GEMM): out[b] = A[b] (J x P) @ X[b] (P x C), and
// dA[b] = dOut[b] (J x C) @ X[b]^T.  Few rows (J = 21), so neither is a matrix-core problem: the forward is a 1024-long reduction per
// output with X[b] read exactly once per channel block, the A-gradient a C-long dot product per output with dOut[b] in LDS.  Fixed
// summation orders (per-lane strided partial sums, xor-shuffle tree, waves combined in order): run-to-run deterministic.
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr int BMM_J = 24;  // accumulator rows held per thread (J <= 24)
// grid (C / 32, B), 1024 threads = 8 channel quads x 128 row lanes; thread (q, pl) adds rows p = pl, pl + 128, ... for all J outputs of its quad (the loop is a
// chain of dependent row loads: 8 trips at 1024 threads, 43 -> ~12 us against 32 trips at 256)
constexpr int BMM_NT = 1024, BMM_PL = BMM_NT / 8, BMM_NW = BMM_NT / 64;
__global__ __launch_bounds__(BMM_NT) void bmm21_fwd_kernel(const float* __restrict__ A, const float* __restrict__ X, float* __restrict__ out, int J, int P, int C) {
  __shared__ float red[BMM_NW][BMM_J][32];
  const int b = blockIdx.y, c0 = blockIdx.x * 32;
  const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const float* Ab = A + (long)b * J * P;
  const float* Xb = X + (long)b * P * C + c0 + 4 * q;
  f32x4 acc[BMM_J];
#pragma unroll
  for (int j = 0; j < BMM_J; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool cv = c0 + 4 * q < C;  // (C % 4 == 0: a quad is inside or outside)
  // branch-free inner loop: the row index is clamped (accumulators beyond J collect a copy of row J - 1 and are never stored) — a guarded load per
  // joint compiled into a chain of 24 branches, each exposing its load's latency: 175 us per call where this form takes ~10
  for (int p = pl; p < P; p += BMM_PL) {
    const f32x4 x = cv ? *reinterpret_cast<const f32x4*>(Xb + (long)p * C) : f32x4{0.f, 0.f, 0.f, 0.f};
    float av[BMM_J];
#pragma unroll
    for (int j = 0; j < BMM_J; ++j) av[j] = Ab[(j < J ? j : J - 1) * P + p];
#pragma unroll
    for (int j = 0; j < BMM_J; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(av[j], x[e], acc[j][e]);
  }
  // the 8 row lanes of a wave (lane bits 3..5), then the 16 waves through LDS, in a fixed order
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < BMM_J; ++j)
    if (j < J) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[j][e];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        acc[j][e] = v;
      }
      if (lane < 8) *reinterpret_cast<f32x4*>(&red[wave][j][4 * lane]) = acc[j];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < J * 32; i += BMM_NT) {
    const int j = i >> 5, c = i & 31;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < BMM_NW; w += 4) sum += (red[w][j][c] + red[w + 1][j][c]) + (red[w + 2][j][c] + red[w + 3][j][c]);
    if (c0 + c < C) out[((long)b * J + j) * C + c0 + c] = sum;
  }
}
// grid (ceil(P / 256), B), 256 threads: thread = one row p of X[b]; dOut[b] ([J][C]) in LDS, the row streamed once in float4 pieces
__global__ __launch_bounds__(256) void bmm21_da_kernel(const float* __restrict__ dOut, const float* __restrict__ X, float* __restrict__ dA, int J, int P, int C) {
  extern __shared__ float sd[];  // dOut[b]: [J][C]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < J * C; i += 256) sd[i] = dOut[(long)b * J * C + i];
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const float* xr = X + ((long)b * P + p) * C;
  float acc[BMM_J];
#pragma unroll
  for (int j = 0; j < BMM_J; ++j) acc[j] = 0.f;
  for (int c = 0; c < C; c += 4) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
    for (int j = 0; j < BMM_J; ++j) {  // (row index clamped instead of a guard per joint: rows beyond J repeat row J - 1 and are not stored)
      const f32x4 d = *reinterpret_cast<const f32x4*>(sd + (j < J ? j : J - 1) * C + c);  // (same address across the wave: an LDS broadcast)
      acc[j] = fmaf(x[3], d[3], fmaf(x[2], d[2], fmaf(x[1], d[1], fmaf(x[0], d[0], acc[j]))));
    }
  }
#pragma unroll
  for (int j = 0; j < BMM_J; ++j)
    if (j < J) dA[((long)b * J + j) * P + p] = acc[j];
}
}  // namespace

namespace {
// (round 6) the same product with FOUR rows p per thread and trip: A[b][j][4 pl .. 4 pl + 3] is one 16-byte load, so a trip is 24 + 4 load instructions for
// 384 multiply-adds (the form above: 24 + 1 for 96 — the address unit, not the memory, bounded it: 32 us per call at B = 32, P = 1024, C = 128 for 17 MB of
// X).  grid (ceil(C / 16), B), 512 threads = 4 channel quads x 128 row lanes (256 workgroups at C = 128: every CU busy).  P % 4 == 0.
constexpr int BMM4_NT = 512, BMM4_NW = BMM4_NT / 64;
__global__ __launch_bounds__(BMM4_NT) void bmm21_fwd4_kernel(const float* __restrict__ A, const float* __restrict__ X, float* __restrict__ out, int J, int P, int C) {
  __shared__ float red[BMM4_NW][BMM_J][16];
  const int b = blockIdx.y, c0 = blockIdx.x * 16;
  const int q = threadIdx.x & 3, pl = threadIdx.x >> 2;
  const float* Ab = A + (long)b * J * P;
  const float* Xb = X + (long)b * P * C + c0 + 4 * q;
  f32x4 acc[BMM_J];
#pragma unroll
  for (int j = 0; j < BMM_J; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool cv = c0 + 4 * q < C;
  for (int p = 4 * pl; p < P; p += 4 * (BMM4_NT / 4)) {
    f32x4 x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = cv ? *reinterpret_cast<const f32x4*>(Xb + (long)(p + r) * C) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < BMM_J; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Ab + (long)(j < J ? j : J - 1) * P + p);  // (rows beyond J repeat row J - 1 and are never stored)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(a[r], x[r][e], acc[j][e]);
    }
  }
  // the 16 row lanes of a wave (lane bits 2..5), then the 8 waves through LDS, in a fixed order
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < BMM_J; ++j)
    if (j < J) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[j][e];
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        acc[j][e] = v;
      }
      if (lane < 4) *reinterpret_cast<f32x4*>(&red[wave][j][4 * lane]) = acc[j];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < J * 16; i += BMM4_NT) {
    const int j = i >> 4, c = i & 15;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < BMM4_NW; w += 4) sum += (red[w][j][c] + red[w + 1][j][c]) + (red[w + 2][j][c] + red[w + 3][j][c]);
    if (c0 + c < C) out[((long)b * J + j) * C + c0 + c] = sum;
  }
}

// (round 6) dA with X[b]'s rows staged: workgroup = 32 rows p; the tile X[b][p0 .. p0 + 31][:] goes to LDS with coalesced 16-byte loads (the form above has
// every lane walk its own 512-byte row: 64 cache lines per load instruction), dOut[b] beside it; thread = (row p, three joints): per channel quad one LDS read
// of x and three broadcast reads of dOut feed 12 multiply-adds.  LDS: (32 (C + 4) + J C) floats.
__global__ __launch_bounds__(256) void bmm21_da_lds_kernel(const float* __restrict__ dOut, const float* __restrict__ X, float* __restrict__ dA, int J, int P, int C) {
  extern __shared__ __attribute__((aligned(16))) float sd[];  // dOut[b]: [J][C] (rows beyond J: zero up to 24), then the X tile [32][C + 4]
  const int b = blockIdx.y, p0 = blockIdx.x * 32;
  const int C4 = C >> 2, ldx = C + 4;
  float* xs = sd + BMM_J * C;
  for (int i = threadIdx.x; i < BMM_J * C; i += 256) sd[i] = i < J * C ? dOut[(long)b * J * C + i] : 0.f;
  for (int i = threadIdx.x; i < 32 * C4; i += 256) {
    const int pp = i / C4, q = i - pp * C4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p0 + pp < P) v = *reinterpret_cast<const f32x4*>(X + ((long)b * P + p0 + pp) * C + 4 * q);
    *reinterpret_cast<f32x4*>(xs + pp * ldx + 4 * q) = v;
  }
  __syncthreads();
  const int pp = threadIdx.x & 31, jg = threadIdx.x >> 5;
  float acc[3] = {0.f, 0.f, 0.f};
  const float* xr = xs + pp * ldx;
  const float* dr = sd + 3 * jg * C;
  for (int q = 0; q < C4; ++q) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(xr + 4 * q);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(dr + k * C + 4 * q);
      acc[k] = fmaf(x[3], d[3], fmaf(x[2], d[2], fmaf(x[1], d[1], fmaf(x[0], d[0], acc[k]))));
    }
  }
  if (p0 + pp < P) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (3 * jg + k < J) dA[((long)b * J + 3 * jg + k) * P + p0 + pp] = acc[k];
  }
}
}  // namespace

extern "C" int kpf_bmm_small_k_fwd(const float* A, const float* X, float* out, int B, int J, int P, int C, void* stream) {
  KPF_REQUIRE(A && X && out && B > 0 && J > 0 && J <= BMM_J && P > 0 && C > 0 && C % 4 == 0 && kpf_aligned16(X), "kpf_bmm_small_k_fwd: bad arguments (J <= 24, C %% 4 == 0)");
  if (P % 4 == 0 && kpf_aligned16(A))
    hipLaunchKernelGGL(bmm21_fwd4_kernel, dim3((C + 15) / 16, B), dim3(BMM4_NT), 0, reinterpret_cast<hipStream_t>(stream), A, X, out, J, P, C);
  else
    hipLaunchKernelGGL(bmm21_fwd_kernel, dim3((C + 31) / 32, B), dim3(BMM_NT), 0, reinterpret_cast<hipStream_t>(stream), A, X, out, J, P, C);
  return kpf_check_launch("kpf_bmm_small_k_fwd");
}

extern "C" int kpf_bmm_small_k_da(const float* dOut, const float* X, float* dA, int B, int J, int P, int C, void* stream) {
  KPF_REQUIRE(dOut && X && dA && B > 0 && J > 0 && J <= BMM_J && P > 0 && C > 0 && C % 4 == 0 && (long)J * C * 4 <= 64 * 1024 && kpf_aligned16(X) && kpf_aligned16(dOut),
              "kpf_bmm_small_k_da: bad arguments (J <= 24, C %% 4 == 0)");
  const size_t lds = ((size_t)BMM_J * C + 32 * (size_t)(C + 4)) * sizeof(float);
  if (lds <= 48 * 1024)  // (C <= 216 at J <= 24: the X tile fits beside dOut)
    hipLaunchKernelGGL(bmm21_da_lds_kernel, dim3((P + 31) / 32, B), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), dOut, X, dA, J, P, C);
  else
    hipLaunchKernelGGL(bmm21_da_kernel, dim3((P + 255) / 256, B), dim3(256), (size_t)J * C * sizeof(float), reinterpret_cast<hipStream_t>(stream), dOut, X, dA, J, P, C);
  return kpf_check_launch("kpf_bmm_small_k_da");
}

// ---------------------------------------------------------------------------------------------------------------
// Layer scale + residual of the ConvNeXt block, out = x + gamma * y (convNeXT/convnext.py:48-51), forward and backward.  x, out and the
// incoming gradient are fp32 (the residual stream), y / dy are in the GEMM's storage type (fp32, or 16-bit under mixed precision: the
// torch expression needs a cast on either side).  dgamma = column sums of g * y: per-workgroup partial sums in registers (a lane owns up
// to 4 channel quads, rows walked in a fixed order), four waves combined through LDS, then the LayerNorm backward's fixed-order reduce.
// ---------------------------------------------------------------------------------------------------------------
namespace {
template <typename TY>
__global__ __launch_bounds__(256) void layer_scale_fwd_kernel(const float* __restrict__ x, const TY* __restrict__ y, const float* __restrict__ gamma,
                                                              float* __restrict__ out, long rows, int C4) {
  const long total = rows * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % C4);
    const f32x4 xv = kpf_ld4(x + 4 * i), yv = kpf_ld4(y + 4 * i), g = kpf_ld4(gamma + 4 * q);
    kpf_st4(out + 4 * i, f32x4{fmaf(g[0], yv[0], xv[0]), fmaf(g[1], yv[1], xv[1]), fmaf(g[2], yv[2], xv[2]), fmaf(g[3], yv[3], xv[3])});
  }
}

template <typename TY>
__global__ __launch_bounds__(256) void layer_scale_bwd_kernel(const float* __restrict__ g, const TY* __restrict__ y, const float* __restrict__ gamma,
                                                              TY* __restrict__ dy, float* __restrict__ part, long rows, int C4, int G) {
  extern __shared__ float ls_lds[];  // [4 waves][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = 4 * C4;
  gamma += (wave % G) * C;  // G > 1 (kpf_layer_scale_backward_g): row r uses parameter set r % G; a wave's rows are all of one set (see ln_bwd_kernel)
  f32x4 acc[LN_MAXQ], gm[LN_MAXQ];
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    gm[i] = lane + 64 * i < C4 ? kpf_ld4(gamma + 4 * (lane + 64 * i)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      if (q < C4) {
        const f32x4 gv = kpf_ld4(g + r * C + 4 * q), yv = kpf_ld4(y + r * C + 4 * q);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = gv[e] * gm[i][e];
          acc[i][e] = fmaf(gv[e], yv[e], acc[i][e]);
        }
        kpf_st4(dy + r * C + 4 * q, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i)
    if (lane + 64 * i < C4) kpf_st4(ls_lds + wave * C + 4 * (lane + 64 * i), acc[i]);
  __syncthreads();
  const int GC = G * C;
  for (int cc = threadIdx.x; cc < GC; cc += 256) {
    const int grp = cc / C, c = cc - grp * C;
    float s = 0.f;
    for (int wv = grp; wv < 4; wv += G) s += ls_lds[wv * C + c];  // (G = 1: waves 0..3 in order)
    part[((long)blockIdx.x * 2) * GC + cc] = s;  // (slot layout of ln_bwd_reduce_kernel: [block][2][G*C]; the second plane is unused here)
    part[((long)blockIdx.x * 2 + 1) * GC + cc] = 0.f;
  }
}

template <typename T>
int layer_scale_fwd_launch(const float* x, const void* y, const float* gamma, float* out, long rows, int C, void* stream) {
  hipLaunchKernelGGL(layer_scale_fwd_kernel<T>, dim3(grid_for(rows * (C / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, static_cast<const T*>(y),
                     gamma, out, rows, C / 4);
  return kpf_check_launch("kpf_layer_scale_forward");
}
template <typename T>
int layer_scale_bwd_launch(const float* g, const void* y, const float* gamma, void* dy, float* dgamma, float* ws, long rows, int C, void* stream,
                           kpf_colsum_desc* defer = nullptr, int G = 1) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nblk = ln_blocks(rows);
  const int GC = G * C;  // (G > 1: rows counts the [rows][C] view of a [rows / G][G*C] tensor; gamma, dgamma hold G sets)
  hipLaunchKernelGGL(layer_scale_bwd_kernel<T>, dim3(nblk), dim3(256), (size_t)4 * C * sizeof(float), st, g, static_cast<const T*>(y), gamma, static_cast<T*>(dy), ws,
                     rows, C / 4, G);
  int rc = kpf_check_launch("kpf_layer_scale_backward");
  if (rc != KPF_OK) return rc;
  if (defer) {
    defer->part = ws, defer->dw = dgamma, defer->db = ws + (long)nblk * 2 * GC, defer->nblk = nblk, defer->C = GC, defer->first_block = 0, defer->reserved = 0;
    return KPF_OK;
  }
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * GC + 63) / 64), dim3(512), 0, st, ws, dgamma, ws + (long)nblk * 2 * GC, nblk, GC);
  return kpf_check_launch("kpf_layer_scale_backward (reduce)");
}
}  // namespace

extern "C" long kpf_layer_scale_ws_floats(long rows, int C) { return (long)ln_blocks(rows) * 2 * C + C; }

extern "C" int kpf_layer_scale_forward(const float* x, const void* y, int y_dtype, const float* gamma, float* out, long rows, int C, void* stream) {
  KPF_REQUIRE(x && y && gamma && out && rows > 0 && C > 0 && C % 4 == 0, "kpf_layer_scale_forward: bad arguments");
#define CALL(T) layer_scale_fwd_launch<T>(x, y, gamma, out, rows, C, stream)
  KPF_DISPATCH_DT(y_dtype, "kpf_layer_scale_forward", CALL);
#undef CALL
}

extern "C" int kpf_layer_scale_backward(const float* g, const void* y, int y_dtype, const float* gamma, void* dy, float* dgamma, float* ws, long ws_floats,
                                        long rows, int C, void* stream) {
  KPF_REQUIRE(g && y && gamma && dy && dgamma && ws && rows > 0 && C > 0 && C % 4 == 0 && C <= 256 * LN_MAXQ, "kpf_layer_scale_backward: bad arguments (C <= 1024)");
  KPF_REQUIRE(ws_floats >= kpf_layer_scale_ws_floats(rows, C), "kpf_layer_scale_backward: workspace too small");
#define CALL(T) layer_scale_bwd_launch<T>(g, y, gamma, dy, dgamma, ws, rows, C, stream)
  KPF_DISPATCH_DT(y_dtype, "kpf_layer_scale_backward", CALL);
#undef CALL
}

extern "C" int kpf_layer_scale_backward_g(const float* g, const void* y, int y_dtype, const float* gamma, void* dy, float* dgamma, float* ws, long ws_floats,
                                          long rows, int C, int G, kpf_colsum_desc* desc, void* stream) {
  KPF_REQUIRE(g && y && gamma && dy && dgamma && ws && rows > 0 && C > 0 && C % 4 == 0 && C <= 256 * LN_MAXQ, "kpf_layer_scale_backward_g: bad arguments (C <= 1024)");
  KPF_REQUIRE((G == 1 || G == 2 || G == 4) && rows % G == 0, "kpf_layer_scale_backward_g: G must be 1, 2 or 4 and divide the row count");
  KPF_REQUIRE(ws_floats >= kpf_layer_scale_ws_floats(rows, G * C), "kpf_layer_scale_backward_g: workspace too small");
#define CALL(T) layer_scale_bwd_launch<T>(g, y, gamma, dy, dgamma, ws, rows, C, stream, desc, G)
  KPF_DISPATCH_DT(y_dtype, "kpf_layer_scale_backward_g", CALL);
#undef CALL
}

extern "C" int kpf_layer_scale_backward_partial(const float* g, const void* y, int y_dtype, const float* gamma, void* dy, float* dgamma, float* ws, long ws_floats,
                                                long rows, int C, kpf_colsum_desc* desc, void* stream) {
  KPF_REQUIRE(g && y && gamma && dy && dgamma && ws && desc && rows > 0 && C > 0 && C % 4 == 0 && C <= 256 * LN_MAXQ, "kpf_layer_scale_backward_partial: bad arguments");
  KPF_REQUIRE(ws_floats >= kpf_layer_scale_ws_floats(rows, C), "kpf_layer_scale_backward_partial: workspace too small");
#define CALL(T) layer_scale_bwd_launch<T>(g, y, gamma, dy, dgamma, ws, rows, C, stream, desc)
  KPF_DISPATCH_DT(y_dtype, "kpf_layer_scale_backward_partial", CALL);
#undef CALL
}

// ---------------------------------------------------------------------------------------------------------------
// The dense-stage loss of the training iteration in one kernel each way (train.py:211-224; util/generateFeature.py:59-84 joint2offset,
// :166-195 offset2joint_weight; model/loss.py:3-26): for stage output pd [B][5J][F][F] (3J unit offsets (j, xyz), J heat maps, J weight
// logits), the depth crop and the ground-truth joints uvd_gt [B][J][3]
//     loss_pixel = mean SmoothL1(pd[:, :4J] - joint2offset(uvd_gt))          loss_coord = mean SmoothL1(decode(pd) - uvd_gt)
// where decode is the masked soft-argmax.  One workgroup per (joint, sample): the F*F pixels are 4 per thread (256 threads up to F = 32, 1024 threads up to
// F = 64: the wide model), the softmax and the
// three weighted sums are block reductions, the target maps are computed on the fly (never materialised); it writes the two partial
// sums of its (sample, joint) and, backward, the gradient of its five channels — every element of d pd is written exactly once.
// The library path was ~200 element-wise launches per iteration over B x 105 x 32 x 32 maps.
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr int LS_PX = 4;  // pixels per thread: F * F <= 4 * threads

__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// the same reductions over NT / 64 waves (NT = 256: the expressions above, bit for bit; NT = 1024: four such groups, added in order)
template <int NT>
__device__ __forceinline__ float block_sum_nt(float v, float* red) {
  if constexpr (NT == 256) return block_sum256(v, red);
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NT / 256; ++q) s += (red[4 * q] + red[4 * q + 1]) + (red[4 * q + 2] + red[4 * q + 3]);
  return s;
}
template <int NT>
__device__ __forceinline__ float block_max_nt(float v, float* red) {
  if constexpr (NT == 256) return block_max256(v, red);
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = red[0];
#pragma unroll
  for (int q = 1; q < NT / 64; ++q) m = fmaxf(m, red[q]);
  return m;
}
__device__ __forceinline__ float sl1(float z) {  // model/loss.py: quadratic below 0.01, 0.01 (|z| - 0.005) from there on
  const float a = fabsf(z);
  return a < 0.01f ? 0.5f * z * z : 0.01f * (a - 0.005f);
}
__device__ __forceinline__ float sl1_grad(float z) { return fabsf(z) < 0.01f ? z : (z > 0.f ? 0.01f : (z < 0.f ? -0.01f : 0.f)); }

// MODE 0: forward (part[b][j] = {pixel sum, coord sum});  MODE 1: backward (dpd), scaled by gp = dL/dloss_pixel / n_pix, gc = dL/dloss_coord / n_coord
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void dense_loss_kernel(const float* __restrict__ pd, const float* __restrict__ img, const float* __restrict__ gt,
                                                         float* __restrict__ part, float* __restrict__ dpd, const float* __restrict__ gscale, int J, int F, int S,
                                                         float ks, float pix_w, float coord_w) {
  __shared__ float red[NT / 64];
  const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int P = F * F, st = S / F;
  const float* pdb = pd + (long)b * 5 * J * P;
  const float* imb = img + (long)b * S * S;
  const float gx = gt[((long)b * J + j) * 3 + 0], gy = gt[((long)b * J + j) * 3 + 1], gz = gt[((long)b * J + j) * 3 + 2];
  float d[LS_PX], m[LS_PX], lg[LS_PX], u[LS_PX], v[LS_PX];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < LS_PX; ++k) {
    const int p = tid + NT * k;
    const bool in = p < P;
    const int py = in ? p / F : 0, px = in ? p - py * F : 0;
    d[k] = in ? imb[(long)(py * st) * S + px * st] : 1.f;  // F.interpolate(nearest): source index = floor(dst * S / F)
    m[k] = d[k] < 0.99f ? 1.f : 0.f;
    u[k] = 2.f * ((float)px + 0.5f) / (float)F - 1.f;
    v[k] = 2.f * ((float)py + 0.5f) / (float)F - 1.f;
    lg[k] = in ? (d[k] > 0.99f ? -1e8f : pdb[(long)(4 * J + j) * P + p]) : -INFINITY;
    mx = fmaxf(mx, lg[k]);
  }
  mx = block_max_nt<NT>(mx, red);
  float w[LS_PX], se = 0.f;
#pragma unroll
  for (int k = 0; k < LS_PX; ++k) {
    w[k] = tid + NT * k < P ? __expf(lg[k] - mx) : 0.f;
    se += w[k];
  }
  se = block_sum_nt<NT>(se, red);
  const float inv = 1.0f / se;
  float heat[LS_PX], dist[LS_PX], un[3][LS_PX], val[3][LS_PX], Jc[3];
#pragma unroll
  for (int k = 0; k < LS_PX; ++k) {
    const int p = tid + NT * k;
    w[k] *= inv;
    heat[k] = p < P ? pdb[(long)(3 * J + j) * P + p] : 0.f;
    dist[k] = ks - heat[k] * m[k] * ks;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LS_PX; ++k) {
      const int p = tid + NT * k;
      un[c][k] = p < P ? pdb[(long)(3 * j + c) * P + p] : 0.f;
      const float coord = c == 0 ? u[k] : (c == 1 ? v[k] : d[k]);
      val[c][k] = un[c][k] * m[k] * dist[k] + coord;
      s = fmaf(val[c][k], w[k], s);
    }
    Jc[c] = block_sum_nt<NT>(s, red);
  }
  const float zc[3] = {Jc[0] - gx, Jc[1] - gy, Jc[2] - gz};
  if (MODE == 0) {
    // pixel term: the target maps of util/generateFeature.py:59-84 on the fly
    float ps = 0.f;
#pragma unroll
    for (int k = 0; k < LS_PX; ++k) {
      if (tid + NT * k >= P) continue;
      const float ox = gx - u[k], oy = gy - v[k], oz = gz - d[k];
      const float dg = sqrtf(ox * ox + oy * oy + oz * oz + 1e-8f);
      const float hm = (ks - dg) / ks;
      const float mg = (hm >= 0.f ? 1.f : 0.f) * m[k];
      ps += sl1(un[0][k] - ox / dg * mg) + sl1(un[1][k] - oy / dg * mg) + sl1(un[2][k] - oz / dg * mg) + sl1(heat[k] - hm * mg);
    }
    ps = block_sum_nt<NT>(ps, red);
    if (tid == 0) {
      part[((long)b * J + j) * 2 + 0] = ps;
      part[((long)b * J + j) * 2 + 1] = sl1(zc[0]) + sl1(zc[1]) + sl1(zc[2]);
    }
  } else {
    const float gp = gscale[0] * pix_w, gcs = gscale[1] * coord_w;
    const float gJ[3] = {gcs * sl1_grad(zc[0]), gcs * sl1_grad(zc[1]), gcs * sl1_grad(zc[2])};
    float* db = dpd + (long)b * 5 * J * P;
#pragma unroll
    for (int k = 0; k < LS_PX; ++k) {
      const int p = tid + NT * k;
      if (p >= P) continue;
      const float ox = gx - u[k], oy = gy - v[k], oz = gz - d[k];
      const float dg = sqrtf(ox * ox + oy * oy + oz * oz + 1e-8f);
      const float hm = (ks - dg) / ks;
      const float mg = (hm >= 0.f ? 1.f : 0.f) * m[k];
      const float tg[3] = {ox / dg * mg, oy / dg * mg, oz / dg * mg};
      float dheat = gp * sl1_grad(heat[k] - hm * mg), dlog = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        db[(long)(3 * j + c) * P + p] = gp * sl1_grad(un[c][k] - tg[c]) + gJ[c] * w[k] * m[k] * dist[k];
        dheat -= gJ[c] * w[k] * un[c][k] * m[k] * m[k] * ks;
        dlog += gJ[c] * w[k] * (val[c][k] - Jc[c]);
      }
      db[(long)(3 * J + j) * P + p] = dheat;
      db[(long)(4 * J + j) * P + p] = d[k] > 0.99f ? 0.f : dlog;  // masked_fill: no gradient into a replaced logit
    }
  }
}
}  // namespace

extern "C" int kpf_dense_loss_forward(const float* pd, const float* img, const float* uvd_gt, float* part, int B, int J, int F, int S, float kernel_size,
                                      void* stream) {
  KPF_REQUIRE(pd && img && uvd_gt && part && B > 0 && J > 0 && F > 0 && F * F <= 1024 * LS_PX && S % F == 0, "kpf_dense_loss_forward: bad arguments (F*F <= 4096)");
  if (F * F <= 256 * LS_PX)
    hipLaunchKernelGGL((dense_loss_kernel<0, 256>), dim3(J, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pd, img, uvd_gt, part, (float*)nullptr,
                       (const float*)nullptr, J, F, S, kernel_size, 0.f, 0.f);
  else
    hipLaunchKernelGGL((dense_loss_kernel<0, 1024>), dim3(J, B), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), pd, img, uvd_gt, part, (float*)nullptr,
                       (const float*)nullptr, J, F, S, kernel_size, 0.f, 0.f);
  return kpf_check_launch("kpf_dense_loss_forward");
}

extern "C" int kpf_dense_loss_backward(const float* pd, const float* img, const float* uvd_gt, const float* grad2, float* dpd, int B, int J, int F, int S,
                                       float kernel_size, void* stream) {
  KPF_REQUIRE(pd && img && uvd_gt && grad2 && dpd && B > 0 && J > 0 && F > 0 && F * F <= 1024 * LS_PX && S % F == 0, "kpf_dense_loss_backward: bad arguments (F*F <= 4096)");
  const float pix_w = 1.0f / ((float)B * 4 * J * F * F), coord_w = 1.0f / ((float)B * J * 3);
  if (F * F <= 256 * LS_PX)
    hipLaunchKernelGGL((dense_loss_kernel<1, 256>), dim3(J, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pd, img, uvd_gt, (float*)nullptr, dpd, grad2, J, F, S,
                       kernel_size, pix_w, coord_w);
  else
    hipLaunchKernelGGL((dense_loss_kernel<1, 1024>), dim3(J, B), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), pd, img, uvd_gt, (float*)nullptr, dpd, grad2, J, F, S,
                       kernel_size, pix_w, coord_w);
  return kpf_check_launch("kpf_dense_loss_backward");
}

// ---------------------------------------------------------------------------------------------------------------
// The rest of the loss (train.py:225-261) in two launches forward and one backward: the four SmoothL1 terms between the fusion blocks'
// joints and xyz_gt, the two spatial-weight terms against the Gaussian heat maps of util/generateFeature.py:584-600 divided by their
// global maximum, the epoch gate of train.py:251 read from the device, the weights, and the total — together with the partial sums the
// two dense-stage kernels leave.  `out` (16 floats): [0] loss, [1..4] pixel_0 coord_0 pixel_1 coord_1, [5..8] coord_2..5, [9..10]
// spatial_0..1 (weighted and gated, as the reference logs them), [12..13] the gates, [14..15] the heat maps' global maxima.
// The library path was ~260 element-wise launches forward and as many backward for a few thousand numbers.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct LossTailArgs {
  const float* dense[2];   // [B][J][2] partial sums of the dense stages (nullable)
  const float* joints[4];  // [B][J][3] (nullable)
  const float* xyz_gt;
  const float* uvd_gt;
  const float* sw[2];      // spatial weights (nullable): element (b, j, p) at b * sw_sb + j * sw_sj + p * sw_sp
  long sw_sb[2], sw_sj[2], sw_sp[2];
  const float* epoch;      // device scalar (nullable: gates = 1)
  float epoch_max[2];
  float sigma[2];
  float std_, coord_w, deconv_w, sp_w[2];
  int B, J, F;
};

// the heat map of joint (b, j) at pixel (x, y): util/generateFeature.py:584-600
__device__ __forceinline__ float hm_sq(float c, float jc, float std_) {
  const float t = (c + 0.5f - jc) / std_;
  return t * t;
}
__device__ __forceinline__ float hm_center(float uv, int F) { return (uv + 1.f) / 2.f * (float)F; }

// global maximum of the heat maps of every (sample, joint): exp is monotone and the exponent separable, so the maximum of a map is at
// (argmin_x, argmin_y) — 2F evaluations per joint instead of F*F; every workgroup recomputes it (B*J*2F values) rather than wait for one
__device__ float heatmap_global_max(const LossTailArgs& a, float sigma, float* red) {
  float best = 0.f;
  for (int q = threadIdx.x; q < a.B * a.J; q += 256) {
    const float jx = hm_center(a.uvd_gt[(long)q * 3 + 0], a.F), jy = hm_center(a.uvd_gt[(long)q * 3 + 1], a.F);
    float mx = INFINITY, my = INFINITY;
    for (int c = 0; c < a.F; ++c) {
      mx = fminf(mx, hm_sq((float)c, jx, a.std_));
      my = fminf(my, hm_sq((float)c, jy, a.std_));
    }
    best = fmaxf(best, expf(-(mx + my) / (2.f * sigma * sigma)));
  }
  return block_max256(best, red);
}

__device__ __forceinline__ float epoch_gate(const LossTailArgs& a, int t) { return a.epoch == nullptr ? 1.f : (a.epoch[0] <= a.epoch_max[t] ? 1.f : 0.f); }

// grid (J, B, 2): MODE 0 -> sp_part[t][b*J + j] = sum of SmoothL1(sw - hm / max);  MODE 1 -> d sw
template <int MODE>
__global__ __launch_bounds__(256) void spatial_loss_kernel(LossTailArgs a, float* __restrict__ sp_part, float* __restrict__ out, float* __restrict__ dsw0,
                                                           float* __restrict__ dsw1, const float* __restrict__ g, float* __restrict__ dj0, float* __restrict__ dj1,
                                                           float* __restrict__ dj2, float* __restrict__ dj3, float* __restrict__ gdense) {
  __shared__ float red[4];
  const int j = blockIdx.x, b = blockIdx.y, t = blockIdx.z, tid = threadIdx.x;
  const int P = a.F * a.F;
  if (MODE == 1 && t == 2) {  // the joints' gradients (and the two scalars the dense-stage backward kernels read), one (sample, joint) per workgroup
    float* dj[4] = {dj0, dj1, dj2, dj3};
    if (tid < 12) {
      const int s = tid / 3, c = tid - 3 * s;
      const long e = ((long)b * a.J + j) * 3 + c;
      if (a.joints[s] != nullptr && dj[s] != nullptr) dj[s][e] = g[0] * a.coord_w / (float)(a.B * a.J * 3) * sl1_grad(a.joints[s][e] - a.xyz_gt[e]);
    }
    if (j == 0 && b == 0 && tid == 64 && gdense != nullptr) {
      gdense[0] = g[0] * a.deconv_w;
      gdense[1] = g[0] * a.coord_w;
    }
    return;
  }
  if (a.sw[t] == nullptr) return;
  const float gmax = MODE == 0 ? heatmap_global_max(a, a.sigma[t], red) : out[14 + t];
  const float jx = hm_center(a.uvd_gt[((long)b * a.J + j) * 3 + 0], a.F), jy = hm_center(a.uvd_gt[((long)b * a.J + j) * 3 + 1], a.F);
  const float* sw = a.sw[t] + (long)b * a.sw_sb[t] + (long)j * a.sw_sj[t];
  float* dsw = MODE == 1 ? (t == 0 ? dsw0 : dsw1) + (long)b * a.sw_sb[t] + (long)j * a.sw_sj[t] : nullptr;
  const float gs = MODE == 1 ? g[0] * a.sp_w[t] * out[12 + t] / (float)((long)a.B * a.J * P) : 0.f;
  float s = 0.f;
  for (int p = tid; p < P; p += 256) {
    const int py = p / a.F, px = p - py * a.F;
    const float h = expf(-(hm_sq((float)px, jx, a.std_) + hm_sq((float)py, jy, a.std_)) / (2.f * a.sigma[t] * a.sigma[t]));
    const float z = sw[(long)p * a.sw_sp[t]] - h / gmax;
    if (MODE == 0) s += sl1(z);
    else dsw[(long)p * a.sw_sp[t]] = gs * sl1_grad(z);
  }
  if (MODE == 0) {
    s = block_sum256(s, red);
    if (tid == 0) {
      sp_part[(long)t * a.B * a.J + (long)b * a.J + j] = s;
      if (j == 0 && b == 0) out[14 + t] = gmax;
    }
  }
}

// one workgroup: every partial sum in a fixed order, the weights, the gates, the total (train.py:224-261 in the reference's order of additions)
__global__ __launch_bounds__(256) void loss_combine_kernel(LossTailArgs a, const float* __restrict__ sp_part, float* __restrict__ out) {
  __shared__ float red[4];
  const int tid = threadIdx.x, BJ = a.B * a.J, P = a.F * a.F;
  float term[10];
  for (int i = 0; i < 2; ++i)
    for (int k = 0; k < 2; ++k) {
      float s = 0.f;
      if (a.dense[i] != nullptr)
        for (int q = tid; q < BJ; q += 256) s += a.dense[i][(long)q * 2 + k];
      s = block_sum256(s, red);
      term[2 * i + k] = k == 0 ? s / (float)((long)BJ * 4 * P) * a.deconv_w : s / (float)(BJ * 3) * a.coord_w;
    }
  for (int i = 0; i < 4; ++i) {
    float s = 0.f;
    if (a.joints[i] != nullptr)
      for (int q = tid; q < BJ * 3; q += 256) s += sl1(a.joints[i][q] - a.xyz_gt[q]);
    s = block_sum256(s, red);
    term[4 + i] = s / (float)(BJ * 3) * a.coord_w;
  }
  float gate[2];
  for (int t = 0; t < 2; ++t) {
    float s = 0.f;
    if (a.sw[t] != nullptr)
      for (int q = tid; q < BJ; q += 256) s += sp_part[(long)t * BJ + q];
    s = block_sum256(s, red);
    gate[t] = epoch_gate(a, t);
    term[8 + t] = s / (float)((long)BJ * P) * a.sp_w[t] * gate[t];
  }
  if (tid == 0) {
    float loss = 0.f;
    loss += term[0] + term[1];
    loss += term[2] + term[3];
    for (int i = 4; i < 10; ++i) loss += term[i];
    out[0] = loss;
    for (int i = 0; i < 10; ++i) out[1 + i] = term[i];
    out[11] = 0.f;
    out[12] = gate[0];
    out[13] = gate[1];
  }
}

int fill_loss_args(LossTailArgs& a, const float* dense0, const float* dense1, const float* const* joints4, const float* xyz_gt, const float* uvd_gt,
                   const float* const* sw2, const long* sw_strides6, const float* epoch, const float* cfg9, int B, int J, int F) {
  a.dense[0] = dense0;
  a.dense[1] = dense1;
  for (int i = 0; i < 4; ++i) a.joints[i] = joints4[i];
  a.xyz_gt = xyz_gt;
  a.uvd_gt = uvd_gt;
  for (int t = 0; t < 2; ++t) {
    a.sw[t] = sw2[t];
    a.sw_sb[t] = sw_strides6[3 * t + 0];
    a.sw_sj[t] = sw_strides6[3 * t + 1];
    a.sw_sp[t] = sw_strides6[3 * t + 2];
    a.epoch_max[t] = cfg9[7 + t];
    a.sigma[t] = cfg9[1 + t];
    a.sp_w[t] = cfg9[5 + t];
  }
  a.epoch = epoch;
  a.std_ = cfg9[0];
  a.coord_w = cfg9[3];
  a.deconv_w = cfg9[4];
  a.B = B;
  a.J = J;
  a.F = F;
  return 0;
}
}  // namespace

extern "C" int kpf_loss_tail_forward(const float* dense0, const float* dense1, const float* const* joints4, const float* xyz_gt, const float* uvd_gt,
                                     const float* const* sw2, const long* sw_strides6, const float* epoch, const float* cfg9, float* sp_part, float* out16,
                                     int B, int J, int F, void* stream) {
  KPF_REQUIRE(joints4 && xyz_gt && uvd_gt && sw2 && sw_strides6 && cfg9 && sp_part && out16 && B > 0 && J > 0 && F > 0, "kpf_loss_tail_forward: bad arguments");
  LossTailArgs a;
  fill_loss_args(a, dense0, dense1, joints4, xyz_gt, uvd_gt, sw2, sw_strides6, epoch, cfg9, B, J, F);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.sw[0] != nullptr || a.sw[1] != nullptr)
    hipLaunchKernelGGL(spatial_loss_kernel<0>, dim3(J, B, 2), dim3(256), 0, st, a, sp_part, out16, (float*)nullptr, (float*)nullptr, (const float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(256), 0, st, a, sp_part, out16);
  return kpf_check_launch("kpf_loss_tail_forward");
}

extern "C" int kpf_loss_tail_backward(const float* const* joints4, const float* xyz_gt, const float* uvd_gt, const float* const* sw2, const long* sw_strides6, const float* cfg9, const float* out16, const float* g,
                                      float* const* djoints4, float* const* dsw2, float* gdense2, int B, int J, int F, void* stream) {
  KPF_REQUIRE(joints4 && xyz_gt && uvd_gt && sw2 && sw_strides6 && cfg9 && out16 && g && djoints4 && dsw2 && B > 0 && J > 0 && F > 0,
              "kpf_loss_tail_backward: bad arguments");
  LossTailArgs a;
  fill_loss_args(a, nullptr, nullptr, joints4, xyz_gt, uvd_gt, sw2, sw_strides6, nullptr, cfg9, B, J, F);
  for (int t = 0; t < 2; ++t)
    if (dsw2[t] == nullptr) a.sw[t] = nullptr;
  hipLaunchKernelGGL(spatial_loss_kernel<1>, dim3(J, B, 3), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, (float*)nullptr, const_cast<float*>(out16),
                     dsw2[0], dsw2[1], g, djoints4[0], djoints4[1], djoints4[2], djoints4[3], gdense2);
  return kpf_check_launch("kpf_loss_tail_backward");
}

// ---------------------------------------------------------------------------------------------------------------
// AdamW (train.py:84-91: torch.optim.AdamW, weight_decay 0.01) over every parameter of the model in a handful of launches: the library's
// fused multi-tensor kernel takes ~30 launches of 40 us for the ~300 tensors of a ConvNeXt-T KPFusion (4 KB of kernel arguments each, a
// few tensors per launch, 0.5 TB/s); here a launch carries KPF_ADAMW_BATCH descriptors by value (read from the kernel-argument segment with
// scalar loads), a workgroup owns 4096 consecutive elements of one tensor, and the learning rate and the step count are DEVICE scalars (a
// captured iteration keeps following the scheduler).  Arithmetic as torch's fused kernel (ADAMW mode, no amsgrad):
//   p -= lr*wd*p;  m += (1-b1)(g-m);  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  t = step + 1.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct AdamwBatch {
  kpf_adamw_desc d[KPF_ADAMW_BATCH];
  const float* lr_dev;
  const float* step_dev;
  const float* inv_scale;  // fp16 loss scaling (kpf_adamw_step_multi_scaled): the gradients are multiplied by *inv_scale as they are read; null = 1
  const int* skip;         // ... and the whole step is a no-op when *skip != 0 (a non-finite gradient was found); null = never
  float lr_host, beta1, beta2, omb1, omb2, eps, wd;  // omb = 1 - beta, rounded from the double difference (1.f - 0.999f is 4.7e-5 off)
  int nd;
};
typedef const __attribute__((address_space(4))) AdamwBatch* adamw_kernarg_t;
constexpr int ADAMW_PER_BLOCK = 4096;

__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamwBatch) {
  adamw_kernarg_t bp = (adamw_kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();  // (indexing the by-value struct would copy it to scratch)
  if (bp->skip && bp->skip[0] != 0) return;  // (uniform: every workgroup of every launch of this step reads the same flag)
  const float gs = bp->inv_scale ? bp->inv_scale[0] : 1.0f;
  const int nd = bp->nd;
  int lo = 0, hi = nd - 1;
  while (lo < hi) {  // last descriptor whose first_block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= bp->d[mid].first_block) lo = mid;
    else hi = mid - 1;
  }
  float* p = bp->d[lo].p;
  const float* g = bp->d[lo].g;
  float* m = bp->d[lo].m;
  float* v = bp->d[lo].v;
  const long n = bp->d[lo].n;
  const long base = (long)((int)blockIdx.x - bp->d[lo].first_block) * ADAMW_PER_BLOCK;
  const float lr = bp->lr_dev ? bp->lr_dev[0] : bp->lr_host;
  const double t = (double)bp->step_dev[0] + 1.0;
  const float b1 = bp->beta1, b2 = bp->beta2, eps = bp->eps, omb1 = bp->omb1, omb2 = bp->omb2;
  const float bc1 = (float)(1.0 - pow((double)b1, t));
  const float bc2s = (float)sqrt(1.0 - pow((double)b2, t));
  const float step_size = lr / bc1, decay = lr * bp->wd;
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
#pragma unroll
  for (int it = 0; it < ADAMW_PER_BLOCK / 1024; ++it) {
    const long i = base + (long)it * 1024 + 4 * threadIdx.x;
    if (i >= n) break;
    if (vec && i + 4 <= n) {
      f32x4 pp = *reinterpret_cast<const f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mm = *reinterpret_cast<const f32x4*>(m + i), vv = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        gg[c] *= gs;
        pp[c] -= decay * pp[c];
        mm[c] += omb1 * (gg[c] - mm[c]);
        vv[c] = b2 * vv[c] + omb2 * gg[c] * gg[c];
        pp[c] -= step_size * mm[c] / (sqrtf(vv[c]) / bc2s + eps);
      }
      *reinterpret_cast<f32x4*>(p + i) = pp;
      *reinterpret_cast<f32x4*>(m + i) = mm;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (long e = i; e < i + 4 && e < n; ++e) {
        float pp = p[e], mm = m[e], vv = v[e];
        const float gg = g[e] * gs;
        pp -= decay * pp;
        mm += omb1 * (gg - mm);
        vv = b2 * vv + omb2 * gg * gg;
        pp -= step_size * mm / (sqrtf(vv) / bc2s + eps);
        p[e] = pp, m[e] = mm, v[e] = vv;
      }
    }
  }
}
}  // namespace

extern "C" int kpf_adamw_step_multi_scaled(const kpf_adamw_desc* descs, int n, const float* lr_dev, float lr_host, const float* step_dev, double beta1,
                                           double beta2, float eps, float weight_decay, const float* inv_scale_dev, const int* skip_dev, void* stream);
extern "C" int kpf_adamw_step_multi(const kpf_adamw_desc* descs, int n, const float* lr_dev, float lr_host, const float* step_dev, double beta1, double beta2,
                                    float eps, float weight_decay, void* stream) {
  return kpf_adamw_step_multi_scaled(descs, n, lr_dev, lr_host, step_dev, beta1, beta2, eps, weight_decay, nullptr, nullptr, stream);
}
extern "C" int kpf_adamw_step_multi_scaled(const kpf_adamw_desc* descs, int n, const float* lr_dev, float lr_host, const float* step_dev, double beta1,
                                           double beta2, float eps, float weight_decay, const float* inv_scale_dev, const int* skip_dev, void* stream) {
  KPF_REQUIRE(n >= 0 && (descs || n == 0) && step_dev, "kpf_adamw_step_multi: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int base = 0; base < n; base += KPF_ADAMW_BATCH) {
    AdamwBatch b;
    b.nd = n - base < KPF_ADAMW_BATCH ? n - base : KPF_ADAMW_BATCH;
    long blocks = 0;
    for (int k = 0; k < b.nd; ++k) {
      b.d[k] = descs[base + k];
      KPF_REQUIRE(b.d[k].p && b.d[k].g && b.d[k].m && b.d[k].v && b.d[k].n > 0, "kpf_adamw_step_multi: bad descriptor %d", base + k);
      b.d[k].first_block = (int)blocks;
      blocks += (b.d[k].n + ADAMW_PER_BLOCK - 1) / ADAMW_PER_BLOCK;
    }
    KPF_REQUIRE(blocks < (1L << 31), "kpf_adamw_step_multi: too many elements");
    b.inv_scale = inv_scale_dev, b.skip = skip_dev;
    b.lr_dev = lr_dev, b.step_dev = step_dev, b.lr_host = lr_host, b.beta1 = (float)beta1, b.beta2 = (float)beta2, b.omb1 = (float)(1.0 - beta1), b.omb2 = (float)(1.0 - beta2), b.eps = eps, b.wd = weight_decay;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
    const int rc = kpf_check_launch("kpf_adamw_step_multi");
    if (rc != KPF_OK) return rc;
  }
  return KPF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Loss scaling for fp16 mixed-precision training (round 5; what torch.cuda.amp.GradScaler does around train.py:262-264, entirely on the device so that a
// captured iteration keeps following it): the loss is multiplied by a device scalar `scale` before backward; afterwards
//   (1) kpf_grad_finite_check_multi: one pass over every gradient (the AdamW descriptors), *found |= any value is inf / nan (an integer OR: order-independent);
//   (2) kpf_adamw_step_multi_scaled: AdamW on g * (1 / scale), or nothing at all when *found;
//   (3) kpf_loss_scale_update: found -> scale *= backoff, growth tracker = 0; else tracker + 1, and scale *= growth every `interval` clean steps;
//       1 / scale refreshed, the optimiser's step count advanced only by a step that happened, the flag cleared for the next iteration.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void grad_finite_check_kernel(const AdamwBatch) {
  adamw_kernarg_t bp = (adamw_kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int nd = bp->nd;
  int lo = 0, hi = nd - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= bp->d[mid].first_block) lo = mid;
    else hi = mid - 1;
  }
  const float* g = bp->d[lo].g;
  const long n = bp->d[lo].n;
  const long base = (long)((int)blockIdx.x - bp->d[lo].first_block) * ADAMW_PER_BLOCK;
  bool bad = false;
  for (long i = base + threadIdx.x; i < n && i < base + ADAMW_PER_BLOCK; i += 256) {
    const float v = g[i];
    bad |= !(fabsf(v) <= 3.4028234e38f);  // inf or nan
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(const_cast<int*>(bp->skip), 1);
}
__global__ void loss_scale_update_kernel(float* scale, float* inv_scale, int* tracker, int* found, float* step, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = scale[0];
  if (found[0] != 0) {
    s *= backoff;
    tracker[0] = 0;
  } else {
    if (step) step[0] += 1.0f;
    const int t = tracker[0] + 1;
    if (t >= interval) {
      s *= growth;
      tracker[0] = 0;
    } else {
      tracker[0] = t;
    }
  }
  s = fminf(fmaxf(s, 1.0f), 16777216.0f);  // [1, 2^24]
  scale[0] = s;
  inv_scale[0] = 1.0f / s;
  found[0] = 0;
}
}  // namespace

extern "C" int kpf_grad_finite_check_multi(const kpf_adamw_desc* descs, int n, int* found_dev, void* stream) {
  KPF_REQUIRE(n >= 0 && (descs || n == 0) && found_dev, "kpf_grad_finite_check_multi: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int base = 0; base < n; base += KPF_ADAMW_BATCH) {
    AdamwBatch b;
    b.nd = n - base < KPF_ADAMW_BATCH ? n - base : KPF_ADAMW_BATCH;
    long blocks = 0;
    for (int k = 0; k < b.nd; ++k) {
      b.d[k] = descs[base + k];
      KPF_REQUIRE(b.d[k].g && b.d[k].n > 0, "kpf_grad_finite_check_multi: bad descriptor %d", base + k);
      b.d[k].first_block = (int)blocks;
      blocks += (b.d[k].n + ADAMW_PER_BLOCK - 1) / ADAMW_PER_BLOCK;
    }
    KPF_REQUIRE(blocks < (1L << 31), "kpf_grad_finite_check_multi: too many elements");
    b.lr_dev = nullptr, b.step_dev = nullptr, b.inv_scale = nullptr, b.skip = found_dev;
    b.lr_host = b.beta1 = b.beta2 = b.omb1 = b.omb2 = b.eps = b.wd = 0.f;
    hipLaunchKernelGGL(grad_finite_check_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
    const int rc = kpf_check_launch("kpf_grad_finite_check_multi");
    if (rc != KPF_OK) return rc;
  }
  return KPF_OK;
}

extern "C" int kpf_loss_scale_update(float* scale_dev, float* inv_scale_dev, int* tracker_dev, int* found_dev, float* step_dev, float growth, float backoff,
                                     int interval, void* stream) {
  KPF_REQUIRE(scale_dev && inv_scale_dev && tracker_dev && found_dev && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval > 0,
              "kpf_loss_scale_update: bad arguments");
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), scale_dev, inv_scale_dev, tracker_dev, found_dev, step_dev,
                     growth, backoff, interval);
  return kpf_check_launch("kpf_loss_scale_update");
}

// ---------------------------------------------------------------------------------------------------------------
// The two analytic maps a fusion block derives from its first joint estimate (model/model.py:300-336 in train mode): the Gaussian heat map
// of the joints' (u, v) (GFM.joint2heatmap, util/generateFeature.py:584-600) and the geometry adjacency 1 / (10 |pixel_xyz - joint_xyz|^2 + 1)
// (dataloader/loader.py:791-819), each with its gradient towards the joints — one kernel each way, one workgroup per (joint, sample),
// instead of ~20 element-wise launches each way over B x 21 x 1024 (x 3) intermediates.  The pixel sums of the backward are block
// reductions in a fixed order.
// ---------------------------------------------------------------------------------------------------------------
namespace {
// MODE 0: hm[b][j][p];  MODE 1: duv[b][j][0..2] = (d/du, d/dv, 0) of sum_p dhm * hm
template <int MODE>
__global__ __launch_bounds__(256) void joint_heatmap_kernel(const float* __restrict__ uvd, const float* __restrict__ dhm, float* __restrict__ out, int J, int F,
                                                            float std_, float sigma) {
  __shared__ float red[4];
  const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int P = F * F;
  const long bj = (long)b * J + j;
  const float jx = hm_center(uvd[bj * 3 + 0], F), jy = hm_center(uvd[bj * 3 + 1], F);
  const float inv2s = 1.f / (2.f * sigma * sigma);
  float gu = 0.f, gv = 0.f;
  for (int p = tid; p < P; p += 256) {
    const int py = p / F, px = p - py * F;
    const float tx = ((float)px + 0.5f - jx) / std_, ty = ((float)py + 0.5f - jy) / std_;
    const float h = expf(-(tx * tx + ty * ty) * inv2s);
    if (MODE == 0) {
      out[bj * P + p] = h;
    } else {
      const float g = dhm[bj * P + p] * h;
      gu += g * tx;
      gv += g * ty;
    }
  }
  if (MODE == 1) {
    // d h / d u = h * (2 tx inv2s) * (1 / std) * (F / 2)      (jx = (u + 1) / 2 * F enters tx with a minus sign, the exponent with another)
    const float k = 2.f * inv2s / std_ * 0.5f * (float)F;
    gu = block_sum256(gu, red) * k;
    gv = block_sum256(gv, red) * k;
    if (tid == 0) {
      out[bj * 3 + 0] = gu;
      out[bj * 3 + 1] = gv;
      out[bj * 3 + 2] = 0.f;
    }
  }
}

// MODE 0: gam[b][j][p] = 1 / (10 |ix[b][p] - jx[b][j]|^2 + 1);  MODE 1: djx[b][j][c] = sum_p dgam * (-20 gam^2 (jx_c - ix_c))
template <int MODE>
__global__ __launch_bounds__(256) void geom_gate_kernel(const float* __restrict__ ix, const float* __restrict__ jx, const float* __restrict__ dgam,
                                                        float* __restrict__ out, int J, int P, const float* __restrict__ par = nullptr, float half_size = 0.f,
                                                        float flip = 1.f) {
  // par != nullptr (kpf_geom_gate_uvd_*): jx holds the joints as crop coordinates uvd in [-1, 1]^3 and par the sample's 16 numbers
  // [M^-1 rows 0-1 (6) | fx fy u0 v0 | centre xyz | cube xyz]; the joint goes through dataloader/loader.py:775-789 (uvd -> pixel -> camera ->
  // cube-normalised xyz) here, and the backward multiplies by that map's Jacobian — ~25 element-wise launches each way in the torch expression.
  __shared__ float red[4];
  const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const long bj = (long)b * J + j;
  float x0 = jx[bj * 3 + 0], x1 = jx[bj * 3 + 1], x2 = jx[bj * 3 + 2];
  float tx = 0.f, ty = 0.f, dep = 0.f;
  const float* pp = par ? par + (long)b * 16 : nullptr;
  if (pp) {
    const float u = (x0 + 1.f) * half_size, v = (x1 + 1.f) * half_size;
    dep = x2 * (pp[15] * 0.5f) + pp[12];
    tx = pp[0] * u + pp[1] * v + pp[2];
    ty = pp[3] * u + pp[4] * v + pp[5];
    const float cx = (tx - pp[8]) * dep / pp[6], cy = flip * (ty - pp[9]) * dep / pp[7];
    x0 = (cx - pp[10]) / (pp[13] * 0.5f);
    x1 = (cy - pp[11]) / (pp[14] * 0.5f);
    x2 = (dep - pp[12]) / (pp[15] * 0.5f);
  }
  const float* ib = ix + (long)b * P * 3;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  for (int p = tid; p < P; p += 256) {
    const float d0 = ib[p * 3 + 0] - x0, d1 = ib[p * 3 + 1] - x1, d2 = ib[p * 3 + 2] - x2;
    const float gam = 1.f / (10.f * ((d0 * d0 + d1 * d1) + d2 * d2) + 1.f);
    if (MODE == 0) {
      out[bj * P + p] = gam;
    } else {
      const float k = dgam[bj * P + p] * 20.f * gam * gam;  // d gam / d jx_c = -gam^2 * 10 * 2 (jx_c - ix_c) = 20 gam^2 (ix_c - jx_c)
      g0 += k * d0;
      g1 += k * d1;
      g2 += k * d2;
    }
  }
  if (MODE == 1) {
    g0 = block_sum256(g0, red);
    g1 = block_sum256(g1, red);
    g2 = block_sum256(g2, red);
    if (tid == 0) {
      if (pp) {  // d/d(uvd) = J^T d/d(xyz_normalised)
        const float gx = g0 / (pp[13] * 0.5f), gy = g1 / (pp[14] * 0.5f), gz = g2 / (pp[15] * 0.5f);
        const float ax = dep / pp[6] * half_size, ay = flip * dep / pp[7] * half_size, hz = pp[15] * 0.5f;
        g0 = gx * pp[0] * ax + gy * pp[3] * ay;
        g1 = gx * pp[1] * ax + gy * pp[4] * ay;
        g2 = (gx * (tx - pp[8]) / pp[6] + gy * flip * (ty - pp[9]) / pp[7] + gz) * hz;
      }
      out[bj * 3 + 0] = g0;
      out[bj * 3 + 1] = g1;
      out[bj * 3 + 2] = g2;
    }
  }
}
}  // namespace

extern "C" int kpf_joint_heatmap_forward(const float* uvd, float* hm, int B, int J, int F, float std_, float sigma, void* stream) {
  KPF_REQUIRE(uvd && hm && B > 0 && J > 0 && F > 0 && std_ > 0.f && sigma > 0.f, "kpf_joint_heatmap_forward: bad arguments");
  hipLaunchKernelGGL(joint_heatmap_kernel<0>, dim3(J, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), uvd, (const float*)nullptr, hm, J, F, std_, sigma);
  return kpf_check_launch("kpf_joint_heatmap_forward");
}
extern "C" int kpf_joint_heatmap_backward(const float* uvd, const float* dhm, float* duvd, int B, int J, int F, float std_, float sigma, void* stream) {
  KPF_REQUIRE(uvd && dhm && duvd && B > 0 && J > 0 && F > 0 && std_ > 0.f && sigma > 0.f, "kpf_joint_heatmap_backward: bad arguments");
  hipLaunchKernelGGL(joint_heatmap_kernel<1>, dim3(J, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), uvd, dhm, duvd, J, F, std_, sigma);
  return kpf_check_launch("kpf_joint_heatmap_backward");
}
extern "C" int kpf_geom_gate_forward(const float* pix_xyz, const float* joint_xyz, float* gam, int B, int J, int P, void* stream) {
  KPF_REQUIRE(pix_xyz && joint_xyz && gam && B > 0 && J > 0 && P > 0, "kpf_geom_gate_forward: bad arguments");
  hipLaunchKernelGGL(geom_gate_kernel<0>, dim3(J, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pix_xyz, joint_xyz, (const float*)nullptr, gam, J, P);
  return kpf_check_launch("kpf_geom_gate_forward");
}
extern "C" int kpf_geom_gate_backward(const float* pix_xyz, const float* joint_xyz, const float* dgam, float* djoint, int B, int J, int P, void* stream) {
  KPF_REQUIRE(pix_xyz && joint_xyz && dgam && djoint && B > 0 && J > 0 && P > 0, "kpf_geom_gate_backward: bad arguments");
  hipLaunchKernelGGL(geom_gate_kernel<1>, dim3(J, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pix_xyz, joint_xyz, dgam, djoint, J, P);
  return kpf_check_launch("kpf_geom_gate_backward");
}
extern "C" int kpf_geom_gate_uvd_forward(const float* pix_xyz, const float* joint_uvd, const float* par16, float* gam, int B, int J, int P, float half_size,
                                         float flip, void* stream) {
  KPF_REQUIRE(pix_xyz && joint_uvd && par16 && gam && B > 0 && J > 0 && P > 0 && half_size > 0.f, "kpf_geom_gate_uvd_forward: bad arguments");
  hipLaunchKernelGGL(geom_gate_kernel<0>, dim3(J, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pix_xyz, joint_uvd, (const float*)nullptr, gam, J, P, par16,
                     half_size, flip);
  return kpf_check_launch("kpf_geom_gate_uvd_forward");
}
extern "C" int kpf_geom_gate_uvd_backward(const float* pix_xyz, const float* joint_uvd, const float* par16, const float* dgam, float* djoint_uvd, int B, int J, int P,
                                          float half_size, float flip, void* stream) {
  KPF_REQUIRE(pix_xyz && joint_uvd && par16 && dgam && djoint_uvd && B > 0 && J > 0 && P > 0 && half_size > 0.f, "kpf_geom_gate_uvd_backward: bad arguments");
  hipLaunchKernelGGL(geom_gate_kernel<1>, dim3(J, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pix_xyz, joint_uvd, dgam, djoint_uvd, J, P, par16, half_size,
                     flip);
  return kpf_check_launch("kpf_geom_gate_uvd_backward");
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the small producers around the odd-width Linear layers of the fusion head (3-d coordinates, the 105 pose channels, the 149-channel
// gate input, the 3-wide joint heads): a row pad in ONE launch (F.pad is a fill plus a strided copy, and its autograd twin a zero fill plus a
// copy) and the pose tokens of model/model.py:308-316 written directly at the padded width.
// ---------------------------------------------------------------------------------------------------------------
namespace {
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void pad_rows_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long rows, int C, int src_ld, int Cp) {
  const long total = rows * Cp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / Cp;
    const int c = (int)(i - r * Cp);
    dst[i] = c < C ? (TD)(float)src[r * src_ld + c] : (TD)0.f;
  }
}

// thread = (point, joint): unit offset joint - point (3), closeness (kernel - |offset|) / kernel, both masked by closeness >= 0 and point depth < 0.99
__global__ __launch_bounds__(256) void pose_tokens_kernel(const float* __restrict__ pw, const float* __restrict__ joint, const float* __restrict__ pcl,
                                                          float* __restrict__ out, long total, int N, int J, int ld, float kernel) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long bn = t / J;
  const int j = (int)(t - bn * J);
  const int b = (int)(bn / N);
  const float* pt = pcl + bn * 3;
  const float* jt = joint + ((long)b * J + j) * 3;
  const float o0 = jt[0] - pt[0], o1 = jt[1] - pt[1], o2 = jt[2] - pt[2];
  const float dis = sqrtf((o0 * o0 + o1 * o1) + o2 * o2);
  const float inv = 1.0f / (dis + 1e-8f);
  float clos = (kernel - dis) / kernel;
  const float m = (clos >= 0.f && pt[2] < 0.99f) ? 1.f : 0.f;
  float* o = out + bn * ld;
  int base = 0;
  if (pw) {
    o[j] = pw[bn * J + j];
    base = J;
  }
  o[base + 3 * j + 0] = o0 * inv * m;
  o[base + 3 * j + 1] = o1 * inv * m;
  o[base + 3 * j + 2] = o2 * inv * m;
  o[base + 3 * J + j] = clos * m;
  const int used = base + 4 * J;
  if (j < ld - used) o[used + j] = 0.f;  // (zero padding up to the row stride: fewer than J channels)
}
}  // namespace

/* dst[r][c] = c < C ? src[r * src_ld + c] : 0 for c < Cp (Cp >= C): a zero pad of the last axis — or, with Cp == C < src_ld, a dense copy of a column slice —
 * with an optional type change (KPF_DT_*). */
extern "C" int kpf_pad_rows(const void* src, int src_dtype, void* dst, int dst_dtype, long rows, int C, int src_ld, int Cp, void* stream) {
  KPF_REQUIRE(src && dst && rows > 0 && C > 0 && Cp >= C && src_ld >= C, "kpf_pad_rows: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(rows * Cp));
#define PR(TS, TD) hipLaunchKernelGGL((pad_rows_kernel<TS, TD>), grid, dim3(256), 0, st, static_cast<const TS*>(src), static_cast<TD*>(dst), rows, C, src_ld, Cp)
  if (src_dtype == KPF_DT_F32 && dst_dtype == KPF_DT_F32) PR(float, float);
  else if (src_dtype == KPF_DT_F32 && dst_dtype == KPF_DT_BF16) PR(float, bf16_t);
  else if (src_dtype == KPF_DT_F32 && dst_dtype == KPF_DT_F16) PR(float, f16_t);
  else if (src_dtype == KPF_DT_BF16 && dst_dtype == KPF_DT_BF16) PR(bf16_t, bf16_t);
  else if (src_dtype == KPF_DT_F16 && dst_dtype == KPF_DT_F16) PR(f16_t, f16_t);
  else if (src_dtype == KPF_DT_BF16 && dst_dtype == KPF_DT_F32) PR(bf16_t, float);
  else if (src_dtype == KPF_DT_F16 && dst_dtype == KPF_DT_F32) PR(f16_t, float);
  else {
    kpf_set_error("kpf_pad_rows: unsupported type pair %d -> %d", src_dtype, dst_dtype);
    return KPF_EINVAL;
  }
#undef PR
  return kpf_check_launch("kpf_pad_rows");
}

// Round 6: the seam between the paired backbones and the fusion head.  The backbones' maps arrive channel-stacked ([rows][G * gs] in the step's storage type, group
// g's C channels at column g * gs); the head wants one dense fp32 map per backbone.  Forward: all G maps in ONE launch (dst [G][rows][C] fp32); backward: the G
// gradients (fp32 [rows][C] each, any of them NULL = zero) back into ONE stacked tensor of the storage type, every element (pad columns included) written once —
// instead of a strided cast per map forward and cast + zero fill + strided copy + fan-in add per map backward (~15 library launches per iteration).
namespace {
struct RestackSrc { const float* g[4]; };
template <typename TS>
__global__ __launch_bounds__(256) void unstack_rows_kernel(const TS* __restrict__ src, float* __restrict__ dst, long rows, int G, int C, int ld, int gs, int hw) {
  // hw == 0: dst [G][rows][C];  hw > 0: dst [G][rows / hw][C][hw] (dense NCHW maps: what the decode and the loss read), pixel index fastest
  const long per = rows * C, total = per * G;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int g = (int)(i / per);
    const long j = i - g * per;
    long r;
    int c;
    if (hw == 0) {
      r = j / C, c = (int)(j - r * C);
    } else {
      const long bc = j / hw;
      const int p = (int)(j - bc * hw);
      const long b = bc / C;
      c = (int)(bc - b * C), r = b * hw + p;
    }
    dst[i] = (float)src[r * ld + g * gs + c];
  }
}
template <typename TD>
__global__ __launch_bounds__(256) void restack_rows_kernel(const RestackSrc s, TD* __restrict__ dst, long rows, int G, int C, int ld, int gs, int hw) {
  const long total = rows * ld;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / ld;
    const int col = (int)(i - r * ld);
    const int g = col / gs, c = col - g * gs;
    float v = 0.f;
    if (g < G && c < C) {
      const float* p = s.g[g];
      if (p) {
        if (hw == 0) v = p[r * C + c];
        else {
          const long b = r / hw;
          v = p[(b * C + c) * hw + (r - b * hw)];
        }
      }
    }
    dst[i] = (TD)v;
  }
}
}  // namespace

namespace {
// four columns per thread for rows of whole channel quads (hw == 0, C % 4 == 0, gs % 4 == 0, ld % 4 == 0): 8- / 16-byte accesses instead of one element per
// thread (the scalar form took 30 us for the 50 MB of the 128-channel feature maps' gradient)
template <typename TS>
__global__ __launch_bounds__(256) void unstack_rows4_kernel(const TS* __restrict__ src, float* __restrict__ dst, long rows, int G, int C4, int ld, int gs) {
  const long per = rows * C4, total = per * G;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int g = (int)(i / per);
    const long j = i - g * per;
    const long r = j / C4;
    const int q = (int)(j - r * C4);
    kpf_st4(dst + 4 * i, kpf_ld4(src + r * ld + g * gs + 4 * q));
  }
}
template <typename TD>
__global__ __launch_bounds__(256) void restack_rows4_kernel(const RestackSrc s, TD* __restrict__ dst, long rows, int G, int C4, int ld4, int gs4) {
  const long total = rows * ld4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / ld4;
    const int col = (int)(i - r * ld4);
    const int g = col / gs4, q = col - g * gs4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (g < G && q < C4) {
      const float* p = s.g[g];
      if (p) v = kpf_ld4(p + (r * C4 + q) * 4);
    }
    kpf_st4(dst + 4 * i, v);
  }
}
}  // namespace

namespace {
// the NCHW forms (hw > 0) as tiled transposes through LDS: workgroup = (64-pixel tile, sample, network); rows of C channels are read / written as whole runs and
// the planes as 256-byte runs — the element-per-thread form above walks one side with a stride of a pixel row (31 / 41 us for the 105-channel maps, now ~8)
constexpr int UT_PX = 64;
template <typename TS>
__global__ __launch_bounds__(256) void unstack_nchw_kernel(const TS* __restrict__ src, float* __restrict__ dst, int Bn, int C, int ld, int gs, int hw) {
  extern __shared__ float ut[];  // [C][UT_PX + 1]
  const int p0 = blockIdx.x * UT_PX, b = blockIdx.y, g = blockIdx.z;
  const int npx = min(UT_PX, hw - p0);
  for (int e = threadIdx.x; e < npx * C; e += 256) {
    const int px = e / C, c = e - px * C;
    ut[c * (UT_PX + 1) + px] = (float)src[((long)b * hw + p0 + px) * ld + g * gs + c];
  }
  __syncthreads();
  float* o = dst + (((long)g * Bn + b) * C) * hw + p0;
  for (int e = threadIdx.x; e < C * UT_PX; e += 256) {
    const int c = e / UT_PX, px = e - c * UT_PX;
    if (px < npx) o[(long)c * hw + px] = ut[c * (UT_PX + 1) + px];
  }
}
template <typename TD>
__global__ __launch_bounds__(256) void restack_nchw_kernel(const RestackSrc s, TD* __restrict__ dst, int C, int ld, int gs, int hw, int G) {
  extern __shared__ float ut[];
  const int p0 = blockIdx.x * UT_PX, b = blockIdx.y, g = blockIdx.z;
  const int npx = min(UT_PX, hw - p0);
  const float* gp = s.g[g];
  for (int e = threadIdx.x; e < C * UT_PX; e += 256) {
    const int c = e / UT_PX, px = e - c * UT_PX;
    ut[c * (UT_PX + 1) + px] = (gp && px < npx) ? gp[((long)b * C + c) * hw + p0 + px] : 0.f;
  }
  __syncthreads();
  const int wcols = g == G - 1 ? ld - g * gs : gs;  // this network's columns of the stacked row: C real ones, then zeros (the last network also owns the row's tail)
  for (int e = threadIdx.x; e < npx * wcols; e += 256) {
    const int px = e / wcols, c = e - px * wcols;
    dst[((long)b * hw + p0 + px) * ld + g * gs + c] = (TD)(c < C ? ut[c * (UT_PX + 1) + px] : 0.f);
  }
}
}  // namespace

extern "C" int kpf_unstack_rows(const void* src, int src_dtype, float* dst, long rows, int G, int C, int ld, int gs, int hw, void* stream) {
  KPF_REQUIRE(src && dst && rows > 0 && G >= 1 && G <= 4 && C > 0 && gs >= C && ld >= (G - 1) * gs + C && hw >= 0 && (hw == 0 || rows % hw == 0), "kpf_unstack_rows: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hw == 0 && C % 4 == 0 && gs % 4 == 0 && ld % 4 == 0 && kpf_aligned16(dst) && (reinterpret_cast<uintptr_t>(src) & 7u) == 0) {
    const dim3 g4(grid_for(rows * (C / 4) * G));
    if (src_dtype == KPF_DT_F32 && kpf_aligned16(src)) hipLaunchKernelGGL(unstack_rows4_kernel<float>, g4, dim3(256), 0, st, static_cast<const float*>(src), dst, rows, G, C / 4, ld, gs);
    else if (src_dtype == KPF_DT_BF16) hipLaunchKernelGGL(unstack_rows4_kernel<bf16_t>, g4, dim3(256), 0, st, static_cast<const bf16_t*>(src), dst, rows, G, C / 4, ld, gs);
    else if (src_dtype == KPF_DT_F16) hipLaunchKernelGGL(unstack_rows4_kernel<f16_t>, g4, dim3(256), 0, st, static_cast<const f16_t*>(src), dst, rows, G, C / 4, ld, gs);
    else goto scalar_form;
    return kpf_check_launch("kpf_unstack_rows");
  }
scalar_form:
  if (hw > 0 && (size_t)C * (UT_PX + 1) * 4 <= 48 * 1024) {
    const dim3 gt((hw + UT_PX - 1) / UT_PX, (unsigned)(rows / hw), G);
    const size_t lds = (size_t)C * (UT_PX + 1) * 4;
    if (src_dtype == KPF_DT_F32) hipLaunchKernelGGL(unstack_nchw_kernel<float>, gt, dim3(256), lds, st, static_cast<const float*>(src), dst, (int)(rows / hw), C, ld, gs, hw);
    else if (src_dtype == KPF_DT_BF16) hipLaunchKernelGGL(unstack_nchw_kernel<bf16_t>, gt, dim3(256), lds, st, static_cast<const bf16_t*>(src), dst, (int)(rows / hw), C, ld, gs, hw);
    else if (src_dtype == KPF_DT_F16) hipLaunchKernelGGL(unstack_nchw_kernel<f16_t>, gt, dim3(256), lds, st, static_cast<const f16_t*>(src), dst, (int)(rows / hw), C, ld, gs, hw);
    else {
      kpf_set_error("kpf_unstack_rows: unsupported dtype %d", src_dtype);
      return KPF_EINVAL;
    }
    return kpf_check_launch("kpf_unstack_rows");
  }
  const dim3 grid(grid_for(rows * C * G));
  if (src_dtype == KPF_DT_F32) hipLaunchKernelGGL(unstack_rows_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(src), dst, rows, G, C, ld, gs, hw);
  else if (src_dtype == KPF_DT_BF16) hipLaunchKernelGGL(unstack_rows_kernel<bf16_t>, grid, dim3(256), 0, st, static_cast<const bf16_t*>(src), dst, rows, G, C, ld, gs, hw);
  else if (src_dtype == KPF_DT_F16) hipLaunchKernelGGL(unstack_rows_kernel<f16_t>, grid, dim3(256), 0, st, static_cast<const f16_t*>(src), dst, rows, G, C, ld, gs, hw);
  else {
    kpf_set_error("kpf_unstack_rows: unsupported dtype %d", src_dtype);
    return KPF_EINVAL;
  }
  return kpf_check_launch("kpf_unstack_rows");
}

extern "C" int kpf_restack_rows(const float* const* grads, void* dst, int dst_dtype, long rows, int G, int C, int ld, int gs, int hw, void* stream) {
  KPF_REQUIRE(grads && dst && rows > 0 && G >= 1 && G <= 4 && C > 0 && gs >= C && ld >= (G - 1) * gs + C && hw >= 0 && (hw == 0 || rows % hw == 0), "kpf_restack_rows: bad arguments");
  RestackSrc s{};
  for (int g = 0; g < G; ++g) s.g[g] = grads[g];
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  bool al = kpf_aligned16(dst);
  for (int g = 0; g < G; ++g) al = al && kpf_aligned16(grads[g]);
  if (hw == 0 && C % 4 == 0 && gs % 4 == 0 && ld % 4 == 0 && al && dst_dtype >= KPF_DT_F32 && dst_dtype <= KPF_DT_F16) {
    const dim3 g4(grid_for(rows * (ld / 4)));
    if (dst_dtype == KPF_DT_F32) hipLaunchKernelGGL(restack_rows4_kernel<float>, g4, dim3(256), 0, st, s, static_cast<float*>(dst), rows, G, C / 4, ld / 4, gs / 4);
    else if (dst_dtype == KPF_DT_BF16) hipLaunchKernelGGL(restack_rows4_kernel<bf16_t>, g4, dim3(256), 0, st, s, static_cast<bf16_t*>(dst), rows, G, C / 4, ld / 4, gs / 4);
    else hipLaunchKernelGGL(restack_rows4_kernel<f16_t>, g4, dim3(256), 0, st, s, static_cast<f16_t*>(dst), rows, G, C / 4, ld / 4, gs / 4);
    return kpf_check_launch("kpf_restack_rows");
  }
  if (hw > 0 && (size_t)C * (UT_PX + 1) * 4 <= 48 * 1024 && dst_dtype >= KPF_DT_F32 && dst_dtype <= KPF_DT_F16) {
    const dim3 gt((hw + UT_PX - 1) / UT_PX, (unsigned)(rows / hw), G);
    const size_t lds = (size_t)C * (UT_PX + 1) * 4;
    if (dst_dtype == KPF_DT_F32) hipLaunchKernelGGL(restack_nchw_kernel<float>, gt, dim3(256), lds, st, s, static_cast<float*>(dst), C, ld, gs, hw, G);
    else if (dst_dtype == KPF_DT_BF16) hipLaunchKernelGGL(restack_nchw_kernel<bf16_t>, gt, dim3(256), lds, st, s, static_cast<bf16_t*>(dst), C, ld, gs, hw, G);
    else hipLaunchKernelGGL(restack_nchw_kernel<f16_t>, gt, dim3(256), lds, st, s, static_cast<f16_t*>(dst), C, ld, gs, hw, G);
    return kpf_check_launch("kpf_restack_rows");
  }
  const dim3 grid(grid_for(rows * ld));
  if (dst_dtype == KPF_DT_F32) hipLaunchKernelGGL(restack_rows_kernel<float>, grid, dim3(256), 0, st, s, static_cast<float*>(dst), rows, G, C, ld, gs, hw);
  else if (dst_dtype == KPF_DT_BF16) hipLaunchKernelGGL(restack_rows_kernel<bf16_t>, grid, dim3(256), 0, st, s, static_cast<bf16_t*>(dst), rows, G, C, ld, gs, hw);
  else if (dst_dtype == KPF_DT_F16) hipLaunchKernelGGL(restack_rows_kernel<f16_t>, grid, dim3(256), 0, st, s, static_cast<f16_t*>(dst), rows, G, C, ld, gs, hw);
  else {
    kpf_set_error("kpf_restack_rows: unsupported dtype %d", dst_dtype);
    return KPF_EINVAL;
  }
  return kpf_check_launch("kpf_restack_rows");
}

/* The pose tokens of a fusion block (model/model.py:308-316, 417): out[b][n] = [pw[b][n][0..J) | unit offsets (j, xyz) 3J | closeness J | zeros up to ld];
 * pw (the J weight logits sampled at the point, may be NULL: then the row starts with the offsets), joint [B][J][3], pcl [B][N][3]; no gradient (the
 * reference detaches all three).  ld >= 5J (4J without pw), ld - 5J < J. */
extern "C" int kpf_pose_tokens_f32(const float* pw, const float* joint, const float* pcl, float* out, int B, int N, int J, int ld, float kernel, void* stream) {
  const int used = (pw ? 5 : 4) * J;
  KPF_REQUIRE(joint && pcl && out && B > 0 && N > 0 && J > 0 && ld >= used && ld - used < J && kernel > 0.f, "kpf_pose_tokens_f32: bad arguments");
  const long total = (long)B * N * J;
  hipLaunchKernelGGL(pose_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pw, joint, pcl, out, total, N, J,
                     ld, kernel);
  return kpf_check_launch("kpf_pose_tokens_f32");
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: DESA's three radii as ONE channel-stacked tensor through the training step (train_graph.TrainGraph.desa: grouped launches instead of three
// chains of small ones).  (1) the backward of the grouping for all three radii in one launch: d3 [B*R][ld3] holds radius i's gradient of the grouped
// feature differences in columns [128 i, 128 i + 128); `start` / `list` are kpf_row_gather_invert's lists of idx [3][B][R] taken as 3 B images of P = N + J
// source rows (image i * B + b).  dX[b][p] (p < N) = sum_i sum_{entries e of row p} d3[b*R + e][128 i + :] — radius ascending, entries ascending: one fixed
// order — and the joint rows additionally lose the centre's sum: dnode[b][j] = that sum for p = N + j  -  sum_i sum_{m < 64} d3[b*R + 64 j + m][128 i + :].
// One wave per source row, the two half-waves on alternate list positions (even + odd, always in that order), lanes over the 32 channel quads.
// (2) max over the 64 members of a group with the winner's index kept (first maximum), and its gather-form backward (every element of dx written once).
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void ball_group_bwd_kernel(const float* __restrict__ d3, const int* __restrict__ start, const int* __restrict__ list,
                                                             float* __restrict__ dX, float* __restrict__ dnode, int B, int N, int Jn, int ld3) {
  const int lane = threadIdx.x & 63, q = lane & 31, phase = lane >> 5;
  const int P = N + Jn, R = Jn * 64;
  const long rows = (long)B * P;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
    const int b = (int)(r / P), p = (int)(r - (long)b * P);
    const float* db = d3 + (long)b * R * ld3 + 4 * q;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 3; ++i) {
      const long img = (long)i * B + b;
      const int s0 = start[img * (P + 1) + p], s1 = start[img * (P + 1) + p + 1];
      const int* lb = list + img * R;
      for (int c0 = s0; c0 < s1; c0 += 64) {
        const int m = min(64, s1 - c0);
        const int ev = lane < m ? lb[c0 + lane] : 0;
        // (lists are very uneven — the slots a small ball leaves unfilled repeat its first member, so a few rows own hundreds of entries: eight row loads in
        //  flight per half-wave, added in position order, as in row_gather_accum_kernel)
        int k = phase;
        for (; (k - phase) + 15 < m; k += 16) {  // (the same trip count for both half-waves: every shuffle runs with the whole wave active)
          int e[8];
          f32x4 g[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) e[u] = __shfl(ev, k + 2 * u, 64);
#pragma unroll
          for (int u = 0; u < 8; ++u) g[u] = kpf_ld4(db + (long)e[u] * ld3 + 128 * i);
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += g[u][t];
        }
        for (; k < m + phase; k += 2) {
          const int e1 = __shfl(ev, min(k, m - 1), 64);
          if (k < m) {
            const f32x4 g1 = kpf_ld4(db + (long)e1 * ld3 + 128 * i);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += g1[t];
          }
        }
      }
    }
    f32x4 cen = {0.f, 0.f, 0.f, 0.f};
    if (p >= N) {  // (wave-uniform)
      const float* cb = db + (long)(p - N) * 64 * ld3;
      for (int i = 0; i < 3; ++i)
#pragma unroll 8
        for (int m = phase; m < 64; m += 2) {
          const f32x4 g = kpf_ld4(cb + (long)m * ld3 + 128 * i);
#pragma unroll
          for (int t = 0; t < 4; ++t) cen[t] += g[t];
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] += __shfl_down(acc[t], 32, 64);
      cen[t] += __shfl_down(cen[t], 32, 64);
    }
    if (phase == 0) {
      if (p < N) kpf_st4(dX + ((long)b * N + p) * 128 + 4 * q, acc);
      else kpf_st4(dnode + ((long)b * Jn + (p - N)) * 128 + 4 * q, f32x4{acc[0] - cen[0], acc[1] - cen[1], acc[2] - cen[2], acc[3] - cen[3]});
    }
  }
}

__global__ __launch_bounds__(256) void group_max_train_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ arg, long n4,
                                                                  int group, int C4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const int q = (int)(i - r * C4);
    const float* p = x + (r * group * C4 + q) * 4;
    f32x4 best = kpf_ld4(p);
    int bi[4] = {0, 0, 0, 0};
#pragma unroll 8
    for (int m = 1; m < group; ++m) {
      const f32x4 v = kpf_ld4(p + (long)m * C4 * 4);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (v[t] > best[t]) {  // strictly greater: the first maximum wins
          best[t] = v[t];
          bi[t] = m;
        }
    }
    kpf_st4(y + i * 4, best);
    *reinterpret_cast<uchar4*>(arg + i * 4) = uchar4{(unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]};
  }
}
__global__ __launch_bounds__(256) void group_max_train_bwd_kernel(const float* __restrict__ dy, int dy_ld4, const unsigned char* __restrict__ arg,
                                                                  float* __restrict__ dx, long n4, int group, int C4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {  // i over [rows][group][C4]
    const int q = (int)(i % C4);
    const long rm = i / C4;
    const int m = (int)(rm % group);
    const long r = rm / group;
    const f32x4 g = kpf_ld4(dy + (r * dy_ld4 + q) * 4);  // (dy may be a column slice of a wider matrix: the concatenation behind the maximum)
    const uchar4 a = *reinterpret_cast<const uchar4*>(arg + (r * C4 + q) * 4);
    kpf_st4(dx + i * 4, f32x4{a.x == m ? g[0] : 0.f, a.y == m ? g[1] : 0.f, a.z == m ? g[2] : 0.f, a.w == m ? g[3] : 0.f});
  }
}
}  // namespace

extern "C" int kpf_ball_group_bwd_f32(const float* d3, int ld3, const int* start, const int* list, float* dX, float* dnode, int B, int N, int Jn, void* stream) {
  KPF_REQUIRE(d3 && start && list && dX && dnode && B > 0 && N > 0 && Jn > 0 && ld3 >= 384 && ld3 % 4 == 0, "kpf_ball_group_bwd_f32: bad arguments");
  KPF_REQUIRE(kpf_aligned16(d3) && kpf_aligned16(dX) && kpf_aligned16(dnode), "kpf_ball_group_bwd_f32: 16-byte aligned rows");
  const long rows = (long)B * (N + Jn);
  hipLaunchKernelGGL(ball_group_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d3, start, list, dX, dnode, B, N,
                     Jn, ld3);
  return kpf_check_launch("kpf_ball_group_bwd_f32");
}

extern "C" int kpf_group_max_train_forward(const float* x, float* y, unsigned char* arg, long rows, int group, int C, void* stream) {
  KPF_REQUIRE(x && y && arg && rows > 0 && group > 0 && group <= 256 && C > 0 && C % 4 == 0, "kpf_group_max_train_forward: bad arguments (C %% 4 == 0, group <= 256)");
  const long n4 = rows * (C / 4);
  hipLaunchKernelGGL(group_max_train_fwd_kernel, dim3(grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, arg, n4, group, C / 4);
  return kpf_check_launch("kpf_group_max_train_forward");
}
extern "C" int kpf_group_max_train_backward(const float* dy, int dy_ld, const unsigned char* arg, float* dx, long rows, int group, int C, void* stream) {
  KPF_REQUIRE(dy && arg && dx && rows > 0 && group > 0 && group <= 256 && C > 0 && C % 4 == 0 && dy_ld >= C && dy_ld % 4 == 0 && kpf_aligned16(dy),
              "kpf_group_max_train_backward: bad arguments");
  const long n4 = rows * group * (C / 4);
  hipLaunchKernelGGL(group_max_train_bwd_kernel, dim3(grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, dy_ld / 4, arg, dx, n4, group, C / 4);
  return kpf_check_launch("kpf_group_max_train_backward");
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: the embedding sums of a fusion block over ONE channel-stacked tensor y [rows][n * C] (n sibling Linear + BatchNorm branches written side by side,
// one BatchNorm pass over n * C channels: training.LinearCat / TrainGraph.bn_g) —  out = relu(S1)  with  S1 = sum of the first n1 blocks   (n2 == 0), or
// out = relu(relu(S1) + S2)  with  S2 = sum of the next n2 blocks (model/model.py:417-422: relu(feat + xyz + pose), then relu(. + rgb feat)).
// Backward: dy block i = d for i >= n1, d * (S1 > 0) for i < n1, with d = dout * (out > 0); S1 is recomputed from y.  One launch each way.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void slices_sum_relu_fwd_kernel(const float* __restrict__ y, float* __restrict__ out, long n4, int C4, int n1, int n2) {
  const int ld4 = C4 * (n1 + n2);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const int q = (int)(i - r * C4);
    const float* p = y + (r * ld4 + q) * 4;
    f32x4 s1 = kpf_ld4(p);
    for (int k = 1; k < n1; ++k) s1 += kpf_ld4(p + (long)k * C4 * 4);
    f32x4 o = {fmaxf(s1[0], 0.f), fmaxf(s1[1], 0.f), fmaxf(s1[2], 0.f), fmaxf(s1[3], 0.f)};
    if (n2 > 0) {
      for (int k = n1; k < n1 + n2; ++k) o += kpf_ld4(p + (long)k * C4 * 4);
      o = f32x4{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
    }
    kpf_st4(out + i * 4, o);
  }
}
__global__ __launch_bounds__(256) void slices_sum_relu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ y,
                                                                  float* __restrict__ dy, long n4, int C4, int n1, int n2) {
  const int ld4 = C4 * (n1 + n2);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const int q = (int)(i - r * C4);
    const f32x4 g = kpf_ld4(dout + i * 4), o = kpf_ld4(out + i * 4);
    const f32x4 d = {o[0] > 0.f ? g[0] : 0.f, o[1] > 0.f ? g[1] : 0.f, o[2] > 0.f ? g[2] : 0.f, o[3] > 0.f ? g[3] : 0.f};
    f32x4 d1 = d;
    if (n2 > 0) {
      const float* p = y + (r * ld4 + q) * 4;
      f32x4 s1 = kpf_ld4(p);
      for (int k = 1; k < n1; ++k) s1 += kpf_ld4(p + (long)k * C4 * 4);
      d1 = f32x4{s1[0] > 0.f ? d[0] : 0.f, s1[1] > 0.f ? d[1] : 0.f, s1[2] > 0.f ? d[2] : 0.f, s1[3] > 0.f ? d[3] : 0.f};
    }
    float* w = dy + (r * ld4 + q) * 4;
    for (int k = 0; k < n1; ++k) kpf_st4(w + (long)k * C4 * 4, d1);
    for (int k = n1; k < n1 + n2; ++k) kpf_st4(w + (long)k * C4 * 4, d);
  }
}
}  // namespace

extern "C" int kpf_slices_sum_relu_forward(const float* y, float* out, long rows, int C, int n1, int n2, void* stream) {
  KPF_REQUIRE(y && out && rows > 0 && C > 0 && C % 4 == 0 && n1 >= 1 && n2 >= 0 && n1 + n2 <= 8, "kpf_slices_sum_relu_forward: bad arguments (C %% 4 == 0, 1 <= n1, n1 + n2 <= 8)");
  const long n4 = rows * (C / 4);
  hipLaunchKernelGGL(slices_sum_relu_fwd_kernel, dim3(grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, out, n4, C / 4, n1, n2);
  return kpf_check_launch("kpf_slices_sum_relu_forward");
}
extern "C" int kpf_slices_sum_relu_backward(const float* dout, const float* out, const float* y, float* dy, long rows, int C, int n1, int n2, void* stream) {
  KPF_REQUIRE(dout && out && y && dy && rows > 0 && C > 0 && C % 4 == 0 && n1 >= 1 && n2 >= 0 && n1 + n2 <= 8, "kpf_slices_sum_relu_backward: bad arguments");
  const long n4 = rows * (C / 4);
  hipLaunchKernelGGL(slices_sum_relu_bwd_kernel, dim3(grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dout, out, y, dy, n4, C / 4, n1, n2);
  return kpf_check_launch("kpf_slices_sum_relu_backward");
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: y = LayerNorm(h + dropout(o)) in one launch each way — the tail of both halves of a BERT layer (model/model.py:30-126: dense -> dropout ->
// residual add -> LayerNorm) and of the decoder layer's feed-forward.  Same wave-per-row arithmetic as ln_fwd_kernel / ln_bwd_kernel on the sum
// xs = h + dropout(o), which is kept for the backward together with the dropout mask (one byte per element; the mask is drawn from the hash of
// (seed, counter, call, element) like the attention probabilities').  Backward: d h = LayerNorm's input gradient, d o = the same times the mask.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void drop_add_ln_fwd_kernel(const float* __restrict__ o, const float* __restrict__ h, const float* __restrict__ w,
                                                              const float* __restrict__ b, float* __restrict__ xs, float* __restrict__ y,
                                                              unsigned char* __restrict__ mask, float* __restrict__ mean, float* __restrict__ rstd, long rows, int C4,
                                                              float eps, float p_drop, const long* __restrict__ rng, int call_id) {
  const int lane = threadIdx.x & 63;
  const int C = 4 * C4;
  const float invC = 1.0f / (float)C;
  const float ks = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  const unsigned thr = p_drop > 0.f ? (unsigned)fminf(p_drop * 4294967296.0f, 4294967295.0f) : 0u;
  const unsigned seed = rng ? (unsigned)rng[0] : 0u, ctr = rng ? (unsigned)rng[1] : 0u;
  const unsigned base = hash32(seed ^ (ctr * 0x9e3779b9U));
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
    f32x4 v[LN_MAXQ];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (q < C4) {
        const long e0 = r * C + 4 * q;
        const f32x4 ov = kpf_ld4(o + e0), hv = kpf_ld4(h + e0);
        unsigned char m[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m[e] = p_drop > 0.f ? (hash32(base ^ hash32((unsigned)call_id * 0x85ebca6bU + (unsigned)(e0 + e))) >= thr) : 1;
          v[i][e] = hv[e] + (m[e] ? ov[e] * ks : 0.f);
        }
        if (mask) *reinterpret_cast<uchar4*>(mask + e0) = uchar4{m[0], m[1], m[2], m[3]};
        kpf_st4(xs + e0, v[i]);
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mu = wave_sum(s) * invC;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i)
      if (lane + 64 * i < C4)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[i][e] - mu;
          sq = fmaf(d, d, sq);
        }
    const float rs = 1.0f / sqrtf(wave_sum(sq) * invC + eps);
    if (lane == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      if (q < C4) {
        const f32x4 g = kpf_ld4(w + 4 * q), be = kpf_ld4(b + 4 * q);
        f32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = (v[i][e] - mu) * rs * g[e] + be[e];
        kpf_st4(y + r * C + 4 * q, ov);
      }
    }
  }
}

// ln_bwd_kernel on xs with a second output d o = d xs * mask * keep_scale (mask == nullptr: no dropout, d o = d xs); part as ln_bwd_kernel (G = 1)
__global__ __launch_bounds__(256) void drop_add_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ w, const unsigned char* __restrict__ mask,
                                                              float* __restrict__ dx, float* __restrict__ dov, float* __restrict__ part, long rows, int C4, float ks) {
  extern __shared__ float ln_lds[];  // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = 4 * C4;
  const float invC = 1.0f / (float)C;
  f32x4 aw[LN_MAXQ], ab[LN_MAXQ], g[LN_MAXQ];
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    aw[i] = ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    g[i] = lane + 64 * i < C4 ? kpf_ld4(w + 4 * (lane + 64 * i)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const float m = mean[r], rs = rstd[r];
    f32x4 xh[LN_MAXQ], d[LN_MAXQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      xh[i] = d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (q < C4) {
        const f32x4 xv = kpf_ld4(x + r * C + 4 * q);
        d[i] = kpf_ld4(dy + r * C + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xv[e] - m) * rs;
          const float gd = d[i][e] * g[i][e];
          s1 += gd;
          s2 = fmaf(gd, xh[i][e], s2);
          aw[i][e] = fmaf(d[i][e], xh[i][e], aw[i][e]);
          ab[i][e] += d[i][e];
        }
      }
    }
    const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      if (q < C4) {
        const long e0 = r * C + 4 * q;
        f32x4 o, od;
        uchar4 mk = uchar4{1, 1, 1, 1};
        if (mask) mk = *reinterpret_cast<const uchar4*>(mask + e0);
        const unsigned char mm[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = rs * (d[i][e] * g[i][e] - m1 - xh[i][e] * m2);
          od[e] = mm[e] ? o[e] * ks : 0.f;
        }
        kpf_st4(dx + e0, o);
        kpf_st4(dov + e0, od);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    if (q < C4) {
      kpf_st4(ln_lds + (wave * 2 + 0) * C + 4 * q, aw[i]);
      kpf_st4(ln_lds + (wave * 2 + 1) * C + 4 * q, ab[i]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) s += ln_lds[(wv * 2 + which) * C + c];
    part[((long)blockIdx.x * 2 + which) * C + c] = s;
  }
}
}  // namespace

/* y = LayerNorm(h + dropout(o)) over the last axis (fp32 rows of C, C % 4 == 0, C <= 1024): xs = h + dropout(o) and the keep mask (one byte per
 * element; may be NULL when p_drop == 0) are kept for the backward; rng = the device-resident (seed, counter) pair of kpf_attn21_forward. */
extern "C" int kpf_drop_add_ln_forward(const float* o, const float* h, const float* w, const float* b, float* xs, float* y, unsigned char* mask, float* mean,
                                       float* rstd, long rows, int C, float eps, float p_drop, const long* rng, int call_id, void* stream) {
  KPF_REQUIRE(o && h && w && b && xs && y && mean && rstd && rows > 0 && C > 0 && C % 4 == 0 && C <= 256 * LN_MAXQ, "kpf_drop_add_ln_forward: bad arguments (C %% 4 == 0, C <= 1024)");
  KPF_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || (rng && mask)), "kpf_drop_add_ln_forward: dropout needs 0 <= p < 1, the rng state and a mask buffer");
  hipLaunchKernelGGL(drop_add_ln_fwd_kernel, dim3(grid_for(rows, 4, 256 * 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), o, h, w, b, xs, y, mask, mean, rstd,
                     rows, C / 4, eps, p_drop, rng, call_id);
  return kpf_check_launch("kpf_drop_add_ln_forward");
}

/* dh = d xs, d_o = d xs * mask / (1 - p); dw / db as kpf_ln_train_backward (desc != NULL: the column-sum reduce is described, not launched). */
extern "C" int kpf_drop_add_ln_backward(const float* dy, const float* xs, const float* mean, const float* rstd, const float* w, const unsigned char* mask, float* dh,
                                        float* d_o, float* dw, float* db, float* ws, long ws_floats, long rows, int C, float p_drop, kpf_colsum_desc* desc,
                                        void* stream) {
  KPF_REQUIRE(dy && xs && mean && rstd && w && dh && d_o && dw && db && ws && rows > 0 && C > 0 && C % 4 == 0 && C <= 256 * LN_MAXQ, "kpf_drop_add_ln_backward: bad arguments");
  KPF_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || mask), "kpf_drop_add_ln_backward: dropout needs the mask");
  const int nblk = ln_blocks(rows);
  KPF_REQUIRE(ws_floats >= (long)nblk * 2 * C, "kpf_drop_add_ln_backward: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(drop_add_ln_bwd_kernel, dim3(nblk), dim3(256), (size_t)8 * C * sizeof(float), st, dy, xs, mean, rstd, w, p_drop > 0.f ? mask : nullptr, dh, d_o, ws,
                     rows, C / 4, p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f);
  int rc = kpf_check_launch("kpf_drop_add_ln_backward");
  if (rc != KPF_OK) return rc;
  if (desc) {
    desc->part = ws, desc->dw = dw, desc->db = db, desc->nblk = nblk, desc->C = C, desc->first_block = 0, desc->reserved = 0;
    return KPF_OK;
  }
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * C + 63) / 64), dim3(512), 0, st, ws, dw, db, nblk, C);
  return kpf_check_launch("kpf_drop_add_ln_backward (reduce)");
}

// ---------------------------------------------------------------------------------------------------------------
// out = relu(scale * (a + b [+ c])) and its backward d = (out > 0) ? scale * dy : 0 (the same tensor for every addend): the embedding sums of a
// fusion block (model/model.py:417-422) and DESA's relu(loc + feat) (model/model.py:190) — two or three library launches each way otherwise.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void add_relu_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ out,
                                                           long n4, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = kpf_ld4(a + 4 * i);
    if (b) v = v + kpf_ld4(b + 4 * i);
    if (c) v = v + kpf_ld4(c + 4 * i);
    kpf_st4(out + 4 * i, f32x4{fmaxf(v[0] * scale, 0.f), fmaxf(v[1] * scale, 0.f), fmaxf(v[2] * scale, 0.f), fmaxf(v[3] * scale, 0.f)});
  }
}
__global__ __launch_bounds__(256) void relu_scale_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out, float* __restrict__ dx, long n4, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 g = kpf_ld4(dy + 4 * i), o = kpf_ld4(out + 4 * i);
    kpf_st4(dx + 4 * i, f32x4{o[0] > 0.f ? g[0] * scale : 0.f, o[1] > 0.f ? g[1] * scale : 0.f, o[2] > 0.f ? g[2] * scale : 0.f, o[3] > 0.f ? g[3] * scale : 0.f});
  }
}
}  // namespace

extern "C" int kpf_add_relu_forward(const float* a, const float* b, const float* c, float* out, long n, float scale, void* stream) {
  KPF_REQUIRE(a && out && n > 0 && n % 4 == 0 && (b || !c), "kpf_add_relu_forward: bad arguments (n %% 4 == 0; c only with b)");
  hipLaunchKernelGGL(add_relu_fwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b, c, out, n / 4, scale);
  return kpf_check_launch("kpf_add_relu_forward");
}
extern "C" int kpf_add_relu_backward(const float* dy, const float* out, float* dx, long n, float scale, void* stream) {
  KPF_REQUIRE(dy && out && dx && n > 0 && n % 4 == 0, "kpf_add_relu_backward: bad arguments (n %% 4 == 0)");
  hipLaunchKernelGGL(relu_scale_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, out, dx, n / 4, scale);
  return kpf_check_launch("kpf_add_relu_backward");
}

// ---------------------------------------------------------------------------------------------------------------
// The gate of a fusion block (model/model.py:334-341) around its two maps: sw = sigmoid(logits), g = sigmoid(weight_dis) * gam + (1 - sigmoid(weight_dis)) * sw,
// gw = g * w_fc[p] — one launch forward (sigmoid x 2, 1 - x, two products, a sum, the product with the pooling weight and the [B, P, J] -> [B, J, P]
// transpose of the logits otherwise), two backward (element-wise part + per-row partial sums; then the two parameter reductions in a fixed order).
// logits [B*P][J] (rows of the atten_spatial GEMM), gam / sw / gw [B][J][P]; weight_dis a scalar parameter, w_fc [P].
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void gate_mix_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ gam, const float* __restrict__ wdis,
                                                           const float* __restrict__ wfc, float* __restrict__ sw, float* __restrict__ gw, int J, int P) {
  const int j = blockIdx.x, b = blockIdx.y;
  const float wd = 1.0f / (1.0f + __expf(-wdis[0]));
  const long row = ((long)b * J + j) * P;
  for (int p = threadIdx.x; p < P; p += 256) {
    const float s = 1.0f / (1.0f + __expf(-logits[((long)b * P + p) * J + j]));
    sw[row + p] = s;
    gw[row + p] = (wd * gam[row + p] + (1.0f - wd) * s) * wfc[p];
  }
}

__global__ __launch_bounds__(256) void gate_mix_bwd_kernel(const float* __restrict__ sw, const float* __restrict__ gam, const float* __restrict__ wdis,
                                                           const float* __restrict__ wfc, const float* __restrict__ dsw, const float* __restrict__ dgw,
                                                           float* __restrict__ dgam, float* __restrict__ dlogits, float* __restrict__ part, int J, int P) {
  __shared__ float red[4];
  const int j = blockIdx.x, b = blockIdx.y;
  const float wd = 1.0f / (1.0f + __expf(-wdis[0]));
  const long row = ((long)b * J + j) * P;
  float acc = 0.f;
  for (int p = threadIdx.x; p < P; p += 256) {
    const float t = dgw[row + p] * wfc[p];  // d g
    const float s = sw[row + p], ga = gam[row + p];
    dgam[row + p] = t * wd;
    const float ds = (dsw ? dsw[row + p] : 0.f) + t * (1.0f - wd);
    dlogits[((long)b * P + p) * J + j] = ds * s * (1.0f - s);
    acc = fmaf(t, ga - s, acc);
  }
  acc = block_sum256(acc, red);
  if (threadIdx.x == 0) part[(long)b * J + j] = acc;
}

// blocks [0, ceil(P / 8)): d w_fc[p] = sum over the B*J rows of dgw * g — 8 columns x 32 row groups per block (rows r = group, group + 32, ...: 21 iterations at
// B = 32; round 6: 32 columns x 8 groups before — 33 workgroups walking 84 iterations each took 28 us for 8 MB), the 32 group sums added in group order;  the last
// block: d weight_dis = wd (1 - wd) * sum of the row partials
__global__ __launch_bounds__(256) void gate_mix_params_kernel(const float* __restrict__ sw, const float* __restrict__ gam, const float* __restrict__ wdis,
                                                              const float* __restrict__ dgw, const float* __restrict__ part, float* __restrict__ dwfc,
                                                              float* __restrict__ dwdis, int R, int P) {
  __shared__ float red[32][8];
  const float wd = 1.0f / (1.0f + __expf(-wdis[0]));
  const int nb = (P + 7) / 8;
  if ((int)blockIdx.x < nb) {
    const int pl = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const int p = blockIdx.x * 8 + pl;
    float acc = 0.f;
    if (p < P)
      for (int r = rg; r < R; r += 32) {
        const long i = (long)r * P + p;
        acc = fmaf(dgw[i], wd * gam[i] + (1.0f - wd) * sw[i], acc);
      }
    red[rg][pl] = acc;
    __syncthreads();
    if (rg == 0 && p < P) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += red[k][pl];
      dwfc[p] = t;
    }
  } else {
    float acc = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) acc += part[r];
    acc = block_sum256(acc, &red[0][0]);
    if (threadIdx.x == 0) dwdis[0] = acc * wd * (1.0f - wd);
  }
}
}  // namespace

extern "C" int kpf_gate_mix_forward(const float* logits, const float* gam, const float* weight_dis, const float* w_fc, float* sw, float* gw, int B, int J, int P,
                                    void* stream) {
  KPF_REQUIRE(logits && gam && weight_dis && w_fc && sw && gw && B > 0 && J > 0 && P > 0, "kpf_gate_mix_forward: bad arguments");
  hipLaunchKernelGGL(gate_mix_fwd_kernel, dim3(J, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits, gam, weight_dis, w_fc, sw, gw, J, P);
  return kpf_check_launch("kpf_gate_mix_forward");
}
/* d_sw may be NULL; ws >= B*J floats; writes d_gam [B][J][P], d_logits [B*P][J], d_w_fc [P], d_weight_dis [1] */
extern "C" int kpf_gate_mix_backward(const float* sw, const float* gam, const float* weight_dis, const float* w_fc, const float* d_sw, const float* d_gw, float* d_gam,
                                     float* d_logits, float* d_w_fc, float* d_weight_dis, float* ws, int B, int J, int P, void* stream) {
  KPF_REQUIRE(sw && gam && weight_dis && w_fc && d_gw && d_gam && d_logits && d_w_fc && d_weight_dis && ws && B > 0 && J > 0 && P > 0, "kpf_gate_mix_backward: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gate_mix_bwd_kernel, dim3(J, B), dim3(256), 0, st, sw, gam, weight_dis, w_fc, d_sw, d_gw, d_gam, d_logits, ws, J, P);
  int rc = kpf_check_launch("kpf_gate_mix_backward");
  if (rc != KPF_OK) return rc;
  hipLaunchKernelGGL(gate_mix_params_kernel, dim3((P + 7) / 8 + 1), dim3(256), 0, st, sw, gam, weight_dis, d_gw, ws, d_w_fc, d_weight_dis, B * J, P);
  return kpf_check_launch("kpf_gate_mix_backward (parameters)");
}
