// Geometry / point-cloud kernels of the keypoint-fusion head.  All are HBM- or latency-bound
// gathers, reductions and small searches: one wave (or one workgroup) per joint / point / pixel tile, wave64 shuffles
// for reductions, no materialised B x N x F^2 intermediates.
//
// Floating-point contraction is OFF in this file: the index-producing comparisons (top-4 nearest pixels, ball-query
// membership) must see squared distances rounded like the oracle's (products rounded individually, left-to-right adds).
#include "kpf_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int J = 21;

struct Cam {  // per-sample crop/camera constants
  float mi00, mi01, mi02, mi10, mi11, mi12;  // first two rows of M^-1
  float cx, cy, cz, hx, hy, hz;              // centre (mm), half cube (mm)
  float fx, fy, u0, v0;
};

// M^-1 with the operation order of the reference's torch.linalg.inv on the CPU (dataloader/loader.py:781), so that the pixel positions
// -- and with them the integer top-4 / ball-query decisions -- come out bit for bit like the reference's.  ATen solves A X = I by
// factoring the row-major matrix as its transpose (no copy) and calling getrs with trans = 'T' (linalg_solve_ex: use_A_T); MKL's
// 3x3 path was pinned empirically (tools/mkl_inv_probe.py, tests/test_inv3x3.py: 10^4 crop matrices, bitwise): partial-pivot LU of
// A^T with column 0 scaled by the reciprocal of the pivot, column 1 by a true division; U^T y = e with reciprocal diagonals;
// x1 = y1 - l21 x2; x0 = y0 - (l10 x1 + l20 x2); row interchanges undone.  FUSED = MKL's FMA code path (Intel hosts): every c - a*b
// is one fused operation and x0 = y0 - fma(l10, x1, l20*x2); otherwise (its generic path, AMD hosts) every product and sum is
// rounded separately.  keypointfusion_amd/inv3x3.py finds out which one the host's library takes.
template <bool FUSED>
__device__ __forceinline__ float msub(float c, float a, float b) {  // c - a*b
  return FUSED ? __builtin_fmaf(-a, b, c) : c - a * b;  // (contraction is off in this file: the second form rounds twice)
}
template <bool FUSED>
__device__ __forceinline__ void inv3x3_lapack_order(const float* __restrict__ m, float (&inv)[3][3]) {
  // rows of A^T = columns of A
  float r0x = m[0], r0y = m[3], r0z = m[6];
  float r1x = m[1], r1y = m[4], r1z = m[7];
  float r2x = m[2], r2y = m[5], r2z = m[8];
  int p0 = 0;
  float best = fabsf(r0x);
  if (fabsf(r1x) > best) { best = fabsf(r1x); p0 = 1; }
  if (fabsf(r2x) > best) p0 = 2;
#define KPF_SWAP3(a, b) do { float t_; t_ = a##x; a##x = b##x; b##x = t_; t_ = a##y; a##y = b##y; b##y = t_; t_ = a##z; a##z = b##z; b##z = t_; } while (0)
  if (p0 == 1) KPF_SWAP3(r0, r1);
  if (p0 == 2) KPF_SWAP3(r0, r2);
  const float rc = 1.0f / r0x;
  r1x = r1x * rc;  // l10
  r2x = r2x * rc;  // l20
  r1y = msub<FUSED>(r1y, r1x, r0y);
  r1z = msub<FUSED>(r1z, r1x, r0z);
  r2y = msub<FUSED>(r2y, r2x, r0y);
  r2z = msub<FUSED>(r2z, r2x, r0z);
  const int p1 = fabsf(r2y) > fabsf(r1y) ? 2 : 1;
  if (p1 == 2) KPF_SWAP3(r1, r2);
#undef KPF_SWAP3
  const float l10 = r1x, l20 = r2x;
  const float l21 = r2y / r1y;
  r2z = msub<FUSED>(r2z, l21, r1z);
  const float rd0 = 1.0f / r0x, rd1 = 1.0f / r1y, rd2 = 1.0f / r2z;
  float x[3][3];  // x[row][rhs column]
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float b0 = c == 0 ? 1.f : 0.f, b1 = c == 1 ? 1.f : 0.f, b2 = c == 2 ? 1.f : 0.f;
    const float y0 = b0 * rd0;
    b1 = b1 - y0 * r0y;
    b2 = b2 - y0 * r0z;
    const float y1 = b1 * rd1;
    b2 = b2 - y1 * r1z;
    const float y2 = b2 * rd2;
    const float x2 = y2;
    const float x1 = msub<FUSED>(y1, x2, l21);
    const float p2 = l20 * x2;
    const float x0 = y0 - (FUSED ? __builtin_fmaf(l10, x1, p2) : l10 * x1 + p2);
    x[0][c] = x0;
    x[1][c] = x1;
    x[2][c] = x2;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {  // undo the row interchanges, last first
    float a0 = x[0][c], a1 = x[1][c], a2 = x[2][c], t;
    if (p1 == 2) { t = a1; a1 = a2; a2 = t; }
    if (p0 == 1) { t = a0; a0 = a1; a1 = t; }
    if (p0 == 2) { t = a0; a0 = a2; a2 = t; }
    inv[0][c] = a0;
    inv[1][c] = a1;
    inv[2][c] = a2;
  }
}

__device__ __forceinline__ Cam load_cam(const float* center, const float* Minv, const float* cube, const float* cam, int b) {
  Cam c;
  const float* mi = Minv + 9 * b;  // M^-1 from kpf_inv3x3_f32 (or the host's torch.linalg.inv): only its first two rows are used
  c.mi00 = mi[0];
  c.mi01 = mi[1];
  c.mi02 = mi[2];
  c.mi10 = mi[3];
  c.mi11 = mi[4];
  c.mi12 = mi[5];
  c.cx = center[3 * b];
  c.cy = center[3 * b + 1];
  c.cz = center[3 * b + 2];
  c.hx = cube[3 * b] / 2.0f;
  c.hy = cube[3 * b + 1] / 2.0f;
  c.hz = cube[3 * b + 2] / 2.0f;
  c.fx = cam[4 * b];
  c.fy = cam[4 * b + 1];
  c.u0 = cam[4 * b + 2];
  c.v0 = cam[4 * b + 3];
  return c;
}

// dataloader/loader.py:775-789: normalised uvd -> normalised xyz (un-crop through M^-1, pinhole back-projection)
__device__ __forceinline__ void uvd2xyz(const Cam& c, float u, float v, float d, float half_img, float flip, float& x, float& y,
                                        float& z) {
  const float up = (u + 1.0f) * half_img, vp = (v + 1.0f) * half_img;
  const float dm = d * c.hz + c.cz;
  const float tu = (c.mi00 * up + c.mi01 * vp) + c.mi02;
  const float tv = (c.mi10 * up + c.mi11 * vp) + c.mi12;
  const float X = (tu - c.u0) * dm / c.fx;
  const float Y = flip * (tv - c.v0) * dm / c.fy;
  x = (X - c.cx) / c.hx;
  y = (Y - c.cy) / c.hy;
  z = (dm - c.cz) / c.hz;
}

__device__ __forceinline__ float block_sum(float v, float* red) {  // 256 threads
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// M^-1 for a batch of crop matrices: computed once per forward, read by every geometry kernel below.
template <bool FUSED>
__global__ __launch_bounds__(64) void inv3x3_kernel(const float* __restrict__ M, float* __restrict__ Minv, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float inv[3][3];
  inv3x3_lapack_order<FUSED>(M + 9 * b, inv);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) Minv[9 * b + 3 * i + k] = inv[i][k];
}

// ---------------------------------------------------------------------------------------------------------------
// a7 + a8: masked soft-argmax decode (model/model.py:466-500) and uvd -> xyz.  grid (J, B), 256 threads.
// offset: NCHW [B][105][F*F]; depth: [B][S][S] (nearest-downsampled on the fly).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void offset2joint_kernel(const float* __restrict__ offset, const float* __restrict__ depth,
                                                           const float* __restrict__ center, const float* __restrict__ Minv,
                                                           const float* __restrict__ cube, const float* __restrict__ cam,
                                                           float* __restrict__ joint_uvd, float* __restrict__ joint_xyz, int S, int F,
                                                           float kernel, float half_img, float flip) {
  __shared__ float red[4];
  const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int P = F * F;
  const float* ob = offset + (long)b * 5 * J * P;
  const float* wp = ob + (4 * J + j) * P;
  const float* hp = ob + (3 * J + j) * P;
  const float* up = ob + (3 * j) * P;
  float lmax = -INFINITY;
  for (int p = tid; p < P; p += 256) {
    const int py = p / F, px = p - py * F;
    const float d = depth[(long)b * S * S + (long)((py * S) / F) * S + (px * S) / F];
    const float w = d > 0.99f ? -1e8f : wp[p];
    lmax = fmaxf(lmax, w);
  }
  const float mx = block_max(lmax, red);
  float se = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int p = tid; p < P; p += 256) {
    const int py = p / F, px = p - py * F;
    const float d = depth[(long)b * S * S + (long)((py * S) / F) * S + (px * S) / F];
    const float w = d > 0.99f ? -1e8f : wp[p];
    const float e = expf(w - mx);
    const float mask = d < 0.99f ? 1.f : 0.f;
    const float dist = kernel - (hp[p] * mask) * kernel;
    const float cu = 2.0f * ((float)px + 0.5f) / (float)F - 1.0f;
    const float cv = 2.0f * ((float)py + 0.5f) / (float)F - 1.0f;
    se += e;
    s0 += ((up[p] * mask) * dist + cu) * e;
    s1 += ((up[P + p] * mask) * dist + cv) * e;
    s2 += ((up[2 * P + p] * mask) * dist + d) * e;
  }
  se = block_sum(se, red);
  s0 = block_sum(s0, red);
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (tid == 0) {
    const float u = s0 / se, v = s1 / se, d = s2 / se;
    float* o = joint_uvd + ((long)b * J + j) * 3;
    o[0] = u;
    o[1] = v;
    o[2] = d;
    const Cam c = load_cam(center, Minv, cube, cam, b);
    float x, y, z;
    uvd2xyz(c, u, v, d, half_img, flip, x, y, z);
    float* q = joint_xyz + ((long)b * J + j) * 3;
    q[0] = x;
    q[1] = y;
    q[2] = z;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// a9: per point the 4 nearest feature pixels (dataloader/loader.py:936-967) without materialising B x N x F^2.
// grid (ceil(N/32), B): 32 points per workgroup, 8 lanes per point; the F*F pixel positions are rebuilt in LDS per workgroup (12 KB at F=32).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void img2pcl_top4_kernel(const float* __restrict__ pcl, const float* __restrict__ depth,
                                                           const float* __restrict__ center, const float* __restrict__ Minv,
                                                           const float* __restrict__ cube, const float* __restrict__ cam,
                                                           float* __restrict__ closeness, int* __restrict__ index,
                                                           float* __restrict__ img_xyz_out, int N, int S, int F, float half_img,
                                                           float flip) {
  extern __shared__ __attribute__((aligned(16))) float pix[];  // [P][3]
  const int b = blockIdx.y, tid = threadIdx.x;
  const int P = F * F;
  const Cam c = load_cam(center, Minv, cube, cam, b);
  for (int p = tid; p < P; p += 256) {
    const int py = p / F, px = p - py * F;
    const float d = depth[(long)b * S * S + (long)((py * S) / F) * S + (px * S) / F];
    const float cu = 2.0f * ((float)px + 0.5f) / (float)F - 1.0f;
    const float cv = 2.0f * ((float)py + 0.5f) / (float)F - 1.0f;
    float x, y, z;
    uvd2xyz(c, cu, cv, d, half_img, flip, x, y, z);
    pix[3 * p] = x;
    pix[3 * p + 1] = y;
    pix[3 * p + 2] = z;
    if (img_xyz_out && blockIdx.x == 0) {
      float* o = img_xyz_out + ((long)b * P + p) * 3;
      o[0] = x;
      o[1] = y;
      o[2] = z;
    }
  }
  __syncthreads();
  // 8 lanes per point: lane s scans the pixels p = s, s+8, s+16, .. (neighbouring lanes read neighbouring LDS words) with a branch-free
  // sorted insert, skipped for the whole wave while no lane has a candidate below its 4th best; the 8 sorted lists are then merged by
  // lane 0 of the group with the comparison (distance, pixel index) — so the result is the list sorted by distance with the lower
  // index first among equal distances, exactly what a sequential scan with a strict < produces.
  const int sub = tid & 7;
  const int n = blockIdx.x * 32 + (tid >> 3);
  const int nc = n < N ? n : N - 1;  // (idle groups still take part in the shuffles)
  const float* q = pcl + ((long)b * N + nc) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY, d3 = INFINITY;
  int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff, i3 = 0x7fffffff;
  auto insert = [&](float dd, int p) {  // (dd, p) < (d_k, i_k) lexicographically
    const bool c0 = dd < d0 || (dd == d0 && p < i0), c1 = dd < d1 || (dd == d1 && p < i1);
    const bool c2 = dd < d2 || (dd == d2 && p < i2), c3 = dd < d3 || (dd == d3 && p < i3);
    d3 = c2 ? d2 : (c3 ? dd : d3);
    i3 = c2 ? i2 : (c3 ? p : i3);
    d2 = c1 ? d1 : (c2 ? dd : d2);
    i2 = c1 ? i1 : (c2 ? p : i2);
    d1 = c0 ? d0 : (c1 ? dd : d1);
    i1 = c0 ? i0 : (c1 ? p : i1);
    d0 = c0 ? dd : d0;
    i0 = c0 ? p : i0;
  };
  for (int p = sub; p < P; p += 8) {
    const float dx = qx - pix[3 * p], dy = qy - pix[3 * p + 1], dz = qz - pix[3 * p + 2];
    const float dd = (dx * dx + dy * dy) + dz * dz;
    if (__builtin_amdgcn_ballot_w64(dd < d3) != 0) {  // wave-uniform: within a lane p only grows, so an equal distance never enters
      const bool c0 = dd < d0, c1 = dd < d1, c2 = dd < d2, c3 = dd < d3;
      d3 = c2 ? d2 : (c3 ? dd : d3);
      i3 = c2 ? i2 : (c3 ? p : i3);
      d2 = c1 ? d1 : (c2 ? dd : d2);
      i2 = c1 ? i1 : (c2 ? p : i2);
      d1 = c0 ? d0 : (c1 ? dd : d1);
      i1 = c0 ? i0 : (c1 ? p : i1);
      d0 = c0 ? dd : d0;
      i0 = c0 ? p : i0;
    }
  }
  const float m0 = d0, m1 = d1, m2 = d2, m3 = d3;
  const int j0 = i0, j1 = i1, j2 = i2, j3 = i3;
  const int base = (tid & 63) & ~7;
#pragma unroll
  for (int s2 = 1; s2 < 8; ++s2) {
    const float e0 = __shfl(m0, base + s2, 64), e1 = __shfl(m1, base + s2, 64), e2 = __shfl(m2, base + s2, 64), e3 = __shfl(m3, base + s2, 64);
    const int k0 = __shfl(j0, base + s2, 64), k1 = __shfl(j1, base + s2, 64), k2 = __shfl(j2, base + s2, 64), k3 = __shfl(j3, base + s2, 64);
    if (sub == 0) {
      insert(e0, k0);
      insert(e1, k1);
      insert(e2, k2);
      insert(e3, k3);
    }
  }
  if (sub != 0 || n >= N) return;
  const float c0 = 1.0f / (d0 + 1e-8f), c1 = 1.0f / (d1 + 1e-8f), c2 = 1.0f / (d2 + 1e-8f), c3 = 1.0f / (d3 + 1e-8f);
  const float cs = (((c0 + c1) + c2) + c3) + 1e-8f;
  float* co = closeness + ((long)b * N + n) * 4;
  int* io = index + ((long)b * N + n) * 4;
  co[0] = c0 / cs; co[1] = c1 / cs; co[2] = c2 / cs; co[3] = c3 / cs;
  io[0] = i0; io[1] = i1; io[2] = i2; io[3] = i3;
}

// ---------------------------------------------------------------------------------------------------------------
// a10 + a11 (gather half): one wave per point builds the two operand rows of the point-embedding GEMMs:
//   A1[b,n] = [ pf(128) | pcl xyz(3) | pw(21) | unit offsets (63) | closeness (21) | 0 0 0 0 ]   (240 floats)
//   A2[b,n] = pf_rgb(128)
// pf = sum_k clos_k * feat[b, idx_k, :] (model/model.py:297-301), pw from the weight-logit planes (:304-306),
// offsets/closeness = pcl_joint2offset (:503-525).
// ---------------------------------------------------------------------------------------------------------------
constexpr int A1_LD = 240;

__global__ __launch_bounds__(256) void point_assemble_kernel(const float* __restrict__ feat_d, const float* __restrict__ feat_rgb,
                                                             const float* __restrict__ offset, const float* __restrict__ pcl,
                                                             const float* __restrict__ joint_xyz, const float* __restrict__ closeness,
                                                             const int* __restrict__ index, float* __restrict__ A1,
                                                             float* __restrict__ A2, int N, int P, float kernel) {
  const int lane = threadIdx.x & 63;
  const long pt = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // global point id b*N+n
  const int b = (int)(pt / N);
  const float* cl = closeness + pt * 4;
  const int* ix = index + pt * 4;
  const float c0 = cl[0], c1 = cl[1], c2 = cl[2], c3 = cl[3];
  const int i0 = ix[0], i1 = ix[1], i2 = ix[2], i3 = ix[3];
  float* r1 = A1 + pt * A1_LD;
  float* r2 = A2 + pt * 128;
  {
    const float* f = feat_d + (long)b * P * 128 + 2 * lane;
    const float2 v0 = *reinterpret_cast<const float2*>(f + (long)i0 * 128), v1 = *reinterpret_cast<const float2*>(f + (long)i1 * 128);
    const float2 v2 = *reinterpret_cast<const float2*>(f + (long)i2 * 128), v3 = *reinterpret_cast<const float2*>(f + (long)i3 * 128);
    float2 o;
    o.x = ((v0.x * c0 + v1.x * c1) + v2.x * c2) + v3.x * c3;
    o.y = ((v0.y * c0 + v1.y * c1) + v2.y * c2) + v3.y * c3;
    *reinterpret_cast<float2*>(r1 + 2 * lane) = o;
  }
  {
    const float* f = feat_rgb + (long)b * P * 128 + 2 * lane;
    const float2 v0 = *reinterpret_cast<const float2*>(f + (long)i0 * 128), v1 = *reinterpret_cast<const float2*>(f + (long)i1 * 128);
    const float2 v2 = *reinterpret_cast<const float2*>(f + (long)i2 * 128), v3 = *reinterpret_cast<const float2*>(f + (long)i3 * 128);
    float2 o;
    o.x = ((v0.x * c0 + v1.x * c1) + v2.x * c2) + v3.x * c3;
    o.y = ((v0.y * c0 + v1.y * c1) + v2.y * c2) + v3.y * c3;
    *reinterpret_cast<float2*>(r2 + 2 * lane) = o;
  }
  const float* q = pcl + pt * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  if (lane < 3) r1[128 + lane] = q[lane];
  if (lane < J) {
    const float* w = offset + ((long)b * 5 * J + 4 * J + lane) * P;
    r1[131 + lane] = ((w[i0] * c0 + w[i1] * c1) + w[i2] * c2) + w[i3] * c3;
    const float* jp = joint_xyz + ((long)b * J + lane) * 3;
    const float ox = jp[0] - qx, oy = jp[1] - qy, oz = jp[2] - qz;
    const float dis = sqrtf((ox * ox + oy * oy) + oz * oz);
    const float inv = dis + 1e-8f;
    const float cls = (kernel - dis) / kernel;
    const float mask = (cls >= 0.f ? 1.f : 0.f) * (qz < 0.99f ? 1.f : 0.f);
    r1[152 + 3 * lane] = (ox / inv) * mask;
    r1[152 + 3 * lane + 1] = (oy / inv) * mask;
    r1[152 + 3 * lane + 2] = (oz / inv) * mask;
    r1[152 + 3 * J + lane] = cls * mask;
  }
  if (lane < 4) r1[236 + lane] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// a11 (pool half): attention = softmax over the N points of the per-joint weight, joint_feat = attention @ pcl_feat
// (model/model.py:319-320).  grid (J, B), 256 threads.  Output row of JA[b*J+j] = [joint_feat(128) | joint xyz(3) | 0].
// ---------------------------------------------------------------------------------------------------------------
constexpr int JA_LD = 132;

__global__ __launch_bounds__(256) void softmax_pool_kernel(const float* __restrict__ A1, const float* __restrict__ X,
                                                           const float* __restrict__ joint_xyz, float* __restrict__ JA, int N) {
  // grid (3, B): 7 joints per workgroup, so the point features X[b] are streamed 3 times instead of 21.  The seven softmaxes share
  // their block reductions (two barriers for the maxima, two for the sums); the pooling loop is split over 8 point groups x 32
  // channel quads (float4 rows, 128 dependent iterations instead of 512) and combined through LDS in a fixed order.
  extern __shared__ __attribute__((aligned(16))) float att[];  // [7][N] | red[4][8] | part[8][7][128]
  float* red = att + 7 * N;
  float* part = red + 32;
  const int jc = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* wbase = A1 + (long)b * N * A1_LD + 131 + jc * 7;
  float mx[7], inv[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) mx[j] = -INFINITY;
  for (int n = tid; n < N; n += 256) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const float w = wbase[(long)n * A1_LD + j];
      att[j * N + n] = w;
      mx[j] = fmaxf(mx[j], w);
    }
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const float m = wave_max(mx[j]);
    if (lane == 0) red[wave * 8 + j] = m;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 7; ++j) mx[j] = fmaxf(fmaxf(red[j], red[8 + j]), fmaxf(red[16 + j], red[24 + j]));
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 7; ++j) inv[j] = 0.f;
  for (int n = tid; n < N; n += 256) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const float e = expf(att[j * N + n] - mx[j]);
      att[j * N + n] = e;
      inv[j] += e;
    }
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const float sm = wave_sum(inv[j]);
    if (lane == 0) red[wave * 8 + j] = sm;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 7; ++j) inv[j] = 1.0f / ((red[j] + red[8 + j]) + (red[16 + j] + red[24 + j]));
  const int cq = tid & 31, ng = tid >> 5;
  const float* xb = X + (long)b * N * 128 + 4 * cq;
  f32x4 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // 16 rows in flight per thread: the loop is latency-bound on the row loads (X[b] is read from L2 / HBM once per joint group)
  int n = ng;
  for (; n + 8 * 15 < N; n += 8 * 16) {
    f32x4 xv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) xv[u] = *reinterpret_cast<const f32x4*>(xb + (long)(n + 8 * u) * 128);
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[j] += att[j * N + n + 8 * u] * xv[u];
  }
  for (; n < N; n += 8) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + (long)n * 128);
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j] += att[j * N + n] * xv;
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) *reinterpret_cast<f32x4*>(part + (ng * 7 + j) * 128 + 4 * cq) = acc[j];
  __syncthreads();
  for (int o = tid; o < 7 * 128; o += 256) {
    const int j = o >> 7, c = o & 127;
    float sacc = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) sacc += part[(g * 7 + j) * 128 + c];
    float scale = inv[0];
#pragma unroll
    for (int jj = 1; jj < 7; ++jj) scale = j == jj ? inv[jj] : scale;
    JA[((long)b * J + jc * 7 + j) * JA_LD + c] = sacc * scale;
  }
  if (tid < 28) {
    const int j = tid >> 2, e = tid & 3;
    JA[((long)b * J + jc * 7 + j) * JA_LD + 128 + e] = e < 3 ? joint_xyz[((long)b * J + jc * 7 + j) * 3 + e] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// a12 (search + gather half): pointnet2 ball query + grouping, one wave per (radius, joint).
// Points = cat(pcl [N], joints [J]); features = cat(X [N][128], JF [J][128]).  For each query joint: first 64 point
// indices in index order with d^2 < r^2 (wave ballot + prefix popcount keeps index order), unfilled slots repeat the first
// hit.  Writes the GEMM operand rows  G[r][(b*J+j)*64 + s] = [ feat[idx]-feat_j (128) | (xyz[idx]-xyz_j)/r (3) | 0 ].
// grid (B*J, 3).
// ---------------------------------------------------------------------------------------------------------------
constexpr int G_LD = 132;

__global__ __launch_bounds__(64) void ball_group_kernel(const float* __restrict__ pcl, const float* __restrict__ joint_xyz,
                                                        const float* __restrict__ X, const float* __restrict__ JF,
                                                        float* __restrict__ GF, float* __restrict__ GX, int* __restrict__ idx_out, int N, int jf_ld,
                                                        long gf_gs, int gf_ld, long gx_gs, int gx_ld, float r0, float r1, float r2) {
  // output layout: radius ri's grouped feature differences go to GF + ri * gf_gs with row stride gf_ld, its scaled offsets (3 + a zero channel) to
  // GX + ri * gx_gs with row stride gx_ld.  kpf_ball_group_f32: [3][rows][132] = [feat 128 | xyz 3 | 0] (gf_ld = gx_ld = 132, GX = GF + 128);
  // kpf_ball_group_stacked_f32: the three radii CHANNEL-STACKED, [rows][3 * 128] and [rows][3 * 4] (the grouped launches of the training step).
  __shared__ int sidx[64];
  const int lane = threadIdx.x;
  const int bj = blockIdx.x, ri = blockIdx.y;
  const int b = bj / J;
  const float radius = ri == 0 ? r0 : (ri == 1 ? r1 : r2);
  const float rad2 = radius * radius;
  const float* qp = joint_xyz + (long)bj * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const int NT = N + J;
  int cnt = 0;
  sidx[lane] = 0;
  __syncthreads();
  // the distance tests of eight 64-point chunks are requested together, then the chunks are taken in index order (the first 64 members in index order, as before)
  for (int base0 = 0; base0 < NT && cnt < 64; base0 += 512) {
    bool hit[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base0 + 64 * u + lane;
      hit[u] = false;
      if (i < NT) {
        const float* p = i < N ? pcl + ((long)b * N + i) * 3 : joint_xyz + ((long)b * J + (i - N)) * 3;
        const float dx = qx - p[0], dy = qy - p[1], dz = qz - p[2];
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        hit[u] = d2 < rad2;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (cnt < 64) {  // (wave-uniform)
        const unsigned long long m = __ballot(hit[u]);
        const int slot = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (hit[u] && slot < 64) sidx[slot] = base0 + 64 * u + lane;
        cnt += __popcll(m);
      }
    }
  }
  __syncthreads();
  if (cnt > 64) cnt = 64;
  const int first = sidx[0];
  const int mine = lane < cnt ? sidx[lane] : first;
  __syncthreads();
  sidx[lane] = mine;
  if (idx_out) idx_out[((long)ri * gridDim.x + bj) * 64 + lane] = mine;
  __syncthreads();
  const float2 fj = *reinterpret_cast<const float2*>(JF + (long)bj * jf_ld + 2 * lane);
  float* gb = GF + (long)ri * gf_gs + (long)bj * 64 * gf_ld;
  float* xb = GX + (long)ri * gx_gs + (long)bj * 64 * gx_ld;
  // eight member rows per step: their loads are all requested before the first store (one member per iteration was a chain of 64 dependent round trips —
  // 45 of the kernel's 52 us at B = 32; round 5)
  const float ql = lane < 3 ? qp[lane] : 0.f;
  for (int s0 = 0; s0 < 64; s0 += 8) {
    float2 f[8];
    float pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = sidx[s0 + u];
      const float* fp = i < N ? X + ((long)b * N + i) * 128 : JF + ((long)b * J + (i - N)) * jf_ld;
      f[u] = *reinterpret_cast<const float2*>(fp + 2 * lane);
      pv[u] = 0.f;
      if (lane < 3) {
        const float* p = i < N ? pcl + ((long)b * N + i) * 3 : joint_xyz + ((long)b * J + (i - N)) * 3;
        pv[u] = p[lane];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float2 o;
      o.x = f[u].x - fj.x;
      o.y = f[u].y - fj.y;
      *reinterpret_cast<float2*>(gb + (long)(s0 + u) * gf_ld + 2 * lane) = o;
      if (lane < 4) xb[(long)(s0 + u) * gx_ld + lane] = lane < 3 ? (pv[u] - ql) / radius : 0.f;
    }
  }
}

// max over groups of `group` consecutive rows: in [rows*group][C] -> out slice (ld, coff) of [rows]
__global__ __launch_bounds__(128) void group_max_kernel(const float* __restrict__ in, float* __restrict__ out, int group, int C,
                                                        int out_ld, int out_coff) {
  const long r = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float* p = in + r * group * C + c;
    float m = -INFINITY;
    for (int s = 0; s < group; ++s) m = fmaxf(m, p[(long)s * C]);
    out[r * out_ld + out_coff + c] = m;
  }
}

// C % 4 == 0, C <= 1024, 16-byte aligned rows (round 5): 256 threads = (channel quad, member subgroup), float4 loads — every member row of the group is
// requested at once instead of one 4-byte load per member and thread (64 x 128 channels: 19.7 -> see DESIGN 4.4) — the subgroups' maxima meet in LDS.
// max is exact and order-free: same bits as the scalar form.
__global__ __launch_bounds__(256) void group_max_vec_kernel(const float* __restrict__ in, float* __restrict__ out, int group, int C, int out_ld,
                                                            int out_coff) {
  __shared__ f32x4 red[256];
  const long r = blockIdx.x;
  const int Q = C >> 2;                 // channel quads
  const int SG = 256 / Q > 0 ? 256 / Q : 1;  // member subgroups (a power of two when Q is; any Q works)
  const int q = threadIdx.x % Q, sg = threadIdx.x / Q;
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  if (sg < SG) {
    const float* p = in + r * group * C + 4 * q;
    for (int s = sg; s < group; s += SG) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + (long)s * C);
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
    }
  }
  red[threadIdx.x] = m;
  __syncthreads();
  if (sg == 0) {
    for (int g = 1; g < SG; ++g) {
      const f32x4 v = red[g * Q + q];
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
    }
    *reinterpret_cast<f32x4*>(out + r * out_ld + out_coff + 4 * q) = m;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// a14 part 1: heat-map, geometry adjacency map, spatial attention and the gate (model/model.py:334-338).
//   hm_j(p)  = exp(-(((x+.5-jx)/std)^2 + ((y+.5-jy)/std)^2) / (2 sigma^2))          util/generateFeature.py:584-600
//   GAM_j(p) = 1 / (gamma * |img_xyz(p) - uvd2xyz(joint_j)|^2 + 1)                    dataloader/loader.py:791-819
//   sw_j(p)  = sigmoid( SF[p][j] + sum_j' Wh[j][j'] hm_j'(p) + bias_j )               (SF = feature part of the 1x1 conv, a GEMM)
//   g        = wd * GAM + (1 - wd) * sw ;  Gw[b][j][p] = g * w_fc[p]
// grid (ceil(P/256), B).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heat_gam_gate_kernel(const float* __restrict__ r3d, const float* __restrict__ img_xyz,
                                                            const float* __restrict__ SF, int sf_ld, const float* __restrict__ Wh,
                                                            const float* __restrict__ bias, const float* __restrict__ weight_dis,
                                                            const float* __restrict__ wfc, const float* __restrict__ center,
                                                            const float* __restrict__ Minv, const float* __restrict__ cube,
                                                            const float* __restrict__ cam, float* __restrict__ sw_out,
                                                            float* __restrict__ Gw, int F, float std_, float sigma, float gamma,
                                                            float half_img, float flip) {
  __shared__ float jht[J][2];
  __shared__ float jxyz[J][3];
  __shared__ float wh[J * J];
  __shared__ float bs[J];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int P = F * F;
  if (tid < J) {
    const float* jp = r3d + ((long)b * J + tid) * 3;
    jht[tid][0] = (jp[0] + 1.0f) / 2.0f * (float)F;
    jht[tid][1] = (jp[1] + 1.0f) / 2.0f * (float)F;
    const Cam c = load_cam(center, Minv, cube, cam, b);
    float x, y, z;
    uvd2xyz(c, jp[0], jp[1], jp[2], half_img, flip, x, y, z);
    jxyz[tid][0] = x;
    jxyz[tid][1] = y;
    jxyz[tid][2] = z;
    bs[tid] = bias[tid];
  }
  for (int i = tid; i < J * J; i += 256) wh[i] = Wh[i];
  __syncthreads();
  const int p = blockIdx.x * 256 + tid;
  if (p >= P) return;
  const int py = p / F, px = p - py * F;
  const float mx = (float)px + 0.5f, my = (float)py + 0.5f;
  const float* ip = img_xyz + ((long)b * P + p) * 3;
  const float ix = ip[0], iy = ip[1], iz = ip[2];
  float hm[J];
  const float den = 2.0f * sigma * sigma;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const float ax = (mx - jht[j][0]) / std_, ay = (my - jht[j][1]) / std_;
    hm[j] = expf(-(ax * ax + ay * ay) / den);
  }
  const float wd = 1.0f / (1.0f + expf(-weight_dis[0]));
  const float fc = wfc[p];
  const float* sf = SF + ((long)b * P + p) * sf_ld;
#pragma unroll 1
  for (int j = 0; j < J; ++j) {
    float s = sf[j] + bs[j];
#pragma unroll
    for (int k = 0; k < J; ++k) s += wh[j * J + k] * hm[k];
    const float sw = 1.0f / (1.0f + expf(-s));
    const float dx = ix - jxyz[j][0], dy = iy - jxyz[j][1], dz = iz - jxyz[j][2];
    const float dist = (dx * dx + dy * dy) + dz * dz;
    const float gam = 1.0f / (gamma * dist + 1.0f);
    const float g = wd * gam + (1.0f - wd) * sw;
    sw_out[((long)b * J + j) * P + p] = sw;
    Gw[((long)b * J + j) * P + p] = g * fc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// a14 part 2: img_feat_j[b][j][c] = sum_p Gw[b][j][p] * relu(feat[b][p][c]) + b_fc   (since the gate is >= 0,
// relu(g*f) = g*relu(f): SURVEY a14), optionally relu((. + prev)/2) for block 2 (model/model.py:343-344).
// Per sample this is the GEMM (21 x P) @ (P x 128) of model/model.py:336-341 on the f32 matrix cores (v_mfma_f32_16x16x4_f32):
// grid (4, B): a workgroup owns 32 channels of one sample, its 4 waves = 2 channel tiles x 2 pixel halves; a wave multiplies both
// joint tiles (21 joints padded to 32) against its 16 channels over P/2 pixels, fragments straight from global memory (the gate
// rows as one 16-byte load per 16 pixels, the features as four 64-byte row segments per load: each 128-byte line is used by the
// workgroup's two channel tiles), the two pixel halves are added through LDS in a fixed order (deterministic).
// Within a 16-pixel block, k-step e of lane group g multiplies pixel 4g+e — the same permutation on both operands.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_reduce_kernel(const float* __restrict__ Gw, const float* __restrict__ feat,
                                                          const float* __restrict__ bfc, const float* __restrict__ prev,
                                                          float* __restrict__ out, int P) {
  __shared__ float part[2][32][17];  // [channel tile][joint][channel] partial sums of the second pixel half
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = wave & 1, ph = wave >> 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int c0 = blockIdx.x * 32 + ct * 16;
  const int j0 = fr, j1 = (16 + fr) < J ? (16 + fr) : J - 1;  // rows of the second joint tile beyond 20 are never stored
  const float* g0 = Gw + ((long)b * J + j0) * P;
  const float* g1 = Gw + ((long)b * J + j1) * P;
  const float* f = feat + (long)b * P * 128 + c0 + fr;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int pbeg = ph * (P / 2), pend = pbeg + P / 2;
  // the loads of block p+16 are in flight while block p is multiplied (explicit register double buffer: one memory round trip per
  // block would otherwise be exposed 32 times)
  f32x4 a0, a1, na0, na1;
  float v[4], nv[4];
  {
    const int pk = pbeg + 4 * fg;
    a0 = *reinterpret_cast<const f32x4*>(g0 + pk);
    a1 = *reinterpret_cast<const f32x4*>(g1 + pk);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = f[(long)(pk + e) * 128];
  }
  for (int p = pbeg; p < pend; p += 16) {
    const int pn = (p + 16 < pend ? p + 16 : p) + 4 * fg;  // (the last block re-loads itself: no branch around the loads)
    na0 = *reinterpret_cast<const f32x4*>(g0 + pn);
    na1 = *reinterpret_cast<const f32x4*>(g1 + pn);
#pragma unroll
    for (int e = 0; e < 4; ++e) nv[e] = f[(long)(pn + e) * 128];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float r = fmaxf(v[e], 0.f);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, a0[e], acc0, 0, 0, 0);  // A = features (row = channel), B = gate (col = joint)
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, a1[e], acc1, 0, 0, 0);
    }
    a0 = na0;
    a1 = na1;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = nv[e];
  }
  // accumulator: column = lane & 15 = joint within the tile, rows 4*fg + r = channel within the tile
  if (ph == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      part[ct][fr][4 * fg + r] = acc0[r];
      part[ct][16 + fr][4 * fg + r] = acc1[r];
    }
  }
  __syncthreads();
  if (ph == 0) {
    const float bb = bfc[0];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = 16 * t + fr;
      if (j >= J) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = ((t ? acc1[r] : acc0[r]) + part[ct][j][4 * fg + r]) + bb;
        const long o = ((long)b * J + j) * 128 + c0 + 4 * fg + r;
        if (prev) v = fmaxf((v + prev[o]) / 2.0f, 0.f);
        out[o] = v;
      }
    }
  }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int kpf_inv3x3_f32(const float* M, float* Minv, int B, int fused, void* stream) {
  KPF_REQUIRE(M && Minv && B > 0 && (fused == 0 || fused == 1), "kpf_inv3x3_f32: bad arguments");
  if (fused)
    hipLaunchKernelGGL(inv3x3_kernel<true>, dim3((B + 63) / 64), dim3(64), 0, ST(stream), M, Minv, B);
  else
    hipLaunchKernelGGL(inv3x3_kernel<false>, dim3((B + 63) / 64), dim3(64), 0, ST(stream), M, Minv, B);
  return kpf_check_launch("kpf_inv3x3_f32");
}

extern "C" int kpf_offset2joint_f32(const float* offset, const float* depth, const float* center, const float* Minv, const float* cube,
                                    const float* cam, float* joint_uvd, float* joint_xyz, int B, int S, int F, float kernel,
                                    int img_size, int flip, void* stream) {
  KPF_REQUIRE(offset && depth && center && Minv && cube && cam && joint_uvd && joint_xyz && B > 0 && F > 0 && S >= F,
              "kpf_offset2joint_f32: bad arguments");
  hipLaunchKernelGGL(offset2joint_kernel, dim3(J, B), dim3(256), 0, ST(stream), offset, depth, center, Minv, cube, cam, joint_uvd,
                     joint_xyz, S, F, kernel, (float)img_size / 2.0f, (float)flip);
  return kpf_check_launch("kpf_offset2joint_f32");
}

extern "C" int kpf_img2pcl_top4_f32(const float* pcl, const float* depth, const float* center, const float* Minv, const float* cube,
                                    const float* cam, float* closeness, int* index, float* img_xyz, int B, int N, int S, int F,
                                    int img_size, int flip, void* stream) {
  KPF_REQUIRE(pcl && depth && center && Minv && cube && cam && closeness && index && B > 0 && N > 0 && F * F >= 4 && F * F * 12 <= 64 * 1024,
              "kpf_img2pcl_top4_f32: bad arguments");
  hipLaunchKernelGGL(img2pcl_top4_kernel, dim3((N + 31) / 32, B), dim3(256), (size_t)F * F * 3 * sizeof(float), ST(stream), pcl,
                     depth, center, Minv, cube, cam, closeness, index, img_xyz, N, S, F, (float)img_size / 2.0f, (float)flip);
  return kpf_check_launch("kpf_img2pcl_top4_f32");
}

extern "C" int kpf_point_assemble_f32(const float* feat_d, const float* feat_rgb, const float* offset, const float* pcl,
                                      const float* joint_xyz, const float* closeness, const int* index, float* A1, float* A2, int B,
                                      int N, int P, float kernel, void* stream) {
  KPF_REQUIRE(feat_d && feat_rgb && offset && pcl && joint_xyz && closeness && index && A1 && A2, "kpf_point_assemble_f32: null pointer");
  KPF_REQUIRE(((long)B * N) % 4 == 0, "kpf_point_assemble_f32: B*N must be a multiple of 4");
  hipLaunchKernelGGL(point_assemble_kernel, dim3((unsigned)((long)B * N / 4)), dim3(256), 0, ST(stream), feat_d, feat_rgb, offset, pcl,
                     joint_xyz, closeness, index, A1, A2, N, P, kernel);
  return kpf_check_launch("kpf_point_assemble_f32");
}

extern "C" int kpf_softmax_pool_f32(const float* A1, const float* X, const float* joint_xyz, float* JA, int B, int N, void* stream) {
  KPF_REQUIRE(A1 && X && joint_xyz && JA && B > 0 && N > 1 && N % 2 == 0 && N <= 2048, "kpf_softmax_pool_f32: bad arguments");
  const size_t lds = (size_t)(7 * N + 32 + 8 * 7 * 128) * sizeof(float);
  static std::atomic<bool> lds_opt_in[KPF_MAX_DEVICES];
  if (lds > 64 * 1024 && !kpf_raise_lds_limit(reinterpret_cast<const void*>(&softmax_pool_kernel), lds_opt_in)) {
    kpf_set_error("kpf_softmax_pool_f32: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  hipLaunchKernelGGL(softmax_pool_kernel, dim3(3, B), dim3(256), lds, ST(stream), A1, X,
                     joint_xyz, JA, N);
  return kpf_check_launch("kpf_softmax_pool_f32");
}

extern "C" int kpf_ball_group_f32(const float* pcl, const float* joint_xyz, const float* X, const float* JF, int jf_ld, float* G,
                                  int* idx_out, int B, int N, float r0, float r1, float r2, void* stream) {
  KPF_REQUIRE(pcl && joint_xyz && X && JF && G && B > 0 && N > 0 && jf_ld >= 128 && jf_ld % 2 == 0, "kpf_ball_group_f32: bad arguments");
  const long g_stride = (long)B * J * 64 * G_LD;
  hipLaunchKernelGGL(ball_group_kernel, dim3(B * J, 3), dim3(64), 0, ST(stream), pcl, joint_xyz, X, JF, G, G + 128, idx_out, N, jf_ld, g_stride, G_LD,
                     g_stride, G_LD, r0, r1, r2);
  return kpf_check_launch("kpf_ball_group_f32");
}

extern "C" int kpf_ball_group_stacked_f32(const float* pcl, const float* joint_xyz, const float* X, const float* JF, int jf_ld, float* GF, float* GX,
                                          int* idx_out, int B, int N, float r0, float r1, float r2, void* stream) {
  KPF_REQUIRE(pcl && joint_xyz && X && JF && GF && GX && B > 0 && N > 0 && jf_ld >= 128 && jf_ld % 2 == 0, "kpf_ball_group_stacked_f32: bad arguments");
  hipLaunchKernelGGL(ball_group_kernel, dim3(B * J, 3), dim3(64), 0, ST(stream), pcl, joint_xyz, X, JF, GF, GX, idx_out, N, jf_ld, 128L, 3 * 128, 4L, 3 * 4,
                     r0, r1, r2);
  return kpf_check_launch("kpf_ball_group_stacked_f32");
}

extern "C" int kpf_group_max_f32(const float* in, float* out, long rows, int group, int C, int out_ld, int out_coff, void* stream) {
  KPF_REQUIRE(in && out && rows > 0 && group > 0 && C > 0 && out_coff + C <= out_ld, "kpf_group_max_f32: bad arguments");
  if (C % 4 == 0 && C <= 1024 && C >= 4 && out_ld % 4 == 0 && out_coff % 4 == 0 && kpf_aligned16(in) && kpf_aligned16(out))
    hipLaunchKernelGGL(group_max_vec_kernel, dim3((unsigned)rows), dim3(256), 0, ST(stream), in, out, group, C, out_ld, out_coff);
  else
    hipLaunchKernelGGL(group_max_kernel, dim3((unsigned)rows), dim3(128), 0, ST(stream), in, out, group, C, out_ld, out_coff);
  return kpf_check_launch("kpf_group_max_f32");
}

extern "C" int kpf_heat_gam_gate_f32(const float* r3d, const float* img_xyz, const float* SF, int sf_ld, const float* Wh,
                                     const float* bias, const float* weight_dis, const float* wfc, const float* center,
                                     const float* Minv, const float* cube, const float* cam, float* sw_out, float* Gw, int B, int F,
                                     int img_size, int flip, void* stream) {
  KPF_REQUIRE(r3d && img_xyz && SF && Wh && bias && weight_dis && wfc && center && Minv && cube && cam && sw_out && Gw && B > 0, "kpf_heat_gam_gate_f32: null pointer");
  hipLaunchKernelGGL(heat_gam_gate_kernel, dim3((F * F + 255) / 256, B), dim3(256), 0, ST(stream), r3d, img_xyz, SF, sf_ld, Wh, bias,
                     weight_dis, wfc, center, Minv, cube, cam, sw_out, Gw, F, 0.8f, 1.0f, 10.0f, (float)img_size / 2.0f, (float)flip);
  return kpf_check_launch("kpf_heat_gam_gate_f32");
}

extern "C" int kpf_gate_reduce_f32(const float* Gw, const float* feat, const float* bfc, const float* prev, float* out, int B, int P,
                                   void* stream) {
  KPF_REQUIRE(Gw && feat && bfc && out && B > 0 && P % 32 == 0 && kpf_aligned16(Gw), "kpf_gate_reduce_f32: bad arguments (P %% 32 == 0)");
  hipLaunchKernelGGL(gate_reduce_kernel, dim3(4, B), dim3(256), 0, ST(stream), Gw, feat, bfc, prev, out, P);
  return kpf_check_launch("kpf_gate_reduce_f32");
}
