// Stand-alone heads of the reference that KPFusion.forward does not call but the north star names:
//   CBAM channel / spatial gates   (model/cbam.py:26-94)            -> pool, gate, compress, spatial-gate, apply kernels
//   Hourglass glue                 (model/hourglass.py:122-149)     -> 2x2 max-pool, nearest-x2 upsample fused with the skip add
//   MANO layer + rotation algebra  (model/mano_head.py:144-225, util/manopth/manopth/manolayer.py:106-273) -> one kernel per batch
// All HBM-bound elementwise / reduction work on NHWC activations (channel quads contiguous -> 16-byte lanes, coalesced).
#include <math.h>

#include "kpf_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

inline unsigned grid_for(long total, int block = 256) {
  long g = (total + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > 65535l * 16 ? 65535l * 16 : g));
}

// ---------------------------------------------------------------------------------------------------------------
// CBAM 1/5: partial global avg / max pool.  grid (S, B): split s reduces pixels [s*chunk, (s+1)*chunk) of sample b for every
// channel; lanes run over channel quads (coalesced), the 256/LP pixel lanes of a workgroup are combined through LDS in a
// fixed order (deterministic — no atomics).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cbam_pool_kernel(const float* __restrict__ x, float* __restrict__ psum, float* __restrict__ pmax,
                                                        int HW, int C4, int chunk) {
  __shared__ f32x4 ls[256], lm[256];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int LP = C4 < 256 ? C4 : 256, NPL = 256 / LP;
  const int q0 = threadIdx.x % LP, pl = threadIdx.x / LP;
  const int p0 = s * chunk, p1 = min(HW, p0 + chunk);
  const float* xb = x + (long)b * HW * C4 * 4;
  for (int qb = 0; qb < C4; qb += LP) {
    const int q = qb + q0;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f}, mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (pl < NPL && q < C4) {
      for (int p = p0 + pl; p < p1; p += NPL) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + ((long)p * C4 + q) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sum[e] += v[e];
          mx[e] = fmaxf(mx[e], v[e]);
        }
      }
    }
    ls[threadIdx.x] = sum;
    lm[threadIdx.x] = mx;
    __syncthreads();
    if (pl == 0 && q < C4) {
      for (int k = 1; k < NPL; ++k) {
        const f32x4 a = ls[k * LP + q0], m = lm[k * LP + q0];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sum[e] += a[e];
          mx[e] = fmaxf(mx[e], m[e]);
        }
      }
      *reinterpret_cast<f32x4*>(psum + (((long)b * S + s) * C4 + q) * 4) = sum;
      *reinterpret_cast<f32x4*>(pmax + (((long)b * S + s) * C4 + q) * 4) = mx;
    }
    __syncthreads();
  }
}

// CBAM 2/5: finish the pools and run the shared MLP on both: scale[b][c] = sigmoid(mlp(avg) + mlp(max)) (model/cbam.py:38-57).
__global__ __launch_bounds__(256) void cbam_gate_kernel(const float* __restrict__ psum, const float* __restrict__ pmax,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ scale, int S, int HW, int C, int Cr) {
  extern __shared__ float sm[];
  float* avg = sm;            // [C]
  float* mx = sm + C;         // [C]
  float* ha = sm + 2 * C;     // [Cr]
  float* hm = sm + 2 * C + Cr;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f, m = -INFINITY;
    for (int k = 0; k < S; ++k) {
      s += psum[((long)b * S + k) * C + c];
      m = fmaxf(m, pmax[((long)b * S + k) * C + c]);
    }
    avg[c] = s / (float)HW;
    mx[c] = m;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int h = wave; h < Cr; h += 4) {
    float da = 0.f, dm = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float w = w1[(long)h * C + c];
      da = fmaf(w, avg[c], da);
      dm = fmaf(w, mx[c], dm);
    }
    da = wave_sum(da);
    dm = wave_sum(dm);
    if (lane == 0) {
      ha[h] = fmaxf(da + b1[h], 0.f);
      hm[h] = fmaxf(dm + b1[h], 0.f);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float oa = 0.f, om = 0.f;
    for (int h = 0; h < Cr; ++h) {
      const float w = w2[(long)c * Cr + h];
      oa = fmaf(w, ha[h], oa);
      om = fmaf(w, hm[h], om);
    }
    scale[(long)b * C + c] = sigmoidf_((oa + b2[c]) + (om + b2[c]));
  }
}

// CBAM 3/5: ChannelPool of the gated activation (model/cbam.py:65-67): comp[b][p] = (max_c, mean_c) of x*scale.  LG lanes per pixel.
template <int LG>
__global__ __launch_bounds__(256) void cbam_compress_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            float* __restrict__ comp, long BHW, int HW, int C4) {
  const long pix = ((long)blockIdx.x * 256 + threadIdx.x) / LG;
  const int l = threadIdx.x % LG;
  const bool live = pix < BHW;
  const long pc = live ? pix : BHW - 1;
  const int b = (int)(pc / HW);
  float mx = -INFINITY, sum = 0.f;
  for (int q = l; q < C4; q += LG) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (pc * C4 + q) * 4);
    const f32x4 s = *reinterpret_cast<const f32x4*>(scale + ((long)b * C4 + q) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float y = v[e] * s[e];
      mx = fmaxf(mx, y);
      sum += y;
    }
  }
#pragma unroll
  for (int o = LG / 2; o > 0; o >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    sum += __shfl_xor(sum, o, 64);
  }
  if (live && l == 0) {
    comp[pix * 2 + 0] = mx;
    comp[pix * 2 + 1] = sum / (float)(C4 * 4);
  }
}

// CBAM 4/5: spatial gate value per pixel: sigmoid(BN(conv7x7_{2->1}(comp))) (model/cbam.py:69-81); w7 is [2][7][7] (OIHW, O=1).
__global__ __launch_bounds__(256) void cbam_sgate_kernel(const float* __restrict__ comp, const float* __restrict__ w7, float bn_s,
                                                         float bn_t, float* __restrict__ sg, int B, int H, int W) {
  __shared__ float w[98];
  if (threadIdx.x < 98) w[threadIdx.x] = w7[threadIdx.x];
  __syncthreads();
  const long total = (long)B * H * W;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int xx = (int)(i % W);
  const int yy = (int)((i / W) % H);
  const long base = i - (long)yy * W - xx;  // first pixel of this sample
  float acc = 0.f;
#pragma unroll
  for (int ch = 0; ch < 2; ++ch)
    for (int ky = 0; ky < 7; ++ky) {
      const int iy = yy + ky - 3;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int ix = xx + kx - 3;
        if ((unsigned)ix >= (unsigned)W) continue;
        acc = fmaf(comp[(base + (long)iy * W + ix) * 2 + ch], w[ch * 49 + ky * 7 + kx], acc);
      }
    }
  sg[i] = sigmoidf_(acc * bn_s + bn_t);
}

// CBAM 5/5: out0 = (x*scale)*s, out1 = (x*scale)*(1-s); without a spatial gate (sg == nullptr): out0 = x*scale.
__global__ __launch_bounds__(256) void cbam_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ sg, float* __restrict__ out0, float* __restrict__ out1,
                                                         long total4, int HW, int C4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const int q = (int)(i % C4);
    const long pix = i / C4;
    const int b = (int)(pix / HW);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    const f32x4 s = *reinterpret_cast<const f32x4*>(scale + ((long)b * C4 + q) * 4);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = v[e] * s[e];
    if (sg) {
      const float g = sg[pix], ng = 1.0f - g;
      f32x4 a, c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = y[e] * g;
        c[e] = y[e] * ng;
      }
      *reinterpret_cast<f32x4*>(out0 + i * 4) = a;
      *reinterpret_cast<f32x4*>(out1 + i * 4) = c;
    } else {
      *reinterpret_cast<f32x4*>(out0 + i * 4) = y;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Hourglass glue
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W,
                                                         int OH, int OW, int C4) {
  const long total = (long)B * OH * OW * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % C4);
    long p = i / C4;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const float* s0 = src + ((((long)b * H + 2 * oy) * W + 2 * ox) * C4 + q) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(s0), c = *reinterpret_cast<const f32x4*>(s0 + (long)C4 * 4);
    const f32x4 d = *reinterpret_cast<const f32x4*>(s0 + (long)W * C4 * 4), e4 = *reinterpret_cast<const f32x4*>(s0 + (long)(W + 1) * C4 * 4);
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a[e], c[e]), fmaxf(d[e], e4[e]));
    *reinterpret_cast<f32x4*>(dst + i * 4) = m;
  }
}

// out[b][y][x] = up1[b][y][x] + low[b][y/2][x/2]   (nn.Upsample(scale 2, nearest) + add, model/hourglass.py:140-149)
__global__ __launch_bounds__(256) void upnearest2x_add_kernel(const float* __restrict__ low, const float* __restrict__ up1,
                                                              float* __restrict__ out, int B, int h, int w, int C4) {
  const int H = 2 * h, W = 2 * w;
  const long total = (long)B * H * W * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % C4);
    long p = i / C4;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    const f32x4 u = *reinterpret_cast<const f32x4*>(up1 + i * 4);
    const f32x4 l = *reinterpret_cast<const f32x4*>(low + ((((long)b * h + (y >> 1)) * w + (x >> 1)) * C4 + q) * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = u[e] + l[e];
    *reinterpret_cast<f32x4*>(out + i * 4) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// MANO: 6D rotations -> rotation matrices -> axis-angle -> (Rodrigues again, as the reference does) -> blend shapes, joint
// regression, kinematic chain, linear-blend skinning.  One 256-thread workgroup per sample; the sample's 778 x 3 vertices live
// in LDS from the shape blend to the final skinning.  Blend-shape bases are read transposed ([k][778*3]) so lanes coalesce.
// ---------------------------------------------------------------------------------------------------------------
constexpr int NV = 778, NJ = 16, NC = NV * 3;
__constant__ int kManoParent[16] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14};
// joints3d[i] = jtr21[ORDER[OBMAN2MANO[i]]] with jtr21 = [16 chain joints, 5 finger-tip vertices]
// (manolayer.py:251-261 then model/mano_head.py:7-16,221)
__constant__ int kManoOut[21] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 17, 18, 20, 19, 16};
__constant__ int kManoTip[5] = {745, 317, 444, 556, 673};

__device__ void rot6d_to_mat(const float* x, float* R) {  // model/mano_head.py:144-153 (b1,b2,b3 are columns)
  float a1[3] = {x[0], x[1], x[2]}, a2[3] = {x[3], x[4], x[5]};
  float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
  float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
  const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
  float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
  float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
  float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
  for (int r = 0; r < 3; ++r) {
    R[r * 3 + 0] = b1[r];
    R[r * 3 + 1] = b2[r];
    R[r * 3 + 2] = b3[r];
  }
}

__device__ void mat_to_aa(const float* R, float* aa) {  // model/mano_head.py:84-141 (on the transpose), :49-81, :171-173
  // t = R^T: t[i][j] = R[j][i]
#define T_(i, j) R[(j) * 3 + (i)]
  const float m00 = T_(0, 0), m11 = T_(1, 1), m22 = T_(2, 2);
  float q[4], t;
  if (m22 < 1e-6f) {
    if (m00 > m11) {
      t = 1 + m00 - m11 - m22;
      q[0] = T_(1, 2) - T_(2, 1); q[1] = t; q[2] = T_(0, 1) + T_(1, 0); q[3] = T_(2, 0) + T_(0, 2);
    } else {
      t = 1 - m00 + m11 - m22;
      q[0] = T_(2, 0) - T_(0, 2); q[1] = T_(0, 1) + T_(1, 0); q[2] = t; q[3] = T_(1, 2) + T_(2, 1);
    }
  } else {
    if (m00 < -m11) {
      t = 1 - m00 - m11 + m22;
      q[0] = T_(0, 1) - T_(1, 0); q[1] = T_(2, 0) + T_(0, 2); q[2] = T_(1, 2) + T_(2, 1); q[3] = t;
    } else {
      t = 1 + m00 + m11 + m22;
      q[0] = t; q[1] = T_(1, 2) - T_(2, 1); q[2] = T_(2, 0) - T_(0, 2); q[3] = T_(0, 1) - T_(1, 0);
    }
  }
#undef T_
  const float inv = 0.5f / sqrtf(t);
  for (int i = 0; i < 4; ++i) q[i] *= inv;
  const float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const float s = sqrtf(s2), c = q[0];
  const float two_theta = 2.0f * (c < 0.0f ? atan2f(-s, -c) : atan2f(s, c));
  const float k = s2 > 0.0f ? two_theta / s : 2.0f;
  for (int i = 0; i < 3; ++i) {
    const float v = q[i + 1] * k;
    aa[i] = isnan(v) ? 0.0f : v;
  }
}

__device__ void rodrigues(const float* aa, float* R) {  // util/manopth/manopth/rodrigues_layer.py:16-57
  const float ax = aa[0] + 1e-8f, ay = aa[1] + 1e-8f, az = aa[2] + 1e-8f;
  const float ang = sqrtf(ax * ax + ay * ay + az * az);
  const float h = ang * 0.5f, sn = sinf(h), cs = cosf(h);
  float q[4] = {cs, sn * (aa[0] / ang), sn * (aa[1] / ang), sn * (aa[2] / ang)};
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
  const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
  R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
  R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2;   R[5] = 2 * yz - 2 * wx;
  R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
}

struct ManoArgs {
  const float* pose6d;  // [B][ld6] (first 96 used)
  const float* betas;   // [B][ldb] (first 10 used)
  int ld6, ldb;
  const float *shapeT, *poseT, *vtmpl, *jreg, *skin, *hands_mean;  // [10][NC], [135][NC], [NC], [16][778], [778][16], [45]
  float *verts, *joints, *rotmat, *aa;                             // [B][778][3], [B][21][3], [B][16][9], [B][48]
};

__global__ __launch_bounds__(256) void mano_kernel(ManoArgs a) {
  __shared__ float vsh[NC];        // v_shaped -> v_posed -> posed vertices (metres)
  __shared__ float Rj[NJ][9];      // rotations the MANO layer uses (Rodrigues of the axis-angle pose)
  __shared__ float pmap[135];
  __shared__ float beta[10];
  __shared__ float Jr[NJ][3];
  __shared__ float G[NJ][12], G2[NJ][12];  // global joint transforms (3x4), and with the rest pose removed
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < 10) beta[t] = a.betas[(long)b * a.ldb + t];
  if (t < NJ) {
    float R[9], aa[3];
    rot6d_to_mat(a.pose6d + (long)b * a.ld6 + t * 6, R);
    mat_to_aa(R, aa);
    for (int i = 0; i < 9; ++i) a.rotmat[((long)b * NJ + t) * 9 + i] = R[i];
    for (int i = 0; i < 3; ++i) a.aa[(long)b * 48 + t * 3 + i] = aa[i];
    if (t > 0)
      for (int i = 0; i < 3; ++i) aa[i] = a.hands_mean[(t - 1) * 3 + i] + aa[i];
    rodrigues(aa, Rj[t]);
    if (t > 0)
      for (int i = 0; i < 9; ++i) pmap[(t - 1) * 9 + i] = Rj[t][i] - ((i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f);
  }
  __syncthreads();
  for (int c = t; c < NC; c += 256) {  // shape blend
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) v = fmaf(a.shapeT[(long)k * NC + c], beta[k], v);
    vsh[c] = v + a.vtmpl[c];
  }
  __syncthreads();
  {  // joint regression from the shaped (un-posed) vertices: 48 dot products of length 778, 12 per wave
    const int wave = t >> 6, lane = t & 63;
    for (int o = wave; o < NJ * 3; o += 4) {
      const int j = o / 3, d = o - j * 3;
      float s = 0.f;
      for (int v = lane; v < NV; v += 64) s = fmaf(a.jreg[(long)j * NV + v], vsh[v * 3 + d], s);
      s = wave_sum(s);
      if (lane == 0) Jr[j][d] = s;
    }
  }
  __syncthreads();
  for (int c = t; c < NC; c += 256) {  // pose blend (in place: each coordinate is touched by exactly one thread)
    float v = 0.f;
    for (int k = 0; k < 135; ++k) v = fmaf(a.poseT[(long)k * NC + c], pmap[k], v);
    vsh[c] += v;
  }
  if (t == 0) {  // kinematic chain, parents precede children
    for (int j = 0; j < NJ; ++j) {
      const int p = kManoParent[j];
      float loc[12];
      for (int r = 0; r < 3; ++r) {
        loc[r * 4 + 0] = Rj[j][r * 3 + 0];
        loc[r * 4 + 1] = Rj[j][r * 3 + 1];
        loc[r * 4 + 2] = Rj[j][r * 3 + 2];
        loc[r * 4 + 3] = p < 0 ? Jr[j][r] : Jr[j][r] - Jr[p][r];
      }
      if (p < 0) {
        for (int i = 0; i < 12; ++i) G[j][i] = loc[i];
      } else {
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 4; ++c) {
            float s = G[p][r * 4 + 0] * loc[0 * 4 + c] + G[p][r * 4 + 1] * loc[1 * 4 + c] + G[p][r * 4 + 2] * loc[2 * 4 + c];
            if (c == 3) s += G[p][r * 4 + 3];
            G[j][r * 4 + c] = s;
          }
      }
      for (int r = 0; r < 3; ++r) {
        const float corr = G[j][r * 4 + 0] * Jr[j][0] + G[j][r * 4 + 1] * Jr[j][1] + G[j][r * 4 + 2] * Jr[j][2];
        G2[j][r * 4 + 0] = G[j][r * 4 + 0];
        G2[j][r * 4 + 1] = G[j][r * 4 + 1];
        G2[j][r * 4 + 2] = G[j][r * 4 + 2];
        G2[j][r * 4 + 3] = G[j][r * 4 + 3] - corr;
      }
    }
  }
  __syncthreads();
  for (int v = t; v < NV; v += 256) {  // linear-blend skinning
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.f;
    for (int j = 0; j < NJ; ++j) {
      const float w = a.skin[(long)v * NJ + j];
#pragma unroll
      for (int i = 0; i < 12; ++i) T[i] = fmaf(w, G2[j][i], T[i]);
    }
    const float px = vsh[v * 3 + 0], py = vsh[v * 3 + 1], pz = vsh[v * 3 + 2];
    float o[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = T[r * 4 + 0] * px + T[r * 4 + 1] * py + T[r * 4 + 2] * pz + T[r * 4 + 3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      vsh[v * 3 + r] = o[r];
      a.verts[((long)b * NV + v) * 3 + r] = o[r] * 1000.0f;
    }
  }
  __syncthreads();
  if (t < 21 * 3) {
    const int i = t / 3, d = t - i * 3;
    const int src = kManoOut[i];
    const float v = src < NJ ? G[src][d * 4 + 3] : vsh[kManoTip[src - NJ] * 3 + d];
    a.joints[((long)b * 21 + i) * 3 + d] = v * 1000.0f;
  }
}

}  // namespace

extern "C" long kpf_cbam_workspace_floats(int B, int HW, int C) {
  const int S = HW >= 64 * 16 ? 64 : (HW + 15) / 16;
  return 2l * B * (S < 1 ? 1 : S) * C;
}

extern "C" int kpf_cbam_channel_gate_f32(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                         float* workspace, float* scale, int B, int HW, int C, int Cr, void* stream) {
  KPF_REQUIRE(x && w1 && b1 && w2 && b2 && workspace && scale, "kpf_cbam_channel_gate_f32: null pointer");
  KPF_REQUIRE(B > 0 && HW > 0 && C > 0 && Cr > 0 && C % 4 == 0, "kpf_cbam_channel_gate_f32: bad shape B=%d HW=%d C=%d Cr=%d", B, HW, C, Cr);
  KPF_REQUIRE(kpf_aligned16(x) && kpf_aligned16(workspace), "kpf_cbam_channel_gate_f32: pointers must be 16-byte aligned");
  KPF_REQUIRE((size_t)(2 * C + 2 * Cr) * sizeof(float) <= 64 * 1024, "kpf_cbam_channel_gate_f32: C=%d too wide", C);
  int S = HW >= 64 * 16 ? 64 : (HW + 15) / 16;
  if (S < 1) S = 1;
  const int chunk = (HW + S - 1) / S;
  float* psum = workspace;
  float* pmax = workspace + (long)B * S * C;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cbam_pool_kernel, dim3(S, B), dim3(256), 0, st, x, psum, pmax, HW, C / 4, chunk);
  hipLaunchKernelGGL(cbam_gate_kernel, dim3(B), dim3(256), (size_t)(2 * C + 2 * Cr) * sizeof(float), st, psum, pmax, w1, b1, w2, b2, scale,
                     S, HW, C, Cr);
  return kpf_check_launch("kpf_cbam_channel_gate_f32");
}

extern "C" int kpf_cbam_spatial_gate_f32(const float* x, const float* scale, const float* w7, float bn_scale, float bn_shift,
                                         float* comp, float* sgate, int B, int H, int W, int C, void* stream) {
  KPF_REQUIRE(x && scale && w7 && comp && sgate, "kpf_cbam_spatial_gate_f32: null pointer");
  KPF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "kpf_cbam_spatial_gate_f32: bad shape");
  KPF_REQUIRE(kpf_aligned16(x) && kpf_aligned16(scale), "kpf_cbam_spatial_gate_f32: pointers must be 16-byte aligned");
  const long BHW = (long)B * H * W;
  const int C4 = C / 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (C4 <= 16)
    hipLaunchKernelGGL((cbam_compress_kernel<16>), dim3(grid_for(BHW * 16)), dim3(256), 0, st, x, scale, comp, BHW, H * W, C4);
  else if (C4 <= 32)
    hipLaunchKernelGGL((cbam_compress_kernel<32>), dim3(grid_for(BHW * 32)), dim3(256), 0, st, x, scale, comp, BHW, H * W, C4);
  else
    hipLaunchKernelGGL((cbam_compress_kernel<64>), dim3(grid_for(BHW * 64)), dim3(256), 0, st, x, scale, comp, BHW, H * W, C4);
  hipLaunchKernelGGL(cbam_sgate_kernel, dim3(grid_for(BHW)), dim3(256), 0, st, comp, w7, bn_scale, bn_shift, sgate, B, H, W);
  return kpf_check_launch("kpf_cbam_spatial_gate_f32");
}

extern "C" int kpf_cbam_apply_f32(const float* x, const float* scale, const float* sgate, float* out0, float* out1, int B, int HW, int C,
                                  void* stream) {
  KPF_REQUIRE(x && scale && out0 && (sgate == nullptr) == (out1 == nullptr), "kpf_cbam_apply_f32: null pointer / gate-output mismatch");
  KPF_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0, "kpf_cbam_apply_f32: bad shape");
  KPF_REQUIRE(kpf_aligned16(x) && kpf_aligned16(scale) && kpf_aligned16(out0) && kpf_aligned16(out1), "kpf_cbam_apply_f32: alignment");
  const long total4 = (long)B * HW * (C / 4);
  hipLaunchKernelGGL(cbam_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, scale, sgate, out0,
                     out1, total4, HW, C / 4);
  return kpf_check_launch("kpf_cbam_apply_f32");
}

extern "C" int kpf_maxpool2x2_f32(const float* src, float* dst, int B, int H, int W, int C, void* stream) {
  KPF_REQUIRE(src && dst && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "kpf_maxpool2x2_f32: bad arguments");
  KPF_REQUIRE(kpf_aligned16(src) && kpf_aligned16(dst), "kpf_maxpool2x2_f32: alignment");
  const int OH = H / 2, OW = W / 2;
  const long total = (long)B * OH * OW * (C / 4);
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, B, H, W, OH, OW,
                     C / 4);
  return kpf_check_launch("kpf_maxpool2x2_f32");
}

extern "C" int kpf_upnearest2x_add_f32(const float* low, const float* up1, float* out, int B, int h, int w, int C, void* stream) {
  KPF_REQUIRE(low && up1 && out && B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "kpf_upnearest2x_add_f32: bad arguments");
  KPF_REQUIRE(kpf_aligned16(low) && kpf_aligned16(up1) && kpf_aligned16(out), "kpf_upnearest2x_add_f32: alignment");
  const long total = (long)B * 4 * h * w * (C / 4);
  hipLaunchKernelGGL(upnearest2x_add_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), low, up1, out, B, h,
                     w, C / 4);
  return kpf_check_launch("kpf_upnearest2x_add_f32");
}

extern "C" int kpf_mano_forward_f32(const float* pose6d, int ld6, const float* betas, int ldb, const float* shapedirs_t,
                                    const float* posedirs_t, const float* v_template, const float* j_regressor, const float* skin_weights,
                                    const float* hands_mean, float* verts, float* joints, float* rotmat, float* pose_aa, int B,
                                    void* stream) {
  KPF_REQUIRE(pose6d && betas && shapedirs_t && posedirs_t && v_template && j_regressor && skin_weights && hands_mean && verts && joints &&
                  rotmat && pose_aa,
              "kpf_mano_forward_f32: null pointer");
  KPF_REQUIRE(B > 0 && ld6 >= 96 && ldb >= 10, "kpf_mano_forward_f32: bad shape B=%d ld6=%d ldb=%d", B, ld6, ldb);
  ManoArgs a;
  a.pose6d = pose6d; a.betas = betas; a.ld6 = ld6; a.ldb = ldb;
  a.shapeT = shapedirs_t; a.poseT = posedirs_t; a.vtmpl = v_template; a.jreg = j_regressor; a.skin = skin_weights; a.hands_mean = hands_mean;
  a.verts = verts; a.joints = joints; a.rotmat = rotmat; a.aa = pose_aa;
  hipLaunchKernelGGL(mano_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  return kpf_check_launch("kpf_mano_forward_f32");
}
