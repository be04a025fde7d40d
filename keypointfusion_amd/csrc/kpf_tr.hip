// The 21-token transformers of the fusion head as single-workgroup fused kernels:
//   kpf_tr_encoder_f32  : KP_Interaction_TR  = Linear embed + position table + 4 post-LN BERT layers (4 heads x 32,
//                         intermediate 16, GELU-erf, LN eps 1e-12) + cls_head/residual 3-vector heads  (model/model.py:30-126)
//   kpf_xattn_layer_f32 : the one observable decoder layer of updatedDecoder (cross attention 21x21, post-LN, ReLU FFN)
//                         (model/transfusion_head.py:137-173, 635-708)
// One workgroup (512 threads) owns one sample: all activations (21 x 128 tokens, Q/K/V, scores) live in LDS for the
// whole stack, weights stream from L2 (they are shared by every workgroup), nothing round-trips through HBM between
// layers and the whole stack is ONE launch instead of ~30.  The work is tiny (12 MFLOP per sample) and latency-bound, so
// it runs on the vector ALUs: each thread owns one output channel for a quarter of the tokens, weights are stored transposed
// [K][N] so a wave reads 256 contiguous bytes per k and the token values are LDS broadcasts.
#include "kpf_common.h"

namespace {

constexpr int T = 21;    // tokens
constexpr int H = 128;   // hidden
constexpr int NH = 4;    // heads
constexpr int HD = 32;   // head dim
constexpr int TG = 6;    // tokens per work item (4 token groups: 6 + 6 + 6 + 3)
constexpr int NG = 4;    // token groups
constexpr int NTHR = 512;  // threads per workgroup: 8 waves = 2 per SIMD, so LDS / L2 latency of one wave hides under the other

// out[t][o] = act( (sum_k in[t][k] * Wt[k][o] + bias[o]) * scale (+ add[t][o]) ), in/out/add in LDS (row strides multiples of 4
// floats), Wt/bias global.  Work item = (output channel o, token half); 8 weight loads are issued together (the k loop would
// otherwise be one dependent L2 round trip per k: ~40 us per 128x128 layer) and the token values come as float4 LDS broadcasts.
template <int ACT>  // 0 none, 1 relu, 2 gelu(erf)
__device__ __forceinline__ void linear(const float* in, int ldin, const float* __restrict__ Wt, const float* __restrict__ bias, int K,
                                       int N, float* out, int ldo, const float* add, int ldadd, float scale) {
  for (int item = threadIdx.x; item < NG * N; item += NTHR) {
    const int o = item % N, g = item / N;
    const int t0 = g * TG;
    const int nt = (T - t0) < TG ? (T - t0) : TG;
    float acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) acc[t] = 0.f;
    const float* ip = in + t0 * ldin;
    const float* wp = Wt + o;
    constexpr int U = 16;  // weights fetched per batch; the next batch is in flight while this one is consumed
    const int KU = K - K % U;
    float w[U], wn[U];
    if (KU > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) w[u] = wp[(long)u * N];
    }
    int k = 0;
    for (; k < KU; k += U) {
      if (k + U < KU) {
#pragma unroll
        for (int u = 0; u < U; ++u) wn[u] = wp[(long)(k + U + u) * N];
      }
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        if (t < nt) {
          float s = acc[t];
#pragma unroll
          for (int v = 0; v < U / 4; ++v) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ip + t * ldin + k + 4 * v);
            s = fmaf(a[0], w[4 * v], s);
            s = fmaf(a[1], w[4 * v + 1], s);
            s = fmaf(a[2], w[4 * v + 2], s);
            s = fmaf(a[3], w[4 * v + 3], s);
          }
          acc[t] = s;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) w[u] = wn[u];
    }
    for (; k < K; ++k) {
      const float wv = wp[(long)k * N];
#pragma unroll
      for (int t = 0; t < TG; ++t)
        if (t < nt) acc[t] = fmaf(ip[t * ldin + k], wv, acc[t]);
    }
    const float bv = bias ? bias[o] : 0.f;
#pragma unroll
    for (int t = 0; t < TG; ++t)
      if (t < nt) {
        float v = (acc[t] + bv) * scale;
        if (add) v += add[(t0 + t) * ldadd + o];
        if (ACT == 1) v = fmaxf(v, 0.f);
        if (ACT == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        out[(t0 + t) * ldo + o] = v;
      }
  }
}

// LayerNorm over H for each of the T tokens (in place allowed); one wave per token round-robin
__device__ __forceinline__ void layernorm_tokens(const float* in, int ldin, const float* __restrict__ w, const float* __restrict__ b,
                                                 float eps, float* out, int ldo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = wave; t < T; t += NTHR / 64) {
    const float a0 = in[t * ldin + lane], a1 = in[t * ldin + 64 + lane];
    const float mean = wave_sum(a0 + a1) * (1.0f / H);
    const float d0 = a0 - mean, d1 = a1 - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / H);
    const float rstd = 1.0f / sqrtf(var + eps);
    out[t * ldo + lane] = d0 * rstd * w[lane] + b[lane];
    out[t * ldo + 64 + lane] = d1 * rstd * w[64 + lane] + b[64 + lane];
  }
}

// ctx[i][h*32+d] = sum_j softmax_j( q[i][h,:].k[j][h,:] * qscale ) v[j][h*32+d];  S is scratch [NH][T][T]
__device__ __forceinline__ void attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float qscale,
                                          float* S, float* ctx, int ldc) {
  for (int item = threadIdx.x; item < NH * T * T; item += NTHR) {
    const int h = item / (T * T), r = item - h * T * T, i = r / T, j = r - i * T;
    const float* qp = q + i * ldq + h * HD;
    const float* kp = k + j * ldk + h * HD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(qp[d], kp[d], s);
    S[item] = s * qscale;
  }
  __syncthreads();
  for (int row = threadIdx.x; row < NH * T; row += NTHR) {
    float* sp = S + row * T;
    float m = -INFINITY;
    for (int j = 0; j < T; ++j) m = fmaxf(m, sp[j]);
    float se = 0.f;
    for (int j = 0; j < T; ++j) {
      const float e = expf(sp[j] - m);
      sp[j] = e;
      se += e;
    }
    const float inv = 1.0f / se;
    for (int j = 0; j < T; ++j) sp[j] *= inv;
  }
  __syncthreads();
  for (int item = threadIdx.x; item < T * H; item += NTHR) {
    const int i = item / H, c = item - i * H, h = c / HD;
    const float* sp = S + (h * T + i) * T;
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < T; ++j) a = fmaf(sp[j], v[j * ldv + c], a);
    ctx[i * ldc + c] = a;
  }
  __syncthreads();
}

// ---- packed weight layout of one encoder (floats) -----------------------------------------------------------------
// [Wemb_t Din x 128][bemb 128][pos 21 x 128] then per layer L (4x):
//   [Wqkv_t 128 x 384][bqkv 384][Wo_t 128 x 128][bo 128][ln1w 128][ln1b 128][Wi_t 128 x 16][bi 16][Wo2_t 16 x 128][bo2 128][ln2w 128][ln2b 128]
// then [Wcls_t 128 x 3][bcls 3][Wres_t Din x 3][bres 3]
constexpr int ENC_LAYER = 128 * 384 + 384 + 128 * 128 + 128 + 128 + 128 + 128 * 16 + 16 + 16 * 128 + 128 + 128 + 128;

__global__ __launch_bounds__(NTHR) void tr_encoder_kernel(const float* __restrict__ x, int ldx, int Din, const float* __restrict__ W,
                                                         float* __restrict__ hout, float* __restrict__ score, float* __restrict__ score2, int s2_ld) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ldi = (Din + 3) & ~3;  // row strides are multiples of 4 floats: float4 LDS reads in linear()
  float* X0 = sm;                // [T][ldi]   input tokens (kept for the residual head)
  float* Hb = X0 + T * ldi;      // [T][132]   hidden state
  float* QKV = Hb + T * 132;     // [T][388]
  float* CTX = QKV + T * 388;    // [T][132]
  float* T1 = CTX + T * 132;     // [T][132]
  float* S = T1 + T * 132;       // [NH*T*T]
  float* IM = S + NH * T * T;    // [T][20]
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < T * Din; i += NTHR) {
    const int t = i / Din, k = i - t * Din;
    X0[t * ldi + k] = x[((long)b * T + t) * ldx + k];
  }
  __syncthreads();
  const float* Wemb = W;
  const float* bemb = Wemb + Din * H;
  const float* pos = bemb + H;
  // h = Linear(x) + pos : feed pos through the "add" operand (T1 <- pos)
  for (int i = threadIdx.x; i < T * H; i += NTHR) T1[(i / H) * 132 + (i % H)] = pos[i];
  __syncthreads();
  linear<0>(X0, ldi, Wemb, bemb, Din, H, Hb, 132, T1, 132, 1.0f);
  __syncthreads();
  const float* L = pos + T * H;
  for (int l = 0; l < 4; ++l, L += ENC_LAYER) {
    const float* Wqkv = L;
    const float* bqkv = Wqkv + 128 * 384;
    const float* Wo = bqkv + 384;
    const float* bo = Wo + 128 * 128;
    const float* ln1w = bo + 128;
    const float* ln1b = ln1w + 128;
    const float* Wi = ln1b + 128;
    const float* bi = Wi + 128 * 16;
    const float* Wo2 = bi + 16;
    const float* bo2 = Wo2 + 16 * 128;
    const float* ln2w = bo2 + 128;
    const float* ln2b = ln2w + 128;
    linear<0>(Hb, 132, Wqkv, bqkv, H, 384, QKV, 388, nullptr, 0, 1.0f);
    __syncthreads();
    attention(QKV, 388, QKV + 128, 388, QKV + 256, 388, 0.17677669529663687f /* 1/sqrt(32) */, S, CTX, 132);
    linear<0>(CTX, 132, Wo, bo, H, H, T1, 132, Hb, 132, 1.0f);  // dense(ctx) + h
    __syncthreads();
    layernorm_tokens(T1, 132, ln1w, ln1b, 1e-12f, Hb, 132);     // h1
    __syncthreads();
    linear<2>(Hb, 132, Wi, bi, H, 16, IM, 20, nullptr, 0, 1.0f);
    __syncthreads();
    linear<0>(IM, 20, Wo2, bo2, 16, H, T1, 132, Hb, 132, 1.0f);  // dense(inter) + h1
    __syncthreads();
    layernorm_tokens(T1, 132, ln2w, ln2b, 1e-12f, Hb, 132);
    __syncthreads();
  }
  const float* Wcls = L;
  const float* bcls = Wcls + H * 3;
  const float* Wres = bcls + 3;
  const float* bres = Wres + Din * 3;
  for (int i = threadIdx.x; i < T * H; i += NTHR) hout[(long)b * T * H + i] = Hb[(i / H) * 132 + (i % H)];
  if (threadIdx.x < T * 3) {
    const int t = threadIdx.x / 3, o = threadIdx.x - t * 3;
    float a = bcls[o];
    for (int k = 0; k < H; ++k) a = fmaf(Hb[t * 132 + k], Wcls[k * 3 + o], a);
    float r = bres[o];
    for (int k = 0; k < Din; ++k) r = fmaf(X0[t * ldi + k], Wres[k * 3 + o], r);
    score[((long)b * T + t) * 3 + o] = a + r;
    if (score2) score2[((long)b * T + t) * s2_ld + o] = a + r;
  }
}

// ---- packed decoder-layer weights --------------------------------------------------------------------------------
// [qpos 21x128][kpos 21x128][Wq_t 128x128][bq 128][Wkv_t 128x256][bkv 256][Wo_t 128x128][bo 128][n2w][n2b][W1_t 128x128][b1][W2_t 128x128][b2][n3w][n3b]
__global__ __launch_bounds__(NTHR) void xattn_layer_kernel(const float* __restrict__ query, const float* __restrict__ key,
                                                          const float* __restrict__ W, float* __restrict__ out, int ldo, int ocoff) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Q0 = sm;               // [T][132] un-embedded query (residual)
  float* QE = Q0 + T * 132;     // [T][132] query + pos, later scratch
  float* KE = QE + T * 132;     // [T][132] key + pos
  float* Qp = KE + T * 132;     // [T][132] projected q
  float* KV = Qp + T * 132;     // [T][260]
  float* CTX = KV + T * 260;    // [T][132]
  float* S = CTX + T * 132;     // [NH*T*T]
  const int b = blockIdx.x;
  const float* qpos = W;
  const float* kpos = qpos + T * H;
  const float* Wq = kpos + T * H;
  const float* bq = Wq + H * H;
  const float* Wkv = bq + H;
  const float* bkv = Wkv + H * 256;
  const float* Wo = bkv + 256;
  const float* bo = Wo + H * H;
  const float* n2w = bo + H;
  const float* n2b = n2w + H;
  const float* W1 = n2b + H;
  const float* b1 = W1 + H * H;
  const float* W2 = b1 + H;
  const float* b2 = W2 + H * H;
  const float* n3w = b2 + H;
  const float* n3b = n3w + H;
  for (int i = threadIdx.x; i < T * H; i += NTHR) {
    const int t = i / H, c = i - t * H;
    const float qv = query[(long)b * T * H + i], kv = key[(long)b * T * H + i];
    Q0[t * 132 + c] = qv;
    QE[t * 132 + c] = qv + qpos[i];
    KE[t * 132 + c] = kv + kpos[i];
  }
  __syncthreads();
  linear<0>(QE, 132, Wq, bq, H, H, Qp, 132, nullptr, 0, 0.17677669529663687f);  // (Wq x + b) * head_dim^-1/2
  linear<0>(KE, 132, Wkv, bkv, H, 256, KV, 260, nullptr, 0, 1.0f);
  __syncthreads();
  attention(Qp, 132, KV, 260, KV + 128, 260, 1.0f, S, CTX, 132);
  linear<0>(CTX, 132, Wo, bo, H, H, QE, 132, Q0, 132, 1.0f);  // query + attn
  __syncthreads();
  layernorm_tokens(QE, 132, n2w, n2b, 1e-5f, Q0, 132);        // x = norm2(.)
  __syncthreads();
  linear<1>(Q0, 132, W1, b1, H, H, CTX, 132, nullptr, 0, 1.0f);
  __syncthreads();
  linear<0>(CTX, 132, W2, b2, H, H, QE, 132, Q0, 132, 1.0f);
  __syncthreads();
  layernorm_tokens(QE, 132, n3w, n3b, 1e-5f, CTX, 132);
  __syncthreads();
  for (int i = threadIdx.x; i < T * H; i += NTHR) out[((long)b * T + i / H) * ldo + ocoff + (i % H)] = CTX[(i / H) * 132 + (i % H)];
}

}  // namespace

extern "C" int kpf_tr_encoder_f32(const float* x, int ldx, int Din, const float* W, float* h, float* score, float* score2,
                                  int score2_ld, int B, void* stream) {
  KPF_REQUIRE(x && W && h && score && B > 0 && Din > 0 && Din <= 256 && ldx >= Din, "kpf_tr_encoder_f32: bad arguments");
  const size_t lds = (size_t)(21 * ((Din + 3) & ~3) + 21 * 132 * 3 + 21 * 388 + 4 * 21 * 21 + 21 * 20) * sizeof(float);
  hipLaunchKernelGGL(tr_encoder_kernel, dim3(B), dim3(NTHR), lds, reinterpret_cast<hipStream_t>(stream), x, ldx, Din, W, h, score,
                     score2, score2_ld);
  return kpf_check_launch("kpf_tr_encoder_f32");
}

extern "C" int kpf_xattn_layer_f32(const float* query, const float* key, const float* W, float* out, int out_ld, int out_coff, int B,
                                   void* stream) {
  KPF_REQUIRE(query && key && W && out && B > 0 && out_coff + 128 <= out_ld, "kpf_xattn_layer_f32: bad arguments");
  const size_t lds = (size_t)(21 * 132 * 5 + 21 * 260 + 4 * 21 * 21) * sizeof(float);
  hipLaunchKernelGGL(xattn_layer_kernel, dim3(B), dim3(NTHR), lds, reinterpret_cast<hipStream_t>(stream), query, key, W, out, out_ld,
                     out_coff);
  return kpf_check_launch("kpf_xattn_layer_f32");
}

extern "C" int kpf_tr_encoder_weight_floats(int Din) { return Din * 128 + 128 + 21 * 128 + 4 * ENC_LAYER + 128 * 3 + 3 + Din * 3 + 3; }
extern "C" int kpf_xattn_weight_floats(void) { return 2 * 21 * 128 + 128 * 128 + 128 + 128 * 256 + 256 + 128 * 128 + 128 + 256 + 128 * 128 + 128 + 128 * 128 + 128 + 256; }
