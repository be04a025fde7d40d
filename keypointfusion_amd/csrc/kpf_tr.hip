// The 21-token transformers of the fusion head as single-workgroup fused kernels:
//   kpf_tr_encoder_f32  : KP_Interaction_TR  = Linear embed + position table + 4 post-LN BERT layers (4 heads x 32,
//                         intermediate 16, GELU-erf, LN eps 1e-12) + cls_head/residual 3-vector heads  (model/model.py:30-126)
//   kpf_xattn_layer_f32 : the one observable decoder layer of updatedDecoder (cross attention 21x21, post-LN, ReLU FFN)
//                         (model/transfusion_head.py:137-173, 635-708)
// One workgroup (512 threads) owns one sample: all activations (21 x 128 tokens, Q/K/V, scores) live in LDS for the
// whole stack, weights stream from L2 (they are shared by every workgroup), nothing round-trips through HBM between
// layers and the whole stack is ONE launch instead of ~30.  The work is tiny (12 MFLOP per sample) and latency-bound on the
// weight stream: every Linear runs on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fmaf chains, tokens = rows padded to
// two 16-row tiles) with its weights, stored transposed [K][N], streamed chunk by chunk through a two-slot LDS ring by LDS-DMA
// (below); softmax / LayerNorm / GELU are wave-shuffle reductions on the vector ALUs.
#include "kpf_common.h"

namespace {

constexpr int T = 21;    // tokens
constexpr int H = 128;   // hidden
constexpr int NH = 4;    // heads
constexpr int HD = 32;   // head dim


constexpr int NTHR = 512;  // threads per workgroup: 8 waves = 2 per SIMD, so LDS / L2 latency of one wave hides under the other

// ---- weight streaming: a two-slot LDS ring filled by LDS-DMA, one chunk ahead of the MFMAs -----------------------------------------
// The stacks are latency-bound: every sample (= workgroup) has to pull ~1.2 MB of weights through one CU's L2 port, and the first
// version did so with 16 loads in flight per thread, one dependent batch after the other (165 us per stack).  Now every Linear is cut
// into chunks of NC output columns x all K rows (<= 32 KB), the chunks of the WHOLE stack form one schedule, and chunk i+1 is in
// flight (global_load_lds_dwordx4, no VGPR round trip) while chunk i is multiplied — also across Linear / attention / LayerNorm
// boundaries.  The products run on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fmaf chains): tokens are the rows (21
// padded to two 16-row tiles; the rows beyond 20 are computed on whatever follows the token array in LDS and never stored), a wave owns
// one 16 x 16 output tile of the chunk.
struct Chunk {
  const float* src;  // first element of the chunk: W[0][col0]
  int rows;          // K rows to fetch (real, un-padded)
  int stride;        // N: floats between rows in global memory
  int nc;            // columns of this chunk: 16 / 32 / 64 / 128
};
constexpr int MAXCH = 48;
constexpr int SCHED_FLOATS = MAXCH * 6 + 4;  // Chunk is 24 bytes; + the chunk count
constexpr int GPAD = 16;                    // floats of padding after every 1-KiB DMA group (spreads the 4 k-groups of a fragment read over the banks)
constexpr int RING_SLOT = 8192 + 32 * GPAD + 64;  // floats per ring slot: 32 KiB of weights = 32 DMA groups + their padding

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;

// LDS image of a chunk: DMA group g (= 256 consecutive floats of the row-major [rows][nc] chunk) sits at g * (256 + GPAD)
__device__ __forceinline__ int ring_off(int k, int col, int nc) {
  const int lin = k * nc + col;
  return (lin >> 8) * (256 + GPAD) + (lin & 255);
}

__device__ __forceinline__ void issue_chunk(const Chunk& c, float* slot) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = 31 - __builtin_clz(c.nc >> 2);  // log2(lanes per row): nc is 16 / 32 / 64 / 128
  const int rpi = 64 >> lg;                      // rows per DMA instruction (1 KiB)
  const int ngroups = (c.rows * c.nc + 255) >> 8;
  const int lrow = lane >> lg, lcol = (lane & ((1 << lg) - 1)) * 4;
  for (int g = wave; g < ngroups; g += NTHR / 64) {
    int row = g * rpi + lrow;
    row = row < c.rows ? row : c.rows - 1;   // tail of the last group: re-fetch the last row (multiplied by zero-padded tokens)
    const float* src = c.src + (long)row * c.stride + lcol;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(slot + g * (256 + GPAD)), 16, 0, 0);
  }
}

// out[t][o] = act( (sum_k in[t][k] * W[k][o] + bias[o]) * scale (+ add[t][o]) ) for the chunks [c0, c0 + nch) of the schedule, which
// cover the N columns of one Linear in order.  in/out/add in LDS; K = padded depth (multiple of 16; in[][K_real..K) are zeros).
// On entry chunk c0 is in flight (or landed) in ring slot c0 & 1; on exit chunk c0 + nch is.
template <int ACT, int NC, int KC>  // ACT: 0 none, 1 relu, 2 gelu(erf); NC: chunk columns; KC: padded depth (0: run time, K)
__device__ __forceinline__ void linear(const float* in, int ldin, int K, const float* __restrict__ bias, float* out, int ldo, const float* add,
                                       int ldadd, float scale, const Chunk* sched, int c0, int nch, int ntotal, float* ring) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int Kd = KC ? KC : K;
  constexpr int LNC = NC == 16 ? 4 : (NC == 32 ? 5 : (NC == 64 ? 6 : 7));
  int col0 = 0;
  for (int c = c0; c < c0 + nch; ++c) {
    __syncthreads();  // chunk c has landed (the barrier's fence drains this wave's DMA; everyone's after the barrier) and every wave is done with slot (c+1)&1
    if (c + 1 < ntotal) issue_chunk(sched[c + 1], ring + ((c + 1) & 1) * RING_SLOT);
    const float* w = ring + (c & 1) * RING_SLOT;
    constexpr int ntiles = 2 * (NC >> 4);  // 2 token tiles x NC/16 column tiles
    for (int tile = wave; tile < ntiles; tile += NTHR / 64) {
      const int tt = tile & 1, ct = tile >> 1;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains: the 16x16x4 MFMA has a 40-cycle dependent latency
      const float* ip = in + (tt * 16 + fr) * ldin + 4 * fg;
      // weights of k = kb + 4 fg + e, column ct*16 + fr: linear index (k << LNC) + col in the [K][NC] chunk, DMA group = index >> 8
      auto wread = [&](int kb, int e) {
        const int lin = ((kb + 4 * fg + e) << LNC) + ct * 16 + fr;
        return w[(lin >> 8) * (256 + GPAD) + (lin & 255)];
      };
      f32x4 a = *reinterpret_cast<const f32x4*>(ip);
      float wv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) wv[e] = wread(0, e);
#pragma unroll 4
      for (int kb = 0; kb < Kd; kb += 16) {
        const int kn = kb + 16 < Kd ? kb + 16 : kb;  // fragments of the next 16-deep block are read while this one is multiplied
        const f32x4 an = *reinterpret_cast<const f32x4*>(ip + kn);
        float wn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) wn[e] = wread(kn, e);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], wv[0], acc0, 0, 0, 0);  // tokens: k = kb + 4 fg + e (the same permutation on the weights)
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], wv[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], wv[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], wv[3], acc1, 0, 0, 0);
        a = an;
#pragma unroll
        for (int e = 0; e < 4; ++e) wv[e] = wn[e];
      }
      const int o = col0 + ct * 16 + fr;  // accumulator: column = fr = output channel, rows 4 fg + r = token within the tile
      const float bv = bias ? bias[o] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = tt * 16 + 4 * fg + r;
        if (t < T) {
          float v = ((acc0[r] + acc1[r]) + bv) * scale;
          if (add) v += add[t * ldadd + o];
          if (ACT == 1) v = fmaxf(v, 0.f);
          if (ACT == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
          out[t * ldo + o] = v;
        }
      }
    }
    col0 += NC;
  }
  __syncthreads();  // `out` is complete
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
    for (int d = 0; d < HD; d += 4) {  // (rows are 16-byte aligned: ldq / ldk are multiples of 4 floats; same order of the 32 multiply-adds as one by one)
      const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + d), kv = *reinterpret_cast<const f32x4*>(kp + d);
      s = fmaf(qv[0], kv[0], s);
      s = fmaf(qv[1], kv[1], s);
      s = fmaf(qv[2], kv[2], s);
      s = fmaf(qv[3], kv[3], s);
    }
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
  for (int item = threadIdx.x; item < T * (H / 4); item += NTHR) {  // four channels per item: one 16-byte read of v per probability (same sums per channel)
    const int i = item / (H / 4), c = 4 * (item - i * (H / 4)), h = c / HD;
    const float* sp = S + (h * T + i) * T;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const float pj = sp[j];
      const f32x4 vv = *reinterpret_cast<const f32x4*>(v + j * ldv + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = fmaf(pj, vv[e], a[e]);
    }
    *reinterpret_cast<f32x4*>(ctx + i * ldc + c) = a;
  }
  __syncthreads();
}

// ---- packed weight layout of one encoder (floats) -----------------------------------------------------------------
// [Wemb_t Din x 128][bemb 128][pos 21 x 128] then per layer L (4x):
//   [Wqkv_t 128 x 384][bqkv 384][Wo_t 128 x 128][bo 128][ln1w 128][ln1b 128][Wi_t 128 x 16][bi 16][Wo2_t 16 x 128][bo2 128][ln2w 128][ln2b 128]
// then [Wcls_t 128 x 3][bcls 3][Wres_t Din x 3][bres 3]
constexpr int ENC_LAYER = 128 * 384 + 384 + 128 * 128 + 128 + 128 + 128 + 128 * 16 + 16 + 16 * 128 + 128 + 128 + 128;

__device__ __forceinline__ int add_chunks(Chunk* sched, int n, const float* W, int rows, int N, int nc) {
  for (int col = 0; col < N; col += nc) sched[n++] = Chunk{W + col, rows, N, nc};
  return n;
}

__global__ __launch_bounds__(NTHR) void tr_encoder_kernel(const float* __restrict__ x, int ldx, int Din, const float* __restrict__ W,
                                                         float* __restrict__ hout, float* __restrict__ score, float* __restrict__ score2, int s2_ld) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ldi = (Din + 15) & ~15;  // padded depth of the embedding Linear (zeros beyond Din)
  Chunk* sched = reinterpret_cast<Chunk*>(sm);  // [MAXCH] chunk schedule (all LDS is dynamic: the 160-KiB opt-in covers dynamic LDS only)
  int& nsched = *reinterpret_cast<int*>(sm + SCHED_FLOATS - 4);
  float* X0 = sm + SCHED_FLOATS;  // [T][ldi]   input tokens (kept for the residual head)
  float* Hb = X0 + T * ldi;      // [T][132]   hidden state
  float* QKV = Hb + T * 132;     // [T][388]
  float* CTX = QKV + T * 388;    // [T][132]
  float* T1 = CTX + T * 132;     // [T][132]
  float* IM = T1 + T * 132;      // [T][20]
  float* S = IM + T * 20;        // [NH*T*T]
  float* ring = S + NH * T * T + 4;  // [2][RING_SLOT]  (every token array above is followed by >= 11 rows of LDS: the MFMA row padding)
  const int b = blockIdx.x;
  const float* Wemb = W;
  const float* bemb = Wemb + Din * H;
  const float* pos = bemb + H;
  const float* L0 = pos + T * H;
  if (threadIdx.x == 0) {  // the chunk schedule of the whole stack
    int n = add_chunks(sched, 0, Wemb, Din, H, 32);
    const float* L = L0;
    for (int l = 0; l < 4; ++l, L += ENC_LAYER) {
      const float* Wqkv = L;
      const float* Wo = Wqkv + 128 * 384 + 384;
      const float* Wi = Wo + 128 * 128 + 128 + 128 + 128;
      const float* Wo2 = Wi + 128 * 16 + 16;
      n = add_chunks(sched, n, Wqkv, 128, 384, 64);
      n = add_chunks(sched, n, Wo, 128, 128, 64);
      n = add_chunks(sched, n, Wi, 128, 16, 16);
      n = add_chunks(sched, n, Wo2, 16, 128, 128);
    }
    nsched = n;
  }
  for (int i = threadIdx.x; i < 2 * RING_SLOT; i += NTHR) ring[i] = 0.f;  // (rows a short chunk never fills are multiplied by zero tokens: keep them finite)
  for (int i = threadIdx.x; i < T * ldi; i += NTHR) {
    const int t = i / ldi, k = i - t * ldi;
    X0[i] = k < Din ? x[((long)b * T + t) * ldx + k] : 0.f;
  }
  // h = Linear(x) + pos : feed pos through the "add" operand (T1 <- pos)
  for (int i = threadIdx.x; i < T * H; i += NTHR) T1[(i / H) * 132 + (i % H)] = pos[i];
  __syncthreads();
  const int ntotal = nsched;
  issue_chunk(sched[0], ring);
  int c = 0;
  linear<0, 32, 0>(X0, ldi, ldi, bemb, Hb, 132, T1, 132, 1.0f, sched, c, 4, ntotal, ring);
  c += 4;
  const float* L = L0;
  for (int l = 0; l < 4; ++l, L += ENC_LAYER) {
    const float* bqkv = L + 128 * 384;
    const float* bo = bqkv + 384 + 128 * 128;
    const float* ln1w = bo + 128;
    const float* ln1b = ln1w + 128;
    const float* bi = ln1b + 128 + 128 * 16;
    const float* bo2 = bi + 16 + 16 * 128;
    const float* ln2w = bo2 + 128;
    const float* ln2b = ln2w + 128;
    linear<0, 64, 128>(Hb, 132, H, bqkv, QKV, 388, nullptr, 0, 1.0f, sched, c, 6, ntotal, ring);
    c += 6;
    attention(QKV, 388, QKV + 128, 388, QKV + 256, 388, 0.17677669529663687f /* 1/sqrt(32) */, S, CTX, 132);
    linear<0, 64, 128>(CTX, 132, H, bo, T1, 132, Hb, 132, 1.0f, sched, c, 2, ntotal, ring);  // dense(ctx) + h
    c += 2;
    layernorm_tokens(T1, 132, ln1w, ln1b, 1e-12f, Hb, 132);  // h1
    __syncthreads();
    linear<2, 16, 128>(Hb, 132, H, bi, IM, 20, nullptr, 0, 1.0f, sched, c, 1, ntotal, ring);
    c += 1;
    linear<0, 128, 16>(IM, 20, 16, bo2, T1, 132, Hb, 132, 1.0f, sched, c, 1, ntotal, ring);  // dense(inter) + h1
    c += 1;
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
  Chunk* sched = reinterpret_cast<Chunk*>(sm);
  float* Q0 = sm + SCHED_FLOATS;  // [T][132] un-embedded query (residual)
  float* QE = Q0 + T * 132;     // [T][132] query + pos, later scratch
  float* KE = QE + T * 132;     // [T][132] key + pos
  float* Qp = KE + T * 132;     // [T][132] projected q
  float* CTX = Qp + T * 132;    // [T][132]
  float* KV = CTX + T * 132;    // [T][260]
  float* S = KV + T * 260;      // [NH*T*T]
  float* ring = S + NH * T * T + 4;  // [2][RING_SLOT]
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
  if (threadIdx.x == 0) {
    int n = add_chunks(sched, 0, Wq, H, H, 64);
    n = add_chunks(sched, n, Wkv, H, 256, 64);
    n = add_chunks(sched, n, Wo, H, H, 64);
    n = add_chunks(sched, n, W1, H, H, 64);
    n = add_chunks(sched, n, W2, H, H, 64);
  }
  constexpr int ntotal = 12;
  for (int i = threadIdx.x; i < T * H; i += NTHR) {
    const int t = i / H, c = i - t * H;
    const float qv = query[(long)b * T * H + i], kv = key[(long)b * T * H + i];
    Q0[t * 132 + c] = qv;
    QE[t * 132 + c] = qv + qpos[i];
    KE[t * 132 + c] = kv + kpos[i];
  }
  __syncthreads();
  issue_chunk(sched[0], ring);
  linear<0, 64, 128>(QE, 132, H, bq, Qp, 132, nullptr, 0, 0.17677669529663687f, sched, 0, 2, ntotal, ring);  // (Wq x + b) * head_dim^-1/2
  linear<0, 64, 128>(KE, 132, H, bkv, KV, 260, nullptr, 0, 1.0f, sched, 2, 4, ntotal, ring);
  attention(Qp, 132, KV, 260, KV + 128, 260, 1.0f, S, CTX, 132);
  linear<0, 64, 128>(CTX, 132, H, bo, QE, 132, Q0, 132, 1.0f, sched, 6, 2, ntotal, ring);  // query + attn
  layernorm_tokens(QE, 132, n2w, n2b, 1e-5f, Q0, 132);        // x = norm2(.)
  __syncthreads();
  linear<1, 64, 128>(Q0, 132, H, b1, CTX, 132, nullptr, 0, 1.0f, sched, 8, 2, ntotal, ring);
  linear<0, 64, 128>(CTX, 132, H, b2, QE, 132, Q0, 132, 1.0f, sched, 10, 2, ntotal, ring);
  layernorm_tokens(QE, 132, n3w, n3b, 1e-5f, CTX, 132);
  __syncthreads();
  for (int i = threadIdx.x; i < T * H; i += NTHR) out[((long)b * T + i / H) * ldo + ocoff + (i % H)] = CTX[(i / H) * 132 + (i % H)];
}

}  // namespace

extern "C" int kpf_tr_encoder_f32(const float* x, int ldx, int Din, const float* W, float* h, float* score, float* score2,
                                  int score2_ld, int B, void* stream) {
  KPF_REQUIRE(x && W && h && score && B > 0 && Din > 0 && Din <= 256 && ldx >= Din, "kpf_tr_encoder_f32: bad arguments");
  const size_t lds = (size_t)(21 * ((Din + 15) & ~15) + 21 * 132 * 3 + 21 * 388 + 4 * 21 * 21 + 21 * 20 + 4 + 2 * RING_SLOT + SCHED_FLOATS) * sizeof(float);
  static std::atomic<bool> lds_opt_in[KPF_MAX_DEVICES];
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(tr_encoder_kernel), lds_opt_in)) {
    kpf_set_error("kpf_tr_encoder_f32: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  KPF_REQUIRE(lds <= 160 * 1024, "kpf_tr_encoder_f32: Din=%d needs too much LDS", Din);
  hipLaunchKernelGGL(tr_encoder_kernel, dim3(B), dim3(NTHR), lds, reinterpret_cast<hipStream_t>(stream), x, ldx, Din, W, h, score,
                     score2, score2_ld);
  return kpf_check_launch("kpf_tr_encoder_f32");
}

extern "C" int kpf_xattn_layer_f32(const float* query, const float* key, const float* W, float* out, int out_ld, int out_coff, int B,
                                   void* stream) {
  KPF_REQUIRE(query && key && W && out && B > 0 && out_coff + 128 <= out_ld, "kpf_xattn_layer_f32: bad arguments");
  const size_t lds = (size_t)(21 * 132 * 5 + 21 * 260 + 4 * 21 * 21 + 4 + 2 * RING_SLOT + SCHED_FLOATS) * sizeof(float);
  static std::atomic<bool> lds_opt_in[KPF_MAX_DEVICES];
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(xattn_layer_kernel), lds_opt_in)) {
    kpf_set_error("kpf_xattn_layer_f32: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  hipLaunchKernelGGL(xattn_layer_kernel, dim3(B), dim3(NTHR), lds, reinterpret_cast<hipStream_t>(stream), query, key, W, out, out_ld,
                     out_coff);
  return kpf_check_launch("kpf_xattn_layer_f32");
}

extern "C" int kpf_tr_encoder_weight_floats(int Din) { return Din * 128 + 128 + 21 * 128 + 4 * ENC_LAYER + 128 * 3 + 3 + Din * 3 + 3; }
extern "C" int kpf_xattn_weight_floats(void) { return 2 * 21 * 128 + 128 * 128 + 128 + 128 * 256 + 256 + 128 * 128 + 128 + 256 + 128 * 128 + 128 + 128 * 128 + 128 + 256; }
