// The four post-LN BERT layers of a KP_Interaction_TR stack in TRAIN mode as ONE launch forward and ONE launch backward
// (model/model.py:30-126 via transformers' BertEncoder: 21 tokens x 128, 4 heads x 32, intermediate 16, GELU-erf, LayerNorm eps 1e-12,
// hidden / attention dropout): round 6, VERDICT r05 item 1(b).  The unfused training step ran a stack as ~30 launches forward and ~40
// backward of 5-10 us each on 672 rows (B = 32) — a dependent chain of ~350 us per stack and iteration that no kernel in it could shorten.
//
// One workgroup owns one sample: the sample's 21 x 128 hidden state, Q | K | V, scores and gradients live in LDS for the whole stack.  The
// workgroup has TWELVE waves with fixed roles:
//   * waves 0-7 (consumers) multiply on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fmaf chains; tokens are the rows, 21 padded
//     to two 16-row tiles), run softmax / LayerNorm / GELU and their backward on the vector ALUs, and write what the other direction (and the
//     deferred weight-gradient launch) needs to HBM;
//   * waves 8-11 (loaders) do nothing but stream the weights: every Linear is cut into 64 x 64 blocks of its [N][K] parameter tensor AS IT
//     LIES IN MEMORY (no packed copies: a training step rewrites the weights every iteration), the blocks of the whole stack form one schedule,
//     and the loaders keep THREE 16-KiB blocks in flight (global_load_lds_dwordx4 into a four-slot LDS ring) ahead of the block being
//     multiplied — across Linear / attention / LayerNorm boundaries.  Because the loaders execute no other vector-memory instruction, their
//     counted `s_waitcnt vmcnt` is exact whatever the consumers store meanwhile (vmcnt is per wave); the inference kernels of kpf_tr.hip let
//     every wave issue and drain at each block, i.e. ONE block in flight: 127 us per stack, bound by the latency of 44 dependent fetches.
//   * one raw s_barrier per block hands it from the loaders to the consumers and the slot of the previous block back.
// The same [N][K] block serves both directions: forward reads it "rows = output channel" (one ds_read_b128 per fragment), backward reads it
// "rows = reduction index" (four ds_read_b32), both bank-conflict-free under one XOR swizzle applied to the DMA's SOURCE addresses.
//
// Dropout masks are the counter-based hash of kpf_train.hip (seed, counter, call id, element index) and are RECOMPUTED in the backward: no mask
// bytes are stored.  Weight gradients are not computed here: the backward writes every Linear's dY beside the X the forward kept, and the
// deferred grouped launch (kpf_linear_wgrad_grouped) turns them into dW / db after backward as before; LayerNorm parameter gradients leave as
// per-sample partial sums for kpf_colsum_reduce_grouped.  Every sum has a fixed order: replays are bit-identical.
#include "kpf_common.h"

namespace {

constexpr int T = 21, H = 128, NH = 4, HD = 32, FF = 16;
constexpr int NCW = 8;                  // consumer waves
constexpr int NLW = 4;                  // loader waves
constexpr int NTHR = 64 * (NCW + NLW);  // 768
constexpr int NCT = 64 * NCW;           // consumer threads
constexpr int SLOT = 64 * 64;           // floats per ring slot: one 64 x 64 block
constexpr int NSF = 6;                  // ring slots of the forward kernel: the block in use + five in flight (its token arrays share one LDS region, below)
constexpr int NSB = 4;                  // ring slots of the backward kernel: the block in use + three in flight
constexpr int LDA = 136;                // row stride of the 128-wide token arrays (34 x 16 B: conflict-free ds_read_b128 fragments)
constexpr int LDQ = 392;                // q | k | v rows
constexpr int LDI = 72;                 // intermediate (16 real columns, zeros up to 64: one block deep)
constexpr int NLAYER = 4;
constexpr int CH_PER_LAYER = 20;        // 12 (q | k | v) + 4 (attention output) + 2 (intermediate) + 2 (output)
constexpr int NCHUNK = NLAYER * CH_PER_LAYER;
constexpr int SCHED_F = NCHUNK * 6;     // Chunk is 24 bytes
constexpr int SPAD = 1792;              // 4 x 21 x 21 = 1764 scores, padded (7 x 256: whole DMA instructions)
constexpr int QPAD = 8448;              // 21 x 392 = 8232 floats of q | k | v, padded to 33 x 256 (backward: the array is filled by LDS-DMA)

struct Chunk {
  const float* src;  // element (0, 0) of the block inside its parameter tensor
  int ld;            // floats between rows of the tensor
  int rows, cols;    // valid extent of the block (<= 64 each, cols % 4 == 0); the rest of the slot is filled from a zero page
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;

__device__ __attribute__((aligned(16))) float kpf_trs_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// tuning aid: wall-clock stamps (100 MHz) of workgroup 0's thread 0 at phase boundaries, into a buffer set by kpf_tr_stack_set_stamps (NULL = off, the default)
__device__ unsigned long long* kpf_trs_stamps = nullptr;
__device__ int kpf_trs_dbg = 0;  // tuning aid (kpf_tr_stack_set_stamps' second argument): 1 = the loaders issue no DMA, 2 = the consumers skip the products
#define TRS_STAMP(i)                                                                                         \
  do {                                                                                                       \
    if (kpf_trs_stamps && blockIdx.x == 0 && threadIdx.x == 0) kpf_trs_stamps[(i)] = wall_clock64();         \
  } while (0)

__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 2)) & 15; }
__device__ __forceinline__ void BAR() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned hash32(unsigned x) {  // "lowbias32" (kpf_train.hip)
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
struct Drop {
  unsigned base, thr;
  float ks;
  bool on;
  __device__ __forceinline__ bool keep(int call, unsigned idx) const { return !on || hash32(base ^ hash32((unsigned)call * 0x85ebca6bU + idx)) >= thr; }
};
__device__ __forceinline__ Drop make_drop(float p, unsigned seed, unsigned ctr) {
  Drop d;
  d.on = p > 0.f;
  d.ks = d.on ? 1.0f / (1.0f - p) : 1.0f;
  d.thr = d.on ? (unsigned)fminf(p * 4294967296.0f, 4294967295.0f) : 0u;
  d.base = hash32(seed ^ (ctr * 0x9e3779b9U));
  return d;
}

__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float v) {
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
  return cdf + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}

// ---- the saved-activation buffer of one stack (floats; M = 21 B rows) -----------------------------------------------------------------
//   H[l], l = 0..4 : [M][128]   hidden state entering layer l (H[0] = dropout(e + pos), H[4] = the stack's output)
//   per layer (1020 M): qkv [M][384] | P [B][4][21][21] = 84 M (softmax, before dropout) | ctx [M][128] | xs1 [M][128] (h + dropout(dense(ctx))) |
//                       st1 [M][4] (mean, rstd) | h1 [M][128] | it [M][16] (before GELU) | g [M][16] | xs2 [M][128] | st2 [M][4]
//   then 4 floats: (seed, counter) of the forward's dropout masks as two unsigned
constexpr long SV_LAYER = 1020;
struct Save {
  long M;
  __device__ __host__ long Hs(int l) const { return (long)l * M * 128; }
  __device__ __host__ long L(int l) const { return 5 * M * 128 + (long)l * M * SV_LAYER; }
  __device__ __host__ long qkv(int l) const { return L(l); }
  __device__ __host__ long P(int l) const { return L(l) + 384 * M; }
  __device__ __host__ long ctx(int l) const { return L(l) + 468 * M; }
  __device__ __host__ long xs1(int l) const { return L(l) + 596 * M; }
  __device__ __host__ long st1(int l) const { return L(l) + 724 * M; }
  __device__ __host__ long h1(int l) const { return L(l) + 728 * M; }
  __device__ __host__ long it(int l) const { return L(l) + 856 * M; }
  __device__ __host__ long g(int l) const { return L(l) + 872 * M; }
  __device__ __host__ long xs2(int l) const { return L(l) + 888 * M; }
  __device__ __host__ long st2(int l) const { return L(l) + 1016 * M; }
  __device__ __host__ long rng() const { return L(NLAYER); }
  __device__ __host__ long total() const { return L(NLAYER) + 4; }
};
// dY buffer of the backward (per layer 656 M): dqkv [M][384] | do1 [M][128] | dit [M][16] | do2 [M][128];  LayerNorm partials: [layer][2 (LN1, LN2)][B][2][128]
constexpr long DY_LAYER = 656;

// parameter table of one stack: 16 pointers per layer
enum { PW_Q = 0, PB_Q, PW_K, PB_K, PW_V, PB_V, PW_O, PB_O, P_G1, P_B1, PW_I, PB_I, PW_O2, PB_O2, P_G2, P_B2, P_PER_LAYER };

struct Sync {
  float* ring;
  const Chunk* sched;
  const float* zero;
  int ntotal;  // blocks in the schedule
  int c;       // next block of the schedule
  int dbg;
  int lane, wave;
  bool loader;
};

template <int NS>
__device__ __forceinline__ void issue_chunk(const Sync& s, int n) {
  const Chunk ck = s.sched[n];
  float* slot = s.ring + (n % NS) * SLOT;
  const int lw = s.wave - NCW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int g = lw * 4 + i;              // 1-KiB DMA group: rows 4 g .. 4 g + 3 of the block
    const int row = g * 4 + (s.lane >> 4);
    const int lc = (s.lane & 15) ^ swz(row);  // the logical 16-byte piece that LDS position (row, lane & 15) holds
    const bool ok = row < ck.rows && 4 * lc < ck.cols;
    const float* src = ok ? ck.src + (long)row * ck.ld + 4 * lc : s.zero;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(slot + g * 256), 16, 0, 0);
  }
}

// Every wave calls this once per block, in schedule order: afterwards block s.c is readable in slot s.c % NS and (loaders) block s.c + NS - 1 is in flight.
template <int NS>
__device__ __forceinline__ void chunk_sync(Sync& s) {
  static_assert(NS >= 3 && NS <= 6, "vmcnt immediates below");
  if (s.loader) {
    const int younger = min(s.ntotal - 1 - s.c, NS - 2);  // blocks issued after s.c so far: four DMA instructions per loader wave each
    if (younger >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  BAR();  // block s.c has landed for everyone; every consumer is done with block s.c - 1
  if (s.loader && s.c + NS - 1 < s.ntotal && !(s.dbg & 1)) issue_chunk<NS>(s, s.c + NS - 1);
}

// one 64-deep block of a product: tokens in[t][roff .. roff + 63] times the block in `slot`.  KN = false: out column = block row (forward, W[n][k]);
// KN = true: out column = block column, reduction over block rows (backward, dX = dY W).
// MMA: 0 = v_mfma_f32_16x16x4_f32 (exact fp32 products: the fp32 training step and every parity test), 1 / 2 = the operands rounded to bf16 / f16 in registers and ONE
// v_mfma_f32_16x16x16_{bf16,f16} per 16-deep step, fp32 accumulation — what torch.autocast does to these Linears — for the mixed-precision step: the fp32
// matrix pipe takes 1024 cycles per block for the two waves of a SIMD (21 tokens padded to 32 rows), which was two thirds of a stack's time.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <bool KN, int MMA>
__device__ __forceinline__ void mma_chunk(const float* in, int ldin, int roff, const float* slot, int tt, int ct, int fr, int fg, f32x4& acc0, f32x4& acc1) {
  const float* ip = in + (tt * 16 + fr) * ldin + roff + 4 * fg;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ip + ks * 16);
    f32x4 b;
    if (!KN) {
      const int row = ct * 16 + fr;
      b = *reinterpret_cast<const f32x4*>(slot + row * 64 + (((ks * 4 + fg) ^ swz(row)) << 2));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = ks * 16 + 4 * fg + e;
        b[e] = slot[row * 64 + (((ct * 4 + (fr >> 2)) ^ swz(row)) << 2) + (fr & 3)];
      }
    }
    if constexpr (MMA == 0) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);  // (both operands use k = 16 ks + 4 fg + e: a permutation of the sum)
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
    } else if constexpr (MMA == 1) {  // lane (l % 16, l / 16) supplies k = 4 (l / 16) .. + 3 of its row / column: exactly the four values it holds
      const bf16x4 ah = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]}, bh = {(bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
      f32x4& acc = (ks & 1) ? acc1 : acc0;
      acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ah), __builtin_bit_cast(s16x4, bh), acc, 0, 0, 0);
    } else {
      const f16x4 ah = {(f16_t)a[0], (f16_t)a[1], (f16_t)a[2], (f16_t)a[3]}, bh = {(f16_t)b[0], (f16_t)b[1], (f16_t)b[2], (f16_t)b[3]};
      f32x4& acc = (ks & 1) ? acc1 : acc0;
      acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc, 0, 0, 0);
    }
  }
}

// nob output blocks of 64 columns, each the sum of nrb 64-deep blocks (schedule order: output block major).  epi(ob, t, col, value) is called for the
// 21 real tokens; t = token, col = ob * 64 + column within the block.
template <int NS, bool KN, int MMA, class Epi>
__device__ __forceinline__ void gemm_op(Sync& s, const float* in, int ldin, int nob, int nrb, Epi&& epi) {
  const int fr = s.lane & 15, fg = s.lane >> 4;
  const int tt = s.wave & 1, ct = (s.wave >> 1) & 3;
  for (int ob = 0; ob < nob; ++ob) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int rb = 0; rb < nrb; ++rb) {
      chunk_sync<NS>(s);
      if (!s.loader && !(s.dbg & 2)) mma_chunk<KN, MMA>(in, ldin, rb * 64, s.ring + (s.c % NS) * SLOT, tt, ct, fr, fg, acc0, acc1);
      ++s.c;
    }
    if (!s.loader) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = tt * 16 + 4 * fg + r;
        if (t < T) epi(ob, t, ob * 64 + ct * 16 + fr, acc0[r] + acc1[r]);
      }
    }
  }
}

// The attention core of one sample, all four heads, on the consumer waves (every wave of the workgroup calls it: the barriers are the workgroup's):
// S = q k^T / sqrt(32) -> row softmax (saved to Pg, before dropout) -> dropout -> ctx = S v, written to `ctx` (LDS, row stride ldc) and to ctx_g (HBM rows of 128).
// q | k | v are the column blocks of QKV [T][LDQ]; element numbering of the dropout hash = the unfused kernel's ((b * 4 + h) * 441 + i * 21 + j).
__device__ __forceinline__ void attention_forward(const Sync& s, bool work, const float* QKV, float* S, float* ctx, int ldc, float* __restrict__ Pg,
                                                  float* __restrict__ ctx_g, int b, long row0, const Drop& dr, int call) {
  const int tid = threadIdx.x;
  BAR();
  if (work) {
    for (int item = tid; item < NH * T * T; item += NCT) {
      const int h = item / (T * T), r = item - h * T * T, i = r / T, j = r - i * T;
      const float* qp = QKV + i * LDQ + h * HD;
      const float* kp = QKV + j * LDQ + H + h * HD;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + d), kv = *reinterpret_cast<const f32x4*>(kp + d);
        a = fmaf(qv[0], kv[0], a);
        a = fmaf(qv[1], kv[1], a);
        a = fmaf(qv[2], kv[2], a);
        a = fmaf(qv[3], kv[3], a);
      }
      S[item] = a * 0.17677669529663687f;
    }
  }
  BAR();
  if (work && tid < NH * T) {
    float* sp = S + tid * T;
    float mx = -INFINITY;
    for (int j = 0; j < T; ++j) mx = fmaxf(mx, sp[j]);
    float se = 0.f;
    for (int j = 0; j < T; ++j) {
      const float ev = __expf(sp[j] - mx);
      sp[j] = ev;
      se += ev;
    }
    const float inv = 1.0f / se;
    const long pbase = ((long)b * NH + tid / T) * (T * T) + (tid % T) * T;  // element index of (b, h, i, 0): the unfused kernel's numbering
    float* pg = Pg + pbase;
    for (int j = 0; j < T; ++j) {
      const float pv = sp[j] * inv;
      pg[j] = pv;
      sp[j] = dr.keep(call, (unsigned)(pbase + j)) ? pv * dr.ks : 0.f;
    }
  }
  BAR();
  if (work) {  // (q and k are dead: the context may go to their columns)
    for (int item = tid; item < T * (H / 4); item += NCT) {
      const int i = item >> 5, c = (item & 31) * 4, h = c / HD;
      const float* sp = S + (h * T + i) * T;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 3
      for (int j = 0; j < T; ++j) {
        const float pj = sp[j];
        const f32x4 vv = *reinterpret_cast<const f32x4*>(QKV + j * LDQ + 2 * H + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = fmaf(pj, vv[k], a[k]);
      }
      *reinterpret_cast<f32x4*>(ctx + i * ldc + c) = a;
      kpf_st4(ctx_g + (row0 + i) * H + c, a);
    }
  }
}

// ================================================================ forward ================================================================
// per-layer parameter VECTORS (biases, LayerNorm weights) staged in LDS: an epilogue that fetched its bias from HBM / L2 exposed ~1 us of load latency eleven
// times per layer.  Layer l + 1's vectors are requested into registers at the top of layer l and written to LDS at its end.
enum { PV_BQKV = 0, PV_BO = 384, PV_G1 = 512, PV_B1 = 640, PV_BI = 768, PV_BO2 = 784, PV_G2 = 912, PV_B2 = 1040, PV_N = 1168, PV_PAD = 1184 };
__device__ __forceinline__ float pv_fetch(const float* const* pl, int i) {
  if (i < PV_BO) return pl[2 * (i >> 7) + 1][i & 127];
  if (i < PV_G1) return pl[PB_O][i - PV_BO];
  if (i < PV_B1) return pl[P_G1][i - PV_G1];
  if (i < PV_BI) return pl[P_B1][i - PV_B1];
  if (i < PV_BO2) return pl[PB_I][i - PV_BI];
  if (i < PV_G2) return pl[PB_O2][i - PV_BO2];
  if (i < PV_B2) return pl[P_G2][i - PV_G2];
  return pl[P_B2][i - PV_B2];
}

template <int MMA>
__global__ __launch_bounds__(NTHR) void tr_stack_fwd_kernel(const float* __restrict__ e, const float* __restrict__ pos, const float* const* __restrict__ P,
                                                          float* __restrict__ save, int B, float p_drop, const long* __restrict__ rng, int call0) {
  constexpr int NS = NSF;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  Chunk* sched = reinterpret_cast<Chunk*>(sm);
  float* PV = sm + SCHED_F;     // [PV_PAD] this layer's parameter vectors
  float* Hb = PV + PV_PAD;      // [T][LDA] hidden state
  float* QKV = Hb + T * LDA;    // [T][LDQ] q | k | v; after the scores exist the q columns hold T1 (the LayerNorm's input), after the attention the k columns held
  float* T1 = QKV;              //          CTX and the v columns hold IM (gelu(intermediate), 16 real columns, zeros up to 64): same rows, stride LDQ
  float* CTX = QKV + H;
  float* IM = QKV + 2 * H;
  float* S = QKV + T * LDQ;     // [NH][T][T]
  float* ring = S + SPAD;       // [NS][SLOT]   (every token array is followed by >= 11 more rows of LDS: the second MFMA row tile reads them)
  const int tid = threadIdx.x;
  Sync s;
  s.lane = tid & 63;
  s.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  s.loader = s.wave >= NCW;
  s.ring = ring;
  s.sched = sched;
  s.zero = kpf_trs_zero16;
  s.c = 0;
  s.ntotal = NCHUNK;
  s.dbg = kpf_trs_dbg;
  const bool work = !s.loader;
  const int b = blockIdx.x;
  Save sv;
  sv.M = (long)B * T;
  const long row0 = (long)b * T;

  if (tid == 0) {
    int n = 0;
    for (int l = 0; l < NLAYER; ++l) {
      const float* const* pl = P + l * P_PER_LAYER;
      for (int ob = 0; ob < 6; ++ob)
        for (int rb = 0; rb < 2; ++rb) sched[n++] = Chunk{pl[2 * (ob >> 1)] + (long)(ob & 1) * 64 * H + rb * 64, H, 64, 64};
      for (int ob = 0; ob < 2; ++ob)
        for (int rb = 0; rb < 2; ++rb) sched[n++] = Chunk{pl[PW_O] + (long)ob * 64 * H + rb * 64, H, 64, 64};
      for (int rb = 0; rb < 2; ++rb) sched[n++] = Chunk{pl[PW_I] + rb * 64, H, FF, 64};
      for (int ob = 0; ob < 2; ++ob) sched[n++] = Chunk{pl[PW_O2] + (long)ob * 64 * FF, FF, 64, FF};
    }
  }
  const unsigned seed = (p_drop > 0.f && rng) ? (unsigned)rng[0] : 0u, ctr = (p_drop > 0.f && rng) ? (unsigned)rng[1] : 0u;
  const Drop dr = make_drop(p_drop, seed, ctr);
  if (b == 0 && tid == 0) {
    unsigned* ru = reinterpret_cast<unsigned*>(save + sv.rng());
    ru[0] = seed;
    ru[1] = ctr;
  }
  float pvr[3] = {0.f, 0.f, 0.f};  // this thread's three elements of the NEXT layer's parameter vectors
  if (work) {
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (tid + u * NCT < PV_N) PV[tid + u * NCT] = pv_fetch(P, tid + u * NCT);
    // H[0] = dropout(e + pos)  (TR_Encoder: embedding + position, then the embedding dropout; model/model.py:78-84)
    for (int i = tid; i < T * (H / 4); i += NCT) {
      const int t = i >> 5, c = (i & 31) * 4;
      const long off = (row0 + t) * H + c;
      const f32x4 ev = kpf_ld4(e + off), pv = kpf_ld4(pos + t * H + c);
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float x = ev[k] + pv[k];
        v[k] = dr.keep(call0 + 12, (unsigned)(off + k)) ? x * dr.ks : 0.f;
      }
      *reinterpret_cast<f32x4*>(Hb + t * LDA + c) = v;
      kpf_st4(save + sv.Hs(0) + off, v);
    }
  }
  BAR();  // schedule, parameter vectors and H[0] visible
  if (s.loader)
    for (int n = 0; n < NS - 1; ++n)
      if (!(s.dbg & 1)) issue_chunk<NS>(s, n);

  for (int l = 0; l < NLAYER; ++l) {
    TRS_STAMP(l * 8 + 0);
    if (work && l + 1 < NLAYER) {
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (tid + u * NCT < PV_N) pvr[u] = pv_fetch(P + (l + 1) * P_PER_LAYER, tid + u * NCT);
    }
    // ---- q | k | v = h W^T + b ----
    {
      float* sq = save + sv.qkv(l);
      gemm_op<NS, false, MMA>(s, Hb, LDA, 6, 2, [&](int, int t, int col, float acc) {
        const float v = acc + PV[PV_BQKV + col];
        QKV[t * LDQ + col] = v;
        sq[(row0 + t) * 384 + col] = v;
      });
    }
    // ---- attention core: softmax(q k^T / sqrt(32)) -> dropout -> . v ----
    TRS_STAMP(l * 8 + 1);
    attention_forward(s, work, QKV, S, CTX, LDQ, save + sv.P(l), save + sv.ctx(l), b, row0, dr, call0 + 3 * l);
    // ---- xs1 = h + dropout(ctx Wo^T + bo)  (into the q columns) ----
    TRS_STAMP(l * 8 + 2);
    {
      float* sx = save + sv.xs1(l);
      gemm_op<NS, false, MMA>(s, CTX, LDQ, 2, 2, [&](int, int t, int col, float acc) {
        const float o = acc + PV[PV_BO + col];
        const long idx = (row0 + t) * H + col;
        const float x = Hb[t * LDA + col] + (dr.keep(call0 + 3 * l + 1, (unsigned)idx) ? o * dr.ks : 0.f);
        T1[t * LDQ + col] = x;
        sx[idx] = x;
      });
    }
    // ---- h1 = LayerNorm(xs1); intermediate; output; h = LayerNorm(xs2) ----
    for (int pass = 0; pass < 2; ++pass) {
      BAR();
      TRS_STAMP(l * 8 + 3 + 3 * pass);
      if (work) {
        const float* gw = PV + (pass == 0 ? PV_G1 : PV_G2);
        const float* gb = PV + (pass == 0 ? PV_B1 : PV_B2);
        float* st = save + (pass == 0 ? sv.st1(l) : sv.st2(l));
        float* hs = save + (pass == 0 ? sv.h1(l) : sv.Hs(l + 1));
        const float w0 = gw[s.lane], w1 = gw[64 + s.lane], b0 = gb[s.lane], b1 = gb[64 + s.lane];
        for (int t = s.wave; t < T; t += NCW) {
          const float a0 = T1[t * LDQ + s.lane], a1 = T1[t * LDQ + 64 + s.lane];
          const float mean = wave_sum(a0 + a1) * (1.0f / H);
          const float d0 = a0 - mean, d1 = a1 - mean;
          const float rstd = 1.0f / sqrtf(wave_sum(fmaf(d0, d0, d1 * d1)) * (1.0f / H) + 1e-12f);
          const float y0 = d0 * rstd * w0 + b0, y1 = d1 * rstd * w1 + b1;
          Hb[t * LDA + s.lane] = y0;
          Hb[t * LDA + 64 + s.lane] = y1;
          hs[(row0 + t) * H + s.lane] = y0;
          hs[(row0 + t) * H + 64 + s.lane] = y1;
          if (s.lane == 0) {
            st[(row0 + t) * 4] = mean;
            st[(row0 + t) * 4 + 1] = rstd;
          }
        }
        if (pass == 0)  // the v columns are dead since the attention: IM's zero padding (columns 16 .. 63 of its block)
          for (int i = tid; i < T * 48; i += NCT) IM[(i / 48) * LDQ + 16 + i % 48] = 0.f;
      }
      if (pass == 1) break;
      // ---- it = h1 Wi^T + bi; g = gelu(it) ----
      TRS_STAMP(l * 8 + 4);
      {
        float* si = save + sv.it(l);
        float* sg = save + sv.g(l);
        gemm_op<NS, false, MMA>(s, Hb, LDA, 1, 2, [&](int, int t, int col, float acc) {
          if (col < FF) {
            const float iv = acc + PV[PV_BI + col];
            const float gv = gelu_exact(iv);
            IM[t * LDQ + col] = gv;
            si[(row0 + t) * FF + col] = iv;
            sg[(row0 + t) * FF + col] = gv;
          }
        });
      }
      // ---- xs2 = h1 + dropout(g Wo2^T + bo2) ----
      TRS_STAMP(l * 8 + 5);
      {
        float* sx = save + sv.xs2(l);
        gemm_op<NS, false, MMA>(s, IM, LDQ, 2, 1, [&](int, int t, int col, float acc) {
          const float o = acc + PV[PV_BO2 + col];
          const long idx = (row0 + t) * H + col;
          const float x = Hb[t * LDA + col] + (dr.keep(call0 + 3 * l + 2, (unsigned)idx) ? o * dr.ks : 0.f);
          T1[t * LDQ + col] = x;
          sx[idx] = x;
        });
      }
    }
    // next layer's parameter vectors: every read of this layer's is behind the LayerNorm above; the next layer's first block barrier publishes them
    BAR();
    TRS_STAMP(l * 8 + 7);
    if (work && l + 1 < NLAYER) {
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (tid + u * NCT < PV_N) PV[tid + u * NCT] = pvr[u];
    }
  }
}

// ================================================================ backward ===============================================================
// What a LayerNorm backward reads from HBM for the (up to) three rows of a wave — x = xs rows and their (mean, rstd) — is requested into registers one phase
// EARLIER than it is used (the loads of the next LayerNorm fly under the GEMM blocks in between): a dependent HBM read costs ~1.5 us, and a layer has six.
struct LnIn {
  float x0[3], x1[3], mean[3], rstd[3];
};
__device__ __forceinline__ void ln_prefetch(const Sync& s, LnIn& in, const float* __restrict__ xs, const float* __restrict__ st, long row0) {
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int t = s.wave + u * NCW;
    const long r = row0 + (t < T ? t : T - 1);
    in.x0[u] = xs[r * H + s.lane];
    in.x1[u] = xs[r * H + 64 + s.lane];
    in.mean[u] = st[r * 4];
    in.rstd[u] = st[r * 4 + 1];
  }
}
// Consumer-side LDS-DMA copy of a saved activation into an LDS array, no registers: instruction k fills LDS floats [256 k, 256 k + 256) of dst; a lane's 16 bytes
// come from row t = f / lld, column c = f % lld of the source (row stride gld floats, ncols valid columns) or from the zero page.  dst must extend to a multiple
// of 256 floats.  The issuing wave orders its later LDS reads with `s_waitcnt vmcnt(0)` + the workgroup barrier.
__device__ __forceinline__ void dma_rows(const Sync& s, const float* __restrict__ src, int nrows, int gld, int ncols, float* dst, int lld) {
  const int ninstr = (nrows * lld + 255) >> 8;
  for (int k = s.wave; k < ninstr; k += NCW) {
    const int f = k * 256 + s.lane * 4;
    const int t = f / lld, c = f - t * lld;
    const float* g = (t < nrows && c < ncols) ? src + (long)t * gld + c : s.zero;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)(dst + k * 256), 16, 0, 0);
  }
}
// One LayerNorm backward over the sample's 21 rows (wave per row): dy rows in `dyl` (LDS), x / stats in registers, gamma in LDS.
//   dxs -> dres (LDS) [the residual branch's gradient] and dout = dxs * keep / (1 - p) -> ddn (LDS) + HBM (the dense layer's dY);
//   per-sample partial sums of d gamma / d beta -> part[2][128] through `scratch` ([NCW][2][128] floats of LDS).
__device__ __forceinline__ void ln_backward(const Sync& s, bool work, const float* dyl, const LnIn& in, const float* gam, float* dres, float* ddn,
                                            float* __restrict__ ddn_g, float* __restrict__ part, float* scratch, long row0, const Drop& dr, int call) {
  BAR();
  if (work) {
    const float g0 = gam[s.lane], g1 = gam[64 + s.lane];
    float aw0 = 0.f, aw1 = 0.f, ab0 = 0.f, ab1 = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int t = s.wave + u * NCW;
      if (t < T) {
        const long r = row0 + t;
        const float mean = in.mean[u], rstd = in.rstd[u];
        const float d0 = dyl[t * LDA + s.lane], d1 = dyl[t * LDA + 64 + s.lane];
        const float xh0 = (in.x0[u] - mean) * rstd, xh1 = (in.x1[u] - mean) * rstd;
        const float gd0 = d0 * g0, gd1 = d1 * g1;
        const float m1 = wave_sum(gd0 + gd1) * (1.0f / H);
        const float m2 = wave_sum(fmaf(gd0, xh0, gd1 * xh1)) * (1.0f / H);
        aw0 = fmaf(d0, xh0, aw0);
        aw1 = fmaf(d1, xh1, aw1);
        ab0 += d0;
        ab1 += d1;
        const float o0 = rstd * (gd0 - m1 - xh0 * m2), o1 = rstd * (gd1 - m1 - xh1 * m2);
        dres[t * LDA + s.lane] = o0;
        dres[t * LDA + 64 + s.lane] = o1;
        const float q0 = dr.keep(call, (unsigned)(r * H + s.lane)) ? o0 * dr.ks : 0.f;
        const float q1 = dr.keep(call, (unsigned)(r * H + 64 + s.lane)) ? o1 * dr.ks : 0.f;
        ddn[t * LDA + s.lane] = q0;
        ddn[t * LDA + 64 + s.lane] = q1;
        ddn_g[r * H + s.lane] = q0;
        ddn_g[r * H + 64 + s.lane] = q1;
      }
    }
    scratch[(s.wave * 2 + 0) * H + s.lane] = aw0;
    scratch[(s.wave * 2 + 0) * H + 64 + s.lane] = aw1;
    scratch[(s.wave * 2 + 1) * H + s.lane] = ab0;
    scratch[(s.wave * 2 + 1) * H + 64 + s.lane] = ab1;
  }
  BAR();
  if (work && threadIdx.x < 2 * H) {
    const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NCW; ++w) a += scratch[(w * 2 + which) * H + c];
    part[which * H + c] = a;
  }
}

// Backward of attention_forward for one sample, in place: on entry QKV holds q | k | v and SP the saved probabilities (both brought in by the caller's
// LDS-DMA: this function waits for the calling wave's share), dctx the context's gradient (LDS, row stride ldd); on exit QKV holds dq | dk | dv, also stored to
// dqkv_g (HBM rows of 384).  SP's sign carries the dropout decision after the first phase.
__device__ __forceinline__ void attention_backward(const Sync& s, bool work, float* QKV, float* SP, float* SD, const float* dctx, int ldd,
                                                   float* __restrict__ dqkv_g, int b, long row0, const Drop& dr, int call) {
  const int tid = threadIdx.x;
  if (work) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of q | k | v and P has landed in LDS
  BAR();
  if (work) {  // dP' = d ctx V^T; SD = dP' * keep / (1 - p); SP <- +-P (sign: kept / dropped)
    const long pb0 = (long)b * NH * T * T;
    for (int item = tid; item < NH * T * T; item += NCT) {
      const int h = item / (T * T), r = item - h * T * T, i = r / T, j = r - i * T;
      const float* gp = dctx + i * ldd + h * HD;
      const float* vp = QKV + j * LDQ + 2 * H + h * HD;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + d), vv = *reinterpret_cast<const f32x4*>(vp + d);
        a = fmaf(gv[0], vv[0], a);
        a = fmaf(gv[1], vv[1], a);
        a = fmaf(gv[2], vv[2], a);
        a = fmaf(gv[3], vv[3], a);
      }
      const bool kp = dr.keep(call, (unsigned)(pb0 + item));
      SD[item] = kp ? a * dr.ks : 0.f;
      if (!kp) SP[item] = -SP[item];
    }
  }
  BAR();
  if (work) {
    // dV[j][c] = sum_i P'[i][j] d ctx[i][c]  (V is dead after the phase above: written in place)
    for (int item = tid; item < T * (H / 4); item += NCT) {
      const int j = item >> 5, c = (item & 31) * 4, h = c / HD;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 3
      for (int i = 0; i < T; ++i) {
        const float pv = fmaxf(SP[(h * T + i) * T + j], 0.f) * dr.ks;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(dctx + i * ldd + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = fmaf(pv, gv[k], a[k]);
      }
      *reinterpret_cast<f32x4*>(QKV + j * LDQ + 2 * H + c) = a;
    }
    // dS = P (dP - <dP, P>) / sqrt(32), row by row
    if (tid < NH * T) {
      float* sd = SD + tid * T;
      const float* sp = SP + tid * T;
      float dot = 0.f;
      for (int j = 0; j < T; ++j) dot = fmaf(sd[j], fabsf(sp[j]), dot);
      for (int j = 0; j < T; ++j) sd[j] = fabsf(sp[j]) * (sd[j] - dot) * 0.17677669529663687f;
    }
  }
  BAR();
  {
    // dQ[i] = sum_j dS[i][j] K[j], dK[i] = sum_j dS[j][i] Q[j]: into registers, then in place
    f32x4 rq[2], rk[2];
    if (work) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int item = tid + u * NCT;
        rq[u] = rk[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (item < T * (H / 4)) {
          const int i = item >> 5, c = (item & 31) * 4, h = c / HD;
#pragma unroll 3
          for (int j = 0; j < T; ++j) {
            const float a = SD[(h * T + i) * T + j], bq = SD[(h * T + j) * T + i];
            const f32x4 kv = *reinterpret_cast<const f32x4*>(QKV + j * LDQ + H + c), qv = *reinterpret_cast<const f32x4*>(QKV + j * LDQ + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              rq[u][k] = fmaf(a, kv[k], rq[u][k]);
              rk[u][k] = fmaf(bq, qv[k], rk[u][k]);
            }
          }
        }
      }
    }
    BAR();
    if (work) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int item = tid + u * NCT;
        if (item < T * (H / 4)) {
          const int i = item >> 5, c = (item & 31) * 4;
          *reinterpret_cast<f32x4*>(QKV + i * LDQ + c) = rq[u];
          *reinterpret_cast<f32x4*>(QKV + i * LDQ + H + c) = rk[u];
        }
      }
    }
  }
  BAR();
  if (work) {
    for (int i = tid; i < T * (384 / 4); i += NCT) {
      const int t = i / 96, c = (i - t * 96) * 4;
      kpf_st4(dqkv_g + (row0 + t) * 384 + c, *reinterpret_cast<const f32x4*>(QKV + t * LDQ + c));
    }
  }
}

template <int MMA>
__global__ __launch_bounds__(NTHR) void tr_stack_bwd_kernel(const float* __restrict__ dh, const float* const* __restrict__ P, const float* __restrict__ save,
                                                          float* __restrict__ dE, float* __restrict__ dys, float* __restrict__ parts, int B, float p_drop, int call0) {
  constexpr int NS = NSB;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  Chunk* sched = reinterpret_cast<Chunk*>(sm);
  float* PV = sm + SCHED_F;    // [2][128] LayerNorm weights of this layer (ln1, ln2)
  float* G = PV + 2 * H;       // [T][LDA] gradient of the layer's output, later of its input
  float* D1 = G + T * LDA;     // [T][LDA]
  float* D2 = D1 + T * LDA;    // [T][LDA]
  float* DI = D2 + T * LDA;    // [T][LDI] d intermediate, zero beyond column 15
  float* QKV = DI + T * LDI;   // [T][LDQ] q | k | v, overwritten by dq | dk | dv
  float* SP = QKV + QPAD;      // [NH][T][T] probabilities (sign = dropped); before that the LayerNorm reduce scratch (with SD)
  float* SD = SP + SPAD;       // [NH][T][T]
  float* ring = SD + SPAD;
  const int tid = threadIdx.x;
  Sync s;
  s.lane = tid & 63;
  s.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  s.loader = s.wave >= NCW;
  s.ring = ring;
  s.sched = sched;
  s.zero = kpf_trs_zero16;
  s.c = 0;
  s.ntotal = NCHUNK;
  s.dbg = kpf_trs_dbg;
  const bool work = !s.loader;
  const int b = blockIdx.x;
  Save sv;
  sv.M = (long)B * T;
  const long M = sv.M;
  const long row0 = (long)b * T;

  if (tid == 0) {
    int n = 0;
    for (int l = NLAYER - 1; l >= 0; --l) {
      const float* const* pl = P + l * P_PER_LAYER;
      for (int rb = 0; rb < 2; ++rb) sched[n++] = Chunk{pl[PW_O2] + (long)rb * 64 * FF, FF, 64, FF};           // d g = d o2 . Wo2      (reduce n, out k < 16)
      for (int ob = 0; ob < 2; ++ob) sched[n++] = Chunk{pl[PW_I] + ob * 64, H, FF, 64};                         // d h1 += d it . Wi     (reduce n < 16)
      for (int ob = 0; ob < 2; ++ob)
        for (int rb = 0; rb < 2; ++rb) sched[n++] = Chunk{pl[PW_O] + (long)rb * 64 * H + ob * 64, H, 64, 64};  // d ctx = d o . Wo
      for (int ob = 0; ob < 2; ++ob)
        for (int rb = 0; rb < 6; ++rb) sched[n++] = Chunk{pl[2 * (rb >> 1)] + (long)(rb & 1) * 64 * H + ob * 64, H, 64, 64};  // d h += d(q|k|v) . [Wq; Wk; Wv]
    }
  }
  unsigned seed = 0u, ctr = 0u;
  if (p_drop > 0.f) {
    const unsigned* ru = reinterpret_cast<const unsigned*>(save + sv.rng());
    seed = ru[0];
    ctr = ru[1];
  }
  const Drop dr = make_drop(p_drop, seed, ctr);
  LnIn ln2, ln1;
  float pvr = 0.f;
  if (work) {
    ln_prefetch(s, ln2, save + sv.xs2(NLAYER - 1), save + sv.st2(NLAYER - 1), row0);
    if (tid < 2 * H) PV[tid] = (P + (NLAYER - 1) * P_PER_LAYER)[tid < H ? P_G1 : P_G2][tid & 127];
    for (int i = tid; i < T * LDI; i += NCT) DI[i] = 0.f;
    for (int i = tid; i < T * (H / 4); i += NCT) {
      const int t = i >> 5, c = (i & 31) * 4;
      *reinterpret_cast<f32x4*>(G + t * LDA + c) = kpf_ld4(dh + (row0 + t) * H + c);
    }
  }
  BAR();
  if (s.loader)
    for (int n = 0; n < NS - 1; ++n)
      if (!(s.dbg & 1)) issue_chunk<NS>(s, n);

  const int fr = s.lane & 15, fg = s.lane >> 4, tt = s.wave & 1;
  for (int l = NLAYER - 1; l >= 0; --l) {
    float* dyl = dys + (long)l * M * DY_LAYER;
    float* d_qkv = dyl;
    float* d_o1 = dyl + 384 * M;
    float* d_it = dyl + 512 * M;
    float* d_o2 = dyl + 528 * M;
    float* part1 = parts + (((long)l * 2 + 0) * B + b) * 2 * H;
    float* part2 = parts + (((long)l * 2 + 1) * B + b) * 2 * H;
    TRS_STAMP(32 + (NLAYER - 1 - l) * 8 + 0);
    // ---- requests whose data is used one or more phases later ----
    float rit[4];  // the pre-GELU intermediate of this lane's four rows (epilogue of the first product)
    if (work) {
      const float* si = save + sv.it(l);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tr = tt * 16 + 4 * fg + u;
        rit[u] = si[(row0 + (tr < T ? tr : T - 1)) * FF + fr];
      }
    }
    // ---- LayerNorm 2: G -> D1 (d h1, residual branch), D2 (d o2) ----
    ln_backward(s, work, G, ln2, PV + H, D1, D2, d_o2, part2, SP, row0, dr, call0 + 3 * l + 2);
    if (work) {
      ln_prefetch(s, ln1, save + sv.xs1(l), save + sv.st1(l), row0);
      dma_rows(s, save + sv.qkv(l) + row0 * 384, T, 384, 384, QKV, LDQ);  // (the q | k | v array is idle until the attention phase: every reader of the
    }                                                                      //  previous layer's last product is behind the barriers of the LayerNorm above)
    // ---- d it = (d o2 . Wo2) * gelu'(it) ----
    TRS_STAMP(32 + (NLAYER - 1 - l) * 8 + 1);
    gemm_op<NS, true, MMA>(s, D2, LDA, 1, 2, [&](int, int t, int col, float acc) {
      if (col < FF) {
        const float v = acc * gelu_grad(rit[t & 3]);  // (t = tt * 16 + 4 fg + r: r = t & 3)
        DI[t * LDI + col] = v;
        d_it[(row0 + t) * FF + col] = v;
      }
    });
    // ---- d h1 += d it . Wi ----
    gemm_op<NS, true, MMA>(s, DI, LDI, 2, 1, [&](int, int t, int col, float acc) { D1[t * LDA + col] += acc; });
    // ---- LayerNorm 1: D1 -> G (d h, residual branch), D2 (d o) ----
    TRS_STAMP(32 + (NLAYER - 1 - l) * 8 + 2);
    ln_backward(s, work, D1, ln1, PV, G, D2, d_o1, part1, SP, row0, dr, call0 + 3 * l + 1);
    BAR();  // (the reduce scratch in SP / SD has been read)
    if (work) dma_rows(s, save + sv.P(l) + (long)b * NH * T * T, 1, NH * T * T, NH * T * T, SP, SPAD);
    // ---- d ctx = d o . Wo -> D1 ----
    TRS_STAMP(32 + (NLAYER - 1 - l) * 8 + 3);
    gemm_op<NS, true, MMA>(s, D2, LDA, 2, 2, [&](int, int t, int col, float acc) { D1[t * LDA + col] = acc; });
    // ---- attention backward (in place on QKV) ----
    TRS_STAMP(32 + (NLAYER - 1 - l) * 8 + 4);
    attention_backward(s, work, QKV, SP, SD, D1, LDA, d_qkv, b, row0, dr, call0 + 3 * l);
    if (work) {
      if (l > 0) {  // the next (lower) layer's first LayerNorm backward: its operands fly under the twelve blocks below
        ln_prefetch(s, ln2, save + sv.xs2(l - 1), save + sv.st2(l - 1), row0);
        if (tid < 2 * H) pvr = (P + (l - 1) * P_PER_LAYER)[tid < H ? P_G1 : P_G2][tid & 127];
      }
    }
    // ---- d h += d(q | k | v) . [Wq; Wk; Wv] ----
    TRS_STAMP(32 + (NLAYER - 1 - l) * 8 + 5);
    gemm_op<NS, true, MMA>(s, QKV, LDQ, 2, 6, [&](int, int t, int col, float acc) { G[t * LDA + col] += acc; });
    TRS_STAMP(32 + (NLAYER - 1 - l) * 8 + 6);
    if (work && l > 0 && tid < 2 * H) PV[tid] = pvr;  // (this layer's two LayerNorms are done; the next ln_backward opens with a barrier)
  }
  BAR();
  if (work) {  // through the embedding dropout: d(e + pos)
    for (int i = tid; i < T * (H / 4); i += NCT) {
      const int t = i >> 5, c = (i & 31) * 4;
      const long off = (row0 + t) * H + c;
      const f32x4 g = *reinterpret_cast<const f32x4*>(G + t * LDA + c);
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = dr.keep(call0 + 12, (unsigned)(off + k)) ? g[k] * dr.ks : 0.f;
      kpf_st4(dE + off, v);
    }
  }
}

// ================================================ the decoder layer (cross attention) ================================================
// updatedDecoder layer 3 in train mode (model/transfusion_head.py:137-173, 437-554): q = (query + qpos) Wq^T, k | v = (key + kpos) Wkv^T, attention with dropout,
// x = LayerNorm(query + dropout(ctx Wo^T)), out = LayerNorm(x + dropout(W2 dropout(relu(W1 x)))), eps 1e-5 — one launch each way on the same engine as the stacks
// above (24 weight blocks per direction; the unfused form: ~14 launches forward, ~20 backward).  Parameter table (14 pointers): in_proj_weight [384][128],
// in_proj_bias, out_proj.weight / bias, norm2.weight / bias, linear1.weight / bias, linear2.weight / bias, norm3.weight / bias, qpos [21][128], kpos [21][128].
// Saved (floats, M = 21 B): qe | ke | qkv [M][384] | P 84 M | ctx | xs2 | st2 [M][4] | x | f1d | xs3 | st3 [M][4] | out | (seed, counter).
enum { XP_WIN = 0, XP_BIN, XP_WO, XP_BO, XP_G2, XP_B2, XP_W1, XP_BB1, XP_W2, XP_BB2, XP_G3, XP_B3, XP_QPOS, XP_KPOS, XP_N };
constexpr int XCHUNK = 24;
struct XSave {
  long M;
  __device__ __host__ long qe() const { return 0; }
  __device__ __host__ long ke() const { return 128 * M; }
  __device__ __host__ long qkv() const { return 256 * M; }
  __device__ __host__ long P() const { return 640 * M; }
  __device__ __host__ long ctx() const { return 724 * M; }
  __device__ __host__ long xs2() const { return 852 * M; }
  __device__ __host__ long st2() const { return 980 * M; }
  __device__ __host__ long x() const { return 984 * M; }
  __device__ __host__ long f1() const { return 1112 * M; }
  __device__ __host__ long xs3() const { return 1240 * M; }
  __device__ __host__ long st3() const { return 1368 * M; }
  __device__ __host__ long out() const { return 1372 * M; }
  __device__ __host__ long rng() const { return 1500 * M; }
  __device__ __host__ long total() const { return 1500 * M + 4; }
};
// backward's dY buffer (floats): dqkv [M][384] | do [M][128] | dpre [M][128] | df [M][128]
enum { XV_BQKV = 0, XV_BO = 384, XV_G2 = 512, XV_B2 = 640, XV_BB1 = 768, XV_BB2 = 896, XV_G3 = 1024, XV_B3 = 1152, XV_N = 1280 };

// LayerNorm over the 128 channels of the sample's 21 rows (wave per row): in (LDS, stride ldi) -> out_l (LDS, stride ldo; nullable) and out_g (HBM rows of 128),
// (mean, rstd) to st_g [row][4]
__device__ __forceinline__ void ln_forward(const Sync& s, bool work, const float* in, int ldi, const float* gw, const float* gb, float eps, float* out_l, int ldo,
                                           float* __restrict__ out_g, float* __restrict__ st_g, long row0) {
  BAR();
  if (work) {
    const float w0 = gw[s.lane], w1 = gw[64 + s.lane], b0 = gb[s.lane], b1 = gb[64 + s.lane];
    for (int t = s.wave; t < T; t += NCW) {
      const float a0 = in[t * ldi + s.lane], a1 = in[t * ldi + 64 + s.lane];
      const float mean = wave_sum(a0 + a1) * (1.0f / H);
      const float d0 = a0 - mean, d1 = a1 - mean;
      const float rstd = 1.0f / sqrtf(wave_sum(fmaf(d0, d0, d1 * d1)) * (1.0f / H) + eps);
      const float y0 = d0 * rstd * w0 + b0, y1 = d1 * rstd * w1 + b1;
      if (out_l) {
        out_l[t * ldo + s.lane] = y0;
        out_l[t * ldo + 64 + s.lane] = y1;
      }
      out_g[(row0 + t) * H + s.lane] = y0;
      out_g[(row0 + t) * H + 64 + s.lane] = y1;
      if (s.lane == 0) {
        st_g[(row0 + t) * 4] = mean;
        st_g[(row0 + t) * 4 + 1] = rstd;
      }
    }
  }
}

template <int MMA>
__global__ __launch_bounds__(NTHR) void xattn_train_fwd_kernel(const float* __restrict__ query, const float* __restrict__ key, const float* const* __restrict__ P,
                                                             float* __restrict__ save, int B, float p_drop, const long* __restrict__ rng, int call0) {
  constexpr int NS = 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  Chunk* sched = reinterpret_cast<Chunk*>(sm);
  float* PV = sm + SCHED_F;    // [XV_N] parameter vectors
  float* Q0 = PV + XV_N;       // [T][LDA] query (the residual of the first LayerNorm)
  float* X1 = Q0 + T * LDA;    // [T][LDA] query + qpos, later the feed-forward's hidden rows
  float* X2 = X1 + T * LDA;    // [T][LDA] key + kpos, later x = LayerNorm 2's output
  float* QKV = X2 + T * LDA;   // [T][LDQ] q | k | v; T1 (a LayerNorm's input) in the q columns, the context in the k columns
  float* T1 = QKV;
  float* CTX = QKV + H;
  float* S = QKV + T * LDQ;
  float* ring = S + SPAD;
  const int tid = threadIdx.x;
  Sync s;
  s.lane = tid & 63;
  s.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  s.loader = s.wave >= NCW;
  s.ring = ring;
  s.sched = sched;
  s.zero = kpf_trs_zero16;
  s.c = 0;
  s.ntotal = XCHUNK;
  s.dbg = 0;
  const bool work = !s.loader;
  const int b = blockIdx.x;
  XSave sv;
  sv.M = (long)B * T;
  const long row0 = (long)b * T;
  if (tid == 0) {
    int n = 0;
    const float* win = P[XP_WIN];
    for (int ob = 0; ob < 6; ++ob)  // q (2 blocks of 64 outputs), then k | v (4)
      for (int rb = 0; rb < 2; ++rb) sched[n++] = Chunk{win + (long)ob * 64 * H + rb * 64, H, 64, 64};
    const int ws[3] = {XP_WO, XP_W1, XP_W2};
    for (int w = 0; w < 3; ++w)
      for (int ob = 0; ob < 2; ++ob)
        for (int rb = 0; rb < 2; ++rb) sched[n++] = Chunk{P[ws[w]] + (long)ob * 64 * H + rb * 64, H, 64, 64};
  }
  const unsigned seed = (p_drop > 0.f && rng) ? (unsigned)rng[0] : 0u, ctr = (p_drop > 0.f && rng) ? (unsigned)rng[1] : 0u;
  const Drop dr = make_drop(p_drop, seed, ctr);
  if (b == 0 && tid == 0) {
    unsigned* ru = reinterpret_cast<unsigned*>(save + sv.rng());
    ru[0] = seed;
    ru[1] = ctr;
  }
  if (work) {
    for (int i = tid; i < XV_N; i += NCT) {  // in_proj_bias (384) | out_proj.bias | norm2.weight | norm2.bias | linear1.bias | linear2.bias | norm3.weight | norm3.bias
      const int which = i < XV_BO ? XP_BIN : (i < XV_BB1 ? XP_BO + ((i - XV_BO) >> 7) : (i < XV_BB2 ? XP_BB1 : (i < XV_G3 ? XP_BB2 : (i < XV_B3 ? XP_G3 : XP_B3))));
      PV[i] = P[which][i < XV_BO ? i : (i & 127)];
    }
    const float* qpos = P[XP_QPOS];
    const float* kpos = P[XP_KPOS];
    for (int i = tid; i < T * (H / 4); i += NCT) {
      const int t = i >> 5, c = (i & 31) * 4;
      const long off = (row0 + t) * H + c;
      const f32x4 qv = kpf_ld4(query + off), kv = kpf_ld4(key + off);
      const f32x4 qe = qv + kpf_ld4(qpos + t * H + c), ke = kv + kpf_ld4(kpos + t * H + c);
      *reinterpret_cast<f32x4*>(Q0 + t * LDA + c) = qv;
      *reinterpret_cast<f32x4*>(X1 + t * LDA + c) = qe;
      *reinterpret_cast<f32x4*>(X2 + t * LDA + c) = ke;
      kpf_st4(save + sv.qe() + off, qe);
      kpf_st4(save + sv.ke() + off, ke);
    }
  }
  BAR();
  if (s.loader)
    for (int n = 0; n < NS - 1; ++n) issue_chunk<NS>(s, n);
  float* sq = save + sv.qkv();
  gemm_op<NS, false, MMA>(s, X1, LDA, 2, 2, [&](int, int t, int col, float acc) {
    const float v = acc + PV[XV_BQKV + col];
    QKV[t * LDQ + col] = v;
    sq[(row0 + t) * 384 + col] = v;
  });
  gemm_op<NS, false, MMA>(s, X2, LDA, 4, 2, [&](int, int t, int col, float acc) {
    const float v = acc + PV[XV_BQKV + H + col];
    QKV[t * LDQ + H + col] = v;
    sq[(row0 + t) * 384 + H + col] = v;
  });
  attention_forward(s, work, QKV, S, CTX, LDQ, save + sv.P(), save + sv.ctx(), b, row0, dr, call0);
  {
    float* sx = save + sv.xs2();
    gemm_op<NS, false, MMA>(s, CTX, LDQ, 2, 2, [&](int, int t, int col, float acc) {
      const float o = acc + PV[XV_BO + col];
      const long idx = (row0 + t) * H + col;
      const float x = Q0[t * LDA + col] + (dr.keep(call0 + 1, (unsigned)idx) ? o * dr.ks : 0.f);
      T1[t * LDQ + col] = x;
      sx[idx] = x;
    });
  }
  ln_forward(s, work, T1, LDQ, PV + XV_G2, PV + XV_B2, 1e-5f, X2, LDA, save + sv.x(), save + sv.st2(), row0);
  {
    float* sf = save + sv.f1();
    gemm_op<NS, false, MMA>(s, X2, LDA, 2, 2, [&](int, int t, int col, float acc) {
      const long idx = (row0 + t) * H + col;
      const float r = fmaxf(acc + PV[XV_BB1 + col], 0.f);
      const float v = dr.keep(call0 + 2, (unsigned)idx) ? r * dr.ks : 0.f;
      X1[t * LDA + col] = v;
      sf[idx] = v;
    });
  }
  {
    float* sx = save + sv.xs3();
    gemm_op<NS, false, MMA>(s, X1, LDA, 2, 2, [&](int, int t, int col, float acc) {
      const float o = acc + PV[XV_BB2 + col];
      const long idx = (row0 + t) * H + col;
      const float x = X2[t * LDA + col] + (dr.keep(call0 + 3, (unsigned)idx) ? o * dr.ks : 0.f);
      T1[t * LDQ + col] = x;
      sx[idx] = x;
    });
  }
  ln_forward(s, work, T1, LDQ, PV + XV_G3, PV + XV_B3, 1e-5f, nullptr, 0, save + sv.out(), save + sv.st3(), row0);
}

template <int MMA>
__global__ __launch_bounds__(NTHR) void xattn_train_bwd_kernel(const float* __restrict__ dout, const float* const* __restrict__ P, const float* __restrict__ save,
                                                             float* __restrict__ dquery, float* __restrict__ dqe, float* __restrict__ dke, float* __restrict__ dys,
                                                             float* __restrict__ parts, int B, float p_drop, int call0) {
  constexpr int NS = 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  Chunk* sched = reinterpret_cast<Chunk*>(sm);
  float* PV = sm + SCHED_F;   // [2][128] LayerNorm weights (norm2, norm3)
  float* G = PV + 2 * H;      // [T][LDA] d out, then d pre (the feed-forward's first product), then d query
  float* D1 = G + T * LDA;
  float* D2 = D1 + T * LDA;
  float* QKV = D2 + T * LDA;  // [QPAD]
  float* SP = QKV + QPAD;
  float* SD = SP + SPAD;
  float* ring = SD + SPAD;
  const int tid = threadIdx.x;
  Sync s;
  s.lane = tid & 63;
  s.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  s.loader = s.wave >= NCW;
  s.ring = ring;
  s.sched = sched;
  s.zero = kpf_trs_zero16;
  s.c = 0;
  s.ntotal = XCHUNK;
  s.dbg = 0;
  const bool work = !s.loader;
  const int b = blockIdx.x;
  XSave sv;
  sv.M = (long)B * T;
  const long M = sv.M, row0 = (long)b * T;
  if (tid == 0) {
    int n = 0;
    const int ws[3] = {XP_W2, XP_W1, XP_WO};  // reduction over the rows (n) of each [N][K] tensor, 64 output columns (k) per block
    for (int w = 0; w < 3; ++w)
      for (int ob = 0; ob < 2; ++ob)
        for (int rb = 0; rb < 2; ++rb) sched[n++] = Chunk{P[ws[w]] + (long)rb * 64 * H + ob * 64, H, 64, 64};
    const float* win = P[XP_WIN];
    for (int ob = 0; ob < 2; ++ob)
      for (int rb = 0; rb < 2; ++rb) sched[n++] = Chunk{win + (long)rb * 64 * H + ob * 64, H, 64, 64};          // d qe = dq . Wq
    for (int ob = 0; ob < 2; ++ob)
      for (int rb = 0; rb < 4; ++rb) sched[n++] = Chunk{win + (long)(128 + rb * 64) * H + ob * 64, H, 64, 64};  // d ke = d(k | v) . [Wk; Wv]
  }
  unsigned seed = 0u, ctr = 0u;
  if (p_drop > 0.f) {
    const unsigned* ru = reinterpret_cast<const unsigned*>(save + sv.rng());
    seed = ru[0];
    ctr = ru[1];
  }
  const Drop dr = make_drop(p_drop, seed, ctr);
  float* d_qkv = dys;
  float* d_o = dys + 384 * M;
  float* d_pre = dys + 512 * M;
  float* d_f = dys + 640 * M;
  LnIn ln3, ln2;
  float rf1[8];  // the dropped ReLU output at this lane's accumulator positions of the feed-forward's first product (its sign is the mask)
  const int fr = s.lane & 15, fg = s.lane >> 4, tt = s.wave & 1, ct = (s.wave >> 1) & 3;
  if (work) {
    ln_prefetch(s, ln3, save + sv.xs3(), save + sv.st3(), row0);
    if (tid < 2 * H) PV[tid] = P[tid < H ? XP_G2 : XP_G3][tid & 127];
    for (int i = tid; i < T * (H / 4); i += NCT) {
      const int t = i >> 5, c = (i & 31) * 4;
      *reinterpret_cast<f32x4*>(G + t * LDA + c) = kpf_ld4(dout + (row0 + t) * H + c);
    }
    const float* sf = save + sv.f1();
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = tt * 16 + 4 * fg + r;
        rf1[ob * 4 + r] = sf[(row0 + (t < T ? t : T - 1)) * H + ob * 64 + ct * 16 + fr];
      }
  }
  BAR();
  if (s.loader)
    for (int n = 0; n < NS - 1; ++n) issue_chunk<NS>(s, n);
  // ---- LayerNorm 3: G -> D1 (d x, residual branch), D2 (d f) ----
  ln_backward(s, work, G, ln3, PV + H, D1, D2, d_f, parts + ((long)1 * B + b) * 2 * H, SP, row0, dr, call0 + 3);
  if (work) {
    ln_prefetch(s, ln2, save + sv.xs2(), save + sv.st2(), row0);
    dma_rows(s, save + sv.qkv() + row0 * 384, T, 384, 384, QKV, LDQ);
  }
  // ---- d pre = (d f . W2) * (hidden > 0) / (1 - p) -> G ----
  gemm_op<NS, true, MMA>(s, D2, LDA, 2, 2, [&](int ob, int t, int col, float acc) {
    const float v = rf1[ob * 4 + (t & 3)] > 0.f ? acc * dr.ks : 0.f;
    G[t * LDA + col] = v;
    d_pre[(row0 + t) * H + col] = v;
  });
  // ---- d x += d pre . W1 ----
  gemm_op<NS, true, MMA>(s, G, LDA, 2, 2, [&](int, int t, int col, float acc) { D1[t * LDA + col] += acc; });
  // ---- LayerNorm 2: D1 -> G (d query, residual branch), D2 (d o) ----
  ln_backward(s, work, D1, ln2, PV, G, D2, d_o, parts + ((long)0 * B + b) * 2 * H, SP, row0, dr, call0 + 1);
  BAR();
  if (work) dma_rows(s, save + sv.P() + (long)b * NH * T * T, 1, NH * T * T, NH * T * T, SP, SPAD);
  // ---- d ctx = d o . Wo -> D1 ----
  gemm_op<NS, true, MMA>(s, D2, LDA, 2, 2, [&](int, int t, int col, float acc) { D1[t * LDA + col] = acc; });
  attention_backward(s, work, QKV, SP, SD, D1, LDA, d_qkv, b, row0, dr, call0);
  // ---- d qe = dq . Wq (also the query's gradient through the projection); d ke = d(k | v) . [Wk; Wv] ----
  gemm_op<NS, true, MMA>(s, QKV, LDQ, 2, 2, [&](int, int t, int col, float acc) {
    dqe[(row0 + t) * H + col] = acc;
    dquery[(row0 + t) * H + col] = G[t * LDA + col] + acc;
  });
  gemm_op<NS, true, MMA>(s, QKV + H, LDQ, 2, 4, [&](int, int t, int col, float acc) { dke[(row0 + t) * H + col] = acc; });
}

constexpr size_t XF_LDS = (size_t)(SCHED_F + XV_N + 3 * T * LDA + T * LDQ + SPAD + 4 * SLOT) * sizeof(float);
constexpr size_t XB_LDS = (size_t)(SCHED_F + 2 * H + 3 * T * LDA + QPAD + 2 * SPAD + 4 * SLOT) * sizeof(float);
static_assert(XF_LDS <= 160 * 1024 && XB_LDS <= 160 * 1024, "the decoder-layer kernels' LDS must fit one CU");

constexpr size_t FWD_LDS = (size_t)(SCHED_F + PV_PAD + T * LDA + T * LDQ + SPAD + NSF * SLOT) * sizeof(float);
constexpr size_t BWD_LDS = (size_t)(SCHED_F + 2 * H + 3 * T * LDA + T * LDI + QPAD + 2 * SPAD + NSB * SLOT) * sizeof(float);
static_assert(FWD_LDS <= 160 * 1024 && BWD_LDS <= 160 * 1024, "the stack kernels' LDS must fit one CU");
static_assert(sizeof(Chunk) == 24, "Chunk layout");
static_assert(2 * SPAD >= NCW * 2 * H, "LayerNorm reduce scratch lives in the score arrays");

}  // namespace

/* tuning aid: 64 x 8-byte stamp slots in device memory (NULL switches the stamps off) */
extern "C" int kpf_tr_stack_set_stamps(void* p) {
  unsigned long long* q = static_cast<unsigned long long*>(p);
  static const int dbg = []() { const char* e = getenv("KPF_TRS_DBG"); return e ? atoi(e) : 0; }();  // (ablation bits, read once: see kpf_trs_dbg)
  if (hipMemcpyToSymbol(HIP_SYMBOL(kpf_trs_dbg), &dbg, sizeof(dbg)) != hipSuccess) return KPF_ELAUNCH;
  return hipMemcpyToSymbol(HIP_SYMBOL(kpf_trs_stamps), &q, sizeof(q)) == hipSuccess ? KPF_OK : KPF_ELAUNCH;
}
extern "C" long kpf_tr_stack_save_floats(int B) {
  Save sv;
  sv.M = (long)B * T;
  return sv.total();
}
extern "C" long kpf_tr_stack_out_offset(int B) {
  Save sv;
  sv.M = (long)B * T;
  return sv.Hs(NLAYER);
}
extern "C" long kpf_tr_stack_dy_floats(int B) { return (long)NLAYER * B * T * DY_LAYER; }
extern "C" long kpf_tr_stack_part_floats(int B) { return (long)NLAYER * 2 * B * 2 * H; }

/* where the backward's operands of the deferred weight gradients live: which = 0 q|k|v input (H[l]), 1 ctx, 2 h1, 3 g (offsets into `save`);
 * 4 dqkv, 5 do1, 6 dit, 7 do2 (offsets into `dys`). */
extern "C" long kpf_tr_stack_offset(int B, int layer, int which) {
  Save sv;
  sv.M = (long)B * T;
  const long M = sv.M;
  switch (which) {
    case 0: return sv.Hs(layer);
    case 1: return sv.ctx(layer);
    case 2: return sv.h1(layer);
    case 3: return sv.g(layer);
    case 4: return (long)layer * M * DY_LAYER;
    case 5: return (long)layer * M * DY_LAYER + 384 * M;
    case 6: return (long)layer * M * DY_LAYER + 512 * M;
    case 7: return (long)layer * M * DY_LAYER + 528 * M;
    default: return -1;
  }
}

extern "C" int kpf_tr_stack_train_forward(const float* e, const float* pos, const void* param_table, float* save, long save_floats, int B, float p_drop,
                                          const long* rng, int call0, int mma, void* stream) {
  KPF_REQUIRE(e && pos && param_table && save && B > 0 && mma >= 0 && mma <= 2, "kpf_tr_stack_train_forward: bad arguments");
  KPF_REQUIRE(save_floats >= kpf_tr_stack_save_floats(B), "kpf_tr_stack_train_forward: save buffer too small");
  KPF_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng), "kpf_tr_stack_train_forward: dropout needs 0 <= p < 1 and the rng state");
  KPF_REQUIRE(kpf_aligned16(e) && kpf_aligned16(pos) && kpf_aligned16(save), "kpf_tr_stack_train_forward: e, pos, save must be 16-byte aligned");
  using K = void (*)(const float*, const float*, const float* const*, float*, int, float, const long*, int);
  const K kern = mma == 0 ? (K)tr_stack_fwd_kernel<0> : (mma == 1 ? (K)tr_stack_fwd_kernel<1> : (K)tr_stack_fwd_kernel<2>);
  static std::atomic<bool> lds_opt_in[3][KPF_MAX_DEVICES];
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(kern), lds_opt_in[mma])) {
    kpf_set_error("kpf_tr_stack_train_forward: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(NTHR), FWD_LDS, reinterpret_cast<hipStream_t>(stream), e, pos, static_cast<const float* const*>(param_table), save, B, p_drop,
                     rng, call0);
  return kpf_check_launch("kpf_tr_stack_train_forward");
}

extern "C" int kpf_tr_stack_train_backward(const float* dh, const void* param_table, const float* save, float* dE, float* dys, float* parts, int B, float p_drop,
                                           int call0, int mma, void* stream) {
  KPF_REQUIRE(dh && param_table && save && dE && dys && parts && B > 0 && mma >= 0 && mma <= 2, "kpf_tr_stack_train_backward: bad arguments");
  KPF_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "kpf_tr_stack_train_backward: 0 <= p < 1");
  KPF_REQUIRE(kpf_aligned16(dh) && kpf_aligned16(save) && kpf_aligned16(dE) && kpf_aligned16(dys), "kpf_tr_stack_train_backward: dh, save, dE, dys must be 16-byte aligned");
  using K = void (*)(const float*, const float* const*, const float*, float*, float*, float*, int, float, int);
  const K kern = mma == 0 ? (K)tr_stack_bwd_kernel<0> : (mma == 1 ? (K)tr_stack_bwd_kernel<1> : (K)tr_stack_bwd_kernel<2>);
  static std::atomic<bool> lds_opt_in[3][KPF_MAX_DEVICES];
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(kern), lds_opt_in[mma])) {
    kpf_set_error("kpf_tr_stack_train_backward: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(NTHR), BWD_LDS, reinterpret_cast<hipStream_t>(stream), dh, static_cast<const float* const*>(param_table), save, dE, dys, parts,
                     B, p_drop, call0);
  return kpf_check_launch("kpf_tr_stack_train_backward");
}

extern "C" long kpf_xattn_train_save_floats(int B) {
  XSave sv;
  sv.M = (long)B * T;
  return sv.total();
}
/* which: 0 qe, 1 ke, 2 ctx, 3 x (LayerNorm 2's output), 4 f1 (dropped ReLU rows), 5 out — offsets into `save`; 6 dqkv [M][384], 7 do, 8 dpre, 9 df — offsets into `dys` */
extern "C" long kpf_xattn_train_offset(int B, int which) {
  XSave sv;
  sv.M = (long)B * T;
  const long M = sv.M;
  switch (which) {
    case 0: return sv.qe();
    case 1: return sv.ke();
    case 2: return sv.ctx();
    case 3: return sv.x();
    case 4: return sv.f1();
    case 5: return sv.out();
    case 6: return 0;
    case 7: return 384 * M;
    case 8: return 512 * M;
    case 9: return 640 * M;
    default: return -1;
  }
}
extern "C" long kpf_xattn_train_dy_floats(int B) { return (long)B * T * 768; }

extern "C" int kpf_xattn_train_forward(const float* query, const float* key, const void* param_table, float* save, long save_floats, int B, float p_drop,
                                       const long* rng, int call0, int mma, void* stream) {
  KPF_REQUIRE(query && key && param_table && save && B > 0 && mma >= 0 && mma <= 2, "kpf_xattn_train_forward: bad arguments");
  KPF_REQUIRE(save_floats >= kpf_xattn_train_save_floats(B), "kpf_xattn_train_forward: save buffer too small");
  KPF_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng), "kpf_xattn_train_forward: dropout needs 0 <= p < 1 and the rng state");
  KPF_REQUIRE(kpf_aligned16(query) && kpf_aligned16(key) && kpf_aligned16(save), "kpf_xattn_train_forward: query, key, save must be 16-byte aligned");
  using K = void (*)(const float*, const float*, const float* const*, float*, int, float, const long*, int);
  const K kern = mma == 0 ? (K)xattn_train_fwd_kernel<0> : (mma == 1 ? (K)xattn_train_fwd_kernel<1> : (K)xattn_train_fwd_kernel<2>);
  static std::atomic<bool> lds_opt_in[3][KPF_MAX_DEVICES];
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(kern), lds_opt_in[mma])) {
    kpf_set_error("kpf_xattn_train_forward: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(NTHR), XF_LDS, reinterpret_cast<hipStream_t>(stream), query, key, static_cast<const float* const*>(param_table), save, B,
                     p_drop, rng, call0);
  return kpf_check_launch("kpf_xattn_train_forward");
}

extern "C" int kpf_xattn_train_backward(const float* dout, const void* param_table, const float* save, float* dquery, float* dqe, float* dke, float* dys,
                                        float* parts, int B, float p_drop, int call0, int mma, void* stream) {
  KPF_REQUIRE(dout && param_table && save && dquery && dqe && dke && dys && parts && B > 0 && mma >= 0 && mma <= 2, "kpf_xattn_train_backward: bad arguments");
  KPF_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "kpf_xattn_train_backward: 0 <= p < 1");
  KPF_REQUIRE(kpf_aligned16(dout) && kpf_aligned16(save) && kpf_aligned16(dys), "kpf_xattn_train_backward: dout, save, dys must be 16-byte aligned");
  using K = void (*)(const float*, const float* const*, const float*, float*, float*, float*, float*, float*, int, float, int);
  const K kern = mma == 0 ? (K)xattn_train_bwd_kernel<0> : (mma == 1 ? (K)xattn_train_bwd_kernel<1> : (K)xattn_train_bwd_kernel<2>);
  static std::atomic<bool> lds_opt_in[3][KPF_MAX_DEVICES];
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(kern), lds_opt_in[mma])) {
    kpf_set_error("kpf_xattn_train_backward: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(NTHR), XB_LDS, reinterpret_cast<hipStream_t>(stream), dout, static_cast<const float* const*>(param_table), save, dquery, dqe,
                     dke, dys, parts, B, p_drop, call0);
  return kpf_check_launch("kpf_xattn_train_backward");
}
