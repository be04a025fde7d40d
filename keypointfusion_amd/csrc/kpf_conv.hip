// Implicit-GEMM convolution / linear layer on the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32).
//
// GEMM view:  out[m][n] = sum_k A[m][k] * W[n][k],  m = output pixel (b,oy,ox), n = output channel,
//             k = (ky,kx,c) with the input channel c fastest (activations are NHWC, so k-runs are contiguous).
//
// MI355X mapping
//  * One workgroup = 4 (or 8) waves computes a BM x BN tile, BM = 16*TM*WM pixels, BN = 16*TN*WN channels; each wave owns
//    TM x TN accumulator tiles of 16x16 (TM*TN*4 registers).  f32-input MFMA runs at the f32 vector rate (157 TF peak, 1/16
//    of bf16), so the kernel is MFMA-issue bound by construction; everything else is arranged to stay out of its way.
//  * Staging is LDS-DMA (global_load_lds_dwordx4): every wave moves 8 rows x 128 B per instruction straight into LDS, no VGPR
//    round trip and no ds_write.  Two LDS buffers: the DMA of K-tile k+1 is issued before the MFMAs of tile k and lands
//    under them; one barrier per K tile.  A 128x128 tile takes 64 KB, so two workgroups share a CU: one's DMA issue (the
//    CU's address unit moves 64 B/clk, ~500 cycles per 32 KB tile), pipeline fill and epilogue run under the other's MFMAs.
//    Measured alternatives (round 1): register staging + 2 barriers 84 TF on the model's shapes; a dedicated loader wave
//    with a 3-deep ring was latency-bound on its single instruction stream (MFMA waves waited ~1000 of 5400 cycles/tile).
//  * The MFMA is issued "transposed": A-operand = weight fragment W[n][k], B-operand = activation fragment X[m][k], so
//    a lane's 4 accumulator registers are 4 *consecutive output channels* of one pixel -> the epilogue reads bias /
//    gamma / residual and writes the result as float4 (16 B per lane, 64 B contiguous per pixel per instruction).
//  * LDS images are [row][32 k] fp32 with the 16-byte chunk index XOR-swizzled by ((row >> 1) & 7) (applied on the DMA's source
//    address): rows are 128 B, so even / odd rows start in different halves of the 256-byte bank row and the 16 rows of a fragment
//    read land in 16 distinct 16-byte slots (a swizzle by row & 7 repeats every 8 rows: 2-way conflicts); ds_read_b128 fragment
//    reads are bank-conflict free and one ds_read_b128 feeds 4 MFMAs (the K order inside
//    a 16-deep step is permuted identically for both operands, which a dot product allows).
//  * Workgroup ids are remapped so that the 8 XCDs (private L2s) each get a contiguous range of tiles, channel tiles
//    fastest: the tiles that re-read one activation panel run on one XCD back to back.
//  * Prologue: eval-BatchNorm + ReLU of the *input* (pre-activation Residual, model/hourglass.py:106-108) is applied to
//    the activation fragments after the LDS read (scale/shift table in LDS).  Epilogues are compiled per kind (linear/ReLU,
//    GELU, residual) with an unguarded float4 fast path for interior tiles.
#include "kpf_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct ConvArgs {
  const float* in;
  const float* w;
  const float* bias;
  const float* ps;
  const float* pt;
  const float* gamma;
  const float* res;
  float* out;
  int M, N, K, Kp;
  int IH, IW, Cin, in_ld, in_coff;
  int OH, OW, ohow, KH, KW, sh, sw, ph, pw;
  int out_ld, out_coff, res_ld, res_coff;
  const float* zero;  // 16 zero bytes in device memory: DMA source of padding / out-of-range chunks
  float w_unscale;    // split path: weights were packed as w * 2^s, the accumulator is multiplied by 2^-s
  unsigned flags;
  int tilesN, nblk;
  int vec;  // 1: output/residual rows are 16-byte aligned -> float4 epilogue
  int dbg;  // tuning aid (KPF_G8_DBG, gemm16_8ph_kernel only): 1 = no activation, 2 = no global stores, 4 = no main loop
  int st_policy;  // output stores of the store-only 16-bit GEMM epilogues: 0 plain, 1 `sc1` (write-through: the line is not kept in the XCD's L2), 2 `nt` (KPF_G8_ST)
  int skew;  // opt-in start skew of the odd workgroup slot of a CU, in units of ~2048 clocks (KPF_STAGGER, fp32 NS = 2 launches; 0 = off, the default)
  // grouped launch (kpf_conv_desc::groups > 1, grid.y = group): group g consumes input channels from in_coff + g * g_in (staging units), uses the weights at
  // w + g * g_w (4-byte words) and bias[g * g_out + n], and writes output channels from out_coff + g * g_out (g_out = N).  All zero for an ordinary launch.
  int groups, g_in, g_out;
  long g_w;
};

// The group's view of the launch arguments (grid.y = group; blockIdx.y is 0 for ordinary launches, whose steps are 0 too): scalar arithmetic only.
__device__ __forceinline__ ConvArgs kpf_group_args(const ConvArgs& a0) {
  ConvArgs a = a0;
  const int g = blockIdx.y;
  a.in_coff += g * a.g_in;
  a.out_coff += g * a.g_out;
  a.res_coff += g * a.g_out;  // (a residual, when present, is stacked like the output)
  a.w += (long)g * a.g_w;
  if (a.bias) a.bias += g * a.g_out;
  return a;
}

// GELU(x) = x/2 * (1 + erf(x/sqrt2)) with erfc(z) = t*(a1 + t*(a2 + ...)) * exp(-z^2), t = 1/(1 + p z)  (Abramowitz-Stegun 7.1.26,
// |error| <= 1.5e-7 absolute on erf, i.e. fp32 rounding level) written on the erfc side so the negative tail does not
// cancel: gelu = max(x,0) - |x/2 * erfc(|x|/sqrt2)|.  12 VALU ops (2 transcendental) instead of libm erff's ~50 with branches:
// the GELU epilogue of a K=96 GEMM would otherwise cost as much as its MFMAs.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float q = 0.5f * x * (p * t) * e;  // x/2 * erfc(|x|/sqrt2), carries the sign of x
  return fmaxf(x, 0.f) - fabsf(q);
}

// GELU for 16-bit outputs: kpf_gelu_h16 (kpf_common.h)
__device__ __forceinline__ float gelu_h16(float x) { return kpf_gelu_h16(x); }
enum { ARITH_F32 = 0, ARITH_SPLIT = 1, ARITH_SPLIT_W = 2, ARITH_BF16 = 3, ARITH_F16 = 4, ARITH_R_BF16 = 5, ARITH_R_F16 = 6 };
// ARITH_R_* (round 6): fp32 STORAGE on both sides, the operands rounded to bf16 / f16 in registers and multiplied on the 16-bit MFMA with fp32 accumulation — what
// torch.autocast does to a Linear whose tensors stay fp32 around it (the fusion head's DESA Linears in the mixed-precision training step: KPF_MMA_BF16 / _F16).
// Separate instantiations: the fp32 kernels' code is untouched.
// the GELU every epilogue of one arithmetic uses (all tile shapes of a storage type must agree bit for bit: a sample's result must not depend on
// which kernel its batch size selects)
template <int ARITH>
__device__ __forceinline__ float gelu_of(float x) {
  if constexpr (ARITH == ARITH_BF16 || ARITH == ARITH_F16) return gelu_h16(x);
  else return gelu_erf(x);
}

enum { EPI_LIN = 0, EPI_GELU = 1, EPI_RES = 2, EPI_GGRAD = 3, EPI_GELU2 = 4 };  // EPI_GELU2 (KPF_ACT_GELU_SAVE): out = gelu(z) AND z = acc + bias to a second buffer
// (passed in the `res` slot): the forward of a training Linear whose GELU's backward needs the pre-activation — one launch instead of GEMM + GELU pass.
//  // EPI_GGRAD (KPF_RES_GELU_GRAD): y = (acc + bias) * gelu'(res) — the data gradient of
// `Linear(gelu(z))` with respect to z in the GEMM's epilogue (training step; res = z).  A separate epilogue value: the residual epilogues of the inference
// kernels keep their code and registers.
__device__ __forceinline__ float gelu_exact_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }  // (csrc/kpf_train.hip gelu_exact)
__device__ __forceinline__ float gelu_grad_erf(float v) {  // d/dv [v Phi(v)] = Phi(v) + v phi(v)  (csrc/kpf_train.hip gelu_grad: same expression)
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
  return cdf + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}
#ifndef STORE4
#define STORE4(p, v) *reinterpret_cast<f32x4*>(p) = (v)
#endif

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;

__device__ __attribute__((aligned(16))) float kpf_zero16[4] = {0.f, 0.f, 0.f, 0.f};  // source of every padding / out-of-range chunk

#ifdef KPF_DBG_TIME  // tuning build (make dbg): per-workgroup stamps read back by tools/f32_tile_time.py
// [0..3] shader-clock stamps (start, main loop start, main loop end, end), [4],[5] 100-MHz real-time stamps at start / end, [6] HW_ID
__device__ unsigned long long kpf_dbg_t[8 * 8192];
#define KPF_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) { kpf_dbg_t[8 * blockIdx.x + (i)] = __builtin_readcyclecounter(); \
    if ((i) == 0) { kpf_dbg_t[8 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime(); kpf_dbg_t[8 * blockIdx.x + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32); } \
    if ((i) == 3) kpf_dbg_t[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define KPF_STAMP(i)
#endif
constexpr int BK = 32;  // K-tile depth: 8 chunks of 16 B per staged row
// Two LDS buffers per workgroup (64 KB at 128x128): two workgroups share a CU, so one's DMA issue, pipeline fill and epilogue run
// under the other's MFMAs.

// SPLIT = true: both operands are in the split format of kpf_common.h ([32 x f16 hi | 32 x f16 lo] per 32-k block, byte-identical
// staging) and a 32-deep K tile is three v_mfma_f32_16x16x32_f16 per fragment pair (hi*hi + hi*lo + lo*hi, fp32 accumulate): the
// dropped lo*lo term and the 22-bit operands leave a per-product error of ~2^-21, below the rounding noise of an fp32 accumulation
// chain, at 3/16 of the f32-input MFMA's cycles.
// ARITH = 2: weights split as above, activations staged as fp32 and split in registers after the LDS read (after the BatchNorm+ReLU
// operand prologue when there is one): any fp32 tensor can feed the f16 matrix cores without a format pass over HBM.
// ARITH = 3 / 4: 16-bit storage (bf16 / f16) of both operands, of the residual and of the output (include/kpf.h, *_h16): rows hold 64
// elements per 128-byte K tile, so the staging is byte-identical again; the two 32-deep k-steps of a tile are one
// v_mfma_f32_16x16x32_{bf16,f16} each.  All staging-side fields of ConvArgs (Cin, in_ld, in_coff, K, Kp) then count 4-byte words
// (= 2 elements), all output-side fields elements.
// NS = LDS stages.  NS = 2: the DMA of tile k+1 flies under the MFMAs of tile k, one __syncthreads per K tile (its fence drains the
// DMA).  NS > 2 (split arithmetic, where a K tile's MFMAs are shorter than the DMA latency): a ring with NS-1 tiles in flight — a
// counted s_waitcnt vmcnt leaves the younger tiles' DMAs outstanding across a raw s_barrier; every wave issues the same number of DMA
// instructions per tile (the staged row count is rounded up to whole passes) so that one immediate count fits all waves.
template <int TM, int TN, int WM, int WN, bool IS1X1, bool HAS_PRO, int EPI, int ARITH, int NS>
__device__ __forceinline__ void igemm_body(const ConvArgs& a) {
  constexpr bool SPLIT = ARITH == ARITH_SPLIT || ARITH == ARITH_SPLIT_W;
  constexpr bool H16 = ARITH == ARITH_BF16 || ARITH == ARITH_F16;
  using TH = typename std::conditional<ARITH == ARITH_BF16, bf16_t, f16_t>::type;  // 16-bit storage type (H16 only)
  constexpr int BM = 16 * TM * WM;
  constexpr int BN = 16 * TN * WN;
  constexpr int NW = WM * WN;               // waves per workgroup (4 or 8)
  constexpr int RPP = 8 * NW;               // rows staged per pass: every wave moves 8 rows x 128 B = one 1-KiB DMA
  constexpr int AP = (BM + RPP - 1) / RPP;  // A staging passes
  constexpr int BP = (BN + RPP - 1) / RPP;  // B staging passes
  constexpr bool R16 = ARITH == ARITH_R_BF16 || ARITH == ARITH_R_F16;
  constexpr bool WHOLE = NS > 2 || ARITH == ARITH_F32 || R16;  // every wave issues every staging pass (no exec-masked DMA branches)
  constexpr int BMR = WHOLE ? AP * RPP : BM;  // staged rows (whole passes; the extra rows are never read)
  constexpr int BNR = WHOLE ? BP * RPP : BN;
  constexpr int TILE = (BMR + BNR) * BK;
  constexpr int CNT = AP + BP;  // DMA instructions per wave per K tile in ring mode
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  static_assert(NS >= 1 && (NS < 2 || (NS - 2) * CNT < 64), "vmcnt is a 6-bit counter");
  static_assert(BM % 8 == 0 && BN % 8 == 0, "tile granularity");

  extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][TILE] (+ [2][Kp] operand prologue scale/shift)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA destinations (M0) become scalar arithmetic
  const int wm = wave % WM, wn = wave / WM;
  KPF_STAMP(0);
  if constexpr (ARITH == ARITH_F32 && NS == 2) {
    // Start skew between the two workgroups that share a CU (a.skew, 0 = off = the default: KPF_STAGGER opts in): launched together and equally long, they run
    // their main loops together (each at half the matrix pipe's rate) and their epilogues together (the pipe idle).  The workgroup in the odd
    // slot of a CU sleeps before its first tile; equal tile times keep the offset through the following rounds, and one workgroup's
    // prologue / epilogue then lies under the other's MFMAs.  Measured +0.1-0.3 % on the headline — inside the noise — and TG_ID parity does not
    // identify "the second workgroup of THIS launch" when another stream's kernel shares the CU, so it is off unless asked for.
    if (a.skew > 0 && (int)blockIdx.x < 512 && blockIdx.y == 0) {
      const unsigned tg = __builtin_amdgcn_s_getreg(4 | (16 << 6) | (3 << 11));  // HW_ID.TG_ID: this workgroup's slot in its CU
      if (tg & 1)
        for (int i = 0; i < a.skew; ++i) __builtin_amdgcn_s_sleep(32);  // ~2048 clocks
    }
  }

  // XCD-aware bijective remap: blocks b, b+8, b+16.. share an XCD -> give each XCD a contiguous range of logical tiles.
  int bid = blockIdx.x;
  {
    const int nb = a.nblk, q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int nt = bid % a.tilesN, mt = bid / a.tilesN;
  const int m0 = mt * BM, n0 = nt * BN;

  // Staging: global -> LDS by DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write).  LDS rows are [32 k] fp32 = 8
  // chunks of 16 B, chunk c of row r stored at position c ^ ((r >> 1) & 7) (bank-conflict-free ds_read_b128 fragments).  The DMA
  // writes lane l of a wave at base + 16*l, so LDS stays linear and the swizzle is applied to the SOURCE: the lane at position
  // (r, c') fetches logical chunk c' ^ ((r >> 1) & 7).  Out-of-range rows / K padding / conv halo read a 16-byte zero page instead.
  const int lr = tid >> 3;                          // row within a staging pass; a wave owns rows 8*wave .. 8*wave+7
  const int kc = (((tid & 7) ^ ((lr >> 1) & 7)) << 2);  // logical k offset (floats) this lane fetches within the K tile

  // IS1X1 (dense 1x1, stride 1, K % 32 == 0): one source pointer per staging pass, fixed for the whole K loop and advanced by the
  // K-tile offset — the per-tile address arithmetic is two VALU ops per DMA.  Rows beyond M / channels beyond N are clamped to
  // the last valid row (their products land in accumulator rows/columns the epilogue never stores), so no zero page is needed.
  // General convolution: receptive-field origin per staged row, tap decoded per K tile, zero page outside the image / K range.
  // General convolution: per staged row the address of its receptive field's origin and the origin's image coordinates; per lane the tap
  // (ky, kx) and channel offset of ITS 16-byte chunk of the current K tile, advanced incrementally from K tile to K tile — BK = q * Cin + r
  // words per step with q, r uniform: c += r, one conditional wrap, tap += q + wrap; ky = tap / KW through a 16-bit reciprocal (tap <= 49) —
  // so that a K tile costs ~10 VALU operations of decode instead of two integer divisions (round 5: at 16-bit storage a 128 x 128 K tile is 512
  // matrix-pipe cycles per wave, and the ~130 VALU instructions of the division form were as long as the MFMAs they feed).  Zero page outside
  // the image / K range.  Any Cin % 4 == 0 and any KH x KW: a lane's chunk never straddles a tap (taps start at multiples of Cin, chunks at
  // multiples of 4 words), different lanes of a K tile may sit in different taps.
  const float* pa[AP];
  const float* pb[BP];
  int riy[AP], rix[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = m0 + lr + RPP * p;
    if (IS1X1) {
      const int mc = m < a.M ? m : a.M - 1;
      pa[p] = a.in + (long)mc * a.in_ld + a.in_coff + kc;
      riy[p] = rix[p] = 0;
    } else if (m < a.M && lr + RPP * p < BM) {  // (ring mode stages whole passes: rows beyond BM read the zero page)
      const int b = m / a.ohow;
      const int r = m - b * a.ohow;
      const int oy = r / a.OW;
      const int ox = r - oy * a.OW;
      riy[p] = oy * a.sh - a.ph;
      rix[p] = ox * a.sw - a.pw;
      pa[p] = a.in + ((long)b * a.IH * a.IW + (long)riy[p] * a.IW + rix[p]) * a.in_ld + a.in_coff;
    } else {
      pa[p] = a.zero;
      riy[p] = -(1 << 20);  // (never inside the image)
      rix[p] = 0;
    }
  }
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int n = n0 + lr + RPP * p;
    const int nc = n < a.N ? n : a.N - 1;
    pb[p] = a.w + (long)nc * a.Kp + kc;
  }
  // this lane's chunk of the K tile that was staged last (general convolution only)
  int s_kt = 0, s_tap = 0, s_c = 0;
  const int s_q = IS1X1 ? 0 : BK / a.Cin, s_r = IS1X1 ? 0 : BK - s_q * a.Cin;   // uniform
  const int s_rcp = IS1X1 ? 0 : (65536 + a.KW - 1) / a.KW;                     // uniform: tap / KW == (tap * s_rcp) >> 16 for tap < 2^16 / KW
  if (!IS1X1) {
    s_tap = kc / a.Cin;
    s_c = kc - s_tap * a.Cin;
  }

  auto stage = [&](int kt, int buf) {
    float* As = lds + buf * TILE;
    float* Bs = As + BMR * BK;
    if (IS1X1) {
#pragma unroll
      for (int p = 0; p < AP; ++p)
        if (WHOLE || RPP * p + 8 * wave < BM)  // wave-uniform
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(pa[p] + kt * BK), (lds_void_t*)(As + (RPP * p + 8 * wave) * BK), 16, 0, 0);
    } else {
      // K tiles are staged in order (every caller passes kt == the previous kt, or the previous kt + 1; the first call is kt = 0)
      const bool adv = kt != s_kt;
      s_kt = kt;
      s_c += adv ? s_r : 0;
      const bool wrap = s_c >= a.Cin;
      s_c -= wrap ? a.Cin : 0;
      s_tap += adv ? s_q + (wrap ? 1 : 0) : 0;
      const int ky = (s_tap * s_rcp) >> 16;
      const int kx = s_tap - ky * a.KW;
      const bool kvalid = s_tap < a.KH * a.KW;  // (K padding: chunks beyond the last tap)
      const int toff = (ky * a.IW + kx) * a.in_ld + s_c;
#pragma unroll
      for (int p = 0; p < AP; ++p) {
        if (WHOLE || RPP * p + 8 * wave < BM) {  // wave-uniform
          const int iy = riy[p] + ky, ix = rix[p] + kx;
          const bool v = kvalid && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
          const float* src = v ? pa[p] + toff : a.zero;
          __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(As + (RPP * p + 8 * wave) * BK), 16, 0, 0);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < BP; ++p)
      if (WHOLE || RPP * p + 8 * wave < BN)  // wave-uniform: BN is a multiple of 8
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(pb[p] + kt * BK), (lds_void_t*)(Bs + (RPP * p + 8 * wave) * BK), 16, 0, 0);
  };

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = zero4;

  const int nk = a.Kp / BK;
  const int fr = lane & 15;  // fragment row (pixel for X, channel for W)
  const int fg = lane >> 4;  // k group 0..3
  const int rsw = (fr >> 1) & 7;  // this lane's row swizzle

  constexpr int EPW = H16 ? 2 : 1;  // elements per staged 4-byte word
  float* pro_s = lds + NS * TILE;  // [Kp] scale, then [Kp] shift (zero beyond Cin: padded k contributes relu(0*x+0) = 0)
  float* pro_t = pro_s + a.Kp * EPW;
  if (NS <= 2) {
    stage(0, 0);
  } else {
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
      if (t < nk) stage(t, t);
  }
  if (HAS_PRO) {
    for (int i = tid; i < a.Kp * EPW; i += 64 * NW) {
      pro_s[i] = i < a.Cin * EPW ? a.ps[i] : 0.f;
      pro_t[i] = i < a.Cin * EPW ? a.pt[i] : 0.f;
    }
  }
  if (NS <= 2) __syncthreads();  // (a pending global_load_lds is an outstanding vmcnt: the barrier's fence drains it)

  // Ring mode with pre-split operands: software-pipelined main loop.  The fragments of tile t+1 are read from LDS into a second
  // register set while the MFMAs of tile t run from the first, so a wave's LDS latency sits under its own MFMAs (the plain loop
  // relies on the SIMD's other wave for that); DMA runs NS-1 tiles ahead.
  KPF_STAMP(1);
  if constexpr (ARITH == ARITH_F32 && NS == 2) {
    // f32-input MFMA main loop, software-pipelined so that the matrix pipe never waits for anything but the barrier's skew.  An
    // f32 MFMA occupies the pipe for 32 cycles and the wave's issue port for a few, so everything else a K tile needs — 16 fragment
    // reads, the next tile's DMA issue (8-16 instructions with their address arithmetic), the barrier — is placed BETWEEN MFMAs of the
    // same wave instead of in front of them (the plain loop exposed ~500 cycles of DMA issue + ~150 of LDS latency per 4096-cycle K
    // tile whenever the CU's two workgroups ran in lockstep).  Per K tile kt (two 16-deep steps s = 0, 1 of four k4 sub-steps e):
    //     read frags(kt, s=1) | 64 MFMAs (kt, s=0) | 48 MFMAs (kt, s=1, e=0..2) | vmcnt(0), barrier: tile kt+1 has landed and every wave
    //     has read tile kt | read frags(kt+1, s=0) | issue the DMA of tile kt+2 into tile kt's buffer, interleaved with the last 16
    //     MFMAs (kt, s=1, e=3), which also cover the LDS latency of the reads just issued.
    auto read_frags = [&](int buf, int s16, f32x4(&xf)[TM], f32x4(&wf)[TN]) {
      const float* xrow = lds + buf * TILE + (wm * TM * 16 + fr) * BK;
      const float* wrow = lds + buf * TILE + BMR * BK + (wn * TN * 16 + fr) * BK;
      const int sc = (((4 * s16 + fg) ^ rsw) << 2);  // 16-deep k-step s16: this lane's k group is logical chunk 4*s16 + fg
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xrow + j * 16 * BK + sc);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const f32x4*>(wrow + i * 16 * BK + sc);
    };
    auto prologue = [&](int kt, int s16, f32x4(&xf)[TM]) {  // eval-BatchNorm + ReLU on the activation operand (k = channel for 1x1)
      if constexpr (HAS_PRO) {
        const int kk = kt * BK + 16 * s16 + 4 * fg;
        const f32x4 sp = *reinterpret_cast<const f32x4*>(pro_s + kk);
        const f32x4 tp = *reinterpret_cast<const f32x4*>(pro_t + kk);
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) xf[j][e] = fmaxf(fmaf(xf[j][e], sp[e], tp[e]), 0.f);
      }
    };
    auto mma = [&](int e, const f32x4(&xf)[TM], const f32x4(&wf)[TN]) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][e], xf[j][e], acc[i][j], 0, 0, 0);
    };
    f32x4 xa[TM], wa[TN], xb[TM], wb[TN];
    read_frags(0, 0, xa, wa);
    if (nk > 1) stage(1, 1);
    for (int kt = 0; kt + 1 < nk; ++kt) {  // (branch-free body: the last tile is peeled, so fragment registers carry over the back edge as they are)
      const int cur = kt & 1;
      read_frags(cur, 1, xb, wb);
      prologue(kt, 0, xa);
#pragma unroll
      for (int e = 0; e < 4; ++e) mma(e, xa, wa);
      prologue(kt, 1, xb);
#pragma unroll
      for (int e = 0; e < 3; ++e) mma(e, xb, wb);
      __builtin_amdgcn_sched_barrier(0);  // (MFMAs are register-only: without this the scheduler sinks them below the barrier)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's part of tile kt+1 (issued a whole K tile ago) has landed
      __builtin_amdgcn_s_barrier();                     // ... and everyone's; every wave's reads of tile kt retired with its MFMAs
      __builtin_amdgcn_sched_barrier(0);
      read_frags(cur ^ 1, 0, xa, wa);
      stage(kt + 2 < nk ? kt + 2 : nk - 1, cur);  // (clamped: the last tiles re-stage a tile nobody reads rather than branch)
      mma(3, xb, wb);
      // issue order of this tail: the fragment reads first, then the DMAs spread evenly between the 16 MFMAs (the CU's address unit takes
      // ~16 cycles per 1-KiB DMA and is shared by the workgroup's waves: bunched up they would stall the wave's issue port)
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
      for (int q = 0; q < AP + BP; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, (TM * TN) / (AP + BP) > 0 ? (TM * TN) / (AP + BP) : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    {
      const int kt = nk - 1, cur = kt & 1;
      read_frags(cur, 1, xb, wb);
      prologue(kt, 0, xa);
#pragma unroll
      for (int e = 0; e < 4; ++e) mma(e, xa, wa);
      prologue(kt, 1, xb);
#pragma unroll
      for (int e = 0; e < 4; ++e) mma(e, xb, wb);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may be in flight into this workgroup's LDS when it ends
  } else
  if constexpr (NS > 2 && ARITH == ARITH_SPLIT && !HAS_PRO) {
    const int ch = ((fg ^ rsw) << 2), cl = (((4 + fg) ^ rsw) << 2);
    auto read_frags = [&](int buf, f16x8(&xh)[TM], f16x8(&xl)[TM], f16x8(&wh)[TN], f16x8(&wl)[TN]) {
      const float* xrow = lds + buf * TILE + (wm * TM * 16 + fr) * BK;
      const float* wrow = lds + buf * TILE + BMR * BK + (wn * TN * 16 + fr) * BK;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        xh[j] = *reinterpret_cast<const f16x8*>(xrow + j * 16 * BK + ch);
        xl[j] = *reinterpret_cast<const f16x8*>(xrow + j * 16 * BK + cl);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        wh[i] = *reinterpret_cast<const f16x8*>(wrow + i * 16 * BK + ch);
        wl[i] = *reinterpret_cast<const f16x8*>(wrow + i * 16 * BK + cl);
      }
    };
    auto mma = [&](const f16x8(&xh)[TM], const f16x8(&xl)[TM], const f16x8(&wh)[TN], const f16x8(&wl)[TN]) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[j], acc[i][j], 0, 0, 0);
        }
    };
    // wait until tile t has landed: the DMAs of tiles t+1 .. min(nk-1, issued) may stay outstanding
    auto wait_tile = [&](int t, int issued) {
      const int younger = (issued < nk - 1 ? issued : nk - 1) - t;
      if (younger >= NS - 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * CNT) : "memory");
      } else if (NS > 3 && younger == NS - 3 && NS - 3 > 0) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * CNT) : "memory");
      } else if (NS > 4 && younger == 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    f16x8 xhA[TM], xlA[TM], whA[TN], wlA[TN], xhB[TM], xlB[TM], whB[TN], wlB[TN];
    wait_tile(0, NS - 2);
    __builtin_amdgcn_s_barrier();
    read_frags(0, xhA, xlA, whA, wlA);
    for (int kt = 0; kt < nk; kt += 2) {
      // even half: MFMAs of tile kt (set A) over the LDS reads of tile kt+1 (set B)
      if (kt + 1 < nk) {
        wait_tile(kt + 1, kt + NS - 2);
        __builtin_amdgcn_s_barrier();  // tile kt+1 is complete in LDS; every wave's reads of tile kt-1 retired long ago
      }
      if (kt + NS - 1 < nk) stage(kt + NS - 1, (kt + NS - 1) % NS);
      if (kt + 1 < nk) read_frags((kt + 1) % NS, xhB, xlB, whB, wlB);
      mma(xhA, xlA, whA, wlA);
      if (kt + 1 < nk) {  // odd half: tile kt+1 (set B) over the reads of tile kt+2 (set A)
        if (kt + 2 < nk) {
          wait_tile(kt + 2, kt + NS - 1);
          __builtin_amdgcn_s_barrier();
        }
        if (kt + NS < nk) stage(kt + NS, (kt + NS) % NS);
        if (kt + 2 < nk) read_frags((kt + 2) % NS, xhA, xlA, whA, wlA);
        mma(xhB, xlB, whB, wlB);
      }
    }
  } else
  for (int kt = 0; kt < nk; ++kt) {
    int cur;
    if (NS == 1) {
      cur = 0;  // one LDS stage: the DMA of the next tile is issued after this tile's reads (second barrier below); overlap comes
                // from the 4-5 workgroups that fit a CU at 32 KB of LDS and <= 128 VGPRs, not from double buffering
    } else if (NS == 2) {
      cur = kt & 1;
      if (kt + 1 < nk) stage(kt + 1, cur ^ 1);  // next tile flies into the other buffer under this tile's MFMAs
    } else {
      cur = kt % NS;
      // tile kt has landed once at most the younger tiles' DMAs are outstanding: NS-2 tiles in steady state, fewer at the tail
      const int younger = nk - 1 - kt;
      if (younger >= NS - 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * CNT) : "memory");
      } else if (NS > 3 && younger == 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (HAS_PRO && kt == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the prologue table's ds_writes
      __builtin_amdgcn_s_barrier();  // everyone's part of tile kt is in LDS; everyone has finished reading tile kt-1's stage
      if (kt + NS - 1 < nk) stage(kt + NS - 1, (kt + NS - 1) % NS);
    }
    const float* xrow = lds + cur * TILE + (wm * TM * 16 + fr) * BK;
    const float* wrow = lds + cur * TILE + BMR * BK + (wn * TN * 16 + fr) * BK;
    if constexpr (H16) {
      // 16-bit rows: the tile holds k = 0..63; lane group fg multiplies k = 8fg..8fg+7 (logical chunk fg) in the first 32-deep step and
      // k = 32+8fg.. (chunk 4+fg) in the second
      const int c0 = ((fg ^ rsw) << 2), c1 = (((4 + fg) ^ rsw) << 2);
      f16x8 x0[TM], x1[TM], w0[TN], w1[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        x0[j] = *reinterpret_cast<const f16x8*>(xrow + j * 16 * BK + c0);
        x1[j] = *reinterpret_cast<const f16x8*>(xrow + j * 16 * BK + c1);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        w0[i] = *reinterpret_cast<const f16x8*>(wrow + i * 16 * BK + c0);
        w1[i] = *reinterpret_cast<const f16x8*>(wrow + i * 16 * BK + c1);
      }
      if constexpr (HAS_PRO) {  // eval-BatchNorm + ReLU on the activation operand, in fp32, rounded back to the storage type
        const int k0 = kt * 2 * BK + 8 * fg, k1 = k0 + 32;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const auto t0 = __builtin_bit_cast(typename std::conditional<ARITH == ARITH_BF16, bf16x8, f16x8>::type, x0[j]);
          const auto t1 = __builtin_bit_cast(typename std::conditional<ARITH == ARITH_BF16, bf16x8, f16x8>::type, x1[j]);
          typename std::conditional<ARITH == ARITH_BF16, bf16x8, f16x8>::type r0, r1;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            r0[e] = (TH)fmaxf(fmaf((float)t0[e], pro_s[k0 + e], pro_t[k0 + e]), 0.f);
            r1[e] = (TH)fmaxf(fmaf((float)t1[e], pro_s[k1 + e], pro_t[k1 + e]), 0.f);
          }
          x0[j] = __builtin_bit_cast(f16x8, r0);
          x1[j] = __builtin_bit_cast(f16x8, r1);
        }
      }
      if constexpr (TM == 8) __builtin_amdgcn_s_setprio(1);  // (keeps hipcc from moving the cluster across the barriers: cdna_hip_programming T5)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          if constexpr (ARITH == ARITH_BF16) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0[i]), __builtin_bit_cast(bf16x8, x0[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1[i]), __builtin_bit_cast(bf16x8, x1[j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[i], x0[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[i], x1[j], acc[i][j], 0, 0, 0);
          }
        }
      if constexpr (TM == 8) __builtin_amdgcn_s_setprio(0);
    } else if (SPLIT) {
      // lane group fg multiplies k = 8fg .. 8fg+7 of the tile: hi halves are logical chunk fg, lo halves chunk 4 + fg of the row
      const int ch = ((fg ^ rsw) << 2), cl = (((4 + fg) ^ rsw) << 2);
      f16x8 xh[TM], xl[TM], wh[TN], wl[TN];
      if (ARITH == ARITH_SPLIT) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          xh[j] = *reinterpret_cast<const f16x8*>(xrow + j * 16 * BK + ch);
          xl[j] = *reinterpret_cast<const f16x8*>(xrow + j * 16 * BK + cl);
        }
      } else {  // fp32 rows: k = 8fg .. 8fg+7 are logical chunks 2fg and 2fg+1
        const int c0 = (((2 * fg) ^ rsw) << 2), c1 = (((2 * fg + 1) ^ rsw) << 2);
        f32x4 sp0, sp1, tp0, tp1;
        if (HAS_PRO) {
          const int kk = kt * BK + 8 * fg;
          sp0 = *reinterpret_cast<const f32x4*>(pro_s + kk);
          sp1 = *reinterpret_cast<const f32x4*>(pro_s + kk + 4);
          tp0 = *reinterpret_cast<const f32x4*>(pro_t + kk);
          tp1 = *reinterpret_cast<const f32x4*>(pro_t + kk + 4);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          f32x4 u0 = *reinterpret_cast<const f32x4*>(xrow + j * 16 * BK + c0);
          f32x4 u1 = *reinterpret_cast<const f32x4*>(xrow + j * 16 * BK + c1);
          if (HAS_PRO) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              u0[e] = fmaxf(fmaf(u0[e], sp0[e], tp0[e]), 0.f);
              u1[e] = fmaxf(fmaf(u1[e], sp1[e], tp1[e]), 0.f);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v0 = __builtin_amdgcn_fmed3f(u0[e], -65504.0f, 65504.0f), v1 = __builtin_amdgcn_fmed3f(u1[e], -65504.0f, 65504.0f);
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            xh[j][e] = h0;
            xh[j][4 + e] = h1;
            xl[j][e] = (_Float16)(v0 - (float)h0);
            xl[j][4 + e] = (_Float16)(v1 - (float)h1);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        wh[i] = *reinterpret_cast<const f16x8*>(wrow + i * 16 * BK + ch);
        wl[i] = *reinterpret_cast<const f16x8*>(wrow + i * 16 * BK + cl);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[j], acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        const int sc = (((4 * s + fg) ^ rsw) << 2);  // 16-deep k-step s: this lane's k group is logical chunk 4s+fg
        f32x4 xf[TM], wf[TN];
#pragma unroll
        for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xrow + j * 16 * BK + sc);
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const f32x4*>(wrow + i * 16 * BK + sc);
        if (HAS_PRO) {  // eval-BatchNorm + ReLU on the activation operand (pre-activation Residual.conv1), k = channel for 1x1
          const int kk = kt * BK + 16 * s + 4 * fg;
          const f32x4 sp = *reinterpret_cast<const f32x4*>(pro_s + kk);
          const f32x4 tp = *reinterpret_cast<const f32x4*>(pro_t + kk);
#pragma unroll
          for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) xf[j][e] = fmaxf(fmaf(xf[j][e], sp[e], tp[e]), 0.f);
        }
        if constexpr (ARITH == ARITH_R_BF16 || ARITH == ARITH_R_F16) {
          // lane (l % 16, l / 16) holds k = 4 (l / 16) .. + 3 of its row of either operand: exactly what one v_mfma_f32_16x16x16 takes from it
          typedef short r16x4 __attribute__((ext_vector_type(4)));
          r16x4 wh[TN], xh[TM];
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            if constexpr (ARITH == ARITH_R_BF16) wh[i] = __builtin_bit_cast(r16x4, bf16x4{(bf16_t)wf[i][0], (bf16_t)wf[i][1], (bf16_t)wf[i][2], (bf16_t)wf[i][3]});
            else wh[i] = __builtin_bit_cast(r16x4, f16x4{(f16_t)wf[i][0], (f16_t)wf[i][1], (f16_t)wf[i][2], (f16_t)wf[i][3]});
          }
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            if constexpr (ARITH == ARITH_R_BF16) xh[j] = __builtin_bit_cast(r16x4, bf16x4{(bf16_t)xf[j][0], (bf16_t)xf[j][1], (bf16_t)xf[j][2], (bf16_t)xf[j][3]});
            else xh[j] = __builtin_bit_cast(r16x4, f16x4{(f16_t)xf[j][0], (f16_t)xf[j][1], (f16_t)xf[j][2], (f16_t)xf[j][3]});
          }
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) {
              if constexpr (ARITH == ARITH_R_BF16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh[i], xh[j], acc[i][j], 0, 0, 0);
              else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, wh[i]), __builtin_bit_cast(f16x4, xh[j]), acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
              for (int j = 0; j < TM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][e], xf[j][e], acc[i][j], 0, 0, 0);
        }
      }
    }
    if (NS == 2) __syncthreads();  // next tile landed (vmcnt drained by the barrier's fence); every wave is done reading `cur`
    if (NS == 1 && kt + 1 < nk) {
      __syncthreads();  // every wave has read tile kt out of the stage
      stage(kt + 1, 0);
      __syncthreads();  // tile kt+1 landed
    }
  }

  KPF_STAMP(2);
  // ---- epilogue: lane holds channels n..n+3 (n = tile + 4*fg) of pixel m (= tile + fr) ----
  const unsigned fl = a.flags;
  const bool interior = (m0 + BM <= a.M) && (n0 + BN <= a.N) && a.vec && !(fl & KPF_OUT_NCHW);  // workgroup-uniform
  if (interior) {
    // fast path: whole tile in range and 16-byte aligned -> no per-element guards, float4 bias / gamma / residual / store.
    // Pixel tiles outermost: a wave's consecutive store instructions then cover the same 16 rows' adjacent 64-byte segments,
    // so L2 merges them into whole 128-byte lines before write-back.
    // Bias / layer scale once per channel quad; ALL residual rows of the tile are requested before the first store (`out` may alias
    // `res`, so the compiler keeps loads behind earlier stores: row by row, each row would expose a full memory round trip).
    f32x4 bvv[TN], gvv[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + (wn * TN + i) * 16 + fg * 4;
      bvv[i] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n) : zero4;
      gvv[i] = (EPI == EPI_RES && (fl & KPF_RES_GAMMA)) ? *reinterpret_cast<const f32x4*>(a.gamma + n) : zero4;
    }
    // 16-bit output without residual, 64-channel wave tiles: the results go through LDS so that the global stores are whole 128-byte
    // rows (a lane's 4 channels are 8 bytes: stored directly, a pixel row would be written in 32-byte pieces by four different
    // instructions — the GELU layers of the 16-bit path write 4C-wide tensors and were bound by exactly that)
    constexpr bool HAS_RES = EPI == EPI_RES || EPI == EPI_GGRAD;  // the epilogue reads a second operand tile
    constexpr bool STAGED = H16 && !HAS_RES && TN == 4;
    constexpr int RS = TN * 16 + 8;  // staging row stride in elements (144 B: the 8-byte writes of a fragment column spread over the banks)
    TH* stg = reinterpret_cast<TH*>(lds) + wave * (TM * 16) * RS;
    if constexpr (STAGED) __syncthreads();  // every wave is done reading the last K tile: LDS becomes the staging area
    f32x4 rvv[HAS_RES ? TM : 1][TN];
    if (HAS_RES) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const long m = m0 + (wm * TM + j) * 16 + fr;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const long o = m * a.res_ld + a.res_coff + n0 + (wn * TN + i) * 16 + fg * 4;
          if constexpr (H16) rvv[j][i] = kpf_ld4(reinterpret_cast<const TH*>(a.res) + o);
          else rvv[j][i] = *reinterpret_cast<const f32x4*>(a.res + o);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const long m = m0 + (wm * TM + j) * 16 + fr;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = n0 + (wn * TN + i) * 16 + fg * 4;
        const f32x4 bv = bvv[i], gv = gvv[i];
        f32x4 v = acc[i][j];
        if (SPLIT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= a.w_unscale;
        }
        if (EPI == EPI_RES) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float y = v[e] + bv[e];
            if (fl & KPF_RES_GAMMA) y *= gv[e];
            y += rvv[HAS_RES ? j : 0][i][e];
            if (fl & KPF_RELU_AFTER_RES) y = fmaxf(y, 0.f);
            v[e] = y;
          }
        } else if (EPI == EPI_GGRAD) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (v[e] + bv[e]) * gelu_grad_erf(rvv[HAS_RES ? j : 0][i][e]);
        } else if (EPI == EPI_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_of<ARITH>(v[e] + bv[e]);
        } else if (EPI == EPI_GELU2) {
          f32x4 z;
#pragma unroll
          for (int e = 0; e < 4; ++e) z[e] = v[e] + bv[e];
          if constexpr (H16) kpf_st4(reinterpret_cast<TH*>(const_cast<float*>(a.res)) + m * a.res_ld + a.res_coff + n, z);
          else STORE4(const_cast<float*>(a.res) + m * a.res_ld + a.res_coff + n, z);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (H16) z[e] = (float)(TH)z[e];  // (gelu of the value as stored: what a separate pass over the 16-bit tensor computes)
            v[e] = gelu_exact_erf(z[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = v[e] + bv[e];
            v[e] = (fl & KPF_ACT_RELU) ? fmaxf(y, 0.f) : y;
          }
          if (fl & KPF_ACT_LEAKY) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.01f * v[e]);
          }
        }
        if constexpr (STAGED)
          kpf_st4(stg + (j * 16 + fr) * RS + i * 16 + fg * 4, v);
        else if constexpr (H16)
          kpf_st4(reinterpret_cast<TH*>(a.out) + m * a.out_ld + a.out_coff + n, v);
        else if (fl & KPF_OUT_SPLIT)
          kpf_store_split4(a.out + m * a.out_ld + a.out_coff, n, v);
        else if (a.st_policy == 2)  // (wave-uniform: large fp32 outputs leave with non-temporal stores, see kpf_conv2d_f32)
          asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(a.out + m * a.out_ld + a.out_coff + n), "v"(v) : "memory");
        else
          STORE4(a.out + m * a.out_ld + a.out_coff + n, v);
      }
    }
    if constexpr (STAGED) {
      __syncthreads();  // (only this wave's own region is read back; the barrier doubles as the LDS fence)
      TH* ob = reinterpret_cast<TH*>(a.out) + (long)(m0 + wm * TM * 16) * a.out_ld + a.out_coff + n0 + wn * TN * 16;
      const int prow = lane >> 3, pch = (lane & 7) * 8;  // 8 lanes x 16 bytes = one 128-byte pixel row of the wave tile, 8 rows per instruction
#pragma unroll
      for (int r = 0; r < TM * 2; ++r) {
        const int px = r * 8 + prow;
        const f32x4 q = *reinterpret_cast<const f32x4*>(stg + px * RS + pch);
        *reinterpret_cast<f32x4*>(ob + (long)px * a.out_ld + pch) = q;
      }
    }
    KPF_STAMP(3);
    return;
  }
  // NCHW output (the 105-channel heads, model/model.py finals: fp32 planes also on the 16-bit path) when the pixel tile lies inside one image:
  // the tile is transposed through LDS, 16 * WN channels at a time, and every channel's BM pixels leave as one contiguous run of float4 stores
  // (the per-element path below writes 64-byte pieces with four scalar stores per lane and quad: 125 us for the 262144 x 105 x 128 head, round 5).
  if constexpr (EPI == EPI_LIN) {
    if ((fl & KPF_OUT_NCHW) && a.ohow % BM == 0 && m0 + BM <= a.M && !(fl & KPF_OUT_SPLIT)) {  // workgroup-uniform
      constexpr int RS = BM + 4;  // row stride in floats: lane groups fg land 16 banks apart
      static_assert(WN * 16 * RS <= NS * TILE, "the transpose staging fits the K buffers");
      const int b = m0 / a.ohow, pix0 = m0 - b * a.ohow;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        __syncthreads();  // every wave is done with the K tiles (first pass) / with the previous pass's staging rows
        const int nq = n0 + (wn * TN + i) * 16 + fg * 4;
        float bq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[e] = (a.bias && nq + e < a.N) ? a.bias[nq + e] : 0.f;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int ml = (wm * TM + j) * 16 + fr;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float y = (SPLIT ? acc[i][j][e] * a.w_unscale : acc[i][j][e]) + bq[e];
            if (fl & KPF_ACT_RELU) y = fmaxf(y, 0.f);
            if (fl & KPF_ACT_LEAKY) y = fmaxf(y, 0.01f * y);
            lds[(wn * 16 + fg * 4 + e) * RS + ml] = y;
          }
        }
        __syncthreads();
        for (int idx = tid; idx < WN * 16 * (BM / 4); idx += 64 * NW) {
          const int r = idx / (BM / 4), c4 = idx - r * (BM / 4);
          const int n = n0 + ((r >> 4) * TN + i) * 16 + (r & 15);
          if (n < a.N) STORE4(a.out + ((long)b * a.N + n) * a.ohow + pix0 + 4 * c4, *reinterpret_cast<const f32x4*>(lds + r * RS + 4 * c4));
        }
      }
      KPF_STAMP(3);
      return;
    }
  }
  // edge / NCHW / unaligned path: per-element guards
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + (wm * TM + j) * 16 + fr;
    if (m >= a.M) continue;
    const int b = m / a.ohow;
    const int pix = m - b * a.ohow;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + (wn * TN + i) * 16 + fg * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= a.N) continue;
        float y = (SPLIT ? acc[i][j][e] * a.w_unscale : acc[i][j][e]) + (a.bias ? a.bias[n + e] : 0.f);
        if (EPI == EPI_GELU) y = gelu_of<ARITH>(y);
        if (EPI == EPI_GELU2) {
          if constexpr (H16) {
            reinterpret_cast<TH*>(const_cast<float*>(a.res))[(long)m * a.res_ld + a.res_coff + n + e] = (TH)y;
            y = (float)(TH)y;
          } else {
            const_cast<float*>(a.res)[(long)m * a.res_ld + a.res_coff + n + e] = y;
          }
          y = gelu_exact_erf(y);
        }
        if (EPI == EPI_LIN && (fl & KPF_ACT_RELU)) y = fmaxf(y, 0.f);
        if (EPI == EPI_LIN && (fl & KPF_ACT_LEAKY)) y = fmaxf(y, 0.01f * y);
        if (EPI == EPI_RES) {
          if (fl & KPF_RES_GAMMA) y *= a.gamma[n + e];
          if constexpr (H16) y += (float)reinterpret_cast<const TH*>(a.res)[(long)m * a.res_ld + a.res_coff + n + e];
          else y += a.res[(long)m * a.res_ld + a.res_coff + n + e];
          if (fl & KPF_RELU_AFTER_RES) y = fmaxf(y, 0.f);
        }
        if (EPI == EPI_GGRAD) {
          if constexpr (H16) y *= gelu_grad_erf((float)reinterpret_cast<const TH*>(a.res)[(long)m * a.res_ld + a.res_coff + n + e]);
          else y *= gelu_grad_erf(a.res[(long)m * a.res_ld + a.res_coff + n + e]);
        }
        if (fl & KPF_OUT_NCHW) {
          a.out[((long)b * a.N + n + e) * a.ohow + pix] = y;  // (fp32 also on the 16-bit path: the heads' output)
        } else if constexpr (H16) {
          reinterpret_cast<TH*>(a.out)[(long)m * a.out_ld + a.out_coff + n + e] = (TH)y;
        } else if (fl & KPF_OUT_SPLIT) {
          const float yc = __builtin_amdgcn_fmed3f(y, -65504.0f, 65504.0f);
          const _Float16 h = (_Float16)yc;
          _Float16* blk = reinterpret_cast<_Float16*>(a.out + (long)m * a.out_ld + a.out_coff + ((n + e) & ~31)) + ((n + e) & 31);
          blk[0] = h;
          blk[32] = (_Float16)(yc - (float)h);
        } else
          a.out[(long)m * a.out_ld + a.out_coff + n + e] = y;
      }
    }
  }
}

// Two kernel names so that profiles tell the arithmetic apart: igemm_f32_kernel = f32-input MFMA, igemm_split_kernel = 3 x f16 MFMA
// on split operands (ARITH 1: activations pre-split in memory, 2: split in registers).
template <int TM, int TN, int WM, int WN, bool IS1X1, bool HAS_PRO, int EPI, int NS>
__global__ __launch_bounds__(64 * WM * WN) void igemm_f32_kernel(const ConvArgs a) {
  igemm_body<TM, TN, WM, WN, IS1X1, HAS_PRO, EPI, ARITH_F32, NS>(kpf_group_args(a));
}
template <int TM, int TN, int WM, int WN, bool IS1X1, int EPI, int ARITH>
__global__ __launch_bounds__(64 * WM * WN) void igemm_r16_kernel(const ConvArgs a) {  // fp32 storage, 16-bit products (ARITH_R_*)
  igemm_body<TM, TN, WM, WN, IS1X1, false, EPI, ARITH, 2>(kpf_group_args(a));
}
template <int TM, int TN, int WM, int WN, bool IS1X1, bool HAS_PRO, int EPI, int ARITH, int NS>
__global__ __launch_bounds__(64 * WM * WN) void igemm_split_kernel(const ConvArgs a) {
  igemm_body<TM, TN, WM, WN, IS1X1, HAS_PRO, EPI, ARITH, NS>(kpf_group_args(a));
}
// Single LDS stage, registers capped for 4 waves per SIMD: with the MFMA phase of a K tile this short, four or five co-resident
// workgroups hide each other's DMA / LDS / barrier latencies better than double buffering inside two (tools/split_probe.hip: 600 vs
// 430-490 TFLOP/s for the bare loop).
template <int TM, int TN, int WM, int WN, bool IS1X1, bool HAS_PRO, int EPI, int ARITH>
__global__ __launch_bounds__(64 * WM * WN, 4) void igemm_split_occ_kernel(const ConvArgs a) {
  igemm_body<TM, TN, WM, WN, IS1X1, HAS_PRO, EPI, ARITH, 1>(kpf_group_args(a));
}

// 16-bit storage (bf16 / f16) GEMMs: igemm_h16_kernel (two LDS stages) / igemm_h16_occ_kernel (one stage, 4 waves per SIMD)
template <int TM, int TN, int WM, int WN, bool IS1X1, bool HAS_PRO, int EPI, int ARITH, int NS>
__global__ __launch_bounds__(64 * WM * WN) void igemm_h16_kernel(const ConvArgs a) {
  igemm_body<TM, TN, WM, WN, IS1X1, HAS_PRO, EPI, ARITH, NS>(kpf_group_args(a));
}
template <int TM, int TN, int WM, int WN, bool IS1X1, bool HAS_PRO, int EPI, int ARITH>
__global__ __launch_bounds__(64 * WM * WN, 4) void igemm_h16_occ_kernel(const ConvArgs a) {
  igemm_body<TM, TN, WM, WN, IS1X1, HAS_PRO, EPI, ARITH, 1>(kpf_group_args(a));
}

template <int TM, int TN, int WM, int WN, bool IS1X1, bool HAS_PRO, int EPI, int ARITH, int NS>
int launch_one(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN, RPP = 8 * WM * WN;
  constexpr bool R16 = ARITH == ARITH_R_BF16 || ARITH == ARITH_R_F16;
  constexpr bool WHOLE = NS > 2 || ARITH == ARITH_F32 || R16;
  constexpr int BMR = WHOLE ? (BM + RPP - 1) / RPP * RPP : BM, BNR = WHOLE ? (BN + RPP - 1) / RPP * RPP : BN;
  constexpr bool H16 = ARITH == ARITH_BF16 || ARITH == ARITH_F16;
  size_t lds = (size_t)(NS * (BMR + BNR) * BK + (HAS_PRO ? 2 * a.Kp * (H16 ? 2 : 1) : 0)) * sizeof(float);
  if (H16 && EPI != EPI_RES && EPI != EPI_GGRAD && TN == 4) {  // LDS-staged epilogue of the 16-bit path (igemm_body: STAGED)
    const size_t stg = (size_t)WM * WN * TM * 16 * (TN * 16 + 8) * 2;
    if (stg > lds) lds = stg;
  }
  void (*kern)(const ConvArgs);
  if constexpr (ARITH == ARITH_F32)
    kern = igemm_f32_kernel<TM, TN, WM, WN, IS1X1, HAS_PRO, EPI, NS>;
  else if constexpr (R16)
    kern = igemm_r16_kernel<TM, TN, WM, WN, IS1X1, EPI, ARITH>;
  else if constexpr (H16 && NS == 1)
    kern = igemm_h16_occ_kernel<TM, TN, WM, WN, IS1X1, HAS_PRO, EPI, ARITH>;
  else if constexpr (H16)
    kern = igemm_h16_kernel<TM, TN, WM, WN, IS1X1, HAS_PRO, EPI, ARITH, NS>;
  else if constexpr (NS == 1)
    kern = igemm_split_occ_kernel<TM, TN, WM, WN, IS1X1, HAS_PRO, EPI, ARITH>;
  else
    kern = igemm_split_kernel<TM, TN, WM, WN, IS1X1, HAS_PRO, EPI, ARITH, NS>;
  static std::atomic<bool> lds_opt_in[KPF_MAX_DEVICES];  // per device: the attribute does not carry over to other GPUs of the process
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(kern), lds_opt_in)) {
    kpf_set_error("kpf_conv2d_f32: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  if (lds > 160 * 1024) {
    kpf_set_error("kpf_conv2d_f32: operand prologue too long for LDS (Kp=%d)", a.Kp);
    return KPF_EINVAL;
  }
  hipLaunchKernelGGL(kern, dim3(a.nblk, a.groups > 1 ? a.groups : 1), dim3(64 * WM * WN), lds, st, a);
  return kpf_check_launch("kpf_conv2d_f32");
}

// Instantiated variants per tile shape (the combinations the model produces):
//   1x1 + operand prologue + linear/ReLU   (Residual.conv1)            1x1 + linear/ReLU        (heads, embeddings, skip convs)
//   1x1 + GELU                             (ConvNeXt pwconv1)          1x1 + residual           (pwconv2, Residual.conv3)
//   conv + linear/ReLU                     (3x3, stems, downsamples)   conv + residual          (BasicBlock.conv2)
// Instantiated per tile shape: {1x1 (+prologue | GELU | residual | linear), general conv (linear | residual)} x {f32 MFMA, split, split weights}.
template <int TM, int TN, int WM, int WN, int ARITH, int NS>
int launch_arith(ConvArgs& a, bool is1x1, hipStream_t st) {
  const bool res = a.flags & KPF_RES_ADD, gelu = a.flags & KPF_ACT_GELU;
  if (a.ps) {
    if (!is1x1 || res || gelu || ARITH == ARITH_SPLIT || (NS == 1 && (ARITH == ARITH_BF16 || ARITH == ARITH_F16))) {
      kpf_set_error("kpf_conv2d_f32: the operand prologue is only supported for 1x1 stride-1 convolutions with Cin %% 32 == 0, fp32 activations and a linear/ReLU epilogue");
      return KPF_EINVAL;
    }
    return launch_one<TM, TN, WM, WN, true, true, EPI_LIN, ARITH == ARITH_SPLIT ? ARITH_F32 : ARITH, NS>(a, st);
  }
  if (gelu) {
    if (!is1x1 || res) {
      kpf_set_error("kpf_conv2d_f32: GELU is only supported on 1x1 convolutions (Cin %% 32 == 0) without residual");
      return KPF_EINVAL;
    }
    if (a.flags & KPF_ACT_GELU_SAVE) {
      if constexpr (ARITH == ARITH_F32 && NS == 2) return launch_one<TM, TN, WM, WN, true, false, EPI_GELU2, ARITH, NS>(a, st);
      kpf_set_error("kpf_conv2d_f32: KPF_ACT_GELU_SAVE needs fp32 arithmetic");
      return KPF_EINVAL;
    }
    return launch_one<TM, TN, WM, WN, true, false, EPI_GELU, ARITH, NS>(a, st);
  }
  if (a.flags & KPF_RES_GELU_GRAD) {
    if constexpr (ARITH == ARITH_F32 && NS == 2) {
      if (a.KH == 1 && a.KW == 1 && a.sh == 1 && a.sw == 1)
        return is1x1 ? launch_one<TM, TN, WM, WN, true, false, EPI_GGRAD, ARITH, NS>(a, st) : launch_one<TM, TN, WM, WN, false, false, EPI_GGRAD, ARITH, NS>(a, st);
    }
    kpf_set_error("kpf_conv2d_f32: KPF_RES_GELU_GRAD needs a 1x1 stride-1 convolution in fp32 arithmetic");
    return KPF_EINVAL;
  }
  if (is1x1) return res ? launch_one<TM, TN, WM, WN, true, false, EPI_RES, ARITH, NS>(a, st) : launch_one<TM, TN, WM, WN, true, false, EPI_LIN, ARITH, NS>(a, st);
  return res ? launch_one<TM, TN, WM, WN, false, false, EPI_RES, ARITH, NS>(a, st) : launch_one<TM, TN, WM, WN, false, false, EPI_LIN, ARITH, NS>(a, st);
}

template <int TM, int TN, int WM, int WN, int NS_SPLIT = 2>
int launch_cfg(ConvArgs& a, bool is1x1, hipStream_t st) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  const int tilesM = (a.M + BM - 1) / BM;
  a.tilesN = (a.N + BN - 1) / BN;
  a.nblk = tilesM * a.tilesN;
  // start skew of co-resident workgroups (igemm_body), in units of 2048 clocks: a fifth of one tile's matrix-pipe time (BM * BN * Kp / 128 clocks), at most 8;
  // KPF_STAGGER = the cap (default 0 = off; 8 was the measured setting).  Short tiles get none.
  static const int stagger = []() { const char* e = getenv("KPF_STAGGER"); return e ? atoi(e) : 0; }();
  const long skew = (long)BM * BN * a.Kp / (128L * 5 * 2048);
  a.skew = a.nblk >= 512 ? (int)(skew < stagger ? skew : stagger) : 0;
  if constexpr ((TM == 4 && TN == 4 && WN == 2 && (WM == 4 || WM == 2)) || (TM == 2 && TN == 4 && WM == 4 && WN == 1)) {
    // 16-bit products on fp32 storage (KPF_MMA_BF16 / _F16): plain 1x1 / linear layers on the three tile shapes the fusion head's wide Linears take; anything
    // else keeps the fp32 products (the flag is a permission, not a demand)
    if ((a.flags & (KPF_MMA_BF16 | KPF_MMA_F16)) && is1x1 && !a.ps && !(a.flags & (KPF_RES_ADD | KPF_ACT_GELU | KPF_IN_SPLIT | KPF_W_SPLIT | KPF_OUT_SPLIT | KPF_OUT_NCHW))) {
      if (a.flags & KPF_MMA_BF16) return launch_one<TM, TN, WM, WN, true, false, EPI_LIN, ARITH_R_BF16, 2>(a, st);
      return launch_one<TM, TN, WM, WN, true, false, EPI_LIN, ARITH_R_F16, 2>(a, st);
    }
  }
  if (a.flags & KPF_IN_SPLIT) return launch_arith<TM, TN, WM, WN, ARITH_SPLIT, NS_SPLIT>(a, is1x1, st);
  if (a.flags & KPF_W_SPLIT) return launch_arith<TM, TN, WM, WN, ARITH_SPLIT_W, NS_SPLIT>(a, is1x1, st);
  return launch_arith<TM, TN, WM, WN, ARITH_F32, NS_SPLIT == 1 ? 2 : 2>(a, is1x1, st);
}

// Tile choice.  The kernel is MFMA-bound and co-resident workgroups share a CU's matrix pipe (they hide each other's bubbles, they
// do not add throughput), so a launch lasts about ceil(blocks / 256 CUs) rounds of one tile's work: 384 blocks cost as much as
// 512.  `pen` is the measured relative cost per FLOP of each tile shape (smaller tiles amortise staging and epilogue worse).
// (A persistent-workgroup variant with cross-tile prefetch was measured 5-14 % slower than letting the dispatcher balance.)
constexpr int KPF_NUM_TILE_CFGS = 18;  // cases of the switch in kpf_conv2d_f32 (9-12: LDS-ring variants of 8, 0, 1, 2 for split operands)
struct Cfg {
  int bm, bn;
  double pen;
};
static const Cfg kCfgs[] = {{128, 128, 1.00}, {128, 96, 1.03}, {128, 64, 1.10}, {256, 48, 1.10}, {128, 112, 1.03},
                            {64, 128, 1.10},  {64, 64, 1.25},  {32, 64, 1.60},  {256, 128, 0.97}};

double cfg_cost(const Cfg& c, long M, long N, int groups = 1) {  // (a grouped launch runs `groups` times the tiles in the same rounds of the chip)
  const long tm = (M + c.bm - 1) / c.bm, tn = (N + c.bn - 1) / c.bn;
  const long blocks = tm * tn * (groups > 1 ? groups : 1);
  const long rounds = (blocks + 255) / 256;
  return (double)rounds * c.bm * c.bn * c.pen;
}

#ifdef KPF_CONV_H16
// ---------------------------------------------------------------------------------------------------------------------------------
// gemm16_8ph_kernel: 256 x 256 x 64 tiles, eight phases per two K tiles, for the dense 1x1 layers of the 16-bit path (round 4).
//
// The structure of cdna_hip_programming.md's "256^2 8-phase template", written for this library's operands and epilogues:
//  * 8 waves = 2 (pixel halves, wr) x 4 (channel quarters, wc); a wave owns 128 pixels x 64 channels = 32 accumulator tiles (128 registers), in four
//    QUADRANTS of 64 pixels x 32 channels; one phase = the 16 MFMAs (v_mfma_f32_16x16x32_{bf16,f16}) of one quadrant over one 64-deep K tile.
//  * Waves 4-7 run ONE barrier behind waves 0-3 (a stagger): every SIMD hosts one wave of each group, and while one issues its 16 MFMAs the other reads
//    its fragments from LDS and issues its share of the LDS-DMA — the matrix pipe of a SIMD alternates between its two waves and never waits for a read.
//  * Two K-tile buffers of 64 KB; a K tile is staged as four 16-KB HALF-TILES, one per phase, each = the rows one phase reads: A half s = pixel rows
//    {wr*128 + s*64 + [0,64)} of both wr, B half s = channels {wc*64 + s*32 + [0,32)} of all four wc (the LDS row <-> tile row permutation lives in the
//    DMA's source addresses).  A half-tile is re-staged two phases after its last read (one phase after for the B half whose reads a counted
//    lgkmcnt(8) retires before the barrier), so three half-tiles are always in flight; the only vmcnt in the loop is a counted vmcnt(6) once per K
//    tile, four phases before the first read of the data it covers.  LDS image and XOR swizzle are the ones igemm_body uses (128-byte rows, 16-byte
//    chunk c of row r at c ^ ((r >> 1) & 7): conflict-free ds_read_b128 fragments), so both kernels read the same packed weights.
//  Schedule of K tile t (buffer b = t & 1), per wave:
//    q1: read B-sub0 (4 x b128) then A-sub0 (8) | DMA A-half1 of tile t+1 -> buffer b^1 | lgkmcnt(8) | barrier | 16 MFMA quadrant (0,0) | barrier
//    q2: read B-sub1 (4)                         | DMA B-half0 of tile t+2 -> buffer b                 | barrier | 16 MFMA quadrant (0,1) | barrier
//    q3: read A-sub1 (8)                         | DMA A-half0 of tile t+2 -> buffer b                 | barrier | 16 MFMA quadrant (1,1) | barrier
//    q4:                                           DMA B-half1 of tile t+2 -> buffer b | vmcnt(6)      | barrier | 16 MFMA quadrant (1,0) | barrier
//  Requirements (the dispatcher checks them): dense 1x1, K % 128 == 0 (an even number of K tiles), N % 256 == 0, 16-byte aligned rows.
// ---------------------------------------------------------------------------------------------------------------------------------
// 16-byte output store under a cache policy chosen at launch (wave-uniform branch).  A GEMM whose output is larger than the L2s writes every byte through its XCD's
// L2 exactly once; with plain stores those lines stay in the L2 and push the weight panels out (M = 65536, N = 2048: 4 MB of results per XCD and round of
// tiles = the whole L2), so the next round fetches the weights again from the Infinity Cache.  `sc1` stores go to memory without keeping the line.
__device__ __forceinline__ void g8_store16(void* p, const f32x4 q, const int policy) {
  if (policy == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
  else if (policy == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(q) : "memory");
  else *reinterpret_cast<f32x4*>(p) = q;
}
constexpr int G8_BUF = 65536, G8_HALF = 16384, G8_B = 32768;  // bytes: K-tile buffer, half-tile, offset of the B halves inside a buffer
template <int V> using ic = std::integral_constant<int, V>;

// PERSIST (linear / GELU epilogues, M % 256 == 0): one workgroup per CU walks tiles id, id + grid, ...; the first two K tiles of the NEXT tile are
// DMA'd into the (idle) K buffers BEFORE the epilogue of the current one, which stages through the 32 KB beyond them — the prologue latency of a tile
// (2-3 us of a 27-us tile at K = 512: profiles/r04_g8_ablation.txt) and the workgroup launch disappear behind the GELU and the stores.
// LNF (KPF_PRO_LN, GELU epilogue): the operand is the RAW output x of the depthwise stencil and the LayerNorm in front of this layer
// (convNeXT/convnext.py:42-44: pwconv1(norm(x))) is folded into the GEMM algebraically —
//     W (gamma * (x - mean_m) * rstd_m + beta) + b  =  rstd_m * (W' x - mean_m * s) + b',   W' = W diag(gamma),  s[n] = sum_k W'[n][k],  b' = W beta + b
// — so the normalisation costs two fused multiply-adds per accumulator in the epilogue instead of a pass over the tensor: a.w holds W', a.bias b', a.pt s and
// a.ps the per-row (mean, rstd) pairs (kpf_ln_stats_merge).  The eight pairs of a lane's rows are requested two K tiles before the main loop ends.
template <int EPI, int ARITH, bool PERSIST, bool LNF = false>
__global__ __launch_bounds__(512, 2) void gemm16_8ph_kernel(const ConvArgs a) {
  static_assert(!PERSIST || EPI != EPI_RES, "the persistent form is built for the store-only epilogues");
  static_assert(!LNF || EPI == EPI_GELU, "the folded LayerNorm exists for pwconv1 (GELU epilogue)");
  using TH = typename std::conditional<ARITH == ARITH_BF16, bf16_t, f16_t>::type;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const LB = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int bid = blockIdx.x;
  {  // XCD-aware bijective remap (igemm_body): an XCD gets a contiguous range of tiles, channel tiles fastest (PERSIST: of every round of tiles)
    const int nb = PERSIST ? (int)gridDim.x : a.nblk, q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  int tile = bid;
  int m0, n0;
  // ---- staging: thread -> (row lr of a 64-row DMA block, chunk position cp); a half-tile is two DMA instructions per wave ----
  const int lr = tid >> 3, cp = tid & 7;
  const int kc = ((cp ^ ((lr >> 1) & 7)) << 2);  // logical k offset (4-byte words) this lane fetches: the swizzle is applied to the SOURCE
  // PERSIST (no M tail): the four rows a lane stages per operand differ by compile-time multiples of the (wave-uniform) row stride, so one pointer
  // per operand stays in registers across the epilogue instead of eight; otherwise rows are clamped to M - 1 one by one
  const float* pa[2][2];
  const float* pb[2][2];
  auto set_tile = [&](int t) {
    const int nt = t % a.tilesN, mt = t / a.tilesN;
    m0 = mt * 256;
    n0 = nt * 256;
    if constexpr (PERSIST) {
      pa[0][0] = a.in + (long)(m0 + lr) * a.in_ld + a.in_coff + kc;
      pb[0][0] = a.w + (long)(n0 + (lr >> 5) * 64 + (lr & 31)) * a.Kp + kc;
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int m = m0 + i * 128 + s * 64 + lr;  // LDS row i*64 + lr of A half s
          m = m < a.M ? m : a.M - 1;           // (rows beyond M: products land in accumulators that are never stored)
          pa[s][i] = a.in + (long)m * a.in_ld + a.in_coff + kc;
          const int n = n0 + (2 * i + (lr >> 5)) * 64 + s * 32 + (lr & 31);  // LDS row i*64 + lr of B half s
          pb[s][i] = a.w + (long)n * a.Kp + kc;
        }
    }
  };
  set_tile(tile);
  auto src_A = [&](int s, int i) { return PERSIST ? pa[0][0] + (long)(i * 128 + s * 64) * a.in_ld : pa[s][i]; };
  auto src_B = [&](int s, int i) { return PERSIST ? pb[0][0] + (long)(i * 128 + s * 32) * a.Kp : pb[s][i]; };
  const int dma_row = wave * 8 * 128;  // this wave's 8 rows of a 64-row DMA block (bytes)
  auto stage_A = [&](auto BUF, auto S, int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src_A(decltype(S)::value, i) + kt * BK),
                                       (lds_void_t*)(LB + decltype(BUF)::value * G8_BUF + decltype(S)::value * G8_HALF + i * 8192 + dma_row), 16, 0, 0);
  };
  auto stage_B = [&](auto BUF, auto S, int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src_B(decltype(S)::value, i) + kt * BK),
                                       (lds_void_t*)(LB + decltype(BUF)::value * G8_BUF + G8_B + decltype(S)::value * G8_HALF + i * 8192 + dma_row), 16, 0, 0);
  };

  // ---- fragments ----
  const int fr = lane & 15, fg = lane >> 4, rsw = (fr >> 1) & 7;
  const int ch0 = ((fg ^ rsw) << 4), ch1 = (((4 + fg) ^ rsw) << 4);  // byte offsets of this lane's chunk in k-step 0 / 1 of a row
  const char* const a_rd = LB + (wr * 64 + fr) * 128;
  const char* const b_rd = LB + G8_B + (wc * 32 + fr) * 128;
  f16x8 xa[2][4];     // [k-step][pixel tile]      of the current A sub-tile
  f16x8 wb[2][2][2];  // [sub][k-step][channel tile]
  f32x4 acc[2][2][2][4];  // [channel sub][channel tile][pixel sub][pixel tile]
  auto read_A = [&](auto BUF, auto S) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xa[0][j] = *reinterpret_cast<const f16x8*>(a_rd + decltype(BUF)::value * G8_BUF + decltype(S)::value * G8_HALF + j * 2048 + ch0);
      xa[1][j] = *reinterpret_cast<const f16x8*>(a_rd + decltype(BUF)::value * G8_BUF + decltype(S)::value * G8_HALF + j * 2048 + ch1);
    }
  };
  auto read_B = [&](auto BUF, auto S) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      wb[decltype(S)::value][0][i] = *reinterpret_cast<const f16x8*>(b_rd + decltype(BUF)::value * G8_BUF + decltype(S)::value * G8_HALF + i * 2048 + ch0);
      wb[decltype(S)::value][1][i] = *reinterpret_cast<const f16x8*>(b_rd + decltype(BUF)::value * G8_BUF + decltype(S)::value * G8_HALF + i * 2048 + ch1);
    }
  };
  auto mma = [&](auto SM, auto SN) {
    constexpr int sm = decltype(SM)::value, sn = decltype(SN)::value;
    __builtin_amdgcn_s_setprio(1);  // (keeps hipcc from moving the cluster across the barriers: cdna_hip_programming T5)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (ARITH == ARITH_BF16)
            acc[sn][i][sm][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[sn][ks][i]), __builtin_bit_cast(bf16x8, xa[ks][j]), acc[sn][i][sm][j], 0, 0, 0);
          else
            acc[sn][i][sm][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[sn][ks][i], xa[ks][j], acc[sn][i][sm][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };
#define G8_BAR() __builtin_amdgcn_s_barrier()
#define G8_FENCE() __builtin_amdgcn_sched_barrier(0)
  // one K tile = four phases; ST1: tile t+1 exists (its A half 1 is staged here), ST2: tile t+2 exists
  auto ktile = [&](auto BUF, auto ST1, auto ST2, int t) {
    constexpr int B = decltype(BUF)::value;
    constexpr bool st1 = decltype(ST1)::value != 0, st2 = decltype(ST2)::value != 0;
    // q1
    read_B(ic<B>{}, ic<0>{});
    G8_FENCE();  // (issue order pinned: the lgkmcnt(8) below must retire exactly the four B reads)
    read_A(ic<B>{}, ic<0>{});
    if constexpr (st1) stage_A(ic<B ^ 1>{}, ic<1>{}, t + 1);
    G8_FENCE();
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // B-sub0 is in registers: B half 0 of this buffer may be re-staged in the NEXT phase
    G8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G8_FENCE();
    mma(ic<0>{}, ic<0>{});
    G8_FENCE();
    G8_BAR();
    // q2
    read_B(ic<B>{}, ic<1>{});
    if constexpr (st2) stage_B(ic<B>{}, ic<0>{}, t + 2);
    G8_FENCE();
    G8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G8_FENCE();
    mma(ic<0>{}, ic<1>{});
    G8_FENCE();
    G8_BAR();
    // q3
    read_A(ic<B>{}, ic<1>{});
    if constexpr (st2) stage_A(ic<B>{}, ic<0>{}, t + 2);
    G8_FENCE();
    G8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G8_FENCE();
    mma(ic<1>{}, ic<1>{});
    G8_FENCE();
    G8_BAR();
    // q4: the wait that retires tile t+1 (first read one phase later, behind two more barriers)
    if constexpr (st2) {
      stage_B(ic<B>{}, ic<1>{}, t + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if constexpr (st1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G8_FENCE();
    G8_BAR();
    G8_FENCE();
    mma(ic<1>{}, ic<0>{});
    G8_FENCE();
    G8_BAR();
  };

  const int nk = a.Kp / BK;  // even, >= 2
  auto prologue = [&]() {  // K tile 0 whole, K tile 1 without its A half 1 (the loop's q1 stages that): 14 DMA instructions per wave
    stage_B(ic<0>{}, ic<0>{}, 0);
    stage_A(ic<0>{}, ic<0>{}, 0);
    stage_B(ic<0>{}, ic<1>{}, 0);
    stage_A(ic<0>{}, ic<1>{}, 0);
    stage_B(ic<1>{}, ic<0>{}, 1);
    stage_A(ic<1>{}, ic<0>{}, 1);
    stage_B(ic<1>{}, ic<1>{}, 1);
  };
  prologue();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // K tile 0 has landed (this wave's part; the barrier below makes it everyone's)
  for (;;) {  // (one pass unless PERSIST)
#pragma unroll
  for (int i = 0; i < 32; ++i) (&acc[0][0][0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  G8_BAR();
  if (wr == 1) G8_BAR();  // the stagger: waves 4-7 run one barrier behind
  int t = (!PERSIST && (a.dbg & 4)) ? nk - 2 : 0;
  for (; t + 3 < nk; t += 2) {
    ktile(ic<0>{}, ic<1>{}, ic<1>{}, t);
    ktile(ic<1>{}, ic<1>{}, ic<1>{}, t + 1);
  }
  // LNF: a lane's accumulators cover pixel fr of eight pixel tiles pt = 4 sm + j, the same eight for the four lanes fg = 0..3 that share fr: lane fg requests
  // the (mean, rstd) pairs of tiles 2 fg and 2 fg + 1 only (4 registers across the last two K tiles instead of 16) and the epilogue fetches a tile's pair from
  // lane fr + 16 (pt >> 1) with two ds_bpermute
  float2 mrq[2];
  if constexpr (LNF) {
    const float2* const mr = reinterpret_cast<const float2*>(a.ps);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pt = 2 * (lane >> 4) + q;
      const int m = m0 + wr * 128 + (pt >> 2) * 64 + (pt & 3) * 16 + (lane & 15);
      mrq[q] = mr[m < a.M ? m : a.M - 1];
    }
  }
  auto row_stats = [&](int pt) -> float2 {  // (pt: compile-time after unrolling)
    if constexpr (LNF) {
      const int src = (lane & 15) + 16 * (pt >> 1);
      return make_float2(__shfl(mrq[pt & 1].x, src, 64), __shfl(mrq[pt & 1].y, src, 64));
    } else {
      return make_float2(0.f, 0.f);
    }
  };
  ktile(ic<0>{}, ic<1>{}, ic<0>{}, t);
  ktile(ic<1>{}, ic<0>{}, ic<0>{}, t + 1);
  if (wr == 0) G8_BAR();  // (both groups have executed the same number of barriers; nobody reads the K buffers any more)

  // ---- epilogue: lane (fr, fg) holds channels 4 fg .. 4 fg + 3 of channel tile (sn, i) for pixel fr of pixel tile (sm, j) ----
  const unsigned fl = a.flags;
  f32x4 bvv[2][2], gvv[2][2];
#pragma unroll
  for (int sn = 0; sn < 2; ++sn)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + wc * 64 + sn * 32 + i * 16 + fg * 4;
      bvv[sn][i] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      gvv[sn][i] = (EPI == EPI_RES && (fl & KPF_RES_GAMMA)) ? *reinterpret_cast<const f32x4*>(a.gamma + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (LNF) gvv[sn][i] = *reinterpret_cast<const f32x4*>(a.pt + n);  // (the layer-scale slot is free in this epilogue: s[n])
    }
  // y = acc + bias, or with the folded LayerNorm rstd * (acc - mean * s) + b'
  auto pre_act = [&](float acc_v, float b_v, float s_v, float2 mr_v) -> float {
    if constexpr (LNF) return fmaf(mr_v.y, fmaf(-mr_v.x, s_v, acc_v), b_v);
    else return acc_v + b_v;
  };
  const int cm0 = m0, cn0 = n0;  // the tile being finished
  bool more = false;
  if constexpr (PERSIST) {
    // the bias is in registers (hipcc waits vmcnt(0) for an ordinary load's result while a DMA is in flight, so it is consumed BEFORE the DMAs go out)
#pragma unroll
    for (int sn = 0; sn < 2; ++sn)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        asm volatile("" : "+v"(bvv[sn][i]));
        if constexpr (LNF) asm volatile("" : "+v"(gvv[sn][i]));
      }
    if constexpr (LNF) asm volatile("" : "+v"(mrq[0].x), "+v"(mrq[0].y), "+v"(mrq[1].x), "+v"(mrq[1].y));
    tile += (int)gridDim.x;
    more = tile < a.nblk;
    if (more) {
      set_tile(tile);
      prologue();  // flies under the epilogue below
    }
  }
  if constexpr (EPI == EPI_RES) {
    // residual layers (pwconv2): the skip rows are read in the accumulator layout, all rows of a pixel sub-tile requested before its first store
    TH* const ob = reinterpret_cast<TH*>(a.out);
    const TH* const rb = reinterpret_cast<const TH*>(a.res);
#pragma unroll
    for (int sm = 0; sm < 2; ++sm) {
      f32x4 rv[4][2][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long m = cm0 + wr * 128 + sm * 64 + j * 16 + fr;
        const long mr = m < a.M ? m : a.M - 1;
#pragma unroll
        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
          for (int i = 0; i < 2; ++i) rv[j][sn][i] = kpf_ld4(rb + mr * a.res_ld + a.res_coff + cn0 + wc * 64 + sn * 32 + i * 16 + fg * 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long m = cm0 + wr * 128 + sm * 64 + j * 16 + fr;
#pragma unroll
        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x4 v = acc[sn][i][sm][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float y = v[e] + bvv[sn][i][e];
              if (fl & KPF_RES_GAMMA) y *= gvv[sn][i][e];
              y += rv[j][sn][i][e];
              if (fl & KPF_RELU_AFTER_RES) y = fmaxf(y, 0.f);
              v[e] = y;
            }
            if (m < a.M) kpf_st4(ob + m * a.out_ld + a.out_coff + cn0 + wc * 64 + sn * 32 + i * 16 + fg * 4, v);
          }
      }
    }
  } else if constexpr (!PERSIST) {
    // linear / ReLU / GELU: results go through a per-wave LDS staging area ([128 pixels][64 channels], 144-byte rows) so that the global stores are whole
    // 128-byte rows (8 lanes x 16 bytes), 8 rows per instruction
    constexpr int RS = 72;  // staging row stride in elements
    TH* const stg = reinterpret_cast<TH*>(LB) + wave * 128 * RS;
#pragma unroll
    for (int sm = 0; sm < 2; ++sm)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float2 mrv = row_stats(4 * sm + j);
#pragma unroll
        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x4 v = acc[sn][i][sm][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y = pre_act(v[e], bvv[sn][i][e], gvv[sn][i][e], mrv);
              if (EPI == EPI_GELU) v[e] = (a.dbg & 1) ? y : gelu_of<ARITH>(y);
              else v[e] = (fl & KPF_ACT_RELU) ? fmaxf(y, 0.f) : ((fl & KPF_ACT_LEAKY) ? fmaxf(y, 0.01f * y) : y);
            }
            kpf_st4(stg + (sm * 64 + j * 16 + fr) * RS + sn * 32 + i * 16 + fg * 4, v);
          }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (own region only: no barrier)
    TH* const ob = reinterpret_cast<TH*>(a.out) + (long)(cm0 + wr * 128) * a.out_ld + a.out_coff + cn0 + wc * 64;
    const int prow = lane >> 3, pch = (lane & 7) * 8;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = r * 8 + prow;
      const f32x4 q = *reinterpret_cast<const f32x4*>(stg + px * RS + pch);
      if (cm0 + wr * 128 + px < a.M && !(a.dbg & 2)) g8_store16(ob + (long)px * a.out_ld + pch, q, a.st_policy);
    }
  } else {
    // persistent form: the K buffers already receive the next tile, so the results are staged 32 pixels at a time through the wave's 4 KB beyond them
    // ([32 pixels][64 channels], 128-byte rows, 16-byte chunk c of row r at c ^ (r & 7)); stores are whole 128-byte rows, exactly 16 per wave and tile
    // (M % 256 == 0: no predicate — the vmcnt(22) at the top of the tile loop counts on it)
    char* const stg = LB + 2 * G8_BUF + wave * 4096;
    TH* const ob = reinterpret_cast<TH*>(a.out) + (long)(cm0 + wr * 128) * a.out_ld + a.out_coff + cn0 + wc * 64;
    const int prow = lane >> 3, pcp = lane & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // chunk c = pixel tiles (sm, j) = (c >> 1, 2 (c & 1) + {0, 1})
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const float2 mrv = row_stats(4 * (c >> 1) + 2 * (c & 1) + jj);
#pragma unroll
        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x4 v = acc[sn][i][c >> 1][2 * (c & 1) + jj];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y = pre_act(v[e], bvv[sn][i][e], gvv[sn][i][e], mrv);
              if (EPI == EPI_GELU) v[e] = gelu_of<ARITH>(y);
              else v[e] = (fl & KPF_ACT_RELU) ? fmaxf(y, 0.f) : ((fl & KPF_ACT_LEAKY) ? fmaxf(y, 0.01f * y) : y);
            }
            const int row = jj * 16 + fr, cidx = sn * 4 + i * 2 + (fg >> 1);
            kpf_st4(reinterpret_cast<TH*>(stg + row * 128 + ((cidx ^ (row & 7)) << 4) + (fg & 1) * 8), v);
          }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r * 8 + prow;
        const f32x4 q = *reinterpret_cast<const f32x4*>(stg + row * 128 + (pcp << 4));
        g8_store16(ob + (long)(c * 32 + row) * a.out_ld + ((pcp ^ (row & 7)) << 3), q, a.st_policy);
      }
    }
  }
  if (!PERSIST || !more) break;
  // next tile: its 14 DMAs were issued BEFORE the 16 row stores of the epilogue above, so "all but the youngest 6 + 16" means "K tile 0 is in LDS"
  asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
  }  // tile loop
}

#undef G8_BAR
#undef G8_FENCE

template <int ARITH>
int launch_8ph(ConvArgs& a, hipStream_t st) {
  const bool res = a.flags & KPF_RES_ADD, gelu = a.flags & KPF_ACT_GELU;
  a.tilesN = a.N / 256;
  a.nblk = ((a.M + 255) / 256) * a.tilesN;
  static const bool no_persist = getenv("KPF_G8_NO_PERSIST") != nullptr;  // tuning aid
  // output stores of the persistent form (g8_store16 above): non-temporal by default (KPF_G8_ST = 0 plain, 1 sc1, 2 nt).  Measured per shape, f16, round 6
  // (profiles/r06_g16_store_policy.txt): FETCH of the K <= 256 GELU layers falls from 3.6 x / 2.4 x their operand bytes to 2.2 x / 1.07 x (sc1: 1.4 x / 1.06 x)
  // and their time by 5-8 % (sc1: equal); the K >= 512 shapes are unchanged either way (their 4.9 x is not output pollution: it does not move with the
  // policy nor with a start skew between the workgroups that share a pixel panel).
  static const int st_env = []() { const char* e = getenv("KPF_G8_ST"); return e ? atoi(e) : -1; }();

  const bool persist = !res && a.M % 256 == 0 && a.nblk > 256 && !no_persist && !a.dbg;
  a.st_policy = persist ? (st_env >= 0 ? st_env : 2) : 0;
  const bool lnf = (a.flags & KPF_PRO_LN) != 0;  // (kpf_conv2d_h16 admits it with the GELU epilogue only)
  void (*kern)(const ConvArgs) = res ? gemm16_8ph_kernel<EPI_RES, ARITH, false>
                                     : (gelu ? (lnf ? (persist ? gemm16_8ph_kernel<EPI_GELU, ARITH, true, true> : gemm16_8ph_kernel<EPI_GELU, ARITH, false, true>)
                                                    : (persist ? gemm16_8ph_kernel<EPI_GELU, ARITH, true> : gemm16_8ph_kernel<EPI_GELU, ARITH, false>))
                                             : (persist ? gemm16_8ph_kernel<EPI_LIN, ARITH, true> : gemm16_8ph_kernel<EPI_LIN, ARITH, false>));
  static std::atomic<bool> lds_opt_in[7][KPF_MAX_DEVICES];
  if (!kpf_raise_lds_limit(reinterpret_cast<const void*>(kern), lds_opt_in[res ? 4 : (gelu ? (lnf ? 5 : 2) : 0) + (persist ? 1 : 0)])) {
    kpf_set_error("kpf_conv2d_h16: cannot raise the dynamic LDS limit");
    return KPF_ELAUNCH;
  }
  // K buffers (128 KB); one-tile store epilogues stage 144 KB through them, the persistent form 32 KB beyond them (160 KB: the whole LDS of a CU)
  const size_t lds = res ? 2 * G8_BUF : (persist ? 2 * G8_BUF + 8 * 4096 : (size_t)8 * 128 * 72 * 2);
  hipLaunchKernelGGL(kern, dim3(persist ? 256 : a.nblk), dim3(512), lds, st, a);
  return kpf_check_launch("kpf_conv2d_h16");
}

// (round 6: gemm16_dfe_kernel — the 256 x 128 deferred-epilogue form of round 5, bit-identical to this kernel and measured equal or slower on every shape,
//  profiles/r05_dfe_ablation.txt — was removed from the build; `git show 2d01941:keypointfusion_amd/csrc/kpf_conv.hip` has it.)

// ---- 16-bit storage path (kpf_conv16.hip compiles this file with KPF_CONV_H16) ----
template <int TM, int TN, int WM, int WN, int ARITH, int NS>
int launch_arith_h16(ConvArgs& a, bool fast1x1, bool pointwise, hipStream_t st) {
  const bool res = a.flags & KPF_RES_ADD, gelu = a.flags & KPF_ACT_GELU;
  if (a.ps) {
    if (!pointwise || res || gelu) {
      kpf_set_error("kpf_conv2d_h16: the operand prologue is only supported for 1x1 stride-1 convolutions with a linear/ReLU epilogue");
      return KPF_EINVAL;
    }
    return fast1x1 ? launch_one<TM, TN, WM, WN, true, true, EPI_LIN, ARITH, 2>(a, st) : launch_one<TM, TN, WM, WN, false, true, EPI_LIN, ARITH, 2>(a, st);
  }
  if (gelu) {
    if (!pointwise || res) {
      kpf_set_error("kpf_conv2d_h16: GELU is only supported on 1x1 convolutions without residual");
      return KPF_EINVAL;
    }
    if (a.flags & KPF_ACT_GELU_SAVE)
      return fast1x1 ? launch_one<TM, TN, WM, WN, true, false, EPI_GELU2, ARITH, NS>(a, st) : launch_one<TM, TN, WM, WN, false, false, EPI_GELU2, ARITH, NS>(a, st);
    return fast1x1 ? launch_one<TM, TN, WM, WN, true, false, EPI_GELU, ARITH, NS>(a, st) : launch_one<TM, TN, WM, WN, false, false, EPI_GELU, ARITH, NS>(a, st);
  }
  if (a.flags & KPF_RES_GELU_GRAD) {
    if constexpr (NS == 2) {
      if (pointwise) return fast1x1 ? launch_one<TM, TN, WM, WN, true, false, EPI_GGRAD, ARITH, NS>(a, st) : launch_one<TM, TN, WM, WN, false, false, EPI_GGRAD, ARITH, NS>(a, st);
    }
    kpf_set_error("kpf_conv2d_h16: KPF_RES_GELU_GRAD needs a 1x1 stride-1 convolution on a two-stage tile");
    return KPF_EINVAL;
  }
  if (fast1x1) return res ? launch_one<TM, TN, WM, WN, true, false, EPI_RES, ARITH, NS>(a, st) : launch_one<TM, TN, WM, WN, true, false, EPI_LIN, ARITH, NS>(a, st);
  return res ? launch_one<TM, TN, WM, WN, false, false, EPI_RES, ARITH, NS>(a, st) : launch_one<TM, TN, WM, WN, false, false, EPI_LIN, ARITH, NS>(a, st);
}

template <int TM, int TN, int WM, int WN, int NS>
int launch_cfg_h16(ConvArgs& a, bool fast1x1, bool pointwise, int dtype, hipStream_t st) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  const int tilesM = (a.M + BM - 1) / BM;
  a.tilesN = (a.N + BN - 1) / BN;
  a.nblk = tilesM * a.tilesN;
  return dtype == KPF_DT_BF16 ? launch_arith_h16<TM, TN, WM, WN, ARITH_BF16, NS>(a, fast1x1, pointwise, st)
                              : launch_arith_h16<TM, TN, WM, WN, ARITH_F16, NS>(a, fast1x1, pointwise, st);
}

#endif

}  // namespace

#ifdef KPF_CONV_H16
// which launches take gemm16_8ph_kernel (one rule for the dispatcher and for callers that label their profiles: kpf_conv2d_h16_kernel)
static bool g8_applies(const kpf_conv_desc* d, bool has_prologue) {
  const unsigned fl = d->flags;
  const bool pointwise = d->KH == 1 && d->KW == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 && d->IH == d->OH && d->IW == d->OW;
  const bool fast1x1 = pointwise && d->Cin % 64 == 0 && d->Kp == d->Cin;
  const bool vec = d->out_ld % 4 == 0 && d->out_coff % 4 == 0 && (!(fl & KPF_RES_ADD) || (d->res_ld % 4 == 0 && d->res_coff % 4 == 0));
  // (the eight-phase kernel has no grouped form and none of the training step's two epilogues: part of the rule, so a forced
  //  KPF_FORCE_CFG16=30 cannot route such a launch to it either)
  const bool plain = d->groups <= 1 && !(fl & (KPF_RES_GELU_GRAD | KPF_ACT_GELU_SAVE));
  return plain && fast1x1 && d->Kp % 128 == 0 && d->N % 256 == 0 && !has_prologue && !(fl & KPF_OUT_NCHW) && vec && d->out_ld % 8 == 0 && d->out_coff % 8 == 0;
}
static bool g8_preferred(const kpf_conv_desc* d) {
  const long M = (long)d->B * d->OH * d->OW;
  return ((M + 255) / 256) * (d->N / 256) >= 224;
}
/* 1 when kpf_conv2d_h16 runs this descriptor on gemm16_8ph_kernel, 0 when on igemm_h16_kernel (profile labels; same rule as the dispatcher) */
extern "C" int kpf_conv2d_h16_uses_8ph(const kpf_conv_desc* d, int has_prologue) {
  static const bool no8 = getenv("KPF_NO_8PH") != nullptr;
  return d && !no8 && g8_applies(d, has_prologue != 0) && g8_preferred(d) ? 1 : 0;
}

/* 1 when kpf_conv2d_h16 accepts this descriptor with KPF_PRO_LN (a rule over the layer's shape only, never its batch: which arithmetic a sample gets must not
   depend on how many samples share its launch) */
extern "C" int kpf_conv2d_h16_ln_fold_supported(const kpf_conv_desc* d) {
  return d && (d->flags & KPF_ACT_GELU) && !(d->flags & (KPF_RES_ADD | KPF_ACT_GELU_SAVE | KPF_OUT_NCHW)) && g8_applies(d, false) ? 1 : 0;
}

extern "C" int kpf_conv2d_h16(const kpf_conv_desc* d, const void* in, const void* w, const float* bias, const float* pro_scale,
                              const float* pro_shift, const float* gamma, const void* res, void* out, int dtype, void* stream) {
  KPF_REQUIRE(d && in && w && out, "kpf_conv2d_h16: null pointer");
  KPF_REQUIRE(dtype == KPF_DT_BF16 || dtype == KPF_DT_F16, "kpf_conv2d_h16: dtype must be KPF_DT_BF16 or KPF_DT_F16");
  KPF_REQUIRE(d->B > 0 && d->OH > 0 && d->OW > 0 && d->N > 0 && d->Cin > 0, "kpf_conv2d_h16: empty shape");
  KPF_REQUIRE(d->Cin % 8 == 0 && d->in_ld % 8 == 0 && d->in_coff % 8 == 0,
              "kpf_conv2d_h16: Cin/in_ld/in_coff must be multiples of 8 elements (got %d/%d/%d)", d->Cin, d->in_ld, d->in_coff);
  KPF_REQUIRE(d->Kp % 64 == 0 && d->Kp >= d->KH * d->KW * d->Cin, "kpf_conv2d_h16: Kp=%d must be a multiple of 64 and >= K=%d", d->Kp,
              d->KH * d->KW * d->Cin);
  KPF_REQUIRE(kpf_aligned16(in) && kpf_aligned16(w) && kpf_aligned16(out), "kpf_conv2d_h16: pointers must be 16-byte aligned");
  const unsigned fl = d->flags;
  KPF_REQUIRE(!(fl & (KPF_IN_SPLIT | KPF_W_SPLIT | KPF_OUT_SPLIT)), "kpf_conv2d_h16: the split-operand flags belong to kpf_conv2d_f32");
  if (!(fl & KPF_OUT_NCHW))
    KPF_REQUIRE(d->out_coff >= 0 && d->out_coff + d->N <= d->out_ld, "kpf_conv2d_h16: bad output slice ld=%d coff=%d N=%d", d->out_ld,
                d->out_coff, d->N);
  if (fl & KPF_RES_ADD)
    KPF_REQUIRE(res && kpf_aligned16(res) && d->res_coff >= 0 && d->res_coff + d->N <= d->res_ld, "kpf_conv2d_h16: bad residual");
  if (fl & KPF_RES_GAMMA) KPF_REQUIRE(gamma && (fl & KPF_RES_ADD), "kpf_conv2d_h16: RES_GAMMA needs gamma and RES_ADD");
  KPF_REQUIRE((pro_scale == nullptr) == (pro_shift == nullptr), "kpf_conv2d_h16: prologue needs both scale and shift");
  if (fl & KPF_PRO_LN)  // folded LayerNorm: pro_scale = (mean, rstd) per pixel, pro_shift = s[n]; the eight-phase kernel's GELU epilogue only
    KPF_REQUIRE(pro_scale && pro_shift && bias && (fl & KPF_ACT_GELU) && !(fl & (KPF_RES_ADD | KPF_ACT_GELU_SAVE | KPF_OUT_NCHW)) && g8_applies(d, false) &&
                    reinterpret_cast<uintptr_t>(pro_scale) % 8 == 0 && kpf_aligned16(pro_shift) && kpf_aligned16(bias),
                "kpf_conv2d_h16: KPF_PRO_LN needs the row statistics, s[n] and b'[n], the GELU epilogue and a layer gemm16_8ph_kernel covers "
                "(dense 1x1, Cin == Kp, Kp %% 128 == 0, N %% 256 == 0: ask kpf_conv2d_h16_uses_8ph)");
  KPF_REQUIRE(((fl & KPF_ACT_RELU) != 0) + ((fl & KPF_ACT_GELU) != 0) + ((fl & KPF_ACT_LEAKY) != 0) <= 1, "kpf_conv2d_h16: one activation only");
  KPF_REQUIRE(!((fl & KPF_RES_ADD) && (fl & (KPF_ACT_RELU | KPF_ACT_GELU | KPF_ACT_LEAKY))), "kpf_conv2d_h16: activation before a residual add is not supported");
  KPF_REQUIRE(!(fl & KPF_RELU_AFTER_RES) || (fl & KPF_RES_ADD), "kpf_conv2d_h16: RELU_AFTER_RES needs RES_ADD");
  KPF_REQUIRE(!(fl & KPF_RES_GELU_GRAD) || ((fl & KPF_RES_ADD) && !(fl & (KPF_RES_GAMMA | KPF_RELU_AFTER_RES | KPF_OUT_NCHW))), "kpf_conv2d_h16: RES_GELU_GRAD goes with RES_ADD alone");
  KPF_REQUIRE(!(fl & KPF_ACT_GELU_SAVE) || ((fl & KPF_ACT_GELU) && !(fl & (KPF_RES_ADD | KPF_OUT_NCHW)) && res && kpf_aligned16(res) && d->res_coff >= 0 && d->res_coff + d->N <= d->res_ld),
              "kpf_conv2d_h16: ACT_GELU_SAVE goes with ACT_GELU and a second output buffer in the res slot (res_ld / res_coff describe it)");
  KPF_REQUIRE((long)d->B * d->OH * d->OW < (1l << 31) && (long)d->B * d->IH * d->IW < (1l << 31), "kpf_conv2d_h16: too many pixels");

  ConvArgs a;  // staging-side fields in 4-byte words (2 elements), output-side fields in elements (see ARITH_BF16 above)
  a.in = static_cast<const float*>(in); a.w = static_cast<const float*>(w); a.bias = bias; a.ps = pro_scale; a.pt = pro_shift; a.gamma = gamma;
  a.res = static_cast<const float*>(res); a.out = static_cast<float*>(out);
  a.M = d->B * d->OH * d->OW; a.N = d->N; a.K = d->KH * d->KW * d->Cin / 2; a.Kp = d->Kp / 2;
  a.IH = d->IH; a.IW = d->IW; a.Cin = d->Cin / 2; a.in_ld = d->in_ld / 2; a.in_coff = d->in_coff / 2;
  a.OH = d->OH; a.OW = d->OW; a.ohow = d->OH * d->OW; a.KH = d->KH; a.KW = d->KW;
  a.sh = d->sh; a.sw = d->sw; a.ph = d->ph; a.pw = d->pw;
  a.out_ld = d->out_ld; a.out_coff = d->out_coff; a.res_ld = d->res_ld; a.res_coff = d->res_coff;
  {
    static const float* zero_of_dev[KPF_MAX_DEVICES] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= KPF_MAX_DEVICES) dev = 0;
    if (!zero_of_dev[dev]) {
      void* p = nullptr;
      if (hipGetSymbolAddress(&p, HIP_SYMBOL(kpf_zero16)) != hipSuccess || !p) {
        kpf_set_error("kpf_conv2d_h16: cannot resolve the zero page");
        return KPF_ELAUNCH;
      }
      zero_of_dev[dev] = static_cast<const float*>(p);
    }
    a.zero = zero_of_dev[dev];
  }
  a.flags = fl; a.tilesN = 0; a.nblk = 0; a.w_unscale = 1.0f; a.skew = 0; a.st_policy = 0;
  a.vec = (d->out_ld % 4 == 0 && d->out_coff % 4 == 0 && (!(fl & KPF_RES_ADD) || (d->res_ld % 4 == 0 && d->res_coff % 4 == 0))) ? 1 : 0;
  a.groups = d->groups > 1 ? d->groups : 1; a.g_in = a.g_out = 0; a.g_w = 0;
  if (a.groups > 1) {  // grouped launch: see ConvArgs
    KPF_REQUIRE(!(fl & (KPF_RES_GAMMA | KPF_OUT_NCHW)) && !pro_scale && d->w_gstride % 8 == 0 && d->N % 2 == 0,
                "kpf_conv2d_h16: a grouped launch takes no layer scale / prologue / NCHW output, and needs w_gstride %% 8 == 0, N %% 2 == 0");
    KPF_REQUIRE(d->in_coff + d->groups * d->Cin <= d->in_ld && d->out_coff + d->groups * d->N <= d->out_ld, "kpf_conv2d_h16: the groups' channel slices exceed the pixel stride");
    KPF_REQUIRE(!(fl & (KPF_RES_ADD | KPF_ACT_GELU_SAVE)) || d->res_coff + d->groups * d->N <= d->res_ld,
                "kpf_conv2d_h16: the groups' residual / saved pre-activation slices exceed the pixel stride");
    a.g_in = d->Cin / 2; a.g_out = d->N; a.g_w = d->w_gstride / 2;
    if ((d->out_coff + d->N) % 8 || ((fl & KPF_RES_ADD) && (d->res_coff + d->N) % 8)) a.vec = 0;  // (the staged epilogue stores 16-byte pieces of every group's slice)
  }
  {
    static const int dbg = []() { const char* e = getenv("KPF_G8_DBG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
  }
  const bool pointwise = d->KH == 1 && d->KW == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 && d->IH == d->OH && d->IW == d->OW;
  const bool fast1x1 = pointwise && d->Cin % 64 == 0 && d->Kp == d->Cin;  // whole 64-element K tiles, no K mask
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  // Tile choice (tools/gemm16_bench.py on the ConvNeXt-B 512^2 shapes): the 16-bit GEMMs are bound by HBM traffic, staging and
  // barriers, not by the matrix pipe, and 128 x 128 tiles with several workgroups per CU beat 256 x 128 (one workgroup per CU) on
  // every heavy shape (65536 x 2048 x 512 + GELU: 585 vs 425 TFLOP/s); non-residual layers take the single-stage, 4-waves-per-SIMD
  // variant; residual layers with a long K take 256 x 128 with a 3-stage LDS ring (two K tiles of DMA in flight: 705 vs 667 at K = 2048)
  int best = 0;
  double bc = 1e30;
  static const int allowed[] = {0, 1, 2, 5, 6, 7};
  for (int i : allowed) {
    // groups not counted here: measured no change on the grouped 16-bit launches of the training step
    const double c = cfg_cost(kCfgs[i], a.M, a.N);
    if (c < bc) { bc = c; best = i; }
  }
  if (a.M >= 4096 && a.N >= 256 && a.N % 128 == 0) best = 0;  // (the round model over-rates the narrow tiles at these sizes)
  const bool occ = !(fl & KPF_RES_ADD) && !pro_scale;
  const bool plain_res = (fl & KPF_RES_ADD) && !(fl & KPF_RES_GELU_GRAD);  // (the GELU-gradient epilogue exists on the two-stage tiles only)
  if (plain_res && fast1x1 && a.Kp * 2 >= 2048 && a.M >= 32768 && a.N >= 256 && a.N % 128 == 0) best = 20;
  // 256 x 256 tiles, 8 waves of 128 x 64 (the geometry of cdna_hip_programming's 256^2 template, two-phase loop, s_setprio around the
  // MFMA cluster): +24 % over case 20 on 65536 x 512 x 2048 (841 vs 679 TFLOP/s), 1049 vs 935 on 16384 x 1024 x 4096
  if (plain_res && fast1x1 && a.Kp * 2 >= 2048 && a.M >= 16384 && a.N >= 256 && a.N % 256 == 0) best = 26;
  // Round 4: the eight-phase 256 x 256 kernel (gemm16_8ph_kernel) for every dense 1x1 layer it covers with at least one full round of tiles
  if (fl & KPF_PRO_LN) return dtype == KPF_DT_BF16 ? launch_8ph<ARITH_BF16>(a, st) : launch_8ph<ARITH_F16>(a, st);  // (no other kernel has that epilogue)
  const bool ok8 = g8_applies(d, pro_scale != nullptr);
  static const bool no8 = getenv("KPF_NO_8PH") != nullptr;  // tuning aid: A/B against the round-3 tile shapes
  if (ok8 && !no8 && g8_preferred(d)) best = 30;
  // Round 5: grids that do not fill the chip with a long K (B = 32 at 128 x 128: pwconv2 / 3x3 / concatenation layers of stages 3-4 and of the
  // low-resolution decoder levels).  With <= 2 workgroups per CU and two LDS stages the K loop runs at one DMA latency per K tile; a 4-stage ring of
  // 64 x 64 tiles (64 KB: two workgroups per CU, three K tiles in flight each) or, below 128 tiles, an 8-stage ring of 32 x 64 tiles shortens it
  // (tools/h16_small_sweep.py, graph replay: 512 x 768 x 3072 23.8 -> 13.3 us, 8192 x 192 x 1728 3x3 21.8 -> 16.9, 2048 x 384 x 1152 9.3 -> 7.7;
  // wide-N short-K layers lose and keep the two-stage tiles).  Same k order in every tile shape: results stay bit-identical.
  static const bool no_ring = getenv("KPF_NO_RING16") != nullptr;  // tuning aid
  if (!no_ring && best != 30 && !pro_scale && !(fl & KPF_RES_GELU_GRAD)) {
    const long nk = a.Kp / BK, b64 = (((long)a.M + 63) / 64) * (((long)a.N + 63) / 64) * a.groups;  // (a grouped launch has that many tiles per group:
                                                                                                     // the training step's paired backbones took the 32 x 64
                                                                                                     // form where their two groups fill the 64 x 64 one —
                                                                                                     // 512 x 768 x 3072 x 2: 25 -> 16 us)
    if (nk >= 6 && 4L * a.Kp >= a.N) {  // (Kp counts 4-byte words: K >= N / 2 elements — every measured winner; wide-N short-K layers are not)
      if (b64 <= 128) best = 44;
      else if (b64 <= 512) best = 41;
    }
  }
  // End of round 5 (per-shape search over every 16-bit launch of configs[4], tools/exp_autotune16_dump.py, the picks then timed in the step,
  // tools/exp_rules16.sh): the k x k convolutions (3 x 3 decoder layers, 2 x 2 downsampling) were left on 128 x 128 tiles; with one round of 256 x 256
  // (else 256 x 128) tiles available they run 33-58 % faster alone — 16384 x 512 x 4608: 187 -> 79 us, 262144 x 256 x 512 (2 x 2): 219 -> 122,
  // 65536 x 256 x 2304: 131 -> 80, 262144 x 128 x 1152: 152 -> 104 — and the 64-channel 3 x 3 layers take 128 x 64 (146 -> 127).  Step: 42.6 -> 41.3 ms.
  static const bool no_kxk = getenv("KPF_NO_KXK16") != nullptr;  // tuning aid
  if (!no_kxk && !pointwise && !pro_scale && a.groups <= 1 && !(fl & (KPF_RES_GELU_GRAD | KPF_ACT_GELU_SAVE | KPF_OUT_NCHW)) && a.M >= 16384) {
    const long tm256 = ((long)a.M + 255) / 256;
    if (a.N % 256 == 0 && tm256 * (a.N / 256) >= 256) best = 26;
    else if (a.N % 128 == 0 && tm256 * (a.N / 128) >= 256) best = 8;
    else if (a.N == 64) best = 2;
  }
  static const int forced = []() { const char* e = getenv("KPF_FORCE_CFG16"); return e ? atoi(e) : -1; }();  // tuning aid only
  if (forced >= 0 && (forced != 30 || ok8)) best = forced;
  if (d->tile_cfg > 0 && (d->tile_cfg != 31 || ok8)) best = d->tile_cfg - 1;  // the caller's choice (tools/h16_small_sweep.py): case index + 1
  if (best == 30) return dtype == KPF_DT_BF16 ? launch_8ph<ARITH_BF16>(a, st) : launch_8ph<ARITH_F16>(a, st);
#ifdef KPF_FAST_BUILD  // tuning aid: one tile shape only (asm inspection / quick syntax builds), never shipped
  return launch_cfg_h16<4, 4, 2, 2, 2>(a, fast1x1, pointwise, dtype, st);
#endif
  switch (best) {
    case 0: return occ ? launch_cfg_h16<4, 4, 2, 2, 1>(a, fast1x1, pointwise, dtype, st) : launch_cfg_h16<4, 4, 2, 2, 2>(a, fast1x1, pointwise, dtype, st);
    case 1: return launch_cfg_h16<4, 3, 2, 2, 2>(a, fast1x1, pointwise, dtype, st);  // 128 x 96
    case 2: return launch_cfg_h16<2, 4, 4, 1, 2>(a, fast1x1, pointwise, dtype, st);  // 128 x 64
    case 5: return launch_cfg_h16<2, 4, 2, 2, 2>(a, fast1x1, pointwise, dtype, st);  // 64 x 128
    case 6: return occ ? launch_cfg_h16<2, 2, 2, 2, 1>(a, fast1x1, pointwise, dtype, st) : launch_cfg_h16<2, 2, 2, 2, 2>(a, fast1x1, pointwise, dtype, st);
    case 8: return launch_cfg_h16<4, 4, 4, 2, 2>(a, fast1x1, pointwise, dtype, st);  // 256 x 128
    case 20: return launch_cfg_h16<4, 4, 4, 2, 3>(a, fast1x1, pointwise, dtype, st);  // 256 x 128, 3-stage LDS ring (144 KB): two K tiles of DMA in flight
    case 21: return launch_cfg_h16<4, 4, 2, 2, 3>(a, fast1x1, pointwise, dtype, st);  // 128 x 128, 3-stage ring (96 KB)
    case 22: return launch_cfg_h16<4, 4, 2, 2, 4>(a, fast1x1, pointwise, dtype, st);  // 128 x 128, 4-stage ring (128 KB)
    case 26: return launch_cfg_h16<8, 4, 2, 4, 2>(a, fast1x1, pointwise, dtype, st);  // 256 x 256, 8 waves of 128 x 64 (2 per SIMD), 128 KB of LDS
    // round 5: LDS rings for launches whose grid does not fill the chip (selection rule above).  Also measured and not kept: 64 x 64 with an 8-stage
    // ring (128 KB, one workgroup per CU: 57 us where the 4-stage form takes 10), 64 x 128 with 3 / 6 stages, 128 x 64 with 6 (all slower than the two
    // below or than the two-stage tiles on every shape of tools/h16_small_sweep.py).
    case 41: return launch_cfg_h16<2, 2, 2, 2, 4>(a, fast1x1, pointwise, dtype, st);  // 64 x 64, 4-stage ring (64 KB: two per CU)
    case 44: return launch_cfg_h16<2, 1, 1, 4, 8>(a, fast1x1, pointwise, dtype, st);  // 32 x 64, 8-stage ring (96 KB)
    default: return launch_cfg_h16<2, 1, 1, 4, 2>(a, fast1x1, pointwise, dtype, st);  // 32 x 64
  }
}
#else
extern "C" int kpf_conv2d_f32(const kpf_conv_desc* d, const float* in, const float* w, const float* bias,
                              const float* pro_scale, const float* pro_shift, const float* gamma, const float* res,
                              float* out, void* stream) {
  KPF_REQUIRE(d && in && w && out, "kpf_conv2d_f32: null pointer");
  KPF_REQUIRE(d->B > 0 && d->OH > 0 && d->OW > 0 && d->N > 0 && d->Cin > 0, "kpf_conv2d_f32: empty shape");
  KPF_REQUIRE(d->Cin % 4 == 0 && d->in_ld % 4 == 0 && d->in_coff % 4 == 0,
              "kpf_conv2d_f32: Cin/in_ld/in_coff must be multiples of 4 (got %d/%d/%d)", d->Cin, d->in_ld, d->in_coff);
  KPF_REQUIRE(d->Kp % 32 == 0 && d->Kp >= d->KH * d->KW * d->Cin, "kpf_conv2d_f32: Kp=%d must be a multiple of 32 and >= K=%d",
              d->Kp, d->KH * d->KW * d->Cin);
  KPF_REQUIRE(kpf_aligned16(in) && kpf_aligned16(w) && kpf_aligned16(out), "kpf_conv2d_f32: pointers must be 16-byte aligned");
  const unsigned fl = d->flags;
  if (!(fl & KPF_OUT_NCHW))
    KPF_REQUIRE(d->out_coff >= 0 && d->out_coff + d->N <= d->out_ld, "kpf_conv2d_f32: bad output slice ld=%d coff=%d N=%d",
                d->out_ld, d->out_coff, d->N);
  if (fl & KPF_RES_ADD)
    KPF_REQUIRE(res && kpf_aligned16(res) && d->res_coff >= 0 && d->res_coff + d->N <= d->res_ld, "kpf_conv2d_f32: bad residual");
  if (fl & KPF_RES_GAMMA) KPF_REQUIRE(gamma && (fl & KPF_RES_ADD), "kpf_conv2d_f32: RES_GAMMA needs gamma and RES_ADD");
  KPF_REQUIRE((pro_scale == nullptr) == (pro_shift == nullptr), "kpf_conv2d_f32: prologue needs both scale and shift");
  KPF_REQUIRE(((fl & KPF_ACT_RELU) != 0) + ((fl & KPF_ACT_GELU) != 0) + ((fl & KPF_ACT_LEAKY) != 0) <= 1, "kpf_conv2d_f32: one activation only");
  KPF_REQUIRE(!((fl & KPF_RES_ADD) && (fl & (KPF_ACT_RELU | KPF_ACT_GELU | KPF_ACT_LEAKY))), "kpf_conv2d_f32: activation before a residual add is not supported");
  KPF_REQUIRE(!((fl & KPF_ACT_LEAKY) && pro_scale), "kpf_conv2d_f32: LeakyReLU is not combined with an operand prologue");
  KPF_REQUIRE(!(fl & KPF_RELU_AFTER_RES) || (fl & KPF_RES_ADD), "kpf_conv2d_f32: RELU_AFTER_RES needs RES_ADD");
  KPF_REQUIRE(!(fl & KPF_RES_GELU_GRAD) || ((fl & KPF_RES_ADD) && !(fl & (KPF_RES_GAMMA | KPF_RELU_AFTER_RES | KPF_OUT_NCHW))), "kpf_conv2d_f32: RES_GELU_GRAD goes with RES_ADD alone");
  KPF_REQUIRE(!(fl & KPF_ACT_GELU_SAVE) || ((fl & KPF_ACT_GELU) && !(fl & (KPF_RES_ADD | KPF_OUT_NCHW)) && res && kpf_aligned16(res) && d->res_coff >= 0 && d->res_coff + d->N <= d->res_ld),
              "kpf_conv2d_f32: ACT_GELU_SAVE goes with ACT_GELU and a second output buffer in the res slot (res_ld / res_coff describe it)");
  KPF_REQUIRE((long)d->B * d->OH * d->OW < (1l << 31) && (long)d->B * d->IH * d->IW < (1l << 31), "kpf_conv2d_f32: too many pixels");

  ConvArgs a;
  a.in = in; a.w = w; a.bias = bias; a.ps = pro_scale; a.pt = pro_shift; a.gamma = gamma; a.res = res; a.out = out;
  a.M = d->B * d->OH * d->OW; a.N = d->N; a.K = d->KH * d->KW * d->Cin; a.Kp = d->Kp;
  a.IH = d->IH; a.IW = d->IW; a.Cin = d->Cin; a.in_ld = d->in_ld; a.in_coff = d->in_coff;
  a.OH = d->OH; a.OW = d->OW; a.ohow = d->OH * d->OW; a.KH = d->KH; a.KW = d->KW;
  a.sh = d->sh; a.sw = d->sw; a.ph = d->ph; a.pw = d->pw;
  a.out_ld = d->out_ld; a.out_coff = d->out_coff; a.res_ld = d->res_ld; a.res_coff = d->res_coff;
  {  // address of the zero page on the current device (one symbol lookup per device per process)
    static const float* zero_of_dev[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!zero_of_dev[dev]) {
      void* p = nullptr;
      if (hipGetSymbolAddress(&p, HIP_SYMBOL(kpf_zero16)) != hipSuccess || !p) {
        kpf_set_error("kpf_conv2d_f32: cannot resolve the zero page");
        return KPF_ELAUNCH;
      }
      zero_of_dev[dev] = static_cast<const float*>(p);
    }
    a.zero = zero_of_dev[dev];
  }
  a.flags = fl; a.tilesN = 0; a.nblk = 0; a.dbg = 0; a.skew = 0; a.st_policy = 0;
  {  // tuning aid KPF_F32_ST_MB: outputs of at least that many MB are written with non-temporal stores (0 = never)
    static const int st_mb = []() { const char* e = getenv("KPF_F32_ST_MB"); return e ? atoi(e) : 0; }();
    if (st_mb > 0 && (long)d->B * d->OH * d->OW * d->N * 4 >= (long)st_mb * 1000000) a.st_policy = 2;
  }
  a.w_unscale = (fl & (KPF_IN_SPLIT | KPF_W_SPLIT)) ? d->w_unscale : 1.0f;
  if (fl & (KPF_IN_SPLIT | KPF_W_SPLIT))
    KPF_REQUIRE(d->w_unscale > 0.f && d->Cin % 32 == 0 && d->in_coff % 4 == 0, "kpf_conv2d_f32: split operands need w_unscale > 0 and Cin %% 32 == 0 (Cin=%d)", d->Cin);
  if (fl & KPF_IN_SPLIT) KPF_REQUIRE(d->in_coff % 32 == 0 && d->in_ld % 32 == 0 && !pro_scale, "kpf_conv2d_f32: split activations need in_ld, in_coff multiples of 32 and no operand prologue");
  if (fl & KPF_OUT_SPLIT)
    KPF_REQUIRE(d->N % 32 == 0 && d->out_coff % 32 == 0 && d->out_ld % 32 == 0 && !(fl & KPF_OUT_NCHW),
                "kpf_conv2d_f32: split output needs N, out_ld, out_coff multiples of 32 (N=%d ld=%d coff=%d)", d->N, d->out_ld, d->out_coff);
  a.vec = (d->out_ld % 4 == 0 && d->out_coff % 4 == 0 && (!(fl & KPF_RES_ADD) || (d->res_ld % 4 == 0 && d->res_coff % 4 == 0))) ? 1 : 0;
  a.groups = d->groups > 1 ? d->groups : 1; a.g_in = a.g_out = 0; a.g_w = 0;
  if (a.groups > 1) {  // grouped launch: see ConvArgs
    KPF_REQUIRE(!(fl & (KPF_RES_GAMMA | KPF_OUT_NCHW | KPF_IN_SPLIT | KPF_W_SPLIT | KPF_OUT_SPLIT)) && !pro_scale && d->w_gstride % 4 == 0,
                "kpf_conv2d_f32: a grouped launch takes no layer scale / prologue / NCHW output / split operands, and needs w_gstride %% 4 == 0");
    KPF_REQUIRE(d->in_coff + d->groups * d->Cin <= d->in_ld && d->out_coff + d->groups * d->N <= d->out_ld, "kpf_conv2d_f32: the groups' channel slices exceed the pixel stride");
    KPF_REQUIRE(!(fl & (KPF_RES_ADD | KPF_ACT_GELU_SAVE)) || d->res_coff + d->groups * d->N <= d->res_ld,
                "kpf_conv2d_f32: the groups' residual / saved pre-activation slices exceed the pixel stride");
    a.g_in = d->Cin; a.g_out = d->N; a.g_w = d->w_gstride;
    if ((d->out_coff + d->N) % 4 || ((fl & KPF_RES_ADD) && (d->res_coff + d->N) % 4)) a.vec = 0;
  }
  // the dense-1x1 fast path also needs whole K tiles (its staging reads 32 channels at a time without a K mask)
  const bool is1x1 = d->KH == 1 && d->KW == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 &&
                     d->IH == d->OH && d->IW == d->OW && d->Cin % 32 == 0 && d->Kp == d->Cin;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  int best = 0;
  double bc = 1e30;
  static const int forced = []() { const char* e = getenv("KPF_FORCE_CFG"); return e ? atoi(e) : -1; }();  // tuning aid only
  for (int i = 0; i < (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); ++i) {
    const double c = cfg_cost(kCfgs[i], a.M, a.N, a.groups);
    if (c < bc) { bc = c; best = i; }
  }
  // f32 arithmetic, 128 x 192 tiles (80 KB of LDS: exactly two workgroups per CU): taken when they make the tile count a whole number of
  // 512-workgroup rounds and the cost model's choice does not (stage-4 pwconv1: 4096 x 3072 -> 512 tiles instead of 768 of 128 x 128,
  // which the dispatcher spreads 2..4 per CU)
  const bool f32_arith = !(fl & (KPF_IN_SPLIT | KPF_W_SPLIT));
  const bool gelu1x1 = f32_arith && is1x1 && (fl & KPF_ACT_GELU) && !(fl & KPF_ACT_GELU_SAVE) && !pro_scale && d->groups <= 1;
  // (The round rule that forced 128 x 192 tiles wherever they make whole 512-workgroup rounds is gone: the per-shape search found it 13-40 % slower than the cost
  //  model's own choice on the layers it caught — 16384 x 1536 x 384 + GELU 174 against 152 us, the training step's 32768 x 384 x 96 data gradient 116 against
  //  70 — and in the overlapped step the tile loses even where it wins alone.  The configuration stays selectable: tile_cfg = 18.)
  // Round 5 (per-shape search over every launch of the headline, tools/exp_autotune_dump.py, then the rule sets timed in the overlapped step,
  // tools/exp_rules.sh): (i) the wide GELU layers (pwconv1: N = 4C) choose between 128 x 128 and the 256 x 128 eight-wave tile by whole rounds of the chip —
  // 16384 x 1536 x 384: 152 us against 174 for the 128 x 192 tile the round rule above used to force, 4096 x 3072 x 768: 157 against 168, 65536 x 768 x 192:
  // 175 against 183; (ii) a long-K layer whose best tile leaves the chip under one round of workgroups takes 32 x 64 tiles when they make two rounds
  // (4096 x 384 x 3456 3x3: 117 against 135 us; 4096 x 384 x 768: 30 against 33).  Step: 18.73 -> 18.25 ms.  Same k order in every tile shape: same bits.
  if (gelu1x1 && a.M >= 1024 && a.N >= 512) {
    // whole-CU-round cost of the two wide tiles; ties go to the smaller one
    auto rounds_cost = [&](int bm, int bn, double pen) { return (double)((((long)(a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) + 255) / 256) * bm * bn * pen; };
    // (128 x 192 is left out on purpose: alone it wins where it makes whole rounds — 4096 x 1536 x 384: 45 us against 58, 16384 x 384 x 1536 + residual: 147
    //  against 153 — and in the overlapped step it loses every time it was tried: its 80 KB of LDS leave no room for the other stream's workgroups;
    //  headline 3471 against 3508 img/s with it on the stage-3 residual layer, full128 9 800 against 9 970 with it on pwconv1)
    const double c0 = rounds_cost(128, 128, 1.00), c8 = rounds_cost(256, 128, 0.97);
    best = c0 <= c8 ? 0 : 8;
  }
  if (f32_arith && best < 9 && a.Kp >= 512) {
    const long bb = ((a.M + kCfgs[best].bm - 1) / kCfgs[best].bm) * ((a.N + kCfgs[best].bn - 1) / kCfgs[best].bn) * a.groups;
    if (bb < 256 && (long)((a.M + 31) / 32) * ((a.N + 63) / 64) * a.groups >= 512) best = 7;
  }
  // split arithmetic without a residual epilogue: the single-stage, 4-waves-per-SIMD variant of the 128 x 128 tile is 4-10 % faster
  // than the double-buffered 128 x 128 / 256 x 128 ones (its residual epilogue would spill at 128 registers, so those keep two stages)
  if ((fl & (KPF_IN_SPLIT | KPF_W_SPLIT)) && !(fl & KPF_RES_ADD) && !pro_scale && (best == 0 || best == 8)) best = 13;
  // split arithmetic, few tiles and a long K (stage-4 layers, 3x3 convolutions on small maps): the cost model's large tile leaves
  // at most two workgroups per CU running ~50-100 K tiles each; 64 x 64 single-stage tiles (4-5 per CU) are 10-25 % faster there
  if ((fl & (KPF_IN_SPLIT | KPF_W_SPLIT)) && !pro_scale && a.Kp >= 1024) {
    const long blocks = ((a.M + kCfgs[best < 9 ? best : 0].bm - 1) / kCfgs[best < 9 ? best : 0].bm) *
                        ((a.N + kCfgs[best < 9 ? best : 0].bn - 1) / kCfgs[best < 9 ? best : 0].bn);
    const long blocks64 = ((a.M + 63) / 64) * ((a.N + 63) / 64);
    if (best < 9 && blocks <= 512 && blocks64 >= 512) best = 16;
  }
  if (forced >= 0) best = forced;
  if (d->tile_cfg > 0) {  // the caller's choice (engine autotuning): tile_cfg = configuration index + 1
    KPF_REQUIRE(d->tile_cfg <= KPF_NUM_TILE_CFGS, "kpf_conv2d_f32: tile_cfg %d out of range", d->tile_cfg);
    best = d->tile_cfg - 1;
  }
#ifdef KPF_FAST_BUILD  // tuning aid: one tile shape only (asm inspection / quick A-B builds), never shipped
  return launch_cfg<4, 4, 2, 2>(a, is1x1, st);
#else
  switch (best) {
    case 0: return launch_cfg<4, 4, 2, 2>(a, is1x1, st);   // 128 x 128
    case 1: return launch_cfg<4, 3, 2, 2>(a, is1x1, st);   // 128 x 96
    case 2: return launch_cfg<2, 4, 4, 1>(a, is1x1, st);   // 128 x 64
    case 3: return launch_cfg<4, 3, 4, 1>(a, is1x1, st);   // 256 x 48
    case 4: return launch_cfg<2, 7, 4, 1>(a, is1x1, st);   // 128 x 112
    case 5: return launch_cfg<2, 4, 2, 2>(a, is1x1, st);   // 64 x 128
    case 6: return launch_cfg<2, 2, 2, 2>(a, is1x1, st);   // 64 x 64
    case 8: return launch_cfg<4, 4, 4, 2>(a, is1x1, st);   // 256 x 128, 8 MFMA waves
    case 9: return launch_cfg<4, 4, 4, 2, 3>(a, is1x1, st);   // split: 256 x 128, 3-stage ring (144 KB, 2 tiles in flight)
    case 10: return launch_cfg<4, 4, 2, 2>(a, is1x1, st);     // (was: 128 x 128 4-stage ring — one workgroup per CU, 1.5x slower; retired)
    case 11: return launch_cfg<4, 3, 2, 2>(a, is1x1, st);     // (was: 128 x 96 4-stage ring; retired)
    case 12: return launch_cfg<2, 4, 4, 1, 3>(a, is1x1, st);  // split: 128 x 64, 3-stage ring (72 KB, two workgroups per CU)
    case 13: return launch_cfg<4, 4, 2, 2, 1>(a, is1x1, st);  // split: 128 x 128, one LDS stage (32 KB), 4 waves per SIMD
    case 14: return launch_cfg<4, 3, 2, 2, 1>(a, is1x1, st);  // split: 128 x 96, one LDS stage (28 KB)
    case 15: return launch_cfg<2, 4, 4, 1, 1>(a, is1x1, st);  // split: 128 x 64, one LDS stage (24 KB)
    case 16: return launch_cfg<2, 2, 2, 2, 1>(a, is1x1, st);  // split: 64 x 64, one LDS stage (16 KB)
    case 17: return launch_cfg<4, 6, 2, 2>(a, is1x1, st);     // 128 x 192 (f32: 80 KB of LDS, two workgroups fill a CU's 160 KB)
    default: return launch_cfg<2, 1, 1, 4>(a, is1x1, st);  // 32 x 64
  }
#endif
}

extern "C" int kpf_conv_num_tile_cfgs(void) { return KPF_NUM_TILE_CFGS; }
#endif  // KPF_CONV_H16

#ifdef KPF_DBG_TIME  // (build ONE of the two translation units with it: the stamp buffer is a plain global)
extern "C" int kpf_dbg_read(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(kpf_dbg_t), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
extern "C" int kpf_dbg_clear(void) {
  static unsigned long long z[8 * 8192];
  return hipMemcpyToSymbol(HIP_SYMBOL(kpf_dbg_t), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
