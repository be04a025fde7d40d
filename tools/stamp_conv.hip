// Implicit-GEMM convolution / linear layer on the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32).
//
// GEMM view:  out[m][n] = sum_k A[m][k] * W[n][k],  m = output pixel (b,oy,ox), n = output channel,
//             k = (ky,kx,c) with the input channel c fastest (activations are NHWC, so k-runs are contiguous).
//
// MI355X mapping
//  * One workgroup = 4 waves (256 threads) computes a BM x BN tile, BM = 16*TM*WM pixels, BN = 16*TN*WN channels;
//    each wave owns TM x TN accumulator tiles of 16x16 (TM*TN*4 VGPRs).  f32 MFMA runs at the f32 vector rate
//    (157 TF peak), i.e. 16x slower than bf16 MFMA, so the kernel is MFMA-issue bound by construction and the design
//    goal is simply to never starve the matrix pipe: operands are staged global -> registers -> LDS one K-tile
//    ahead (the loads fly under the current tile's MFMAs), and 2-3 workgroups per CU cover barrier bubbles and the
//    epilogue VALU work (GELU) of their neighbours.
//  * The MFMA is issued "transposed": A-operand = weight fragment W[n][k], B-operand = activation fragment X[m][k], so
//    a lane's 4 accumulator registers are 4 *consecutive output channels* of one pixel -> the epilogue reads bias /
//    gamma / residual and writes the result as float4 (16 B per lane, 64 B contiguous per pixel per instruction).
//  * LDS images are [row][32 k] fp32 with the 16-byte chunk index XOR-swizzled by (row & 7): ds_write_b128 when
//    staging and ds_read_b128 when building fragments are both bank-conflict free; one ds_read_b128 feeds 4 MFMAs
//    (the K order inside a 16-deep step is permuted identically for both operands, which a dot product allows).
//  * Workgroup ids are remapped so that the 8 XCDs (private L2s) each get a contiguous range of tiles, channel tiles
//    fastest: the tiles that re-read one activation panel run on one XCD back to back.
//  * Prologue: eval-BatchNorm + ReLU of the *input* (pre-activation Residual, model/hourglass.py:106-108) is applied
//    in registers between the global load and the LDS store; zero padding stays zero.
//    Epilogue: bias, ReLU / GELU(erf), layer-scale * y + residual, ReLU-after-add, NHWC slice or NCHW store.
#include "../keypointfusion_amd/csrc/kpf_common.h"
#include <stdlib.h>

__device__ unsigned long long kpf_stamps[8];
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
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
  unsigned flags;
  int tilesN, nblk;
  int vec;  // 1: output/residual rows are 16-byte aligned -> float4 epilogue
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;

__device__ __attribute__((aligned(16))) float kpf_zero16[4] = {0.f, 0.f, 0.f, 0.f};  // source of every padding / out-of-range chunk

constexpr int BK = 32;  // K-tile depth: 8 chunks of 16 B per staged row
constexpr int NB = 3;   // LDS ring depth (loader runs 2 tiles ahead)

template <int TM, int TN, int WM, int WN, bool IS1X1, bool HAS_PRO>
__global__ __launch_bounds__(64 * (WM * WN + 1)) void igemm_f32_kernel(const ConvArgs a) {
  constexpr int BM = 16 * TM * WM;
  constexpr int BN = 16 * TN * WN;
  constexpr int AQ = BM / 8;  // DMA instructions per A tile: each moves 8 rows x 128 B = 1 KiB
  constexpr int BQ = BN / 8;
  constexpr int TILE = (BM + BN) * BK;
  constexpr int NW = WM * WN;   // MFMA waves (4 or 8); wave NW is the loader
  constexpr int ND = AQ + BQ;   // DMA instructions per K tile
  static_assert(NW == 4 || NW == 8, "4 or 8 MFMA waves per workgroup");
  static_assert(BM % 8 == 0 && BN % 8 == 0 && ND <= 63, "tile granularity / vmcnt range");

  extern __shared__ __attribute__((aligned(16))) float lds[];  // [NB][TILE] ring (+ [2][Kp] operand prologue scale/shift)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;  // 0..NW-1: MFMA waves, NW: loader wave

  // XCD-aware bijective remap: blocks b, b+8, b+16.. share an XCD -> give each XCD a contiguous range of logical tiles.
  int bid = blockIdx.x;
  {
    const int nb = a.nblk, q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int nt = bid % a.tilesN, mt = bid / a.tilesN;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = a.Kp / BK;
  float* pro_s = lds + NB * TILE;  // [Kp] scale, then [Kp] shift (zero beyond Cin: padded k contributes relu(0*x+0) = 0)
  float* pro_t = pro_s + a.Kp;

  if (HAS_PRO) {
    for (int i = tid; i < a.Kp; i += 64 * (NW + 1)) {
      pro_s[i] = i < a.Cin ? a.ps[i] : 0.f;
      pro_t[i] = i < a.Cin ? a.pt[i] : 0.f;
    }
  }

  if (wave == NW) {
    // ---------------------------------------------------------------------------------------------------------------
    // Loader wave: streams K tiles global -> LDS with global_load_lds_dwordx4 (no VGPR round trip), one tile ahead of the
    // MFMA waves, into the buffer they are not reading.  A wave's VMEM issue is paced by the CU's address unit (64 B/clk,
    // ~500 cycles per 32 KB tile); giving that job to its own wave keeps those stalls out of the MFMA waves' instruction
    // streams (measured: 811 of 5368 cycles per K tile when each MFMA wave issued its own share).
    // LDS rows are [32 k] fp32 = 8 chunks of 16 B, chunk c of row r stored at position c ^ (r & 7) (conflict-free
    // ds_read_b128 fragments).  The DMA writes lane l at base + 16*l, so LDS stays linear and the swizzle is applied to the
    // SOURCE: the lane at position (r, c') fetches logical chunk c' ^ (r & 7).
    // ---------------------------------------------------------------------------------------------------------------
    const int rr = lane >> 3;                           // row within the 8-row group of one DMA
    const int kc = (((lane & 7) ^ (rr & 7)) << 2);      // logical k offset (floats) this lane fetches within the K tile
    int rbase[AQ], riy[AQ], rix[AQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
      const int m = m0 + 8 * q + rr;
      if (m < a.M) {
        if (IS1X1) {
          rbase[q] = m;
          riy[q] = 0;
          rix[q] = 0;
        } else {
          const int b = m / a.ohow;
          const int r = m - b * a.ohow;
          const int oy = r / a.OW;
          const int ox = r - oy * a.OW;
          rbase[q] = b * a.IH * a.IW;
          riy[q] = oy * a.sh - a.ph;
          rix[q] = ox * a.sw - a.pw;
        }
      } else {
        rbase[q] = -1;
        riy[q] = 0;
        rix[q] = 0;
      }
    }
    auto stage = [&](int kt, int buf) {
      float* As = lds + buf * TILE;
      float* Bs = As + BM * BK;
      const int k = kt * BK + kc;
      const bool kvalid = k < a.K;
      int c = k, ky = 0, kx = 0;
      if (!IS1X1) {
        const int tap = k / a.Cin;
        c = k - tap * a.Cin;
        ky = tap / a.KW;
        kx = tap - ky * a.KW;
      }
#pragma unroll
      for (int q = 0; q < AQ; ++q) {
        bool v = kvalid && rbase[q] >= 0;
        long off;
        if (IS1X1) {
          off = (long)rbase[q] * a.in_ld + a.in_coff + c;
        } else {
          const int iy = riy[q] + ky, ix = rix[q] + kx;
          v = v && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
          off = ((long)rbase[q] + (long)iy * a.IW + ix) * a.in_ld + a.in_coff + c;
        }
        const float* src = v ? a.in + off : a.zero;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(As + 8 * q * BK), 16, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < BQ; ++q) {
        const int n = n0 + 8 * q + rr;
        const float* src = (n < a.N) ? a.w + ((long)n * a.Kp + k) : a.zero;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(Bs + 8 * q * BK), 16, 0, 0);
      }
    };
    // Ring of NB = 3 buffers, the loader runs two tiles ahead.  Only counted waits: `s_waitcnt vmcnt(ND)` = "all but the
    // youngest tile's DMAs have landed"; raw s_barrier (a __syncthreads() would drain vmcnt to 0 and serialise the ring).
    stage(0, 0);
    if (nk > 1) {
      stage(1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // B0: tile 0 (and the prologue table) visible to the MFMA waves
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 2 < nk) {
        stage(kt + 2, (kt + 2) % NB);  // that buffer held tile kt-1: released by the barrier that ended iteration kt-1
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");  // tile kt+1 landed, tile kt+2 in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // -----------------------------------------------------------------------------------------------------------------
  // MFMA waves: pure ds_read_b128 + v_mfma loop, one barrier per K tile.
  // -----------------------------------------------------------------------------------------------------------------
  const int wm = wave % WM, wn = wave / WM;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = zero4;

  const int fr = lane & 15;  // fragment row (pixel for X, channel for W)
  const int fg = lane >> 4;  // k group 0..3
  const int rsw = fr & 7;    // this lane's row swizzle

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // B0: tile 0 (and the prologue table) in LDS
  unsigned long long t0, t1, t2, acc_mm = 0, acc_bar = 0, tbeg, tend;
  STAMP(tbeg);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt % NB;
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t0);
    __builtin_amdgcn_sched_barrier(0);
    const float* xrow = lds + cur * TILE + (wm * TM * 16 + fr) * BK;
    const float* wrow = lds + cur * TILE + BM * BK + (wn * TN * 16 + fr) * BK;
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
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][e], xf[j][e], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                        // ... and the loader's tile kt+1 has landed
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t2);
    acc_mm += t1 - t0; acc_bar += t2 - t1;
  }
  STAMP(tend);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    kpf_stamps[0] = 0; kpf_stamps[1] = acc_mm; kpf_stamps[2] = acc_bar; kpf_stamps[4] = tend - tbeg; kpf_stamps[5] = nk;
  }

  // ---- epilogue: lane holds channels n..n+3 (n = tile + 4*fg) of pixel m (= tile + fr) ----
  const unsigned fl = a.flags;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + (wm * TM + j) * 16 + fr;
    if (m >= a.M) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + (wn * TN + i) * 16 + fg * 4;
      if (n >= a.N) continue;
      f32x4 v = acc[i][j];
      const bool full = (n + 3 < a.N) && a.vec;
      f32x4 bv = zero4, gv = {1.f, 1.f, 1.f, 1.f}, rv = zero4;
      if (full) {
        if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
        if (fl & KPF_RES_GAMMA) gv = *reinterpret_cast<const f32x4*>(a.gamma + n);
        if (fl & KPF_RES_ADD) rv = *reinterpret_cast<const f32x4*>(a.res + (long)m * a.res_ld + a.res_coff + n);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < a.N) {
            if (a.bias) bv[e] = a.bias[n + e];
            if (fl & KPF_RES_GAMMA) gv[e] = a.gamma[n + e];
            if (fl & KPF_RES_ADD) rv[e] = a.res[(long)m * a.res_ld + a.res_coff + n + e];
          }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = v[e] + bv[e];
        if (fl & KPF_ACT_RELU) y = fmaxf(y, 0.f);
        if (fl & KPF_ACT_GELU) y = gelu_erf(y);
        if (fl & KPF_RES_GAMMA) y = y * gv[e];
        if (fl & KPF_RES_ADD) y = rv[e] + y;
        if (fl & KPF_RELU_AFTER_RES) y = fmaxf(y, 0.f);
        v[e] = y;
      }
      if (fl & KPF_OUT_NCHW) {
        const int b = m / a.ohow;
        const int pix = m - b * a.ohow;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < a.N) a.out[((long)b * a.N + n + e) * a.ohow + pix] = v[e];
      } else if (full) {
        *reinterpret_cast<f32x4*>(a.out + (long)m * a.out_ld + a.out_coff + n) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < a.N) a.out[(long)m * a.out_ld + a.out_coff + n + e] = v[e];
      }
    }
  }
}

template <int TM, int TN, int WM, int WN, bool IS1X1, bool HAS_PRO>
int launch_one(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  const size_t lds = (size_t)(NB * (BM + BN) * BK + (HAS_PRO ? 2 * a.Kp : 0)) * sizeof(float);
  auto kern = igemm_f32_kernel<TM, TN, WM, WN, IS1X1, HAS_PRO>;
  static bool attr_set = false;  // > 64 KiB of dynamic LDS needs an opt-in; benign if two threads race to set it
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      kpf_set_error("kpf_conv2d_f32: cannot raise the dynamic LDS limit");
      return KPF_ELAUNCH;
    }
    attr_set = true;
  }
  if (lds > 160 * 1024) {
    kpf_set_error("kpf_conv2d_f32: operand prologue too long for LDS (Kp=%d)", a.Kp);
    return KPF_EINVAL;
  }
  hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(64 * (WM * WN + 1)), lds, st, a);
  return kpf_check_launch("kpf_conv2d_f32");
}

template <int TM, int TN, int WM, int WN>
int launch_cfg(ConvArgs& a, bool is1x1, hipStream_t st) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  const int tilesM = (a.M + BM - 1) / BM;
  a.tilesN = (a.N + BN - 1) / BN;
  a.nblk = tilesM * a.tilesN;
  // the operand prologue only occurs on 1x1 convolutions (pre-activation Residual.conv1): 3 instantiations per tile shape
  if (a.ps && !is1x1) {
    kpf_set_error("kpf_conv2d_f32: the operand prologue is only supported for 1x1 stride-1 convolutions");
    return KPF_EINVAL;
  }
  if (a.ps) return launch_one<TM, TN, WM, WN, true, true>(a, st);
  if (is1x1) return launch_one<TM, TN, WM, WN, true, false>(a, st);
  return launch_one<TM, TN, WM, WN, false, false>(a, st);
}

// Tile choice.  The kernel is MFMA-bound, so a launch takes about ceil(blocks / 256 CUs) rounds of one tile's work
// (co-resident workgroups share a CU's matrix pipe: they hide bubbles, they do not add throughput).  Padding waste is
// inside blocks*bm*bn; small tiles amortise staging/epilogue worse, hence the mild penalty.
struct Cfg {
  int bm, bn;
};
static const Cfg kCfgs[] = {{128, 128}, {128, 96}, {128, 64}, {256, 48}, {128, 112}, {64, 128}, {64, 64}, {32, 64}, {256, 128}};

double cfg_cost(const Cfg& c, long M, long N) {
  const long tm = (M + c.bm - 1) / c.bm, tn = (N + c.bn - 1) / c.bn;
  const long blocks = tm * tn;
  const double rounds = blocks <= 256 ? 1.0 : (double)blocks / 256.0;  // beyond one round the tail overlaps
  const double small_pen = 1.0 + 0.05 * (128.0 * 128.0 / (c.bm * c.bn) - 1.0);
  return rounds * c.bm * c.bn * small_pen;
}

}  // namespace

extern "C" int kpf_debug_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kpf_stamps), 64); }

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
  KPF_REQUIRE(!((fl & KPF_ACT_RELU) && (fl & KPF_ACT_GELU)), "kpf_conv2d_f32: one activation only");
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
  a.flags = fl; a.tilesN = 0; a.nblk = 0;
  a.vec = (d->out_ld % 4 == 0 && d->out_coff % 4 == 0 && (!(fl & KPF_RES_ADD) || (d->res_ld % 4 == 0 && d->res_coff % 4 == 0))) ? 1 : 0;
  const bool is1x1 = d->KH == 1 && d->KW == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 &&
                     d->IH == d->OH && d->IW == d->OW;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  int best = 0;
  double bc = 1e30;
  static const int forced = []() { const char* e = getenv("KPF_FORCE_CFG"); return e ? atoi(e) : -1; }();  // tuning aid only
  for (int i = 0; i < (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); ++i) {
    const double c = cfg_cost(kCfgs[i], a.M, a.N);
    if (c < bc) { bc = c; best = i; }
  }
  if (forced >= 0) best = forced;
  switch (best) {
    case 0: return launch_cfg<4, 4, 2, 2>(a, is1x1, st);   // 128 x 128
    case 1: return launch_cfg<4, 3, 2, 2>(a, is1x1, st);   // 128 x 96
    case 2: return launch_cfg<2, 4, 4, 1>(a, is1x1, st);   // 128 x 64
    case 3: return launch_cfg<4, 3, 4, 1>(a, is1x1, st);   // 256 x 48
    case 4: return launch_cfg<2, 7, 4, 1>(a, is1x1, st);   // 128 x 112
    case 5: return launch_cfg<2, 4, 2, 2>(a, is1x1, st);   // 64 x 128
    case 6: return launch_cfg<2, 2, 2, 2>(a, is1x1, st);   // 64 x 64
    case 8: return launch_cfg<4, 4, 4, 2>(a, is1x1, st);   // 256 x 128, 8 MFMA waves
    default: return launch_cfg<2, 1, 1, 4>(a, is1x1, st);  // 32 x 64
  }
}
