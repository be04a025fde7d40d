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
#include "kpf_common.h"

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
  unsigned flags;
  int tilesN, nblk;
  int vec;  // 1: output/residual rows are 16-byte aligned -> float4 epilogue
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int TM, int TN, int WM, int WN, bool IS1X1, bool HAS_PRO>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const ConvArgs a) {
  constexpr int BM = 16 * TM * WM;
  constexpr int BN = 16 * TN * WN;
  constexpr int BK = 32;
  constexpr int AP = BM / 32;         // A staging passes (rows per pass = 32)
  constexpr int BP = (BN + 31) / 32;  // B staging passes
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(BM % 32 == 0, "BM multiple of 32");

  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * BK];
  float* As = lds;
  float* Bs = lds + BM * BK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;

  // XCD-aware bijective remap: blocks b, b+8, b+16.. share an XCD -> give each XCD a contiguous range of logical tiles.
  int bid = blockIdx.x;
  {
    const int nb = a.nblk, q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int nt = bid % a.tilesN, mt = bid / a.tilesN;
  const int m0 = mt * BM, n0 = nt * BN;

  const int lr = tid >> 3;        // staging row within a pass, 0..31
  const int kc = (tid & 7) * 4;   // staging k offset within the K tile
  const int wcol = (((tid & 7) ^ (lr & 7)) << 2);  // swizzled LDS column of this thread's 16-byte chunk

  // per staged A row: where its receptive field starts
  int rbase[AP], riy[AP], rix[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = m0 + lr + 32 * p;
    if (m < a.M) {
      if (IS1X1) {
        rbase[p] = m;
        riy[p] = 0;
        rix[p] = 0;
      } else {
        const int b = m / a.ohow;
        const int r = m - b * a.ohow;
        const int oy = r / a.OW;
        const int ox = r - oy * a.OW;
        rbase[p] = b * a.IH * a.IW;
        riy[p] = oy * a.sh - a.ph;
        rix[p] = ox * a.sw - a.pw;
      }
    } else {
      rbase[p] = -1;
      riy[p] = 0;
      rix[p] = 0;
    }
  }

  // Staging registers.  Loads are issued unconditionally (from a safe address when the element is padding / out of range) and
  // nothing touches their results until store_tile(), so the whole next K-tile stays in flight under the current tile's MFMAs;
  // validity travels in a bit mask and the operand prologue is applied at store time.
  f32x4 ra[AP], rb[BP];
  f32x4 ps4 = {0.f, 0.f, 0.f, 0.f}, pt4 = {0.f, 0.f, 0.f, 0.f};
  unsigned amask = 0, bmask = 0;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_tile = [&](int kt) {
    const int k = kt * BK + kc;
    const bool kvalid = k < a.K;
    int c = k, ky = 0, kx = 0;
    if (!IS1X1) {
      const int tap = k / a.Cin;
      c = k - tap * a.Cin;
      ky = tap / a.KW;
      kx = tap - ky * a.KW;
    }
    if (HAS_PRO) {
      const int cs = kvalid ? c : 0;
      ps4 = *reinterpret_cast<const f32x4*>(a.ps + cs);
      pt4 = *reinterpret_cast<const f32x4*>(a.pt + cs);
    }
    amask = 0;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      bool v = kvalid && rbase[p] >= 0;
      long off;
      if (IS1X1) {
        off = (long)rbase[p] * a.in_ld + a.in_coff + c;
      } else {
        const int iy = riy[p] + ky, ix = rix[p] + kx;
        v = v && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
        off = ((long)rbase[p] + (long)iy * a.IW + ix) * a.in_ld + a.in_coff + c;
      }
      ra[p] = *reinterpret_cast<const f32x4*>(a.in + (v ? off : 0l));
      amask |= (v ? 1u : 0u) << p;
    }
    bmask = 0;
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int rl = lr + 32 * p;
      const int n = n0 + rl;
      const bool v = rl < BN && n < a.N;
      rb[p] = *reinterpret_cast<const f32x4*>(a.w + (v ? ((long)n * a.Kp + kt * BK + kc) : 0l));
      bmask |= (v ? 1u : 0u) << p;
    }
  };

  auto store_tile = [&]() {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      f32x4 x = ra[p];
      if (HAS_PRO) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = fmaxf(fmaf(x[e], ps4[e], pt4[e]), 0.f);
      }
      if (!((amask >> p) & 1u)) x = zero4;
      *reinterpret_cast<f32x4*>(As + (lr + 32 * p) * BK + wcol) = x;
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int rl = lr + 32 * p;
      f32x4 x = rb[p];
      if (!((bmask >> p) & 1u)) x = zero4;
      if (rl < BN) *reinterpret_cast<f32x4*>(Bs + rl * BK + wcol) = x;
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = zero4;

  const int nk = a.Kp / BK;
  const int fr = lane & 15;   // fragment row (pixel for X, channel for W)
  const int fg = lane >> 4;   // k group 0..3
  const float* xrow = As + (wm * TM * 16 + fr) * BK;
  const float* wrow = Bs + (wn * TN * 16 + fr) * BK;
  const int sw0 = ((fg ^ (fr & 7)) << 2);        // k-step 0: chunk fg
  const int sw1 = (((4 + fg) ^ (fr & 7)) << 2);  // k-step 1: chunk 4+fg

  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int sc = s ? sw1 : sw0;
      f32x4 xf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xrow + j * 16 * BK + sc);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const f32x4*>(wrow + i * 16 * BK + sc);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][e], xf[j][e], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
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

template <int TM, int TN, int WM, int WN>
int launch_cfg(ConvArgs& a, bool is1x1, hipStream_t st) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  const int tilesM = (a.M + BM - 1) / BM;
  a.tilesN = (a.N + BN - 1) / BN;
  a.nblk = tilesM * a.tilesN;
  // the operand prologue only occurs on 1x1 convolutions (pre-activation Residual.conv1), so 3 instantiations suffice
  if (is1x1 && a.ps)
    hipLaunchKernelGGL((igemm_f32_kernel<TM, TN, WM, WN, true, true>), dim3(a.nblk), dim3(256), 0, st, a);
  else if (is1x1)
    hipLaunchKernelGGL((igemm_f32_kernel<TM, TN, WM, WN, true, false>), dim3(a.nblk), dim3(256), 0, st, a);
  else if (!a.ps)
    hipLaunchKernelGGL((igemm_f32_kernel<TM, TN, WM, WN, false, false>), dim3(a.nblk), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((igemm_f32_kernel<TM, TN, WM, WN, false, true>), dim3(a.nblk), dim3(256), 0, st, a);
  return kpf_check_launch("kpf_conv2d_f32");
}

// Tile choice.  The kernel is MFMA-bound, so a launch takes about ceil(blocks / 256 CUs) rounds of one tile's work
// (co-resident workgroups share a CU's matrix pipe: they hide bubbles, they do not add throughput).  Padding waste is
// inside blocks*bm*bn; small tiles amortise staging/epilogue worse, hence the mild penalty.
struct Cfg {
  int bm, bn;
};
static const Cfg kCfgs[] = {{128, 128}, {128, 96}, {128, 64}, {256, 48}, {128, 112}, {64, 128}, {64, 64}, {32, 64}};

double cfg_cost(const Cfg& c, long M, long N) {
  const long tm = (M + c.bm - 1) / c.bm, tn = (N + c.bn - 1) / c.bn;
  const long blocks = tm * tn;
  const double rounds = blocks <= 256 ? 1.0 : (double)blocks / 256.0;  // beyond one round the tail overlaps
  const double small_pen = 1.0 + 0.05 * (128.0 * 128.0 / (c.bm * c.bn) - 1.0);
  return rounds * c.bm * c.bn * small_pen;
}

}  // namespace

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
  a.flags = fl; a.tilesN = 0; a.nblk = 0;
  a.vec = (d->out_ld % 4 == 0 && d->out_coff % 4 == 0 && (!(fl & KPF_RES_ADD) || (d->res_ld % 4 == 0 && d->res_coff % 4 == 0))) ? 1 : 0;
  const bool is1x1 = d->KH == 1 && d->KW == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 &&
                     d->IH == d->OH && d->IW == d->OW;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  int best = 0;
  double bc = 1e30;
  for (int i = 0; i < (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); ++i) {
    const double c = cfg_cost(kCfgs[i], a.M, a.N);
    if (c < bc) { bc = c; best = i; }
  }
  switch (best) {
    case 0: return launch_cfg<4, 4, 2, 2>(a, is1x1, st);   // 128 x 128
    case 1: return launch_cfg<4, 3, 2, 2>(a, is1x1, st);   // 128 x 96
    case 2: return launch_cfg<2, 4, 4, 1>(a, is1x1, st);   // 128 x 64
    case 3: return launch_cfg<4, 3, 4, 1>(a, is1x1, st);   // 256 x 48
    case 4: return launch_cfg<2, 7, 4, 1>(a, is1x1, st);   // 128 x 112
    case 5: return launch_cfg<2, 4, 2, 2>(a, is1x1, st);   // 64 x 128
    case 6: return launch_cfg<2, 2, 2, 2>(a, is1x1, st);   // 64 x 64
    default: return launch_cfg<2, 1, 1, 4>(a, is1x1, st);  // 32 x 64
  }
}
