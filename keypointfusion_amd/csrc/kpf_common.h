// Shared helpers for the gfx950 kernels of libkpf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/kpf.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

void kpf_set_error(const char* fmt, ...);

#define KPF_REQUIRE(cond, ...)    \
  do {                            \
    if (!(cond)) {                \
      kpf_set_error(__VA_ARGS__); \
      return KPF_EINVAL;          \
    }                             \
  } while (0)

static inline int kpf_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    kpf_set_error("%s: %s", what, hipGetErrorString(e));
    return KPF_ELAUNCH;
  }
  return KPF_OK;
}

// Opt a kernel into > 64 KiB of dynamic LDS on the CURRENT device, once per (kernel instantiation, device): the attribute is per
// device, and one process may drive several GPUs (torch.nn.DataParallel, the reference's own wrapper).  `done` is the caller's
// function-local static array, one flag per device ordinal.
#include <atomic>
constexpr int KPF_MAX_DEVICES = 64;
static inline bool kpf_raise_lds_limit(const void* kern, std::atomic<bool>* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= KPF_MAX_DEVICES) return false;
  if (done[dev].load(std::memory_order_acquire)) return true;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  done[dev].store(true, std::memory_order_release);
  return true;
}

static inline bool kpf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// wave64 butterfly sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// "Split" operand format for the 3 x f16 MFMA path (kpf_conv.hip): a row of C fp32 values (C % 32 == 0) occupies the same
// C*4 bytes as C/32 blocks of [32 x f16 hi | 32 x f16 lo] with x ~= hi + lo (22 significant bits; |x| is clamped to the
// f16 range).  Producers (GEMM / LayerNorm epilogues) write it, the GEMM's LDS-DMA staging reads it byte-for-byte like fp32.
// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// 16-bit activation storage (KPF_DT_BF16 / KPF_DT_F16 of include/kpf.h): kernels compute in fp32 and are templated on the storage
// type of their activation pointers; 4 consecutive channels are one 16-byte (fp32) or 8-byte (16-bit) access.
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 kpf_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 kpf_ld4(const f16_t* p) {
  const f16x4 h = *reinterpret_cast<const f16x4*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ f32x4 kpf_ld4(const bf16_t* p) {  // bf16 -> fp32 is a 16-bit shift
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ void kpf_st4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void kpf_st4(f16_t* p, const f32x4 v) {
  // The value is pinned as an fp32 register first: left to itself the compiler fuses the multiply / add that produced it with the conversion
  // (v_fma_mixlo_f16: ONE rounding) in some instantiations and not in others (v_mul_f32 + v_cvt_pk_f16_f32: two) — 3e-5 of the elements then differ
  // by an ulp between two kernels that compute the same layer, and a sample's result must not depend on which kernel its batch size selects.
  f32x4 u = v;
  asm("" : "+v"(u));
  *reinterpret_cast<f16x4*>(p) = f16x4{(f16_t)u[0], (f16_t)u[1], (f16_t)u[2], (f16_t)u[3]};
}
__device__ __forceinline__ void kpf_st4(bf16_t* p, const f32x4 v) {  // round to nearest even (v_cvt_pk_bf16_f32)
  *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}

// GELU for 16-bit outputs, ONE definition for every kernel that rounds its result to bf16 / f16 (a sample's result must not depend on which kernel its batch
// size selects): x * sigmoid(x (c1 + c3 x^2 + c5 x^4)), a minimax fit of x Phi(x) on [-9, 9] with |error| <= 2.6e-5 ABSOLUTE (tools/gelu_fit.py), in 9 VALU
// operations (2 transcendental) where the fp32-accurate erfc form takes 16.  Round 5: (i) the polynomial's argument is CLAMPED to [-9, 9] — its x^4 term is
// negative, so beyond |x| = 11.1 the unclamped exponent changed sign and the function returned 0 for x = 12 and x for x = -12 (found while folding constants; the
// tests' operands never left |x| < 6, a trained ConvNeXt's hidden pre-activations do); outside the interval sigmoid is 1 - 2^-38 / 2^-38, i.e. the result is x / 0 to
// fp32 rounding; (ii) -log2(e) is folded into the coefficients, which pays for the clamp.  What the error means: the activations are stored in 16 bits and then
// summed by pwconv2, so it is the ABSOLUTE error of a hidden value that reaches the output; 2.6e-5 is a tenth of the rounding of an O(1) f16 value (2.4e-4) and a
// 75th of a bf16 one.  In the negative tail, where |GELU| itself drops below 1e-3, the RELATIVE error of the fit reaches 15 % — of values that small; the per-operation
// test bounds relative error + this absolute term.  fp32 storage always uses the erfc form.
__device__ __forceinline__ float kpf_gelu_h16(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -9.0f, 9.0f);
  const float x2 = xc * xc;
  const float t = xc * fmaf(x2, fmaf(x2, 1.014263055e-03f, -1.067757239e-01f), -2.301121342f);  // -log2(e) * x (c1 + c3 x^2 + c5 x^4)
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

__device__ __forceinline__ void kpf_store_split4(float* row, int c, const f32x4 v) {  // c % 4 == 0
  f16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = __builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
    h[e] = (_Float16)x;
    l[e] = (_Float16)(x - (float)h[e]);
  }
  _Float16* blk = reinterpret_cast<_Float16*>(row + (c & ~31)) + (c & 31);
  *reinterpret_cast<f16x4*>(blk) = h;
  *reinterpret_cast<f16x4*>(blk + 32) = l;
}
