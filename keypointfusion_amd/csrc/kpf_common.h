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
