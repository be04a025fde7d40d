// Shared helpers for the gfx950 kernels of libkpf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/kpf.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

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
