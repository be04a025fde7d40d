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
