// Tuning aid: raw f32 MFMA issue rate on this GPU (no memory traffic).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float x = a + threadIdx.x, y = b - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float x = a + threadIdx.x, y = b - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// f16 16x16x32 with the split GEMM's dependency pattern: 3 dependent MFMAs per accumulator (DEP3) or one per accumulator
template <int NACC, bool DEP3>
__global__ __launch_bounds__(256) void kf16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f16x8 x, y, z;
  for (int j = 0; j < 8; ++j) {
    x[j] = (_Float16)(a + threadIdx.x + j);
    y[j] = (_Float16)(b - threadIdx.x);
    z[j] = (_Float16)(b * 0.001f);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(z, y, acc[i], 0, 0, 0);
      if (DEP3) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, z, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double flop_per_thread_block) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.3f ms  %.1f TFLOP/s\n", name, ms, flop_per_thread_block / ms / 1e9);
}
int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 20000;
  for (int blocks : {256, 512, 1024}) {
    printf("blocks=%d (x4 waves)\n", blocks);
    run("16x16x4 acc=4", [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 2048.0 * 4 * iters * blocks * 4);
    run("16x16x4 acc=16", [&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, out, iters / 4, 1.f, 2.f); }, 2048.0 * 16 * (iters / 4) * blocks * 4);
    run("f16 16x16x32 acc=16", [&] { hipLaunchKernelGGL((kf16<16, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 16384.0 * 16 * iters * blocks * 4);
    run("f16 16x16x32 acc=16 dep3", [&] { hipLaunchKernelGGL((kf16<16, true>), dim3(blocks), dim3(256), 0, 0, out, iters / 2, 1.f, 2.f); }, 16384.0 * 48 * (iters / 2) * blocks * 4);
    run("32x32x2 acc=4", [&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters / 2, 1.f, 2.f); }, 4096.0 * 4 * (iters / 2) * blocks * 4);
  }
  return 0;
}
