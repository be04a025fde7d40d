// Register-only MFMA throughput under sustained load (no memory traffic): what the matrix pipe delivers at the clock the chip
// actually holds.  usage: hipcc -O3 --offload-arch=gfx950 tools/mfma_issue_rate.hip -o tools/bin/mfma_issue_rate && tools/bin/mfma_issue_rate   (prints TFLOP/s and the implied clock for bf16 16x16x32, f16 16x16x32 and f32 16x16x4)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 32x32x16 bf16: a 32-cycle instruction (8 independent 16-register accumulators per wave)
__global__ __launch_bounds__(256) void k32(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  f16x8 ah, bh;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); ah[e] = (_Float16)(threadIdx.x * 0.001f); bh[e] = (_Float16)e; }
  const float af = threadIdx.x * 0.001f, bf = 1.5f;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      else if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, double flop_per_mfma, int wgs) {
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, (size_t)wgs * 256 * 4); (void)hipMalloc(&clk, 16);
  const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, 1000, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double fl = (double)wgs * 4 * iters * 16 * flop_per_mfma;
  printf("%-22s wgs=%d  %.3f ms  %.0f TFLOP/s  in-kernel clock %.0f MHz (shader cycles per 100 MHz tick)\n", name, wgs, ms, fl / ms / 1e9,
         (double)h[0] / (double)h[1] * 100.0);
  hipFree(out); hipFree(clk);
}

void run32(int wgs) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, (size_t)wgs * 256 * 4); hipMalloc(&clk, 16);
  const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k32, dim3(wgs), dim3(256), 0, 0, out, 1000, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k32, dim3(wgs), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double fl = (double)wgs * 4 * iters * 8 * (32.0 * 32 * 16 * 2);
  printf("%-22s wgs=%d  %.3f ms  %.0f TFLOP/s  in-kernel clock %.0f MHz\n", "bf16 32x32x16", wgs, ms, fl / ms / 1e9, (double)h[0] / (double)h[1] * 100.0);
  hipFree(out); hipFree(clk);
}

int main() {
  for (int wgs : {256, 512}) {
    run32(wgs);
    run<0>("bf16 16x16x32", 16.0 * 16 * 32 * 2, wgs);
    run<1>("f16 16x16x32", 16.0 * 16 * 32 * 2, wgs);
    run<2>("f32 16x16x4", 16.0 * 16 * 4 * 2, wgs);
  }
  return 0;
}
