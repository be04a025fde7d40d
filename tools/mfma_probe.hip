// Tuning aid: which ingredient of the GEMM inner loop costs MFMA issue rate?  (no global memory traffic)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e)); return 1; } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;
// 0: register operands only  1: + ds_read_b128 fragments  2: + barrier per K tile  3: + LDS-DMA staging of the next tile (src, stride)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed, const float* src, long tile_stride, long wrap) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 8192; i += 256) lds[i] = seed + i;
  __syncthreads();
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  f32x4 xf[4], wf[4];
  for (int j = 0; j < 4; ++j) { xf[j] = f32x4{seed + j, seed, seed + lane, 1.f}; wf[j] = f32x4{seed - j, 2.f, seed * lane, 3.f}; }
  const int fr = lane & 15, fg = lane >> 4;
  const float* gsrc = src + ((long)blockIdx.x * 977 % 64) * 8192 + tid * 4;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
      const float* g = gsrc + ((long)it * tile_stride) % wrap;
#pragma unroll
      for (int p = 0; p < 8; ++p)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(g + p * 1024), (lds_void_t*)(lds + ((it + 1) & 1) * 8192 + p * 1024 + wave * 256), 16, 0, 0);
    }
    const float* xrow = lds + (it & 1) * 8192 + ((wave & 1) * 64 + fr) * 32;
    const float* wrow = lds + (it & 1) * 8192 + 4096 + ((wave >> 1) * 64 + fr) * 32;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (MODE >= 1) {
        const int sc = (((4 * s + fg) ^ (fr & 7)) << 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xrow + j * 16 * 32 + sc);
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f32x4*>(wrow + i * 16 * 32 + sc);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][e], xf[j][e], acc[i][j], 0, 0, 0);
    }
    if (MODE >= 2) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
int run(const char* name, int blocks, float* out, const float* src = nullptr, long stride = 0, long wrap = 1) {
  const int iters = 400;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, src, stride, wrap);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, src, stride, wrap);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-34s blocks=%4d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, 2048.0 * 128 * iters * blocks * 4 / ms / 1e9);
  return 0;
}
int main() {
  float* out; CK(hipMalloc(&out, 4096 * 256 * 4));
  float* big; CK(hipMalloc(&big, (1l << 30) + (1 << 24))); CK(hipMemset(big, 0, (1l << 30) + (1 << 24)));
  for (int blocks : {256, 512, 1024}) {
    run<0>("regs only", blocks, out);
    run<1>("+ ds_read_b128 fragments", blocks, out);
    run<2>("+ barrier per K tile", blocks, out);
    run<3>("+ DMA 32KB/tile (L2-resident src)", blocks, out, big, 0, 1);
    run<3>("+ DMA 32KB/tile (streaming 1 GiB)", blocks, out, big, 8192 * 64, (1l << 28) - 8192 * 80);
  }
  return 0;
}
