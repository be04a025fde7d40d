// Single-wave MFMA issue cadence, measured on a hand-placed instruction stream (round 3).
// tools/mfma_issue_rate.hip (round 2) let the compiler allocate the 16 accumulators: its loop came out with v_accvgpr_mov copies through
// a[0:3] and s_nop 0 / 2 / 5 pads between the MFMAs (dump: hipcc -S --cuda-device-only), so the ~29 cycles per v_mfma_f32_16x16x32_bf16 it
// reported for one wave per SIMD measured that code, not the matrix pipe.  Here the loop body is inline assembly: 16 back-to-back MFMAs
// on 16 distinct accumulator quads, operands either one A/B register pair for all (SAME) or four alternating pairs (DISTINCT), and a
// dependent chain on ONE accumulator (CHAIN).  Cycles per MFMA = s_memtime delta / MFMAs issued by the wave.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_issue_rate2.hip -o tools/bin/mfma_issue_rate2 && tools/bin/mfma_issue_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MF(ACC, A, B) "v_mfma_f32_16x16x32_bf16 %" #ACC ", %" #A ", %" #B ", %" #ACC "\n"

template <int MODE>  // 0 SAME operands, 1 DISTINCT operand pairs, 2 CHAIN (one accumulator)
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0, c8 = c0, c9 = c0, c10 = c0, c11 = c0, c12 = c0, c13 = c0, c14 = c0, c15 = c0;
  bf16x8 a0, b0, a1, b1, a2, b2, a3, b3;
  for (int e = 0; e < 8; ++e) {
    a0[e] = (__bf16)(threadIdx.x * 0.001f + e); b0[e] = (__bf16)(e * 0.5f);
    a1[e] = (__bf16)(threadIdx.x * 0.002f - e); b1[e] = (__bf16)(e * 0.25f);
    a2[e] = (__bf16)(threadIdx.x * 0.003f + 1); b2[e] = (__bf16)(1.f - e * 0.125f);
    a3[e] = (__bf16)(threadIdx.x * 0.004f - 2); b3[e] = (__bf16)(e * 0.0625f);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      asm volatile(MF(0, 16, 17) MF(1, 16, 17) MF(2, 16, 17) MF(3, 16, 17) MF(4, 16, 17) MF(5, 16, 17) MF(6, 16, 17) MF(7, 16, 17) MF(8, 16, 17) MF(9, 16, 17)
                   MF(10, 16, 17) MF(11, 16, 17) MF(12, 16, 17) MF(13, 16, 17) MF(14, 16, 17) MF(15, 16, 17)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(c8), "+v"(c9), "+v"(c10), "+v"(c11), "+v"(c12),
                     "+v"(c13), "+v"(c14), "+v"(c15)
                   : "v"(a0), "v"(b0));
    } else if (MODE == 1) {
      asm volatile(MF(0, 16, 17) MF(1, 18, 19) MF(2, 20, 21) MF(3, 22, 23) MF(4, 16, 17) MF(5, 18, 19) MF(6, 20, 21) MF(7, 22, 23) MF(8, 16, 17) MF(9, 18, 19)
                   MF(10, 20, 21) MF(11, 22, 23) MF(12, 16, 17) MF(13, 18, 19) MF(14, 20, 21) MF(15, 22, 23)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(c8), "+v"(c9), "+v"(c10), "+v"(c11), "+v"(c12),
                     "+v"(c13), "+v"(c14), "+v"(c15)
                   : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3));
    } else {
      asm volatile(MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2)
                   MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2) MF(0, 1, 2)
                   : "+v"(c0)
                   : "v"(a0), "v"(b0));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + c8 + c9 + c10 + c11 + c12 + c13 + c14 + c15;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int wgs, int threads) {
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, (size_t)wgs * 256 * 4); (void)hipMalloc(&clk, 16);
  const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, out, 1000, clk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, out, iters, clk);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double waves = (double)wgs * threads / 64;
  const double fl = waves * iters * 16 * (16.0 * 16 * 32 * 2);
  printf("%-34s wgs=%4d x %3d thr  %.3f ms  %6.0f TFLOP/s  %5.1f shader cycles per MFMA per wave  clock %.0f MHz\n", name, wgs, threads, ms, fl / ms / 1e9,
         (double)h[0] / ((double)iters * 16), (double)h[0] / (double)h[1] * 100.0);
  (void)hipFree(out); (void)hipFree(clk);
}

int main() {
  run<0>("SAME A/B, 16 accumulators", 1, 64);        // one wave on the whole chip: the pure issue cadence
  run<1>("DISTINCT A/B, 16 accumulators", 1, 64);
  run<2>("CHAIN, 1 accumulator", 1, 64);
  run<0>("SAME A/B, 16 accumulators", 256, 256);     // one wave per SIMD, every CU busy
  run<1>("DISTINCT A/B, 16 accumulators", 256, 256);
  run<2>("CHAIN, 1 accumulator", 256, 256);
  run<1>("DISTINCT A/B, 16 accumulators", 512, 256); // two waves per SIMD
  return 0;
}
