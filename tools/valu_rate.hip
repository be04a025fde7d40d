// Issue rate of the vector-ALU forms a depthwise stencil can be written in (round 5): v_fma_f32, v_fma_mix_f32 (f16 operand taken straight from a packed
// register), v_pk_fma_f32, and v_cvt_f32_f16 + v_fma_f32 — 16 independent accumulators per thread, N waves per SIMD, every CU busy.
// Reports cycles per wave-instruction per SIMD (wall time x clock / instructions issued per SIMD) and effective FMA lanes per clock per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/bin/valu_rate && tools/bin/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define FMA(i) "v_fma_f32 %" #i ", %16, %17, %" #i "\n"
#define MIXL(i) "v_fma_mix_f32 %" #i ", %16, %17, %" #i " op_sel_hi:[1,0,0]\n"
#define MIXH(i) "v_fma_mix_f32 %" #i ", %16, %17, %" #i " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"

#define DOT2H(i) "v_dot2_f32_f16 %" #i ", %16, %17, %" #i "\n"
#define DOT2B(i) "v_dot2_f32_bf16 %" #i ", %16, %17, %" #i "\n"
#define PERM(i) "v_perm_b32 %" #i ", %" #i ", %16, %17\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float c[16];
  for (int i = 0; i < 16; ++i) c[i] = threadIdx.x * 1e-3f + i;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f;
  unsigned h = 0x3c003c00u + threadIdx.x;  // two halves near 1.0
  unsigned h2 = 0x3bff3bffu - threadIdx.x;
  f32x2 p[8], pa = {a, a}, pb = {b, b};
  for (int i = 0; i < 8; ++i) p[i] = f32x2{c[2 * i], c[2 * i + 1]};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      asm volatile(R16(FMA) : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]),
                   "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]) : "v"(a), "v"(b));
    } else if (MODE == 1) {
      asm volatile(MIXL(0) MIXH(1) MIXL(2) MIXH(3) MIXL(4) MIXH(5) MIXL(6) MIXH(7) MIXL(8) MIXH(9) MIXL(10) MIXH(11) MIXL(12) MIXH(13) MIXL(14) MIXH(15)
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]),
                     "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]) : "v"(h), "v"(b));
    } else if (MODE == 2) {  // 8 packed FMAs = the same 16 multiply-adds
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pa), "v"(pb));
    } else if (MODE == 4) {  // two f16 x f16 products + an f32 addend per lane and instruction: a depthwise row's 7 taps as 4 of these (tap pairs per channel)
      asm volatile(R16(DOT2H) : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]),
                   "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]) : "v"(h), "v"(h2));
    } else if (MODE == 5) {
      asm volatile(R16(DOT2B) : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]),
                   "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]) : "v"(h), "v"(h2));
    } else if (MODE == 6) {  // the byte permute that would pair taps: (x[w][c0 c1], x[w+1][c0 c1]) -> (x[w][c0] x[w+1][c0])
      unsigned* u = reinterpret_cast<unsigned*>(c);
      asm volatile(R16(PERM) : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]), "+v"(u[8]), "+v"(u[9]), "+v"(u[10]),
                   "+v"(u[11]), "+v"(u[12]), "+v"(u[13]), "+v"(u[14]), "+v"(u[15]) : "v"(h), "v"(0x05040100u));
    } else {  // MODE 3: 4 conversions feed 16 FMAs (the ratio of a 7-tap row: 14 x 4 conversions per 224 FMAs)
      float f0, f1, f2, f3;
      asm volatile("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                   "v_cvt_f32_f16 %2, %5\n v_cvt_f32_f16_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                   : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3) : "v"(h), "v"(h + it));
      asm volatile("v_fma_f32 %0, %16, %20, %0\n v_fma_f32 %1, %17, %20, %1\n v_fma_f32 %2, %18, %20, %2\n v_fma_f32 %3, %19, %20, %3\n"
                   "v_fma_f32 %4, %16, %20, %4\n v_fma_f32 %5, %17, %20, %5\n v_fma_f32 %6, %18, %20, %6\n v_fma_f32 %7, %19, %20, %7\n"
                   "v_fma_f32 %8, %16, %20, %8\n v_fma_f32 %9, %17, %20, %9\n v_fma_f32 %10, %18, %20, %10\n v_fma_f32 %11, %19, %20, %11\n"
                   "v_fma_f32 %12, %16, %20, %12\n v_fma_f32 %13, %17, %20, %13\n v_fma_f32 %14, %18, %20, %14\n v_fma_f32 %15, %19, %20, %15\n"
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]),
                     "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(b));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int instr_per_iter) {
  const int wgs = 256 * waves_per_simd, iters = 4000;  // 256-thread workgroups: one wave per SIMD each
  float* out;
  (void)hipMalloc(&out, (size_t)wgs * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)waves_per_simd * iters * instr_per_iter;  // wave-instructions one SIMD issued
  const double ns = ms * 1e6 / per_simd;
  printf("%-34s %d waves/SIMD  %8.3f ms  %6.2f ns per wave-instruction  = %5.2f clk at 2.4 GHz   %6.1f G multiply-adds/s per SIMD\n", name, waves_per_simd, ms, ns,
         ns * 2.4, (double)waves_per_simd * iters * 16 * 64 / (ms * 1e6));
  (void)hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", w, 16);
    run<1>("v_fma_mix_f32 (f16 lo/hi)", w, 16);
    run<2>("v_pk_fma_f32 (8 = 16 FMAs)", w, 8);
    run<3>("4 v_cvt_f32_f16 + 16 v_fma_f32", w, 20);
    run<4>("v_dot2_f32_f16 (2 products each)", w, 16);
    run<5>("v_dot2_f32_bf16 (2 products each)", w, 16);
    run<6>("v_perm_b32", w, 16);
  }
  return 0;
}
