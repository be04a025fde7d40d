// Tuning aid: cycles per K tile of the split-GEMM inner loop (3 x v_mfma_f32_16x16x32_f16 per fragment pair) as its ingredients are
// added: 0 MFMAs on register operands, 1 + ds_read_b128 fragments, 2 + barrier per K tile, 3 + LDS-DMA staging of the next tile
// (L2-resident source), 4 = 3 with the DMA issued AFTER the fragment reads, 5 = 3 with a counted vmcnt + raw barrier on a 3-stage ring.
// hipcc --offload-arch=gfx950 -O3 split_probe.hip -o split_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;

constexpr int TILE = 8192;  // floats per stage: (128 + 128) rows x 32

template <int MODE, int NS>
__global__ __launch_bounds__(256, 4) void probe(float* out, unsigned long long* cyc, int iters, float seed, const float* src) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NS * TILE; i += 256) lds[i] = seed + i;
  __syncthreads();
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  f16x8 xh[4], xl[4], wh[4], wl[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 8; ++e) { xh[j][e] = (_Float16)(seed + j); xl[j][e] = (_Float16)(seed * 0.001f); wh[j][e] = (_Float16)(seed - e); wl[j][e] = (_Float16)(0.002f); }
  const int fr = lane & 15, fg = lane >> 4, rsw = (fr >> 1) & 7;
  const float* gsrc = src + ((long)(blockIdx.x * 977) % 64) * TILE + tid * 4;
  auto dma = [&](int it, int buf) {
    const float* g = gsrc + ((long)it * 7 % 64) * TILE;
#pragma unroll
    for (int p = 0; p < 8; ++p)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(g + p * 1024), (lds_void_t*)(lds + buf * TILE + p * 1024 + wave * 256), 16, 0, 0);
  };
  auto reads = [&](int buf) {
    const float* xrow = lds + buf * TILE + ((wave & 1) * 64 + fr) * 32;
    const float* wrow = lds + buf * TILE + 4096 + ((wave >> 1) * 64 + fr) * 32;
    const int ch = ((fg ^ rsw) << 2), cl = (((4 + fg) ^ rsw) << 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[j] = *reinterpret_cast<const f16x8*>(xrow + j * 16 * 32 + ch);
      xl[j] = *reinterpret_cast<const f16x8*>(xrow + j * 16 * 32 + cl);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wh[i] = *reinterpret_cast<const f16x8*>(wrow + i * 16 * 32 + ch);
      wl[i] = *reinterpret_cast<const f16x8*>(wrow + i * 16 * 32 + cl);
    }
  };
  auto mma = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[j], acc[i][j], 0, 0, 0);
      }
  };
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 6 || MODE == 7) {  // register double-buffered fragments: reads of tile it+1 are issued before the MFMAs of tile it
    f16x8 xh2[4], xl2[4], wh2[4], wl2[4];
    auto reads2 = [&](int buf) {
      const float* xrow = lds + buf * TILE + ((wave & 1) * 64 + fr) * 32;
      const float* wrow = lds + buf * TILE + 4096 + ((wave >> 1) * 64 + fr) * 32;
      const int ch = ((fg ^ rsw) << 2), cl = (((4 + fg) ^ rsw) << 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh2[j] = *reinterpret_cast<const f16x8*>(xrow + j * 16 * 32 + ch);
        xl2[j] = *reinterpret_cast<const f16x8*>(xrow + j * 16 * 32 + cl);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wh2[i] = *reinterpret_cast<const f16x8*>(wrow + i * 16 * 32 + ch);
        wl2[i] = *reinterpret_cast<const f16x8*>(wrow + i * 16 * 32 + cl);
      }
    };
    auto mma2 = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl2[i], xh2[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh2[i], xl2[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh2[i], xh2[j], acc[i][j], 0, 0, 0);
        }
    };
    dma(0, 0);
    dma(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    reads(0);
    for (int it = 0; it < iters; it += 2) {
      if (MODE == 7) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
      dma(it + 2, (it + 2) % 3);
      reads2((it + 1) % 3);
      mma();
      if (MODE == 7) {  // interleave: (3 MFMA, 1 DS read, 3 MFMA, 1 DS read, 1 VMEM) x 8
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      dma(it + 3, (it + 3) % 3);
      reads((it + 2) % 3);
      mma2();
      if (MODE == 7) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    }
  } else if (MODE == 8) {  // single LDS stage, two barriers per K tile: relies on 3-4 co-resident workgroups for overlap
    for (int it = 0; it < iters; ++it) {
      dma(it, 0);
      __syncthreads();  // tile landed
      reads(0);
      mma();
      __syncthreads();  // everyone done reading before the next DMA overwrites the stage
    }
  } else if (MODE == 5) {
    dma(0, 0);
    dma(1, 1);
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      dma(it + 2, (it + 2) % 3);
      reads(it % 3);
      mma();
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      if (MODE == 3) dma(it, (it + 1) & 1);
      if (MODE >= 1) reads(it & 1);
      if (MODE == 4) dma(it, (it + 1) & 1);
      mma();
      if (MODE >= 2) __syncthreads();
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NS>
int run(const char* name, int blocks, float* out, unsigned long long* cyc, const float* src) {
  const int iters = 400;
  auto kern = probe<MODE, NS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), NS * TILE * 4, 0, out, cyc, iters, 1.f, src);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), NS * TILE * 4, 0, out, cyc, iters, 1.f, src);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  static unsigned long long h[8192];
  CK(hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double c = 0;
  for (int i = 0; i < blocks; ++i) c += (double)h[i];
  const double flop = 2.0 * 128 * 128 * 32 * (double)iters * blocks;  // algorithmic (one product per split triple)
  printf("%-44s blocks=%4d  %.3f ms  %7.1f TF-eq  %6.0f cycles per K tile per workgroup\n", name, blocks, ms, flop / ms / 1e9, c / blocks / iters);
  return 0;
}

int main() {
  float *out, *src;
  unsigned long long* cyc;
  CK(hipMalloc(&out, 4096 * 256 * 4));
  CK(hipMalloc(&cyc, 4096 * 8));
  CK(hipMalloc(&src, 80 * TILE * 4));
  CK(hipMemset(src, 0, 80 * TILE * 4));
  for (int blocks : {768, 1024, 1280}) {
    run<8, 1>("8 single LDS stage, 2 barriers", blocks, out, cyc, src);
    run<3, 2>("3 double-buffered (64 KB)", blocks, out, cyc, src);
  }
  for (int blocks : {256, 512}) {
    run<0, 2>("0 mfma only (register operands)", blocks, out, cyc, src);
    run<1, 2>("1 + ds_read_b128 fragments", blocks, out, cyc, src);
    run<2, 2>("2 + __syncthreads per K tile", blocks, out, cyc, src);
    run<3, 2>("3 + LDS-DMA of the next tile (before reads)", blocks, out, cyc, src);
    run<4, 2>("4   same, DMA issued after the reads", blocks, out, cyc, src);
    if (blocks == 256) run<5, 3>("5 3-stage ring, counted vmcnt, raw barrier", blocks, out, cyc, src);
    if (blocks == 256) run<6, 3>("6 ring + register-prefetched fragments", blocks, out, cyc, src);
    if (blocks == 256) run<7, 3>("7 = 6 + sched_group_barrier interleave", blocks, out, cyc, src);
  }
  return 0;
}
