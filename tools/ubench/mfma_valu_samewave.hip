// Micro-benchmark: VALU instructions threaded between ONE wave's own MFMAs (gfx950), one wave per SIMD.
// Per iteration 48 MFMAs (3 accumulators round-robin) with NV independent fp32 FMAs after each MFMA (NV = 0, 1, 2, 4, 8).
// If the VALU ops issue in the MFMAs' shadow the time stays at 48 x (MFMA cycles); if not it grows by 4 cycles per FMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, int FLAVOUR>      // FLAVOUR 0: v_mfma_f32_16x16x4_f32 ; 1: v_mfma_f32_16x16x32_f16 ; 2: v_mfma_f32_32x32x2_f32
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 big[3];
  for (int r = 0; r < 3; ++r) for (int e = 0; e < 16; ++e) big[r][e] = 0.f;
  float a[16], b[16], v[8];
  f16x8 ah[4], bh[4];
  for (int i = 0; i < 16; ++i) { a[i] = a0 + threadIdx.x + i; b[i] = b0 + i; }
  for (int i = 0; i < 8; ++i) v[i] = a0 * (threadIdx.x + i) * 1e-3f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { ah[i][e] = (_Float16)(a0 + i + e); bh[i][e] = (_Float16)(b0 + e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (FLAVOUR == 0) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[r], 0, 0, 0);
        else if (FLAVOUR == 1) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s & 3], bh[s & 3], acc[r], 0, 0, 0);
        else big[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], big[r], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b0), "v"(a0));
      }
  }
  float res = 0.f;
  for (int r = 0; r < 3; ++r) { for (int e = 0; e < 4; ++e) res += acc[r][e]; for (int e = 0; e < 16; ++e) res += big[r][e]; }
  for (int i = 0; i < 8; ++i) res += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = res;
}

template <int NV, int FLAVOUR>
static float run() {
  float* d; hipMalloc(&d, 1024 * 256 * 4);
  const int iters = 4000, grid = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, FLAVOUR>), dim3(grid), dim3(256), 0, 0, d, 10, 1.f, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, FLAVOUR>), dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms * 1e-3f * 2.4e9f / iters / 48.f;       // cycles per (MFMA + NV FMAs) at 2.4 GHz
}
template <int FLAVOUR>
static void row(const char* name) {
  printf("%-28s cycles per MFMA slot with 0 / 1 / 2 / 4 / 8 v_fma_f32 after each MFMA: %5.1f %5.1f %5.1f %5.1f %5.1f\n", name,
         run<0, FLAVOUR>(), run<1, FLAVOUR>(), run<2, FLAVOUR>(), run<4, FLAVOUR>(), run<8, FLAVOUR>());
}
int main() {
  row<0>("v_mfma_f32_16x16x4_f32");
  row<1>("v_mfma_f32_16x16x32_f16");
  row<2>("v_mfma_f32_32x32x2_f32");
  return 0;
}
