// Micro-benchmark: v_mfma_f32_16x16x4_f32 issue patterns on gfx950 (IL accumulators round-robin), 1 or 2 waves/SIMD.
// Reports cycles per MFMA per SIMD at the nominal 2.4 GHz (ideal: 32).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int IL>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[IL];
  for (int r = 0; r < IL; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = a0 + threadIdx.x + i; b[i] = b0 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int r = 0; r < IL; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[r], 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < IL; ++r) for (int e = 0; e < 4; ++e) s += acc[r][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int IL>
void run(int threads) {
  float* d; hipMalloc(&d, 64 * 512 * 4);
  const int iters = 20000 / IL, grid = 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<IL>, dim3(grid), dim3(threads), 0, 0, d, 10, 1.f, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<IL>, dim3(grid), dim3(threads), 0, 0, d, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 8 * IL * (threads / 256);
  printf("IL=%2d waves/SIMD=%d : %.3f ms  %.1f cycles/MFMA/SIMD (2.4 GHz)\n", IL, threads / 256, ms,
         ms * 1e-3 * 2.4e9 / mfma_per_simd);
  hipFree(d);
}
int main() {
  run<1>(256); run<2>(256); run<3>(256); run<4>(256); run<8>(256); run<13>(256);
  run<1>(512); run<2>(512); run<3>(512); run<4>(512); run<8>(512); run<13>(512);
  return 0;
}
