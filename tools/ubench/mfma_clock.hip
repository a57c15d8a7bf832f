// Micro-benchmark: effective matrix-core clock under load. One wave per SIMD issues independent
// v_mfma_f32_32x32x2_f32 (64 cycles each, 4 accumulators interleaved) for ~several ms; the same kernel is run on
// 16 workgroups (16 CUs busy) and on 256 / 512 (whole chip). cycles = iters * 64 MFMAs * 64 ; clock = cycles / time.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int r = 0; r < 4; ++r) for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r], 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 4; ++r) for (int e = 0; e < 16; ++e) s += acc[r][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

void run(int grid, int iters) {
  float* d; hipMalloc(&d, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 10, 1.f, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int rounds = (grid + 255) / 256;
  double cycles = (double)iters * 64 * 64 * rounds;
  double flop = (double)grid * 4 * iters * 64 * (2.0 * 32 * 32 * 2);
  printf("grid=%4d iters=%6d : %8.3f ms  %6.1f TFLOP/s  effective MFMA clock %.2f GHz\n", grid, iters, ms, flop / ms / 1e9,
         cycles / ms / 1e6);
  hipFree(d);
}
int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run(16, 2000); run(16, 20000); run(256, 2000); run(256, 20000); run(256, 100000); run(512, 20000);
  }
  return 0;
}
