// Micro-benchmark: issue patterns of v_mfma_f32_32x32x2_f32 (exact-fp32 matrix cores) on gfx950.
// dep16: 16 dependent MFMAs on one accumulator, then the next accumulator (the v1 conv kernel's pattern);
// il2 / il4: 2 / 4 accumulators interleaved.  Prints TFLOP/s for 256 CUs x 4 waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int IL>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int r = 0; r < 4; ++r) for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
    if (IL == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r], 0, 0, 0);
    } else if (IL == 2) {
#pragma unroll
      for (int r = 0; r < 4; r += 2)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r], 0, 0, 0);
          acc[r + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r + 1], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 4; ++r) for (int e = 0; e < 16; ++e) s += acc[r][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int IL>
void run(const char* name, int wg_per_cu) {
  float* d; hipMalloc(&d, 256 * 1024 * 4 * 4);
  const int iters = 2000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<IL>, dim3(grid), dim3(256), 0, 0, d, 10, 1.f, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<IL>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)grid * 4 * iters * 64 * (2.0 * 32 * 32 * 2);
  printf("%-6s wg/cu=%d : %.3f ms  %.1f TFLOP/s\n", name, wg_per_cu, ms, flop / ms / 1e9);
  hipFree(d);
}
int main() {
  run<1>("dep16", 1); run<2>("il2", 1); run<4>("il4", 1);
  run<1>("dep16", 2); run<2>("il2", 2); run<4>("il4", 2);
  return 0;
}
