// permlane_swap_check.hip — v + (v of lane ^ 16), then ^ 32, by v_permlane16_swap / v_permlane32_swap (inline asm) against __shfl_xor: bit-equal on
// gfx950. (The __builtin_amdgcn_permlane16_swap / 32_swap builtins of this compiler return their first result twice: 128 mismatches.)
// Standalone: hipcc --offload-arch=gfx950 -O3 -o permlane_swap_check permlane_swap_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float xor_sum16(float s) {
  unsigned a = __builtin_bit_cast(unsigned, s), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float xor_sum32(float s) {
  unsigned a = __builtin_bit_cast(unsigned, s), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__global__ void k(float* o) {
  float s = o[threadIdx.x];
  float t = xor_sum16(s);
  float u = xor_sum32(t);
  o[64 + threadIdx.x] = t; o[128 + threadIdx.x] = u;
  o[192 + threadIdx.x] = s + __shfl_xor(s, 16, 64);
  float v = s + __shfl_xor(s, 16, 64);
  o[256 + threadIdx.x] = v + __shfl_xor(v, 32, 64);
}
int main() {
  float h[320]; for (int i = 0; i < 64; ++i) h[i] = 1.0f + i * 0.37f + (i % 7) * 1e-3f;
  float* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) { if (h[64 + i] != h[192 + i]) ++bad; if (h[128 + i] != h[256 + i]) ++bad; }
  printf("mismatches %d ; lane0 %g %g %g %g\n", bad, h[64], h[192], h[128], h[256]);
  return 0;
}
