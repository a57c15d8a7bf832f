// Micro-benchmark: does VALU work of one wave run under another wave's MFMAs on the same SIMD (gfx950)?
// Workgroup = 8 waves = 2 per SIMD. Waves 0-3 issue fp32 MFMAs (v_mfma_f32_16x16x4_f32, 3 accumulators interleaved, 48 per
// iteration); waves 4-7 run a VALU block per iteration: KIND 0 = plain fp32 FMAs (full rate), KIND 1 = transcendentals
// (v_exp_f32 + v_rcp_f32, quarter rate), KIND 2 = LDS reads. Each role alone, then both together (no barrier between them):
// together = max(alone) means the two overlap, together = sum means they do not.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, bool F16>
__global__ __launch_bounds__(512) void k(float* out, int iters, int role_mask, float a0, float b0) {
  __shared__ float lds[4096];
  if (role_mask & 4) { if ((threadIdx.x >> 6) >= 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }   // VALU waves first
  const int wv = threadIdx.x >> 6;
  lds[threadIdx.x] = a0; lds[threadIdx.x + 512] = b0;
  __syncthreads();
  float res = 0.f;
  if (wv < 4) {
    if (!(role_mask & 1)) return;
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = a0 + threadIdx.x + i; b[i] = b0 + i; }
    f16x8 ah[4], bh[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { ah[i][e] = (_Float16)(a0 + i + e); bh[i][e] = (_Float16)(b0 + e); }
    for (int it = 0; it < iters; ++it) {
      if (F16) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
          for (int r = 0; r < 3; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s & 3], bh[s & 3], acc[r], 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
          for (int r = 0; r < 3; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[r], 0, 0, 0);
      }
    }
    for (int r = 0; r < 3; ++r) for (int e = 0; e < 4; ++e) res += acc[r][e];
  } else {
    if (!(role_mask & 2)) return;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 * (threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
      if (KIND == 0) {
#pragma unroll
        for (int rep = 0; rep < 48; ++rep)
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, b0);         // 384 full-rate VALU ops = 1536 cycles
      } else if (KIND == 1) {
#pragma unroll
        for (int rep = 0; rep < 6; ++rep)
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[i]));   // 96 quarter-rate ops
      } else {
#pragma unroll
        for (int rep = 0; rep < 12; ++rep)
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += lds[(threadIdx.x * 4 + 64 * i + (int)v[i]) & 4095];
      }
    }
    for (int i = 0; i < 8; ++i) res += v[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int KIND, bool F16>
static void run(const char* name) {
  float* d; hipMalloc(&d, 1024 * 512 * 4);
  const int iters = 4000, grid = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[4];
  float ms7 = 0.f;
  {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, F16>), dim3(grid), dim3(512), 0, 0, d, iters, 7, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms7, e0, e1);
  }
  for (int mask = 1; mask <= 3; ++mask) {
    hipLaunchKernelGGL((k<KIND, F16>), dim3(grid), dim3(512), 0, 0, d, 10, mask, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, F16>), dim3(grid), dim3(512), 0, 0, d, iters, mask, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[mask], e0, e1);
  }
  printf("   (both, VALU waves at s_setprio 3: %.3f ms)\n", ms7);
  printf("%-28s MFMA wave alone %7.3f ms (%.0f cycles/iter at 2.4 GHz) | VALU wave alone %7.3f ms (%.0f) | both %7.3f ms (%.0f)  -> %s\n",
         name, ms[1], ms[1] * 1e-3 * 2.4e9 / iters, ms[2], ms[2] * 1e-3 * 2.4e9 / iters, ms[3], ms[3] * 1e-3 * 2.4e9 / iters,
         ms[3] < 0.5f * (ms[1] + ms[2]) + 0.5f * (ms[1] > ms[2] ? ms[1] : ms[2]) ? "overlap" : "no overlap");
  hipFree(d);
}
int main() {
  printf("48 x v_mfma_f32_16x16x4_f32 per iteration against:\n");
  run<0, false>("fp32 FMA (full rate)");
  run<1, false>("exp2 + rcp (quarter rate)");
  run<2, false>("LDS reads");
  printf("48 x v_mfma_f32_16x16x32_f16 per iteration against:\n");
  run<0, true>("fp32 FMA (full rate)");
  run<1, true>("exp2 + rcp (quarter rate)");
  run<2, true>("LDS reads");
  return 0;
}
