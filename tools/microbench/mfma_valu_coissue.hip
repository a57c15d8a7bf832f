// Do fp32 MFMAs of one wave and vector instructions of ANOTHER wave on the same SIMD run concurrently on gfx950?
// Workgroup = 8 waves on one CU (2 per SIMD): waves 0-3 issue `nm` v_mfma_f32_16x16x4_f32 (4 independent accumulators), waves 4-7 issue
// `nv` instructions of one kind. Three launches per kind: MFMA waves only, vector waves only, both; if the two kinds of work overlap,
// t(both) ~ max(t(mfma), t(vec)); if they serialise, t(both) ~ t(mfma) + t(vec).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/microbench/mfma_valu_coissue.hip && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
static int g_swap = 0, g_half = 0;

template <int KIND, bool SWAP, bool HALF>
__global__ __launch_bounds__(512) void k(float* out, int nm, int nv, int run_m, int run_v) {
  const int wv = threadIdx.x >> 6;
  float r = 0.0f;
  if (SWAP ? wv >= 4 : wv < 4) {                 // SWAP: the MFMA waves are the YOUNGER four
    if (!run_m) return;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
    if (HALF) {                                  // 16-bit MFMA (v_mfma_f32_16x16x32_f16: 8 passes)
      h8 hx, hy;
      for (int e = 0; e < 8; ++e) { hx[e] = (_Float16)x; hy[e] = (_Float16)y; }
      for (int i = 0; i < nm; i += 4) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hx, hy, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hx, hy, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hx, hy, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hx, hy, a3, 0, 0, 0);
      }
    } else {
      for (int i = 0; i < nm; i += 4) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      }
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    if (!run_v) return;
    __shared__ float sm[4][64 * 4];
    float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    float* my = &sm[wv & 3][(threadIdx.x & 63)];
    my[0] = v0; my[64] = v1; my[128] = v2; my[192] = v3;
    for (int i = 0; i < nv; i += 4) {
      if (KIND == 0) {          // fp32 FMA
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      } else if (KIND == 1) {   // integer add
        asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
      } else if (KIND == 2) {   // fp32 add
        asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      } else if (KIND == 3) {   // transcendental
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      } else if (KIND == 4) {   // LDS read
        v0 += my[0]; v1 += my[64]; v2 += my[128]; v3 += my[192];
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      } else if (KIND == 5) {   // v_max (the ReLU)
        asm volatile("v_max_f32 %0, %0, %0\n v_max_f32 %1, %1, %1\n v_max_f32 %2, %2, %2\n v_max_f32 %3, %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      } else if (KIND == 6) {   // packed fp32 FMA
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p0 = {v0, v1}, p1 = {v2, v3};
        asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1" : "+v"(p0), "+v"(p1));
        v0 = p0[0]; v1 = p0[1]; v2 = p1[0]; v3 = p1[1];
      } else {                  // v_mov
        asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      }
    }
    r = v0 + v1 + v2 + v3 + (float)(i0 + i1 + i2 + i3);
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

// ONE wave per SIMD issuing both kinds itself: per MFMA, q independent v_fma_f32 behind it
template <int Q>
__global__ __launch_bounds__(256) void same_wave(float* out, int nm, int with_mfma, int with_valu) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
  const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
  for (int i = 0; i < nm; i += 2) {
    if (with_mfma) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    if (with_valu) for (int q = 0; q < Q; q += 4) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    if (with_mfma) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    if (with_valu) for (int q = 0; q < Q; q += 4) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
  }
  const float r = a0[0] + a1[1] + v0 + v1 + v2 + v3;
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int Q>
void run_same(float* out, int nm) {
  float t[3];
  for (int mode = 0; mode < 3; ++mode) {
    const int rm = mode != 1, rv = mode != 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(same_wave<Q>, dim3(256), dim3(256), 0, 0, out, nm, rm, rv);
    hipEventRecord(e0, 0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(same_wave<Q>, dim3(256), dim3(256), 0, 0, out, nm, rm, rv);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&t[mode], e0, e1); t[mode] /= 5;
  }
  printf("same wave, %d v_fma_f32 per MFMA: mfma only %.1f us | vector only %.1f us | both %.1f us  -> both / (mfma + vec) = %.2f, both / max = %.2f\n",
         Q, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[2] / (t[0] + t[1]), t[2] / (t[0] > t[1] ? t[0] : t[1]));
}

template <int KIND, bool SWAP = false, bool HALF = false>
void run(const char* name, float* out, int nm, int nv) {
  float t[3];
  for (int mode = 0; mode < 3; ++mode) {
    const int rm = mode != 1, rv = mode != 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<KIND, SWAP, HALF>), dim3(256), dim3(512), 0, 0, out, nm, nv, rm, rv);
    hipEventRecord(e0, 0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<KIND, SWAP, HALF>), dim3(256), dim3(512), 0, 0, out, nm, nv, rm, rv);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&t[mode], e0, e1); t[mode] /= 5;
  }
  printf("%s%s%-14s nm=%d nv=%d: mfma only %.1f us | vector only %.1f us | both %.1f us  -> both / (mfma + vec) = %.2f, both / max = %.2f\n",
         SWAP ? "[mfma waves younger] " : "", HALF ? "[f16 16x16x32 mfma] " : "", name, nm, nv, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[2] / (t[0] + t[1]), t[2] / (t[0] > t[1] ? t[0] : t[1]));
}

int main() {
  float* out; hipMalloc(&out, 4096);
  const int nm = 1 << 15;                       // 32768 MFMAs x 32 cycles ~ 1.05 M cycles ~ 0.44 ms
  run<0>("v_fma_f32", out, nm, 1 << 18);        // 262144 x 4 cycles ~ the same
  run<2>("v_add_f32", out, nm, 1 << 18);
  run<5>("v_max_f32", out, nm, 1 << 18);
  run<6>("v_pk_fma_f32", out, nm, 1 << 17);
  run<1>("v_add_u32", out, nm, 1 << 18);
  run<7>("v_mov_b32", out, nm, 1 << 18);
  run<3>("v_exp_f32", out, nm, 1 << 16);
  run<4>("ds_read_b32", out, nm, 1 << 16);
  run<0, true>("v_fma_f32", out, nm, 1 << 18);
  run<1, true>("v_add_u32", out, nm, 1 << 18);
  run<4, true>("ds_read_b32", out, nm, 1 << 16);
  run<0, false, true>("v_fma_f32", out, nm, 1 << 18);
  run<1, false, true>("v_add_u32", out, nm, 1 << 18);
  run<4, false, true>("ds_read_b32", out, nm, 1 << 16);
  run_same<4>(out, nm);
  run_same<8>(out, nm);
  return 0;
}
