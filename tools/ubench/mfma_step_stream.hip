// mfma_step_stream.hip — what one step of backbone_lp_t_kernel's (tap, chunk) loop costs a wave, piece by piece.
// A step = the 6 v_mfma_f32_16x16x32_f16 of one (chunk, row tile): two accumulators x three split passes, on weight
// fragments W[0..3] (registers) and activation fragments U[0..1] (LDS, requested one step ahead into the other buffer).
// Cycles per MFMA by s_memtime (shader cycles), one workgroup per CU, 1 or 2 waves per SIMD, for the streams
//   0  bare: 6 MFMAs per step, 14 accumulators rotating (7 row tiles x 2 column tiles)
//   1  + s_waitcnt lgkmcnt(0) at the head and two ds_read_b128 after the first MFMA
//   2  + two scalar liveness tests (s_bitcmp + s_cbranch, never taken)
//   3  stream 1 with the ds_reads BEFORE the first MFMA (the round-3 order)
//   4  stream 1 with the MFMAs of two row tiles interleaved (4 accumulators per step of 12)
//   5  stream 2 + the weight prefetch of the kernel every 7 steps (4 global_load_dwordx4 into the other W set, vmcnt(4))
//   6  stream 5 where the second wave of each SIMD owns 6 row tiles (6 steps per chunk), like row group 1
//   7  stream 5 without the liveness tests
//   16    stream 14 + the kernel's entry head every 7 steps: the next weight tile requested from global memory into the idle
//         register set (the MFMAs' B operands alternate between the two sets), a schedule word read from LDS
//   12-15 the exact-fp32 backbone's step (14: the fragment address clamped per step as in the kernel, requests after the group;
//         15: the same requests after the FIRST MFMA of the group): 16 v_mfma_f32_16x16x4_f32 (two accumulators x 8 k-steps) bare / with its wait,
//         liveness test and two ds_read_b128 of the next row tile's fragment
//   8-11  the kernel's partly-live step (requests first, ONE test, wait only in a live step, weight prefetch as in 5) with
//         7 / 5 / 3 / 1 of the wave's 7 row tiles live
// Standalone: hipcc --offload-arch=gfx950 -O3 -o mfma_step_stream mfma_step_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define MF(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
#define FENCE __builtin_amdgcn_sched_barrier(0);

template <int STREAM>
__global__ __launch_bounds__(512) void stream_kernel(float* out, unsigned long long* cyc, int iters, int live, const h8* wts) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 210 * 272];
  for (int e = threadIdx.x; e < (int)sizeof(lds) / 4; e += blockDim.x) reinterpret_cast<float*>(lds)[e] = 0.001f * (e & 255);
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const char* xa = lds + j * 272 + 16 * g;
  h8 W[4];
  for (int q = 0; q < 4; ++q) for (int e = 0; e < 8; ++e) W[q][e] = (_Float16)(0.01f * (q + 1) + 0.001f * e + 0.0001f * lane);
  f4 acc[7][2];
  for (int r = 0; r < 7; ++r) { acc[r][0] = f4{0, 0, 0, 0}; acc[r][1] = f4{0, 0, 0, 0}; }
  h8 ua[2], ub[2];
  ua[0] = *reinterpret_cast<const h8*>(xa); ua[1] = *reinterpret_cast<const h8*>(xa + 57120);
  ub[0] = ua[0]; ub[1] = ua[1];
  int lv = __builtin_amdgcn_readfirstlane(live);
  const unsigned long long t0 = __builtin_readcyclecounter();
#define XLOAD(R, V) { V[0] = *reinterpret_cast<const h8*>(xa + (R) * 32 * 272); V[1] = *reinterpret_cast<const h8*>(xa + 57120 + (R) * 32 * 272); }
#define SIX_TAIL(R, U) MF(W[2], U[0], acc[R][1]); MF(W[1], U[0], acc[R][0]); MF(W[3], U[0], acc[R][1]); MF(W[0], U[1], acc[R][0]); MF(W[2], U[1], acc[R][1]);
#define STEP(R, U, LOADNEXT)                                                                  \
  FENCE                                                                                       \
  if constexpr (STREAM >= 1) __builtin_amdgcn_s_waitcnt(0xC07F);                              \
  if constexpr (STREAM == 3) { LOADNEXT FENCE }                                               \
  if constexpr (STREAM == 2) {                                                                \
    int l_ = lv; asm volatile("" : "+s"(l_));                                                 \
    if (l_ & (1 << (R))) MF(W[0], U[0], acc[R][0]);                                           \
    FENCE LOADNEXT FENCE                                                                      \
    asm volatile("" : "+s"(l_));                                                              \
    if (l_ & (1 << (R))) { SIX_TAIL(R, U) }                                                   \
  } else {                                                                                    \
    MF(W[0], U[0], acc[R][0]);                                                                \
    FENCE if constexpr (STREAM == 1) { LOADNEXT } FENCE                                       \
    SIX_TAIL(R, U)                                                                            \
  }                                                                                           \
  FENCE
#define PAIR(R, U, V, LOADNEXT)                                                               \
  FENCE __builtin_amdgcn_s_waitcnt(0xC07F);                                                   \
  MF(W[0], U[0], acc[R][0]); MF(W[2], U[0], acc[R][1]); MF(W[0], V[0], acc[R + 1][0]); MF(W[2], V[0], acc[R + 1][1]); \
  MF(W[1], U[0], acc[R][0]); MF(W[3], U[0], acc[R][1]); MF(W[1], V[0], acc[R + 1][0]); MF(W[3], V[0], acc[R + 1][1]); \
  MF(W[0], U[1], acc[R][0]); MF(W[2], U[1], acc[R][1]); MF(W[0], V[1], acc[R + 1][0]); MF(W[2], V[1], acc[R + 1][1]); \
  FENCE LOADNEXT FENCE
  if constexpr (STREAM >= 5) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const h8* wsrc = wts + ((w & 3) * 64 + lane) * 4;
    h8 WB[4];
    for (int q = 0; q < 4; ++q) WB[q] = W[q];
    int tile = 0;
#define WPF(WN) { const h8* src_ = wsrc + (size_t)tile * 1024; _Pragma("unroll") for (int q = 0; q < 4; ++q) WN[q] = src_[q]; \
                  tile = tile + 9 < 700 ? tile + 9 : tile - 690; } __builtin_amdgcn_s_waitcnt(0x0F74);
#define STEPW(R, U, WS, LOADNEXT)                                                             \
  FENCE __builtin_amdgcn_s_waitcnt(0xC07F);                                                   \
  if constexpr (STREAM == 7) {                                                                \
    MF(WS[0], U[0], acc[R][0]); FENCE LOADNEXT FENCE                                          \
    MF(WS[2], U[0], acc[R][1]); MF(WS[1], U[0], acc[R][0]); MF(WS[3], U[0], acc[R][1]); MF(WS[0], U[1], acc[R][0]); MF(WS[2], U[1], acc[R][1]); \
  } else {                                                                                    \
    int l_ = lv; asm volatile("" : "+s"(l_));                                                 \
    if (l_ & (1 << (R))) MF(WS[0], U[0], acc[R][0]);                                          \
    FENCE LOADNEXT FENCE                                                                      \
    asm volatile("" : "+s"(l_));                                                              \
    if (l_ & (1 << (R))) { MF(WS[2], U[0], acc[R][1]); MF(WS[1], U[0], acc[R][0]); MF(WS[3], U[0], acc[R][1]); MF(WS[0], U[1], acc[R][0]); MF(WS[2], U[1], acc[R][1]); } \
  }                                                                                           \
  FENCE
#define STEPL(R, U, WS, LOADNEXT)                                                             \
  FENCE LOADNEXT FENCE                                                                        \
  { int l_ = lv; asm volatile("" : "+s"(l_));                                                 \
    if (l_ & (1 << (R))) { __builtin_amdgcn_s_waitcnt(0xC27F);                                \
      MF(WS[0], U[0], acc[R][0]); MF(WS[2], U[0], acc[R][1]); MF(WS[1], U[0], acc[R][0]); MF(WS[3], U[0], acc[R][1]); MF(WS[0], U[1], acc[R][0]); MF(WS[2], U[1], acc[R][1]); } } \
  FENCE
    if constexpr (STREAM >= 8) {
      for (int it = 0; it < iters; ++it) {
        WPF(WB)
        STEPL(0, ua, W, XLOAD(1, ub)) STEPL(1, ub, W, XLOAD(2, ua)) STEPL(2, ua, W, XLOAD(3, ub)) STEPL(3, ub, W, XLOAD(4, ua))
        STEPL(4, ua, W, XLOAD(5, ub)) STEPL(5, ub, W, XLOAD(6, ua)) STEPL(6, ua, W, XLOAD(0, ub))
        WPF(W)
        STEPL(0, ub, WB, XLOAD(1, ua)) STEPL(1, ua, WB, XLOAD(2, ub)) STEPL(2, ub, WB, XLOAD(3, ua)) STEPL(3, ua, WB, XLOAD(4, ub))
        STEPL(4, ub, WB, XLOAD(5, ua)) STEPL(5, ua, WB, XLOAD(6, ub)) STEPL(6, ub, WB, XLOAD(0, ua))
      }
    } else {
    const bool six = STREAM == 6 && w >= 4;
    for (int it = 0; it < iters; ++it) {
      if (!six) {
        WPF(WB)
        STEPW(0, ua, W, XLOAD(1, ub)) STEPW(1, ub, W, XLOAD(2, ua)) STEPW(2, ua, W, XLOAD(3, ub)) STEPW(3, ub, W, XLOAD(4, ua))
        STEPW(4, ua, W, XLOAD(5, ub)) STEPW(5, ub, W, XLOAD(6, ua)) STEPW(6, ua, W, XLOAD(0, ub))
        WPF(W)
        STEPW(0, ub, WB, XLOAD(1, ua)) STEPW(1, ua, WB, XLOAD(2, ub)) STEPW(2, ub, WB, XLOAD(3, ua)) STEPW(3, ua, WB, XLOAD(4, ub))
        STEPW(4, ub, WB, XLOAD(5, ua)) STEPW(5, ua, WB, XLOAD(6, ub)) STEPW(6, ub, WB, XLOAD(0, ua))
      } else {
        WPF(WB)
        STEPW(0, ua, W, XLOAD(1, ub)) STEPW(1, ub, W, XLOAD(2, ua)) STEPW(2, ua, W, XLOAD(3, ub)) STEPW(3, ub, W, XLOAD(4, ua))
        STEPW(4, ua, W, XLOAD(5, ub)) STEPW(5, ub, W, XLOAD(0, ua))
        WPF(W)
        STEPW(0, ua, WB, XLOAD(1, ub)) STEPW(1, ub, WB, XLOAD(2, ua)) STEPW(2, ua, WB, XLOAD(3, ub)) STEPW(3, ub, WB, XLOAD(4, ua))
        STEPW(4, ua, WB, XLOAD(5, ub)) STEPW(5, ub, WB, XLOAD(0, ua))
      }
    }
    }
  } else
  for (int it = 0; it < iters; ++it) {
    if constexpr (STREAM == 4) {
      h8 uc[2] = {ua[0], ua[1]}, ud[2] = {ub[0], ub[1]};
      PAIR(0, ua, ub, XLOAD(2, uc) XLOAD(3, ud))
      PAIR(2, uc, ud, XLOAD(4, ua) XLOAD(5, ub))
      PAIR(4, ua, ub, XLOAD(0, uc) XLOAD(1, ud))
      PAIR(0, uc, ud, XLOAD(2, ua) XLOAD(3, ub))
      PAIR(2, ua, ub, XLOAD(4, uc) XLOAD(5, ud))
      PAIR(4, uc, ud, XLOAD(0, ua) XLOAD(1, ub))
    } else {
      STEP(0, ua, XLOAD(1, ub))
      STEP(1, ub, XLOAD(2, ua))
      STEP(2, ua, XLOAD(3, ub))
      STEP(3, ub, XLOAD(4, ua))
      STEP(4, ua, XLOAD(5, ub))
      STEP(5, ub, XLOAD(6, ua))
      STEP(6, ua, XLOAD(0, ub))
      STEP(0, ub, XLOAD(1, ua))
      STEP(1, ua, XLOAD(2, ub))
      STEP(2, ub, XLOAD(3, ua))
      STEP(3, ua, XLOAD(4, ub))
      STEP(4, ub, XLOAD(5, ua))
      STEP(5, ua, XLOAD(6, ub))
      STEP(6, ub, XLOAD(0, ua))
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
  for (int r = 0; r < 7; ++r) s += acc[r][0][0] + acc[r][1][1] + acc[r][0][2] + acc[r][1][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int STREAM>
__global__ __launch_bounds__(512) void stream_f32_kernel(float* out, unsigned long long* cyc, int iters, int live, const float* wts) {
  __shared__ __attribute__((aligned(16))) float lds[210 * 132];
  for (int e = threadIdx.x; e < 210 * 132; e += blockDim.x) lds[e] = 0.001f * (e & 255);
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const float* xa = lds + j * 132 + 8 * g;
  float bf0[8], bf1[8];
  for (int e = 0; e < 8; ++e) { bf0[e] = 0.01f * e + 0.0001f * lane; bf1[e] = 0.02f * e + 0.0002f * lane; }
  f4 acc[7][2];
  for (int r = 0; r < 7; ++r) { acc[r][0] = f4{0, 0, 0, 0}; acc[r][1] = f4{0, 0, 0, 0}; }
  float4 ua[2], ub[2];
  ua[0] = *reinterpret_cast<const float4*>(xa); ua[1] = *reinterpret_cast<const float4*>(xa + 4); ub[0] = ua[0]; ub[1] = ua[1];
  int lv = __builtin_amdgcn_readfirstlane(live);
  const unsigned long long t0 = __builtin_readcyclecounter();
#define XLF(R, V) { V[0] = *reinterpret_cast<const float4*>(xa + (R) * 32 * 132); V[1] = *reinterpret_cast<const float4*>(xa + (R) * 32 * 132 + 4); }
#define MF4(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, C, 0, 0, 0)
#define GROUP16(R, U)                                                                         \
  _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                             \
    MF4(U[q].x, bf0[4 * q], acc[R][0]); MF4(U[q].x, bf1[4 * q], acc[R][1]);                   \
    MF4(U[q].y, bf0[4 * q + 1], acc[R][0]); MF4(U[q].y, bf1[4 * q + 1], acc[R][1]);           \
    MF4(U[q].z, bf0[4 * q + 2], acc[R][0]); MF4(U[q].z, bf1[4 * q + 2], acc[R][1]);           \
    MF4(U[q].w, bf0[4 * q + 3], acc[R][0]); MF4(U[q].w, bf1[4 * q + 3], acc[R][1]); }
#define XLC(R, V) { const int o_ = min(max(arow + dby + (R) * (32 * 132 * 4), alo), ahi);                       \
                    const float4* ap_ = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lds) + o_); V[0] = ap_[0]; V[1] = ap_[1]; }
#define STEPF(R, U, LOADNEXT)                                                                 \
  FENCE                                                                                       \
  if constexpr (STREAM == 15) {                                                               \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                       \
    MF4(U[0].x, bf0[0], acc[R][0]);                                                           \
    FENCE LOADNEXT FENCE                                                                      \
    MF4(U[0].x, bf1[0], acc[R][1]);                                                           \
    MF4(U[0].y, bf0[1], acc[R][0]); MF4(U[0].y, bf1[1], acc[R][1]);                           \
    MF4(U[0].z, bf0[2], acc[R][0]); MF4(U[0].z, bf1[2], acc[R][1]);                           \
    MF4(U[0].w, bf0[3], acc[R][0]); MF4(U[0].w, bf1[3], acc[R][1]);                           \
    MF4(U[1].x, bf0[4], acc[R][0]); MF4(U[1].x, bf1[4], acc[R][1]);                           \
    MF4(U[1].y, bf0[5], acc[R][0]); MF4(U[1].y, bf1[5], acc[R][1]);                           \
    MF4(U[1].z, bf0[6], acc[R][0]); MF4(U[1].z, bf1[6], acc[R][1]);                           \
    MF4(U[1].w, bf0[7], acc[R][0]); MF4(U[1].w, bf1[7], acc[R][1]);                           \
  } else if constexpr (STREAM >= 13) {                                                               \
    __builtin_amdgcn_s_waitcnt(0xC07F | (2 << 8));                                            \
    int l_ = lv; asm volatile("" : "+s"(l_));                                                 \
    if (l_ & (1 << (R))) { GROUP16(R, U) }                                                    \
    FENCE LOADNEXT                                                                            \
  } else { GROUP16(R, U) }                                                                    \
  FENCE
  const int arow = (j * 132 + 8 * g) * 4, alo = arow - (j + 1) * 132 * 4, ahi = arow + (208 - j) * 132 * 4;
  int dby = __builtin_amdgcn_readfirstlane(live) & 0x1000 ? 132 * 4 : 0;
  asm volatile("" : "+s"(dby));
  if constexpr (STREAM == 13 || STREAM == 14) { XLF(1, ub) }
  if constexpr (STREAM == 14) {
    for (int it = 0; it < iters; ++it) {
      STEPF(0, ua, XLC(2, ua)) STEPF(1, ub, XLC(3, ub)) STEPF(2, ua, XLC(4, ua)) STEPF(3, ub, XLC(5, ub))
      STEPF(4, ua, XLC(6, ua)) STEPF(5, ub, XLC(0, ub)) STEPF(6, ua, XLC(1, ua))
      STEPF(0, ub, XLC(2, ub)) STEPF(1, ua, XLC(3, ua)) STEPF(2, ub, XLC(4, ub)) STEPF(3, ua, XLC(5, ua))
      STEPF(4, ub, XLC(6, ub)) STEPF(5, ua, XLC(0, ua)) STEPF(6, ub, XLC(1, ub))
    }
  } else if constexpr (STREAM == 16 || STREAM == 17) {
    __shared__ int sch[64];
    if (threadIdx.x < 64) sch[threadIdx.x] = (threadIdx.x * 37 + 11) & 63;
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* wsrc = wts + ((wv & 3) * 32 + j) * 32 + 8 * g;
    float4 bA[4], bB[4];
    for (int q = 0; q < 4; ++q) { bA[q] = make_float4(bf0[0], bf0[1], bf0[2], bf0[3]); bB[q] = bA[q]; }
    f4 dummy[28];
    if constexpr (STREAM == 17) { for (int i = 0; i < 28; ++i) { dummy[i] = f4{bf0[i & 7], bf1[i & 7], (float)i, 1.0f}; asm volatile("" : "+v"(dummy[i])); } }
    int en = 1, tile = 0;
#define GROUP16W(R, U, BC)                                                                    \
  _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                             \
    const float w0_[4] = {BC[q].x, BC[q].y, BC[q].z, BC[q].w}, w1_[4] = {BC[2 + q].x, BC[2 + q].y, BC[2 + q].z, BC[2 + q].w}; \
    MF4(U[q].x, w0_[0], acc[R][0]); MF4(U[q].x, w1_[0], acc[R][1]);                           \
    MF4(U[q].y, w0_[1], acc[R][0]); MF4(U[q].y, w1_[1], acc[R][1]);                           \
    MF4(U[q].z, w0_[2], acc[R][0]); MF4(U[q].z, w1_[2], acc[R][1]);                           \
    MF4(U[q].w, w0_[3], acc[R][0]); MF4(U[q].w, w1_[3], acc[R][1]); }
#define STEPW16(R, U, BC, LOADNEXT)                                                           \
  FENCE __builtin_amdgcn_s_waitcnt(0xC07F | (2 << 8));                                        \
  { int l_ = lv; asm volatile("" : "+s"(l_)); if (l_ & (1 << (R))) { GROUP16W(R, U, BC) } }   \
  FENCE LOADNEXT FENCE
#define ENTRY16(UA, UB, BC, BN)                                                               \
  { const int nxt_ = en & 63; const int env_ = sch[nxt_];                                     \
    { const float* src_ = wsrc + (size_t)tile * 4096; tile = tile + 9 < 700 ? tile + 9 : tile - 690;                \
      BN[0] = *reinterpret_cast<const float4*>(src_); BN[1] = *reinterpret_cast<const float4*>(src_ + 4);           \
      BN[2] = *reinterpret_cast<const float4*>(src_ + 512); BN[3] = *reinterpret_cast<const float4*>(src_ + 516); } \
    STEPW16(0, UA, BC, XLC(2, UA)) STEPW16(1, UB, BC, XLC(3, UB))                             \
    en = __builtin_amdgcn_readfirstlane(env_);                                                \
    STEPW16(2, UA, BC, XLC(4, UA)) STEPW16(3, UB, BC, XLC(5, UB)) STEPW16(4, UA, BC, XLC(6, UA)) \
    STEPW16(5, UB, BC, XLC(0, UB)) STEPW16(6, UA, BC, XLC(1, UA)) }
    XLF(1, ub)
    for (int it = 0; it < iters; ++it) { ENTRY16(ua, ub, bA, bB) ENTRY16(ub, ua, bB, bA) }
    if constexpr (STREAM == 17) { for (int i = 0; i < 28; ++i) { asm volatile("" : "+v"(dummy[i])); acc[i % 7][0] += dummy[i]; } }
  } else if constexpr (STREAM == 15) {
    for (int it = 0; it < iters; ++it) {
      STEPF(0, ua, XLC(1, ub)) STEPF(1, ub, XLC(2, ua)) STEPF(2, ua, XLC(3, ub)) STEPF(3, ub, XLC(4, ua))
      STEPF(4, ua, XLC(5, ub)) STEPF(5, ub, XLC(6, ua)) STEPF(6, ua, XLC(0, ub))
      STEPF(0, ub, XLC(1, ua)) STEPF(1, ua, XLC(2, ub)) STEPF(2, ub, XLC(3, ua)) STEPF(3, ua, XLC(4, ub))
      STEPF(4, ub, XLC(5, ua)) STEPF(5, ua, XLC(6, ub)) STEPF(6, ub, XLC(0, ua))
    }
  } else
  for (int it = 0; it < iters; ++it) {
    STEPF(0, ua, XLF(2, ua)) STEPF(1, ub, XLF(3, ub)) STEPF(2, ua, XLF(4, ua)) STEPF(3, ub, XLF(5, ub))
    STEPF(4, ua, XLF(6, ua)) STEPF(5, ub, XLF(0, ub)) STEPF(6, ua, XLF(1, ua))
    STEPF(0, ub, XLF(2, ub)) STEPF(1, ua, XLF(3, ua)) STEPF(2, ub, XLF(4, ub)) STEPF(3, ua, XLF(5, ua))
    STEPF(4, ub, XLF(6, ub)) STEPF(5, ua, XLF(0, ua)) STEPF(6, ub, XLF(1, ub))
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sres = 0.0f;
  for (int r = 0; r < 7; ++r) sres += acc[r][0][0] + acc[r][1][1] + acc[r][0][2] + acc[r][1][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sres;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int STREAM>
void run_f32(const char* name, int threads) {
  float* out; unsigned long long* cyc;
  const int iters = 1000, nw = threads / 64;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* wts; CK(hipMalloc(&wts, 720 * 4096 * 4)); CK(hipMemset(wts, 0, 720 * 4096 * 4));
  stream_f32_kernel<STREAM><<<256, threads>>>(out, cyc, iters, 0x7f, wts);
  CK(hipEventRecord(e0));
  stream_f32_kernel<STREAM><<<256, threads>>>(out, cyc, iters, 0x7f, wts);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[256 * 8];
  CK(hipMemcpy(h, cyc, 256 * nw * 8, hipMemcpyDeviceToHost));
  double m0 = 0.0, m1 = 0.0;
  for (int i = 0; i < 256 * nw; ++i) { if (i % nw < 4) m0 += (double)h[i]; else m1 += (double)h[i]; }
  m0 /= 256 * 4; m1 /= 256 * 4;
  const double mfma_per_wave = (double)iters * 14 * 16, last = nw == 8 ? (m0 > m1 ? m0 : m1) : m0;
  printf("%-58s %d wave(s)/SIMD: %5.1f ticks per MFMA of the SIMD ; launch %.3f ms, tick rate %.2f GHz", name, nw / 4, last / (mfma_per_wave * (nw / 4)), ms, last / (ms * 1e6));
  if (nw == 8) printf(" ; older waves done after %.0f %% of it", 100.0 * (m0 < m1 ? m0 : m1) / last);
  printf("\n");
  CK(hipFree(out)); CK(hipFree(cyc));
}

template <int STREAM>
void run(const char* name, int threads, int live = 0x7f) {
  float* out; unsigned long long* cyc; h8* wts;
  CK(hipMalloc(&wts, 720 * 1024 * sizeof(h8))); CK(hipMemset(wts, 0, 720 * 1024 * sizeof(h8)));
  const int iters = 4000, nw = threads / 64;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  stream_kernel<STREAM><<<256, threads>>>(out, cyc, iters, live, wts);
  CK(hipEventRecord(e0));
  stream_kernel<STREAM><<<256, threads>>>(out, cyc, iters, live, wts);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[256 * 8];
  CK(hipMemcpy(h, cyc, 256 * nw * 8, hipMemcpyDeviceToHost));
  double mean = 0.0, m0 = 0.0, m1 = 0.0;
  for (int i = 0; i < 256 * nw; ++i) { mean += (double)h[i]; if (i % nw < 4) m0 += (double)h[i]; else m1 += (double)h[i]; }
  mean /= 256 * nw; m0 /= 256 * 4; m1 /= 256 * 4;
  const double mfma_per_wave = (double)iters * (STREAM == 4 ? 72 : STREAM == 6 ? 78 : 12 * __builtin_popcount(live));


  // the launch ends with the slowest wave: ticks per MFMA of the SIMD and the tick rate come from the LAST wave of a SIMD to finish
  const double last = nw == 8 ? (m0 > m1 ? m0 : m1) : m0;
  printf("%-58s %d wave(s)/SIMD: %5.1f ticks per MFMA of the SIMD ; launch %.3f ms = %.2f ns per MFMA of the SIMD, tick rate %.2f GHz",
         name, nw / 4, last / (mfma_per_wave * (nw / 4)), ms, ms * 1e6 / (mfma_per_wave * (nw / 4)), last / (ms * 1e6));
  if (nw == 8) printf(" ; older waves done after %.0f %% of it", 100.0 * (m0 < m1 ? m0 : m1) / last);
  if (STREAM >= 5) printf(" ; %.0f ticks per 7-step chunk", last / (2.0 * iters));
  printf("\n");
  CK(hipFree(out)); CK(hipFree(cyc)); CK(hipFree(wts));
}

int main() {
  for (int threads = 256; threads <= 512; threads += 256) {
    run<0>("0 bare 6-MFMA steps, 14 accumulators", threads);
    run<1>("1 + lgkmcnt(0) + 2 ds_read_b128 after the first MFMA", threads);
    run<2>("2 + two scalar liveness tests per step", threads);
    run<3>("3 as 1, ds_reads before the first MFMA", threads);
    run<4>("4 two row tiles per step (4 accumulators, 12 MFMAs)", threads);
    run<5>("5 stream 2 + weight prefetch per 7 steps", threads);
    run<6>("6 stream 5, waves 4-7 with 6 steps per chunk", threads);
    run<7>("7 stream 5 without liveness tests", threads);
    run<8>("8 partly-live step, 7 of 7 live", threads, 0x7f);
    run<8>("9 partly-live step, 5 of 7 live", threads, 0x1f);
    run<8>("10 partly-live step, 3 of 7 live", threads, 0x07);
    run<8>("11 partly-live step, 1 of 7 live", threads, 0x01);
    run_f32<12>("12 fp32: bare 16-MFMA steps (16x16x4), 14 accumulators", threads);
    run_f32<13>("13 fp32: + wait, liveness test, 2 ds_read_b128", threads);
    run_f32<14>("14 fp32: 13 with the clamped address computed per step", threads);
    run_f32<15>("15 fp32: requests (clamped) after the first MFMA, no test", threads);
    run_f32<16>("16 fp32: 14 + entry head (weight tile from global, schedule word)", threads);
    run_f32<17>("17 fp32: 16 with 112 more live VGPRs", threads);
  }
  return 0;
}
