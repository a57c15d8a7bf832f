// svdd_lp_common.h — shared by svdd_lp_backbone.hip, svdd_lp_tower.hip, svdd_lp_gru_tail.hip: split-precision variants of the net kernels of svdd_nets.hip on the 16-bit matrix cores of gfx950
// (v_mfma_f32_16x16x32_f16 / _bf16, fp32 accumulate, 16x the rate of the exact-fp32 MFMA).
//
// The exact-fp32 kernels stay the default and the parity reference. These are an explicit opt-in
// (Diffusion.precision, svdd_amd/fused.py), in four modes:
//     f16x3 / bf16x3   every fp32 operand v is split on the fly into hi = rn16(v), lo = rn16(v - hi); a product
//                      a*b is evaluated as ahi*bhi + ahi*blo + alo*bhi on the matrix cores (3 MFMAs, fp32 accumulate;
//                      the dropped alo*blo term is below fp32 round-off for f16, ~2^-17 relative for bf16).
//                      Measured on the device (tools/ubench/mfma_lp_probe.hip, profiles/r02_mfma_lp_probe.txt), K = 1152:
//                      error 1.5e-7 (f16x3) / 5.3e-7 (bf16x3) of sum|a b|, against 1.8e-7 for the fp32 MFMA chain itself —
//                      the 16-bit MFMA accumulates the 32 products of an instruction exactly and rounds once.
//     f16 / bf16       one pass on hi only (plain 16-bit operands), error ~3e-5 / ~3e-4 of sum|a b|.
// f16 operands are pre-scaled by powers of two (exact) so that the lo parts stay normal numbers and nothing
// overflows: activations by `sa` at the LayerNorm write, weights by the host; the accumulator is scaled back by
// `inv` = 1 / (sa * sw) in the epilogue. For bf16 all scales are 1.
//
// Everything that is not a matrix product (first-layer table lookup, LayerNorm statistics, residual stream, ReLU,
// biases, the 128 -> 5 output map) is computed in fp32 exactly as in the fp32 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "svdd_hip.h"
#include "svdd_spt.h"

extern "C" void svdd_internal_timed_events(int k, hipEvent_t* e0, hipEvent_t* e1);   // svdd_kernels.hip (profiling)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));

template <typename T> struct Lp;
template <> struct Lp<_Float16> {
  typedef h8 V8; typedef h2 V2;
  static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Lp<__bf16> {
  typedef b8 V8; typedef b2 V2;
  static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

// hi/lo split of two adjacent channels, packed for one 4-byte LDS store each
template <typename T>
__device__ __forceinline__ void split2(float v0, float v1, typename Lp<T>::V2& hi, typename Lp<T>::V2& lo) {
  const T h0 = (T)v0, h1 = (T)v1;
  hi[0] = h0; hi[1] = h1;
  lo[0] = (T)(v0 - (float)h0); lo[1] = (T)(v1 - (float)h1);
}

template <int N>
__device__ __forceinline__ float row_ror(float v) {            // rotate right by N inside each 16-lane DPP row
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float group16_sum(float v) {
  v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
  return v;
}

constexpr int TW_ROWS = 208;               // 13 row tiles of 16
constexpr int TW_RT = 13;
constexpr int BB_C = 128;
constexpr int BB_AP = BB_C + 4;            // fp32 row stride of the final-stage image
constexpr int BB_MAXL = 32;
constexpr int LPS = 144;                   // 16-bit row stride of an operand plane (288 B: conflict-free ds_read_b128)
constexpr int LPSB = LPS * 2;
constexpr int PLANE_B = (TW_ROWS + 2) * LPSB;              // 60,480 B: rows -1 .. TW_ROWS
constexpr int IMG_REGION_B = 2 * PLANE_B;                  // >= the fp32 final-stage image (208 x 132 x 4 = 109,824 B)

// arguments of the split-precision backbone kernels (svdd_lp_backbone.hip, svdd_lp_backbone2.hip)
struct BackboneLpArgs {
  const uint8_t* x;        // [n, L] tokens 0..4
  const float* table0;     // [9][5][128]
  const void* tiles;       // [nl*36 + 4] tiles of [4 cg][64 lanes][2 ct][NPARTS][8] 16-bit
  const float* vec;        // [nl + 2][4][128] as in svdd_backbone_cnn_f32
  const float* lscale;     // [nl + 1][2] = {sa: activation scale, inv: 1 / (sa * weight scale)}
  const float* w2;         // [5][128] then b2 [5]
  float* out;              // [n, L, 5]
  int n, L, spt, nl;
  int dil[BB_MAXL];
  const int* count;        // device scalar: valid rows (NULL: n) — exact work-skipping on a compacted batch
  const int* row_idx;      // [count] (NULL: identity): compact row r reads the tokens of sequence row_idx[r] of x ...
  int out_scatter;         // ... and writes its logits to row row_idx[r] of out (1) or to row r (0)
  int auto_spt, ncu;       // auto_spt: the workgroups pick the sequences per tile from the device-side row count (svdd_spt.h)
  SvddTilePlan plan;       // several sequences per tile (interleaved, backbone_lp_t_kernel<.., IL = true>): which tile takes how many
  int fixed_half;          // ... and the cost model's fixed part for a plan made on the device
};

}  // namespace
