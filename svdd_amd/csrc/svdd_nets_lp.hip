// svdd_nets_lp.hip — split-precision variants of the net kernels of svdd_nets.hip on the 16-bit matrix cores of gfx950
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
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "svdd_hip.h"

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

// ---------------------------------------------------------------------------------- backbone, split precision ----
// Same decomposition as backbone_kernel (svdd_nets.hip): one workgroup (8 waves) per tile of whole sequences; wave w
// owns 32 output channels (column group w & 3) of the row tiles of parity w >> 2 and keeps its part of the residual
// stream in registers in the MFMA C/D layout; the LayerNorm'd image lives in LDS as TWO 16-bit planes (hi, lo) and
// feeds the A operands directly; weights stream from L2 into the B operands one (layer, chunk, tap) tile ahead.
// Differences that the 16-bit MFMA brings:
//   * one v_mfma_f32_16x16x32 covers a whole 32-channel chunk (K = 32), so a (row tile, tap, chunk) costs 2 column
//     tiles x NP instructions instead of 16;
//   * a lane's two output channels are ADJACENT (32 cg + 2 j, + 1) so that the hi (lo) halves of both go to LDS in one
//     4-byte store;
//   * the weight tile is stored by the host in exactly the order the lanes consume it: [cg][lane][ct][hi|lo][8]
//     (64 B per lane, 4 KB per wave, fully coalesced).
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
};

template <typename T, int NP, bool SPT1>
__global__ __launch_bounds__(512, 2) void backbone_lp_kernel(BackboneLpArgs a) {
  typedef typename Lp<T>::V8 V8;
  typedef typename Lp<T>::V2 V2;
  constexpr int NPARTS = NP == 3 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  char* plane = smem_b + LPSB;                               // byte address of (row 0, channel 0) of the hi plane
  float* img32 = reinterpret_cast<float*>(smem_b);           // final stage: fp32 image [TW_ROWS][BB_AP] over the planes
  float* Bs = reinterpret_cast<float*>(smem_b + IMG_REGION_B);   // [9][5][128] the first layer's lookup table
  float* psum = Bs + 9 * 5 * BB_C;                           // [4][TW_ROWS]
  float* rstat = psum + 4 * TW_ROWS;                         // [TW_ROWS]
  int* toks = reinterpret_cast<int*>(rstat + TW_ROWS);       // [TW_ROWS]
  int* rpos = toks + TW_ROWS;                                // [TW_ROWS]
  int* sdil = rpos + TW_ROWS;                                // [BB_MAXL + 1]
  int* sched = sdil + BB_MAXL + 1;                           // [(nl + 1) * 36]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = w & 3, rh = w >> 2;
  const int j = lane & 15, g = lane >> 4;
  const int c0 = 32 * cg + 2 * j;                            // this lane's output channels: c0 and c0 + 1
  const int L = a.L;
  const int tile_rows = a.spt * L;
  const int64_t row0 = (int64_t)blockIdx.x * tile_rows;
  const int64_t total_rows = (int64_t)a.n * L;
  const int nl = a.nl;
  const int it_end = (nl + 1) * 36;
  constexpr int NR = 7;

  for (int e = tid; e < TW_ROWS; e += 512) {
    toks[e] = (e < tile_rows && row0 + e < total_rows) ? a.x[row0 + e] : -1;
    rpos[e] = e < tile_rows ? e % L : -(1 << 20);
  }
  // zero rows -1 and TW_ROWS of both planes (a tap that leaves the sequence reads them)
  for (int e = tid; e < LPSB / 4; e += 512) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      reinterpret_cast<int*>(smem_b + p * PLANE_B)[e] = 0;
      reinterpret_cast<int*>(smem_b + p * PLANE_B + (TW_ROWS + 1) * LPSB)[e] = 0;
    }
  }
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < BB_MAXL; ++i) sdil[i] = a.dil[i];
    sdil[BB_MAXL] = 1;
  }
  for (int e = tid; e < 9 * 5 * BB_C; e += 512) Bs[e] = a.table0[e];
  __syncthreads();
  // schedule (identical to backbone_kernel): sched[k] = 0 for a tap that only sees zero padding, else
  //   bits 0-12 live row tiles ; 13-14 chunk ; 15-18 tap ; 19-28 index of the next live k
  for (int k = tid; k < it_end; k += 512) {
    auto entry = [&](int kk) {
      const int layer = kk / 36, t = kk % 9;
      if (layer >= nl) return t == 4 ? 0x1fff : 0;
      const int d = (t - 4) * sdil[layer];
      const int lo = d < 0 ? -d : 0, hi = d > 0 ? L - d : L;
      if (lo >= hi) return 0;
      int m = 0;
      if (SPT1) {
        for (int r = 0; r < TW_RT; ++r) if (lo < 16 * r + 16 && hi > 16 * r) m |= 1 << r;
      } else {
        for (int row = 0; row < TW_ROWS; ++row) {
          const int pp = rpos[row];
          if (pp >= lo && pp < hi) m |= 1 << (row >> 4);
        }
      }
      return m;
    };
    const int m = entry(k);
    int nx = k + 1;
    while (nx < it_end && entry(nx) == 0) ++nx;
    sched[k] = m ? (m | ((k % 36) / 9) << 13 | (k % 9) << 15 | nx << 19) : 0;
  }

  // ---- first layer: table lookup, fp32 (dnaconv.py:177,184)
  f32x4 f[NR][2], acc[NR][2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = c0 + ct;
    const float b0 = a.vec[col];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * (rh + 2 * r) + 4 * g + e;
        float v = b0;
        if (row < TW_ROWS) {
          const int pos = rpos[row];
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int p = pos + t - 4;
            const int tk = (p >= 0 && p < L) ? toks[row + t - 4] : -1;
            if (tk >= 0) v += Bs[(t * 5 + tk) * BB_C + col];
          }
        }
        f[r][ct][e] = row < tile_rows ? fmaxf(v, 0.0f) : 0.0f;
      }
  }
  __syncthreads();                                        // sched is visible

  // A operand addressing (bytes inside a plane): this lane feeds row 16 (rh + 2 r) + j, channels 32 c + 8 g .. + 8
  const int arow0 = (16 * rh + j) * LPSB + 16 * g;
  const int a_lo = arow0 - (16 * rh + j + 1) * LPSB;      // row -1
  const int a_hi = arow0 + (TW_ROWS - 16 * rh - j) * LPSB;    // row TW_ROWS
  int apos[SPT1 ? 1 : NR];
  if (!SPT1) {
#pragma unroll
    for (int r = 0; r < NR; ++r) apos[r] = 16 * (rh + 2 * r) + j < TW_ROWS ? rpos[16 * (rh + 2 * r) + j] : -(1 << 20);
  }

  // weight stream: 2 * NPARTS 16-byte pieces per lane per tile, contiguous
  constexpr int TILE_V8 = 4 * 64 * 2 * NPARTS;
  const V8* wsrc = reinterpret_cast<const V8*>(a.tiles) + (cg * 64 + lane) * (2 * NPARTS);
  auto tile_of = [&](int k) { return k < nl * 36 ? k : nl * 36 + (k - nl * 36) / 9; };
  int it = 0;
  while (it < it_end && sched[it] == 0) ++it;
  it = __builtin_amdgcn_readfirstlane(it);
  int en = __builtin_amdgcn_readfirstlane(sched[it]);
  V8 bn[2 * NPARTS];
  {
    const V8* src = wsrc + (size_t)tile_of(it) * TILE_V8;
#pragma unroll
    for (int q = 0; q < 2 * NPARTS; ++q) bn[q] = src[q];
  }

  for (int layer = 0; layer <= nl; ++layer) {             // layer == nl: the first 1x1 conv of final_conv
    const float* vl = a.vec + (size_t)(layer + 1) * 4 * BB_C;
    const float sa = a.lscale[2 * layer], inv = a.lscale[2 * layer + 1];
    if (layer < nl) {
      const float tb0 = vl[BB_C + c0], tb1 = vl[BB_C + c0 + 1];
      // pass 1: row means
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          const float sm = group16_sum((f[r][0][e] + tb0) + (f[r][1][e] + tb1));
          if (j == 0 && row < TW_ROWS) psum[cg * TW_ROWS + row] = sm;
        }
      __syncthreads();
      if (tid < TW_ROWS)
        rstat[tid] = ((psum[tid] + psum[TW_ROWS + tid]) + (psum[2 * TW_ROWS + tid] + psum[3 * TW_ROWS + tid])) * (1.0f / BB_C);
      __syncthreads();
      // pass 2: centred second moment
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          const float mean = row < TW_ROWS ? rstat[row] : 0.0f;
          const float d0 = f[r][0][e] + tb0 - mean, d1 = f[r][1][e] + tb1 - mean;
          acc[r][0][e] = d0; acc[r][1][e] = d1;
          const float sq = group16_sum(d0 * d0 + d1 * d1);
          if (j == 0 && row < TW_ROWS) psum[cg * TW_ROWS + row] = sq;
        }
      __syncthreads();
      if (tid < TW_ROWS)
        rstat[tid] = rsqrtf(((psum[tid] + psum[TW_ROWS + tid]) + (psum[2 * TW_ROWS + tid] + psum[3 * TW_ROWS + tid])) *
                            (1.0f / BB_C) + 1e-5f);
      __syncthreads();
      const float gm0 = vl[2 * BB_C + c0] * sa, gm1 = vl[2 * BB_C + c0 + 1] * sa;     // sa is a power of two: exact
      const float bt0 = vl[3 * BB_C + c0] * sa, bt1 = vl[3 * BB_C + c0 + 1] * sa;
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          if (row < TW_ROWS) {
            const float rs = rstat[row];
            const float v0 = row < tile_rows ? acc[r][0][e] * rs * gm0 + bt0 : 0.0f;
            const float v1 = row < tile_rows ? acc[r][1][e] * rs * gm1 + bt1 : 0.0f;
            V2 hi, lo;
            split2<T>(v0, v1, hi, lo);
            *reinterpret_cast<V2*>(plane + row * LPSB + 2 * c0) = hi;
            if constexpr (NP == 3) *reinterpret_cast<V2*>(plane + PLANE_B + row * LPSB + 2 * c0) = lo;
          }
        }
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          if (row < TW_ROWS) {
            V2 hi, lo;
            split2<T>(f[r][0][e] * sa, f[r][1][e] * sa, hi, lo);
            *reinterpret_cast<V2*>(plane + row * LPSB + 2 * c0) = hi;
            if constexpr (NP == 3) *reinterpret_cast<V2*>(plane + PLANE_B + row * LPSB + 2 * c0) = lo;
          }
        }
    }
    // ---- implicit GEMM over (chunk, live tap)
#pragma unroll
    for (int r = 0; r < NR; ++r) { acc[r][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[r][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    const int dil = __builtin_amdgcn_readfirstlane(sdil[layer < nl ? layer : BB_MAXL]);
    const int layer_end = (layer + 1) * 36;
    __syncthreads();                                      // the image is complete
    while (it < layer_end) {
      const int nxt = en >> 19;
      const int en_next_v = sched[nxt < it_end ? nxt : it];
      V8 bc[2 * NPARTS];
#pragma unroll
      for (int q = 0; q < 2 * NPARTS; ++q) bc[q] = bn[q];
      if (nxt < it_end) {
        const V8* src = wsrc + (size_t)tile_of(nxt) * TILE_V8;
#pragma unroll
        for (int q = 0; q < 2 * NPARTS; ++q) bn[q] = src[q];
      }
      const int delta = (((en >> 15) & 15) - 4) * dil;
      const int coff = ((en >> 13) & 3) * 64;
      const int dbytes = delta * LPSB + coff;
      const int live = en >> rh;                          // bit 2 r = owned tile r
#define LP_ALOAD(R, V)                                                                                       \
      { int o_;                                                                                              \
        if (SPT1) o_ = min(max(arow0 + dbytes + (R) * (32 * LPSB), a_lo + coff), a_hi + coff);               \
        else o_ = ((unsigned)(apos[SPT1 ? 0 : (R)] + delta) < (unsigned)L ? arow0 + dbytes + (R) * (32 * LPSB) \
                                                                            : a_hi + coff);                  \
        V[0] = *reinterpret_cast<const V8*>(plane + o_);                                                     \
        if constexpr (NP == 3) V[1] = *reinterpret_cast<const V8*>(plane + PLANE_B + o_); }
#define LP_WAIT(NOUT) __builtin_amdgcn_s_waitcnt(0xC07F | ((NOUT) << 8));
      // bc: [ct][part] -> bc[ct * NPARTS + part]
#define LP_MM(R, U, NOUT)                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      LP_WAIT(NOUT)                                                                                          \
      if (live & (1 << (2 * (R)))) {                                                                         \
        acc[R][0] = Lp<T>::mfma(U[0], bc[0], acc[R][0]);                                                     \
        acc[R][1] = Lp<T>::mfma(U[0], bc[NPARTS], acc[R][1]);                                                \
        if constexpr (NP == 3) {                                                                             \
          acc[R][0] = Lp<T>::mfma(U[0], bc[1], acc[R][0]);                                                   \
          acc[R][1] = Lp<T>::mfma(U[0], bc[NPARTS + 1], acc[R][1]);                                          \
          acc[R][0] = Lp<T>::mfma(U[1], bc[0], acc[R][0]);                                                   \
          acc[R][1] = Lp<T>::mfma(U[1], bc[NPARTS], acc[R][1]);                                              \
        }                                                                                                    \
      }                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);
      V8 ua[2], ub[2];
      LP_ALOAD(0, ua) LP_ALOAD(1, ub)
      LP_MM(0, ua, NPARTS)
      LP_ALOAD(2, ua)
      LP_MM(1, ub, NPARTS)
      LP_ALOAD(3, ub)
      LP_MM(2, ua, NPARTS)
      LP_ALOAD(4, ua)
      LP_MM(3, ub, NPARTS)
      LP_ALOAD(5, ub)
      LP_MM(4, ua, NPARTS)
      if (rh == 0) {
        LP_ALOAD(6, ua)
        LP_MM(5, ub, NPARTS)
        LP_MM(6, ua, 0)
      } else {
        LP_MM(5, ub, 0)
      }
#undef LP_MM
#undef LP_WAIT
#undef LP_ALOAD
      it = nxt;
      en = __builtin_amdgcn_readfirstlane(en_next_v);
    }
    __syncthreads();                                      // every wave is done reading the image
    const float bl0 = vl[c0], bl1 = vl[c0 + 1];
    if (layer < nl) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {                     // relu(conv + b) + f
          f[r][0][e] = fmaxf(acc[r][0][e] * inv + bl0, 0.0f) + f[r][0][e];
          f[r][1][e] = fmaxf(acc[r][1][e] * inv + bl1, 0.0f) + f[r][1][e];
        }
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          if (row < TW_ROWS) {                            // relu(W1 f + b1), fp32 image over the (dead) planes
            img32[row * BB_AP + c0] = fmaxf(acc[r][0][e] * inv + bl0, 0.0f);
            img32[row * BB_AP + c0 + 1] = fmaxf(acc[r][1][e] * inv + bl1, 0.0f);
          }
        }
    }
  }
  __syncthreads();
  // ---- last 1x1 conv 128 -> 5 in fp32
  for (int e = tid; e < tile_rows * 5; e += 512) {
    const int row = e / 5, v = e - 5 * row;
    if (row0 + row >= total_rows) continue;
    const float* hr = img32 + row * BB_AP;
    const float* wv = a.w2 + v * BB_C;
    float sm = a.w2[5 * BB_C + v];
#pragma unroll 8
    for (int k = 0; k < BB_C; ++k) sm += hr[k] * wv[k];
    a.out[(row0 + row) * 5 + v] = sm;
  }
}


// ------------------------------------------------------------------------ conv tower, split precision ----
// conv_tower_kernel / conv_tower_win_kernel of svdd_nets.hip on the 16-bit matrix cores. Same decomposition: one
// workgroup (8 waves) per tile of whole sequences (WIN: per candidate window), wave w owns the 16 channels of column
// tile w & 3 and the row tiles of parity w >> 2; the activation image lives in LDS as two 16-bit planes (hi, lo) and is
// the A operand directly; the wave keeps its accumulators AND its residual in fp32 registers. Differences:
//   * the input is the TOKEN row (u8), not the fp32 one-hot: the one-hot is built in LDS (exact in 16 bits, so the stem
//     needs only A * Bhi + A * Blo);
//   * one MFMA covers a whole (tap, 32-channel chunk);
//   * WIN: candidates can be addressed through a compacted index list whose length lives on the device
//     (live_idx / count: exact work-skipping without a host round trip); `count` == NULL means all n.
constexpr int TW_C = 64;
constexpr int TLSB = 160;                       // bytes per row of a 16-bit plane (64 channels + 32 B pad: conflict-free b128)
constexpr int TPLANE_B = (TW_ROWS + 2) * TLSB;  // rows -1 .. TW_ROWS
constexpr int TW_AP = TW_C + 4;                 // fp32 staging row stride (final store)
constexpr int TW_MAXL = 8;

struct TowerLpArgs {
  const uint8_t* tok;      // [n, L] tokens (0..3, 4 = MASK -> zero row)
  const void* tiles;       // [2 + 10*nlayers] tiles of [4 cs][64 lanes][P][8] 16-bit
  const float* bias;       // [1 + nlayers][64]
  const float* inv;        // [1 + nlayers] 1 / weight scale of the stage
  float* out;              // [n, L, 64]   (WIN with live_idx: [count, L, 64] compact)
  int n, L, spt, nlayers, residual_mask;
  // WIN only
  const int* win;          // [n][2]
  const float* parent_out; // [n / M, L, 64]
  int M;
  const int* live_idx;     // [count] candidate ids, or NULL (identity)
  const int* count;        // device scalar, or NULL (= n)
};

template <typename T, int NP, bool SPT1, bool WIN>
__global__ __launch_bounds__(512, 2) void tower_lp_kernel(TowerLpArgs a) {
  typedef typename Lp<T>::V8 V8;
  typedef typename Lp<T>::V2 V2;
  constexpr int NPARTS = NP == 3 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  char* plane = smem_b + TLSB;                                   // (row 0, channel 0) of the hi plane
  float* stage = reinterpret_cast<float*>(smem_b);               // final fp32 staging [TW_ROWS][TW_AP] over the planes
  T* xs = reinterpret_cast<T*>(smem_b + 2 * TPLANE_B) + 8 * 4;   // one-hot rows -8 .. TW_ROWS + 8, [row][4]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cs = w & 3, rh = w >> 2;
  const int j = lane & 15, g = lane >> 4;
  const int L = a.L;

  int cand = blockIdx.x, slot = blockIdx.x;
  int w0 = 0, w1 = 0;
  if (WIN) {
    if (a.count && (int)blockIdx.x >= __builtin_amdgcn_readfirstlane(*a.count)) return;
    if (a.live_idx) cand = __builtin_amdgcn_readfirstlane(a.live_idx[blockIdx.x]);
    w0 = __builtin_amdgcn_readfirstlane(a.win[2 * cand]);
    w1 = __builtin_amdgcn_readfirstlane(a.win[2 * cand + 1]);
  }
  const int nt = (w1 - w0) >> 4;
  const int tile_rows = WIN ? min(L, w1) - w0 : a.spt * L;       // valid local rows
  const int64_t row0 = WIN ? 0 : (int64_t)blockIdx.x * tile_rows;
  const int64_t total_rows = (int64_t)a.n * L;
  const int keep_lo = !WIN ? 0 : (nt == 0 ? 0 : (w0 == 0 ? 0 : w0 + 10));
  const int keep_hi = !WIN ? 0 : (nt == 0 ? 0 : (w1 >= L ? L : w1 - 10));
  float* outc = WIN ? a.out + (size_t)slot * L * TW_C : a.out;

  if (WIN) {
    const float* par = a.parent_out + (size_t)(cand / a.M) * L * TW_C;
    for (int e = tid; e < L * 16; e += 512) {                    // rows that are the parent's
      const int row = e >> 4;
      if (row < keep_lo || row >= keep_hi)
        *reinterpret_cast<float4*>(outc + (size_t)row * TW_C + 4 * (e & 15)) = *reinterpret_cast<const float4*>(par + (size_t)row * TW_C + 4 * (e & 15));
    }
    if (nt == 0) return;
  }

  // one-hot of the tokens, rows -8 .. TW_ROWS + 8 (zero outside the sequence / tile; MASK = zero row)
  {
    const uint8_t* tk = a.tok + (WIN ? (size_t)cand * L : 0);
    for (int e = tid - 8; e < TW_ROWS + 8; e += 512) {
      int t = 4;
      if (WIN) { const int gl = w0 + e; if (gl >= 0 && gl < L) t = tk[gl]; }
      else if (e >= 0 && e < tile_rows && row0 + e < total_rows) t = tk[row0 + e];
      typedef T T4 __attribute__((ext_vector_type(4)));
      T4 v; v[0] = (T)(t == 0 ? 1.0f : 0.0f); v[1] = (T)(t == 1 ? 1.0f : 0.0f); v[2] = (T)(t == 2 ? 1.0f : 0.0f); v[3] = (T)(t == 3 ? 1.0f : 0.0f);
      *reinterpret_cast<T4*>(xs + 4 * e) = v;
    }
  }
  for (int e = tid; e < 2 * TPLANE_B / 4; e += 512) reinterpret_cast<int*>(smem_b)[e] = 0;   // planes incl. the zero rows

  constexpr int TILE_V8 = 4 * 64 * NPARTS;
  const V8* wsrc = reinterpret_cast<const V8*>(a.tiles) + (cs * 64 + lane) * NPARTS;
  V8 bn[NPARTS];
#pragma unroll
  for (int q = 0; q < NPARTS; ++q) bn[q] = wsrc[q];

  const int arow0 = 16 * rh + j;                                 // this lane feeds rows arow0 + 32 r as the A operand
  int apos[(SPT1 || WIN) ? 1 : 7];
  if (!SPT1 && !WIN) {
#pragma unroll
    for (int r = 0; r < 7; ++r)
      apos[r] = (rh + 2 * r < TW_RT && arow0 + 32 * r < tile_rows) ? (arow0 + 32 * r) % L : -(1 << 20);
  }
  const int abase = (16 * rh + j) * TLSB + 16 * g;               // byte offset of (row 16 rh + j, channel 8 g)
  const int a_lo = abase - (16 * rh + j + 1) * TLSB;             // row -1
  const int a_hi = abase + (TW_ROWS - 16 * rh - j) * TLSB;       // row TW_ROWS
  const int nown = rh == 0 ? 7 : 6;
  const int nlive = WIN ? (nt - rh + 1) >> 1 : nown;             // owned live tiles rh + 2 r, r < nlive
  const int nit = 2 + 10 * a.nlayers;
  int it = 0;
  f32x4 acc[7], res[7];
#pragma unroll
  for (int r = 0; r < 7; ++r) res[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  __syncthreads();

#define TL_WAIT(NOUT) __builtin_amdgcn_s_waitcnt(0xC07F | ((NOUT) << 8));
  for (int layer = -1; layer < a.nlayers; ++layer) {
#pragma unroll
    for (int r = 0; r < 7; ++r) acc[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int niter = layer < 0 ? 2 : 10;
    for (int ci = 0; ci < niter; ++ci, ++it) {
      V8 bc[NPARTS];
#pragma unroll
      for (int q = 0; q < NPARTS; ++q) bc[q] = bn[q];
      if (it + 1 < nit) {
        const V8* src = wsrc + (size_t)(it + 1) * TILE_V8;
#pragma unroll
        for (int q = 0; q < NPARTS; ++q) bn[q] = src[q];
      }
      if (layer < 0) {
        // k = 32 ci + 8 g + e: taps t0 = 8 ci + 2 g and t0 + 1, 4 channels each = two adjacent one-hot rows
        const int t0 = 8 * ci + 2 * g;
        typedef T T4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int r = 0; r < 7; ++r) {
          if (r >= nlive) continue;
          T4 x0, x1;
          const int rr = arow0 + 32 * r + t0 - 7;
          if (WIN) {
            x0 = *reinterpret_cast<const T4*>(xs + 4 * rr); x1 = *reinterpret_cast<const T4*>(xs + 4 * (rr + 1));
          } else if (SPT1) {
            x0 = *reinterpret_cast<const T4*>(xs + 4 * min(max(rr, -1), TW_ROWS));
            x1 = *reinterpret_cast<const T4*>(xs + 4 * min(max(rr + 1, -1), TW_ROWS));
          } else {
            const int p = apos[(SPT1 || WIN) ? 0 : r] + t0 - 7;
            x0 = *reinterpret_cast<const T4*>(xs + 4 * ((unsigned)p < (unsigned)L ? rr : TW_ROWS));
            x1 = *reinterpret_cast<const T4*>(xs + 4 * ((unsigned)(p + 1) < (unsigned)L ? rr + 1 : TW_ROWS));
          }
          V8 af;
          af[0] = x0[0]; af[1] = x0[1]; af[2] = x0[2]; af[3] = x0[3]; af[4] = x1[0]; af[5] = x1[1]; af[6] = x1[2]; af[7] = x1[3];
          acc[r] = Lp<T>::mfma(af, bc[0], acc[r]);
          if constexpr (NP == 3) acc[r] = Lp<T>::mfma(af, bc[1], acc[r]);    // the one-hot has no lo part
        }
      } else {
        const int c = ci / 5, delta = ci - 5 * c - 2;
        const int dbytes = delta * TLSB + c * 64;
#define TL_ALOAD(R, V)                                                                                        \
        if ((R) < nlive) {                                                                                    \
          int o_;                                                                                             \
          if (SPT1 || WIN) o_ = min(max(abase + dbytes + (R) * (32 * TLSB), a_lo + c * 64), a_hi + c * 64);  \
          else o_ = (unsigned)(apos[(SPT1 || WIN) ? 0 : (R)] + delta) < (unsigned)L ? abase + dbytes + (R) * (32 * TLSB) \
                                                                                    : a_hi + c * 64;          \
          V[0] = *reinterpret_cast<const V8*>(plane + o_);                                                    \
          if constexpr (NP == 3) V[1] = *reinterpret_cast<const V8*>(plane + TPLANE_B + o_);                  \
        }
#define TL_MM(R, U)                                                                                           \
        if ((R) < nlive) {                                                                                    \
          acc[R] = Lp<T>::mfma(U[0], bc[0], acc[R]);                                                          \
          if constexpr (NP == 3) { acc[R] = Lp<T>::mfma(U[0], bc[1], acc[R]); acc[R] = Lp<T>::mfma(U[1], bc[0], acc[R]); } \
        }
        // tile groups (0,1,2)(3,4)(5,6), each group's reads one group ahead of its MFMAs
        V8 fa[3][2], fb[2][2];
        TL_ALOAD(0, fa[0]) TL_ALOAD(1, fa[1]) TL_ALOAD(2, fa[2])
        TL_ALOAD(3, fb[0]) TL_ALOAD(4, fb[1])
        __builtin_amdgcn_sched_barrier(0);
        if (nlive >= 5) { TL_WAIT(2 * NPARTS) } else { TL_WAIT(0) }
        TL_MM(0, fa[0]) TL_MM(1, fa[1]) TL_MM(2, fa[2])
        __builtin_amdgcn_sched_barrier(0);
        TL_ALOAD(5, fa[0]) TL_ALOAD(6, fa[1])
        __builtin_amdgcn_sched_barrier(0);
        if (nlive >= 7) { TL_WAIT(2 * NPARTS) } else if (nlive == 6) { TL_WAIT(NPARTS) } else { TL_WAIT(0) }
        TL_MM(3, fb[0]) TL_MM(4, fb[1])
        __builtin_amdgcn_sched_barrier(0);
        TL_WAIT(0)
        TL_MM(5, fa[0]) TL_MM(6, fa[1])
        __builtin_amdgcn_sched_barrier(0);
#undef TL_MM
#undef TL_ALOAD
      }
    }
    // every wave must be done reading the image before its owners overwrite it (the stem reads xs, not the image)
    if (layer >= 0) __syncthreads();
    const bool rs = layer >= 0 && ((a.residual_mask >> layer) & 1);
    const float bl = a.bias[(layer + 1) * TW_C + 16 * cs + j];
    const float inv = a.inv[layer + 1];
    const bool last = layer + 1 == a.nlayers;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      if (r >= nlive) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {                              // C/D layout: reg e -> row 4 g + e, column j
        const int row = 16 * (rh + 2 * r) + 4 * g + e;
        const float t = acc[r][e] * inv + bl + (rs ? res[r][e] : 0.0f);
        v[e] = row < tile_rows ? fmaxf(t, 0.0f) : 0.0f;
        res[r][e] = v[e];
      }
      if (last) {
#pragma unroll
        for (int e = 0; e < 4; ++e) stage[(16 * (rh + 2 * r) + 4 * g + e) * TW_AP + 16 * cs + j] = v[e];
      } else {
        // 16-bit image: adjacent channels are in adjacent lanes -> exchange so that every lane stores packed pairs:
        // even lanes store rows e = 0, 2 of channels (j, j + 1), odd lanes rows e = 1, 3 of channels (j - 1, j)
#pragma unroll
        for (int e2 = 0; e2 < 4; e2 += 2) {
          const float send = (j & 1) ? v[e2] : v[e2 + 1];
          const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
          const float p0 = (j & 1) ? recv : v[e2], p1 = (j & 1) ? v[e2 + 1] : recv;
          const int row = 16 * (rh + 2 * r) + 4 * g + e2 + (j & 1);
          V2 hi, lo;
          split2<T>(p0, p1, hi, lo);
          char* dst = plane + row * TLSB + 2 * (16 * cs + (j & ~1));
          *reinterpret_cast<V2*>(dst) = hi;
          if constexpr (NP == 3) *reinterpret_cast<V2*>(dst + TPLANE_B) = lo;
        }
      }
    }
    __syncthreads();                                             // the image (or the staging tile) is complete
  }
#undef TL_WAIT
  if (WIN) {
    for (int e = tid; e < L * 16; e += 512) {
      const int row = e >> 4, q = e & 15;
      if (row >= keep_lo && row < keep_hi)
        *reinterpret_cast<float4*>(outc + (size_t)row * TW_C + 4 * q) = *reinterpret_cast<const float4*>(stage + (row - w0) * TW_AP + 4 * q);
    }
  } else {
    for (int e = tid; e < tile_rows * 16; e += 512) {
      const int row = e >> 4, q = e & 15;
      if (row0 + row < total_rows)
        *reinterpret_cast<float4*>(a.out + (row0 + row) * TW_C + 4 * q) = *reinterpret_cast<const float4*>(stage + row * TW_AP + 4 * q);
    }
  }
}


// ------------------------------------------------------------------------------ bidirectional GRU, split precision ----
// gru_bidir_kernel of svdd_nets.hip on the 16-bit matrix cores: one workgroup (4 waves) per (tile of 16 sequences,
// direction); wave w owns hidden units 16 w .. 16 w + 15 of all three gates, its gate weights stay in registers as MFMA
// B operands (hi and lo: 96 VGPRs), the hidden state lives in LDS as two double-buffered 16-bit planes. One
// v_mfma_f32_16x16x32 covers half of K = 64, so a step costs 36 MFMAs per wave (x3 modes) instead of 96 fp32 ones; the
// kernel is bound by the serial chain barrier -> read h -> MFMAs -> gates -> write h, so several workgroups share a CU.
// Gates (sigmoid / tanh via v_exp_f32 / v_rcp_f32), the state update and the outputs are fp32 as in the fp32 kernel.
// `count` (device scalar, may be NULL = n): number of valid sequences — exact work-skipping without a host round trip.
constexpr int GLSB = 160;                  // bytes per hidden-state row of a 16-bit plane (64 x 2 B + 32 B pad)

__device__ __forceinline__ float sigmoid_fast(float a) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * a));
}
__device__ __forceinline__ float tanh_fast(float a) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177793f * a));
}

struct GruLpArgs {
  const float* x;          // [n, L, 64]
  const void* wpack;       // [2 dirs][4 waves][64 lanes][6 mtx][2 chunks][P][8] 16-bit ; mtx order ir, hr, iz, hz, in, hn
  const float* bpack;      // [2][4][64]: b_ir + b_hr, b_iz + b_hz, b_in, b_hn
  const float* inv;        // [2] 1 / weight scale per direction
  float* out;              // [2][n, L, 64]
  int n, L;
  const int* count;
};

template <typename T, int NP>
__global__ __launch_bounds__(256) void gru_lp_kernel(GruLpArgs a) {
  typedef typename Lp<T>::V8 V8;
  typedef typename Lp<T>::V2 V2;
  constexpr int NPARTS = NP == 3 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char hbuf[2][2][16 * GLSB];          // [buffer][hi | lo][row][160 B]
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int dir = blockIdx.y;
  const int j = lane & 15, g = lane >> 4;
  const int n = a.count ? __builtin_amdgcn_readfirstlane(*a.count) : a.n;
  const int seq0 = blockIdx.x * 16;
  if (seq0 >= n) return;
  const int L = a.L;

  V8 wb[6][2][NPARTS];
  {
    const V8* wp = reinterpret_cast<const V8*>(a.wpack) + (((size_t)dir * 4 + w) * 64 + lane) * (12 * NPARTS);
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < NPARTS; ++q) wb[m][c][q] = wp[(m * 2 + c) * NPARTS + q];
  }
  const int u = 16 * w + j;
  const float b_r = a.bpack[(dir * 4 + 0) * 64 + u], b_z = a.bpack[(dir * 4 + 1) * 64 + u];
  const float b_nx = a.bpack[(dir * 4 + 2) * 64 + u], b_nh = a.bpack[(dir * 4 + 3) * 64 + u];
  const float inv = a.inv[dir];

  const int arow = min(seq0 + j, n - 1);                          // clamped for the ragged last tile
  const float* xrow = a.x + (size_t)arow * L * 64 + 8 * g;
  float hprev[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = threadIdx.x; i < 2 * 16 * GLSB / 4; i += 256) reinterpret_cast<int*>(&hbuf[0][0][0])[i] = 0;   // h_0 = 0

  const int t0 = dir == 0 ? 0 : L - 1;
  const int dt = dir == 0 ? 1 : -1;
  float4 xn[4];                                                   // x_{t+1}: channels 8 g .. + 8 and 32 + 8 g .. + 8
  auto load_x = [&](int t) {
    const float4* xp = reinterpret_cast<const float4*>(xrow + (size_t)t * 64);
    xn[0] = xp[0]; xn[1] = xp[1]; xn[2] = xp[8]; xn[3] = xp[9];
  };
  auto split8 = [&](const float4& p, const float4& q, V8& hi, V8& lo) {
    const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) { const T h = (T)v[e]; hi[e] = h; lo[e] = (T)(v[e] - (float)h); }
  };
  f32x4 acc_r, acc_z, acc_nx;
  auto input_proj = [&]() {                                       // acc = W_i* x (scaled), from xn
    V8 xh[2], xl[2];
    split8(xn[0], xn[1], xh[0], xl[0]);
    split8(xn[2], xn[3], xh[1], xl[1]);
    acc_r = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc_z = acc_r; acc_nx = acc_r;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      acc_r = Lp<T>::mfma(xh[c], wb[0][c][0], acc_r);
      acc_z = Lp<T>::mfma(xh[c], wb[2][c][0], acc_z);
      acc_nx = Lp<T>::mfma(xh[c], wb[4][c][0], acc_nx);
      if constexpr (NP == 3) {
        acc_r = Lp<T>::mfma(xh[c], wb[0][c][1], acc_r);
        acc_z = Lp<T>::mfma(xh[c], wb[2][c][1], acc_z);
        acc_nx = Lp<T>::mfma(xh[c], wb[4][c][1], acc_nx);
        acc_r = Lp<T>::mfma(xl[c], wb[0][c][0], acc_r);
        acc_z = Lp<T>::mfma(xl[c], wb[2][c][0], acc_z);
        acc_nx = Lp<T>::mfma(xl[c], wb[4][c][0], acc_nx);
      }
    }
  };
  load_x(t0);
  input_proj();
  __syncthreads();

  for (int step = 0; step < L; ++step) {
    const int t = t0 + dt * step;
    const int cur = step & 1;
    if (step + 1 < L) load_x(t + dt);
    V8 hh[2], hl[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      hh[c] = *reinterpret_cast<const V8*>(&hbuf[cur][0][j * GLSB + 64 * c + 16 * g]);
      if constexpr (NP == 3) hl[c] = *reinterpret_cast<const V8*>(&hbuf[cur][1][j * GLSB + 64 * c + 16 * g]);
    }
    f32x4 acc_nh = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      acc_nh = Lp<T>::mfma(hh[c], wb[5][c][0], acc_nh);
      acc_r = Lp<T>::mfma(hh[c], wb[1][c][0], acc_r);
      acc_z = Lp<T>::mfma(hh[c], wb[3][c][0], acc_z);
      if constexpr (NP == 3) {
        acc_nh = Lp<T>::mfma(hh[c], wb[5][c][1], acc_nh);
        acc_r = Lp<T>::mfma(hh[c], wb[1][c][1], acc_r);
        acc_z = Lp<T>::mfma(hh[c], wb[3][c][1], acc_z);
        acc_nh = Lp<T>::mfma(hl[c], wb[5][c][0], acc_nh);
        acc_r = Lp<T>::mfma(hl[c], wb[1][c][0], acc_r);
        acc_z = Lp<T>::mfma(hl[c], wb[3][c][0], acc_z);
      }
    }
    float hn[4];
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {                           // C/D layout: reg rho -> sequence 4 g + rho, unit u
      const float r = sigmoid_fast(acc_r[rho] * inv + b_r);
      const float z = sigmoid_fast(acc_z[rho] * inv + b_z);
      const float nn = tanh_fast(acc_nx[rho] * inv + b_nx + r * (acc_nh[rho] * inv + b_nh));
      hn[rho] = (1.0f - z) * nn + z * hprev[rho];
      hprev[rho] = hn[rho];
      const int srow = 4 * g + rho;
      if (seq0 + srow < n) a.out[(((size_t)dir * a.n + seq0 + srow) * L + t) * 64 + u] = hn[rho];
    }
    // 16-bit state for the next step: adjacent units sit in adjacent lanes -> exchange, store packed pairs
#pragma unroll
    for (int e2 = 0; e2 < 4; e2 += 2) {
      const float send = (j & 1) ? hn[e2] : hn[e2 + 1];
      const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, false));
      const float p0 = (j & 1) ? recv : hn[e2], p1 = (j & 1) ? hn[e2 + 1] : recv;
      const int srow = 4 * g + e2 + (j & 1);
      V2 hi, lo;
      split2<T>(p0, p1, hi, lo);
      const int o = srow * GLSB + 2 * (16 * w + (j & ~1));
      *reinterpret_cast<V2*>(&hbuf[cur ^ 1][0][o]) = hi;
      if constexpr (NP == 3) *reinterpret_cast<V2*>(&hbuf[cur ^ 1][1][o]) = lo;
    }
    if (step + 1 < L) input_proj();                               // next step's input projections: off the serial chain
    __syncthreads();
  }
}

// ------------------------------------------------------------------ value-net tail, split precision ----
// value_tail_kernel of svdd_nets.hip with the 64 -> 128 map on the 16-bit matrix cores (W1' in registers as hi / lo B
// operands); direction sum, LayerNorm statistics, ReLU, the collapsed 128 -> n_tasks map and the mean stay fp32.
template <typename T, int NP, int TT>
__global__ __launch_bounds__(256, 2) void tail_lp_kernel(const float* __restrict__ hf, const float* __restrict__ hb,
                                                         const void* __restrict__ w1pack, const float* __restrict__ b1,
                                                         const float* __restrict__ weff, const float* __restrict__ beff,
                                                         float inv, float* __restrict__ out, int n_alloc, int L,
                                                         const int* __restrict__ count) {
  typedef typename Lp<T>::V8 V8;
  constexpr int NPARTS = NP == 3 ? 2 : 1;
  const int lane = threadIdx.x & 63;
  const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n = count ? *count : n_alloc;
  if (seq >= n) return;
  const int j = lane & 15, g = lane >> 4;
  V8 wb[8][2][NPARTS];                                           // W1'[16 ct + j][32 c + 8 g + e] (scaled)
  {
    const V8* wp = reinterpret_cast<const V8*>(w1pack) + (size_t)lane * (16 * NPARTS);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < NPARTS; ++q) wb[ct][c][q] = wp[(ct * 2 + c) * NPARTS + q];
  }
  float bias1[8], we[8][TT];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) {
    bias1[ct] = b1[16 * ct + j];
#pragma unroll
    for (int t = 0; t < TT; ++t) we[ct][t] = weff[(16 * ct + j) * TT + t];
  }
  float part[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) part[t] = 0.0f;
  const float* pf = hf + (size_t)seq * L * 64 + 8 * g;
  const float* pb = hb + (size_t)seq * L * 64 + 8 * g;
  const int ntiles = (L + 15) / 16;
  float4 xa[4], xb[4];
  auto load_rows = [&](int tile) {
    const int row = min(16 * tile + j, L - 1);
    const float4* a4 = reinterpret_cast<const float4*>(pf + (size_t)row * 64);
    const float4* b4 = reinterpret_cast<const float4*>(pb + (size_t)row * 64);
    xa[0] = a4[0]; xa[1] = a4[1]; xa[2] = a4[8]; xa[3] = a4[9];
    xb[0] = b4[0]; xb[1] = b4[1]; xb[2] = b4[8]; xb[3] = b4[9];
  };
  load_rows(0);
  for (int tile = 0; tile < ntiles; ++tile) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[4 * i] = xa[i].x + xb[i].x; v[4 * i + 1] = xa[i].y + xb[i].y;
      v[4 * i + 2] = xa[i].z + xb[i].z; v[4 * i + 3] = xa[i].w + xb[i].w;
    }
    if (tile + 1 < ntiles) load_rows(tile + 1);
    float sm = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sm += v[i];
    sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);
    const float mean = sm * (1.0f / 64.0f);
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] -= mean; sq += v[i] * v[i]; }
    sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq * (1.0f / 64.0f) + 1e-5f);
    V8 ah[2], al[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = v[8 * c + e] * rstd;
        const T h = (T)x;
        ah[c][e] = h; al[c][e] = (T)(x - (float)h);
      }
    f32x4 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[ct] = Lp<T>::mfma(ah[c], wb[ct][c][0], acc[ct]);
      if constexpr (NP == 3) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[ct] = Lp<T>::mfma(ah[c], wb[ct][c][1], acc[ct]);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[ct] = Lp<T>::mfma(al[c], wb[ct][c][0], acc[ct]);
      }
    }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      if (16 * tile + 4 * g + rho < L) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
          const float z = fmaxf(acc[ct][rho] * inv + bias1[ct], 0.0f);
#pragma unroll
          for (int t = 0; t < TT; ++t) part[t] += z * we[ct][t];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    float tot = part[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
    if (lane == 0) out[(size_t)seq * TT + t] = tot / (float)L + beff[t];
  }
}

}  // namespace

extern "C" int svdd_backbone_cnn_lp(const uint8_t* x, const float* table0, const void* tiles, const float* vec,
                                    const float* lscale, const float* w2, float* out, int n, int L, int nlayers,
                                    const int* dilations, int prec, void* stream) {
  if (!x || !table0 || !tiles || !vec || !lscale || !w2 || !out || !dilations || n <= 0 || L <= 0 || L > TW_ROWS ||
      nlayers <= 0 || nlayers > BB_MAXL || prec < SVDD_PREC_F16X3 || prec > SVDD_PREC_BF16)
    return SVDD_E_ARG;
  BackboneLpArgs a;
  a.x = x; a.table0 = table0; a.tiles = tiles; a.vec = vec; a.lscale = lscale; a.w2 = w2; a.out = out;
  a.n = n; a.L = L; a.spt = TW_ROWS / L; a.nl = nlayers;
  for (int i = 0; i < nlayers; ++i) if (dilations[i] <= 0) return SVDD_E_ARG;
  for (int i = 0; i < BB_MAXL; ++i) a.dil[i] = i < nlayers ? dilations[i] : 1;
  const size_t lds = (size_t)IMG_REGION_B + sizeof(float) * (9 * 5 * (size_t)BB_C + 4 * (size_t)TW_ROWS + 3 * (size_t)TW_ROWS +
                                                              BB_MAXL + 1 + (size_t)(nlayers + 1) * 36);
  hipEvent_t e0, e1;
  svdd_internal_timed_events(6, &e0, &e1);
  const dim3 grid((unsigned)((n + a.spt - 1) / a.spt));
  const bool spt1 = a.spt == 1;
#define LP_LAUNCH(TT, NPP)                                                                                          \
  do {                                                                                                               \
    if (spt1) {                                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_lp_kernel<TT, NPP, true>),                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
      hipExtLaunchKernelGGL((backbone_lp_kernel<TT, NPP, true>), grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a);  \
    } else {                                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_lp_kernel<TT, NPP, false>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
      hipExtLaunchKernelGGL((backbone_lp_kernel<TT, NPP, false>), grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a); \
    }                                                                                                                \
  } while (0)
  switch (prec) {
    case SVDD_PREC_F16X3: LP_LAUNCH(_Float16, 3); break;
    case SVDD_PREC_BF16X3: LP_LAUNCH(__bf16, 3); break;
    case SVDD_PREC_F16: LP_LAUNCH(_Float16, 1); break;
    default: LP_LAUNCH(__bf16, 1); break;
  }
#undef LP_LAUNCH
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

static int launch_tower_lp(const TowerLpArgs& a, bool win, int prec, unsigned grid, void* stream) {
  const size_t lds = 2 * (size_t)TPLANE_B + (size_t)(TW_ROWS + 16) * 4 * 2;
  hipEvent_t e0, e1;
  svdd_internal_timed_events(5, &e0, &e1);
#define TL_LAUNCH(TT, NPP, S1, WW)                                                                                 \
  do {                                                                                                              \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tower_lp_kernel<TT, NPP, S1, WW>),                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
    hipExtLaunchKernelGGL((tower_lp_kernel<TT, NPP, S1, WW>), dim3(grid), dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a); \
  } while (0)
#define TL_MODE(TT, NPP)                                                                                            \
  do {                                                                                                              \
    if (win) TL_LAUNCH(TT, NPP, true, true);                                                                        \
    else if (a.spt == 1) TL_LAUNCH(TT, NPP, true, false);                                                           \
    else TL_LAUNCH(TT, NPP, false, false);                                                                          \
  } while (0)
  switch (prec) {
    case SVDD_PREC_F16X3: TL_MODE(_Float16, 3); break;
    case SVDD_PREC_BF16X3: TL_MODE(__bf16, 3); break;
    case SVDD_PREC_F16: TL_MODE(_Float16, 1); break;
    default: TL_MODE(__bf16, 1); break;
  }
#undef TL_MODE
#undef TL_LAUNCH
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

extern "C" int svdd_conv_tower_lp(const uint8_t* tok, const void* tiles, const float* bias, const float* inv, float* out,
                                  int n, int L, int nlayers, int residual_mask, int prec, void* stream) {
  if (!tok || !tiles || !bias || !inv || !out || n <= 0 || L <= 0 || L > TW_ROWS || nlayers <= 0 || nlayers > TW_MAXL ||
      prec < SVDD_PREC_F16X3 || prec > SVDD_PREC_BF16)
    return SVDD_E_ARG;
  const int spt = TW_ROWS / L;
  TowerLpArgs a{tok, tiles, bias, inv, out, n, L, spt, nlayers, residual_mask, nullptr, nullptr, 1, nullptr, nullptr};
  return launch_tower_lp(a, false, prec, (unsigned)((n + spt - 1) / spt), stream);
}

extern "C" int svdd_conv_tower_windows_lp(const uint8_t* cand, const void* tiles, const float* bias, const float* inv,
                                          const int32_t* win, const float* parent_out, float* out, int n, int L, int M,
                                          int nlayers, int residual_mask, const int32_t* live_idx, const int32_t* count,
                                          int prec, void* stream) {
  if (!cand || !tiles || !bias || !inv || !win || !parent_out || !out || n <= 0 || M <= 0 || n % M || L <= TW_ROWS / 2 ||
      L > TW_ROWS || nlayers != 5 || prec < SVDD_PREC_F16X3 || prec > SVDD_PREC_BF16)
    return SVDD_E_ARG;
  TowerLpArgs a{cand, tiles, bias, inv, out, n, L, 1, nlayers, residual_mask, win, parent_out, M, live_idx, count};
  return launch_tower_lp(a, true, prec, (unsigned)n, stream);
}

extern "C" int svdd_gru_bidir_lp(const float* x, const void* wpack, const float* bpack, const float* inv, float* out,
                                 int n, int L, const int32_t* count, int prec, void* stream) {
  if (!x || !wpack || !bpack || !inv || !out || n <= 0 || L <= 0 || prec < SVDD_PREC_F16X3 || prec > SVDD_PREC_BF16)
    return SVDD_E_ARG;
  GruLpArgs a{x, wpack, bpack, inv, out, n, L, count};
  hipEvent_t e0, e1;
  svdd_internal_timed_events(3, &e0, &e1);
  const dim3 grid((unsigned)((n + 15) / 16), 2);
  switch (prec) {
    case SVDD_PREC_F16X3: hipExtLaunchKernelGGL((gru_lp_kernel<_Float16, 3>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a); break;
    case SVDD_PREC_BF16X3: hipExtLaunchKernelGGL((gru_lp_kernel<__bf16, 3>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a); break;
    case SVDD_PREC_F16: hipExtLaunchKernelGGL((gru_lp_kernel<_Float16, 1>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a); break;
    default: hipExtLaunchKernelGGL((gru_lp_kernel<__bf16, 1>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a); break;
  }
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

extern "C" int svdd_value_tail_lp(const float* h_fwd, const float* h_bwd, const void* w1pack, const float* b1,
                                  const float* w_eff, const float* b_eff, float inv, float* out, int n, int L, int n_tasks,
                                  const int32_t* count, int prec, void* stream) {
  if (!h_fwd || !h_bwd || !w1pack || !b1 || !w_eff || !b_eff || !out || n <= 0 || L <= 0 || n_tasks < 1 || n_tasks > 4 ||
      prec < SVDD_PREC_F16X3 || prec > SVDD_PREC_BF16)
    return SVDD_E_ARG;
  hipEvent_t e0, e1;
  svdd_internal_timed_events(7, &e0, &e1);
  const dim3 grid((unsigned)((n + 3) / 4)), block(256);
#define TAIL_LP(TY, NPP, TT)                                                                                           \
  hipExtLaunchKernelGGL((tail_lp_kernel<TY, NPP, TT>), grid, block, 0, (hipStream_t)stream, e0, e1, 0, h_fwd, h_bwd, w1pack, \
                        b1, w_eff, b_eff, inv, out, n, L, count)
#define TAIL_LP_T(TY, NPP)                                                                                             \
  switch (n_tasks) { case 1: TAIL_LP(TY, NPP, 1); break; case 2: TAIL_LP(TY, NPP, 2); break;                           \
                     case 3: TAIL_LP(TY, NPP, 3); break; default: TAIL_LP(TY, NPP, 4); break; }
  switch (prec) {
    case SVDD_PREC_F16X3: TAIL_LP_T(_Float16, 3) break;
    case SVDD_PREC_BF16X3: TAIL_LP_T(__bf16, 3) break;
    case SVDD_PREC_F16: TAIL_LP_T(_Float16, 1) break;
    default: TAIL_LP_T(__bf16, 1) break;
  }
#undef TAIL_LP_T
#undef TAIL_LP
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}
