// svdd_lp_backbone.hip — the one-launch dilated-CNN backbone, split precision
// (split-precision net kernels on the 16-bit matrix cores: see svdd_lp_common.h for the arithmetic)
#include "svdd_lp_common.h"

extern "C" int svdd_internal_num_cus();      // svdd_nets.hip
extern "C" int svdd_internal_fixed_spt();    // svdd_nets.hip (svdd_set_backbone_packing)

static int g_bb_lp_version = 2;     // svdd_set_option(SVDD_OPT_BACKBONE_LP_VERSION, 1): the round-2 kernel for every shape (A/B)
static int g_bb_lp_rg = 2;          // ... 21 / 22 / 23: the transposed kernel with 1 / 2 / 3 row groups (waves per SIMD); measured
                                    // (f16x3 / bf16, B = 256, L = 200, round 4d): 2: 0.652 / 0.339 ms ; 1: 0.74 / 0.51 ms (512 registers, but
                                    // nothing hides its own stalls) ; 3: spills at 168 VGPRs. All three produce the same bits (tests/test_lp_gpu.py)
extern "C" void svdd_internal_set_bb_lp_version(int v) {
  if (v >= 21 && v <= 23) { g_bb_lp_version = 2; g_bb_lp_rg = v - 20; }
  else g_bb_lp_version = v == 1 ? 1 : 2;
}

namespace {
constexpr int WAVE_SZ = 64;

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
template <typename T, int NP, bool SPT1>
__global__ __launch_bounds__(512, 2) void backbone_lp_kernel(BackboneLpArgs a) {
  typedef typename Lp<T>::V8 V8;
  typedef typename Lp<T>::V2 V2;
  constexpr int NPARTS = NP == 3 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  char* plane = smem_b + LPSB;                               // byte address of (row 0, channel 0) of the hi plane
  float* img32 = reinterpret_cast<float*>(smem_b);           // final stage: fp32 image [TW_ROWS][BB_AP] over the planes
  float* Bs = reinterpret_cast<float*>(smem_b + IMG_REGION_B);   // [9][5][128] the first layer's lookup table
  float* psum = Bs + 9 * 5 * BB_C;                           // [8][TW_ROWS]: first / second moment partials of the 4 column groups
  float* rstat = psum + 8 * TW_ROWS;                         // [TW_ROWS]
  float* rmean = rstat + TW_ROWS;                            // [TW_ROWS] row means (this layer's shift = last layer's mean)
  int* toks = reinterpret_cast<int*>(rmean + TW_ROWS);       // [TW_ROWS]
  int* rpos = toks + TW_ROWS;                                // [TW_ROWS]
  int* sdil = rpos + TW_ROWS;                                // [BB_MAXL + 1]
  int* sched = sdil + BB_MAXL + 1;                           // [(nl + 1) * 36]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = w & 3, rh = w >> 2;
  const int j = lane & 15, g = lane >> 4;
  const int c0 = 32 * cg + 2 * j;                            // this lane's output channels: c0 and c0 + 1
  const int L = a.L;
  const int nvalid = a.count ? __builtin_amdgcn_readfirstlane(*a.count) : a.n;
  const int spt = (!SPT1 && a.auto_spt) ? __builtin_amdgcn_readfirstlane(svdd_choose_spt(nvalid, L, a.ncu, NP == 3 ? 26 : 40)) : a.spt;
  const int tile_rows = spt * L;
  const int64_t row0 = (int64_t)blockIdx.x * tile_rows;
  const int64_t total_rows = (int64_t)nvalid * L;
  if (row0 >= total_rows) return;
  const int nl = a.nl;
  const int it_end = (nl + 1) * 36;
  constexpr int NR = 7;

  for (int e = tid; e < TW_ROWS; e += 512) {
    int tk = -1;
    if (e < tile_rows && row0 + e < total_rows) {
      if (a.row_idx) { const int sq = e / L; tk = a.x[(int64_t)a.row_idx[blockIdx.x * spt + sq] * L + (e - sq * L)]; }
      else tk = a.x[row0 + e];
    }
    toks[e] = tk;
    rmean[e] = 0.0f;
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
      // ONE statistics pass: first and second moment about a per-row shift K = the row's mean one layer earlier (the
      // residual stream drifts slowly, so |mean - K| stays of the order of the row's sigma and var = S2/n - (S1/n)^2 loses
      // nothing; layer 0: K = 0). The two-pass form (kept in the exact-fp32 kernel, where the phases weigh 8 %) cost this
      // kernel two more barriers and an LDS round trip per layer.
      // The four rows of a tile that a lane owns are adjacent: one 16-byte read per tile, issued one tile ahead.
      float4 st4 = *reinterpret_cast<const float4*>(rmean + min(16 * rh + 4 * g, TW_ROWS - 4));
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float4 cur4 = st4;
        if (r + 1 < NR) st4 = *reinterpret_cast<const float4*>(rmean + min(16 * (rh + 2 * r + 2) + 4 * g, TW_ROWS - 4));
        const float k4[4] = {cur4.x, cur4.y, cur4.z, cur4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          const float d0 = f[r][0][e] + tb0 - k4[e], d1 = f[r][1][e] + tb1 - k4[e];
          const float s1 = group16_sum(d0 + d1);
          const float s2 = group16_sum(d0 * d0 + d1 * d1);
          if (j == 0 && row < TW_ROWS) { psum[cg * TW_ROWS + row] = s1; psum[(4 + cg) * TW_ROWS + row] = s2; }
        }
      }
      __syncthreads();
      if (tid < TW_ROWS) {
        const float m = ((psum[tid] + psum[TW_ROWS + tid]) + (psum[2 * TW_ROWS + tid] + psum[3 * TW_ROWS + tid])) * (1.0f / BB_C);
        const float q = ((psum[4 * TW_ROWS + tid] + psum[5 * TW_ROWS + tid]) + (psum[6 * TW_ROWS + tid] + psum[7 * TW_ROWS + tid])) * (1.0f / BB_C);
        rmean[tid] += m;
        rstat[tid] = rsqrtf(fmaxf(q - m * m, 0.0f) + 1e-5f);
      }
      __syncthreads();
      const float gm0 = vl[2 * BB_C + c0] * sa, gm1 = vl[2 * BB_C + c0 + 1] * sa;     // sa is a power of two: exact
      const float bt0 = vl[3 * BB_C + c0] * sa, bt1 = vl[3 * BB_C + c0 + 1] * sa;
      st4 = *reinterpret_cast<const float4*>(rstat + min(16 * rh + 4 * g, TW_ROWS - 4));
      float4 mn4 = *reinterpret_cast<const float4*>(rmean + min(16 * rh + 4 * g, TW_ROWS - 4));
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float4 cur4 = st4, curm = mn4;
        if (r + 1 < NR) {
          st4 = *reinterpret_cast<const float4*>(rstat + min(16 * (rh + 2 * r + 2) + 4 * g, TW_ROWS - 4));
          mn4 = *reinterpret_cast<const float4*>(rmean + min(16 * (rh + 2 * r + 2) + 4 * g, TW_ROWS - 4));
        }
        const float rs4[4] = {cur4.x, cur4.y, cur4.z, cur4.w};
        const float mean4[4] = {curm.x, curm.y, curm.z, curm.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          if (row < TW_ROWS) {
            const float rs = rs4[e], mean = mean4[e];
            const float v0 = row < tile_rows ? (f[r][0][e] + tb0 - mean) * rs * gm0 + bt0 : 0.0f;
            const float v1 = row < tile_rows ? (f[r][1][e] + tb1 - mean) * rs * gm1 + bt1 : 0.0f;
            V2 hi, lo;
            split2<T>(v0, v1, hi, lo);
            *reinterpret_cast<V2*>(plane + row * LPSB + 2 * c0) = hi;
            if constexpr (NP == 3) *reinterpret_cast<V2*>(plane + PLANE_B + row * LPSB + 2 * c0) = lo;
          }
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
    if (a.row_idx && a.out_scatter) { const int sq = row / L; a.out[((int64_t)a.row_idx[blockIdx.x * spt + sq] * L + (row - sq * L)) * 5 + v] = sm; }
    else a.out[(row0 + row) * 5 + v] = sm;
  }
}


// ------------------------------------------------------------------- backbone, split precision, TRANSPOSED tiles ----
// Round 3. Same tile decomposition, LDS image, weight stream and schedule as backbone_lp_kernel<.., SPT1 = true> above, but
// the two MFMA operands trade places: A = the weight fragment, B = the activation fragment (the very same registers:
// both operand layouts are "lane (j, g) holds index j of the tile and k = 8 g .. 8 g + 7"), so the accumulator tile comes
// out TRANSPOSED: lane (j, g) register e holds output channel 4 g + e of the column tile at POSITION 16 R + j.
// With the host's weight order (tile row j of column tile ct = channel 32 cg + 2 j + ct) a lane then owns EIGHT ADJACENT
// channels 32 cg + 8 g .. + 7 (value (ct, e) <-> channel offset 2 e + ct) of ONE position per row tile, instead of two
// channels of four positions. What that buys — the ablation of the first version put the LayerNorm phases at 0.19 ms, the
// first-layer lookup at 0.05 ms and the MFMAs at 0.39 ms of its 0.77 ms (profiles/r02_exp_ablations.txt):
//   * LayerNorm statistics: a position's 128 channels are 8 values in a lane (summed with packed fp32 adds), 4 lanes
//     (g) and 4 waves: two cross-lane steps per row tile instead of 4 DPP steps for each of 4 rows and 2 moments;
//   * the image write is one 16-byte LDS store per plane and row tile (8 halves) instead of four 4-byte stores, with
//     one mean / rstd read per tile instead of four;
//   * the first layer reads its 8 channels of a (tap, token) as two 16-byte LDS loads: 18 loads per row tile, not 72;
//   * per-channel parameters are two 16-byte loads per vector.
// One sequence per tile only (104 < L <= 208); shorter sequences keep the kernel above.
typedef float f32x8 __attribute__((ext_vector_type(8)));

// v + (v of lane ^ X), X = 16 or 32, on the VALU: gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd 16-lane rows
// (the upper 32 lanes) of one register with the even rows (the lower lanes) of another, so after swapping two copies of v each
// lane holds its own value in one and its partner's in the other. __shfl_xor compiles to ds_bpermute_b32 (an LDS round trip
// per exchange, four in a row per row tile of the LayerNorm statistics); the builtin of the swap returns its first result
// twice with this compiler (tools/ubench/permlane_swap_check.hip), hence the asm with the two wait states it needs.
template <int X>
__device__ __forceinline__ float xor_sum(float v) {
  static_assert(X == 16 || X == 32, "row or half-wave exchange");
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  if constexpr (X == 16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

// bits RG r, r < n: the row tiles a wave that owns n of them sees in the schedule's live mask (shifted by its row group)
constexpr int lpt_own_mask(int n, int rgs) { int m = 0; for (int r = 0; r < n; ++r) m |= 1 << (rgs * r); return m; }

// BEGIN generated by tools/gen_lpt_taps.py (do not edit by hand)
#define LPT_TAP13(S)                                                                                              \
      LPT_XLOAD(0, 0, ua)                                                                                         \
      S(0, ua, wA, LPT_XLOAD(0, 1, ub), LPT_WLD0(wB, t0 + 1 * tstep))                                             \
      S(1, ub, wA, LPT_XLOAD(0, 2, ua), LPT_WLD(wB, 1))                                                           \
      S(2, ua, wA, LPT_XLOAD(0, 3, ub), LPT_WLD(wB, 2))                                                           \
      S(3, ub, wA, LPT_XLOAD(0, 4, ua), LPT_WLD(wB, 3))                                                           \
      S(4, ua, wA, LPT_XLOAD(0, 5, ub), )                                                                         \
      S(5, ub, wA, LPT_XLOAD(0, 6, ua), )                                                                         \
      S(6, ua, wA, LPT_XLOAD(0, 7, ub), )                                                                         \
      S(7, ub, wA, LPT_XLOAD(0, 8, ua), )                                                                         \
      S(8, ua, wA, LPT_XLOAD(0, 9, ub), )                                                                         \
      S(9, ub, wA, LPT_XLOAD(0, 10, ua), )                                                                        \
      S(10, ua, wA, LPT_XLOAD(0, 11, ub), )                                                                       \
      S(11, ub, wA, LPT_XLOAD(0, 12, ua), )                                                                       \
      S(12, ua, wA, LPT_XLOAD(1, 0, ub), )                                                                        \
      S(0, ub, wB, LPT_XLOAD(1, 1, ua), LPT_WLD0(wA, t0 + 2 * tstep))                                             \
      S(1, ua, wB, LPT_XLOAD(1, 2, ub), LPT_WLD(wA, 1))                                                           \
      S(2, ub, wB, LPT_XLOAD(1, 3, ua), LPT_WLD(wA, 2))                                                           \
      S(3, ua, wB, LPT_XLOAD(1, 4, ub), LPT_WLD(wA, 3))                                                           \
      S(4, ub, wB, LPT_XLOAD(1, 5, ua), )                                                                         \
      S(5, ua, wB, LPT_XLOAD(1, 6, ub), )                                                                         \
      S(6, ub, wB, LPT_XLOAD(1, 7, ua), )                                                                         \
      S(7, ua, wB, LPT_XLOAD(1, 8, ub), )                                                                         \
      S(8, ub, wB, LPT_XLOAD(1, 9, ua), )                                                                         \
      S(9, ua, wB, LPT_XLOAD(1, 10, ub), )                                                                        \
      S(10, ub, wB, LPT_XLOAD(1, 11, ua), )                                                                       \
      S(11, ua, wB, LPT_XLOAD(1, 12, ub), )                                                                       \
      S(12, ub, wB, LPT_XLOAD(2, 0, ua), )                                                                        \
      S(0, ua, wA, LPT_XLOAD(2, 1, ub), LPT_WLD0(wB, t0 + 3 * tstep))                                             \
      S(1, ub, wA, LPT_XLOAD(2, 2, ua), LPT_WLD(wB, 1))                                                           \
      S(2, ua, wA, LPT_XLOAD(2, 3, ub), LPT_WLD(wB, 2))                                                           \
      S(3, ub, wA, LPT_XLOAD(2, 4, ua), LPT_WLD(wB, 3))                                                           \
      S(4, ua, wA, LPT_XLOAD(2, 5, ub), )                                                                         \
      S(5, ub, wA, LPT_XLOAD(2, 6, ua), )                                                                         \
      S(6, ua, wA, LPT_XLOAD(2, 7, ub), )                                                                         \
      S(7, ub, wA, LPT_XLOAD(2, 8, ua), )                                                                         \
      S(8, ua, wA, LPT_XLOAD(2, 9, ub), )                                                                         \
      S(9, ub, wA, LPT_XLOAD(2, 10, ua), )                                                                        \
      S(10, ua, wA, LPT_XLOAD(2, 11, ub), )                                                                       \
      S(11, ub, wA, LPT_XLOAD(2, 12, ua), )                                                                       \
      S(12, ua, wA, LPT_XLOAD(3, 0, ub), )                                                                        \
      S(0, ub, wB, LPT_XLOAD(3, 1, ua), LPT_WLD0(wA, t1))                                                         \
      S(1, ua, wB, LPT_XLOAD(3, 2, ub), LPT_WLD(wA, 1))                                                           \
      S(2, ub, wB, LPT_XLOAD(3, 3, ua), LPT_WLD(wA, 2))                                                           \
      S(3, ua, wB, LPT_XLOAD(3, 4, ub), LPT_WLD(wA, 3))                                                           \
      S(4, ub, wB, LPT_XLOAD(3, 5, ua), )                                                                         \
      S(5, ua, wB, LPT_XLOAD(3, 6, ub), )                                                                         \
      S(6, ub, wB, LPT_XLOAD(3, 7, ua), )                                                                         \
      S(7, ua, wB, LPT_XLOAD(3, 8, ub), )                                                                         \
      S(8, ub, wB, LPT_XLOAD(3, 9, ua), )                                                                         \
      S(9, ua, wB, LPT_XLOAD(3, 10, ub), )                                                                        \
      S(10, ub, wB, LPT_XLOAD(3, 11, ua), )                                                                       \
      S(11, ua, wB, LPT_XLOAD(3, 12, ub), )                                                                       \
      S(12, ub, wB, , )
#define LPT_TAP7(S)                                                                                               \
      LPT_XLOAD(0, 0, ua)                                                                                         \
      S(0, ua, wA, LPT_XLOAD(0, 1, ub), LPT_WLD0(wB, t0 + 1 * tstep))                                             \
      S(1, ub, wA, LPT_XLOAD(0, 2, ua), LPT_WLD(wB, 1))                                                           \
      S(2, ua, wA, LPT_XLOAD(0, 3, ub), LPT_WLD(wB, 2))                                                           \
      S(3, ub, wA, LPT_XLOAD(0, 4, ua), LPT_WLD(wB, 3))                                                           \
      S(4, ua, wA, LPT_XLOAD(0, 5, ub), )                                                                         \
      S(5, ub, wA, LPT_XLOAD(0, 6, ua), )                                                                         \
      S(6, ua, wA, LPT_XLOAD(1, 0, ub), )                                                                         \
      S(0, ub, wB, LPT_XLOAD(1, 1, ua), LPT_WLD0(wA, t0 + 2 * tstep))                                             \
      S(1, ua, wB, LPT_XLOAD(1, 2, ub), LPT_WLD(wA, 1))                                                           \
      S(2, ub, wB, LPT_XLOAD(1, 3, ua), LPT_WLD(wA, 2))                                                           \
      S(3, ua, wB, LPT_XLOAD(1, 4, ub), LPT_WLD(wA, 3))                                                           \
      S(4, ub, wB, LPT_XLOAD(1, 5, ua), )                                                                         \
      S(5, ua, wB, LPT_XLOAD(1, 6, ub), )                                                                         \
      S(6, ub, wB, LPT_XLOAD(2, 0, ua), )                                                                         \
      S(0, ua, wA, LPT_XLOAD(2, 1, ub), LPT_WLD0(wB, t0 + 3 * tstep))                                             \
      S(1, ub, wA, LPT_XLOAD(2, 2, ua), LPT_WLD(wB, 1))                                                           \
      S(2, ua, wA, LPT_XLOAD(2, 3, ub), LPT_WLD(wB, 2))                                                           \
      S(3, ub, wA, LPT_XLOAD(2, 4, ua), LPT_WLD(wB, 3))                                                           \
      S(4, ua, wA, LPT_XLOAD(2, 5, ub), )                                                                         \
      S(5, ub, wA, LPT_XLOAD(2, 6, ua), )                                                                         \
      S(6, ua, wA, LPT_XLOAD(3, 0, ub), )                                                                         \
      S(0, ub, wB, LPT_XLOAD(3, 1, ua), LPT_WLD0(wA, t1))                                                         \
      S(1, ua, wB, LPT_XLOAD(3, 2, ub), LPT_WLD(wA, 1))                                                           \
      S(2, ub, wB, LPT_XLOAD(3, 3, ua), LPT_WLD(wA, 2))                                                           \
      S(3, ua, wB, LPT_XLOAD(3, 4, ub), LPT_WLD(wA, 3))                                                           \
      S(4, ub, wB, LPT_XLOAD(3, 5, ua), )                                                                         \
      S(5, ua, wB, LPT_XLOAD(3, 6, ub), )                                                                         \
      S(6, ub, wB, , )
#define LPT_TAP6(S)                                                                                               \
      LPT_XLOAD(0, 0, ua)                                                                                         \
      S(0, ua, wA, LPT_XLOAD(0, 1, ub), LPT_WLD0(wB, t0 + 1 * tstep))                                             \
      S(1, ub, wA, LPT_XLOAD(0, 2, ua), LPT_WLD(wB, 1))                                                           \
      S(2, ua, wA, LPT_XLOAD(0, 3, ub), LPT_WLD(wB, 2))                                                           \
      S(3, ub, wA, LPT_XLOAD(0, 4, ua), LPT_WLD(wB, 3))                                                           \
      S(4, ua, wA, LPT_XLOAD(0, 5, ub), )                                                                         \
      S(5, ub, wA, LPT_XLOAD(1, 0, ua), )                                                                         \
      S(0, ua, wB, LPT_XLOAD(1, 1, ub), LPT_WLD0(wA, t0 + 2 * tstep))                                             \
      S(1, ub, wB, LPT_XLOAD(1, 2, ua), LPT_WLD(wA, 1))                                                           \
      S(2, ua, wB, LPT_XLOAD(1, 3, ub), LPT_WLD(wA, 2))                                                           \
      S(3, ub, wB, LPT_XLOAD(1, 4, ua), LPT_WLD(wA, 3))                                                           \
      S(4, ua, wB, LPT_XLOAD(1, 5, ub), )                                                                         \
      S(5, ub, wB, LPT_XLOAD(2, 0, ua), )                                                                         \
      S(0, ua, wA, LPT_XLOAD(2, 1, ub), LPT_WLD0(wB, t0 + 3 * tstep))                                             \
      S(1, ub, wA, LPT_XLOAD(2, 2, ua), LPT_WLD(wB, 1))                                                           \
      S(2, ua, wA, LPT_XLOAD(2, 3, ub), LPT_WLD(wB, 2))                                                           \
      S(3, ub, wA, LPT_XLOAD(2, 4, ua), LPT_WLD(wB, 3))                                                           \
      S(4, ua, wA, LPT_XLOAD(2, 5, ub), )                                                                         \
      S(5, ub, wA, LPT_XLOAD(3, 0, ua), )                                                                         \
      S(0, ua, wB, LPT_XLOAD(3, 1, ub), LPT_WLD0(wA, t1))                                                         \
      S(1, ub, wB, LPT_XLOAD(3, 2, ua), LPT_WLD(wA, 1))                                                           \
      S(2, ua, wB, LPT_XLOAD(3, 3, ub), LPT_WLD(wA, 2))                                                           \
      S(3, ub, wB, LPT_XLOAD(3, 4, ua), LPT_WLD(wA, 3))                                                           \
      S(4, ua, wB, LPT_XLOAD(3, 5, ub), )                                                                         \
      S(5, ub, wB, , )
#define LPT_TAP5(S)                                                                                               \
      LPT_XLOAD(0, 0, ua)                                                                                         \
      S(0, ua, wA, LPT_XLOAD(0, 1, ub), LPT_WLD0(wB, t0 + 1 * tstep))                                             \
      S(1, ub, wA, LPT_XLOAD(0, 2, ua), LPT_WLD(wB, 1))                                                           \
      S(2, ua, wA, LPT_XLOAD(0, 3, ub), LPT_WLD(wB, 2))                                                           \
      S(3, ub, wA, LPT_XLOAD(0, 4, ua), LPT_WLD(wB, 3))                                                           \
      S(4, ua, wA, LPT_XLOAD(1, 0, ub), )                                                                         \
      S(0, ub, wB, LPT_XLOAD(1, 1, ua), LPT_WLD0(wA, t0 + 2 * tstep))                                             \
      S(1, ua, wB, LPT_XLOAD(1, 2, ub), LPT_WLD(wA, 1))                                                           \
      S(2, ub, wB, LPT_XLOAD(1, 3, ua), LPT_WLD(wA, 2))                                                           \
      S(3, ua, wB, LPT_XLOAD(1, 4, ub), LPT_WLD(wA, 3))                                                           \
      S(4, ub, wB, LPT_XLOAD(2, 0, ua), )                                                                         \
      S(0, ua, wA, LPT_XLOAD(2, 1, ub), LPT_WLD0(wB, t0 + 3 * tstep))                                             \
      S(1, ub, wA, LPT_XLOAD(2, 2, ua), LPT_WLD(wB, 1))                                                           \
      S(2, ua, wA, LPT_XLOAD(2, 3, ub), LPT_WLD(wB, 2))                                                           \
      S(3, ub, wA, LPT_XLOAD(2, 4, ua), LPT_WLD(wB, 3))                                                           \
      S(4, ua, wA, LPT_XLOAD(3, 0, ub), )                                                                         \
      S(0, ub, wB, LPT_XLOAD(3, 1, ua), LPT_WLD0(wA, t1))                                                         \
      S(1, ua, wB, LPT_XLOAD(3, 2, ub), LPT_WLD(wA, 1))                                                           \
      S(2, ub, wB, LPT_XLOAD(3, 3, ua), LPT_WLD(wA, 2))                                                           \
      S(3, ua, wB, LPT_XLOAD(3, 4, ub), LPT_WLD(wA, 3))                                                           \
      S(4, ub, wB, , )
#define LPT_TAP4(S)                                                                                               \
      LPT_XLOAD(0, 0, ua)                                                                                         \
      S(0, ua, wA, LPT_XLOAD(0, 1, ub), LPT_WLD0(wB, t0 + 1 * tstep))                                             \
      S(1, ub, wA, LPT_XLOAD(0, 2, ua), LPT_WLD(wB, 1))                                                           \
      S(2, ua, wA, LPT_XLOAD(0, 3, ub), LPT_WLD(wB, 2))                                                           \
      S(3, ub, wA, LPT_XLOAD(1, 0, ua), LPT_WLD(wB, 3))                                                           \
      S(0, ua, wB, LPT_XLOAD(1, 1, ub), LPT_WLD0(wA, t0 + 2 * tstep))                                             \
      S(1, ub, wB, LPT_XLOAD(1, 2, ua), LPT_WLD(wA, 1))                                                           \
      S(2, ua, wB, LPT_XLOAD(1, 3, ub), LPT_WLD(wA, 2))                                                           \
      S(3, ub, wB, LPT_XLOAD(2, 0, ua), LPT_WLD(wA, 3))                                                           \
      S(0, ua, wA, LPT_XLOAD(2, 1, ub), LPT_WLD0(wB, t0 + 3 * tstep))                                             \
      S(1, ub, wA, LPT_XLOAD(2, 2, ua), LPT_WLD(wB, 1))                                                           \
      S(2, ua, wA, LPT_XLOAD(2, 3, ub), LPT_WLD(wB, 2))                                                           \
      S(3, ub, wA, LPT_XLOAD(3, 0, ua), LPT_WLD(wB, 3))                                                           \
      S(0, ua, wB, LPT_XLOAD(3, 1, ub), LPT_WLD0(wA, t1))                                                         \
      S(1, ub, wB, LPT_XLOAD(3, 2, ua), LPT_WLD(wA, 1))                                                           \
      S(2, ua, wB, LPT_XLOAD(3, 3, ub), LPT_WLD(wA, 2))                                                           \
      S(3, ub, wB, , LPT_WLD(wA, 3))
// END generated by tools/gen_lpt_taps.py

// RG = row groups = waves per SIMD (1, 2 or 3): wave w = (column group w & 3, row group w >> 2) owns the row tiles
// rg, rg + RG, rg + 2 RG, ... The body is compiled once per row group (MYRG is the wave's own, the kernel below branches on
// it once): a row group's tile count is then a constant and its code a region of its own. As ONE body with run-time
// "if (rg == 0) 7 tiles else 6 tiles" blocks in the tap loop, times the all-live / partly-live variants of a tap, the
// register allocator gave the 14 accumulators different registers in different blocks (52 v_mov_b64 per tap to reconcile
// them) and spilled a quarter of the residual stream around the loop.
// IL (round 5): several whole sequences of L <= 104 positions per tile, INTERLEAVED position-major (tile row = position * il + q is
// position `row / il` of the tile's sequence q): a dilated tap is then a row offset of il * (t - 4) * dil that leaves the tile
// exactly when the position leaves the sequence, i.e. the tile behaves like ONE sequence of il * L rows with every dilation
// multiplied by il — and the short sequences run on this kernel (clamped addressing, tile-granular liveness) instead of the
// round-2 kernel with a position test per fragment row.
template <typename T, int NP, int RG, int MYRG, bool IL>
__device__ __forceinline__ void backbone_lp_t_body(const BackboneLpArgs& a, char* smem_b) {
  constexpr int NTH = 256 * RG;
  typedef typename Lp<T>::V8 V8;
  constexpr int NPARTS = NP == 3 ? 2 : 1;
  char* plane = smem_b + LPSB;                               // byte address of (row 0, channel 0) of the hi plane
  float* img32 = reinterpret_cast<float*>(smem_b);           // final stage: fp32 image [TW_ROWS][BB_AP] over the planes
  float* Bs = reinterpret_cast<float*>(smem_b + IMG_REGION_B);   // [9][5][128] the first layer's lookup table
  float* psum = Bs + 9 * 5 * BB_C;                           // [8][TW_ROWS]: first / second moment partials of the 4 column groups
  float* rstat = psum + 8 * TW_ROWS;                         // [TW_ROWS]
  float* rmean = rstat + TW_ROWS;                            // [TW_ROWS] row means (this layer's shift = last layer's mean)
  int* toks = reinterpret_cast<int*>(rmean + TW_ROWS);       // [TW_ROWS]
  int* sdil = toks + 2 * TW_ROWS;                            // [BB_MAXL + 1]   (same map as the kernel above)
  int* sched = sdil + BB_MAXL + 1;                           // [(nl + 1) * 36]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = w & 3;
  int rg = MYRG;                                             // opaque: as a constant it lets the scheduler hoist every row tile's loads of a phase at once (spills)
  asm volatile("" : "+s"(rg));
  const int j = lane & 15, g = lane >> 4;
  const int cb = 32 * cg + 8 * g;                            // this lane's channels cb .. cb + 7: (ct, e) <-> cb + 2 e + ct
  const int L = a.L;
  const int nvalid = a.count ? __builtin_amdgcn_readfirstlane(*a.count) : a.n;
  int seq0 = blockIdx.x, il = 1;                             // this tile's first (compact) sequence and how many it interleaves
  if (IL) {
    if ((int)blockIdx.x >= nvalid) return;                   // a tile takes at least one sequence: no plan needed to know this one is empty
    SvddTilePlan pl = a.plan;
    if (a.auto_spt) pl = svdd_plan_tiles(nvalid, L, a.ncu, a.fixed_half);
    svdd_plan_tile(pl, (int)blockIdx.x, seq0, il);
    seq0 = __builtin_amdgcn_readfirstlane(seq0);
    il = __builtin_amdgcn_readfirstlane(il);
  }
  if (seq0 >= nvalid) return;
  const int R = il * L;                                      // rows of the tile
  const int nl = a.nl;
  const int it_end = (nl + 1) * 9;                           // (layer, tap) entries
  constexpr int NR = (TW_RT - MYRG + RG - 1) / RG;           // row tiles of this row group: MYRG, MYRG + RG, ...

  for (int e = tid; e < TW_ROWS; e += NTH) {
    int tk = -1;
    const int q = IL ? e % il : 0, pos = IL ? e / il : e;
    if (e < R && seq0 + q < nvalid) {
      const int64_t sq = a.row_idx ? (int64_t)a.row_idx[seq0 + q] : (int64_t)(seq0 + q);
      tk = (int)a.x[sq * L + pos];
    }
    toks[e] = tk;
    rmean[e] = 0.0f;
  }
  for (int e = tid; e < LPSB / 4; e += NTH) {                // zero rows -1 and TW_ROWS of both planes
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
  for (int e = tid; e < 9 * 5 * BB_C; e += NTH) Bs[e] = a.table0[e];
  __syncthreads();
  // schedule, one entry per (layer, tap): 0 for a tap that only sees zero padding, else
  //   bits 0-12 live row tiles ; 15-18 tap ; 19-28 index of the next live entry. The four 32-channel chunks of a tap share it.
  for (int k = tid; k < it_end; k += NTH) {
    auto entry = [&](int kk) {
      const int layer = kk / 9, t = kk % 9;
      if (layer >= nl) return t == 4 ? 0x1fff : 0;
      const int d = (t - 4) * sdil[layer] * il;              // in tile rows
      const int lo = d < 0 ? -d : 0, hi = d > 0 ? R - d : R;
      if (lo >= hi) return 0;
      int m = 0;
      for (int r = 0; r < TW_RT; ++r) if (lo < 16 * r + 16 && hi > 16 * r) m |= 1 << r;
      return m;
    };
    const int m = entry(k);
    int nx = k + 1;
    while (nx < it_end && entry(nx) == 0) ++nx;
    sched[k] = m ? (m | (k % 9) << 15 | nx << 19) : 0;
  }

  // ---- first layer: table lookup, fp32 (dnaconv.py:177,184); position 16 (rh + 2 r) + j, channels cb .. cb + 7
  f32x4 f[NR][2], acc[NR][2];
  {
    const f32x4 b0a = *reinterpret_cast<const f32x4*>(a.vec + cb), b0b = *reinterpret_cast<const f32x4*>(a.vec + cb + 4);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int p = 16 * (rg + RG * r) + j;
      f32x4 va = b0a, vb = b0b;
      if (rg + RG * r < TW_RT) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int q = p + (t - 4) * il;                  // the row of position + t - 4 of the same sequence
          const int tk = (q >= 0 && q < R) ? toks[q < 0 ? 0 : (q < TW_ROWS ? q : TW_ROWS - 1)] : -1;
          const float* row = Bs + (t * 5 + (tk < 0 ? 0 : tk)) * BB_C + cb;
          const f32x4 ta = *reinterpret_cast<const f32x4*>(row), tb = *reinterpret_cast<const f32x4*>(row + 4);
          const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
          va += tk >= 0 ? ta : z;
          vb += tk >= 0 ? tb : z;
        }
      }
      const bool in = p < R;
      f[r][0] = f32x4{in ? fmaxf(va[0], 0.0f) : 0.0f, in ? fmaxf(va[2], 0.0f) : 0.0f, in ? fmaxf(vb[0], 0.0f) : 0.0f, in ? fmaxf(vb[2], 0.0f) : 0.0f};
      f[r][1] = f32x4{in ? fmaxf(va[1], 0.0f) : 0.0f, in ? fmaxf(va[3], 0.0f) : 0.0f, in ? fmaxf(vb[1], 0.0f) : 0.0f, in ? fmaxf(vb[3], 0.0f) : 0.0f};
    }
  }
  __syncthreads();                                        // sched is visible

  // activation-fragment addressing (bytes inside a plane): this lane feeds position 16 (rh + 2 r) + j, channels 32 c + 8 g .. + 8
  const int arow0 = (16 * rg + j) * LPSB + 16 * g;
  const int a_lo = arow0 - (16 * rg + j + 1) * LPSB;      // row -1
  const int a_hi = arow0 + (TW_ROWS - 16 * rg - j) * LPSB;    // row TW_ROWS

  constexpr int TILE_V8 = 4 * 64 * 2 * NPARTS;
  const V8* wsrc = reinterpret_cast<const V8*>(a.tiles) + (cg * 64 + lane) * (2 * NPARTS);
  const char* wbase = reinterpret_cast<const char*>(a.tiles);
  const unsigned woff = (unsigned)((cg * 64 + lane) * (2 * NPARTS) * sizeof(V8));
  // weight tile of (entry k = 9 layer + tap, chunk c): the host's order is [layer][chunk][tap], then the 4 chunks of the 1x1 conv
  auto tile_of = [&](int k, int c) { const int ly = k / 9; return ly < nl ? ly * 36 + c * 9 + (k - 9 * ly) : nl * 36 + c; };
  int it = 0;
  while (it < it_end && sched[it] == 0) ++it;
  it = __builtin_amdgcn_readfirstlane(it);
  int en = __builtin_amdgcn_readfirstlane(sched[it]);
  V8 wA[2 * NPARTS], wB[2 * NPARTS];                       // the two weight-fragment sets: chunks alternate between them
  {
    const V8* src = wsrc + (size_t)tile_of(it, 0) * TILE_V8;
#pragma unroll
    for (int q = 0; q < 2 * NPARTS; ++q) wA[q] = src[q];
  }
  // per-channel vectors of this lane in channel order -> (ct, e) order
  auto chan8 = [&](const float* v, int cbx, f32x4& c0, f32x4& c1, float scale) {
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(v + cbx) * scale, hi4 = *reinterpret_cast<const f32x4*>(v + cbx + 4) * scale;
    c0 = f32x4{lo4[0], lo4[2], hi4[0], hi4[2]};
    c1 = f32x4{lo4[1], lo4[3], hi4[1], hi4[3]};
  };
  auto store_image = [&](int wrow0, int r, f32x4 v0, f32x4 v1) {   // (ct, e) values of position 16 (rh + 2 r) + j -> both planes
    const f32x8 v = {v0[0], v1[0], v0[1], v1[1], v0[2], v1[2], v0[3], v1[3]};
    const V8 hi = __builtin_convertvector(v, V8);
    *reinterpret_cast<V8*>(plane + wrow0 + r * (16 * RG * LPSB)) = hi;
    if constexpr (NP == 3) {
      const V8 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8), V8);
      *reinterpret_cast<V8*>(plane + PLANE_B + wrow0 + r * (16 * RG * LPSB)) = lo;
    }
  };

  for (int layer = 0; layer <= nl; ++layer) {             // layer == nl: the first 1x1 conv of final_conv
    const float* vl = a.vec + (size_t)(layer + 1) * 4 * BB_C;
    const float sa = a.lscale[2 * layer], inv = a.lscale[2 * layer + 1];
    // The phase code below addresses LDS by (row tile, lane): 7 tiles x 5 arrays of loop-invariant addresses, which the
    // compiler otherwise hoists out of the layer loop and then SPILLS across the MFMA loop. Re-deriving them from lane ids
    // it cannot see through costs a few VALU ops per layer and keeps the MFMA loop's registers free.
    int jo = j, cbo = cb;
    asm volatile("" : "+v"(jo), "+v"(cbo));
    const int wrow = (16 * rg + jo) * LPSB + 2 * cbo;
    if (layer < nl) {
      f32x4 tb0, tb1;
      chan8(vl + BB_C, cbo, tb0, tb1, 1.0f);
      // ONE statistics pass about the per-row shift K = the row's mean one layer earlier (see the kernel above)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int p = 16 * (rg + RG * r) + jo;
        const float K = rmean[p < TW_ROWS ? p : TW_ROWS - 1];
        const f32x4 d0 = f[r][0] + tb0 - K, d1 = f[r][1] + tb1 - K;
        const f32x4 s = d0 + d1, q = d0 * d0 + d1 * d1;
        float s1 = (s[0] + s[1]) + (s[2] + s[3]), s2 = (q[0] + q[1]) + (q[2] + q[3]);
        s1 = xor_sum<16>(s1); s2 = xor_sum<16>(s2);       // + the lanes g ^ 1, then g ^ 2: the position's 32 channels of this wave
        s1 = xor_sum<32>(s1); s2 = xor_sum<32>(s2);
        if (cbo == 32 * cg && p < TW_ROWS) { psum[cg * TW_ROWS + p] = s1; psum[(4 + cg) * TW_ROWS + p] = s2; }
        if (r & 1) __builtin_amdgcn_sched_barrier(0);       // two tiles' worth of loads in flight, not seven (VGPR pressure)
      }
      __syncthreads();
      if (tid < TW_ROWS) {
        const float m = ((psum[tid] + psum[TW_ROWS + tid]) + (psum[2 * TW_ROWS + tid] + psum[3 * TW_ROWS + tid])) * (1.0f / BB_C);
        const float q = ((psum[4 * TW_ROWS + tid] + psum[5 * TW_ROWS + tid]) + (psum[6 * TW_ROWS + tid] + psum[7 * TW_ROWS + tid])) * (1.0f / BB_C);
        rmean[tid] += m;
        rstat[tid] = rsqrtf(fmaxf(q - m * m, 0.0f) + 1e-5f);
      }
      __syncthreads();
      f32x4 gm0, gm1, bt0, bt1;
      chan8(vl + 2 * BB_C, cbo, gm0, gm1, sa);              // sa is a power of two: exact
      chan8(vl + 3 * BB_C, cbo, bt0, bt1, sa);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int p = 16 * (rg + RG * r) + jo;
        if (rg + RG * r < TW_RT) {
          const float rs = rstat[p], mean = rmean[p];
          const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
          const f32x4 v0 = p < R ? (f[r][0] + tb0 - mean) * rs * gm0 + bt0 : z;
          const f32x4 v1 = p < R ? (f[r][1] + tb1 - mean) * rs * gm1 + bt1 : z;
          store_image(wrow, r, v0, v1);
        }
        if (r & 1) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r)
        if (rg + RG * r < TW_RT) store_image(wrow, r, f[r][0] * sa, f[r][1] * sa);
    }
    // ---- implicit GEMM over (chunk, live tap): acc^T[channel, position] += W[channel, k] X^T[k, position]
#pragma unroll
    for (int r = 0; r < NR; ++r) { acc[r][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[r][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    const int dil = __builtin_amdgcn_readfirstlane(sdil[layer < nl ? layer : BB_MAXL]) * il;   // in tile rows
    const int layer_end = (layer + 1) * 9;
    __syncthreads();                                      // the image is complete
    while (it < layer_end) {                              // one live tap: 4 chunks x the wave's row tiles
      const int nxt = en >> 19;
      const int en_next_v = sched[nxt < it_end ? nxt : it];
      const int delta = (((en >> 15) & 15) - 4) * dil;
      const int live = en >> rg;                          // bit RG r = owned tile r
      int xa[NR];                                         // clamped fragment address of every owned tile, chunk 0
#pragma unroll
      for (int r = 0; r < NR; ++r) xa[r] = min(max(arow0 + delta * LPSB + r * (16 * RG * LPSB), a_lo), a_hi);
      // weight tiles of this tap: chunk c is tile t0 + c tstep; t1 = chunk 0 of the next live tap (after the last one: this tap's,
      // requested and never used — a request behind a branch makes the compiler's counter model merge "requested" with "not
      // requested", and it then waits for vmcnt(0), the request just made, in every step of the chunk)
      const int t0 = tile_of(it, 0), tstep = it < nl * 9 ? 9 : 1, t1 = tile_of(nxt < it_end ? nxt : it, 0);
      const char* wn_;
      // the chunk's 64-byte step is an immediate offset of the LDS read
#define LPT_XLOAD(C, R, V)                                                                                   \
      { V[0] = *reinterpret_cast<const V8*>(plane + xa[R] + 64 * (C));                                       \
        if constexpr (NP == 3) V[1] = *reinterpret_cast<const V8*>(plane + PLANE_B + xa[R] + 64 * (C)); }
      // the next chunk's weight tile, one 16-byte piece per step into the idle set (no wait here: the compiler counts them)
      // (uniform tile base + the lane's 32-bit byte offset: one address VGPR per request instead of a 64-bit pair built per tile)
#define LPT_WLD0(WN, TILE) { wn_ = wbase + (size_t)(TILE) * (TILE_V8 * sizeof(V8)); WN[0] = *reinterpret_cast<const V8*>(wn_ + woff); }
#define LPT_WLD(WN, Q) { if constexpr ((Q) < 2 * NPARTS) WN[Q] = *reinterpret_cast<const V8*>(wn_ + woff + 16 * (Q)); }
      // One step = the 6 (2) MFMAs of one (chunk, row tile) on fragments requested during the step before it. What is not an
      // MFMA sits BETWEEN the MFMAs, where it issues while the matrix pipe is busy: the request for the next step's fragments
      // after the first, the piece of the next weight tile after the third. A wave alone on its SIMD then issues an MFMA every
      // 16.6 cycles (tools/ubench/mfma_step_stream.hip); behind the group the same instructions cost it a third of the pipe,
      // and the SIMD's other wave cannot make that up: MFMA issue goes to the OLDER wave whenever it is ready.
      // W: [ct][part] -> W[ct * NPARTS + part]; pass order as in the kernel above: hi hi, w_lo x_hi, w_hi x_lo
      // LPT_STEP_ALL: every row tile of the tap is live (all of them in the dilation 1 / 4 layers, the inner taps elsewhere).
#define LPT_STEP_ALL(R, U, W, LOADNEXT, EXTRA)                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      __builtin_amdgcn_s_waitcnt(0xC07F);                   /* lgkmcnt(0): this step's fragments */          \
      acc[R][0] = Lp<T>::mfma(W[0], U[0], acc[R][0]);                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      LOADNEXT                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      acc[R][1] = Lp<T>::mfma(W[NPARTS], U[0], acc[R][1]);                                                   \
      if constexpr (NP == 3) acc[R][0] = Lp<T>::mfma(W[1], U[0], acc[R][0]);                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      EXTRA                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      if constexpr (NP == 3) {                                                                               \
        acc[R][1] = Lp<T>::mfma(W[NPARTS + 1], U[0], acc[R][1]);                                             \
        acc[R][0] = Lp<T>::mfma(W[0], U[1], acc[R][0]);                                                      \
        acc[R][1] = Lp<T>::mfma(W[NPARTS], U[1], acc[R][1]);                                                 \
      }                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);
      // LPT_STEP_LIVE: for taps that leave two or more of the wave's row tiles in the zero padding (dilation 16 / 64). A dead
      // step requests the next step's fragments and nothing else — no wait (a wait without MFMAs to hide it is a whole LDS round
      // trip: the version that waited in every step spent as long in a dead step as in a live one) and ONE scalar test, on an
      // SGPR copy of the mask that the compiler cannot see through (else it hoists the tests out of the chunk loop as lane
      // masks and re-materialises each through a VGPR). The requests go out BEFORE the MFMAs here, so a live step waits with
      // the next step's two (one) reads still outstanding.
#define LPT_STEP_LIVE(R, U, W, LOADNEXT, EXTRA)                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      LOADNEXT                                                                                               \
      EXTRA                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      { int lv_ = live;                                                                                      \
        asm volatile("" : "+s"(lv_));                                                                        \
        if (lv_ & (1 << (RG * (R)))) {                                                                       \
          if constexpr (sizeof(#LOADNEXT) > 1) __builtin_amdgcn_s_waitcnt(0xC07F | (NPARTS << 8));           \
          else __builtin_amdgcn_s_waitcnt(0xC07F);                                                           \
          acc[R][0] = Lp<T>::mfma(W[0], U[0], acc[R][0]);                                                    \
          acc[R][1] = Lp<T>::mfma(W[NPARTS], U[0], acc[R][1]);                                               \
          if constexpr (NP == 3) {                                                                           \
            acc[R][0] = Lp<T>::mfma(W[1], U[0], acc[R][0]);                                                  \
            acc[R][1] = Lp<T>::mfma(W[NPARTS + 1], U[0], acc[R][1]);                                         \
            acc[R][0] = Lp<T>::mfma(W[0], U[1], acc[R][0]);                                                  \
            acc[R][1] = Lp<T>::mfma(W[NPARTS], U[1], acc[R][1]);                                             \
          }                                                                                                  \
        } }                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);
      // a tap with at most one dead tile takes the all-live path: the dead tile's fragments are the zero rows, its MFMAs add
      // +0 to accumulators that are never -0 — the same bits, and six MFMAs are cheaper than a test in each of the tap's steps
      // (a tap that is dead for ALL of this wave's tiles — the outermost taps of the dilation-64 layers for the odd row group —
      // only requests the next tap's first weight tile: the weight stream of those layers is what bounds them, 1 KB per wave
      // instruction through the vector memory pipe, twice per column group because both row groups need every tile)
#define LPT_TAP(N)                                                                                           \
      { constexpr int all_ = lpt_own_mask(N, RG);                                                            \
        if (__builtin_popcount(all_ & ~live) <= 1) { LPT_TAP##N(LPT_STEP_ALL) }                              \
        else if ((live & all_) != 0) { LPT_TAP##N(LPT_STEP_LIVE) }                                           \
        else { LPT_WLD0(wA, t1) LPT_WLD(wA, 1) LPT_WLD(wA, 2) LPT_WLD(wA, 3) } }
      V8 ua[2], ub[2];                                    // activation fragments, one (chunk, row tile) step ahead of the MFMAs
      static_assert(TW_RT == 13, "the generated listings are for 13 row tiles");
      if constexpr (NR == 13) LPT_TAP(13)                 // one wave per SIMD: all 13 row tiles
      if constexpr (NR == 7) LPT_TAP(7)
      if constexpr (NR == 6) LPT_TAP(6)
      if constexpr (NR == 5) LPT_TAP(5)
      if constexpr (NR == 4) LPT_TAP(4)
#undef LPT_TAP
#undef LPT_STEP_LIVE
#undef LPT_STEP_ALL
#undef LPT_WLD
#undef LPT_WLD0
#undef LPT_XLOAD
      it = nxt;
      en = __builtin_amdgcn_readfirstlane(en_next_v);
    }
    // No barrier here between conv layers: what follows (epilogue, the next layer's statistics) touches registers and psum
    // only; the image is not written before the two barriers of the LayerNorm, which every wave reaches after its loop. The
    // row group that wins the matrix pipe (the older wave of each SIMD) leaves the loop ~30 % earlier than its partner and used
    // to wait here; now its VALU phases run under the partner's remaining MFMAs.
    // The last conv layer and the 1x1 stage keep it: the 1x1 stage has no LayerNorm, its image write follows this layer's epilogue
    // directly, and the fp32 image after it overwrites the planes (tests/test_lp_gpu.py: with three row groups per SIMD the
    // early writer's rows are read by the late waves' dilation-64 taps; with two, a shift by 64 rows keeps the tile parity and
    // the race happens to be invisible for this dilation list).
    if (layer >= nl - 1) __syncthreads();
    f32x4 bl0, bl1;
    chan8(vl, cbo, bl0, bl1, 1.0f);
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (layer < nl) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {                      // relu(conv + b) + f
        f[r][0] = __builtin_elementwise_max(acc[r][0] * inv + bl0, z4) + f[r][0];
        f[r][1] = __builtin_elementwise_max(acc[r][1] * inv + bl1, z4) + f[r][1];
      }
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int p = 16 * (rg + RG * r) + jo;
        if (rg + RG * r < TW_RT) {                         // relu(W1 f + b1), fp32 image over the (dead) planes
          const f32x4 o0 = __builtin_elementwise_max(acc[r][0] * inv + bl0, z4), o1 = __builtin_elementwise_max(acc[r][1] * inv + bl1, z4);
          *reinterpret_cast<f32x4*>(img32 + p * BB_AP + cbo) = f32x4{o0[0], o1[0], o0[1], o1[1]};
          *reinterpret_cast<f32x4*>(img32 + p * BB_AP + cbo + 4) = f32x4{o0[2], o1[2], o0[3], o1[3]};
        }
      }
    }
  }
  __syncthreads();
  // ---- last 1x1 conv 128 -> 5 in fp32
  for (int e = tid; e < R * 5; e += NTH) {
    const int row = e / 5, v = e - 5 * row;
    const int q = IL ? row % il : 0, pos = IL ? row / il : row;   // tile row -> (sequence seq0 + q, position)
    if (seq0 + q >= nvalid) continue;
    const float* hr = img32 + row * BB_AP;
    const float* wv = a.w2 + v * BB_C;
    float sm = a.w2[5 * BB_C + v];
#pragma unroll 8
    for (int k = 0; k < BB_C; ++k) sm += hr[k] * wv[k];
    const int64_t sq = (a.row_idx && a.out_scatter) ? (int64_t)a.row_idx[seq0 + q] : (int64_t)(seq0 + q);
    a.out[(sq * L + pos) * 5 + v] = sm;
  }
}

template <typename T, int NP, int RG, bool IL = false>
__global__ __launch_bounds__(256 * RG, RG) void backbone_lp_t_kernel(BackboneLpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  const int rg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);     // wave w = (column group w & 3, row group w >> 2)
  // every row group executes the same sequence of workgroup barriers (the bodies differ in tile counts only)
  if constexpr (RG == 1) backbone_lp_t_body<T, NP, RG, 0, IL>(a, smem_b);
  if constexpr (RG == 2) { if (rg == 0) backbone_lp_t_body<T, NP, RG, 0, IL>(a, smem_b); else backbone_lp_t_body<T, NP, RG, 1, IL>(a, smem_b); }
  if constexpr (RG == 3) {
    if (rg == 0) backbone_lp_t_body<T, NP, RG, 0, IL>(a, smem_b);
    else if (rg == 1) backbone_lp_t_body<T, NP, RG, 1, IL>(a, smem_b);
    else backbone_lp_t_body<T, NP, RG, 2, IL>(a, smem_b);
  }
}

}  // namespace

extern "C" int svdd_backbone_cnn_lp(const uint8_t* x, const float* table0, const void* tiles, const float* vec,
                                    const float* lscale, const float* w2, float* out, int n, int L, int nlayers,
                                    const int* dilations, int prec, const int32_t* count, const int32_t* row_idx,
                                    int out_scatter, void* stream) {
  if (!x || !table0 || !tiles || !vec || !lscale || !w2 || !out || !dilations || n <= 0 || L <= 0 || L > TW_ROWS ||
      nlayers <= 0 || nlayers > BB_MAXL || prec < SVDD_PREC_F16X3 || prec > SVDD_PREC_BF16)
    return SVDD_E_ARG;
  BackboneLpArgs a;
  a.x = x; a.table0 = table0; a.tiles = tiles; a.vec = vec; a.lscale = lscale; a.w2 = w2; a.out = out;
  a.n = n; a.L = L; a.spt = TW_ROWS / L; a.nl = nlayers; a.count = count; a.row_idx = row_idx; a.out_scatter = out_scatter;
  for (int i = 0; i < nlayers; ++i) if (dilations[i] <= 0) return SVDD_E_ARG;
  for (int i = 0; i < BB_MAXL; ++i) a.dil[i] = i < nlayers ? dilations[i] : 1;
  const size_t lds = (size_t)IMG_REGION_B + sizeof(float) * (9 * 5 * (size_t)BB_C + 8 * (size_t)TW_ROWS + 4 * (size_t)TW_ROWS +
                                                              BB_MAXL + 1 + (size_t)(nlayers + 1) * 36);
  hipEvent_t e0, e1;
  svdd_internal_timed_events(6, &e0, &e1);
  a.auto_spt = 0; a.ncu = svdd_internal_num_cus();
  a.fixed_half = (prec == SVDD_PREC_F16X3 || prec == SVDD_PREC_BF16X3) ? 58 : 90;   // svdd_spt.h: calibrated on the interleaved kernel
  unsigned nwg = (unsigned)((n + a.spt - 1) / a.spt);
  a.plan = SvddTilePlan{a.spt, 0, a.spt};
  const int fixed = svdd_internal_fixed_spt();
  // Short sequences (several fit a tile) run INTERLEAVED on the transposed-accumulator kernel (round 5; version 1 = the round-2
  // kernel with the sequences stacked, for A/B). Every packing gives a row the same bits (exact work-skipping needs that).
  const bool interleaved = TW_ROWS / L > 1 && g_bb_lp_version != 1;
  if (a.spt > 1 && fixed <= 0) {                         // several sequences fit a tile: which tile takes how many (svdd_spt.h)
    if (fixed < 0) { a.spt = -fixed < a.spt ? -fixed : a.spt; a.plan = SvddTilePlan{a.spt, 0, a.spt}; nwg = (unsigned)((n + a.spt - 1) / a.spt); }
    else if (count) { a.auto_spt = 1; nwg = (unsigned)n; }   // decided on the device from *count; grid for one sequence per tile
    else if (interleaved) { a.plan = svdd_plan_tiles(n, L, a.ncu, a.fixed_half); a.spt = a.plan.s2; nwg = (unsigned)svdd_plan_num_tiles(a.plan, n); }
    else { a.spt = svdd_choose_spt(n, L, a.ncu, a.fixed_half); nwg = (unsigned)((n + a.spt - 1) / a.spt); }
  }
  const dim3 grid(nwg);
  const bool spt1 = a.spt == 1 && !a.auto_spt && (a.plan.n1 == 0 || a.plan.s1 == 1);
  // the transposed-accumulator kernel (round 3) where one sequence per tile is the ONLY packing (104 < L <= 208)
  const bool transposed = spt1 && TW_ROWS / L == 1 && g_bb_lp_version != 1;
#define LPT_LAUNCH_ONE(TT, NPP, RGG)                                                                                \
  { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_lp_t_kernel<TT, NPP, RGG>),                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
    hipExtLaunchKernelGGL((backbone_lp_t_kernel<TT, NPP, RGG>), grid, dim3(256 * RGG), lds, (hipStream_t)stream, e0, e1, 0, a); }
#define LPT_LAUNCH_RG(TT, NPP)                                                                                      \
  if (g_bb_lp_rg == 3) LPT_LAUNCH_ONE(TT, NPP, 3) else if (g_bb_lp_rg == 1) LPT_LAUNCH_ONE(TT, NPP, 1) else LPT_LAUNCH_ONE(TT, NPP, 2)
#define LP_LAUNCH(TT, NPP)                                                                                          \
  do {                                                                                                               \
    if (transposed) {                                                                                                \
      LPT_LAUNCH_RG(TT, NPP)                                                                                         \
    } else if (interleaved) {                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_lp_t_kernel<TT, NPP, 2, true>),               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
      hipExtLaunchKernelGGL((backbone_lp_t_kernel<TT, NPP, 2, true>), grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a); \
    } else if (spt1) {                                                                                               \
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

