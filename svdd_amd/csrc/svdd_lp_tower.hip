// svdd_lp_tower.hip — the value net's conv tower (whole sequences and candidate windows), split precision
// (split-precision net kernels on the 16-bit matrix cores: see svdd_lp_common.h for the arithmetic)
#include "svdd_lp_common.h"

namespace {

// ------------------------------------------------------------------------ conv tower, split precision ----
// conv_tower_kernel / conv_tower_win_kernel of svdd_nets.hip on the 16-bit matrix cores: one workgroup (8 waves) per tile
// of whole sequences (WIN: per candidate window); the activation image lives in LDS as two 16-bit planes (hi, lo) and is
// the A operand directly. What differs from the fp32 kernels (measured: profiles/r02_exp_ablations.txt, tower_lp):
//   * the input is the TOKEN row (u8), not the fp32 one-hot: the one-hot is built in LDS (exact in 16 bits, so the stem
//     needs only A * Bhi + A * Blo); one MFMA covers a whole (tap, 32-channel chunk);
//   * wave w owns 32 output channels (column pair w & 1) of the row tiles rq + 4 r, rq = w >> 1: an activation fragment feeds
//     two column tiles, which halves the LDS read traffic — the largest single cost once the MFMAs are 16x cheaper. The
//     accumulators are TRANSPOSED (round 4d: weights as the A operand, activations as B, like backbone_lp_t_kernel): with the
//     host's weight order (tile row m of column tile ct = channel 32 cp + 2 m + ct) a lane holds eight ADJACENT channels of
//     one position per row tile, and a layer's epilogue is one 16-byte LDS load and store per plane and tile instead of four
//     4-byte ones per row (same bits; windows 364 -> 338 us in f16x3, profiles/r04_bb_lpt_phase_timing.txt section 9);
//   * the number of live row tiles of a wave (4 / 3 for whole sequences, 0..4 for a window) is a TEMPLATE parameter of
//     the layer loop, chosen once per wave: straight-line code, no per-tile predicates (the first version, with runtime
//     predicates, ran the 66 %-live windows SLOWER than whole sequences);
//   * the residual is re-read from the 16-bit planes (hi + lo) instead of being kept in registers: <= 128 VGPRs, so two
//     workgroups share a CU and cover each other's barriers, prologue and output copy;
//   * the OUTPUT is written as the 16-bit planes themselves, [row][hi | lo][64]: its only consumer, the split-precision
//     GRU, feeds them to its MFMAs as they are;
//   * WIN: candidates can be addressed through a compacted index list whose length lives on the device
//     (live_idx / count: exact work-skipping without a host round trip); `count` == NULL means all n.
constexpr int TW_C = 64;
constexpr int TLSB = 160;                       // bytes per row of a 16-bit plane (64 channels + 32 B pad: conflict-free b128)
constexpr int TPLANE_B = (TW_ROWS + 2) * TLSB;  // rows -1 .. TW_ROWS
constexpr int TW_MAXL = 8;

struct TowerLpArgs {
  const uint8_t* tok;      // [n, L] tokens (0..3, 4 = MASK -> zero row)
  const void* tiles;       // [2 + 10*nlayers] tiles of [2 cp][64 lanes][2 ct][P][8] 16-bit
  const float* bias;       // [1 + nlayers][64]
  const float* inv;        // [1 + nlayers] 1 / weight scale of the stage
  void* out;               // [n, L, P, 64] 16-bit planes (hi, lo)   (WIN with live_idx: [count, ...] compact)
  int n, L, spt, nlayers, residual_mask;
  // WIN only
  const int* win;          // [n][2]
  const void* parent_out;  // [n / M, L, P, 64] as written by the non-WIN kernel
  int M;
  const int* live_idx;     // [count] candidate ids, or NULL (identity)
  const int* count;        // device scalar, or NULL (= n)
};

// Everything a wave needs inside the layer loop.
template <typename T>
struct TowerCtx {
  char* plane;             // (row 0, channel 0) of the hi plane
  const T* xs;             // one-hot rows, [row][4]
  const typename Lp<T>::V8* wsrc;
  const float* bias; const float* inv;
  int L, tile_rows, nlayers, residual_mask;
  int rq, cp, j, g;
  int apos[4];             // !SPT1: position of this lane's A row inside its sequence, per owned tile
};

// The stem + nlayers residual conv layers for a wave that owns NL live row tiles (rq + 4 r, r < NL).
template <typename T, int NP, bool CLAMP, int NL>
__device__ __forceinline__ void tower_layers(const TowerCtx<T>& c) {
  typedef typename Lp<T>::V8 V8;
  typedef T T4 __attribute__((ext_vector_type(4)));
  constexpr int NPARTS = NP == 3 ? 2 : 1;
  constexpr int TILE_V8 = 2 * 64 * 2 * NPARTS;
  const int j = c.j, g = c.g, L = c.L;
  const int arow0 = 16 * c.rq + j;                               // this lane feeds rows arow0 + 64 r as the A operand
  const int abase = arow0 * TLSB + 16 * g;                       // byte offset of (row arow0, channel 8 g)
  const int a_lo = abase - (arow0 + 1) * TLSB;                   // row -1
  const int a_hi = abase + (TW_ROWS - arow0) * TLSB;             // row TW_ROWS
  const int nit = 2 + 10 * c.nlayers;
  V8 bn[2 * NPARTS];
#pragma unroll
  for (int q = 0; q < 2 * NPARTS; ++q) bn[q] = c.wsrc[q];
  int it = 0;
  f32x4 acc[NL > 0 ? NL : 1][2];

  for (int layer = -1; layer < c.nlayers; ++layer) {
#pragma unroll
    for (int r = 0; r < NL; ++r) { acc[r][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[r][1] = acc[r][0]; }
    const int niter = layer < 0 ? 2 : 10;
    for (int ci = 0; ci < niter; ++ci, ++it) {
      V8 bc[2 * NPARTS];                                         // [ct][part]
#pragma unroll
      for (int q = 0; q < 2 * NPARTS; ++q) bc[q] = bn[q];
      if (it + 1 < nit) {
        const V8* src = c.wsrc + (size_t)(it + 1) * TILE_V8;
#pragma unroll
        for (int q = 0; q < 2 * NPARTS; ++q) bn[q] = src[q];
      }
      V8 ah[NL > 0 ? NL : 1], al[NL > 0 ? NL : 1];
      if (layer < 0) {
        // k = 32 ci + 8 g + e: taps t0 = 8 ci + 2 g and t0 + 1, 4 channels each = two adjacent one-hot rows
        const int t0 = 8 * ci + 2 * g;
#pragma unroll
        for (int r = 0; r < NL; ++r) {
          const int rr = arow0 + 64 * r + t0 - 7;
          T4 x0, x1;
          if (CLAMP) {
            x0 = *reinterpret_cast<const T4*>(c.xs + 4 * rr);
            x1 = *reinterpret_cast<const T4*>(c.xs + 4 * (rr + 1));
          } else {
            const int p = c.apos[r] + t0 - 7;
            x0 = *reinterpret_cast<const T4*>(c.xs + 4 * ((unsigned)p < (unsigned)L ? rr : TW_ROWS + 7));
            x1 = *reinterpret_cast<const T4*>(c.xs + 4 * ((unsigned)(p + 1) < (unsigned)L ? rr + 1 : TW_ROWS + 7));
          }
          ah[r][0] = x0[0]; ah[r][1] = x0[1]; ah[r][2] = x0[2]; ah[r][3] = x0[3];
          ah[r][4] = x1[0]; ah[r][5] = x1[1]; ah[r][6] = x1[2]; ah[r][7] = x1[3];
        }
#pragma unroll
        for (int r = 0; r < NL; ++r) {                           // the one-hot has no lo part
          acc[r][0] = Lp<T>::mfma(bc[0], ah[r], acc[r][0]);
          acc[r][1] = Lp<T>::mfma(bc[NPARTS], ah[r], acc[r][1]);
          if constexpr (NP == 3) {
            acc[r][0] = Lp<T>::mfma(bc[1], ah[r], acc[r][0]);
            acc[r][1] = Lp<T>::mfma(bc[NPARTS + 1], ah[r], acc[r][1]);
          }
        }
      } else {
        const int ch = ci / 5, delta = ci - 5 * ch - 2;
        const int dbytes = delta * TLSB + ch * 64;
#pragma unroll
        for (int r = 0; r < NL; ++r) {
          int o;
          if (CLAMP) o = min(max(abase + dbytes + r * (64 * TLSB), a_lo + ch * 64), a_hi + ch * 64);
          else o = (unsigned)(c.apos[r] + delta) < (unsigned)L ? abase + dbytes + r * (64 * TLSB) : a_hi + ch * 64;
          ah[r] = *reinterpret_cast<const V8*>(c.plane + o);
          if constexpr (NP == 3) al[r] = *reinterpret_cast<const V8*>(c.plane + TPLANE_B + o);
        }
#pragma unroll
        for (int r = 0; r < NL; ++r) {
          acc[r][0] = Lp<T>::mfma(bc[0], ah[r], acc[r][0]);
          acc[r][1] = Lp<T>::mfma(bc[NPARTS], ah[r], acc[r][1]);
          if constexpr (NP == 3) {
            acc[r][0] = Lp<T>::mfma(bc[1], ah[r], acc[r][0]);
            acc[r][1] = Lp<T>::mfma(bc[NPARTS + 1], ah[r], acc[r][1]);
            acc[r][0] = Lp<T>::mfma(bc[0], al[r], acc[r][0]);
            acc[r][1] = Lp<T>::mfma(bc[NPARTS], al[r], acc[r][1]);
          }
        }
      }
    }
    // every wave must be done reading the image before its owners overwrite it (the stem reads xs, not the image)
    if (layer >= 0) __syncthreads();
    const bool rs = layer >= 0 && ((c.residual_mask >> layer) & 1);
    // TRANSPOSED accumulators (round 4d; A = weight fragment, B = activation fragment — the same registers either way, the same
    // products in the same order, as in backbone_lp_t_kernel): lane (j, g) register e of column tile ct holds channel
    // cb + 2 e + ct, cb = 32 cp + 8 g, of ONE position per row tile, so a layer's epilogue reads the residual and writes the
    // new image as one 16-byte LDS access per plane and tile (position-major it was four 4-byte loads and stores each).
    const int cb = 32 * c.cp + 8 * g;
    const f32x4 b_lo = *reinterpret_cast<const f32x4*>(c.bias + (layer + 1) * TW_C + cb);
    const f32x4 b_hi = *reinterpret_cast<const f32x4*>(c.bias + (layer + 1) * TW_C + cb + 4);
    const float bl[8] = {b_lo[0], b_lo[1], b_lo[2], b_lo[3], b_hi[0], b_hi[1], b_hi[2], b_hi[3]};
    const float inv = c.inv[layer + 1];
#pragma unroll
    for (int r = 0; r < NL; ++r) {
      const int row = 16 * (c.rq + 4 * r) + j;
      char* dst = c.plane + row * TLSB + 2 * cb;
      float res[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (rs) {                                                  // residual = the layer's own input: hi + lo
        const V8 ph = *reinterpret_cast<const V8*>(dst);
#pragma unroll
        for (int k = 0; k < 8; ++k) res[k] = (float)ph[k];
        if constexpr (NP == 3) {
          const V8 pl = *reinterpret_cast<const V8*>(dst + TPLANE_B);
#pragma unroll
          for (int k = 0; k < 8; ++k) res[k] += (float)pl[k];
        }
      }
      V8 hi, lo;
#pragma unroll
      for (int k = 0; k < 8; ++k) {                              // channel cb + k = (ct, e) = (k & 1, k >> 1)
        const float v = row < c.tile_rows ? fmaxf(acc[r][k & 1][k >> 1] * inv + bl[k] + res[k], 0.0f) : 0.0f;
        const T h = (T)v;
        hi[k] = h; lo[k] = (T)(v - (float)h);
      }
      *reinterpret_cast<V8*>(dst) = hi;
      if constexpr (NP == 3) *reinterpret_cast<V8*>(dst + TPLANE_B) = lo;
    }
    __syncthreads();                                             // the image is complete
  }
}

template <typename T, int NP, bool SPT1, bool WIN>
__global__ __launch_bounds__(512, 4) void tower_lp_kernel(TowerLpArgs a) {
  typedef typename Lp<T>::V8 V8;
  constexpr int NPARTS = NP == 3 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  char* plane = smem_b + TLSB;                                   // (row 0, channel 0) of the hi plane
  T* xs = reinterpret_cast<T*>(smem_b + 2 * TPLANE_B) + 8 * 4;   // one-hot rows -8 .. TW_ROWS + 8, [row][4]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L;

  int cand = blockIdx.x;
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
  const int64_t total_rows = (int64_t)((!WIN && a.count) ? __builtin_amdgcn_readfirstlane(*a.count) : a.n) * L;
  if (!WIN && row0 >= total_rows) return;
  const int keep_lo = !WIN ? 0 : (nt == 0 ? 0 : (w0 == 0 ? 0 : w0 + 10));
  const int keep_hi = !WIN ? 0 : (nt == 0 ? 0 : (w1 >= L ? L : w1 - 10));
  constexpr int ROW16 = NPARTS * 8;                              // 16-byte pieces per output row
  uint4* outc = reinterpret_cast<uint4*>(a.out) + (WIN ? (size_t)blockIdx.x * L * ROW16 : 0);

  if (WIN) {
    const uint4* par = reinterpret_cast<const uint4*>(a.parent_out) + (size_t)(cand / a.M) * L * ROW16;
    for (int e = tid; e < L * ROW16; e += 512) {                 // rows that are the parent's
      const int row = e / ROW16;
      if (row < keep_lo || row >= keep_hi) outc[e] = par[e];
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

  TowerCtx<T> c;
  c.plane = plane; c.xs = xs; c.bias = a.bias; c.inv = a.inv;
  c.L = L; c.tile_rows = tile_rows; c.nlayers = a.nlayers; c.residual_mask = a.residual_mask;
  c.cp = w & 1; c.rq = w >> 1; c.j = lane & 15; c.g = lane >> 4;
  c.wsrc = reinterpret_cast<const V8*>(a.tiles) + (c.cp * 64 + lane) * (2 * NPARTS);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * (c.rq + 4 * r) + c.j;
    c.apos[r] = (!SPT1 && !WIN && row < tile_rows) ? row % L : -(1 << 20);
  }
  const int nlive = WIN ? (nt - c.rq + 3) >> 2 : (c.rq == 0 ? 4 : 3);    // owned live tiles rq + 4 r, r < nlive
  __syncthreads();
  constexpr bool CLAMP = SPT1 || WIN;                            // one sequence per image: a tap is a clamped row offset
  switch (nlive) {                                               // wave-uniform; every path runs the same barriers
    case 4: tower_layers<T, NP, CLAMP, 4>(c); break;
    case 3: tower_layers<T, NP, CLAMP, 3>(c); break;
    case 2: if constexpr (WIN) tower_layers<T, NP, CLAMP, 2>(c); break;
    case 1: if constexpr (WIN) tower_layers<T, NP, CLAMP, 1>(c); break;
    default: if constexpr (WIN) tower_layers<T, NP, CLAMP, 0>(c); break;
  }
  // the last image IS the output: rows [hi 128 B | lo 128 B], 16 bytes per thread
  if (WIN) {
    for (int e = tid; e < L * ROW16; e += 512) {
      const int row = e / ROW16, q = e - row * ROW16;
      if (row >= keep_lo && row < keep_hi)
        outc[e] = *reinterpret_cast<const uint4*>(plane + (q >> 3) * TPLANE_B + (row - w0) * TLSB + 16 * (q & 7));
    }
  } else {
    for (int e = tid; e < tile_rows * ROW16; e += 512) {
      const int row = e / ROW16, q = e - row * ROW16;
      if (row0 + row < total_rows)
        outc[(row0 + row) * ROW16 + q] = *reinterpret_cast<const uint4*>(plane + (q >> 3) * TPLANE_B + row * TLSB + 16 * (q & 7));
    }
  }
}

}  // namespace

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

extern "C" int svdd_conv_tower_lp(const uint8_t* tok, const void* tiles, const float* bias, const float* inv, void* out,
                                  int n, int L, int nlayers, int residual_mask, const int32_t* count, int prec, void* stream) {
  if (!tok || !tiles || !bias || !inv || !out || n <= 0 || L <= 0 || L > TW_ROWS || nlayers <= 0 || nlayers > TW_MAXL ||
      prec < SVDD_PREC_F16X3 || prec > SVDD_PREC_BF16)
    return SVDD_E_ARG;
  const int spt = TW_ROWS / L;
  TowerLpArgs a{tok, tiles, bias, inv, out, n, L, spt, nlayers, residual_mask, nullptr, nullptr, 1, nullptr, count};
  return launch_tower_lp(a, false, prec, (unsigned)((n + spt - 1) / spt), stream);
}

extern "C" int svdd_conv_tower_windows_lp(const uint8_t* cand, const void* tiles, const float* bias, const float* inv,
                                          const int32_t* win, const void* parent_out, void* out, int n, int L, int M,
                                          int nlayers, int residual_mask, const int32_t* live_idx, const int32_t* count,
                                          int prec, void* stream) {
  if (!cand || !tiles || !bias || !inv || !win || !parent_out || !out || n <= 0 || M <= 0 || n % M || L <= TW_ROWS / 2 ||
      L > TW_ROWS || nlayers != 5 || prec < SVDD_PREC_F16X3 || prec > SVDD_PREC_BF16)
    return SVDD_E_ARG;
  TowerLpArgs a{cand, tiles, bias, inv, out, n, L, 1, nlayers, residual_mask, win, parent_out, M, live_idx, count};
  return launch_tower_lp(a, true, prec, (unsigned)n, stream);
}

