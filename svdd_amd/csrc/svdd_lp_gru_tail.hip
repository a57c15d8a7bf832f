// svdd_lp_gru_tail.hip — the value net's bidirectional GRU and tail, split precision
// (split-precision net kernels on the 16-bit matrix cores: see svdd_lp_common.h for the arithmetic)
#include "svdd_lp_common.h"

namespace {

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
  const float* x;          // [n, L, 64] fp32, or NULL when x16 is given
  const void* x16;         // [n, L, P, 64] 16-bit planes (hi, lo) written by the split-precision tower, or NULL
  const void* wpack;       // [2 dirs][4 waves][64 lanes][6 mtx][2 chunks][P][8] 16-bit ; mtx order ir, hr, iz, hz, in, hn
  const float* bpack;      // [2][4][64]: b_ir + b_hr, b_iz + b_hz, b_in, b_hn
  const float* inv;        // [2] 1 / weight scale per direction
  float* out;              // [2][n, L, 64]
  int n, L;
  const int* count;
};

// Workgroup = 8 waves on one (tile of 16 sequences, direction): waves 0-3 are CONSUMERS and run the recurrence
// (read h_t and the input projections of step t from LDS -> 18 MFMAs -> gates -> write h_{t+1}), waves 4-7 are
// PRODUCERS and compute the input projections W_i* x of step t + 1 into an LDS ring one step ahead, from x rows they
// prefetched two further steps ahead. Consumer w and producer w + 4 share a SIMD and own the same 16 hidden units, so
// the producer's accumulator layout is exactly what the consumer needs (each lane reads back the 12 floats its
// partner lane wrote). A step is then only as long as the consumer's serial chain; in the first version one wave did
// both halves back to back and the chain was 2.4x longer (profiles/r02_exp_ablations.txt, gru_lp: 314 us with ONE workgroup per
// CU, i.e. latency- not throughput-bound; x loads and the x split alone were 40 % of it).
template <typename T, int NP>
__global__ __launch_bounds__(512) void gru_lp_kernel(GruLpArgs a) {
  typedef typename Lp<T>::V8 V8;
  typedef typename Lp<T>::V2 V2;
  constexpr int NPARTS = NP == 3 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char hbuf[2][2][16 * GLSB];          // [buffer][hi | lo][row][160 B]
  __shared__ __attribute__((aligned(16))) float xproj[2][4][3][64 * 4];        // [slot][wave][gate][lane x 4]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wv >= 4;
  const int w = wv & 3;
  const int dir = blockIdx.x & 1;                                 // unit u = (tile u >> 1, direction u & 1): live units of a
                                                                  // compacted batch form a dense prefix of the grid
  const int j = lane & 15, g = lane >> 4;
  const int n = a.count ? __builtin_amdgcn_readfirstlane(*a.count) : a.n;
  const int seq0 = (blockIdx.x >> 1) * 16;
  if (seq0 >= n) return;
  const int L = a.L;
  const int t0 = dir == 0 ? 0 : L - 1;
  const int dt = dir == 0 ? 1 : -1;

  // this wave's half of the gate weights: producers W_ir, W_iz, W_in (mtx 0, 2, 4), consumers W_hr, W_hz, W_hn (1, 3, 5)
  V8 wb[3][2][NPARTS];
  {
    const V8* wp = reinterpret_cast<const V8*>(a.wpack) + (((size_t)dir * 4 + w) * 64 + lane) * (12 * NPARTS);
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < NPARTS; ++q) wb[m][c][q] = wp[((2 * m + (producer ? 0 : 1)) * 2 + c) * NPARTS + q];
  }
  for (int i = threadIdx.x; i < 2 * 16 * GLSB / 4; i += 512) reinterpret_cast<int*>(&hbuf[0][0][0])[i] = 0;   // h_0 = 0
  float* myproj = &xproj[0][w][0][lane * 4];
  constexpr int SLOT = 4 * 3 * 64 * 4;                            // floats per ring slot

  if (producer) {
    const int arow = min(seq0 + j, n - 1);                        // clamped for the ragged last tile
    V8 xh[2][2], xl[2][2];                                        // two x rows in flight: [buffer][chunk]
    auto load_x = [&](int t, int bufi) {
      if (a.x16) {
        const V8* xp = reinterpret_cast<const V8*>(a.x16) + ((size_t)arow * L + t) * (NPARTS * 8) + g;
        xh[bufi][0] = xp[0]; xh[bufi][1] = xp[4];
        if constexpr (NP == 3) { xl[bufi][0] = xp[8]; xl[bufi][1] = xp[12]; }
      } else {
        const float4* xp = reinterpret_cast<const float4*>(a.x + ((size_t)arow * L + t) * 64 + 8 * g);
        const float4 p[4] = {xp[0], xp[1], xp[8], xp[9]};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float v[8] = {p[2 * c].x, p[2 * c].y, p[2 * c].z, p[2 * c].w, p[2 * c + 1].x, p[2 * c + 1].y, p[2 * c + 1].z, p[2 * c + 1].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) { const T h = (T)v[e]; xh[bufi][c][e] = h; xl[bufi][c][e] = (T)(v[e] - (float)h); }
        }
      }
    };
    auto project = [&](int bufi, int slot) {                      // W_i* x (scaled) of one step -> ring slot
      f32x4 pr = {0.0f, 0.0f, 0.0f, 0.0f}, pz = pr, pn = pr;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        pr = Lp<T>::mfma(xh[bufi][c], wb[0][c][0], pr);
        pz = Lp<T>::mfma(xh[bufi][c], wb[1][c][0], pz);
        pn = Lp<T>::mfma(xh[bufi][c], wb[2][c][0], pn);
        if constexpr (NP == 3) {
          pr = Lp<T>::mfma(xh[bufi][c], wb[0][c][1], pr);
          pz = Lp<T>::mfma(xh[bufi][c], wb[1][c][1], pz);
          pn = Lp<T>::mfma(xh[bufi][c], wb[2][c][1], pn);
          pr = Lp<T>::mfma(xl[bufi][c], wb[0][c][0], pr);
          pz = Lp<T>::mfma(xl[bufi][c], wb[1][c][0], pz);
          pn = Lp<T>::mfma(xl[bufi][c], wb[2][c][0], pn);
        }
      }
      float* dst = myproj + slot * SLOT;
      *reinterpret_cast<f32x4*>(dst) = pr;
      *reinterpret_cast<f32x4*>(dst + 256) = pz;
      *reinterpret_cast<f32x4*>(dst + 512) = pn;
    };
    // prologue: projections of step 0 in slot 0; x of steps 1 and 2 in flight
    load_x(t0, 0);
    if (L > 1) load_x(t0 + dt, 1);
    project(0, 0);
    if (L > 2) load_x(t0 + 2 * dt, 0);
    __syncthreads();
    // step s (consumers work on slot s & 1): project x_{s+1} into slot (s + 1) & 1, prefetch x_{s+3}
    for (int s = 0; s < L; s += 2) {
      if (s + 1 < L) project(1, 1);
      if (s + 3 < L) load_x(t0 + (s + 3) * dt, 1);
      __syncthreads();
      if (s + 1 >= L) break;
      if (s + 2 < L) project(0, 0);
      if (s + 4 < L) load_x(t0 + (s + 4) * dt, 0);
      __syncthreads();
    }
    return;
  }

  // ---- consumers
  const int u = 16 * w + j;
  const float b_r = a.bpack[(dir * 4 + 0) * 64 + u], b_z = a.bpack[(dir * 4 + 1) * 64 + u];
  const float b_nx = a.bpack[(dir * 4 + 2) * 64 + u], b_nh = a.bpack[(dir * 4 + 3) * 64 + u];
  const float inv = a.inv[dir];
  float hprev[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  __syncthreads();                                                // h_0 and slot 0 are ready
  for (int step = 0; step < L; ++step) {
    const int t = t0 + dt * step;
    const int cur = step & 1;
    V8 hh[2], hl[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      hh[c] = *reinterpret_cast<const V8*>(&hbuf[cur][0][j * GLSB + 64 * c + 16 * g]);
      if constexpr (NP == 3) hl[c] = *reinterpret_cast<const V8*>(&hbuf[cur][1][j * GLSB + 64 * c + 16 * g]);
    }
    const float* src = myproj + cur * SLOT;
    f32x4 acc_r = *reinterpret_cast<const f32x4*>(src);
    f32x4 acc_z = *reinterpret_cast<const f32x4*>(src + 256);
    const f32x4 acc_nx = *reinterpret_cast<const f32x4*>(src + 512);
    f32x4 acc_nh = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      acc_nh = Lp<T>::mfma(hh[c], wb[2][c][0], acc_nh);
      acc_r = Lp<T>::mfma(hh[c], wb[0][c][0], acc_r);
      acc_z = Lp<T>::mfma(hh[c], wb[1][c][0], acc_z);
      if constexpr (NP == 3) {
        acc_nh = Lp<T>::mfma(hh[c], wb[2][c][1], acc_nh);
        acc_r = Lp<T>::mfma(hh[c], wb[0][c][1], acc_r);
        acc_z = Lp<T>::mfma(hh[c], wb[1][c][1], acc_z);
        acc_nh = Lp<T>::mfma(hl[c], wb[2][c][0], acc_nh);
        acc_r = Lp<T>::mfma(hl[c], wb[0][c][0], acc_r);
        acc_z = Lp<T>::mfma(hl[c], wb[1][c][0], acc_z);
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
    }
    // 16-bit state for the next step first (it is what the other waves wait for), then the fp32 outputs
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
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      const int srow = 4 * g + rho;
      if (seq0 + srow < n) a.out[(((size_t)dir * a.n + seq0 + srow) * L + t) * 64 + u] = hn[rho];
    }
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

extern "C" int svdd_gru_bidir_lp(const float* x, const void* x16, const void* wpack, const float* bpack, const float* inv,
                                 float* out, int n, int L, const int32_t* count, int prec, void* stream) {
  if ((!x && !x16) || !wpack || !bpack || !inv || !out || n <= 0 || L <= 0 || prec < SVDD_PREC_F16X3 || prec > SVDD_PREC_BF16)
    return SVDD_E_ARG;
  GruLpArgs a{x16 ? nullptr : x, x16, wpack, bpack, inv, out, n, L, count};
  hipEvent_t e0, e1;
  svdd_internal_timed_events(3, &e0, &e1);
  const dim3 grid(2 * (unsigned)((n + 15) / 16));
  switch (prec) {
    case SVDD_PREC_F16X3: hipExtLaunchKernelGGL((gru_lp_kernel<_Float16, 3>), grid, dim3(512), 0, (hipStream_t)stream, e0, e1, 0, a); break;
    case SVDD_PREC_BF16X3: hipExtLaunchKernelGGL((gru_lp_kernel<__bf16, 3>), grid, dim3(512), 0, (hipStream_t)stream, e0, e1, 0, a); break;
    case SVDD_PREC_F16: hipExtLaunchKernelGGL((gru_lp_kernel<_Float16, 1>), grid, dim3(512), 0, (hipStream_t)stream, e0, e1, 0, a); break;
    default: hipExtLaunchKernelGGL((gru_lp_kernel<__bf16, 1>), grid, dim3(512), 0, (hipStream_t)stream, e0, e1, 0, a); break;
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

