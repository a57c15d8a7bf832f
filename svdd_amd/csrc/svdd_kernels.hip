// svdd_kernels.hip — gfx950 (MI355X, CDNA4) kernels + C ABI for the SVDD decode hot path.
//
// Implements include/svdd_hip.h. One translation unit, built with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
// (-ffp-contract=off: every fp32 operation of the reference must round exactly once; the
// arithmetic contract is stated in DESIGN.md and mirrored by the CPU oracle).
//
// Kernels (all HBM/latency-bound integer/byte work; no MFMA here by design):
//   K1 propose_kernel    SUBS log-probs -> q_xs -> M exponential-race draws -> tokens + one-hot
//   K2 select_kernel     per-sample softmax over M soft values, argmax / multinomial, row gather
//   K3 x0hat_kernel      Tweedie posterior-mean candidate one-hot (transposed)
//   K5 finalize_kernel   noise-removal argmax
//   K6 transform_kernel  tokens -> one-hot
//   K7 subs_logp_kernel  SUBS re-parameterisation alone
//   K4 tds_resample_kernel  SMC/TDS resampling (baseline)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "svdd_hip.h"

namespace {

constexpr int V = SVDD_VOCAB;
constexpr int MASK = SVDD_MASK;
constexpr float NEG_INF_F = -1000000.0f;  // Diffusion.neg_infinity, diffusion_gosai.py:161
constexpr int WAVE = 64;

// ------------------------------------------------------------------ arithmetic contract ----
// exp/log correctly rounded to fp32: evaluate in fp64 (ocml, <=1 ulp in double) and round once.
__device__ __forceinline__ float expf_cr(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float logf_cr(float x) { return (float)log((double)x); }

__device__ __forceinline__ int64_t at(int layout, int64_t b, int64_t l, int v, int64_t L) {
  return layout == SVDD_LAYOUT_BLV ? (b * L + l) * V + v : (b * V + v) * L + l;
}

// Diffusion._subs_parameterization for one position (diffusion_gosai.py:286-304).
__device__ __forceinline__ void subs_logp_1(const float (&z)[V], int xt, float (&lp)[V]) {
  float zz[V];
#pragma unroll
  for (int v = 0; v < V; ++v) zz[v] = z[v];
  zz[MASK] = zz[MASK] + NEG_INF_F;
  float mx = zz[0];
#pragma unroll
  for (int v = 1; v < V; ++v) mx = zz[v] > mx ? zz[v] : mx;
  if (isinf(mx)) mx = 0.0f;
  float s = expf_cr(zz[0] - mx);
#pragma unroll
  for (int v = 1; v < V; ++v) s = s + expf_cr(zz[v] - mx);
  const float lse = logf_cr(s) + mx;
#pragma unroll
  for (int v = 0; v < V; ++v) lp[v] = zz[v] - lse;
  if (xt != MASK) {
#pragma unroll
    for (int v = 0; v < V; ++v) lp[v] = (v == xt) ? 0.0f : NEG_INF_F;
  }
}

// ------------------------------------------------------------------------------ Philox ----
// Philox4x32-10; counter layout documented in include/svdd_hip.h / DESIGN.md.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float u24(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void philox_uniform5(uint64_t seed, uint64_t pos, uint32_t step, uint32_t m,
                                                float (&u)[V]) {
  uint32_t c[4] = {(uint32_t)pos, (uint32_t)(pos >> 32), (step << 16) | m, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  u[0] = u24(c[0]); u[1] = u24(c[1]); u[2] = u24(c[2]); u[3] = u24(c[3]);
  uint32_t d[4] = {(uint32_t)pos, (uint32_t)(pos >> 32), (step << 16) | m, 1u};
  philox4x32_10(d, (uint32_t)seed, (uint32_t)(seed >> 32));
  u[4] = u24(d[0]);
}

// _sample_categorical for one position (diffusion_gosai.py:30-34): exponential race, first max wins.
__device__ __forceinline__ int sample_categorical_1(const float (&q)[V], const float (&u)[V]) {
  int best = 0;
  float rbest = 0.0f;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const float a = u[v] + 1e-10f;
    const float g = 1e-10f - logf_cr(a);
    const float r = __fdiv_rn(q[v], g);
    if (v == 0 || r > rbest) { rbest = r; best = v; }
  }
  return best;
}

// ------------------------------------------------------------------------- K1 propose ----
// Block = 256 threads = 4 waves over one tile of 64 consecutive (b,l) positions; wave 0 builds
// q_xs for the tile into LDS (once, not per candidate), then the 4 waves split the M candidates.
// Stores: cand 1 B/lane (64 B per wave-store), one-hot float4/lane (1 KiB per wave-store).
struct ProposeArgs {
  const float* logits; const uint8_t* x; float dm, mcs; int B, L, M, layout;
  int rng_kind; uint32_t step; const float* uniforms; uint64_t seed, row_offset;
  uint8_t* cand; float* onehot; float* q_xs;
};

__global__ __launch_bounds__(256) void propose_kernel(ProposeArgs a) {
  __shared__ float sq[V][WAVE];
  __shared__ int sx[WAVE];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t N = (int64_t)a.B * a.L;
  const int64_t n = (int64_t)blockIdx.x * WAVE + lane;
  const bool valid = n < N;
  const int64_t b = valid ? n / a.L : 0;
  const int64_t l = valid ? n - b * a.L : 0;

  if (wave == 0 && valid) {
    float z[V], lp[V], q[V];
#pragma unroll
    for (int v = 0; v < V; ++v) z[v] = a.logits[at(a.layout, b, l, v, a.L)];
    const int xt = a.x[n];
    if (xt != MASK && a.q_xs == nullptr) {
      // q is never read for an unmasked position (copy_flag wins); skip the transcendental work
#pragma unroll
      for (int v = 0; v < V; ++v) q[v] = 0.0f;
    } else {
      subs_logp_1(z, xt, lp);
#pragma unroll
      for (int v = 0; v < V; ++v) q[v] = expf_cr(lp[v]) * a.dm;   // diffusion_gosai.py:1194
      q[MASK] = a.mcs;                                             // :1196
      if (a.q_xs) {
#pragma unroll
        for (int v = 0; v < V; ++v) a.q_xs[at(a.layout, b, l, v, a.L)] = q[v];
      }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) sq[v][lane] = q[v];
    sx[lane] = xt;
  }
  __syncthreads();
  if (!valid) return;

  float q[V];
#pragma unroll
  for (int v = 0; v < V; ++v) q[v] = sq[v][lane];
  const int xt = sx[lane];
  const uint64_t pos = (a.row_offset + (uint64_t)b) * (uint64_t)a.L + (uint64_t)l;

  for (int m = wave; m < a.M; m += 4) {
    int c = xt;
    if (xt == MASK) {
      float u[V];
      if (a.rng_kind == SVDD_RNG_REPLAY) {
        const float* ub = a.uniforms + (int64_t)m * N * V;
#pragma unroll
        for (int v = 0; v < V; ++v) u[v] = ub[at(a.layout, b, l, v, a.L)];
      } else {
        philox_uniform5(a.seed, pos, a.step, (uint32_t)m, u);
      }
      c = sample_categorical_1(q, u);
    }
    const int64_t o = ((int64_t)b * a.M + m) * a.L + l;
    a.cand[o] = (uint8_t)c;
    float4 oh;
    oh.x = (c == 0) ? 1.0f : 0.0f; oh.y = (c == 1) ? 1.0f : 0.0f;
    oh.z = (c == 2) ? 1.0f : 0.0f; oh.w = (c == 3) ? 1.0f : 0.0f;
    reinterpret_cast<float4*>(a.onehot)[o] = oh;                   // transform_samples, :1462-1470
  }
}

// -------------------------------------------------------------------------- K2 select ----
// One wave per sample row. Lane j holds candidates m = j, j+64, ... (<= 16 chunks, M <= 1024).
// max / argmax use wave xor-shuffles (order independent, exact); the normaliser is summed in
// candidate order via readlane so that it is the same fp32 sequence as the oracle's loop.
constexpr int MAX_CHUNKS = SVDD_MAX_M / WAVE;

struct SelectArgs {
  const float* scores; const uint8_t* cand; int B, L, M, mode;
  uint32_t step; uint64_t seed, row_offset;
  uint8_t* x_next; float* soft; int32_t* idx;
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off, WAVE);
    v = o > v ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(256) void select_kernel(SelectArgs a) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B) return;
  const float* s = a.scores + (int64_t)row * a.M;
  const int nchunk = (a.M + WAVE - 1) / WAVE;

  float e[MAX_CHUNKS];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAX_CHUNKS; ++c) {
    const int m = c * WAVE + lane;
    e[c] = (c < nchunk && m < a.M) ? s[m] : -INFINITY;
    mx = e[c] > mx ? e[c] : mx;
  }
  mx = wave_max(mx);
#pragma unroll
  for (int c = 0; c < MAX_CHUNKS; ++c) {
    const int m = c * WAVE + lane;
    e[c] = (c < nchunk && m < a.M) ? expf_cr(e[c] - mx) : 0.0f;
  }
  // sum in candidate order: ((e0 + e1) + e2) + ...
  float sum = 0.0f;
#pragma unroll
  for (int c = 0; c < MAX_CHUNKS; ++c) {
    if (c < nchunk) {
      const int cnt = min(WAVE, a.M - c * WAVE);
      for (int j = 0; j < cnt; ++j) {
        const float ej = __shfl(e[c], j, WAVE);
        sum = (c == 0 && j == 0) ? ej : sum + ej;
      }
    }
  }
  const float r = __fdiv_rn(1.0f, sum);
  float p[MAX_CHUNKS];
#pragma unroll
  for (int c = 0; c < MAX_CHUNKS; ++c) p[c] = e[c] * r;   // ATen CPU softmax: e * (1/sum)

  int best = 0;
  if (a.mode == SVDD_SELECT_ARGMAX) {
    // first maximal index: reduce (value, index) pairs, larger value wins, ties -> smaller index
    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) {
      const int m = c * WAVE + lane;
      if (c < nchunk && m < a.M && p[c] > bv) { bv = p[c]; bi = m; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, WAVE);
      const int oi = __shfl_xor(bi, off, WAVE);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    best = bi;
  } else {
    // multinomial: u * total < inclusive prefix (candidate order), first hit
    uint32_t ctr[4] = {(uint32_t)(a.row_offset + row), (uint32_t)((a.row_offset + (uint64_t)row) >> 32),
                       (a.step << 16), 2u};
    philox4x32_10(ctr, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const float u = u24(ctr[0]);
    float tot = 0.0f;
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) {
      if (c < nchunk) {
        const int cnt = min(WAVE, a.M - c * WAVE);
        for (int j = 0; j < cnt; ++j) {
          const float pj = __shfl(p[c], j, WAVE);
          tot = (c == 0 && j == 0) ? pj : tot + pj;
        }
      }
    }
    const float thr = u * tot;
    float run = 0.0f;
    best = a.M - 1;
    bool found = false;
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) {
      if (c < nchunk && !found) {
        const int cnt = min(WAVE, a.M - c * WAVE);
        for (int j = 0; j < cnt; ++j) {
          const float pj = __shfl(p[c], j, WAVE);
          run = (c == 0 && j == 0) ? pj : run + pj;
          if (thr < run) { best = c * WAVE + j; found = true; break; }
        }
      }
    }
  }

  if (a.soft) {
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) {
      const int m = c * WAVE + lane;
      if (c < nchunk && m < a.M) a.soft[(int64_t)row * a.M + m] = p[c];
    }
  }
  if (a.idx && lane == 0) a.idx[row] = best;

  // gather the winning candidate row (index-gather compaction, diffusion_gosai.py:1226-1227)
  const uint8_t* src = a.cand + ((int64_t)row * a.M + best) * a.L;
  uint8_t* dst = a.x_next + (int64_t)row * a.L;
  if ((a.L & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(dst);
    for (int i = lane; i < (a.L >> 2); i += WAVE) d4[i] = s4[i];
  } else {
    for (int i = lane; i < a.L; i += WAVE) dst[i] = src[i];
  }
}

// ------------------------------------------------- K3 / K5 / K6 / K7: per-position kernels ----
struct PosArgs {
  const float* logits; const uint8_t* x; int R, L, layout;
  float* out_f; uint8_t* out_u8; int64_t* out_i64; int transposed;
};

__global__ __launch_bounds__(256) void x0hat_kernel(PosArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (int64_t)a.R * a.L) return;
  const int64_t r = n / a.L, l = n - r * a.L;
  const int xt = a.x[n];
  int c = xt;
  if (xt == MASK) {
    float z[V], lp[V];
#pragma unroll
    for (int v = 0; v < V; ++v) z[v] = a.logits[at(a.layout, r, l, v, a.L)];
    subs_logp_1(z, xt, lp);
    int best = 0;
#pragma unroll
    for (int v = 1; v < V; ++v) if (lp[v] > lp[best]) best = v;   // argmax(dim=2), :1416
    c = best;
  }
  if (a.out_u8) a.out_u8[n] = (uint8_t)c;
#pragma unroll
  for (int v = 0; v < 4; ++v) a.out_f[(r * 4 + v) * a.L + l] = (c == v) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void finalize_kernel(PosArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (int64_t)a.R * a.L) return;
  const int64_t r = n / a.L, l = n - r * a.L;
  const int xt = a.x[n];
  int c = xt;
  if (xt == MASK) {
    float z[V], lp[V];
#pragma unroll
    for (int v = 0; v < V; ++v) z[v] = a.logits[at(a.layout, r, l, v, a.L)];
    subs_logp_1(z, xt, lp);
    int best = 0;
#pragma unroll
    for (int v = 1; v < 4; ++v) if (lp[v] > lp[best]) best = v;   // logits[:,:,:-1].argmax(-1), :1060
    c = best;
  }
  if (a.out_i64) a.out_i64[n] = c;
  if (a.out_u8) a.out_u8[n] = (uint8_t)c;
}

__global__ __launch_bounds__(256) void transform_kernel(PosArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (int64_t)a.R * a.L) return;
  const int c = a.x[n];
  if (a.transposed) {
    const int64_t r = n / a.L, l = n - r * a.L;
#pragma unroll
    for (int v = 0; v < 4; ++v) a.out_f[(r * 4 + v) * a.L + l] = (c == v) ? 1.0f : 0.0f;
  } else {
    float4 oh;
    oh.x = (c == 0) ? 1.0f : 0.0f; oh.y = (c == 1) ? 1.0f : 0.0f;
    oh.z = (c == 2) ? 1.0f : 0.0f; oh.w = (c == 3) ? 1.0f : 0.0f;
    reinterpret_cast<float4*>(a.out_f)[n] = oh;
  }
}

__global__ __launch_bounds__(256) void subs_logp_kernel(PosArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (int64_t)a.R * a.L) return;
  const int64_t r = n / a.L, l = n - r * a.L;
  float z[V], lp[V];
#pragma unroll
  for (int v = 0; v < V; ++v) z[v] = a.logits[at(a.layout, r, l, v, a.L)];
  subs_logp_1(z, a.x[n], lp);
#pragma unroll
  for (int v = 0; v < V; ++v) a.out_f[at(a.layout, r, l, v, a.L)] = lp[v];
}

// -------------------------------------------------------------------- K4 TDS resample ----
// numpy's pairwise float32 sum (np.add.reduce), the order `ratio.sum()` uses at :1282.
__device__ float np_pairwise_sum_f32(const float* a, int64_t n) {
  // explicit stack instead of recursion: sizes halve, depth <= 40
  struct Frame { const float* p; int64_t n; };
  Frame stack[48];
  float vals[48];
  int sp = 0, vp = 0;
  // post-order evaluation with an operand stack: push (p,n); leaves produce values; internal
  // nodes are encoded by pushing a marker frame (p == nullptr) that adds the two top values.
  stack[sp++] = {a, n};
  while (sp > 0) {
    Frame f = stack[--sp];
    if (f.p == nullptr) {            // combine marker: left value is below right value
      const float right = vals[--vp];
      const float left = vals[--vp];
      vals[vp++] = left + right;
      continue;
    }
    if (f.n < 8) {
      float res = 0.0f;
      for (int64_t i = 0; i < f.n; ++i) res += f.p[i];
      vals[vp++] = res;
    } else if (f.n <= 128) {
      float r[8];
      for (int k = 0; k < 8; ++k) r[k] = f.p[k];
      int64_t i;
      for (i = 8; i < f.n - (f.n % 8); i += 8)
        for (int k = 0; k < 8; ++k) r[k] += f.p[i + k];
      float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      for (; i < f.n; ++i) res += f.p[i];
      vals[vp++] = res;
    } else {
      int64_t n2 = f.n / 2;
      n2 -= n2 % 8;
      stack[sp++] = {nullptr, 0};                 // evaluated last: left + right
      stack[sp++] = {f.p + n2, f.n - n2};         // right (evaluated second)
      stack[sp++] = {f.p, n2};                    // left (evaluated first)
    }
  }
  return vals[0];
}

struct TdsArgs {
  const float* num; const float* den; float alpha; const uint8_t* sample; const double* u;
  int B, L; uint8_t* x_next; int32_t* idx; double* work;
};

// Single workgroup (the resample is a cross-particle exchange over one shard's B particles):
// phase 1 parallel ratio, phase 2 one lane replays numpy's summation orders, phase 3 parallel
// searchsorted + one wave per particle row copy.
__global__ __launch_bounds__(1024) void tds_resample_kernel(TdsArgs a) {
  float* ratio = reinterpret_cast<float*>(a.work + a.B);   // work: [B] f64 cdf + [B] f32 ratio
  const float inv_alpha = (float)(1.0 / (double)a.alpha);
  for (int b = threadIdx.x; b < a.B; b += blockDim.x)
    ratio[b] = expf_cr(inv_alpha * (a.num[b] - a.den[b]));          // :1280
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = np_pairwise_sum_f32(ratio, a.B);               // ratio.sum()
    double c = 0.0;
    for (int b = 0; b < a.B; ++b) { c += (double)__fdiv_rn(ratio[b], tot); a.work[b] = c; }  // p.cumsum()
  }
  __syncthreads();
  const double last = a.work[a.B - 1];
  __syncthreads();
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) a.work[b] = a.work[b] / last;   // cdf /= cdf[-1]
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  for (int j = wave; j < a.B; j += nwave) {
    const double uj = a.u[j];
    int lo = 0, hi = a.B;                                            // searchsorted(side='right')
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.work[mid] <= uj) lo = mid + 1; else hi = mid; }
    const int k = lo < a.B ? lo : a.B - 1;
    if (a.idx && lane == 0) a.idx[j] = k;
    for (int i = lane; i < a.L; i += WAVE) a.x_next[(int64_t)j * a.L + i] = a.sample[(int64_t)k * a.L + i];
  }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH; }
inline bool bad_layout(int layout) { return layout != SVDD_LAYOUT_BLV && layout != SVDD_LAYOUT_BVL; }

}  // namespace

// ================================================================================ C ABI ====
extern "C" {

int svdd_abi_version(void) { return SVDD_ABI_VERSION; }

int svdd_device_info(char* arch, int arch_len, int* num_cu) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SVDD_E_NODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return SVDD_E_NODEVICE;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  if (num_cu) *num_cu = prop.multiProcessorCount;
  return SVDD_OK;
}

int svdd_propose(const float* logits, const uint8_t* x, float dm, float mcs, int B, int L, int M, int layout,
                 const svdd_rng_t* rng, uint8_t* cand, float* onehot, float* q_xs, void* stream) {
  if (!logits || !x || !rng || !cand || !onehot || B <= 0 || L <= 0 || M <= 0 || M > 65535 || bad_layout(layout))
    return SVDD_E_ARG;
  if (rng->kind == SVDD_RNG_REPLAY ? rng->uniforms == nullptr : rng->kind != SVDD_RNG_PHILOX) return SVDD_E_ARG;
  ProposeArgs a{logits, x, dm, mcs, B, L, M, layout, rng->kind, rng->step, rng->uniforms, rng->seed,
                rng->row_offset, cand, onehot, q_xs};
  const int64_t N = (int64_t)B * L;
  const unsigned grid = (unsigned)((N + WAVE - 1) / WAVE);
  hipLaunchKernelGGL(propose_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch();
}

int svdd_select(const float* scores, const uint8_t* cand, int B, int L, int M, int mode, const svdd_rng_t* rng,
                uint8_t* x_next, float* soft, int32_t* idx, void* stream) {
  if (!scores || !cand || !x_next || B <= 0 || L <= 0 || M <= 0 || M > SVDD_MAX_M) return SVDD_E_ARG;
  if (mode != SVDD_SELECT_ARGMAX && mode != SVDD_SELECT_MULTINOMIAL) return SVDD_E_ARG;
  if (mode == SVDD_SELECT_MULTINOMIAL && (!rng || rng->kind != SVDD_RNG_PHILOX)) return SVDD_E_ARG;
  SelectArgs a{scores, cand, B, L, M, mode, rng ? rng->step : 0u, rng ? rng->seed : 0ull,
               rng ? rng->row_offset : 0ull, x_next, soft, idx};
  hipLaunchKernelGGL(select_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch();
}

static int launch_pos(void (*k)(PosArgs), const PosArgs& a, void* stream) {
  const int64_t N = (int64_t)a.R * a.L;
  hipLaunchKernelGGL(k, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch();
}

int svdd_x0hat(const float* logits, const uint8_t* xt, int R, int L, int layout, float* onehot_t, uint8_t* x0hat,
               void* stream) {
  if (!logits || !xt || !onehot_t || R <= 0 || L <= 0 || bad_layout(layout)) return SVDD_E_ARG;
  return launch_pos(x0hat_kernel, PosArgs{logits, xt, R, L, layout, onehot_t, x0hat, nullptr, 1}, stream);
}

int svdd_finalize(const float* logits, const uint8_t* x, int B, int L, int layout, int64_t* out_i64,
                  uint8_t* out_u8, void* stream) {
  if (!logits || !x || (!out_i64 && !out_u8) || B <= 0 || L <= 0 || bad_layout(layout)) return SVDD_E_ARG;
  return launch_pos(finalize_kernel, PosArgs{logits, x, B, L, layout, nullptr, out_u8, out_i64, 0}, stream);
}

int svdd_transform_samples(const uint8_t* tok, int R, int L, int transposed, float* out, void* stream) {
  if (!tok || !out || R <= 0 || L <= 0) return SVDD_E_ARG;
  return launch_pos(transform_kernel, PosArgs{nullptr, tok, R, L, 0, out, nullptr, nullptr, transposed}, stream);
}

int svdd_subs_logp(const float* logits, const uint8_t* x, int B, int L, int layout, float* logp, void* stream) {
  if (!logits || !x || !logp || B <= 0 || L <= 0 || bad_layout(layout)) return SVDD_E_ARG;
  return launch_pos(subs_logp_kernel, PosArgs{logits, x, B, L, layout, logp, nullptr, nullptr, 0}, stream);
}

int svdd_tds_resample(const float* reward_num, const float* reward_den, float alpha, const uint8_t* sample,
                      const double* u, int B, int L, uint8_t* x_next, int32_t* idx, double* work, void* stream) {
  if (!reward_num || !reward_den || !sample || !u || !x_next || !work || B <= 0 || L <= 0 || !(alpha != 0.0f))
    return SVDD_E_ARG;
  TdsArgs a{reward_num, reward_den, alpha, sample, u, B, L, x_next, idx, work};
  hipLaunchKernelGGL(tds_resample_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  return check_launch();
}

}  // extern "C"
